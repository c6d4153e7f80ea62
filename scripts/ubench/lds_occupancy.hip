// How much dynamic LDS may a 256-thread workgroup use and still fit three (or two) to a CU?  (gfx950: 160 KB per CU)
#include <hip/hip_runtime.h>
#include <cstdio>
extern "C" __global__ void __launch_bounds__(256, 3) k(float *p) {
    extern __shared__ float sm[];
    sm[threadIdx.x] = p[threadIdx.x];
    __syncthreads();
    p[threadIdx.x] = sm[255 - threadIdx.x];
}
int main() {
    int prev = -1;
    for (int lds = 40000; lds <= 82000; lds += 128) {
        int nb = 0;
        hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, 256, lds);
        if (nb != prev) { printf("dynamic LDS %d bytes: %d workgroups per CU\n", lds, nb); prev = nb; }
    }
    return 0;
}
