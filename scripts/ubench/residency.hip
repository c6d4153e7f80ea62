// How many 256-thread workgroups with a given dynamic LDS size are really resident at once?  (The occupancy API answers
// from sizes; this asks the device: every workgroup counts itself in and waits until all have arrived or 20 ms pass.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <int MODE> // 0: few registers; 1: 168 VGPRs (the budget of three waves per SIMD); 2: + 264 B of scratch per lane; 3: + 96 SGPRs
__global__ void __launch_bounds__(256, 3) k(unsigned *cnt, unsigned *seen, unsigned target) {
    extern __shared__ float sm[];
    sm[threadIdx.x] = 1.0f;
    if constexpr (MODE >= 1) asm volatile("v_mov_b32 v167, 0" ::: "v167");
    if constexpr (MODE >= 3) asm volatile("s_mov_b32 s95, 0" ::: "s95");
    if constexpr (MODE >= 2) {
        volatile int priv[66];
        for (int i = 0; i < 66; ++i) priv[i] = i + (int)threadIdx.x;
        sm[threadIdx.x] += (float)priv[(threadIdx.x * 7) % 66];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(cnt, 1u);
        const long long t0 = wall_clock64();
        unsigned v;
        while ((v = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < target && wall_clock64() - t0 < 2000000) __builtin_amdgcn_s_sleep(8);
        atomicMax(seen, v);
    }
    __syncthreads();
}
template <int MODE>
int run(unsigned *cnt, unsigned *seen, const char *what) {
    const int sizes[] = {40960, 54032, 54592, 55296};
    for (int lds : sizes) {
        int api = 0;
        CHECK(hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, k<MODE>, 256, lds));
        const unsigned grid = 256u * (unsigned)api;
        CHECK(hipMemset(cnt, 0, 4)); CHECK(hipMemset(seen, 0, 4));
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), lds, 0, cnt, seen, grid);
        CHECK(hipDeviceSynchronize());
        unsigned h = 0; CHECK(hipMemcpy(&h, seen, 4, hipMemcpyDeviceToHost));
        printf("%-34s dynamic LDS %6d B: occupancy API %d per CU -> grid %4u, resident together (first 20 ms): %4u%s\n", what, lds, api, grid, h, h < grid ? "   <-- fewer than the API says" : "");
    }
    return 0;
}
int main() {
    unsigned *cnt, *seen;
    CHECK(hipMalloc(&cnt, 4)); CHECK(hipMalloc(&seen, 4));
    if (run<0>(cnt, seen, "few registers")) return 1;
    if (run<1>(cnt, seen, "168 VGPRs")) return 1;
    if (run<2>(cnt, seen, "168 VGPRs + scratch")) return 1;
    if (run<3>(cnt, seen, "168 VGPRs + scratch + 96 SGPRs")) return 1;
    return 0;
}
