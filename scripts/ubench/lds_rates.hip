// Micro-benchmark: LDS b64 read / write throughput for the access patterns of the decoder's BP passes
// (linear = jagged-diagonal CN side, random = VN side, same = dead positions), 4 / 8 / 16 waves per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP 32
#define ITER 100
template <int OP> __global__ void k(long long *out, int pattern) {
    extern __shared__ double lds[];
    __shared__ unsigned long long tmin, tmax;
    for (int i = threadIdx.x; i < 6144; i += blockDim.x) lds[i] = i;
    if (threadIdx.x == 0) { tmin = ~0ull; tmax = 0; }
    __syncthreads();
    uint32_t addr[REP];
    for (int r = 0; r < REP; ++r) {
        uint32_t slot;
        if (pattern == 0) slot = (threadIdx.x + r * 160) % 6144;                                  // linear
        else if (pattern == 1) slot = ((threadIdx.x * 2654435761u + r * 40503u) >> 7) % 6144;    // random
        else if (pattern == 2) slot = 17 + (threadIdx.x >> 6);                                    // one address per wave
        else slot = ((threadIdx.x & 63) * 17 + (threadIdx.x >> 6) * 1090 + r * 131) % 6144;      // stride 17 (odd): conflict-free, scattered
        addr[r] = slot * 8;
    }
    double acc = 0;
    long long t0 = clock64();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            if (OP == 0) { double t; asm volatile("ds_read_b64 %0, %1" : "=v"(t) : "v"(addr[r])); asm volatile("" :: "v"(t)); }
            if (OP == 1) { asm volatile("ds_write_b64 %0, %1" :: "v"(addr[r]), "v"(acc)); }
        }
        asm volatile("s_waitcnt lgkmcnt(0)");
    }
    long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) { atomicMin(&tmin, (unsigned long long)t0); atomicMax(&tmax, (unsigned long long)t1); }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (long long)(tmax - tmin);
    if (acc == 123.0) out[0] = 0;
}
int main() {
    long long *d; (void)hipMalloc(&d, 8 * 4096);
    const char *pn[] = {"linear", "random", "same address per wave", "stride 17"};
    for (int threads : {256, 512, 1024}) {
        printf("== %d threads per block (waves per CU = %d), cycles per wave-instruction = block time / (REP*ITER) ; per-CU cycles per instr = that / waves\n", threads, threads / 64);
        for (int op = 0; op < 2; ++op)
            for (int pat = 0; pat < 4; ++pat) {
                if (op == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 6144 * 8, 0, d, pat);
                else hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 6144 * 8, 0, d, pat);
                (void)hipDeviceSynchronize();
                long long h[256]; (void)hipMemcpy(h, d, 8 * 256, hipMemcpyDeviceToHost);
                double avg = 0; for (int i = 0; i < 256; ++i) avg += h[i]; avg /= 256;
                const double per = avg / (REP * ITER);
                printf("  %-12s %-24s %7.2f per wave, %6.2f LDS-pipe cycles per wave-instruction\n", op ? "ds_write_b64" : "ds_read_b64", pn[pat], per, per / (threads / 64));
            }
    }
    return 0;
}
