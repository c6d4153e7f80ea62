// Micro-benchmark: issue cost (cycles per wave-instruction, one wave per SIMD and 2 waves per SIMD)
// of the VALU/LDS operations the decoder's inner loops are made of.  hipcc --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP 64
#define ITER 200
template <int OP> __global__ void k(long long *out, double a0, double b0, int n) {
    __shared__ double lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i * 1.5;
    __syncthreads();
    double a = a0 + threadIdx.x, b = b0, c = 1.0, d = 2.0;
    uint64_t ua = (uint64_t)threadIdx.x * 7919u + 1, ub = 12345;
    uint32_t x = threadIdx.x, y = 3;
    int idx = threadIdx.x;
    long long t0 = clock64();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            if (OP == 0) { asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(b)); }
            if (OP == 1) { asm volatile("v_min_f64 %0, %0, %1" : "+v"(a) : "v"(b)); }
            if (OP == 2) { asm volatile("v_max_f64 %0, %0, %1" : "+v"(a) : "v"(b)); }
            if (OP == 3) { asm volatile("v_cmp_lt_f64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc" : : "v"(a), "v"(b), "v"(x), "v"(y) : "vcc"); }
            if (OP == 4) { asm volatile("v_cmp_lt_u64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc" : : "v"(ua), "v"(ub), "v"(x), "v"(y) : "vcc"); }
            if (OP == 5) { asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(y) : "vcc"); }
            if (OP == 6) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(y) : "vcc"); }
            if (OP == 7) { asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(b)); }
            if (OP == 8) { asm volatile("v_min_u32 %0, %0, %1" : "+v"(x) : "v"(y)); }
            if (OP == 9) { asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(y)); }
            if (OP == 10) { // dependent LDS read chain (latency)
                asm volatile("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)" : "+v"(idx)); idx = (idx & 1023) * 4; }
            if (OP == 11) { // independent LDS b64 reads (throughput)
                double t; asm volatile("ds_read_b64 %0, %1" : "=v"(t) : "v"((threadIdx.x & 511) * 8 + r * 8)); c += 0; asm volatile("" :: "v"(t)); }
            if (OP == 12) { asm volatile("ds_write_b64 %0, %1" :: "v"((threadIdx.x & 255) * 8 + (r & 7) * 2048), "v"(a)); }
            if (OP == 13) { asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c)); }
            if (OP == 14) { int t = __shfl_xor((int)x, 16, 64); x = t; }
            if (OP == 15) { asm volatile("v_max_f64 %0, %0, %0" : "+v"(a)); }
        }
        if (OP == 11 || OP == 12) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (a == 123.456 && x == 77 && ua == 3 && idx == -5 && c == 9 && d == 1) out[0] = 0; // keep values live
}
int main() {
    long long *d; hipMalloc(&d, 8 * 4096);
    const char *names[] = {"v_add_f64", "v_min_f64", "v_max_f64", "v_cmp_lt_f64+cndmask", "v_cmp_lt_u64+cndmask", "v_cmp_lt_u32+cndmask",
                           "v_cndmask_b32", "v_mul_f64", "v_min_u32", "v_and_b32", "ds_read_b32 dependent (latency)", "ds_read_b64 indep",
                           "ds_write_b64", "v_fma_f64", "__shfl_xor(dependent)", "v_max_f64 x,x (canonicalize)"};
    void (*ks[])(long long *, double, double, int) = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>, k<7>, k<8>, k<9>, k<10>, k<11>, k<12>, k<13>, k<14>, k<15>};
    for (int threads : {64, 256, 512}) {
        printf("== %d threads per block, 1 block per CU (waves/SIMD = %g)\n", threads, threads / 256.0);
        for (int op = 0; op < 16; ++op) {
            hipLaunchKernelGGL(ks[op], dim3(256), dim3(threads), 0, 0, d, 1.0, 2.0, 0);
            hipDeviceSynchronize();
            long long h[256]; hipMemcpy(h, d, 8 * 256, hipMemcpyDeviceToHost);
            double avg = 0; for (int i = 0; i < 256; ++i) avg += h[i]; avg /= 256;
            printf("  %-34s %7.2f cycles per wave-instruction (wave 0 view)\n", names[op], avg / (REP * ITER));
        }
    }
    return 0;
}
