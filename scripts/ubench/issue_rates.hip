// How many instructions per cycle does one SIMD issue, by mix and by waves per SIMD?  (gfx950)
// Each wave runs ITER trips of a 32-instruction block of independent instructions; one workgroup per CU, W waves per SIMD.
//   mix 0: 32 v_add_f64            mix 1: 16 v_add_f64 + 16 s_add_u32 (interleaved)      mix 2: 32 s_add_u32
//   mix 3: 24 v_add_f64 + 8 ds_read_b64 (one waitcnt per block)          mix 4: 32 v_add_u32 (32-bit VALU)
//   mix 5: 32 ds_read_b64 (waitcnt per 8)   mix 6: 32 ds_write_b64   mix 7: 16 ds_read_b64 + 16 ds_write_b64 (the BP passes' LDS mix)
//   mix 8: 32 ds_read_b128                  mix 9: what a "one record per check" check pass would issue for the same 8 edges as
//   mix 7: 8 + 8 ds_read_b64 (check pass reads; variable-node pass re-reads its own messages) + 8 ds_read_b128 (the checks' records:
//   two minima) + 8 ds_write_b64 (variable-node pass) + 1 ds_write_b128 (one record per ~8 edges) = 33 instructions
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int ITER = 4096;

template <int MIX>
__global__ void __launch_bounds__(1024) k(long long *cycles, double *sink) {
    __shared__ double lds[2048 + 512];
    double a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, c = 1e-9;
    uint32_t u0 = threadIdx.x, u1 = 1, u2 = 2, u3 = 3, u4 = 4, u5 = 5, u6 = 6, u7 = 7;
    uint32_t s0 = 0, s1 = 1, s2 = 2, s3 = 3;
    double l0 = 0, l1 = 0;
    lds[threadIdx.x] = threadIdx.x; lds[threadIdx.x + 1024] = 1.0;
    __syncthreads();
    const uint32_t la = (threadIdx.x & 63) * 8;
    const long long t0 = clock64();
    for (int it = 0; it < ITER; ++it) {
        if constexpr (MIX == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                             "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        } else if constexpr (MIX == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                asm volatile("v_add_f64 %0, %0, %8\n s_add_u32 %4, %4, 1\n v_add_f64 %1, %1, %8\n s_add_u32 %5, %5, 1\n"
                             "v_add_f64 %2, %2, %8\n s_add_u32 %6, %6, 1\n v_add_f64 %3, %3, %8\n s_add_u32 %7, %7, 1\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(c) : "scc");
        } else if constexpr (MIX == 2) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
                asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n"
                             : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        } else if constexpr (MIX == 3) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                asm volatile("ds_read_b64 %6, %8\n v_add_f64 %0, %0, %9\n v_add_f64 %1, %1, %9\n v_add_f64 %2, %2, %9\n"
                             "ds_read_b64 %7, %8 offset:8192\n v_add_f64 %3, %3, %9\n v_add_f64 %4, %4, %9\n v_add_f64 %5, %5, %9\n s_waitcnt lgkmcnt(0)\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "=&v"(l0), "=&v"(l1) : "v"(la), "v"(c));
            a6 += l0 + l1;
        } else if constexpr (MIX == 5) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                asm volatile("ds_read_b64 %0, %2\n ds_read_b64 %1, %2 offset:512\n ds_read_b64 %0, %2 offset:1024\n ds_read_b64 %1, %2 offset:1536\n"
                             "ds_read_b64 %0, %2 offset:2048\n ds_read_b64 %1, %2 offset:2560\n ds_read_b64 %0, %2 offset:3072\n ds_read_b64 %1, %2 offset:3584\n s_waitcnt lgkmcnt(0)\n"
                             : "=&v"(l0), "=&v"(l1) : "v"(la));
        } else if constexpr (MIX == 6) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                asm volatile("ds_write_b64 %0, %1\n ds_write_b64 %0, %1 offset:512\n ds_write_b64 %0, %1 offset:1024\n ds_write_b64 %0, %1 offset:1536\n"
                             "ds_write_b64 %0, %1 offset:2048\n ds_write_b64 %0, %1 offset:2560\n ds_write_b64 %0, %1 offset:3072\n ds_write_b64 %0, %1 offset:3584\n s_waitcnt lgkmcnt(0)\n"
                             : : "v"(la), "v"(a1) : "memory");
        } else if constexpr (MIX == 7) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                asm volatile("ds_read_b64 %0, %2\n ds_read_b64 %1, %2 offset:512\n ds_read_b64 %0, %2 offset:1024\n ds_read_b64 %1, %2 offset:1536\n s_waitcnt lgkmcnt(0)\n"
                             "ds_write_b64 %2, %3 offset:2048\n ds_write_b64 %2, %3 offset:2560\n ds_write_b64 %2, %3 offset:3072\n ds_write_b64 %2, %3 offset:3584\n s_waitcnt lgkmcnt(0)\n"
                             : "=&v"(l0), "=&v"(l1) : "v"(la), "v"(a1) : "memory");
        } else if constexpr (MIX == 8) {
            double q0, q1, q2, q3;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:1024\n ds_read_b128 %0, %2 offset:2048\n ds_read_b128 %1, %2 offset:3072\n"
                             "ds_read_b128 %0, %2 offset:4096\n ds_read_b128 %1, %2 offset:5120\n ds_read_b128 %0, %2 offset:6144\n ds_read_b128 %1, %2 offset:7168\n s_waitcnt lgkmcnt(0)\n"
                             : "=&v"(*(__attribute__((ext_vector_type(2))) double *)&q0), "=&v"(*(__attribute__((ext_vector_type(2))) double *)&q2) : "v"(la * 2));
            (void)q1; (void)q3;
        } else if constexpr (MIX == 9) {
            typedef double d2 __attribute__((ext_vector_type(2)));
            d2 q0, q1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                asm volatile("ds_read_b64 %0, %2\n ds_read_b64 %1, %2 offset:512\n s_waitcnt lgkmcnt(0)\n"
                             : "=&v"(l0), "=&v"(l1) : "v"(la));
                asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b64 %2, %5 offset:2048\n ds_read_b64 %3, %5 offset:2560\n s_waitcnt lgkmcnt(0)\n"
                             "ds_write_b64 %5, %6 offset:3072\n ds_write_b64 %5, %6 offset:3584\n s_waitcnt lgkmcnt(0)\n"
                             : "=&v"(q0), "=&v"(q1), "=&v"(l0), "=&v"(l1) : "v"(la * 2), "v"(la), "v"(a1) : "memory");
            }
            asm volatile("ds_write_b128 %0, %1 offset:8192\n s_waitcnt lgkmcnt(0)\n" : : "v"(la * 2), "v"(q0) : "memory");
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                             "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(u0));
        }
    }
    const long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (double)(u0 + u1 + u2 + u3 + u4 + u5 + u6 + u7 + s0 + s1 + s2 + s3);
}

template <int MIX>
int run(const char *name) {
    long long *cyc; double *sink;
    CHECK(hipMalloc(&cyc, 256 * 16 * 8)); CHECK(hipMalloc(&sink, 256 * 1024 * 8));
    for (int w = 1; w <= 4; ++w) {
        CHECK(hipMemset(cyc, 0, 256 * 16 * 8));
        hipLaunchKernelGGL(k<MIX>, dim3(256), dim3(256 * w), 0, 0, cyc, sink);
        CHECK(hipDeviceSynchronize());
        std::vector<long long> h(256 * 16);
        CHECK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
        long long mx = 0; for (long long v : h) mx = v > mx ? v : mx;
        // per SIMD: w waves x ITER x 32 instructions in mx cycles (clock64 = s_memtime, 100 MHz reference scaled? report both)
        printf("%-44s %d wave(s)/SIMD: %8lld ticks for %d instructions per wave -> %.3f instructions per tick per SIMD\n", name, w, mx, ITER * 32, (double)w * ITER * 32 / (double)mx);
    }
    return 0;
}
int main() {
    if (run<0>("32 v_add_f64")) return 1;
    if (run<4>("32 v_add_u32")) return 1;
    if (run<2>("32 s_add_u32")) return 1;
    if (run<1>("16 v_add_f64 + 16 s_add_u32 interleaved")) return 1;
    if (run<3>("24 v_add_f64 + 8 ds_read_b64 + waitcnt")) return 1;
    if (run<5>("32 ds_read_b64")) return 1;
    if (run<6>("32 ds_write_b64")) return 1;
    if (run<7>("message form, 8 edges: 16 rd b64 + 16 wr b64")) return 1;
    if (run<8>("32 ds_read_b128")) return 1;
    if (run<9>("record form, 8 edges: 16 rd b64 + 8 rd b128 + 8 wr b64 + 1 wr b128")) return 1;
    return 0;
}
