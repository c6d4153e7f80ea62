// What do the wider / paired LDS stores and loads cost next to ds_write_b64 (6.14) / ds_read_b64 (2.24 cycles per wave-instruction and
// CU, lds_same_addr.hip)?  One workgroup of 1024 threads per CU, ITER trips of 32 instructions per wave.
//   0 ds_write_b64 lanes 8 B apart        1 ds_write_b128 lanes 16 B apart      2 ds_write_b128 lanes 48 B apart (six cells per node)
//   3 ds_write2_b64 adjacent cells, lanes 16 B apart     4 ds_write2_b64 cells 512 B apart, lanes 8 B apart     5 ds_write_b32 lanes 4 B apart
//   6 ds_write2_b32 adjacent words, lanes 8 B apart      7 ds_write_b96 lanes 16 B apart
//   10 ds_read_b64    11 ds_read_b128 lanes 16 B apart    12 ds_read_b128 lanes 48 B apart    13 ds_read2_b64 adjacent, lanes 16 B apart
//   14 ds_read2_b64 cells 512 B apart, lanes 8 B apart    15 ds_read_b32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int ITER = 1024;
typedef double d2 __attribute__((ext_vector_type(2)));
typedef float f3 __attribute__((ext_vector_type(3)));

#define REP8(S) S S S S S S S S
__global__ void __launch_bounds__(1024) k(int mode, long long *cycles, double *sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *lds = (double *)smem;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 8192; i += 1024) lds[i] = i;
    __syncthreads();
    // every wave its own 4 KB region (64 lanes x up to 48 B + a second cell 512 B up)
    uint32_t ad = wave * 4096;
    switch (mode) {
        case 0: case 4: case 6: case 10: case 14: ad += lane * 8; break;
        case 1: case 3: case 7: case 11: case 13: ad += lane * 16; break;
        case 2: case 12: ad += lane * 48; break;
        default: ad += lane * 4; break;
    }
    double v = threadIdx.x;
    d2 v2 = {v, v + 1};
    float f = threadIdx.x;
    f3 v3 = {f, f, f};
    double acc = 0;
    const long long t0 = clock64();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            switch (mode) {
                case 0: asm volatile(REP8("ds_write_b64 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : : "v"(ad), "v"(v) : "memory"); break;
                case 1: case 2: asm volatile(REP8("ds_write_b128 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : : "v"(ad), "v"(v2) : "memory"); break;
                case 3: asm volatile(REP8("ds_write2_b64 %0, %1, %2 offset0:0 offset1:1\n") "s_waitcnt lgkmcnt(0)\n" : : "v"(ad), "v"(v), "v"(acc) : "memory"); break;
                case 4: asm volatile(REP8("ds_write2_b64 %0, %1, %2 offset0:0 offset1:64\n") "s_waitcnt lgkmcnt(0)\n" : : "v"(ad), "v"(v), "v"(acc) : "memory"); break;
                case 5: asm volatile(REP8("ds_write_b32 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : : "v"(ad), "v"(f) : "memory"); break;
                case 6: asm volatile(REP8("ds_write2_b32 %0, %1, %2 offset0:0 offset1:1\n") "s_waitcnt lgkmcnt(0)\n" : : "v"(ad), "v"(f), "v"(f) : "memory"); break;
                case 7: asm volatile(REP8("ds_write_b96 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : : "v"(ad), "v"(v3) : "memory"); break;
                case 10: { double l0; asm volatile(REP8("ds_read_b64 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : "=&v"(l0) : "v"(ad) : "memory"); acc += l0; } break;
                case 11: case 12: { d2 l0; asm volatile(REP8("ds_read_b128 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : "=&v"(l0) : "v"(ad) : "memory"); acc += l0.x + l0.y; } break;
                case 13: { d2 l0; asm volatile(REP8("ds_read2_b64 %0, %1 offset0:0 offset1:1\n") "s_waitcnt lgkmcnt(0)\n" : "=&v"(l0) : "v"(ad) : "memory"); acc += l0.x + l0.y; } break;
                case 14: { d2 l0; asm volatile(REP8("ds_read2_b64 %0, %1 offset0:0 offset1:64\n") "s_waitcnt lgkmcnt(0)\n" : "=&v"(l0) : "v"(ad) : "memory"); acc += l0.x + l0.y; } break;
                default: { float l0; asm volatile(REP8("ds_read_b32 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : "=&v"(l0) : "v"(ad) : "memory"); acc += l0; } break;
            }
        }
    }
    const long long t1 = clock64();
    if (lane == 0) cycles[blockIdx.x * 16 + wave] = t1 - t0;
    if (acc == 12345.678) sink[0] = acc;
}

int main() {
    long long *d; double *sink;
    int cus = 0;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    CHECK(hipMalloc(&d, cus * 16 * sizeof(long long))); CHECK(hipMalloc(&sink, 8));
    CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    std::vector<long long> h(cus * 16);
    const int modes[] = {0, 1, 2, 3, 4, 5, 6, 7, 10, 11, 12, 13, 14, 15};
    const char *names[] = {"ds_write_b64", "ds_write_b128 (16 B apart)", "ds_write_b128 (48 B apart)", "ds_write2_b64 adjacent", "ds_write2_b64 512 B apart", "ds_write_b32",
                           "ds_write2_b32 adjacent", "ds_write_b96", "", "", "ds_read_b64", "ds_read_b128 (16 B apart)", "ds_read_b128 (48 B apart)", "ds_read2_b64 adjacent",
                           "ds_read2_b64 512 B apart", "ds_read_b32"};
    for (int mode : modes) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k, dim3(cus), dim3(1024), 65536, 0, mode, d, sink);
            CHECK(hipDeviceSynchronize());
        }
        CHECK(hipMemcpy(h.data(), d, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
        double mx = 0;
        for (auto c : h) mx = c > mx ? c : mx;
        printf("mode %2d %-28s: %.2f cycles per wave-instruction and CU (16 waves)\n", mode, names[mode], mx / (double)(ITER * 32) / 16.0);
    }
    return 0;
}
