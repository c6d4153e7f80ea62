// What does an LDS store cost when many lanes of the wave write the SAME address (the BP passes' dead positions all point at the wave's
// zero / far slot)?  One workgroup of 1024 threads per CU, ITER trips of 32 stores (or loads) per wave; cycles per wave-instruction and CU.
//   mode 0: every lane its own consecutive 8-byte cell      mode 1: every lane the same cell
//   mode 2: lanes 0-15 their own cells, lanes 16-63 one shared cell       mode 3: as 2, the 48 shared lanes switched off (exec mask)
//   mode 4: as 2 with random own cells (gather-like)        modes 10-14: the same address patterns with ds_read_b64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int ITER = 2048;

__global__ void __launch_bounds__(1024) k(int mode, long long *cycles, double *sink) {
    __shared__ double lds[4096];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += 1024) lds[i] = i;
    __syncthreads();
    const int m = mode % 10;
    uint32_t cell;
    bool on = true;
    if (m == 0) cell = wave * 64 + lane;
    else if (m == 1) cell = wave * 64;
    else if (m == 2 || m == 3) { cell = lane < 16 ? wave * 64 + lane : wave * 64 + 63; on = m == 2 || lane < 16; }
    else cell = lane < 16 ? (uint32_t)((wave * 64 + lane * 37 + 11) % 1024) : wave * 64 + 63 + 1024;
    const uint32_t ad = cell * 8;
    double v = threadIdx.x, acc = 0;
    const long long t0 = clock64();
    if (on) {
        for (int it = 0; it < ITER; ++it) {
            if (mode < 10) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    asm volatile("ds_write_b64 %0, %1\n ds_write_b64 %0, %1 offset:8192\n ds_write_b64 %0, %1 offset:16384\n ds_write_b64 %0, %1 offset:24576\n"
                                 "ds_write_b64 %0, %1\n ds_write_b64 %0, %1 offset:8192\n ds_write_b64 %0, %1 offset:16384\n ds_write_b64 %0, %1 offset:24576\n s_waitcnt lgkmcnt(0)\n"
                                 : : "v"(ad), "v"(v) : "memory");
            } else {
                double l0, l1;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    asm volatile("ds_read_b64 %0, %2\n ds_read_b64 %1, %2 offset:8192\n ds_read_b64 %0, %2 offset:16384\n ds_read_b64 %1, %2 offset:24576\n"
                                 "ds_read_b64 %0, %2\n ds_read_b64 %1, %2 offset:8192\n ds_read_b64 %0, %2 offset:16384\n ds_read_b64 %1, %2 offset:24576\n s_waitcnt lgkmcnt(0)\n"
                                 : "=&v"(l0), "=&v"(l1) : "v"(ad) : "memory");
                    acc += l0 + l1;
                }
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
    std::vector<long long> h(cus * 16);
    const int modes[] = {0, 1, 2, 3, 4, 10, 11, 12, 13, 14};
    for (int mode : modes) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k, dim3(cus), dim3(1024), 0, 0, mode, d, sink);
            CHECK(hipDeviceSynchronize());
        }
        CHECK(hipMemcpy(h.data(), d, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
        double mx = 0;
        for (auto c : h) mx = c > mx ? c : mx;
        // 16 waves per CU issue ITER * 32 instructions each
        printf("mode %2d (%s): %.2f cycles per wave-instruction and CU (16 waves), wave time %.0f cycles\n", mode, mode < 10 ? "ds_write_b64" : "ds_read_b64 ",
               mx / (double)(ITER * 32) / 16.0, mx);
    }
    return 0;
}
