// Which VALU instructions of the BP loops run at the fp64 rate (4 cycles per wave64 instruction) and which at the 32-bit rate (2),
// and which rocprofv3 counter each one is counted under?  (gfx950)
//   plain run:   issue rate per SIMD for each instruction kind at 2 and 4 waves per SIMD (one workgroup per CU)
//   under `rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64
//          SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT`: the per-kernel counter values say where each kind is counted
//   (kernel names carry the kind: valu_kind<K>).
// Kinds: 0 v_add_f64  1 v_min_f64  2 v_max_f64  3 v_min_f64 with |x| modifier  4 v_cmp_ge_f64 (to vcc)  5 v_mul_f64  6 v_fma_f64
//        7 v_cndmask_b32  8 v_addc_co_u32  9 v_add_u32  10 v_and_b32  11 v_lshlrev_b32  12 v_mov_b32  13 v_bfe_u32  14 v_lshlrev_b64
//        15 v_cmp_lt_u32  16 v_xor_b32  17 v_or3_b32  18 v_mov_b32 dpp quad_perm  19 v_readfirstlane_b32  20 v_cvt_f64_i32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int ITER = 2048;
static const char *kNames[] = {"v_add_f64", "v_min_f64", "v_max_f64", "v_min_f64 |x|", "v_cmp_ge_f64", "v_mul_f64", "v_fma_f64", "v_cndmask_b32",
                               "v_addc_co_u32", "v_add_u32", "v_and_b32", "v_lshlrev_b32", "v_mov_b32", "v_bfe_u32", "v_lshlrev_b64",
                               "v_cmp_lt_u32", "v_xor_b32", "v_or3_b32", "v_mov_b32 dpp", "v_readfirstlane_b32", "v_cvt_f64_i32"};
constexpr int NK = 21;

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
template <int K>
__global__ void __launch_bounds__(1024) valu_kind(long long *cycles, double *sink) {
    double a[8];
    uint32_t u[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x + i; u[i] = threadIdx.x * 7 + i; }
    double c = 1.0000001;
    uint32_t s0 = 0;
    const long long t0 = clock64();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#define ASM_D(i) asm volatile(OPSTR : "+v"(a[i]) : "v"(c));
#define ASM_U(i) asm volatile(OPSTR : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
            if constexpr (K == 0) {
#define OPSTR "v_add_f64 %0, %0, %1"
                REP8(ASM_D)
#undef OPSTR
            } else if constexpr (K == 1) {
#define OPSTR "v_min_f64 %0, %0, %1"
                REP8(ASM_D)
#undef OPSTR
            } else if constexpr (K == 2) {
#define OPSTR "v_max_f64 %0, %0, %1"
                REP8(ASM_D)
#undef OPSTR
            } else if constexpr (K == 3) {
#define OPSTR "v_min_f64 %0, |%0|, %1"
                REP8(ASM_D)
#undef OPSTR
            } else if constexpr (K == 4) {
#define OPSTR "v_cmp_ge_f64 vcc, %0, %1"
#define ASM_C(i) asm volatile(OPSTR : : "v"(a[i]), "v"(c) : "vcc");
                REP8(ASM_C)
#undef OPSTR
            } else if constexpr (K == 5) {
#define OPSTR "v_mul_f64 %0, %0, %1"
                REP8(ASM_D)
#undef OPSTR
            } else if constexpr (K == 6) {
#define OPSTR "v_fma_f64 %0, %0, %1, %1"
                REP8(ASM_D)
#undef OPSTR
            } else if constexpr (K == 7) {
#define OPSTR "v_cndmask_b32 %0, %0, %1, vcc"
                REP8(ASM_U)
#undef OPSTR
            } else if constexpr (K == 8) {
#define OPSTR "v_addc_co_u32 %0, vcc, %0, %1, vcc"
#define ASM_UC(i) asm volatile(OPSTR : "+v"(u[i]) : "v"(u[(i + 1) & 7]) : "vcc");
                REP8(ASM_UC)
#undef OPSTR
            } else if constexpr (K == 9) {
#define OPSTR "v_add_u32 %0, %0, %1"
                REP8(ASM_U)
#undef OPSTR
            } else if constexpr (K == 10) {
#define OPSTR "v_and_b32 %0, %0, %1"
                REP8(ASM_U)
#undef OPSTR
            } else if constexpr (K == 11) {
#define OPSTR "v_lshlrev_b32 %0, 1, %0"
#define ASM_U1(i) asm volatile(OPSTR : "+v"(u[i]));
                REP8(ASM_U1)
#undef OPSTR
            } else if constexpr (K == 12) {
#define OPSTR "v_mov_b32 %0, %1"
                REP8(ASM_U)
#undef OPSTR
            } else if constexpr (K == 13) {
#define OPSTR "v_bfe_u32 %0, %0, 3, 9"
                REP8(ASM_U1)
#undef OPSTR
            } else if constexpr (K == 14) {
#define OPSTR "v_lshlrev_b64 %0, 1, %0"
#define ASM_D1(i) asm volatile(OPSTR : "+v"(a[i]));
                REP8(ASM_D1)
#undef OPSTR
            } else if constexpr (K == 15) {
#define OPSTR "v_cmp_lt_u32 vcc, %0, %1"
#define ASM_CU(i) asm volatile(OPSTR : : "v"(u[i]), "v"(u[(i + 1) & 7]) : "vcc");
                REP8(ASM_CU)
#undef OPSTR
            } else if constexpr (K == 16) {
#define OPSTR "v_xor_b32 %0, %0, %1"
                REP8(ASM_U)
#undef OPSTR
            } else if constexpr (K == 17) {
#define OPSTR "v_or3_b32 %0, %0, %1, %1"
                REP8(ASM_U)
#undef OPSTR
            } else if constexpr (K == 18) {
#define OPSTR "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                REP8(ASM_U)
#undef OPSTR
            } else if constexpr (K == 19) {
#define OPSTR "v_readfirstlane_b32 %0, %1"
#define ASM_S(i) asm volatile(OPSTR : "=s"(s0) : "v"(u[i]));
                REP8(ASM_S)
#undef OPSTR
            } else if constexpr (K == 20) {
#define OPSTR "v_cvt_f64_i32 %0, %1"
#define ASM_CV(i) asm volatile(OPSTR : "=v"(a[i]) : "v"(u[i]));
                REP8(ASM_CV)
#undef OPSTR
            }
        }
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    double acc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += a[i] + (double)u[i];
    if (acc == 123.456 + s0) sink[0] = acc;
}

template <int K>
int run(long long *dcyc, double *dsink, int cus) {
    for (int waves : {2, 4}) {
        hipLaunchKernelGGL((valu_kind<K>), dim3(cus), dim3(256 * waves), 0, 0, dcyc, dsink);
        CHECK(hipDeviceSynchronize());
        std::vector<long long> h(cus);
        CHECK(hipMemcpy(h.data(), dcyc, cus * sizeof(long long), hipMemcpyDeviceToHost));
        long long mx = 0;
        for (auto v : h) mx = std::max(mx, v);
        const double insts = (double)ITER * 32;
        printf("%-22s %d waves/SIMD: %9lld ticks for %.0f instructions per wave -> %.3f instructions per tick per SIMD = %.2f ticks per instruction\n",
               kNames[K], waves, mx, insts, insts * waves / mx, mx / (insts * waves));
    }
    return 0;
}

template <int K>
int run_all(long long *dcyc, double *dsink, int cus) {
    if (run<K>(dcyc, dsink, cus)) return 1;
    if constexpr (K + 1 < NK) return run_all<K + 1>(dcyc, dsink, cus);
    return 0;
}

int main() {
    int cus = 0;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    long long *dcyc; double *dsink;
    CHECK(hipMalloc(&dcyc, cus * sizeof(long long)));
    CHECK(hipMalloc(&dsink, 8));
    return run_all<0>(dcyc, dsink, cus);
}
