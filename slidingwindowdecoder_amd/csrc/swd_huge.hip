// osd_window on graphs beyond every kernel variant (more than 1024 checks, 9216 columns or 65 535 edges): the reference's
// mod2sparse has no size limit (/root/reference/src/include/mod2sparse.c:52-80, osd_window.pyx:20-63), e.g. the un-windowed
// [[288,12,18]] detector error model (2736 rows).  One 1024-thread workgroup per decode, EVERY array in the workgroup's slice of an
// HBM buffer (fp64 messages of both directions, the 4-slot posterior history, sort keys, the m x m transform matrix of the
// elimination), LDS only for block-wide scans.  A thread walks several checks / nodes; nothing is tuned -- this is the general
// path that keeps the class surface free of size refusals, bit-exact with the same oracle as the tuned kernels:
//   osd_window.decode 158-199, bp_decode_llr 381-485, vn_set_value 340-368, peel 306-338, osd 201-284,
//   mod2sparse_decomp_osd / LU_forward_backward_solve  src/include/mod2sparse_extra.cpp:78-376.
#include <math.h>
#include <string.h>

#include <memory>
#include <mutex>

#include "swd_host.h"
#include "swd_plan.h"

namespace swd {

struct SwdHugeArgs {
    int32_t m, n, E, new_n, rank, wm, npad;
    int32_t pre_iter, post_iter, osd_method, osd_order, B;
    double alpha;
    const int32_t *row_ptr, *col_idx;        // CSR, columns ascending inside a row
    const int32_t *col_ptr, *row_idx, *c2r;  // CSC (rows ascending inside a column), CSC position -> CSR edge
    const double *llr;
    const uint8_t *synd; int64_t synd_stride;
    uint8_t *out; int64_t out_stride;
    int32_t *stats; double *min_pm;
    double *hist; int32_t hist_is_state;     // nullable [B][4][n]
    uint8_t *osd0, *bp_dec;                  // nullable [B][n]
    uint8_t *scratch; int64_t scratch_stride;
    // offsets inside a workgroup's scratch slice
    int64_t o_b2c, o_c2b, o_hist, o_key, o_idx, o_pos, o_cnval, o_cndeg, o_vn, o_hard, o_bak, o_lv, o_lc, o_T, o_pc, o_pr,
        o_rowof, o_plist, o_ht, o_ycand, o_pm, o_best, o_tmp;
};

static constexpr int HNT = 1024;

__device__ __forceinline__ uint64_t huge_f2key(double x) {
    x = x + 0.0; // -0.0 -> +0.0: equal doubles get equal keys (the reference's stable sort compares with <)
    uint64_t u = (uint64_t)__double_as_longlong(x);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

struct HugeLds {
    int scan[HNT / 64 + 1];
    int flag[4];
    unsigned long long red[HNT / 64];
    int redi[HNT / 64];
    unsigned long long y[64 * 16]; // reduced columns of a batch (wm <= 64 words each)
    int piv[16];
};

// exclusive prefix sum of one int per thread over the block; *total = sum.  Two barriers.
__device__ __forceinline__ int huge_scan(int x, HugeLds &s, int *total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int v = x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(v, o, 64); if (lane >= o) v += t; }
    __syncthreads();
    if (lane == 63) s.scan[w] = v;
    __syncthreads();
    int base = 0, tot = 0;
    for (int i = 0; i < HNT / 64; ++i) { const int c = s.scan[i]; if (i < w) base += c; tot += c; }
    *total = tot;
    return base + v - x;
}

__device__ __forceinline__ bool huge_any(bool p, HugeLds &s) {
    __syncthreads();
    if (threadIdx.x == 0) s.flag[0] = 0;
    __syncthreads();
    if (p) s.flag[0] = 1;
    __syncthreads();
    return s.flag[0] != 0;
}

// indices i in [0, count) with pred(i), ascending, into list[]; returns how many (every thread a contiguous chunk)
template <class F>
__device__ __forceinline__ int huge_compact(int count, int32_t *list, HugeLds &s, F pred) {
    const int ch = (count + HNT - 1) / HNT, i0 = min(count, (int)threadIdx.x * ch), i1 = min(count, i0 + ch);
    int c = 0;
    for (int i = i0; i < i1; ++i) c += pred(i) ? 1 : 0;
    int tot;
    int o = huge_scan(c, s, &tot);
    for (int i = i0; i < i1; ++i) if (pred(i)) list[o++] = i;
    __syncthreads();
    return tot;
}

// sum of llr[v] over the listed nodes IN LIST ORDER (ascending v: "pm" sums of osd_window.pyx run over v ascending), by one thread
__device__ __forceinline__ double huge_ordered_sum(const int32_t *list, int cnt, const double *llr) {
    double pm = 0.0;
    for (int i = 0; i < cnt; ++i) pm += llr[list[i]];
    return pm;
}

struct HugeView {
    double *b2c, *c2b, *hist;
    uint64_t *key; int32_t *idx, *pos, *cnval, *cndeg, *vn, *lv, *lc, *pc, *pr, *rowof, *plist, *ht, *best, *tmp;
    uint8_t *hard, *bak;
    uint64_t *T, *ycand;
    double *pm;
};

// masked min-sum (osd_window.pyx:381-485): `iters` flooding iterations at most; lists: live checks / live nodes (nlc / nlv entries).
// Returns 1 when H * decision == syndrome after an iteration; *done = iterations executed.
__device__ int huge_bp(const SwdHugeArgs &a, const HugeView &v, const uint8_t *synd, int iters, const int32_t *lc, int nlc,
                       const int32_t *lv, int nlv, HugeLds &s, int *done) {
    const int tid = threadIdx.x;
    *done = 0;
    for (int it = 0; it < iters; ++it) {
        // check pass: first and second minimum of the clipped magnitudes over the live edges, parity of the non-positive ones
        for (int q = tid; q < nlc; q += HNT) {
            const int c = lc[q];
            const int e0 = a.row_ptr[c], e1 = a.row_ptr[c + 1];
            double min1 = 1e308, min2 = 1e308;
            int arg = -1, neg = (v.cnval[c] == 1) ? 1 : 0;
            for (int e = e0; e < e1; ++e) {
                if (v.vn[a.col_idx[e]] != -1) continue;
                double x = v.b2c[e];
                x = (x > 50.0) ? 50.0 : ((x < -50.0) ? -50.0 : x);
                const double ax = fabs(x);
                if (ax < min1) { min2 = min1; min1 = ax; arg = e; }
                else if (ax < min2) min2 = ax;
                neg += (x <= 0) ? 1 : 0;
            }
            for (int e = e0; e < e1; ++e) {
                if (v.vn[a.col_idx[e]] != -1) continue;
                double x = v.b2c[e];
                const int sg = (neg - ((x <= 0) ? 1 : 0)) & 1; // (clipping keeps the sign)
                const double mag = (e == arg) ? min2 : min1;   // minimum over the OTHER live edges (none: the 1e308 sentinel)
                v.c2b[e] = mag * (sg ? -a.alpha : a.alpha);
            }
        }
        __syncthreads();
        // variable-node pass: prefix / suffix sums in row order, posterior into history slot it % 4
        double *hs = v.hist + (size_t)(it & 3) * a.n;
        for (int q = tid; q < nlv; q += HNT) {
            const int x = lv[q];
            const int k0 = a.col_ptr[x], k1 = a.col_ptr[x + 1];
            double temp = a.llr[x];
            for (int k = k0; k < k1; ++k) {
                if (v.cnval[a.row_idx[k]] == -1) continue;
                const int e = a.c2r[k];
                v.b2c[e] = temp;
                temp += v.c2b[e];
            }
            hs[x] = temp;
            v.hard[x] = (temp <= 0) ? 1 : 0;
            temp = 0.0;
            for (int k = k1 - 1; k >= k0; --k) {
                if (v.cnval[a.row_idx[k]] == -1) continue;
                const int e = a.c2r[k];
                v.b2c[e] += temp;
                temp += v.c2b[e];
            }
        }
        __syncthreads();
        // H * decision == syndrome over the FULL matrix (decided nodes included)
        bool bad = false;
        for (int c = tid; c < a.m; c += HNT) {
            int p = 0;
            for (int e = a.row_ptr[c]; e < a.row_ptr[c + 1]; ++e) p ^= v.hard[a.col_idx[e]];
            if (p != (synd[c] ? 1 : 0)) bad = true;
        }
        *done = it + 1;
        if (!huge_any(bad, s)) return 1;
    }
    return 0;
}

// ascending bitonic sort of (key, idx) pairs, lexicographic = the reference's stable ascending argsort (bpgd.cpp:384-389)
__device__ void huge_sort(uint64_t *key, int32_t *idx, int npad) {
    for (int k = 2; k <= npad; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < npad; i += HNT) {
                const int l = i ^ j;
                if (l > i) {
                    const uint64_t ki = key[i], kl = key[l];
                    const int32_t ii = idx[i], il = idx[l];
                    const bool up = (i & k) == 0;
                    const bool gt = ki > kl || (ki == kl && ii > il);
                    if (gt == up) { key[i] = kl; key[l] = ki; idx[i] = il; idx[l] = ii; }
                }
            }
            __syncthreads();
        }
}

// vn_set_value (osd_window.pyx:340-368) by ONE thread, on the live state
__device__ int huge_set_value(const SwdHugeArgs &a, const HugeView &v, int x, int value) {
    if (v.vn[x] != -1) return (v.vn[x] == value) ? 0 : -1;
    v.vn[x] = value;
    v.hard[x] = (uint8_t)value;
    for (int k = a.col_ptr[x]; k < a.col_ptr[x + 1]; ++k) {
        const int c = a.row_idx[k];
        if (v.cnval[c] == -1) continue;
        const int deg = v.cndeg[c] - 1;
        if (value) v.cnval[c] = 1 - v.cnval[c];
        if (deg == 0) {
            if (v.cnval[c] != 0) return -1;
            v.cnval[c] = -1;
        }
        v.cndeg[c] = deg;
    }
    return 0;
}

__global__ void __launch_bounds__(HNT) huge_kernel(const SwdHugeArgs a) {
    __shared__ HugeLds s;
    const int tid = threadIdx.x, m = a.m, n = a.n, wm = a.wm;
    uint8_t *base = a.scratch + (int64_t)blockIdx.x * a.scratch_stride;
    HugeView v;
    v.b2c = (double *)(base + a.o_b2c); v.c2b = (double *)(base + a.o_c2b); v.hist = (double *)(base + a.o_hist);
    v.key = (uint64_t *)(base + a.o_key); v.idx = (int32_t *)(base + a.o_idx); v.pos = (int32_t *)(base + a.o_pos);
    v.cnval = (int32_t *)(base + a.o_cnval); v.cndeg = (int32_t *)(base + a.o_cndeg); v.vn = (int32_t *)(base + a.o_vn);
    v.hard = base + a.o_hard; v.bak = base + a.o_bak; v.lv = (int32_t *)(base + a.o_lv); v.lc = (int32_t *)(base + a.o_lc);
    v.T = (uint64_t *)(base + a.o_T); v.pc = (int32_t *)(base + a.o_pc); v.pr = (int32_t *)(base + a.o_pr);
    v.rowof = (int32_t *)(base + a.o_rowof); v.plist = (int32_t *)(base + a.o_plist); v.ht = (int32_t *)(base + a.o_ht);
    v.ycand = (uint64_t *)(base + a.o_ycand); v.pm = (double *)(base + a.o_pm); v.best = (int32_t *)(base + a.o_best);
    v.tmp = (int32_t *)(base + a.o_tmp);
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const uint8_t *synd = a.synd + (int64_t)b * a.synd_stride;
        uint8_t *out = a.out + (int64_t)b * a.out_stride;
        double *hio = a.hist ? a.hist + (int64_t)b * 4 * n : nullptr;
        __syncthreads();
        // reset (osd_window.pyx:288-303) + bp_init
        for (int c = tid; c < m; c += HNT) { v.cnval[c] = synd[c] ? 1 : 0; v.cndeg[c] = a.row_ptr[c + 1] - a.row_ptr[c]; v.lc[c] = c; }
        for (int x = tid; x < n; x += HNT) { v.vn[x] = -1; v.hard[x] = 0; v.lv[x] = x; }
        for (int i = tid; i < 4 * n; i += HNT) v.hist[i] = (hio && a.hist_is_state) ? hio[i] : 0.0; // a new reference object starts from zeros
        for (int e = tid; e < a.E; e += HNT) v.b2c[e] = a.llr[a.col_idx[e]];
        __syncthreads();
        int it_pre = 0, it_post = 0, exit_class = -1, conv = 0, nlv = 0, nlc = 0, nle = 0, rowadds = 0;
        double min_pm = 0.0;
        const uint8_t *ret = v.hard;
        conv = huge_bp(a, v, synd, a.pre_iter, v.lc, m, v.lv, n, s, &it_pre);
        if (conv) exit_class = SWD_EXIT_PRE;
        else {
            // history sum in slot order, stable sort, decimation of cols[new_n:] to 0 (osd_window.pyx:172-181)
            for (int i = tid; i < a.npad; i += HNT) {
                if (i < n) { v.key[i] = huge_f2key(((v.hist[i] + v.hist[n + i]) + v.hist[2 * (size_t)n + i]) + v.hist[3 * (size_t)n + i]); v.idx[i] = i; }
                else { v.key[i] = ~0ull; v.idx[i] = 0x7FFFFFFF; }
            }
            __syncthreads();
            huge_sort(v.key, v.idx, a.npad);
            for (int i = tid; i < n; i += HNT) v.pos[v.idx[i]] = i;
            __syncthreads();
            // a check whose live nodes are ALL decimated reaches degree 0 when the last of them (in sorted order) is set: with a residual
            // value of 1 that is the "setting vn failed" exit, at the first such position
            if (tid == 0) s.flag[1] = 0x7FFFFFFF;
            __syncthreads();
            for (int c = tid; c < m; c += HNT) {
                int cnt = 0, last = -1;
                for (int e = a.row_ptr[c]; e < a.row_ptr[c + 1]; ++e) { const int p = v.pos[a.col_idx[e]]; if (p >= a.new_n) { ++cnt; last = max(last, p); } }
                v.tmp[c] = cnt;
                if (cnt > 0 && cnt == v.cndeg[c] && v.cnval[c] == 1) atomicMin(&s.flag[1], last);
            }
            __syncthreads();
            const int fail_at = s.flag[1];
            if (fail_at != 0x7FFFFFFF) {
                for (int x = tid; x < n; x += HNT) if (v.pos[x] >= a.new_n && v.pos[x] <= fail_at) v.hard[x] = 0;
                exit_class = SWD_EXIT_FAIL_SET;
                __syncthreads();
            } else {
                for (int x = tid; x < n; x += HNT) if (v.pos[x] >= a.new_n) { v.vn[x] = 0; v.hard[x] = 0; }
                for (int c = tid; c < m; c += HNT) {
                    const int d = v.cndeg[c] - v.tmp[c];
                    v.cndeg[c] = d;
                    if (d == 0 && v.tmp[c] > 0) v.cnval[c] = -1; // (value 0: the check is met and leaves the graph)
                }
                __syncthreads();
                // peel (osd_window.pyx:306-338): parallel rounds; the closure does not depend on the order unless a contradiction
                // appears -- then the backup is restored and one thread replays the reference's sweep to the point where it stops
                int32_t *bvn = (int32_t *)v.bak, *bcv = bvn + n, *bcd = bcv + m;
                uint8_t *bh = (uint8_t *)(bcd + m);
                for (int x = tid; x < n; x += HNT) { bvn[x] = v.vn[x]; bh[x] = v.hard[x]; }
                for (int c = tid; c < m; c += HNT) { bcv[c] = v.cnval[c]; bcd[c] = v.cndeg[c]; }
                __syncthreads();
                bool contra = false;
                for (;;) {
                    bool work = false, bad = false;
                    for (int c = tid; c < m; c += HNT) v.tmp[c] = -1;
                    __syncthreads();
                    for (int c = tid; c < m; c += HNT) {
                        if (v.cnval[c] == -1 || v.cndeg[c] >= 2) continue;
                        work = true;
                        int x = -1;
                        for (int e = a.row_ptr[c]; e < a.row_ptr[c + 1]; ++e) if (v.vn[a.col_idx[e]] == -1) { x = a.col_idx[e]; break; }
                        if (x < 0) { bad = true; continue; }
                        v.tmp[c] = x; // proposal: node x takes the check's residual value
                    }
                    __syncthreads();
                    // a node proposed by several checks takes the value of the lowest check (the first in the reference's sweep); the
                    // others see their degree reach 0 in the update below and are met or contradicted like in the serial order
                    for (int c = tid; c < m; c += HNT) {
                        const int x = v.tmp[c];
                        if (x < 0) continue;
                        bool first = true;
                        for (int k = a.col_ptr[x]; k < a.col_ptr[x + 1]; ++k) { const int c2 = a.row_idx[k]; if (c2 < c && v.tmp[c2] == x) { first = false; break; } }
                        if (first) { v.vn[x] = v.cnval[c]; v.hard[x] = (uint8_t)v.cnval[c]; v.pos[x] = -2 - v.cnval[c]; } // (pos marks "decided in this round")
                    }
                    __syncthreads();
                    for (int c = tid; c < m; c += HNT) {
                        if (v.cnval[c] == -1) continue;
                        int dec = 0, flip = 0;
                        for (int e = a.row_ptr[c]; e < a.row_ptr[c + 1]; ++e) { const int p = v.pos[a.col_idx[e]]; if (p <= -2) { ++dec; flip ^= (p == -3) ? 1 : 0; } }
                        if (!dec) continue;
                        const int d = v.cndeg[c] - dec, val = v.cnval[c] ^ flip;
                        v.cndeg[c] = d;
                        if (d == 0) { if (val != 0) bad = true; v.cnval[c] = -1; } else v.cnval[c] = val;
                    }
                    __syncthreads();
                    for (int x = tid; x < n; x += HNT) if (v.pos[x] <= -2) v.pos[x] = 0; // (positions are not needed any more)
                    contra = huge_any(bad, s);
                    if (contra) break;
                    if (!huge_any(work, s)) break;
                }
                if (contra) {
                    for (int x = tid; x < n; x += HNT) { v.vn[x] = bvn[x]; v.hard[x] = bh[x]; }
                    for (int c = tid; c < m; c += HNT) { v.cnval[c] = bcv[c]; v.cndeg[c] = bcd[c]; }
                    __syncthreads();
                    if (tid == 0) {
                        int rc = 0;
                        for (;;) {
                            int degree_check = 1;
                            for (int c = 0; c < m && rc == 0; ++c) {
                                if (v.cnval[c] == -1 || v.cndeg[c] >= 2) continue;
                                degree_check = 0;
                                int x = -1;
                                for (int e = a.row_ptr[c]; e < a.row_ptr[c + 1]; ++e) if (v.vn[a.col_idx[e]] == -1) { x = a.col_idx[e]; break; }
                                if (x < 0 || huge_set_value(a, v, x, v.cnval[c]) == -1) rc = -1;
                            }
                            if (rc || degree_check) break;
                        }
                        s.flag[2] = rc;
                    }
                    __syncthreads();
                    if (s.flag[2]) exit_class = SWD_EXIT_FAIL_PEEL; // (a contradiction the parallel rounds saw is one the sweep meets too)
                    __syncthreads();
                }
                if (exit_class < 0) {
                    // the shortened graph: live lists, bp_init of the live nodes (osd_window.pyx:187), post-processing BP
                    nlv = huge_compact(n, v.lv, s, [&](int x) { return v.vn[x] == -1; });
                    nlc = huge_compact(m, v.lc, s, [&](int c) { return v.cnval[c] != -1; });
                    int le = 0;
                    for (int q = tid; q < nlv; q += HNT) {
                        const int x = v.lv[q];
                        for (int k = a.col_ptr[x]; k < a.col_ptr[x + 1]; ++k) { v.b2c[a.c2r[k]] = a.llr[x]; le += (v.cnval[a.row_idx[k]] != -1) ? 1 : 0; }
                    }
                    int dummy;
                    (void)huge_scan(le, s, &nle);
                    (void)dummy;
                    __syncthreads();
                    conv = huge_bp(a, v, synd, a.post_iter, v.lc, nlc, v.lv, nlv, s, &it_post);
                    if (conv) exit_class = SWD_EXIT_POST;
                    else if (a.osd_order < 0) exit_class = SWD_EXIT_NO_OSD;
                }
            }
        }
        if (exit_class == SWD_EXIT_PRE || exit_class == SWD_EXIT_POST) {
            const int cnt = huge_compact(n, v.plist, s, [&](int x) { return v.hard[x] != 0; });
            if (tid == 0) v.pm[0] = huge_ordered_sum(v.plist, cnt, a.llr);
            __syncthreads();
            min_pm = v.pm[0];
        }
        if (exit_class < 0) {
            // ---- OSD (osd_window.pyx:201-284) ----
            exit_class = SWD_EXIT_OSD;
            if (a.bp_dec) for (int x = tid; x < n; x += HNT) a.bp_dec[(int64_t)b * n + x] = v.hard[x];
            const uint64_t kp = huge_f2key(1000.0), km = huge_f2key(-1000.0);
            for (int i = tid; i < a.npad; i += HNT) {
                if (i < n) {
                    const int st = v.vn[i];
                    v.key[i] = st == 1 ? km : (st == 0 ? kp : huge_f2key(((v.hist[i] + v.hist[n + i]) + v.hist[2 * (size_t)n + i]) + v.hist[3 * (size_t)n + i]));
                    v.idx[i] = i;
                } else { v.key[i] = ~0ull; v.idx[i] = 0x7FFFFFFF; }
            }
            __syncthreads();
            huge_sort(v.key, v.idx, a.npad); // idx = orig_cols
            // transform matrix T (word-major: T[w * m + j] = word w of column j), identity
            for (int i = tid; i < wm * m; i += HNT) { const int w = i / m, j = i - w * m; v.T[i] = (j >> 6) == w ? (1ull << (j & 63)) : 0ull; }
            for (int x = tid; x < n; x += HNT) v.rowof[x] = -1;
            uint64_t *pivmask = (uint64_t *)v.pm; // [wm] (the path metrics are not needed before the sweep)
            for (int w = tid; w < wm; w += HNT) pivmask[w] = 0ull;
            __syncthreads();
            // greedy first-independent columns in sorted order; pivot row = lowest unpivoted row with a 1 (mod2sparse_extra.cpp:113-376).
            // Sixteen columns are reduced against T at a time (one wave each); the first of them with a pivot is applied, the scan
            // resumes behind it (the reduced forms of the later ones are stale then).
            int np = 0, kcol = 0;
            const int wv = tid >> 6, lane = tid & 63;
            while (np < a.rank && kcol < n) {
                {
                    const int kk = kcol + wv;
                    uint64_t y = 0ull;
                    if (kk < n && lane < wm) {
                        const int col = v.idx[kk];
                        for (int k = a.col_ptr[col]; k < a.col_ptr[col + 1]; ++k) y ^= v.T[(size_t)lane * m + a.row_idx[k]];
                    }
                    s.y[wv * 64 + lane] = y;
                    const uint64_t cand = (lane < wm) ? (y & ~pivmask[lane]) : 0ull;
                    const unsigned long long bal = __ballot(cand != 0ull);
                    if (lane == 0) s.piv[wv] = -1;
                    if (bal) {
                        const int w0 = __ffsll((long long)bal) - 1;
                        const uint64_t cw = __shfl(cand, w0, 64);
                        if (lane == 0) s.piv[wv] = w0 * 64 + (__ffsll((long long)cw) - 1);
                    }
                }
                __syncthreads();
                int first = -1;
                for (int q = 0; q < 16; ++q) if (s.piv[q] >= 0) { first = q; break; }
                if (first < 0) { kcol += 16; __syncthreads(); continue; }
                const int r = s.piv[first], col = v.idx[kcol + first];
                // Gauss-Jordan step in transform form: every column j of T with bit r set takes S = y with bit r cleared
                const unsigned long long *yv = &s.y[first * 64];
                for (int j = tid; j < m; j += HNT) {
                    if ((v.T[(size_t)(r >> 6) * m + j] >> (r & 63)) & 1ull) {
                        for (int w = 0; w < wm; ++w) {
                            uint64_t sv = (uint64_t)yv[w];
                            if (w == (r >> 6)) sv &= ~(1ull << (r & 63));
                            if (sv) v.T[(size_t)w * m + j] ^= sv;
                        }
                    }
                }
                if (tid == 0) {
                    v.pc[np] = col; v.pr[np] = r; v.rowof[col] = r;
                    pivmask[r >> 6] |= 1ull << (r & 63);
                    int ra = 0;
                    for (int w = 0; w < wm; ++w) ra += __popcll((unsigned long long)yv[w]);
                    s.flag[3] = ra - 1;
                }
                __syncthreads();
                rowadds += s.flag[3];
                ++np;
                kcol += first + 1;
                __syncthreads();
            }
            // base = T * syndrome; the OSD-0 solution: pivot column i takes bit pr[i] of it, every other column 0
            uint64_t *basev = (uint64_t *)(v.pm) + wm; // [wm]
            {
                const int cnt = huge_compact(m, v.plist, s, [&](int c) { return synd[c] != 0; });
                for (int w = tid; w < wm; w += HNT) {
                    uint64_t y = 0ull;
                    for (int i = 0; i < cnt; ++i) y ^= v.T[(size_t)w * m + v.plist[i]];
                    basev[w] = y;
                }
                __syncthreads();
            }
            uint8_t *o0 = v.bak; // [n] osd0_decoding
            for (int x = tid; x < n; x += HNT) { const int r = v.rowof[x]; o0[x] = (r >= 0 && ((basev[r >> 6] >> (r & 63)) & 1ull)) ? 1 : 0; }
            __syncthreads();
            const int npiv = huge_compact(n, v.plist, s, [&](int x) { return v.rowof[x] >= 0; }); // pivot columns, ascending
            {
                const int cnt = huge_compact(n, v.lv, s, [&](int x) { return o0[x] != 0; });
                if (tid == 0) v.best[2] = cnt, ((double *)v.best)[2] = huge_ordered_sum(v.lv, cnt, a.llr);
                __syncthreads();
            }
            min_pm = ((double *)v.best)[2];
            if (a.osd0) for (int x = tid; x < n; x += HNT) a.osd0[(int64_t)b * n + x] = o0[x];
            ret = o0;
            int bestc = -1;
            if (a.osd_order > 0) {
                // candidate columns: the first k = new_n - rank non-pivot columns among the first new_n of the sorted order (:243-256)
                const int k = a.new_n - a.rank;
                int nht = huge_compact(a.new_n, v.lc, s, [&](int i) { return v.rowof[v.idx[i]] < 0; }); // positions in sorted order
                nht = min(nht, k);
                for (int j = tid; j < nht; j += HNT) v.ht[j] = v.idx[v.lc[j]];
                __syncthreads();
                const int nyc = (a.osd_method == 2) ? nht : min(nht, a.osd_order); // columns whose reduced form the sweep needs
                for (int j = wv; j < nyc; j += 16) {
                    uint64_t y = 0ull;
                    if (lane < wm) { const int col = v.ht[j]; for (int q = a.col_ptr[col]; q < a.col_ptr[col + 1]; ++q) y ^= v.T[(size_t)lane * m + a.row_idx[q]]; }
                    if (lane < wm) v.ycand[(size_t)j * wm + lane] = y;
                }
                __syncthreads();
                // candidates in the reference's order: osd_cs -- k of weight one, then the pairs i < j < order (:134-155); osd_e -- every
                // pattern of the first `order` columns, pattern l = the binary digits of l (:128-132)
                const int w = a.osd_order;
                const long long ncand = (a.osd_method == 2) ? (long long)k + (long long)w * (w - 1) / 2 : (1ll << w);
                double bpm = min_pm;
                long long bidx = -1;
                for (long long l = tid; l < ncand; l += HNT) {
                    int mem[16], nm = 0;
                    if (a.osd_method == 2) {
                        if (l < k) mem[nm++] = (int)l;
                        else { long long q = l - k; int i = 0; while (q >= w - 1 - i) { q -= w - 1 - i; ++i; } mem[nm++] = i; mem[nm++] = i + 1 + (int)q; }
                    } else {
                        for (int bit = 0; bit < w; ++bit) if ((l >> bit) & 1) mem[nm++] = bit;
                    }
                    // members beyond the candidate columns that exist contribute nothing (enc rows are k long: they cannot occur)
                    int nmv = 0;
                    int memc[16];
                    for (int q = 0; q < nm; ++q) if (mem[q] < nht) { mem[nmv] = mem[q]; memc[nmv] = v.ht[mem[q]]; ++nmv; }
                    // candidate columns in ascending column order for the ordered sum
                    for (int p = 1; p < nmv; ++p) { const int cc = memc[p], mm2 = mem[p]; int q = p - 1; while (q >= 0 && memc[q] > cc) { memc[q + 1] = memc[q]; mem[q + 1] = mem[q]; --q; } memc[q + 1] = cc; mem[q + 1] = mm2; }
                    double pm = 0.0;
                    int nx = 0;
                    for (int i = 0; i < npiv; ++i) {
                        const int col = v.plist[i], r = v.rowof[col];
                        while (nx < nmv && memc[nx] < col) pm += a.llr[memc[nx++]];
                        uint64_t bit = (basev[r >> 6] >> (r & 63)) & 1ull;
                        for (int q = 0; q < nmv; ++q) bit ^= (v.ycand[(size_t)mem[q] * wm + (r >> 6)] >> (r & 63)) & 1ull;
                        if (bit) pm += a.llr[col];
                    }
                    while (nx < nmv) pm += a.llr[memc[nx++]];
                    if (pm < bpm) { bpm = pm; bidx = l; } // (ascending l per thread: strict < keeps the earliest)
                }
                // block minimum of (pm, index): the reference keeps the first candidate that is strictly better than everything before
                for (int o = 32; o > 0; o >>= 1) {
                    const double op = __shfl_xor(bpm, o, 64);
                    const long long oi = __shfl_xor(bidx, o, 64);
                    if (oi >= 0 && (bidx < 0 || op < bpm || (op == bpm && oi < bidx))) { bpm = op; bidx = oi; }
                }
                __syncthreads();
                if (lane == 0) { s.red[wv] = (unsigned long long)__double_as_longlong(bpm); ((long long *)s.y)[wv] = bidx; }
                __syncthreads();
                bpm = min_pm; bidx = -1;
                for (int q = 0; q < HNT / 64; ++q) {
                    const double op = __longlong_as_double((long long)s.red[q]);
                    const long long oi = ((long long *)s.y)[q];
                    if (oi >= 0 && (bidx < 0 || op < bpm || (op == bpm && oi < bidx))) { bpm = op; bidx = oi; }
                }
                __syncthreads();
                if (bidx >= 0 && bpm < min_pm) {
                    bestc = 1;
                    min_pm = bpm;
                    int mem[16], nm = 0;
                    const long long l = bidx;
                    if (a.osd_method == 2) {
                        if (l < k) mem[nm++] = (int)l;
                        else { long long q = l - k; int i = 0; while (q >= w - 1 - i) { q -= w - 1 - i; ++i; } mem[nm++] = i; mem[nm++] = i + 1 + (int)q; }
                    } else {
                        for (int bit = 0; bit < w; ++bit) if ((l >> bit) & 1) mem[nm++] = bit;
                    }
                    uint8_t *ow = v.hard; // (the BP decisions went out above)
                    for (int x = tid; x < n; x += HNT) {
                        const int r = v.rowof[x];
                        uint64_t bit = 0;
                        if (r >= 0) {
                            bit = (basev[r >> 6] >> (r & 63)) & 1ull;
                            for (int q = 0; q < nm; ++q) if (mem[q] < nht) bit ^= (v.ycand[(size_t)mem[q] * wm + (r >> 6)] >> (r & 63)) & 1ull;
                        }
                        ow[x] = (uint8_t)bit;
                    }
                    __syncthreads();
                    if (tid == 0) for (int q = 0; q < nm; ++q) if (mem[q] < nht) ow[v.ht[mem[q]]] = 1;
                    __syncthreads();
                    ret = ow;
                }
            }
            (void)bestc;
        }
        __syncthreads();
        for (int x = tid; x < n; x += HNT) out[x] = ret[x];
        if (hio) for (int i = tid; i < 4 * n; i += HNT) hio[i] = v.hist[i];
        if (tid == 0) {
            if (a.stats) {
                int32_t *st = a.stats + (int64_t)b * SWD_STAT_WORDS;
                st[0] = exit_class | (conv ? SWD_STATUS_CONVERGE : 0);
                st[1] = it_pre + it_post; st[2] = it_pre; st[3] = it_post; st[4] = nlv; st[5] = nlc; st[6] = nle; st[7] = rowadds;
            }
            if (a.min_pm) a.min_pm[b] = min_pm;
        }
    }
}

struct Huge : HugeIface {
    int device = 0, wm = 0, npad = 0, E = 0;
    swd_osdw_params p{};
    DevBuf graph, scratch;
    SwdHugeArgs tmpl{};
    int64_t stride = 0;
    int grid_max = 0;
    std::mutex mu;

    int decode_dev(int32_t B, const uint8_t *synd, int64_t synd_stride, uint8_t *out, int64_t out_stride, int32_t *stats,
                   double *min_pm, double *hist, int32_t hist_is_state, uint8_t *osd0, uint8_t *bp_dec, void *stream) override {
        std::lock_guard<std::mutex> lk(mu); // one scratch area: launches of one handle run one after the other
        SWD_HIP(hipSetDevice(device));
        const int grid = std::max(1, std::min(B, grid_max));
        if (scratch.reserve((size_t)grid * (size_t)stride)) return -1;
        SwdHugeArgs a = tmpl;
        a.B = B; a.synd = synd; a.synd_stride = synd_stride ? synd_stride : m; a.out = out; a.out_stride = out_stride ? out_stride : n;
        a.stats = stats; a.min_pm = min_pm; a.hist = hist; a.hist_is_state = hist_is_state; a.osd0 = osd0; a.bp_dec = bp_dec;
        a.scratch = scratch.as<uint8_t>(); a.scratch_stride = stride;
        hipStream_t st = (hipStream_t)stream;
        // (the scratch area is shared by consecutive launches of this handle: order them on the device too)
        if (last_stream_set && last_stream != st) SWD_HIP(hipStreamSynchronize(last_stream));
        hipLaunchKernelGGL(huge_kernel, dim3(grid), dim3(HNT), 0, st, a);
        SWD_HIP(hipGetLastError());
        last_stream = st; last_stream_set = true;
        return 0;
    }
    hipStream_t last_stream = nullptr;
    bool last_stream_set = false;
};

// builds the general form for a graph beyond the kernel variants; NULL (with a message) when even that cannot take it
HugeIface *huge_create(const swd_graph_desc *g, const swd_osdw_params *p, int device) {
    const int m = g->m, n = g->n, E = g->nnz;
    if (m <= 0 || n <= 0 || E <= 0 || g->row_ptr[0] != 0 || g->row_ptr[m] != E) { set_error("empty or inconsistent check matrix"); return nullptr; }
    if (m > 4096) { set_error("m=%d exceeds the general form's limit of 4096 checks (64 words per column of the elimination's transform matrix)", m); return nullptr; }
    if ((long long)n > (1 << 22)) { set_error("n=%d exceeds the general form's limit of 4194304 columns", n); return nullptr; }
    std::unique_ptr<Huge> h(new Huge());
    h->device = device; h->p = *p; h->m = m; h->n = n; h->E = E;
    std::vector<int32_t> row_ptr(g->row_ptr, g->row_ptr + m + 1), col_idx(g->col_idx, g->col_idx + E);
    for (int r = 0; r < m; ++r) {
        if (row_ptr[r + 1] < row_ptr[r]) { set_error("row_ptr not monotone at row %d", r); return nullptr; }
        std::sort(col_idx.begin() + row_ptr[r], col_idx.begin() + row_ptr[r + 1]);
        for (int e = row_ptr[r]; e < row_ptr[r + 1]; ++e) {
            if (col_idx[e] < 0 || col_idx[e] >= n) { set_error("column index out of range in row %d", r); return nullptr; }
            if (e > row_ptr[r] && col_idx[e] == col_idx[e - 1]) { set_error("duplicate entry in row %d", r); return nullptr; }
        }
    }
    std::vector<int32_t> col_ptr(n + 1, 0), row_idx(E), c2r(E), fill(n, 0);
    for (int e = 0; e < E; ++e) col_ptr[col_idx[e] + 1]++;
    for (int v = 0; v < n; ++v) col_ptr[v + 1] += col_ptr[v];
    for (int c = 0; c < m; ++c)
        for (int e = row_ptr[c]; e < row_ptr[c + 1]; ++e) { const int v = col_idx[e], k = col_ptr[v] + fill[v]++; row_idx[k] = c; c2r[k] = e; }
    std::vector<double> llr(n);
    for (int v = 0; v < n; ++v) llr[v] = log((1 - g->channel_probs[v]) / g->channel_probs[v]); // osd_window.pyx:113
    h->rank = gf2_rank(m, n, row_ptr, col_idx);
    h->new_n = (p->new_n <= 0) ? std::min(n, 2 * m) : std::min(p->new_n, n); // osd_window.pyx:60-63
    if (h->p.osd_method == 0) h->p.osd_order = 0;
    if (h->p.osd_order > h->new_n - h->rank) {
        set_error("For this code, the OSD order should be set in the range 0<=osd_oder<=%d.", h->new_n - h->rank);
        return nullptr;
    }
    if (h->p.osd_method == 1 && h->p.osd_order > 15) { set_error("osd_e supports osd_order <= 15 on the device"); return nullptr; }
    const int wm = (m + 63) / 64;
    int npad = 2; while (npad < n) npad <<= 1;
    h->wm = wm; h->npad = npad;
    if (hipSetDevice(device) != hipSuccess) { set_error("hipSetDevice(%d) failed", device); return nullptr; }
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t g_rp = 0, g_ci = al((size_t)(m + 1) * 4), g_cp = g_ci + al((size_t)E * 4), g_ri = g_cp + al((size_t)(n + 1) * 4),
                 g_cr = g_ri + al((size_t)E * 4), g_llr = g_cr + al((size_t)E * 4), g_tot = g_llr + al((size_t)n * 8);
    if (h->graph.reserve(g_tot)) return nullptr;
    char *gd = (char *)h->graph.p;
    auto up = [&](size_t off, const void *src, size_t bytes) { return hipMemcpy(gd + off, src, bytes, hipMemcpyHostToDevice) == hipSuccess; };
    if (!up(g_rp, row_ptr.data(), (size_t)(m + 1) * 4) || !up(g_ci, col_idx.data(), (size_t)E * 4) || !up(g_cp, col_ptr.data(), (size_t)(n + 1) * 4) ||
        !up(g_ri, row_idx.data(), (size_t)E * 4) || !up(g_cr, c2r.data(), (size_t)E * 4) || !up(g_llr, llr.data(), (size_t)n * 8)) {
        set_error("hipMemcpy of the graph failed");
        return nullptr;
    }
    SwdHugeArgs &a = h->tmpl;
    a.m = m; a.n = n; a.E = E; a.new_n = h->new_n; a.rank = h->rank; a.wm = wm; a.npad = npad;
    a.pre_iter = h->p.pre_max_iter; a.post_iter = h->p.post_max_iter; a.osd_method = h->p.osd_method; a.osd_order = h->p.osd_order;
    a.alpha = h->p.ms_scaling_factor;
    a.row_ptr = (const int32_t *)(gd + g_rp); a.col_idx = (const int32_t *)(gd + g_ci); a.col_ptr = (const int32_t *)(gd + g_cp);
    a.row_idx = (const int32_t *)(gd + g_ri); a.c2r = (const int32_t *)(gd + g_cr); a.llr = (const double *)(gd + g_llr);
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += al(bytes); return (int64_t)at; };
    const int k = std::max(h->new_n - h->rank, 0);
    const int nyc = h->p.osd_order <= 0 ? 0 : (h->p.osd_method == 2 ? k : std::min(k, h->p.osd_order));
    a.o_b2c = take((size_t)E * 8); a.o_c2b = take((size_t)E * 8); a.o_hist = take((size_t)4 * n * 8);
    a.o_key = take((size_t)npad * 8); a.o_idx = take((size_t)npad * 4); a.o_pos = take((size_t)n * 4);
    a.o_cnval = take((size_t)m * 4); a.o_cndeg = take((size_t)m * 4); a.o_vn = take((size_t)n * 4); a.o_hard = take((size_t)n);
    a.o_bak = take((size_t)n * 4 + (size_t)m * 8 + (size_t)n + 64); a.o_lv = take((size_t)n * 4); a.o_lc = take((size_t)std::max(m, h->new_n) * 4);
    a.o_T = take((size_t)wm * m * 8); a.o_pc = take((size_t)(h->rank + 1) * 4); a.o_pr = take((size_t)(h->rank + 1) * 4);
    a.o_rowof = take((size_t)n * 4); a.o_plist = take((size_t)std::max(n, m) * 4); a.o_ht = take((size_t)(k + 1) * 4);
    a.o_ycand = take((size_t)std::max(nyc, 1) * wm * 8); a.o_pm = take((size_t)(2 * wm + 4) * 8); a.o_best = take(64); a.o_tmp = take((size_t)m * 4);
    h->stride = (int64_t)al(o);
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) cus = 64;
    h->grid_max = std::max(1, cus);
    return h.release();
}

} // namespace swd
