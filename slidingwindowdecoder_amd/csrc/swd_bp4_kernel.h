// Quaternary min-sum BP + one OSD per basis on the device: bp4_osd.decode
// (/root/reference/src/bp4_osd.pyx:197-221) for one pair of syndromes per workgroup.
//   bp_init 425-442, bp4_decode_llr 444-481, cn_update_all 483-529, vn_update 533-589, osd 261-368,
//   log1pexp / logaddexp  /root/reference/src/include/bpgd.cpp:399-416.
// Messages of both Tanner graphs (Hx and Hz share the variable nodes) live in LDS.  The variable-node update
// needs exp / log1p: they are evaluated with the algorithms of the C library the reference links against
// (swd_libm.h: glibc's table-driven exp in its FMA build, fdlibm's log1p), so posteriors, decisions and OSD
// orderings are bit-identical to a reference run on such a host, not merely "equal up to the math library".
#pragma once
#include "swd_libm.h"
#include "swd_osdw_kernel.h"

struct SwdBp4Layout {
    int32_t off_msgz;   // msgX at 0 (Ex+1 doubles), msgZ here (Ez+1 doubles); both inside the scratch region
    int32_t off_jptrx, off_jptrz, off_cnx, off_cnz, off_parx, off_parz, off_sxo, off_szo, off_decx, off_decz,
        off_hard, off_misc, total;
};

struct SwdBp4Args {
    SwdGraphDev gx, gz;   // Hx / Hz; their llr arrays hold prior_llr_x / prior_llr_z (OSD path metrics)
    SwdLdsLayout Lx, Lz;  // OSD scratch layouts (npad, off_idx, off_aux, off_cs, cs_par) per basis
    SwdBp4Layout L;
    const double *llr_x, *llr_y, *llr_z; // [n] channel LLRs (bp4_osd.pyx:131-133)
    int32_t max_iter, osd_method, osd_order, B;
    double alpha;
    const uint8_t *sx, *sz;  // [B][mx], [B][mz]
    uint8_t *out;            // [B][2][n]  rows: X string, Z string
    uint8_t *osd0;           // nullable [B][2][n]
    uint8_t *bp_dec;         // nullable [B][2][n] BP hard decisions at exit
    int32_t *stats;          // [B][SWD_STAT_WORDS]
    double *lpr;             // [B][3][n] posterior LLRs (x, y, z); also the OSD ordering input
    // camel_decode (bp4_osd.pyx:223-247): 4 workgroups per shot, workgroup 4b+v fixes the last qubit to Pauli v
    // (0 I, 1 X, 2 Z, 3 Y) and runs plain BP4; out/osd0 are unused, lpr is [4B][3][n] scratch
    // decodes that leave BP unconverged are queued here and finished by bp4_osd_kernel on the same stream: the BP kernel then carries
    // neither the elimination's code nor its scratch arrays (17.3 -> 20.4 M decodes/s on [[144,12,12]])
    int32_t *osd_list;       // [B]
    uint32_t *osd_count;     // zeroed before the launch
    int32_t camel;
    // round 6: units are drawn by ticket (the first gridDim.x statically) in the order `order` lists them -- heaviest syndrome first.
    // A decode that does not converge runs max_iter iterations against 1-3 of the others (80x a normal decode at the notebooks'
    // noise rates): with a static share of the units per workgroup a launch ended on the workgroups that had met one, the device
    // 60 % empty on average.  Both NULL: static shares (camel runs).
    uint32_t *ticket;        // zeroed before the launch
    const uint32_t *order;   // [B] decode numbers by decreasing syndrome weight (bp4_weight_kernel + shot_order_kernel)
    int32_t split;           // FAST instantiation: two threads per qubit -- thread v the Hx edges of qubit v, thread n + v its Hz edges
    int32_t lpr_wanted;      // the caller reads lpr: every decode stores its posteriors (else only those the OSD kernel finishes)
    uint8_t *camel_dec;      // [4B][2][n] decisions of every run
    double *camel_pm;        // [4B] cal_pm of the converged runs
    int32_t *camel_st;       // [4B][2] converged, iterations
};

namespace swd {

#ifdef SWD_BP4_CALLS
#define SWD_BP4_FN __device__ __attribute__((noinline))
#else
#define SWD_BP4_FN __device__ __forceinline__
#endif
// (bpgd.cpp:399-416.  Written without branches: the reference's two cases of each routine differ in the ARGUMENT of one and the same
// evaluation, so a lane selects the argument, evaluates once and selects the result -- the same operations on the same values per
// lane, and a wave whose lanes disagree about the case no longer walks both inlined copies of exp / log1p, four per logaddexp.)
SWD_BP4_FN double bp4_log1pexp(double x, const uint64_t *tab = nullptr) {
    const bool big = x > 36.04365338911715; // -log(DBL_EPSILON)
    const double r = swd_log1p(swd_exp_from(big ? -x : x, tab));
    return big ? x + r : r;
}
SWD_BP4_FN double bp4_logaddexp(double x, double y, const uint64_t *tab = nullptr) {
    const double tmp = x - y;
    const bool gt = tmp > 0, le = tmp <= 0;
    const double r = (gt ? x : y) + bp4_log1pexp(gt ? -tmp : tmp, tab);
    if (x == y) return x + 0.693147180559945309417232121458176568;
    return (gt || le) ? r : tmp; // (NaN: neither case)
}

// block_any (swd_osdw_kernel.h) for a workgroup whose size is a launch parameter
__device__ __forceinline__ bool bp4_block_any(bool p, Lds &s, int nwaves) {
    const int par = (s.fpar++) & 1;
    const unsigned long long b = __ballot(p);
    if ((threadIdx.x & 63) == 0) s.flags[par * 16 + (threadIdx.x >> 6)] = (b != 0ull);
    __syncthreads();
    int r = 0;
    for (int w = 0; w < nwaves; ++w) r |= s.flags[par * 16 + w];
    return __builtin_amdgcn_readfirstlane(r) != 0;
}

#ifndef SWD_BP4_OWN_CHECK
#define SWD_BP4_OWN_CHECK 1
#endif
// one check of the plain min-sum CN pass (bp4_osd.pyx:483-529): true when its parity word was still set
__device__ __forceinline__ bool bp4_cn_one(double *msg, const uint16_t *jptr, const int8_t *cn, uint32_t *par, int l, int deg, int it, double alpha) {
    bool unsat = false;
    {
        const int cv = cn[l];
        if (it > 0 && par[l] != 0u) unsat = true;
        par[l] = (uint32_t)cv;
        double min1 = 1e308, min2 = 1e308;
        int arg = -1;
        uint64_t negm = 0;
        for (int k = 0; k < deg; ++k) {
            double x = msg[jptr[k] + l];
            // (clip and comparisons as in bp4_cn_pass below; asking for every message before the first is used -- a fixed unroll of
            // eight predicated positions -- made the pass 1.5x slower)
            x = (x > 50.0) ? 50.0 : ((x < -50.0) ? -50.0 : x);
            const double ax = (x != x) ? 1e308 : fabs(x);
            arg = (ax < min1) ? k : arg;
            min2 = fmin(min2, fmax(min1, ax));
            min1 = fmin(min1, ax);
            negm |= (x <= 0) ? (1ull << k) : 0ull;
        }
        const int sg = (cv ^ __popcll(negm)) & 1;
        for (int k = 0; k < deg; ++k) {
            const double mag = (k == arg) ? min2 : min1;
            const int sgn = sg ^ (int)((negm >> k) & 1ull);
            msg[jptr[k] + l] = mag * (sgn ? -alpha : alpha);
        }
    }
    return unsat;
}

// plain min-sum CN pass over one graph: lanes [lane0, lane0 + g.m) of the block own its checks
__device__ __forceinline__ bool bp4_cn_pass(int nthreads, const SwdGraphDev &g, double *msg, const uint16_t *jptr, const int8_t *cn,
                                            uint32_t *par, int lane0, int it, double alpha) {
    bool unsat = false;
    for (int l = (int)threadIdx.x - lane0; l < g.m; l += nthreads) {
        if (l < 0) continue;
        const int cv = cn[l];
        if (it > 0 && par[l] != 0u) unsat = true;
        par[l] = (uint32_t)cv;
        const int deg = g.row_deg[l];
        double min1 = 1e308, min2 = 1e308;
        int arg = -1;
        uint64_t negm = 0;
        for (int k = 0; k < deg; ++k) {
            double x = msg[jptr[k] + l];
            // the reference clips and compares with plain < and > (bp4_osd.pyx:503-509): a NaN message (inf - inf
            // in vn_update when a qubit collects +-1e308 sentinels of degree-1 checks) passes the clip, never
            // lowers the running minimum and counts as positive -- fmin/fmax would turn it into -50
            x = (x > 50.0) ? 50.0 : ((x < -50.0) ? -50.0 : x);
            const double ax = (x != x) ? 1e308 : fabs(x);
            arg = (ax < min1) ? k : arg;
            min2 = fmin(min2, fmax(min1, ax));
            min1 = fmin(min1, ax);
            negm |= (x <= 0) ? (1ull << k) : 0ull;
        }
        const int sg = (cv ^ __popcll(negm)) & 1;
        for (int k = 0; k < deg; ++k) {
            const double mag = (k == arg) ? min2 : min1;
            const int sgn = sg ^ (int)((negm >> k) & 1ull);
            msg[jptr[k] + l] = mag * (sgn ? -alpha : alpha);
        }
    }
    return unsat;
}

#ifndef SWD_BP4_ROLLED
#define SWD_BP4_ROLLED 1 // the variable-node update's per-edge loops as loops (one helper body per basis; SHYPS r = 3 +10 %, the BB codes unchanged, half the compile time)
#endif
#ifndef SWD_BP4_LAZY_MSG
#define SWD_BP4_LAZY_MSG 1 // the node update's message half runs only when another check pass follows (round 6; needs SWD_BP4_ROLLED)
#endif
#ifndef SWD_BP4_LAZY_ITS
#define SWD_BP4_LAZY_ITS 3 // ... in a decode's first iterations (the iterations of the decodes that stop early)
#endif
#ifndef SWD_BP4_PRIO_IT
#define SWD_BP4_PRIO_IT 3 // iteration from which a decode's waves run at raised priority (-1: never)
#endif
#ifndef SWD_BP4_WAVES
// waves per SIMD the register allocation leaves room for (kernels of up to 256 / up to 512 threads).  Round 4 (static shares of the
// units): 1 / 2 / 3 / 4 / 6 / 8 -> 6.9 / 10.0 / 13.4 / 15.3 / 15.4 / 15.0 M decodes/s on [[144,12,12]].  Round 6 (ticket-scheduled units,
// gpurun_out/r06e -> profiles/r06_bp4_codes_rate.log): 4 / 5 / 6 -> [[144]] 35.2 / 36.4 / 35.7 M, [[72]] 44.8 / 43.5 / 41.4, SHYPS r = 3
// 26.4 / 24.8 / 23.8 (all one- to three-wave workgroups: five); [[288]] 16.9 / 19.8 / 20.2, [[360]] 15.3 / 14.6 / 15.9 (up to eight waves: six)
#define SWD_BP4_WAVES(wmax) ((wmax) <= 4 ? 5 : 6)
#endif
// WMAX: the most waves a workgroup of this instantiation is launched with (4 / 8: up to 256 / 512 threads, SWD_BP4_WAVES waves per SIMD;
// 16: up to 1024 threads, 128 registers).  The workgroup size itself is a launch parameter: ceil(n / 64) waves while that is at most 16, so that every qubit has
// a thread of its own and the node's edges / LLRs / bp_init messages stay in registers.
// FAST (round 6): the launch of the notebooks' codes -- every qubit has a thread (n <= NT), every check of both graphs has a thread
// (mx + mz <= NT), no camel run -- as its own instantiation: the generic paths (strided node loops, per-graph check passes, the
// decided qubit of camel_decode, loads of the graph tables inside the iterations) are compiled out.
// LAZY (round 6): the two-half variable-node update in a decode's first iterations (below).  An instantiation of its own, compiled for fewer
// waves per SIMD than the fused form: it pays when the register budget has room for both forms of the update (at the fused form's budgets it
// cost ~10 % on the eight-wave workgroups) -- [[144]] 58 -> 72 M decodes/s with two launches in flight, [[756]] 7.1 -> 9.3.  The launcher takes
// it for every launch with one thread per qubit (swd_bp4.hip, bp4_dispatch_nt); the two-threads-per-qubit launches keep the fused update.
// The LAZY instantiation of up to four waves is compiled for three waves per SIMD (139 registers, 48 B of scratch) instead of five (96 and
// 224 B): its short decodes are chains of dependent steps that want their values in registers more than they want neighbours --
// [[144]], 1 / 2 / 3 / 4 launches in flight: 36.7 / 67.4 / 73.5 / 78.9 -> 37.4 / 73.5 / 78.2 / 84.3 M decodes/s (four waves: 38.0 / 69.7 / 76.8 / 81.8).
#ifndef SWD_BP4_WAVES_LAZY
#define SWD_BP4_WAVES_LAZY 3
#endif
template <int WMAX, int DM, bool FAST, bool LAZY = false>
#ifndef SWD_BP4_WAVES_LAZY8
#define SWD_BP4_WAVES_LAZY8 4 // ... and the five- to eight-wave ones for four (128 registers, 96 B; six: 80 registers, the fused form's choice): [[288]] 19.6 -> 22.8 M decodes/s one launch at a time, 26.8 -> 32.0 M with two in flight; [[360]] 15.7 -> 20.7, 20.5 -> 27.7 (three waves: 15.4 / 19.9, 14.1 / 18.3)
#endif
__global__ void __launch_bounds__(WMAX * 64, (WMAX <= 8 ? (LAZY ? (WMAX <= 4 ? SWD_BP4_WAVES_LAZY : SWD_BP4_WAVES_LAZY8) : SWD_BP4_WAVES(WMAX)) : 1)) bp4_kernel(const SwdBp4Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // glibc's exp table (swd_libm.h) in LDS: tail and scale of an entry are one aligned 16-byte read
    __shared__ __attribute__((aligned(16))) uint64_t s_exptab[256];
    const int NT = (int)blockDim.x;
    for (int i = threadIdx.x; i < 256; i += NT) s_exptab[i] = swd_exp_tab_dev[i];
    __syncthreads(); // (the channel-only messages below already evaluate exp)
    const uint64_t *const xt = s_exptab;
    const bool camel_run = !FAST && a.camel;
    const SwdGraphDev &gx = a.gx, &gz = a.gz;
    const int tid = threadIdx.x, n = gx.n, mx = gx.m, mz = gz.m;
    const int fixed = camel_run ? n - 1 : -1; // the decided qubit of a camel run
    const int nunits = camel_run ? 4 * a.B : a.B;
    double *msgx = (double *)smem, *msgz = (double *)(smem + a.L.off_msgz);
    uint16_t *jpx = (uint16_t *)(smem + a.L.off_jptrx), *jpz = (uint16_t *)(smem + a.L.off_jptrz);
    int8_t *cnx = (int8_t *)(smem + a.L.off_cnx), *cnz = (int8_t *)(smem + a.L.off_cnz);
    uint32_t *parx = (uint32_t *)(smem + a.L.off_parx), *parz = (uint32_t *)(smem + a.L.off_parz);
    uint8_t *decx = (uint8_t *)(smem + a.L.off_decx), *decz = (uint8_t *)(smem + a.L.off_decz);
    Lds s;
    s.scratch = smem; s.msg = msgx; s.hard = (uint8_t *)(smem + a.L.off_hard);
    s.flags = (int *)(smem + a.L.off_misc); s.scal = s.flags + 32; s.dbl = (double *)(s.scal + 32); s.iaux = (int *)(s.dbl + 24);
    s.ctid = s.vtid = threadIdx.x;
    // what does not depend on the syndrome is set up once per workgroup: the grid is persistent (as many workgroups as the device
    // holds, each taking units blockIdx.x, blockIdx.x + gridDim.x, ...), and when every variable node has a thread of its own
    // (n <= NT: the BB and SHYPS codes of the notebooks) its edges, degrees and channel LLRs stay in registers -- no global load
    // and no posterior store inside the iterations (the posteriors of the last update are stored once, after the loop)
    for (int j = tid; j <= gx.K; j += NT) jpx[j] = gx.jptr[j];
    for (int j = tid; j <= gz.K; j += NT) jpz[j] = gz.jptr[j];
    const bool one = FAST || n <= NT;
    // Round 6: two threads per qubit (a.split; FAST launches with 2 n <= 1024).  Both compute the node's three posteriors from all of its
    // messages (same operations in the same order), thread v then updates the Hx edges, thread n + v the Hz edges: the chain of eight
    // dependent exp / log1p evaluations per node and iteration -- what a launch waits for once its units are ticket-scheduled: the
    // decodes that run all max_iter iterations -- becomes two chains of four.
    const bool split = FAST && a.split != 0;
    const int vt = (split && tid >= n) ? tid - n : tid;          // the thread's qubit
    const int hsel = !split ? 3 : (tid < n ? 1 : 2);             // bit 0: the Hx edges are this thread's, bit 1: the Hz edges
    const bool mine = one && vt < n;
    int c_dx = 0, c_dz = 0;
    uint32_t c_ex[DM], c_ez[DM];
    double c_lx = 0.0, c_ly = 0.0, c_lz = 0.0;
#pragma unroll
    for (int k = 0; k < DM; ++k) { c_ex[k] = 0u; c_ez[k] = 0u; }
    if (mine) {
        c_dx = gx.col_deg[vt]; c_dz = gz.col_deg[vt];
#pragma unroll
        for (int k = 0; k < DM; ++k) {
            c_ex[k] = (k < c_dx) ? gx.vn_edge[k * n + vt] : 0u;
            c_ez[k] = (k < c_dz) ? gz.vn_edge[k * n + vt] : 0u;
        }
        c_lx = a.llr_x[vt]; c_ly = a.llr_y[vt]; c_lz = a.llr_z[vt];
    }
    // (the messages of bp_init depend on the channel only: three of a decode's ~14 exp / log1p evaluations, hoisted out of the unit loop)
    double c_mx = 0.0, c_mz = 0.0;
    if (mine) {
        c_mx = bp4_log1pexp(-1. * c_lx, xt) - bp4_logaddexp(-1. * c_ly, -1. * c_lz, xt);
        c_mz = bp4_log1pexp(-1. * c_lz, xt) - bp4_logaddexp(-1. * c_ly, -1. * c_lz, xt); // sic (bp4_osd.pyx:438)
    }
    // Round 5: when every check of both graphs has a thread (mx + mz <= NT: the notebooks' codes), a thread keeps ONE check for the
    // whole launch -- the checks of Hx, then those of Hz, dealt to the waves in equal shares -- with its degree, its syndrome index
    // and its arrays in registers: the check pass is one walk per wave instead of two half-empty ones (the wave that held the end
    // of Hx and the start of Hz walked both: 2x the others, which waited for it at the barrier), and neither the pass nor the
    // reset waits for a load of the graph's tables any more.
    const int mtot = mx + mz;
    const bool own = FAST || (SWD_BP4_OWN_CHECK && mtot <= NT);
    int o_l = -1, o_deg = 0, o_perm = 0;
    bool o_z = false;
    if (own) {
        const int nw = NT >> 6, q = (mtot + nw - 1) / nw, c = (tid >> 6) * q + (tid & 63);
        if ((tid & 63) < q && c < mtot) {
            o_z = c >= mx;
            o_l = o_z ? c - mx : c;
            o_deg = o_z ? (int)gz.row_deg[o_l] : (int)gx.row_deg[o_l];
            o_perm = o_z ? (int)gz.perm[o_l] : (int)gx.perm[o_l];
        }
    }
    double *const o_msg = o_z ? msgz : msgx;
    const uint16_t *const o_jp = o_z ? jpz : jpx;
    int8_t *const o_cn = o_z ? cnz : cnx;
    uint32_t *const o_par = o_z ? parz : parx;
#ifdef SWD_BP4PROF // diagnostic build: cycles (s_memtime) per region of a decode, summed over the units of this workgroup
    long long q_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, q0_ = clock64();
    long long q_iters = 0, q_units = 0;
#define BP4T(i) { const long long t_ = clock64(); q_[i] += t_ - q0_; q0_ = t_; }
#else
#define BP4T(i)
#endif
    uint32_t tk_next = 0; // (thread 0) the ticket drawn for the workgroup's next unit
    auto next_unit = [&](uint32_t tk) -> uint32_t {
        if (!a.ticket) return tk + gridDim.x;
        if (tid == 0) s.scal[1] = (int)tk_next;
        __syncthreads();
        return (uint32_t)s.scal[1];
    };
    for (uint32_t tk = blockIdx.x; tk < (uint32_t)nunits; tk = next_unit(tk)) {
    const int unit = (a.ticket && a.order) ? (int)a.order[tk] : (int)tk;
    const int b = camel_run ? unit >> 2 : unit;
    s.fpar = 0;
    const uint8_t *sx_b = a.sx + (int64_t)b * mx, *sz_b = a.sz + (int64_t)b * mz;
    double *lpr_b = a.lpr + (int64_t)unit * 3 * n;
    double p_x = 0.0, p_y = 0.0, p_z = 0.0; // (one node per thread) the posteriors of the last update
    bool p_set = false;
    __syncthreads(); // the previous unit is done with LDS

    // reset + bp_init (bp4_osd.pyx:371-386, 425-442)
    if (own) {
        if (o_l >= 0) o_cn[o_l] = (int8_t)((o_z ? sz_b : sx_b)[o_perm] ? 1 : 0);
    } else {
        for (int l = tid; l < mx; l += NT) cnx[l] = (int8_t)(sx_b[gx.perm[l]] ? 1 : 0);
        for (int l = tid; l < mz; l += NT) cnz[l] = (int8_t)(sz_b[gz.perm[l]] ? 1 : 0);
    }
    for (int v = vt; v < n; v += NT) {
        if (hsel & 1) { decx[v] = 0; decz[v] = 0; }
        double m_x = c_mx, m_z = c_mz;
        if (!one) {
            const double llrx = a.llr_x[v], llry = a.llr_y[v], llrz = a.llr_z[v];
            m_x = bp4_log1pexp(-1. * llrx, xt) - bp4_logaddexp(-1. * llry, -1. * llrz, xt);
            m_z = bp4_log1pexp(-1. * llrz, xt) - bp4_logaddexp(-1. * llry, -1. * llrz, xt); // sic (bp4_osd.pyx:438)
        }
        const int dx = one ? c_dx : (int)gx.col_deg[v], dz = one ? c_dz : (int)gz.col_deg[v];
        if (one) {
#pragma unroll
            for (int k = 0; k < DM; ++k) { if (k < dx && (hsel & 1)) msgx[swd_edge_slot(c_ex[k])] = m_x; if (k < dz && (hsel & 2)) msgz[swd_edge_slot(c_ez[k])] = m_z; }
        } else {
            for (int k = 0; k < dx; ++k) msgx[swd_edge_slot(gx.vn_edge[k * n + v])] = m_x;
            for (int k = 0; k < dz; ++k) msgz[swd_edge_slot(gz.vn_edge[k * n + v])] = m_z;
        }
    }
    __syncthreads();
    if (camel_run) { // vn_set_value(n - 1, value) after bp_init (bp4_osd.pyx:234-236, 388-423): the qubit keeps its prior messages
        const int value = unit & 3, x = value & 1, z = value >> 1;
        if (tid == 0) { decx[fixed] = (uint8_t)x; decz[fixed] = (uint8_t)z; }
        if (z) for (int k = tid; k < gx.col_deg[fixed]; k += NT) cnx[swd_edge_lane(gx.vn_edge[k * n + fixed])] ^= 1;
        if (x) for (int k = tid; k < gz.col_deg[fixed]; k += NT) cnz[swd_edge_lane(gz.vn_edge[k * n + fixed])] ^= 1;
        __syncthreads();
    }

    BP4T(0) // reset + bp_init
    // bp4_decode_llr (bp4_osd.pyx:444-481)
    int conv = 0, iters = 0;
    const int lane0z = (mx + mz <= NT) ? mx : 0; // Hz checks on the lanes after the Hx ones when both fit
    // Round 6: in a decode's first SWD_BP4_LAZY_ITS iterations the variable-node update runs in two halves with the convergence test between them.  Half A is what the test and the outputs
    // need -- sums of the check messages, the three posteriors, the hard decision, its parity flips; half B, the new bit-to-check
    // messages (a log1pexp and a logaddexp per edge: most of a decode's arithmetic), only feeds the NEXT check pass and is skipped when the
    // decode stops here -- for the notebooks' noise rates after the first update in three decodes out of four.  Outputs are those of
    // the reference's "update everything, then test" (bp4_osd.pyx:444-481): nothing reads the messages of a decode that has ended.
    // A decode that is still running after those iterations is one of the few that run for long: it goes on with the fused update of
    // rounds 4-5 (two barriers per iteration instead of four; the test of an update is made by the next check pass).
    for (int it = 0; it < a.max_iter; ++it) {
#if SWD_BP4_LAZY_MSG && SWD_BP4_ROLLED
        if constexpr (LAZY) if (it < SWD_BP4_LAZY_ITS) {
        if (SWD_BP4_PRIO_IT >= 0 && it == SWD_BP4_PRIO_IT) __builtin_amdgcn_s_setprio(2);
        if (own) {
            if (o_l >= 0) (void)bp4_cn_one(o_msg, o_jp, o_cn, o_par, o_l, o_deg, 0, a.alpha);
        } else {
            (void)bp4_cn_pass(NT, gx, msgx, jpx, cnx, parx, 0, 0, a.alpha);
            (void)bp4_cn_pass(NT, gz, msgz, jpz, cnz, parz, lane0z, 0, a.alpha);
        }
        BP4T(1) // check passes
        __syncthreads();
        BP4T(2)
        // the node's edge words, check messages and sums (bp4_osd.pyx:533-558); the same operations in the same order in both halves
        auto node_sums = [&](int v, uint32_t (&ex)[DM], uint32_t (&ez)[DM], int &dx, int &dz, double &lx, double &ly, double &lz) {
            dx = one ? c_dx : (int)gx.col_deg[v]; dz = one ? c_dz : (int)gz.col_deg[v];
            double cx[DM], cz[DM];
#pragma unroll
            for (int k = 0; k < DM; ++k) {
                ex[k] = one ? c_ex[k] : ((k < dx) ? gx.vn_edge[k * n + v] : 0u);
                ez[k] = one ? c_ez[k] : ((k < dz) ? gz.vn_edge[k * n + v] : 0u);
            }
            const double v_lx = one ? c_lx : a.llr_x[v], v_ly = one ? c_ly : a.llr_y[v], v_lz = one ? c_lz : a.llr_z[v];
#pragma unroll
            for (int k = 0; k < DM; ++k) { cx[k] = msgx[swd_edge_slot(ex[k])]; cz[k] = msgz[swd_edge_slot(ez[k])]; }
            double llrx_hx = 0.0, llrz_hz = 0.0;
#pragma unroll
            for (int k = 0; k < DM; ++k) if (k < dz) llrx_hx += cz[k];
#pragma unroll
            for (int k = 0; k < DM; ++k) if (k < dx) llrz_hz += cx[k];
            ly = llrx_hx + llrz_hz + v_ly;
            lx = llrx_hx + v_lx;
            lz = llrz_hz + v_lz;
        };
        auto pick = [&](const uint32_t (&ev)[DM], int k) { uint32_t e = ev[0];
#pragma unroll
            for (int j = 1; j < DM; ++j) e = (k == j) ? ev[j] : e;
            return e; };
        for (int v = vt; v < n; v += NT) { // half A
            if (v == fixed) continue; // (decided, bp4_osd.pyx:456-458: its parity flips went into the checks' syndrome bits)
            uint32_t ex[DM], ez[DM];
            int dx, dz;
            double llrx_hx, llry_all, llrz_hz;
            node_sums(v, ex, ez, dx, dz, llrx_hx, llry_all, llrz_hz);
            if (one) { p_x = llrx_hx; p_y = llry_all; p_z = llrz_hz; p_set = (hsel & 1) != 0; }
            else { lpr_b[v] = llrx_hx; lpr_b[n + v] = llry_all; lpr_b[2 * n + v] = llrz_hz; }
            int idx;
            if (0 < llrx_hx && 0 < llry_all && 0 < llrz_hz) idx = 0;
            else if (llrx_hx < llry_all && llrx_hx < llrz_hz) idx = 1;
            else if (llry_all > llrz_hz) idx = 2;
            else idx = 3;
            const int bx = idx & 1, bz = idx >> 1;
            if (hsel & 1) { decx[v] = (uint8_t)bx; decz[v] = (uint8_t)bz; }
            if (bz && (hsel & 1)) // Hx * z-string
#pragma unroll 1
                for (int k = 0; k < dx; ++k) atomicXor(&parx[swd_edge_lane(pick(ex, k))], 1u);
            if (bx && (hsel & 2)) // Hz * x-string
#pragma unroll 1
                for (int k = 0; k < dz; ++k) atomicXor(&parz[swd_edge_lane(pick(ez, k))], 1u);
        }
        BP4T(3) // node: message loads, sums, posteriors, decision, parity flips
        __syncthreads();
        bool unsat = false;
        if (own) { if (o_l >= 0) unsat = o_par[o_l] != 0u; }
        else {
            for (int l = tid; l < mx; l += NT) if (parx[l] != 0u) unsat = true;
            for (int l = tid; l < mz; l += NT) if (parz[l] != 0u) unsat = true;
        }
        const bool any = bp4_block_any(unsat, s, NT >> 6);
        BP4T(7)
#ifdef SWD_BP4PROF
        ++q_iters;
#endif
        if (!any) { conv = 1; iters = it + 1; break; }
        if (it + 1 >= a.max_iter) break; // (the messages of the last update feed no check pass)
        for (int v = vt; v < n; v += NT) { // half B: the new bit-to-check messages (bp4_osd.pyx:571-589)
            if (v == fixed) { // decided: its bit-to-check messages stay the priors of bp_init; the CN pass has overwritten the shared
                              // slots with check-to-bit values, so put them back
                const double llrx = a.llr_x[v], llry = a.llr_y[v], llrz = a.llr_z[v];
                const double m_x = bp4_log1pexp(-1. * llrx, xt) - bp4_logaddexp(-1. * llry, -1. * llrz, xt);
                const double m_z = bp4_log1pexp(-1. * llrz, xt) - bp4_logaddexp(-1. * llry, -1. * llrz, xt);
                for (int k = 0; k < gx.col_deg[v]; ++k) msgx[swd_edge_slot(gx.vn_edge[k * n + v])] = m_x;
                for (int k = 0; k < gz.col_deg[v]; ++k) msgz[swd_edge_slot(gz.vn_edge[k * n + v])] = m_z;
                continue;
            }
            uint32_t ex[DM], ez[DM];
            int dx, dz;
            double llrx_hx, llry_all, llrz_hz;
            // (the sums of half A again, from the check messages it left untouched: holding them across the test costs registers this
            // kernel does not have -- [[144]], 4 launches in flight: 73 -> 79 M decodes/s without them)
            node_sums(v, ex, ez, dx, dz, llrx_hx, llry_all, llrz_hz);
            const double num_hx = (hsel & 1) ? bp4_log1pexp(-1. * llrx_hx, xt) : 0.0;
            BP4T(4) // log1pexp
#pragma unroll 1
            for (int k = 0; k < ((hsel & 1) ? dx : 0); ++k) {
                const uint32_t e = pick(ex, k);
                const double c = msgx[swd_edge_slot(e)];
                const double aa = llrz_hz - c, bb = llry_all - c;
                msgx[swd_edge_slot(e)] = num_hx - bp4_logaddexp(-1. * aa, -1. * bb, xt);
            }
            BP4T(5) // Hx edges: logaddexp + store each
            const double num_hz = (hsel & 2) ? bp4_log1pexp(-1. * llrz_hz, xt) : 0.0;
            BP4T(4)
#pragma unroll 1
            for (int k = 0; k < ((hsel & 2) ? dz : 0); ++k) {
                const uint32_t e = pick(ez, k);
                const double c = msgz[swd_edge_slot(e)];
                const double aa = llrx_hx - c, bb = llry_all - c;
                msgz[swd_edge_slot(e)] = num_hz - bp4_logaddexp(-1. * aa, -1. * bb, xt);
            }
            BP4T(6) // Hz edges
        }
        __syncthreads();
            continue;
        }
#endif
        // a decode that is still running after a few iterations is one of the few that will run for long: its waves issue ahead of
        // the others on their SIMDs from here on (what ends a launch is the last of these decodes, not the device's throughput)
        if (SWD_BP4_PRIO_IT >= 0 && it == SWD_BP4_PRIO_IT) __builtin_amdgcn_s_setprio(2);
        bool unsat = false;
        if (own) {
            if (o_l >= 0) unsat = bp4_cn_one(o_msg, o_jp, o_cn, o_par, o_l, o_deg, it, a.alpha);
        } else {
            unsat = bp4_cn_pass(NT, gx, msgx, jpx, cnx, parx, 0, it, a.alpha);
            unsat |= bp4_cn_pass(NT, gz, msgz, jpz, cnz, parz, lane0z, it, a.alpha);
        }
        BP4T(1) // check passes
        const bool any = bp4_block_any(unsat, s, NT >> 6);
        BP4T(2) // flags + barrier
        if (it > 0 && !any) { conv = 1; iters = it; break; }
        for (int v = vt; v < n; v += NT) { // vn_update (bp4_osd.pyx:533-589)
            if (v == fixed) { // decided (bp4_osd.pyx:456-458): its bit-to-check messages stay the priors of bp_init; the
                              // CN pass has just overwritten the shared slots with check-to-bit values, so put them back
                const double llrx = a.llr_x[v], llry = a.llr_y[v], llrz = a.llr_z[v];
                const double m_x = bp4_log1pexp(-1. * llrx, xt) - bp4_logaddexp(-1. * llry, -1. * llrz, xt);
                const double m_z = bp4_log1pexp(-1. * llrz, xt) - bp4_logaddexp(-1. * llry, -1. * llrz, xt);
                for (int k = 0; k < gx.col_deg[v]; ++k) msgx[swd_edge_slot(gx.vn_edge[k * n + v])] = m_x;
                for (int k = 0; k < gz.col_deg[v]; ++k) msgz[swd_edge_slot(gz.vn_edge[k * n + v])] = m_z;
                continue;
            }
            const int dx = one ? c_dx : (int)gx.col_deg[v], dz = one ? c_dz : (int)gz.col_deg[v];
            uint32_t ex[DM], ez[DM];
            double cx[DM], cz[DM];
#pragma unroll
            for (int k = 0; k < DM; ++k) {
                ex[k] = one ? c_ex[k] : ((k < dx) ? gx.vn_edge[k * n + v] : 0u);
                ez[k] = one ? c_ez[k] : ((k < dz) ? gz.vn_edge[k * n + v] : 0u);
            }
            const double v_lx = one ? c_lx : a.llr_x[v], v_ly = one ? c_ly : a.llr_y[v], v_lz = one ? c_lz : a.llr_z[v];
#pragma unroll
            for (int k = 0; k < DM; ++k) { cx[k] = msgx[swd_edge_slot(ex[k])]; cz[k] = msgz[swd_edge_slot(ez[k])]; }
            double llrx_hx = 0.0, llrz_hz = 0.0;
#pragma unroll
            for (int k = 0; k < DM; ++k) if (k < dz) llrx_hx += cz[k];
#pragma unroll
            for (int k = 0; k < DM; ++k) if (k < dx) llrz_hz += cx[k];
            const double llry_all = llrx_hx + llrz_hz + v_ly;
            llrx_hx = llrx_hx + v_lx;
            llrz_hz = llrz_hz + v_lz;
            if (one) { p_x = llrx_hx; p_y = llry_all; p_z = llrz_hz; p_set = (hsel & 1) != 0; }
            else { lpr_b[v] = llrx_hx; lpr_b[n + v] = llry_all; lpr_b[2 * n + v] = llrz_hz; }
            int idx;
            if (0 < llrx_hx && 0 < llry_all && 0 < llrz_hz) idx = 0;
            else if (llrx_hx < llry_all && llrx_hx < llrz_hz) idx = 1;
            else if (llry_all > llrz_hz) idx = 2;
            else idx = 3;
            const int bx = idx & 1, bz = idx >> 1;
            if (hsel & 1) { decx[v] = (uint8_t)bx; decz[v] = (uint8_t)bz; }
            BP4T(3) // node: message loads, sums, posteriors, decision
            const double num_hx = (hsel & 1) ? bp4_log1pexp(-1. * llrx_hx, xt) : 0.0;
            BP4T(4) // log1pexp
#if SWD_BP4_ROLLED // one body of the helper per basis instead of DM: the edge word is picked by a select chain, the message re-read from LDS
            auto pick = [&](const uint32_t (&ev)[DM], int k) { uint32_t e = ev[0];
#pragma unroll
                for (int j = 1; j < DM; ++j) e = (k == j) ? ev[j] : e;
                return e; };
#pragma unroll 1
            for (int k = 0; k < ((hsel & 1) ? dx : 0); ++k) {
                const uint32_t e = pick(ex, k);
                const double c = msgx[swd_edge_slot(e)];
                const double aa = llrz_hz - c, bb = llry_all - c;
                msgx[swd_edge_slot(e)] = num_hx - bp4_logaddexp(-1. * aa, -1. * bb, xt);
                if (bz) atomicXor(&parx[swd_edge_lane(e)], 1u); // Hx * z-string
            }
            BP4T(5) // Hx edges: logaddexp + store + parity flip each
            const double num_hz = (hsel & 2) ? bp4_log1pexp(-1. * llrz_hz, xt) : 0.0;
            BP4T(4)
#pragma unroll 1
            for (int k = 0; k < ((hsel & 2) ? dz : 0); ++k) {
                const uint32_t e = pick(ez, k);
                const double c = msgz[swd_edge_slot(e)];
                const double aa = llrx_hx - c, bb = llry_all - c;
                msgz[swd_edge_slot(e)] = num_hz - bp4_logaddexp(-1. * aa, -1. * bb, xt);
                if (bx) atomicXor(&parz[swd_edge_lane(e)], 1u); // Hz * x-string
            }
            BP4T(6) // Hz edges
#else
#pragma unroll
            for (int k = 0; k < DM; ++k)
                if (k < dx && (hsel & 1)) {
                    const double aa = llrz_hz - cx[k], bb = llry_all - cx[k];
                    msgx[swd_edge_slot(ex[k])] = num_hx - bp4_logaddexp(-1. * aa, -1. * bb, xt);
                    if (bz) atomicXor(&parx[swd_edge_lane(ex[k])], 1u); // Hx * z-string
                }
            const double num_hz = bp4_log1pexp(-1. * llrz_hz, xt);
#pragma unroll
            for (int k = 0; k < DM; ++k)
                if (k < dz && (hsel & 2)) {
                    const double aa = llrx_hx - cz[k], bb = llry_all - cz[k];
                    msgz[swd_edge_slot(ez[k])] = num_hz - bp4_logaddexp(-1. * aa, -1. * bb, xt);
                    if (bx) atomicXor(&parz[swd_edge_lane(ez[k])], 1u); // Hz * x-string
                }
#endif
        }
        __syncthreads();
        BP4T(7) // barrier behind the node pass
#ifdef SWD_BP4PROF
        ++q_iters;
#endif
        }
    if (SWD_BP4_PRIO_IT >= 0) __builtin_amdgcn_s_setprio(0);
    // the next unit's ticket: drawn as soon as this decode's iterations are over, so that the atomic's round trip overlaps the stores below
    if (a.ticket && tid == 0) tk_next = gridDim.x + atomicAdd(a.ticket, 1u);
    if (!conv) {
        bool unsat = false;
        if (a.max_iter > 0) {
            for (int l = tid; l < mx; l += NT) if (parx[l] != 0u) unsat = true;
            for (int l = tid; l < mz; l += NT) if (parz[l] != 0u) unsat = true;
        } else unsat = true;
        const bool any = bp4_block_any(unsat, s, NT >> 6);
        iters = a.max_iter;
        conv = any ? 0 : 1;
    }
    // the posteriors of the last update: an output when the caller asked for them, otherwise only the OSD kernel reads them -- a decode
    // that converged (all but a few per thousand at the notebooks' noise rates) stores nothing (24 n bytes per decode, 226 MB per
    // 65 536-decode launch of the 144-qubit code before)
    if (p_set && (a.lpr_wanted || (!conv && !camel_run && a.osd_order >= 0))) { lpr_b[vt] = p_x; lpr_b[n + vt] = p_y; lpr_b[2 * n + vt] = p_z; }
    if (camel_run) {
        uint8_t *dst = a.camel_dec + (int64_t)unit * 2 * n;
        for (int v = tid; v < n; v += NT) { dst[v] = decx[v]; dst[n + v] = decz[v]; }
        if (tid == 0) {
            double pm = 0.0; // cal_pm (bp4_osd.pyx:249-258), summed in qubit order
            if (conv)
                for (int v = 0; v < n; ++v) {
                    if (decx[v] && decz[v]) pm += a.llr_y[v];
                    else if (decx[v]) pm += a.llr_x[v];
                    else if (decz[v]) pm += a.llr_z[v];
                }
            a.camel_pm[unit] = pm;
            a.camel_st[2 * unit] = conv; a.camel_st[2 * unit + 1] = iters;
        }
        continue;
    }
    uint8_t *out_b = a.out + (int64_t)b * 2 * n;
    if (a.bp_dec)
        for (int v = tid; v < n; v += NT) { a.bp_dec[(int64_t)b * 2 * n + v] = decx[v]; a.bp_dec[(int64_t)b * 2 * n + n + v] = decz[v]; }
    int exit_class = SWD_EXIT_PRE;
    if (conv || a.osd_order < 0) {
        for (int v = tid; v < n; v += NT) { out_b[v] = decx[v]; out_b[n + v] = decz[v]; }
        if (a.osd0 && conv)
            for (int v = tid; v < n; v += NT) { a.osd0[(int64_t)b * 2 * n + v] = decx[v]; a.osd0[(int64_t)b * 2 * n + n + v] = decz[v]; }
        if (!conv) exit_class = SWD_EXIT_NO_OSD;
    } else {
        exit_class = SWD_EXIT_OSD; // finished by bp4_osd_kernel from the posteriors stored above
        if (tid == 0) a.osd_list[atomicAdd(a.osd_count, 1u)] = b;
    }
    if (tid == 0) {
        int32_t *st = a.stats + (int64_t)b * SWD_STAT_WORDS;
        st[0] = exit_class | (conv ? SWD_STATUS_CONVERGE : 0);
        st[1] = iters; st[2] = iters; st[3] = 0; st[4] = n; st[5] = mx + mz; st[6] = gx.E + gz.E; st[7] = 0;
    }
    BP4T(8) // exit test, result stores
#ifdef SWD_BP4PROF
    ++q_units;
#endif
    } // next unit
#ifdef SWD_BP4PROF
    if ((tid & 63) == 0 && (blockIdx.x % 97) == 0)
        printf("bp4prof block %d wave %d: units %lld node passes %lld | reset+init %lld | check passes %lld | flags+barrier %lld | node head %lld | log1pexp x2 %lld | Hx edges %lld | Hz edges %lld | end barrier %lld | tail %lld cycles\n",
               (int)blockIdx.x, tid >> 6, q_units, q_iters, q_[0], q_[1], q_[2], q_[3], q_[4], q_[5], q_[6], q_[7], q_[8]);
#endif
}

// The decodes the BP kernel queued (bp4_osd.pyx:212-219: osd('x') and osd('z') when BP did not converge), one per workgroup at a time:
// ordering keys from the stored posteriors, elimination + higher-order sweep per basis (osd_run), the Z string from Hx and the X string
// from Hz.  Same LDS layout as the BP kernel.
template <int NT, int DM>
__global__ void __launch_bounds__(NT) bp4_osd_kernel(const SwdBp4Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const SwdGraphDev &gx = a.gx, &gz = a.gz;
    const int tid = threadIdx.x, n = gx.n, mx = gx.m, mz = gz.m;
    uint8_t *sxo = (uint8_t *)(smem + a.L.off_sxo), *szo = (uint8_t *)(smem + a.L.off_szo);
    Lds s;
    s.scratch = smem; s.msg = (double *)smem; s.hard = (uint8_t *)(smem + a.L.off_hard);
    s.flags = (int *)(smem + a.L.off_misc); s.scal = s.flags + 32; s.dbl = (double *)(s.scal + 32); s.iaux = (int *)(s.dbl + 24);
    s.ctid = s.vtid = threadIdx.x;
    const int count = (int)*a.osd_count;
    for (int q = blockIdx.x; q < count; q += gridDim.x) {
        const int b = a.osd_list[q];
        s.fpar = 0;
        const uint8_t *sx_b = a.sx + (int64_t)b * mx, *sz_b = a.sz + (int64_t)b * mz;
        const double *lpr_b = a.lpr + (int64_t)b * 3 * n;
        uint8_t *out_b = a.out + (int64_t)b * 2 * n;
        __syncthreads(); // the previous unit is done with LDS
        for (int r = tid; r < mx; r += NT) sxo[r] = sx_b[r] ? 1 : 0;
        for (int r = tid; r < mz; r += NT) szo[r] = sz_b[r] ? 1 : 0;
        int rowadds = 0;
        SwdDecodeParams P{};
        P.osd_method = a.osd_method; P.osd_order = a.osd_order;
        long long t0, t1;
        // osd('x'): Hx, synd_x -> Z string; osd('z'): Hz, synd_z -> X string (bp4_osd.pyx:261-296)
        for (int basis = 0; basis < 2; ++basis) {
            const SwdGraphDev &g = basis == 0 ? gx : gz;
            const SwdLdsLayout &L = basis == 0 ? a.Lx : a.Lz;
            uint64_t *key = (uint64_t *)s.scratch;
            uint16_t *idx = (uint16_t *)(s.scratch + L.off_idx);
            s.aux = s.scratch + L.off_aux; // the OSD arrays of this basis' layout
            __syncthreads();
            for (int v = tid; v < L.npad; v += NT) {
                if (v < n) {
                    const double lx = lpr_b[v], ly = lpr_b[n + v], lz = lpr_b[2 * n + v];
                    const double post = basis == 0 ? bp4_log1pexp(-1. * lx) - bp4_logaddexp(-1. * ly, -1. * lz)
                                                   : bp4_log1pexp(-1. * lz) - bp4_logaddexp(-1. * ly, -1. * lx);
                    key[v] = f2key(post);
                    idx[v] = (uint16_t)v;
                } else { key[v] = ~0ull; idx[v] = 0xFFFF; }
            }
            __syncthreads();
            int ra;
            uint8_t *o0 = a.osd0 ? a.osd0 + (int64_t)b * 2 * n + (basis == 0 ? n : 0) : nullptr;
            osd_run<NT, DM, false, true>(g, L, P, s, basis == 0 ? sxo : szo, o0, ra, t0, t1); // (the scratch region is LDS: check matrices of up to 256 rows take the four-wave elimination)
            rowadds += ra;
            uint8_t *dst = out_b + (basis == 0 ? n : 0);
            for (int v = tid; v < n; v += NT) dst[v] = s.hard[v];
            __syncthreads();
        }
        if (tid == 0) a.stats[(int64_t)b * SWD_STAT_WORDS + 7] = rowadds;
    }
}

// syndrome weight of every decode (both bases), for the start order of a launch: one wave per decode
__global__ void __launch_bounds__(256) bp4_weight_kernel(const uint8_t *sx, const uint8_t *sz, int mx, int mz, int B, uint32_t *wt) {
    const int b = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    const uint8_t *px = sx + (int64_t)b * mx, *pz = sz + (int64_t)b * mz;
    int c = 0;
    for (int r = lane; r < mx; r += 64) c += px[r] ? 1 : 0;
    for (int r = lane; r < mz; r += 64) c += pz[r] ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if (lane == 0) wt[b] = (uint32_t)min(c, 1023);
}

// camel_decode's choice among the four runs of a shot: the converged run of smallest path metric, strict < keeps
// the earliest (bp4_osd.pyx:237-243); converge = min_pm < 9999 (:244-245); bp_iteration = that of the last run.
// A shot without a converged run returns the zero vectors of a new reference object.
__global__ void __launch_bounds__(256) bp4_camel_select(int n, const uint8_t *dec, const double *pm, const int32_t *st,
                                                        uint8_t *out, int32_t *stats, double *min_pm) {
    const int b = blockIdx.x;
    double best = 10000.0;
    int arg = -1;
    for (int v = 0; v < 4; ++v)
        if (st[2 * (4 * b + v)] && pm[4 * b + v] < best) { best = pm[4 * b + v]; arg = v; }
    uint8_t *o = out + (int64_t)b * 2 * n;
    const uint8_t *src = dec + (int64_t)(4 * b + (arg < 0 ? 0 : arg)) * 2 * n;
    for (int i = threadIdx.x; i < 2 * n; i += 256) o[i] = arg < 0 ? (uint8_t)0 : src[i];
    if (threadIdx.x == 0) {
        int32_t *s = stats + (int64_t)b * SWD_STAT_WORDS;
        const int it = st[2 * (4 * b + 3) + 1];
        s[0] = SWD_EXIT_PRE | (best < 9999.0 ? SWD_STATUS_CONVERGE : 0);
        s[1] = it; s[2] = it; s[3] = 0; s[4] = n; s[5] = arg; s[6] = 0; s[7] = 0;
        if (min_pm) min_pm[b] = best;
    }
}

} // namespace swd
