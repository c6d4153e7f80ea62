// Guessing decoders on the device (included inside namespace swd by swd_osdw_kernel.h):
//   bpgdg_decoder, single-thread gdg()   /root/reference/src/bp_guessing_decoder.pyx:160-442
//   bpgd_decoder, gd()                   /root/reference/src/bp_guessing_decoder.pyx:473-571
//   bp_history_decoder                   /root/reference/src/bp_guessing_decoder.pyx:5-158
// with the BPGD engine of /root/reference/src/include/bpgd.cpp (reset 199-239, min_sum_log 97-197,
// vn_set_value 51-80, peel 13-49, set_masks 241-248, get_pm 250-256, decimate_vn_reliable 258-286).
//
// One workgroup per shot.  The decimation tree of a shot is explored in exactly the reference's
// order (main branch, then the saved snapshots in stack order with the min_converge_depth pruning),
// because that order decides which hypothesis wins; the parallelism is inside each BP block and
// across shots.  The multi-thread ensemble of the reference is racy and is not a
// parity target.
//
// The sub-matrix "pcm" of BPGD::reset (first new_n columns in sorted order) is never built: the
// selected columns keep their slots in the window graph, `pos_lv[j]` = column at sorted position j,
// and every scan that the reference does "for vn in range(new_n)" runs over positions.
#pragma once

struct GdgLds {
    uint16_t *pos_lv;  // [new_n] sorted position -> column
    uint16_t *plist;   // [new_n] scratch list of positions (ordered sums, decimation queue)
    int16_t *dec_vn;   // [64] snapshot stack: guessed position
    int16_t *alt_depth;// [64]
    uint8_t *best_err; // [new_n] bpgd_error
    uint8_t *bp_hard;  // [n] pre-processing BP decisions (returned if BPGD::reset fails)
    int8_t *dec_val;   // [64]
    uint8_t *cat;      // [new_n] select_vn classification per position
};

__device__ __forceinline__ void gdg_bind(GdgLds &G, char *smem, const SwdLdsLayout &L, int n, int new_n) {
    char *b = smem + L.off_gdg;
    G.pos_lv = (uint16_t *)b; b += new_n * 2;
    G.plist = (uint16_t *)b; b += new_n * 2;
    G.dec_vn = (int16_t *)b; b += 64 * 2;
    G.alt_depth = (int16_t *)b; b += 64 * 2;
    G.best_err = (uint8_t *)b; b += new_n;
    G.bp_hard = (uint8_t *)b; b += n;
    G.dec_val = (int8_t *)b; b += 64;
    G.cat = (uint8_t *)b;
}

// vn_set_value for an arbitrary VN incl. the "already decided" pre-check of BPGD::vn_set_value
// (bpgd.cpp:52-55).  Wave 0 only.  Returns true on failure.
__device__ __forceinline__ bool gdg_set_value_wave(const SwdGraphDev &g, Lds &s, int vn, int value) {
    const int cur = s.vn_val[vn];
    if (cur != -1) return cur != value;
    return vn_set_value_wave(g, s, vn, value);
}

// lexicographic (key, pos) minimum over the block; every thread gets the result
template <int NT>
__device__ __forceinline__ void block_argmin(double &key, int &pos, Lds &s) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const double ok = __shfl_xor(key, d, 64);
        const int op = __shfl_xor(pos, d, 64);
        if (ok < key || (ok == key && op < pos)) { key = ok; pos = op; }
    }
    double *wk = s.dbl + 4; // [16]
    int *wp = s.iaux;       // [16]
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { wk[threadIdx.x >> 6] = key; wp[threadIdx.x >> 6] = pos; }
    __syncthreads();
    double bk = wk[0];
    int bp = wp[0];
#pragma unroll
    for (int w = 1; w < NT / 64; ++w) {
        const double ok = wk[w];
        const int op = wp[w];
        if (ok < bk || (ok == bk && op < bp)) { bk = ok; bp = op; }
    }
    key = bk; pos = bp;
    __syncthreads();
}

// Compact the still-active selected VNs (position order) into s.lv and (re)build the per-phase
// register caches.  Messages are untouched (they persist across decimation steps, bpgd.cpp:97-197).
template <int NT, int VF, int DM, int KG>
__device__ __forceinline__ int gdg_build_caches(const SwdGraphDev &g, Lds &s, const GdgLds &G, VnCache<VF, DM> &vc,
                                                CnCache<KG> &cn) {
    const int tid = threadIdx.x, m = g.m, new_n = g.new_n;
    const int ch = (new_n + NT - 1) / NT;
    const int j0 = tid * ch, j1 = min(new_n, j0 + ch);
    int cnt = 0;
    for (int j = j0; j < j1; ++j) cnt += (s.vn_val[G.pos_lv[j]] == -1) ? 1 : 0;
    int nlive;
    int pos = block_exscan<NT>(cnt, s, nlive);
    for (int j = j0; j < j1; ++j) {
        const int v = G.pos_lv[j];
        if (s.vn_val[v] == -1) s.lv[pos++] = (uint16_t)v;
    }
    for (int l = tid; l < m; l += NT)
        if (s.cn_val[l] >= 0) {
            uint64_t mk = s.livemask[l];
            int k = 0;
            while (mk) {
                const int j = __ffsll((long long)mk) - 1;
                mk &= mk - 1;
                s.lslot[k * m + l] = (uint16_t)(s.jptr[j] + l);
                ++k;
            }
        }
    __syncthreads();
    vn_cache_load<NT, VF, DM, false>(g, s, nlive, vc);
    cn_cache_load<NT, KG, false>(g, s, true, s.ctid < g.m ? s.ctid : -1, 0, 1, cn);
    __syncthreads();
    return nlive;
}

// get_pm (bpgd.cpp:250-256): sum of llr_prior over error == 1 in POSITION order
template <int NT>
__device__ __forceinline__ double gdg_get_pm(const SwdGraphDev &g, Lds &s, const GdgLds &G) {
    const int new_n = g.new_n;
    const int ch = (new_n + NT - 1) / NT;
    const int j0 = threadIdx.x * ch, j1 = min(new_n, j0 + ch);
    int cnt = 0;
    for (int j = j0; j < j1; ++j) cnt += s.hard[G.pos_lv[j]] ? 1 : 0;
    int total;
    int pos = block_exscan<NT>(cnt, s, total);
    for (int j = j0; j < j1; ++j)
        if (s.hard[G.pos_lv[j]]) G.plist[pos++] = G.pos_lv[j];
    __syncthreads();
    double *dres = s.dbl;
    if (threadIdx.x == 0) {
        double pm = 0;
        for (int i = 0; i < total; ++i) pm += g.llr[G.plist[i]];
        *dres = pm;
    }
    __syncthreads();
    return *dres;
}

// Snapshot record in HBM: vn state by position | cn_val | cn_deg | livemask
__device__ __forceinline__ int64_t gdg_snap_bytes(int m, int new_n) { return ((new_n + 2 * m + 7) & ~7) + 8 * (int64_t)m; }

template <int NT>
__device__ __forceinline__ void gdg_snap_save(const SwdGraphDev &g, Lds &s, const GdgLds &G, uint8_t *rec) {
    const int m = g.m, new_n = g.new_n;
    for (int j = threadIdx.x; j < new_n; j += NT) rec[j] = (uint8_t)s.vn_val[G.pos_lv[j]];
    uint64_t *lm = (uint64_t *)(rec + ((new_n + 2 * m + 7) & ~7));
    for (int l = threadIdx.x; l < m; l += NT) {
        rec[new_n + l] = (uint8_t)s.cn_val[l];
        rec[new_n + m + l] = s.cn_deg[l];
        lm[l] = s.livemask[l];
    }
}

// set_masks (bpgd.cpp:241-248) minus the message re-initialisation, which the caller does after
// the branch decision and peeling (only messages of still-active VNs are ever read)
template <int NT>
__device__ __forceinline__ void gdg_snap_load(const SwdGraphDev &g, Lds &s, const GdgLds &G, const uint8_t *rec) {
    const int m = g.m, new_n = g.new_n;
    __syncthreads();
    for (int j = threadIdx.x; j < new_n; j += NT) {
        const int v = G.pos_lv[j];
        const int8_t val = (int8_t)rec[j];
        s.vn_val[v] = val;
        s.hard[v] = (val == 1) ? 1 : 0;
    }
    const uint64_t *lm = (const uint64_t *)(rec + ((new_n + 2 * m + 7) & ~7));
    for (int l = threadIdx.x; l < m; l += NT) {
        s.cn_val[l] = (int8_t)rec[new_n + l];
        s.cn_deg[l] = rec[new_n + m + l];
        s.livemask[l] = lm[l];
    }
    __syncthreads();
}

// bpgdg_decoder.select_vn (bp_guessing_decoder.pyx:340-442).  Returns -1 on failure, else 0.
// Classification of a position is independent of the scan order (num_flip only looks at active
// neighbour checks, and a live VN has no inactive ones), the decimations are applied by wave 0 in
// position order so that a contradiction stops exactly where the reference's scan stops.
template <int NT>
__device__ __forceinline__ int gdg_select_vn(const SwdGraphDev &g, const SwdDecodeParams &P, Lds &s, const GdgLds &G,
                                             const double *hist_b, bool side, int depth, int min_converge_depth,
                                             int &used_guess, uint8_t *snap_b) {
    const int tid = threadIdx.x, n = g.n, new_n = g.new_n;
    const double A = side ? 0.0 : -3.0;
    double A_sum = side ? -10.0 : -12.0;
    if (depth == 0) A_sum = -16.0;
    const double C = 30.0, D = 3.0;
    double best_all = 10000.0, best_neg = 10000.0;
    int pos_all = 0x7fffffff, pos_neg = 0x7fffffff;
    for (int j = tid; j < new_n; j += NT) {
        const int v = G.pos_lv[j];
        uint8_t cat = 0;
        if (s.vn_val[v] == -1) {
            const int deg = g.col_deg[v];
            if (deg > 2) {
                int num_flip = 0;
                for (int k = 0; k < deg; ++k) {
                    const uint32_t e = g.vn_edge[k * n + v];
                    const int l = swd_edge_lane(e);
                    if (s.cn_val[l] >= 0 && s.par[l] != 0u) ++num_flip;
                }
                bool smaller_A = true, all_neg = true, larger_C = true, larger_D = true;
                double hsum = 0.0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const double llr = hist_b[i * n + v];
                    hsum += llr;
                    if (llr < C) larger_C = false;
                    if (llr < D) larger_D = false;
                    if (llr > A) smaller_A = false;
                    if (llr > 0.0) all_neg = false;
                }
                const bool aggr = !P.low_error_mode;
                if (aggr && larger_C && depth < 4) cat = 1;                 // decide 0
                else if (aggr && num_flip >= 3 && larger_D) cat = 1;        // decide 0
                else if (aggr && smaller_A && hsum < A_sum) cat = 2;        // decide 1
                else {
                    cat = 3;
                    // positions ascend inside a thread, so strict "<" keeps the earliest minimum
                    if (hsum < best_all) { best_all = hsum; pos_all = j; }
                    if (all_neg && hsum < best_neg) { best_neg = hsum; pos_neg = j; }
                }
            }
        }
        G.cat[j] = cat;
    }
    // thresholds of the reference are strict "<" against the running minimum initialised to 10000:
    // a candidate with sum >= 10000 never becomes the minimum
    if (!(best_all < 10000.0)) { best_all = 10000.0; pos_all = 0x7fffffff; }
    if (!(best_neg < 10000.0)) { best_neg = 10000.0; pos_neg = 0x7fffffff; }
    block_argmin<NT>(best_all, pos_all, s);
    block_argmin<NT>(best_neg, pos_neg, s);
    // ordered queue of the aggressive decimations
    {
        const int ch = (new_n + NT - 1) / NT;
        const int j0 = tid * ch, j1 = min(new_n, j0 + ch);
        int cnt = 0;
        for (int j = j0; j < j1; ++j) cnt += (G.cat[j] == 1 || G.cat[j] == 2) ? 1 : 0;
        int total;
        int pos = block_exscan<NT>(cnt, s, total);
        for (int j = j0; j < j1; ++j)
            if (G.cat[j] == 1 || G.cat[j] == 2) G.plist[pos++] = (uint16_t)j;
        __syncthreads();
        if (tid < 64) {
            bool bad = false;
            for (int i = 0; i < total && !bad; ++i) {
                const int j = G.plist[i];
                bad = gdg_set_value_wave(g, s, G.pos_lv[j], G.cat[j] == 2 ? 1 : 0);
            }
            if (!bad) bad = peel_wave(g, s);
            if (tid == 0) s.scal[1] = bad ? 1 : 0;
        }
        __syncthreads();
        if (s.scal[1]) return -1;
    }
    int guess_pos, favor;
    if (pos_neg != 0x7fffffff) { guess_pos = pos_neg; favor = 1; }
    else { guess_pos = pos_all; favor = (best_all > 0) ? 0 : 1; }
    bool guess = true;
    if (depth > min_converge_depth) guess = false;
    if (!side && depth >= P.max_side_depth) guess = false;
    if (side && depth > P.max_tree_depth) guess = false;
    if (guess && used_guess < P.max_guess) {
        if (tid == 0) {
            G.dec_val[used_guess] = (int8_t)(1 - favor);
            G.dec_vn[used_guess] = (int16_t)(guess_pos == 0x7fffffff ? -1 : guess_pos);
            G.alt_depth[used_guess] = (int16_t)(depth + 1);
        }
        gdg_snap_save<NT>(g, s, G, snap_b + (int64_t)used_guess * gdg_snap_bytes(g.m, new_n));
        used_guess += 1;
    }
    if (guess_pos == 0x7fffffff) return -1; // no candidate left (the reference would index vn_mask[-1])
    __syncthreads();
    if (tid < 64) {
        bool bad = gdg_set_value_wave(g, s, G.pos_lv[guess_pos], favor);
        if (!bad) bad = peel_wave(g, s);
        if (tid == 0) s.scal[1] = bad ? 1 : 0;
    }
    __syncthreads();
    return s.scal[1] ? -1 : 0;
}

// BPGD::decimate_vn_reliable (bpgd.cpp:258-286): largest |posterior of slot 3| among active VNs
template <int NT>
__device__ __forceinline__ int gdg_decimate_reliable(const SwdGraphDev &g, Lds &s, const GdgLds &G, const double *hist_b) {
    const int tid = threadIdx.x, n = g.n, new_n = g.new_n;
    double best = 0.0; // stored negated so that block_argmin (a minimum) finds the largest magnitude
    int bpos = 0x7fffffff;
    for (int j = tid; j < new_n; j += NT) {
        const int v = G.pos_lv[j];
        if (s.vn_val[v] != -1) continue;
        const double a = fabs(hist_b[3 * n + v]);
        if (a > -best) { best = -a; bpos = j; } // strict ">" (bpgd.cpp:270), earliest position wins ties
    }
    if (!(best < 0.0)) { best = 0.0; bpos = 0x7fffffff; }
    block_argmin<NT>(best, bpos, s);
    if (bpos == 0x7fffffff) return -1;
    const int v = G.pos_lv[bpos];
    const int val = (hist_b[3 * n + v] > 0) ? 0 : 1;
    __syncthreads();
    if (tid < 64) {
        bool bad = gdg_set_value_wave(g, s, v, val);
        if (!bad) bad = peel_wave(g, s);
        if (tid == 0) s.scal[1] = bad ? 1 : 0;
    }
    __syncthreads();
    return s.scal[1] ? -1 : 0;
}

// bpgdg_decoder.decode / bpgd_decoder.decode / bp_history_decoder for one syndrome.  On return
// s.hard[0..n) is the returned vector.
template <int NT, int VF, int DM, int KG>
__device__ __forceinline__ void decode_window_gdg(const SwdGraphDev &g, const SwdLdsLayout &L, const SwdDecodeParams &P, Lds &s,
                                                  const uint8_t *synd, double *hist_b, uint8_t *snap_b, WinResult &R) {
    const int tid = threadIdx.x, m = g.m, n = g.n, new_n = g.new_n;
    GdgLds G;
    gdg_bind(G, s.scratch, L, n, new_n);
#pragma unroll
    for (int i = 0; i < 9; ++i) R.t[i] = 0;
    R.t[0] = wall_clock64();
    for (int l = tid; l < m; l += NT) {
        const int d = g.row_deg[l];
        s.cn_val[l] = (int8_t)(synd[g.perm[l]] ? 1 : 0);
        s.cn_deg[l] = (uint8_t)d;
        s.cn_deg0[l] = (uint8_t)d;
        s.livemask[l] = (d >= 64) ? ~0ull : ((1ull << d) - 1ull);
    }
    for (int v = tid; v < n; v += NT) { s.vn_val[v] = -1; s.hard[v] = 0; }
    for (int j = tid; j <= g.K; j += NT) s.jptr[j] = g.jptr[j];
    if (P.zero_hist)
        for (int i = tid; i < 4 * n; i += NT) hist_b[i] = 0.0;
    __syncthreads();
    VnCache<VF, DM> vc;
    CnCache<KG> cn;
    vn_cache_load<NT, VF, DM, true>(g, s, n, vc);
    bp_init<VF, DM>(s, vc);
    cn_cache_load<NT, KG, true>(g, s, false, s.ctid < g.m ? s.ctid : -1, 0, 1, cn);
    __syncthreads();
    int it = 0;
    R.conv = 0; R.pm = 0.0; R.pre_it = R.post_it = 0;
    R.live_vn = n; R.live_cn = m; R.live_e = g.E; R.osd_rowadds = 0;
    R.t[1] = wall_clock64();
    // bp_history_decoder.bp_decode_llr (bp_guessing_decoder.pyx:48-139)
    R.conv = bp_run<NT, VF, DM, KG, true>(g, P, s, P.pre_iter, n, vc, cn, hist_b, it, P.alpha, false);
    R.pre_it = it; R.total_it = it;
    R.t[2] = wall_clock64();
    if (R.conv) { R.exit_class = SWD_EXIT_PRE; return; }
    if (P.kind == 3) { R.exit_class = SWD_EXIT_NO_OSD; return; }

    // ---- order by summed history, keep the first new_n columns (pyx:259-271, bpgd.cpp:199-239)
    uint64_t *key = (uint64_t *)s.scratch;
    uint16_t *idx = (uint16_t *)(s.scratch + L.off_idx);
    __syncthreads();
    for (int v = tid; v < L.npad; v += NT) {
        if (v < n) {
            const double sum = ((hist_b[v] + hist_b[n + v]) + hist_b[2 * n + v]) + hist_b[3 * n + v];
            key[v] = f2key(sum);
            idx[v] = (uint16_t)v;
        } else { key[v] = ~0ull; idx[v] = 0xFFFF; }
    }
    for (int v = tid; v < n; v += NT) G.bp_hard[v] = s.hard[v];
    __syncthreads();
    // BPGD keeps its own 4-slot history (bpgd.cpp:357-358, never initialised by the reference) and
    // restarts at slot 0 in every block (bpgd.cpp:166): with fewer than 4 iterations per block
    // the upper slots are never written.  The oracle defines them as zero; the pre-phase
    // values this buffer still holds there must not leak into select_vn / decimate_vn_reliable.
    if (P.max_iter_per_step < 4)
        for (int i = P.max_iter_per_step * n + tid; i < 4 * n; i += NT) hist_b[i] = 0.0;
    sort_pairs<NT>(key, idx, L.npad);
    for (int i = tid; i < n; i += NT) {
        const int v = idx[i];
        if (i < new_n) G.pos_lv[i] = (uint16_t)v;
        else { s.vn_val[v] = 0; s.hard[v] = 0; G.bp_hard[v] = 0; }
    }
    __syncthreads();
    bool dead_unsat = false;
    for (int l = tid; l < m; l += NT) {
        const int d = g.row_deg[l];
        uint64_t mk = 0;
        int cntl = 0;
        for (int j = 0; j < d; ++j) {
            const int v = g.row_col[s.jptr[j] + l];
            if (s.vn_val[v] < 0) { mk |= 1ull << j; ++cntl; }
        }
        s.livemask[l] = mk;
        s.cn_deg[l] = (uint8_t)cntl;
        if (cntl == 0) { // degree-0 check: deactivated without a contradiction test (bpgd.cpp:210-217)
            if (s.cn_val[l] != 0) dead_unsat = true;
            s.cn_val[l] = -1;
        }
    }
    for (int j = tid; j < new_n; j += NT) s.hard[G.pos_lv[j]] = 0; // error[] = 0 (bpgd.cpp:234)
    dead_unsat = block_any<NT>(dead_unsat, s);
    R.t[3] = wall_clock64();
    if (tid < 64) {
        const bool bad = peel_wave(g, s);
        if (tid == 0) s.scal[1] = bad ? 1 : 0;
    }
    __syncthreads();
    if (s.scal[1]) { // BPGD::reset failed: decode returns the BP vector with cols[new_n:] zeroed
        for (int v = tid; v < n; v += NT) s.hard[v] = G.bp_hard[v];
        __syncthreads();
        R.exit_class = SWD_EXIT_FAIL_PEEL;
        return;
    }
    int nlive = gdg_build_caches<NT, VF, DM, KG>(g, s, G, vc, cn);
    bp_init<VF, DM>(s, vc);
    __syncthreads();

    double min_pm = 10000.0;
    int used_guess = 0, min_converge_depth = P.max_step, converge = 0, blocks = 0;
    const bool gdg = (P.kind == 1);
    // ---- phase 1: main branch
    for (int depth = 0; depth < P.max_step; ++depth) {
        if (depth > 0) nlive = gdg_build_caches<NT, VF, DM, KG>(g, s, G, vc, cn);
        const int cv = bp_run<NT, VF, DM, KG, false>(g, P, s, P.max_iter_per_step, nlive, vc, cn, hist_b, it, P.gdg_factor, dead_unsat);
        ++blocks; R.post_it += it;
        if (cv) {
            converge = 1; min_converge_depth = depth;
            min_pm = gdg_get_pm<NT>(g, s, G);
            for (int j = tid; j < new_n; j += NT) G.best_err[j] = s.hard[G.pos_lv[j]];
            break;
        }
        const int rc = gdg ? gdg_select_vn<NT>(g, P, s, G, hist_b, false, depth, min_converge_depth, used_guess, snap_b)
                           : gdg_decimate_reliable<NT>(g, s, G, hist_b);
        if (rc == -1) break;
    }
    if (!converge)
        for (int j = tid; j < new_n; j += NT) G.best_err[j] = s.hard[G.pos_lv[j]];
    __syncthreads();
    // ---- phase 2: side branches in stack order (pyx:301-335)
    for (int i = 0; gdg && i < used_guess; ++i) {
        int depth = G.alt_depth[i];
        if (depth > min_converge_depth) continue;
        gdg_snap_load<NT>(g, s, G, snap_b + (int64_t)i * gdg_snap_bytes(m, new_n));
        if (tid < 64) {
            const int gp = G.dec_vn[i];
            bool bad = (gp < 0) ? true : gdg_set_value_wave(g, s, G.pos_lv[gp], G.dec_val[i]);
            if (!bad) bad = peel_wave(g, s);
            if (tid == 0) s.scal[1] = bad ? 1 : 0;
        }
        __syncthreads();
        if (s.scal[1]) continue;
        for (int j = 0; j < P.max_side_branch_step; ++j) {
            depth = G.alt_depth[i] + j;
            nlive = gdg_build_caches<NT, VF, DM, KG>(g, s, G, vc, cn);
            if (j == 0) { bp_init<VF, DM>(s, vc); __syncthreads(); } // set_masks re-initialises the messages
            const int cv = bp_run<NT, VF, DM, KG, false>(g, P, s, P.max_iter_per_step, nlive, vc, cn, hist_b, it, P.gdg_factor, dead_unsat);
            ++blocks; R.post_it += it;
            if (cv) {
                converge = 1;
                const double pm = gdg_get_pm<NT>(g, s, G);
                if (pm < min_pm) {
                    if (depth < min_converge_depth) min_converge_depth = depth;
                    for (int q = tid; q < new_n; q += NT) G.best_err[q] = s.hard[G.pos_lv[q]];
                    min_pm = pm;
                }
                break;
            }
            if (depth > min_converge_depth + 2) break;
            if (gdg_select_vn<NT>(g, P, s, G, hist_b, true, depth, min_converge_depth, used_guess, snap_b) == -1) break;
        }
        __syncthreads();
    }
    __syncthreads();
    for (int v = tid; v < n; v += NT) s.hard[v] = 0;
    __syncthreads();
    for (int j = tid; j < new_n; j += NT) s.hard[G.pos_lv[j]] = G.best_err[j];
    __syncthreads();
    R.conv = converge; R.pm = min_pm; R.total_it = R.pre_it + R.post_it;
    R.live_vn = used_guess; R.live_cn = blocks; R.live_e = min_converge_depth;
    R.exit_class = SWD_EXIT_POST;
    R.t[5] = wall_clock64();
}
