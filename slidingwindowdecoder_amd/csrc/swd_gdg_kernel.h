// Guessing decoders on the device (included inside namespace swd by swd_osdw_kernel.h):
//   bpgdg_decoder, single-thread gdg()   /root/reference/src/bp_guessing_decoder.pyx:160-442
//   bpgd_decoder, gd()                   /root/reference/src/bp_guessing_decoder.pyx:473-571
//   bp_history_decoder                   /root/reference/src/bp_guessing_decoder.pyx:5-158
// with the BPGD engine of /root/reference/src/include/bpgd.cpp (reset 199-239, min_sum_log 97-197,
// vn_set_value 51-80, peel 13-49, set_masks 241-248, get_pm 250-256, decimate_vn_reliable 258-286).
//
// The decimation tree of a (shot, window) is explored with the RESULT of the reference's order (main branch,
// then the saved snapshots in stack order with the min_converge_depth pruning), because that order decides
// which hypothesis wins.  Two execution forms:
//   serial    one workgroup walks the whole tree in that order (decode_window_gdg, phase 2 below);
//   parallel  the owner workgroup runs the main branch, then every saved snapshot becomes a task on the
//             persistent grid's queue (gdg_run_task, any workgroup): a side branch run with the most permissive
//             pruning state (min_converge_depth as it was after the main branch) does a superset of what the
//             serial order would do with it -- BP blocks and decimations of a branch depend on nothing outside
//             the branch; only "is a snapshot pushed here" and "stop here" depend on the evolving
//             min_converge_depth / snapshot count.  Every step's outcome is recorded and the owner replays the
//             serial bookkeeping (gdg_replay: pruning, capacity, strict-< arg-min in stack order) over the
//             records, so the returned vector, converge flag and statistics are those of the serial order.
// The reference's multi-thread ensemble (bpgd.cpp:419-688, row a23) explores all leaves concurrently and keeps
// the smallest path metric; it is racy and no parity target.  `ensemble` replays the same
// records without pruning / capacity: every hypothesis counts, ties to the earliest in stack order.
//
// The sub-matrix "pcm" of BPGD::reset (first new_n columns in sorted order) is never built: the
// selected columns keep their slots in the window graph, `pos_lv[j]` = column at sorted position j,
// and every scan that the reference does "for vn in range(new_n)" runs over positions.
#pragma once

#define SWD_GDG_SLOTS 64     // snapshots per (shot, window) the parallel form can hold; beyond: serial fallback
#define SWD_GDG_MAXGUESS 160 // snapshot stack of the serial form (max_guess = 2 (2^D - 1) + S - D, bp_guessing_decoder.pyx:181: D = 6, S <= 40)
#define SWD_GDG_MAXSTEP 64   // max_side_branch_step the parallel form records
#define SWD_GDG_REC_BYTES (16 + 4 * SWD_GDG_MAXSTEP)
#define SWD_GDG_HDR_BYTES 256

// agent-scope (sc1) word accesses: everything one workgroup publishes for another goes through these
__device__ __forceinline__ uint32_t ag_ld(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ag_st(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// all stores of this workgroup are acknowledged; then a barrier (callers publish a counter / queue entry after it)
__device__ __forceinline__ void ag_publish_barrier() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
}

// Work items of the persistent grid in the parallel form (32-bit payloads of the launch's queue):
//   UNIT  (shot, window)   decode the window: pre-processing BP and the main branch; a tree with side branches is
//                          parked in a context (below) and the workgroup goes on to other items
//   SIDE  (context, slot)  one side branch of a parked tree
//   FINAL (context)        every branch of the tree is accounted for: pick the winner, commit, hand the shot on
// Nothing ever waits for another item: a unit is queued when its predecessor window has committed, a FINAL when the
// last branch has been replayed.
#define SWD_ITEM_UNIT 0u
#define SWD_ITEM_SIDE 1u
#define SWD_ITEM_FINAL 2u
#define SWD_ITEM_NONE 0xFFFFFFFEu
#define SWD_ITEM_EXIT 0xFFFFFFFFu
// a UNIT item holds the window in 8 bits and the shot in 22: plans with more windows / launches with more shots take the serial
// form (Plan::finalize, launch in swd_osdw.hip)
#define SWD_GDG_ITEM_MAX_WINDOWS 256
#define SWD_GDG_ITEM_MAX_SHOTS (1 << 22)
__device__ __forceinline__ uint32_t item_unit(int b, int wi) { return (SWD_ITEM_UNIT << 30) | ((uint32_t)wi << 22) | (uint32_t)b; }
__device__ __forceinline__ uint32_t item_side(int ctxid, int slot) { return (SWD_ITEM_SIDE << 30) | ((uint32_t)ctxid << 8) | (uint32_t)slot; }
__device__ __forceinline__ uint32_t item_final(int ctxid) { return (SWD_ITEM_FINAL << 30) | (uint32_t)ctxid; }

// Context of a parked tree in HBM (SwdGdgPar::ctx + id * ctx_stride; ids come from the free ring), 32-bit words:
//   hdr  0 lock  1 -  2 snapshot slots allocated  3 overflow
//        4 window  5 shot  6 dead_unsat  7 published bound on min_converge_depth (side branches prune against it)
//        scheduler state, touched only under the lock:  8 frontier i  9 nseq  10 used_guess  11 min_converge_depth
//        12 converge  13 BP blocks  14 iterations  15 best slot  16,17 min_pm  18 tasks launched  19 tasks finished
//        20,21 launched mask  22 FINAL queued  23 min_converge_depth after the main branch
//        24 pre-processing iterations  25 snapshots of the main branch  26..41 seq[64]: slots in serial stack order
//   pos  u16 pos_lv[new_n] (packed)            rec   SWD_GDG_SLOTS records of SWD_GDG_REC_BYTES:
//     w0 = alt_depth | dec_val << 8 | (dec_vn + 1) << 16
//     w1 = start_failed | done << 1 | pruned_before_start << 2 | nsteps << 8 | (conv_step + 1) << 16   (written last)
//     w2,w3 = path metric of the converged block               w4.. per step: iterations | fail_before_push << 8 |
//     fail_after_push << 9 | (child slot + 1) << 16
//   err   (SWD_GDG_SLOTS + 1) error vectors by position (slot SWD_GDG_SLOTS = the main branch's)
//   the snapshots of the tree: SwdGdgPar::csnap + id * csnap_stride
struct GdgCtx {
    uint32_t *hdr, *pos, *rec, *err;
    uint8_t *snap;
    int err_words; // words per error vector
};
__device__ __forceinline__ GdgCtx gdg_ctx(const SwdGdgPar &gp, int id) {
    uint8_t *b = gp.ctx + (int64_t)id * gp.ctx_stride;
    GdgCtx c;
    c.hdr = (uint32_t *)b; c.pos = (uint32_t *)(b + gp.off_pos); c.rec = (uint32_t *)(b + gp.off_rec);
    c.err = (uint32_t *)(b + gp.off_err); c.err_words = gp.err_stride >> 2;
    c.snap = gp.csnap + (int64_t)id * gp.csnap_stride;
    return c;
}
__device__ __forceinline__ uint32_t *gdg_rec(const GdgCtx &c, int slot) { return c.rec + slot * (SWD_GDG_REC_BYTES / 4); }

#ifdef SWD_GDG_DEBUG // diagnostic build: counters in the decoder's status array (words 1..)
#define GDG_COUNT(word, v) atomicAdd(&dbg_status[word], (uint32_t)(v))
#else
#define GDG_COUNT(word, v) do { } while (0)
#endif
#ifdef SWD_SELPROF // diagnostic build: 100 MHz ticks inside select_vn / the cache rebuild, summed per unit in s.scal[20..27]
#define SEL_T0() long long sel_t_ = wall_clock64()
#define SEL_T(k) do { const long long sel_n_ = wall_clock64(); if (threadIdx.x == 0) s.scal[20 + (k)] += (int)(sel_n_ - sel_t_); sel_t_ = sel_n_; } while (0)
#else
#define SEL_T0() do { } while (0)
#define SEL_T(k) do { } while (0)
#endif
#ifdef SWD_GDG_CHECKS // diagnostic build: invariant violations are reported in the status word (bits 8..) instead of followed
#define GDG_CHECK(cond, code) do { if (!(cond)) atomicOr(chk_status, 1u << (8 + (code))); } while (0)
#else
#define GDG_CHECK(cond, code) do { } while (0)
#endif

// multi-producer multi-consumer ring: r[0] head, r[1] tail, r[2], r[3] free for the user, then (sequence << 32 | payload)
__device__ __forceinline__ void ring_push(uint32_t *r, uint32_t mask, uint32_t payload) { // one thread
    const uint32_t t = atomicAdd(&r[1], 1u);
    unsigned long long *ring = (unsigned long long *)(r + 4);
    __hip_atomic_store(&ring[t & mask], ((unsigned long long)(t + 1u) << 32) | payload, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Blocking pop: the consumer takes the next sequence number unconditionally (no compare-and-swap race among idle
// consumers: with hundreds of idle workgroups contending for every item a CAS loop starved single workgroups for
// milliseconds) and waits on ITS OWN ring slot until a producer fills it.  Every waiter is eventually served: the
// workgroup that completes the launch's last unit pushes one SWD_ITEM_EXIT per workgroup.  A 20 s bound (100 MHz
// ticks) turns a bug into SWD_ITEM_EXIT + a status flag instead of a hung device.
__device__ __forceinline__ uint32_t ring_pop_wait(uint32_t *r, uint32_t mask, uint32_t *status, uint32_t *launch_fault) { // one thread
    const uint32_t h = atomicAdd(&r[0], 1u);
    unsigned long long *ring = (unsigned long long *)(r + 4);
    unsigned long long e;
    const long long t0 = wall_clock64();
    while ((uint32_t)((e = __hip_atomic_load(&ring[h & mask], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != h + 1u) {
        __builtin_amdgcn_s_sleep(16);
        if (wall_clock64() - t0 > 2000000000ll) { atomicOr(status, 2u); atomicOr(launch_fault, 2u); return 0xFFFFFFFFu; }
    }
    return (uint32_t)e;
}

__device__ __forceinline__ int ring_pop(uint32_t *r, uint32_t mask, uint32_t *payload) { // one thread; 0 if empty (free-context ring)
    for (;;) {
        const uint32_t h = ag_ld(&r[0]), t = ag_ld(&r[1]);
        if ((int32_t)(t - h) <= 0) return 0;
        if (atomicCAS(&r[0], h, h + 1u) != h) continue;
        unsigned long long *ring = (unsigned long long *)(r + 4);
        unsigned long long e;
        while ((uint32_t)((e = __hip_atomic_load(&ring[h & mask], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != h + 1u)
            __builtin_amdgcn_s_sleep(2);
        *payload = (uint32_t)e;
        return 1;
    }
}

struct GdgLds {
    uint16_t *pos_lv;  // [new_n] sorted position -> column
    uint16_t *plist;   // [new_n] scratch list of positions (ordered sums, decimation queue)
    int16_t *dec_vn;   // [SWD_GDG_MAXGUESS] snapshot stack: guessed position
    int16_t *alt_depth;// [SWD_GDG_MAXGUESS]
    uint8_t *best_err; // [new_n] bpgd_error
    uint8_t *bp_hard;  // [n] pre-processing BP decisions (returned if BPGD::reset fails)
    int8_t *dec_val;   // [SWD_GDG_MAXGUESS]
    uint8_t *cat;      // [new_n] select_vn classification per position
};

// smem: the workgroup's LDS block (NOT the scratch region, which large-graph kernels keep in HBM): gdg_lds_base
__device__ __forceinline__ char *gdg_lds_base(const Lds &s, const SwdLdsLayout &L) { return (char *)s.hard - L.off_hard; }
__device__ __forceinline__ void gdg_bind(GdgLds &G, char *smem, const SwdLdsLayout &L, int n, int new_n) {
    char *b = smem + L.off_gdg;
    G.pos_lv = (uint16_t *)b; b += new_n * 2;
    G.plist = (uint16_t *)b; b += new_n * 2;
    G.dec_vn = (int16_t *)b; b += SWD_GDG_MAXGUESS * 2;
    G.alt_depth = (int16_t *)b; b += SWD_GDG_MAXGUESS * 2;
    G.best_err = (uint8_t *)b; b += new_n;
    G.bp_hard = (uint8_t *)b; b += n;
    G.dec_val = (int8_t *)b; b += SWD_GDG_MAXGUESS;
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

// two lexicographic minima at once (same barriers as one).  LEAN: the caller guarantees a barrier between any earlier access to the
// staging words and this call, and another one after it before they are touched again -- only the middle barrier remains.
template <int NT, bool LEAN = false>
__device__ __forceinline__ void block_argmin2(double &k1, int &p1, double &k2, int &p2, Lds &s) {
    if constexpr (NT / 64 > 8) { // (the per-wave staging area holds 16 keys)
        block_argmin<NT>(k1, p1, s);
        block_argmin<NT>(k2, p2, s);
    } else {
        constexpr int NW = NT / 64;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            const double o1 = __shfl_xor(k1, d, 64), o2 = __shfl_xor(k2, d, 64);
            const int q1 = __shfl_xor(p1, d, 64), q2 = __shfl_xor(p2, d, 64);
            if (o1 < k1 || (o1 == k1 && q1 < p1)) { k1 = o1; p1 = q1; }
            if (o2 < k2 || (o2 == k2 && q2 < p2)) { k2 = o2; p2 = q2; }
        }
        double *wk = s.dbl + 4; // [16]: first keys, then second keys
        int *wp = s.iaux;       // [16]
        if constexpr (!LEAN) __syncthreads();
        if ((threadIdx.x & 63) == 0) {
            const int w = threadIdx.x >> 6;
            wk[w] = k1; wp[w] = p1; wk[NW + w] = k2; wp[NW + w] = p2;
        }
        __syncthreads();
        double b1 = wk[0], b2 = wk[NW];
        int c1 = wp[0], c2 = wp[NW];
#pragma unroll
        for (int w = 1; w < NW; ++w) {
            const double o1 = wk[w], o2 = wk[NW + w];
            const int q1 = wp[w], q2 = wp[NW + w];
            if (o1 < b1 || (o1 == b1 && q1 < c1)) { b1 = o1; c1 = q1; }
            if (o2 < b2 || (o2 == b2 && q2 < c2)) { b2 = o2; c2 = q2; }
        }
        k1 = b1; p1 = c1; k2 = b2; p2 = c2;
        if constexpr (!LEAN) __syncthreads();
    }
}

// Compact the still-active selected VNs (position order) into s.lv and (re)build the per-phase
// register caches.  Messages are untouched (they persist across decimation steps, bpgd.cpp:97-197).
// BP register caches of the guessing decoders: 16-bit slot numbers, two per register, classic parity words
template <int VF, int DM> using GdgVC = VnCacheP<VF, DM, 3, false>;
template <int KG> using GdgCC = CnCacheP<KG, 3>;

// Which check a thread serves during a tree walk: thread `sub` of the `grp` (1, 2 or 4 adjacent lanes) that share check `lc`
// takes the live positions number sub, sub + grp, ...  Chosen once per window (or per task) by gdg_cn_map -- checks dealt by
// decreasing live degree, heavy ones shared, exactly like the osd_window post phase (cn_assign) -- and kept while the tree is
// walked: degrees only go down from there, so the bound on a thread's walk stays valid, and the order stays roughly sorted.
struct GdgCnMap { int lc, sub, grp; };
template <int NT, int KG>
__device__ __forceinline__ GdgCnMap gdg_cn_map(const SwdGraphDev &g, Lds &s, const GdgLds &G) {
    GdgCnMap mp{s.ctid < g.m ? s.ctid : -1, 0, 1};
    // scratch for the counting sort: the position list of the path-metric sums (free until a block converges)
    int *dhist = (int *)(((uintptr_t)G.plist + 3) & ~(uintptr_t)3);
    uint16_t *cord = (uint16_t *)(dhist + 66);
    if ((char *)(cord + g.m) <= (char *)(G.plist + g.new_n)) // (uniform)
        cn_assign<NT, KG>(g, s, dhist, cord, true, false, -1, mp.lc, mp.sub, mp.grp);
    return mp;
}

// The check cache of a thread straight from the check's live-position mask: slot of position j = jptr[j] + lane.  Groups of four
// beyond the wave's longest walk are filled without looking at the mask.
template <int NT, int KG, int SH>
__device__ __forceinline__ void gdg_cn_cache_from_mask(const SwdGraphDev &g, Lds &s, const GdgCnMap &mp, CnCacheP<KG, SH> &cc) {
    const int dummy = swd_slot_far(g);
    const int lc = mp.lc;
    const bool act = (lc >= 0) && (lc < g.m) && (s.cn_val[lc >= 0 ? lc : 0] >= 0);
    const int l = act ? lc : 0;
    cc.l = act ? lc : -1;
    cc.sub = mp.sub;
    cc.grp = mp.grp;
    uint64_t mk = act ? s.livemask[l] : 0ull;
    const int live = act ? (int)s.cn_deg[l] : 0;
    const int cnt = (live > mp.sub) ? (live - mp.sub + mp.grp - 1) / mp.grp : 0;
    cc.cnt = cnt;
    cc.live = live;
    for (int k = 0; k < mp.sub; ++k) mk &= mk - 1; // this thread's first position: the sub-th live one (0 stays 0)
    const int wmax = __builtin_amdgcn_readfirstlane(wave_max(cnt));
#pragma unroll
    for (int gq = 0; gq < KG; ++gq) {
        if (gq * 4 < wmax) { // wave-uniform
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int sv = dummy;
                if (mk) {
                    const int j = __ffsll((long long)mk) - 1;
                    sv = (int)s.jptr[j] + l;
                    mk &= mk - 1;
                    if (mp.grp >= 2) mk &= mk - 1; // (the positions in between belong to the check's other threads)
                    if (mp.grp == 4) { mk &= mk - 1; mk &= mk - 1; }
                }
                cc.set_slot(gq * 4 + u, sv);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) cc.set_slot(gq * 4 + u, dummy);
        }
    }
}

// Static form of the shortened graph's node cache (depth-2 caches: new_n <= 2 NT): position vtid + i NT belongs to this thread
// for the whole tree walk, with EVERY edge of its column (gdg_static_load, one gather per window / task); after a decimation
// the working cache is re-derived from it and the state arrays in LDS -- a decided node or an edge to a deactivated check
// turns into dead positions -- without compaction, prefix sums or loads from the graph in HBM.  The node list then holds n
// at the places of decided nodes (bp_run<..., SPARSE>).
struct GdgNoStatic {};
template <int NT, int VF, int DM>
__device__ __forceinline__ void gdg_static_load(const SwdGraphDev &g, Lds &s, const GdgLds &G, GdgVC<VF, DM> &st) {
#pragma unroll
    for (int i = 0; i < VF; ++i) {
        const int idx = s.vtid + i * NT;
        if (idx < g.new_n) s.lv[idx] = G.pos_lv[idx]; // (read back by this thread only)
    }
    vn_cache_load<NT, VF, DM, false, true>(g, s, g.new_n, st);
}
template <int NT, int VF, int DM, int KG>
__device__ __forceinline__ int gdg_refresh_caches(const SwdGraphDev &g, Lds &s, const GdgLds &G, const GdgVC<VF, DM> &st, GdgVC<VF, DM> &vc, GdgCC<KG> &cn, const GdgCnMap &mp) {
    const int m = g.m, n = g.n, new_n = g.new_n;
    const uint32_t deadslot = (uint32_t)swd_slot_zero<NT>(g);
    SEL_T0();
#pragma unroll
    for (int i = 0; i < VF; ++i) {
        const int idx = s.vtid + i * NT;
        const bool inr = idx < new_n;
        const int v = inr ? (int)G.pos_lv[idx] : 0;
        const bool alive = inr && s.vn_val[v] == -1;
        if (inr) s.lv[idx] = (uint16_t)(alive ? v : n);
        vc.llr[i] = alive ? st.llr[i] : 0.0;
#pragma unroll
        for (int k2 = 0; k2 < DM / 2; ++k2) {
            const uint32_t pw = st.par[i][k2], ew = st.edp[i][k2];
            const int l0 = (int)(pw & 0xFFFFu), l1 = (int)(pw >> 16);
            const bool live0 = alive && l0 < m && s.cn_val[min(l0, m - 1)] >= 0;
            const bool live1 = alive && l1 < m && s.cn_val[min(l1, m - 1)] >= 0;
            vc.edp[i][k2] = (live0 ? (ew & 0xFFFFu) : deadslot) | ((live1 ? (ew >> 16) : deadslot) << 16);
            vc.par[i][k2] = (uint32_t)(live0 ? l0 : m) | ((uint32_t)(live1 ? l1 : m) << 16);
        }
    }
    SEL_T(6);
    gdg_cn_cache_from_mask<NT>(g, s, mp, cn);
    __syncthreads();
    SEL_T(7);
    return new_n;
}

template <int NT, int VF, int DM, int KG, class VC, class CC>
__device__ __forceinline__ int gdg_build_caches(const SwdGraphDev &g, Lds &s, const GdgLds &G, VC &vc, CC &cn, const GdgCnMap &mp) {
    const int tid = threadIdx.x, new_n = g.new_n;
    const int ch = (new_n + NT - 1) / NT;
    const int j0 = tid * ch, j1 = min(new_n, j0 + ch);
    int cnt = 0;
    SEL_T0();
    for (int j = j0; j < j1; ++j) cnt += (s.vn_val[G.pos_lv[j]] == -1) ? 1 : 0;
    int nlive;
    int pos = block_exscan<NT>(cnt, s, nlive);
    for (int j = j0; j < j1; ++j) {
        const int v = G.pos_lv[j];
        if (s.vn_val[v] == -1) s.lv[pos++] = (uint16_t)v;
    }
    __syncthreads();
    SEL_T(5);
    vn_cache_load<NT, VF, DM, false>(g, s, nlive, vc);
#ifdef SWD_SELPROF
    asm volatile("" : "+v"(vc.edp[0][0]), "+v"(vc.llr[0]));
#endif
    SEL_T(6);
    gdg_cn_cache_from_mask<NT>(g, s, mp, cn);
    __syncthreads();
    SEL_T(7);
    return nlive;
}

// the static cache type of a tree walk with register caches of depth VF, and the cache (re)build that goes with it
template <int VF, int DM> using GdgStatic = std::conditional_t<(VF <= 2), GdgVC<(VF <= 2 ? VF : 1), DM>, GdgNoStatic>;
template <int NT, int VF, int DM, int KG, class ST>
__device__ __forceinline__ int gdg_caches(const SwdGraphDev &g, Lds &s, const GdgLds &G, const ST &st, GdgVC<VF, DM> &vc, GdgCC<KG> &cn, const GdgCnMap &mp) {
    if constexpr (std::is_same_v<ST, GdgNoStatic>) return gdg_build_caches<NT, VF, DM, KG>(g, s, G, vc, cn, mp);
    else return gdg_refresh_caches<NT, VF, DM, KG>(g, s, G, st, vc, cn, mp);
}
template <int NT, int VF, int DM, class ST>
__device__ __forceinline__ void gdg_static_init(const SwdGraphDev &g, Lds &s, const GdgLds &G, ST &st) {
    if constexpr (!std::is_same_v<ST, GdgNoStatic>) gdg_static_load<NT, VF, DM>(g, s, G, st);
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
    if (threadIdx.x < 64) {
        const double pm = ordered_llr_sum_wave(g.llr, G.plist, total);
        if (threadIdx.x == 0) *dres = pm;
    }
    __syncthreads();
    return *dres;
}

// Snapshot record in HBM: vn state by position | cn_val | cn_deg | livemask
__device__ __forceinline__ int64_t gdg_snap_bytes(int m, int new_n) { return ((new_n + 2 * m + 7) & ~7) + 8 * (int64_t)m; }

// written and read with agent-scope word accesses: in the parallel form another workgroup (another XCD) reads it
template <int NT>
__device__ __forceinline__ void gdg_snap_save(const SwdGraphDev &g, Lds &s, const GdgLds &G, uint8_t *rec) {
    const int m = g.m, new_n = g.new_n;
    const int nb = (new_n + 2 * m + 7) & ~7;
    uint32_t *w = (uint32_t *)rec;
    for (int i = threadIdx.x; i < nb / 4; i += NT) {
        uint32_t x = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int q = 4 * i + k;
            uint32_t byte = 0;
            if (q < new_n) byte = (uint8_t)s.vn_val[G.pos_lv[q]];
            else if (q < new_n + m) byte = (uint8_t)s.cn_val[q - new_n];
            else if (q < new_n + 2 * m) byte = s.cn_deg[q - new_n - m];
            x |= byte << (8 * k);
        }
        ag_st(&w[i], x);
    }
    uint32_t *lm = w + nb / 4;
    for (int l = threadIdx.x; l < m; l += NT) {
        const uint64_t v = s.livemask[l];
        ag_st(&lm[2 * l], (uint32_t)v); ag_st(&lm[2 * l + 1], (uint32_t)(v >> 32));
    }
}

// set_masks (bpgd.cpp:241-248) minus the message re-initialisation, which the caller does after
// the branch decision and peeling (only messages of still-active VNs are ever read)
template <int NT>
__device__ __forceinline__ void gdg_snap_load(const SwdGraphDev &g, Lds &s, const GdgLds &G, const uint8_t *rec) {
    const int m = g.m, new_n = g.new_n;
    const int nb = (new_n + 2 * m + 7) & ~7;
    const uint32_t *w = (const uint32_t *)rec;
    __syncthreads();
    for (int i = threadIdx.x; i < nb / 4; i += NT) {
        const uint32_t x = ag_ld(&w[i]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int q = 4 * i + k;
            const int8_t val = (int8_t)(uint8_t)(x >> (8 * k));
            if (q < new_n) { const int v = G.pos_lv[q]; s.vn_val[v] = val; s.hard[v] = (val == 1) ? 1 : 0; }
            else if (q < new_n + m) s.cn_val[q - new_n] = val;
            else if (q < new_n + 2 * m) s.cn_deg[q - new_n - m] = (uint8_t)val;
        }
    }
    const uint32_t *lm = w + nb / 4;
    for (int l = threadIdx.x; l < m; l += NT) s.livemask[l] = (uint64_t)ag_ld(&lm[2 * l]) | ((uint64_t)ag_ld(&lm[2 * l + 1]) << 32);
    __syncthreads();
}

// bpgdg_decoder.select_vn (bp_guessing_decoder.pyx:340-442).  Returns -1 on failure, else 0.
// Classification of a position is independent of the scan order (num_flip only looks at active
// neighbour checks, and a live VN has no inactive ones), the decimations are applied by wave 0 in
// position order so that a contradiction stops exactly where the reference's scan stops.
// sp != nullptr (a side branch run as a task): min_converge_depth is a bound at least as permissive as the serial
// order's value at this point, the snapshot slot comes from the tree's shared counter instead of used_guess / max_guess,
// and what happened is reported in sp (the scheduler's replay decides what counts and queues the child in its turn).
struct GdgSpec {
    GdgCtx ctx;
    int fail_before, fail_after, child; // out
};

// The scan itself -- classification of every position, the aggressive decimations in position order, the peeling after
// them, and the candidate to guess -- shared by the Cython routine above and by BPGD::select_vn (bpgd.cpp:288-355, thresholds
// A / A_sum of the calling thread; C = 30, D = 3).  Returns -1 when a decimation or the peeling fails; else guess_pos (0x7fffffff:
// no candidate) and the favoured value.
template <int NT, class ST = GdgNoStatic>
__device__ __forceinline__ int gdg_select_core(const SwdGraphDev &g, const SwdDecodeParams &P, Lds &s, const GdgLds &G,
                                               const double *hist_b, double A, double A_sum, int depth, int &guess_pos, int &favor,
                                               const ST &st = ST{}, const double (*h4)[4] = nullptr) { // h4: the history values of this thread's positions (static cache)
    const int tid = threadIdx.x, n = g.n, new_n = g.new_n;
    const double C = 30.0, D = 3.0;
    double best_all = 10000.0, best_neg = 10000.0;
    int pos_all = 0x7fffffff, pos_neg = 0x7fffffff;
    SEL_T0();
    // classification of one active position from the four history values, its degree and its number of unsatisfied live checks
    auto classify = [&](int j, const double (&hl)[4], int deg, int num_flip) {
        uint8_t cat = 0;
        if (deg > 2) {
            bool smaller_A = true, all_neg = true, larger_C = true, larger_D = true;
            double hsum = 0.0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const double llr = hl[i];
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
        return cat;
    };
    if constexpr (!std::is_same_v<ST, GdgNoStatic>) {
        // the thread's own positions (static cache: every edge of the column, check numbers in the parity fields)
        constexpr int VFS = (int)(sizeof(st.llr) / sizeof(double)), DMS = (int)(sizeof(st.edp[0]) / sizeof(uint32_t)) * 2;
        int vv[VFS];
        bool act[VFS];
        double hl[VFS][4];
#pragma unroll
        for (int u = 0; u < VFS; ++u) {
            const int j = s.vtid + u * NT;
            vv[u] = G.pos_lv[min(j, new_n - 1)];
            act[u] = j < new_n && s.vn_val[vv[u]] == -1;
#pragma unroll
            for (int i = 0; i < 4; ++i) hl[u][i] = h4[u][i];
        }
#pragma unroll
        for (int u = 0; u < VFS; ++u) {
            const int j = s.vtid + u * NT;
            if (j >= new_n) continue;
            uint8_t cat = 0;
            if (act[u]) {
                int deg = 0, num_flip = 0;
#pragma unroll
                for (int k = 0; k < DMS; ++k) {
                    const int l = (int)((st.par[u][k >> 1] >> (16 * (k & 1))) & 0xFFFFu);
                    if (l < g.m) {
                        ++deg;
                        if (s.cn_val[l] >= 0 && s.par[l] != 0u) ++num_flip;
                    }
                }
                cat = classify(j, hl[u], deg, num_flip);
            }
            G.cat[j] = cat;
        }
    } else {
    // two positions per round, every global load of both (edge table rows -- padded beyond a column's degree --, the four
    // history slots) issued before anything depends on them
    const int Dg = g.D;
    for (int jb = tid; jb < new_n; jb += 2 * NT) {
        int vv[2];
        bool act[2];
        uint32_t ev[2][SWD_DMAX];
        double hl[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = jb + u * NT;
            const int v = G.pos_lv[min(j, new_n - 1)];
            vv[u] = v;
            act[u] = j < new_n && s.vn_val[v] == -1;
#pragma unroll
            for (int k = 0; k < SWD_DMAX; ++k) ev[u][k] = (k < Dg) ? g.vn_edge[k * n + v] : SWD_PAD_EDGE; // (k < Dg: uniform)
#pragma unroll
            for (int i = 0; i < 4; ++i) hl[u][i] = hist_b[i * n + v];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = jb + u * NT;
            if (j >= new_n) continue;
            uint8_t cat = 0;
            if (act[u]) {
                int deg = 0, num_flip = 0;
#pragma unroll
                for (int k = 0; k < SWD_DMAX; ++k) {
                    const uint32_t e = ev[u][k];
                    if (e != SWD_PAD_EDGE) {
                        ++deg;
                        const int l = swd_edge_lane(e);
                        if (s.cn_val[l] >= 0 && s.par[l] != 0u) ++num_flip;
                    }
                }
                cat = classify(j, hl[u], deg, num_flip);
            }
            G.cat[j] = cat;
        }
    }
    }
    // thresholds of the reference are strict "<" against the running minimum initialised to 10000:
    // a candidate with sum >= 10000 never becomes the minimum
    if (!(best_all < 10000.0)) { best_all = 10000.0; pos_all = 0x7fffffff; }
    if (!(best_neg < 10000.0)) { best_neg = 10000.0; pos_neg = 0x7fffffff; }
    SEL_T(0);
    // (the staging words were last touched before the block that just ran -- barriers in between -- and the barrier after wave 0's
    // section below comes before anything else touches them; the remaining barrier also publishes G.cat)
    block_argmin2<NT, true>(best_all, pos_all, best_neg, pos_neg, s);
    SEL_T(1);
    // the aggressive decimations in position order, then the peeling: wave 0 walks the classification itself (lane q owns the
    // positions [q CH, (q + 1) CH); the next one to apply is the smallest pending position of the first lane that has any)
    if (tid < 64) {
        const int CH = (new_n + 63) >> 6, jend = min(new_n, (tid + 1) * CH);
        auto pending = [&](int from) { // smallest position >= from of this lane's chunk that is to be decimated
            for (int j = from; j < jend; ++j) { const uint8_t c = G.cat[j]; if (c == 1 || c == 2) return j; }
            return 0x7fffffff;
        };
        int next = pending(tid * CH);
        bool bad = false;
        for (;;) {
            const unsigned long long have = __ballot(next != 0x7fffffff);
            if (have == 0ull) break;
            const int src = __ffsll((long long)have) - 1;
            const int j = __builtin_amdgcn_readlane(next, src);
            bad = gdg_set_value_wave(g, s, G.pos_lv[j], G.cat[j] == 2 ? 1 : 0);
            if (bad) break;
            if (tid == src) next = pending(j + 1);
        }
        if (!bad) bad = peel_wave<false, NT>(g, s);
        if (tid == 0) s.scal[1] = bad ? 1 : 0;
    }
    __syncthreads();
    SEL_T(2);
    if (s.scal[1]) return -1;
    if (pos_neg != 0x7fffffff) { guess_pos = pos_neg; favor = 1; }
    else { guess_pos = pos_all; favor = (best_all > 0) ? 0 : 1; }
    return 0;
}

template <int NT, class ST = GdgNoStatic>
__device__ __forceinline__ int gdg_select_vn(const SwdGraphDev &g, const SwdDecodeParams &P, Lds &s, const GdgLds &G,
                                             const double *hist_b, bool side, int depth, int min_converge_depth,
                                             int &used_guess, uint8_t *snap_b, GdgSpec *sp = nullptr, const ST &st = ST{}, const double (*h4)[4] = nullptr) {
    const int tid = threadIdx.x, new_n = g.new_n;
    const double A = side ? 0.0 : -3.0;
    double A_sum = side ? -10.0 : -12.0;
    if (depth == 0) A_sum = -16.0;
    int guess_pos, favor;
    if (gdg_select_core<NT>(g, P, s, G, hist_b, A, A_sum, depth, guess_pos, favor, st, h4) == -1) { if (sp) sp->fail_before = 1; return -1; }
    bool guess = true;
    SEL_T0();
    if (depth > min_converge_depth) guess = false;
    if (!side && depth >= P.max_side_depth) guess = false;
    if (side && depth > P.max_tree_depth) guess = false;
    if (guess && sp) {
        __syncthreads();
        if (tid == 0) s.iaux[4] = (int)atomicAdd(&sp->ctx.hdr[2], 1u);
        __syncthreads();
        const int slot = s.iaux[4];
        if (slot >= SWD_GDG_SLOTS) {
            if (tid == 0) ag_st(&sp->ctx.hdr[3], 1u); // more snapshots than the context holds: the unit is redone serially
        } else {
            if (tid == 0) {
                uint32_t *r = gdg_rec(sp->ctx, slot);
                ag_st(&r[0], (uint32_t)(depth + 1) | ((uint32_t)(uint8_t)(1 - favor) << 8) |
                                 ((uint32_t)(uint16_t)((guess_pos == 0x7fffffff ? -1 : guess_pos) + 1) << 16));
                ag_st(&r[1], 0u);
            }
            gdg_snap_save<NT>(g, s, G, sp->ctx.snap + (int64_t)slot * gdg_snap_bytes(g.m, new_n));
            sp->child = slot;
        }
    } else if (guess && used_guess < P.max_guess) {
        if (tid == 0) {
            G.dec_val[used_guess] = (int8_t)(1 - favor);
            G.dec_vn[used_guess] = (int16_t)(guess_pos == 0x7fffffff ? -1 : guess_pos);
            G.alt_depth[used_guess] = (int16_t)(depth + 1);
        }
        gdg_snap_save<NT>(g, s, G, snap_b + (int64_t)used_guess * gdg_snap_bytes(g.m, new_n));
        used_guess += 1;
    }
    SEL_T(3);
    if (sp) sp->fail_after = 1; // cleared below when the favoured value goes through
    if (guess_pos == 0x7fffffff) return -1; // no candidate left (the reference would index vn_mask[-1])
    __syncthreads();
    if (tid < 64) {
        bool bad = gdg_set_value_wave(g, s, G.pos_lv[guess_pos], favor);
        if (!bad) bad = peel_wave<false, NT>(g, s);
        if (tid == 0) s.scal[1] = bad ? 1 : 0;
    }
    __syncthreads();
    SEL_T(4);
    if (sp && !s.scal[1]) sp->fail_after = 0;
    return s.scal[1] ? -1 : 0;
}

// BPGD::decimate_vn_reliable (bpgd.cpp:258-286): largest |posterior of slot 3| among active VNs
template <int NT, class ST = GdgNoStatic>
__device__ __forceinline__ int gdg_decimate_reliable(const SwdGraphDev &g, Lds &s, const GdgLds &G, const double *hist_b,
                                                     const ST &st = ST{}, const double (*h4)[4] = nullptr) {
    const int tid = threadIdx.x, n = g.n, new_n = g.new_n;
    double best = 0.0; // stored negated so that block_argmin (a minimum) finds the largest magnitude
    int bpos = 0x7fffffff;
    [[maybe_unused]] double own3 = 0.0; // static cache: slot 3 of the candidate this thread offers
    if constexpr (!std::is_same_v<ST, GdgNoStatic>) {
        constexpr int VFS = (int)(sizeof(st.llr) / sizeof(double));
#pragma unroll
        for (int u = 0; u < VFS; ++u) {
            const int j = s.vtid + u * NT;
            if (j >= new_n || s.vn_val[G.pos_lv[j]] != -1) continue;
            const double a = fabs(h4[u][3]);
            if (a > -best) { best = -a; bpos = j; own3 = h4[u][3]; }
        }
    } else {
        for (int j = tid; j < new_n; j += NT) {
            const int v = G.pos_lv[j];
            if (s.vn_val[v] != -1) continue;
            const double a = fabs(hist_b[3 * n + v]);
            if (a > -best) { best = -a; bpos = j; } // strict ">" (bpgd.cpp:270), earliest position wins ties
        }
    }
    if (!(best < 0.0)) { best = 0.0; bpos = 0x7fffffff; }
    const int mypos = bpos;
    block_argmin<NT>(best, bpos, s);
    if (bpos == 0x7fffffff) return -1;
    const int v = G.pos_lv[bpos];
    int val;
    if constexpr (!std::is_same_v<ST, GdgNoStatic>) {
        if (mypos == bpos) s.scal[5] = (own3 > 0) ? 0 : 1; // (one thread owns the position)
        __syncthreads();
        val = s.scal[5];
    } else {
        val = (hist_b[3 * n + v] > 0) ? 0 : 1;
        __syncthreads();
    }
    if (tid < 64) {
        bool bad = gdg_set_value_wave(g, s, v, val);
        if (!bad) bad = peel_wave<false, NT>(g, s);
        if (tid == 0) s.scal[1] = bad ? 1 : 0;
    }
    __syncthreads();
    return s.scal[1] ? -1 : 0;
}

// error vector of the current hypothesis by position, packed four per word, to the owner's context
template <int NT>
__device__ __forceinline__ void gdg_store_err(Lds &s, const GdgLds &G, int new_n, uint32_t *dst) {
    for (int i = threadIdx.x; i < (new_n + 3) / 4; i += NT) {
        uint32_t x = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int q = 4 * i + k; if (q < new_n) x |= (uint32_t)(s.hard[G.pos_lv[q]] ? 1 : 0) << (8 * k); }
        ag_st(&dst[i], x);
    }
}

// The serial bookkeeping of bpgdg_decoder.gdg's second phase (bp_guessing_decoder.pyx:301-335), run incrementally over
// the branch records of a parked tree by whoever holds the tree's lock: advance the frontier over the records that
// are complete, in serial stack order (which snapshots the serial order would have run / pushed / pruned, which
// hypothesis wins -- strict "<" on the path metric), queue the next branches in that order (at most inflight_max of a
// tree at a time: speculation beyond the frontier is bounded), publish the current min_converge_depth as the
// pruning bound for running branches, and queue the tree's FINAL item when every record is accounted for.
__device__ __forceinline__ bool gdg_sched_step(const SwdGdgPar &gp, const SwdDecodeParams &P, const GdgCtx &c, int ctxid) {
    uint32_t *h = c.hdr;
    int i = (int)ag_ld(&h[8]), nseq = (int)ag_ld(&h[9]), used = (int)ag_ld(&h[10]), mcd = (int)ag_ld(&h[11]);
    int converge = (int)ag_ld(&h[12]), blocks = (int)ag_ld(&h[13]), post_it = (int)ag_ld(&h[14]), best = (int)ag_ld(&h[15]);
    double min_pm = __longlong_as_double((long long)((unsigned long long)ag_ld(&h[16]) | ((unsigned long long)ag_ld(&h[17]) << 32)));
    int launched = (int)ag_ld(&h[18]);
    unsigned long long lmask = (unsigned long long)ag_ld(&h[20]) | ((unsigned long long)ag_ld(&h[21]) << 32);
    const int finalp = (int)ag_ld(&h[22]), mcd0 = (int)ag_ld(&h[23]);
#ifdef SWD_GDG_CHECKS
    uint32_t *chk_status = gp.chk_status;
#endif
    GDG_CHECK(!finalp, 0);                                  // scheduler run after the FINAL item was queued
    GDG_CHECK(nseq >= 0 && nseq <= SWD_GDG_SLOTS && i >= 0 && i <= nseq, 1);
    GDG_CHECK(launched >= (int)ag_ld(&h[19]), 2);
    GDG_CHECK(ag_ld(&h[1]) == 1u, 3);                       // context state: 1 = owned by a parked tree
    if (finalp) return false;
    const bool ens = gp.ensemble != 0;
    uint32_t seqw[SWD_GDG_SLOTS / 4];
#pragma unroll
    for (int k = 0; k < SWD_GDG_SLOTS / 4; ++k) seqw[k] = ag_ld(&h[26 + k]);
    auto seq_get = [&](int k) { return (int)((seqw[k >> 2] >> (8 * (k & 3))) & 0xFFu); };
    auto seq_set = [&](int k, int v) { seqw[k >> 2] = (seqw[k >> 2] & ~(0xFFu << (8 * (k & 3)))) | ((uint32_t)v << (8 * (k & 3))); };
    const int maxj = min(P.max_side_branch_step, SWD_GDG_MAXSTEP);
    while (i < nseq) {
        const uint32_t *r = gdg_rec(c, seq_get(i));
        const int alt = (int)(ag_ld(&r[0]) & 0xFFu);
        if (alt > (ens ? mcd0 : mcd)) { ++i; continue; } // pruned by the serial order (a running copy is ignored)
        const uint32_t w1 = ag_ld(&r[1]);
        if (!(w1 & 2u)) break; // not finished yet: the frontier stops here
        if (!(w1 & 5u)) {      // the decision and the peeling after it went through
            const int nsteps = (int)((w1 >> 8) & 0xFFu), conv = (int)((w1 >> 16) & 0xFFu) - 1;
            for (int j = 0; j < maxj && j < nsteps; ++j) {
                const uint32_t sw = ag_ld(&r[4 + j]);
                const int depth = alt + j;
                ++blocks; post_it += (int)(sw & 0xFFu);
                if (conv == j) {
                    converge = 1;
                    const double pm = __longlong_as_double((long long)((unsigned long long)ag_ld(&r[2]) | ((unsigned long long)ag_ld(&r[3]) << 32)));
                    if (pm < min_pm) {
                        if (depth < mcd) mcd = depth;
                        best = seq_get(i);
                        min_pm = pm;
                    }
                    break;
                }
                if (depth > (ens ? mcd0 : mcd) + 2) break;
                if (sw & 0x100u) break; // select_vn failed before its snapshot
                const bool guess = !(depth > (ens ? mcd0 : mcd)) && !(depth > P.max_tree_depth);
                if (guess && (ens || used < P.max_guess)) {
                    const int child = (int)((sw >> 16) & 0xFFu) - 1;
                    if (child >= 0 && nseq < SWD_GDG_SLOTS) { seq_set(nseq, child); ++nseq; }
                    ++used;
                }
                if (sw & 0x200u) break; // ... or after it
            }
        }
        ++i;
    }
    const int fin = (int)ag_ld(&h[19]);
    // (workgroups waiting for an item: every branch that may still count is queued -- speculation costs nothing then)
    // (diagnostics, SWD_GDG_ADAPTIVE=0: inflight_max negated = fixed bound)
    const int lim = gp.inflight_max < 0 ? -gp.inflight_max : (((int32_t)(ag_ld(&gp.q[1]) - ag_ld(&gp.q[0])) < 0) ? SWD_GDG_SLOTS : gp.inflight_max);
    for (int k = i; k < nseq && launched - fin < lim; ++k) {
        const int b = seq_get(k);
        if ((lmask >> b) & 1ull) continue;
        if ((int)(ag_ld(&gdg_rec(c, b)[0]) & 0xFFu) > (ens ? mcd0 : mcd)) continue;
        GDG_CHECK(b >= 0 && b < SWD_GDG_SLOTS, 4);
        lmask |= 1ull << b; ++launched;
        ring_push(gp.q, gp.qmask, item_side(ctxid, b));
    }
    const bool complete = (i == nseq) && (launched == fin);
    ag_st(&h[8], (uint32_t)i); ag_st(&h[9], (uint32_t)nseq); ag_st(&h[10], (uint32_t)used); ag_st(&h[11], (uint32_t)mcd);
    ag_st(&h[12], (uint32_t)converge); ag_st(&h[13], (uint32_t)blocks); ag_st(&h[14], (uint32_t)post_it); ag_st(&h[15], (uint32_t)best);
    const unsigned long long pb = (unsigned long long)__double_as_longlong(min_pm);
    ag_st(&h[16], (uint32_t)pb); ag_st(&h[17], (uint32_t)(pb >> 32));
    ag_st(&h[18], (uint32_t)launched); ag_st(&h[20], (uint32_t)lmask); ag_st(&h[21], (uint32_t)(lmask >> 32));
#pragma unroll
    for (int k = 0; k < SWD_GDG_SLOTS / 4; ++k) ag_st(&h[26 + k], seqw[k]);
    ag_st(&h[7], (uint32_t)(ens ? mcd0 : mcd));
    if (complete) ag_st(&h[22], 1u);
    return complete;
}

// Run the scheduler of a tree (one thread).  The lock is a short spin lock: every holder is a running workgroup that
// leaves after one step.  A finishing side branch counts itself as finished INSIDE the critical section, so the step
// that sees "every launched branch finished" is the last access any branch makes to the context; the FINAL item is
// queued after the lock is open, and nothing touches the context after that but the FINAL handler (which frees it).
__device__ __forceinline__ void gdg_sched(const SwdGdgPar &gp, const SwdDecodeParams &P, const GdgCtx &c, int ctxid, bool finished_one) {
    while (atomicCAS(&c.hdr[0], 0u, 1u) != 0u) __builtin_amdgcn_s_sleep(1);
    if (finished_one) ag_st(&c.hdr[19], ag_ld(&c.hdr[19]) + 1u);
    const bool complete = gdg_sched_step(gp, P, c, ctxid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the state is written back before the lock opens
    ag_st(&c.hdr[0], 0u);
    if (complete) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ring_push(gp.q, gp.qmask, item_final(ctxid));
    }
}

// One side branch (a saved snapshot) of a parked tree, run by whichever workgroup popped the item.  Uses the
// workgroup's LDS from scratch and its own history scratch; reads the tree's context / snapshot, writes the branch
// record back and runs the tree's scheduler.
// VFP: depth of the register cache for the shortened graph (2 when every window keeps new_n <= 2 NT columns -- the plan decides,
// Plan::post_depth2 -- else VF): the BP blocks of the post-processing only ever see the live nodes among the first new_n.
template <int NT, int VF, int DM, int KG, int VFP = VF>
__device__ __forceinline__ void gdg_run_task(const SwdPipeArgs &a, char *smem, uint32_t payload, int ctid, int vtid) {
    const int tid = threadIdx.x;
    const SwdGdgPar &gp = a.gdgp;
    const int ctxid = (int)((payload >> 8) & 0x3FFFFFu), slot = (int)(payload & 0xFFu);
#ifdef SWD_GDG_DEBUG
    uint32_t *dbg_status = gp.chk_status;
    const long long dbg_t0 = wall_clock64();
#endif
    GdgSpec sp;
    sp.ctx = gdg_ctx(gp, ctxid);
#ifdef SWD_GDG_CHECKS
    uint32_t *chk_status = gp.chk_status;
    GDG_CHECK(ctxid >= 0 && ctxid < gp.nctx && slot < SWD_GDG_SLOTS, 5);
    GDG_CHECK(ag_ld(&sp.ctx.hdr[1]) == 1u, 6);             // the context is owned by a parked tree
    GDG_CHECK((int)ag_ld(&sp.ctx.hdr[4]) < a.W && (int)ag_ld(&sp.ctx.hdr[5]) < a.B, 7);
    GDG_CHECK((ag_ld(&gdg_rec(sp.ctx, slot)[1]) & 2u) == 0u, 8); // this branch has not run before
    if (!(ctxid >= 0 && ctxid < gp.nctx && slot < SWD_GDG_SLOTS && (int)ag_ld(&sp.ctx.hdr[4]) < a.W)) return;
#endif
    const int wi = (int)ag_ld(&sp.ctx.hdr[4]);
    const bool dead_unsat = ag_ld(&sp.ctx.hdr[6]) != 0u;
    const SwdWindowDev &w = a.wins[wi];
    const SwdGraphDev &g = w.g; // (a task of three or four steps does not repay staging row_col / perm in LDS: measured)
    const SwdLdsLayout &L = w.L;
    const SwdDecodeParams &P = a.P;
    const int m = g.m, n = g.n, new_n = g.new_n;
    uint32_t *rec = gdg_rec(sp.ctx, slot);
    const uint32_t w0 = ag_ld(&rec[0]);
    const int alt = (int)(w0 & 0xFFu), dval = (int)(int8_t)(uint8_t)(w0 >> 8), gpos = (int)(w0 >> 16) - 1;
    int nsteps = 0, conv_step = -1;
    double pm = 0.0;
    bool start_failed = false;
    // The pruning bound changes while branches run (the scheduler lowers it): ONE thread reads it and the workgroup
    // takes that value, or its threads would disagree about leaving the loop.  bcast: two words below the syndrome bytes.
    const int bound_word = gp.static_bound ? 23 : 7; // diagnostics: prune against the main branch's value only
    uint32_t *bcast = (uint32_t *)(smem + a.off_det) - 4 + 1;
    __syncthreads();
    if (tid == 0) *bcast = ag_ld(&sp.ctx.hdr[bound_word]);
    __syncthreads();
    const bool pruned = alt > (int)*bcast; // the serial order has already passed this snapshot by
    if (!pruned) {
        Lds s;
        lds_bind(s, smem, L, smem);
        s.fpar = 0; s.ctid = ctid; s.vtid = vtid;
        GdgLds G;
        gdg_bind(G, s.scratch, L, n, new_n);
        double *hist_b = a.hist + (int64_t)blockIdx.x * a.hist_stride;
        __syncthreads();
        for (int l = tid; l < m; l += NT) s.cn_deg0[l] = g.row_deg[l];
        for (int v = tid; v < n; v += NT) { s.vn_val[v] = 0; s.hard[v] = 0; }
        for (int j = tid; j <= g.K; j += NT) s.jptr[j] = g.jptr[j];
        for (int i = tid; i < (new_n + 1) / 2; i += NT) {
            const uint32_t x = ag_ld(&sp.ctx.pos[i]);
            G.pos_lv[2 * i] = (uint16_t)x;
            if (2 * i + 1 < new_n) G.pos_lv[2 * i + 1] = (uint16_t)(x >> 16);
        }
        if (P.max_iter_per_step < 4)
            for (int i = P.max_iter_per_step * n + tid; i < 4 * n; i += NT) hist_b[i] = 0.0;
        gdg_snap_load<NT>(g, s, G, sp.ctx.snap + (int64_t)slot * gdg_snap_bytes(m, new_n));
        if (tid < 64) {
            bool bad = (gpos < 0) ? true : gdg_set_value_wave(g, s, G.pos_lv[gpos], dval);
            if (!bad) bad = peel_wave<false, NT>(g, s);
            if (tid == 0) s.scal[1] = bad ? 1 : 0;
        }
        __syncthreads();
        start_failed = s.scal[1] != 0;
        if (!start_failed) {
            GdgVC<VFP, DM> vc;
            GdgCC<KG> cn;
            using ST = GdgStatic<VFP, DM>;
            ST vst;
            gdg_static_init<NT, VFP, DM>(g, s, G, vst);
            double h4[VFP][4] = {}; // history of the last four iterations of a block, for this thread's positions (static cache only)
            const GdgCnMap cmap = gdg_cn_map<NT, KG>(g, s, G);
            const int maxj = min(P.max_side_branch_step, SWD_GDG_MAXSTEP);
            for (int j = 0; j < maxj; ++j) {
                const int depth = alt + j;
                const int nlive = gdg_caches<NT, VFP, DM, KG>(g, s, G, vst, vc, cn, cmap);
                if (j == 0) { bp_init<VFP, DM>(s, vc); __syncthreads(); }
                int it;
                const int cv = bp_run<NT, VFP, DM, KG, false, false, false, !std::is_same_v<ST, GdgNoStatic>>(g, P, s, P.max_iter_per_step, nlive, vc, cn, hist_b, it, P.gdg_factor, dead_unsat, nullptr, h4);
                uint32_t sw = (uint32_t)it & 0xFFu;
                nsteps = j + 1;
                if (cv) {
                    pm = gdg_get_pm<NT>(g, s, G);
                    gdg_store_err<NT>(s, G, new_n, sp.ctx.err + (int64_t)slot * sp.ctx.err_words);
                    conv_step = j;
                    if (tid == 0) ag_st(&rec[4 + j], sw);
                    break;
                }
                // the bound is the serial order's min_converge_depth as far as the frontier has got: at least as
                // permissive as the value the serial order has when it reaches this branch
                __syncthreads();
                if (tid == 0) *bcast = ag_ld(&sp.ctx.hdr[bound_word]);
                __syncthreads();
                const int bound = (int)*bcast;
                if (depth > bound + 2) { if (tid == 0) ag_st(&rec[4 + j], sw); break; }
                sp.fail_before = sp.fail_after = 0; sp.child = -1;
                int dummy = 0;
                const int rc = gdg_select_vn<NT>(g, P, s, G, hist_b, true, depth, bound, dummy, nullptr, &sp, vst, h4);
                sw |= (sp.fail_before ? 0x100u : 0u) | (sp.fail_after ? 0x200u : 0u) | ((uint32_t)(sp.child + 1) << 16);
                if (tid == 0) ag_st(&rec[4 + j], sw);
                if (rc == -1) break;
            }
        }
    }
    if (tid == 0) {
        const unsigned long long pb = (unsigned long long)__double_as_longlong(pm);
        ag_st(&rec[2], (uint32_t)pb); ag_st(&rec[3], (uint32_t)(pb >> 32));
    }
    ag_publish_barrier(); // the record's body, the error vector and the child snapshots are out ...
    if (tid == 0) {
        ag_st(&rec[1], (start_failed ? 1u : 0u) | 2u | (pruned ? 4u : 0u) | ((uint32_t)nsteps << 8) | ((uint32_t)(conv_step + 1) << 16));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // ... before the record reads "done"
#ifdef SWD_GDG_DEBUG
        const long long dbg_t1 = wall_clock64();
#endif
        gdg_sched(gp, P, sp.ctx, ctxid, true);
#ifdef SWD_GDG_DEBUG
        GDG_COUNT(1, 1); GDG_COUNT(2, dbg_t1 - dbg_t0); GDG_COUNT(3, wall_clock64() - dbg_t1); GDG_COUNT(4, nsteps); GDG_COUNT(5, pruned ? 1 : 0);
#endif
    }
    __syncthreads();
}

// FINAL item: the result of a parked tree.  Fills R and s.hard like decode_window_gdg would have; false if the tree's
// snapshot area overflowed (the caller decodes the unit again in the serial form).
template <int NT>
__device__ __forceinline__ bool gdg_finalize(const SwdGraphDev &g, const SwdDecodeParams &P, Lds &s, const GdgCtx &c, WinResult &R) {
    const int tid = threadIdx.x, n = g.n, new_n = g.new_n;
    if (ag_ld(&c.hdr[3]) != 0u) return false;
    R = WinResult{};
    R.exit_class = SWD_EXIT_POST;
    R.conv = (int)ag_ld(&c.hdr[12]);
    R.pm = __longlong_as_double((long long)((unsigned long long)ag_ld(&c.hdr[16]) | ((unsigned long long)ag_ld(&c.hdr[17]) << 32)));
    R.pre_it = (int)ag_ld(&c.hdr[24]); R.post_it = (int)ag_ld(&c.hdr[14]); R.total_it = R.pre_it + R.post_it;
    R.live_vn = (int)ag_ld(&c.hdr[10]); R.live_cn = (int)ag_ld(&c.hdr[13]); R.live_e = (int)ag_ld(&c.hdr[11]);
    R.osd_rowadds = (int)ag_ld(&c.hdr[25]); // statistics word 7: snapshots of the main branch that became tasks (0 in the serial form)
    const int best = (int)ag_ld(&c.hdr[15]);
    for (int v = tid; v < n; v += NT) s.hard[v] = 0;
    __syncthreads();
    const uint32_t *ev = c.err + (int64_t)best * c.err_words;
    for (int i = tid; i < (new_n + 3) / 4; i += NT) {
        const uint32_t x = ag_ld(&ev[i]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int q = 4 * i + k;
            if (q < new_n && ((x >> (8 * k)) & 1u)) {
                const uint32_t pw = ag_ld(&c.pos[q >> 1]);
                s.hard[(q & 1) ? (pw >> 16) : (pw & 0xFFFFu)] = 1;
            }
        }
    }
    __syncthreads();
    return true;
}

// The reference's threaded ensemble (bpgdg_decoder(multi_thread=True): bp_guessing_decoder.pyx:238-251 over
// BPGD_main_thread / BPGD_tree_thread / BPGD_side_thread::do_work, bpgd.cpp:435-688) on one workgroup, the thread bodies in
// the order main, tree threads by id, side threads by index -- the order of the test oracle's restatement (gdg_multi_run), which is
// pinned to the reference's real threads.  Every body is a pure function of its inputs; only the strict-< update of the
// shared best depends on thread timing in the reference, and only when two converged hypotheses carry the same path metric
// with different vectors (statistics word 7 counts those: such shots have no single reference answer).
//   main   thresholds (-3, -16 at depth 0 else -12); select_vn runs BEFORE the convergence test (:630-633); at depths D..S-1 the
//          masks before the guess go to side thread depth - D together with the unfavoured value
//   tree   id = 1 .. 2^D - 1: at depth d < D bit (D-1-d) of id picks favour / unfavour (unfavour: thresholds (0, -10) from then on,
//          NO re-initialisation of the messages); at depth D the masks are saved; after max_tree_branch_step + D + 1 steps the
//          saved masks are restored with freshly initialised messages and the unfavoured value, max_tree_branch_step more steps
//   side   thresholds (0, -10): the handed-over masks, messages = priors, the unfavoured value, max_side_branch_step steps
// Snapshot slots in snap_b: 0 = state after BPGD::reset, 1 = the tree thread's saved masks, 2 + j = side thread j's.
// Entered with the state after reset (peeled, caches built, messages initialised).  Leaves the winner in G.best_err.
// (Round 3 ran these bodies one after the other on the unit's workgroup -- gdg_ensemble_ref, removed; the oracle's gdg_multi_run is that form.)

// The same ensemble with the work its threads share done once: the walk of the PREFIX TREE.
// A tree thread's first D steps are fixed by the direction bits of its id, and two threads whose ids share the first d bits do
// exactly the same work through depth d - 1 -- same masks, same messages (an unfavoured guess does not re-initialise them), same
// thresholds (they depend on the depth and on whether a 1-bit has been passed), hence the same BP block, the same select_vn
// scan and the same guess at depth d.  The main thread walks the all-favoured path with the thresholds of a tree thread that has
// not left it.  So the 2^D hypotheses (main = leaf 0, tree thread p = leaf p) are the leaves of a binary tree of depth D whose
// inner node (d, prefix) is one BP block + one scan, shared by the 2^(D-d) threads below it: 2^D - 1 shared steps instead of
// D 2^D.  (Oracle statistics on the [[144,12,12]] (3,1) windows: 153.6 -> 59.2 BP blocks per ensemble at D = 5 / S = 6,
// 39.4 -> 24.1 at D = 3 / S = 10.)  Leaves are visited in increasing order (favoured child first): that is the order main, tree
// threads by id in which the thread-by-thread form -- and the oracle -- make their strict-< offers, so winner and tie count come
// out the same; an inner node that converges makes the offer for every thread below it (same metric, same vector: the first id
// wins, the others neither win nor count as ties), after the main thread's own on the all-favoured path (main scans BEFORE its
// convergence test, so its vector may carry more decimations than the tree threads': the post-block vector is stashed).  At a
// fork the state AFTER the scan -- masks and all messages -- is saved; the unfavoured child is walked from it when its first leaf
// comes up.  BP blocks and iterations are counted once per thread that would have run them.  From depth D on every thread is on
// its own and the code is that of the thread-by-thread form; so are the side threads.
// Snapshot area (per workgroup): slot 0 unused, 1 = the solo tree thread's saved masks, 2 + j = side thread j's, then D fork
// records (masks + message cells), the stash vector and the main thread's exit vector.
// A fork record's message part: with the static tree-walk cache (every position has its thread, with every edge of its column:
// new_n <= 2 NT) each thread keeps the cells of its own positions -- VF x DM cells per thread, coalesced, every access in flight
// at once; decided positions and padded edges copy the wave's zero slot, which is re-armed before anybody reads it.  Without the
// static cache: all message cells of the window, eight per thread in flight.  (A first version copied every cell with one
// dependent agent-scope access per loop turn: 25 us per fork, more than two BP blocks.)
__device__ __forceinline__ int gdg_ens_fork_cells(int nmsg, int nt, int vf, int dm) { return max(nmsg, vf * dm * nt); }
__device__ __forceinline__ int64_t gdg_ens_fork_bytes(int m, int new_n, int cells) { return ((gdg_snap_bytes(m, new_n) + 15) & ~(int64_t)15) + (((int64_t)cells * 8 + 15) & ~(int64_t)15); }
template <int NT, int VF, int DM, class ST>
__device__ __forceinline__ void gdg_msg_save(const Lds &s, const ST &st, int nmsg, uint8_t *dst) {
    unsigned long long *d = (unsigned long long *)dst;
    if constexpr (!std::is_same_v<ST, GdgNoStatic>) {
#pragma unroll
        for (int i = 0; i < VF; ++i) {
            uint32_t ad[DM];
            st.get_ed(i, ad);
#pragma unroll
            for (int k = 0; k < DM; ++k)
                __hip_atomic_store(&d[(i * DM + k) * NT + s.vtid], *(const unsigned long long *)((const char *)s.msg + ad[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (by position: another workgroup may deal its waves differently)
        }
    } else {
        const unsigned long long *src = (const unsigned long long *)s.msg;
        for (int i0 = threadIdx.x; i0 < nmsg; i0 += 8 * NT) {
            unsigned long long v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[min(i0 + u * NT, nmsg - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) if (i0 + u * NT < nmsg) __hip_atomic_store(&d[i0 + u * NT], v[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
template <int NT, int VF, int DM, class ST>
__device__ __forceinline__ void gdg_msg_load(Lds &s, const ST &st, int nmsg, const uint8_t *srcb) {
    const unsigned long long *src = (const unsigned long long *)srcb;
    if constexpr (!std::is_same_v<ST, GdgNoStatic>) {
        unsigned long long v[VF][DM];
#pragma unroll
        for (int i = 0; i < VF; ++i)
#pragma unroll
            for (int k = 0; k < DM; ++k) v[i][k] = __hip_atomic_load(&src[(i * DM + k) * NT + s.vtid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int i = 0; i < VF; ++i) {
            uint32_t ad[DM];
            st.get_ed(i, ad);
#pragma unroll
            for (int k = 0; k < DM; ++k) *(unsigned long long *)((char *)s.msg + ad[k]) = v[i][k];
        }
    } else {
        unsigned long long *d = (unsigned long long *)s.msg;
        for (int i0 = threadIdx.x; i0 < nmsg; i0 += 8 * NT) {
            unsigned long long v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = __hip_atomic_load(&src[min(i0 + u * NT, nmsg - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int u = 0; u < 8; ++u) if (i0 + u * NT < nmsg) d[i0 + u * NT] = v[u];
        }
    }
    __syncthreads();
}

// ---- the ensemble's leaves as work items (kernel kind 7 on the work-item ring) ------------------------------------------------
// A hard window's ensemble takes up to 16 ms on one workgroup (64 hypotheses, most of them walking all their steps), a shot's windows
// are sequential, and a launch ended with a 60 ms tail in which one to three workgroups walked the last monster shots.  The 2^D
// leaves are independent once the shared prefix has been walked, so the OWNER of a unit (ROLE 1) walks the prefix tree, the main
// thread's own steps and the side threads, and hands every tree thread to the queue: at a fork of depth D - 1 it stores the fork
// record in the unit's CONTEXT (HBM: header, position list, fork table, offer table, one vector per hypothesis, 2^(D-1) fork
// records) and queues one TASK per leaf below it (ROLE 2: restore, decide the leaf's value, walk the thread's own steps).  Nobody
// applies an offer: every role writes (metric, vector) into the offer table at the hypothesis' id, and whoever takes the last
// token of the context queues FINAL, whose handler replays the table in the oracle's order -- main, tree threads by id, side
// threads -- with the strict "<" and the tie count.  ROLE 0 is the whole ensemble on one workgroup (no context free, W = 1 calls).
struct EnsCtx {
    uint32_t *hdr, *pos, *node, *etab; // header words: 0 tokens  4 window  5 shot  6 dead_unsat  8 BP blocks  9 iterations  10 side threads run
    uint8_t *vec, *fork;               //               11 main's exit vector present  12 pre-processing iterations
    int vecb, nh, id;
};
__device__ __forceinline__ EnsCtx gdg_ens_ctx(const SwdGdgPar &gp, int id) {
    uint8_t *b = gp.ctx + (int64_t)id * gp.ctx_stride;
    EnsCtx c;
    c.hdr = (uint32_t *)b; c.pos = (uint32_t *)(b + gp.off_pos); c.node = (uint32_t *)(b + gp.off_node); c.etab = (uint32_t *)(b + gp.off_rec);
    c.vec = b + gp.off_err; c.vecb = gp.err_stride; c.nh = gp.ens_hyps; c.id = id;
    c.fork = gp.csnap + (int64_t)id * gp.csnap_stride;
    return c;
}

// Written as ONE loop whose body runs a single step -- [restore a saved state] [decide a node + peel] BP block [scan] [offers, saves]
// -- with the scalar bookkeeping around it deciding what the next step is: the BP block, the scan, the decimation and the
// snapshot routines are each inlined exactly once (as straight-line copies of the thread bodies the kernel took 25 minutes to
// compile and its code no longer fit the instruction cache).
// role 0: the whole ensemble here.  1: owner (tree threads become tasks).  2: task = tree thread `task_p` from its fork on.
// The role is a run-time value and the kernel inlines this walk at ONE place (pipeline_kernel, where a unit that needs its ensemble
// and a task of a parked ensemble meet).  Measured on the way: three inlined copies (one per role) or two -- every launch 1.3 to
// 1.5 times slower, the copies evict each other from the instruction cache that the workgroups of a CU share; a real, non-inlined
// function -- twice as slow, the BP caches and the walk's state live on the stack.  Everything a walk keeps in registers (static
// cache, check map, BP caches) is built in here.
template <int NT, int VF, int DM, int KG>
__device__ __forceinline__ void gdg_ensemble_tree(int ROLE, const SwdGraphDev *gptr, const SwdDecodeParams *Pptr, Lds s, GdgLds G,
                                                            double *hist_b, uint8_t *snap_b, WinResult *Rptr, int dead_unsat_i,
                                                            const SwdGdgPar *gp, const EnsCtx *ec, int task_p) {
    const SwdGraphDev &g = *gptr;
    const SwdDecodeParams &P = *Pptr;
    WinResult &R = *Rptr;
    const bool dead_unsat = dead_unsat_i != 0;
    using ST = GdgStatic<VF, DM>;
    ST st;
    gdg_static_init<NT, VF, DM>(g, s, G, st);
    const GdgCnMap cmap = gdg_cn_map<NT, KG>(g, s, G); // (the degrees of the state this walk starts from bound every later state of it)
    GdgVC<VF, DM> vc;
    GdgCC<KG> cn;
    const int tid = threadIdx.x, m = g.m, new_n = g.new_n;
    const int Dp = P.max_tree_depth, S = P.max_side_depth;
    const int T = (1 << Dp) - 1, NS = max(S - Dp, 0);
    const int64_t rec = gdg_snap_bytes(m, new_n), rec16 = (rec + 15) & ~(int64_t)15;
    const int nmsg = g.E + 1 + 2 * (NT / 64); // message cells incl. the sink and the waves' far / zero slots
    const int64_t forkb = gdg_ens_fork_bytes(m, new_n, gdg_ens_fork_cells(nmsg, NT, VF, DM)), vecb = ((int64_t)new_n + 15) & ~(int64_t)15;
    uint8_t *fork0 = snap_b + (((int64_t)(2 + NS) * rec + 15) & ~(int64_t)15);
    uint32_t *stash = (uint32_t *)(fork0 + (int64_t)Dp * forkb); // post-block vector of a converged node on the all-favoured path
    uint32_t *main_fb = (uint32_t *)((uint8_t *)stash + vecb);   // the main thread's error vector when it ends unconverged
    const int NONE = 0x7fffffff;
    enum { M_DESC, M_MAIN, M_TREE1, M_TREE2, M_SIDE };                                  // what the next step belongs to
    enum { C_NONE, C_LEAF, C_DESC, C_MAIN, C_TREE1, C_BK, C_TREE2, C_SIDE0, C_SIDE };   // whose decimation precedes it
    enum { ADV_NONE, ADV_LEAF, ADV_REPLAY, ADV_SIDE };                                  // where to go when a walk ends
    enum { R_NONE, R_FORK, R_BK, R_SIDE, R_CTX };
    double best = 10000.0;
    int winner = -1, ties = 0, blocks = 0, sides_run = 0, it = 0;
    bool main_alive = true, main_conv = false, have_main_fb = false;
    int *nd_gpos = s.iaux + 16, *nd_fav = s.iaux + 24; // the guess of the fork at depth d (d < 6)
    double h4[VF][4] = {};
    auto vec_byte = [&](const uint32_t *v, int j) { return (uint8_t)((ag_ld(&v[j >> 2]) >> (8 * (j & 3))) & 0xFFu); };
    // walk state
    int mode = Dp > 0 ? M_DESC : M_MAIN, p = 0, d = 0, iter = 0, jside = -1, dead_depth = NONE;
    bool first = true;                   // the next block starts from the priors
    int adv = ADV_NONE, restore = R_NONE, rslot = 0, set_ctx = C_NONE, set_pos = NONE, set_val = 0;
    bool prev_with_main = false;         // the pending decimation is also the main thread's
    bool want_main_end = false;          // the main thread has left its loop: keep its vector (before any state is restored)
    bool saved = false;                  // tree thread on its own: masks saved at depth D
    double own_pm = 10000.0;
    int bk_pos = NONE, bk_val = 0;
#ifdef SWD_GDGPROF
    long long gp_t_ = wall_clock64();
#define EPT(k) do { const long long gp_n_ = wall_clock64(); R.gp[k] += gp_n_ - gp_t_; gp_t_ = gp_n_; } while (0)
#else
#define EPT(k) do { } while (0)
#endif
    R.post_it = (ROLE == 0) ? R.post_it : 0;
    const int nforks = Dp >= 1 ? (1 << (Dp - 1)) : 0;
    if (ROLE == 2 && task_p <= T) { // tree thread task_p: from the fork at depth D - 1 above it, through the child its last direction bit names
        const int q = task_p >> 1, gp_ = (int)ag_ld(&ec->node[2 * q]), fv_ = (int)ag_ld(&ec->node[2 * q + 1]);
        p = task_p; d = Dp; mode = M_TREE1; first = false; main_alive = false;
        restore = R_CTX; rslot = q; set_ctx = C_LEAF; set_pos = gp_; set_val = (task_p & 1) ? 1 - fv_ : fv_;
    } else if (ROLE == 2) { // side thread task_p - T - 1: the masks the main thread handed over, messages = priors, the unfavoured value
        const uint32_t *sd = ec->node + 2 * nforks + 3 * (task_p - T - 1);
        jside = task_p - T - 1; mode = M_SIDE; d = (int)ag_ld(&sd[2]); iter = 0; first = true; main_alive = false; sides_run = 1;
        restore = R_CTX; rslot = -1 - jside; set_ctx = C_SIDE0; set_pos = (int)ag_ld(&sd[0]); set_val = (int)ag_ld(&sd[1]);
    }
    // an offer: applied at once when the whole ensemble runs here, else written into the context's table at the hypothesis' id
    auto emit = [&](int who, int count, double pm) {
        gdg_store_err<NT>(s, G, new_n, (uint32_t *)(ec->vec + (int64_t)who * ec->vecb));
        if (tid == 0) {
            uint32_t *e = ec->etab + 4 * who;
            const unsigned long long pb = (unsigned long long)__double_as_longlong(pm);
            ag_st(&e[2], (uint32_t)pb); ag_st(&e[3], (uint32_t)(pb >> 32)); ag_st(&e[0], 1u | ((uint32_t)count << 8));
        }
    };
    for (;;) {
        EPT(4);
        if (ROLE == 2 && (adv == ADV_LEAF || adv == ADV_SIDE)) break; // the thread's walk is over
        if (want_main_end) { // the vector returned if nothing converges (bpgd.cpp:677-682)
            if (!main_conv) { gdg_store_err<NT>(s, G, new_n, ROLE == 1 ? (uint32_t *)(ec->vec + (int64_t)ec->nh * ec->vecb) : main_fb); ag_publish_barrier(); have_main_fb = true; }
            main_alive = false; want_main_end = false;
        }
        if (adv == ADV_LEAF) { // the next leaf whose fork is alive: its path leaves that fork through the unfavoured child
            int dstar = 0;
            for (++p; p <= T; ++p) {
                dstar = Dp - __ffs(p);
                if (ROLE == 1 && dstar == Dp - 1) continue; // (owner: the leaves below a fork of depth D - 1 are tasks)
                if (dead_depth > dstar) break;
            }
            if (p > T) adv = ADV_SIDE;
            else {
                dead_depth = NONE;
                restore = R_FORK; rslot = dstar;
                set_ctx = C_LEAF; set_pos = nd_gpos[dstar]; set_val = 1 - nd_fav[dstar];
                d = dstar + 1; first = false;
                mode = d < Dp ? M_DESC : M_TREE1;
                saved = false; own_pm = 10000.0;
                adv = ADV_NONE;
            }
        }
        if (adv == ADV_REPLAY) { // tree thread p: back to the masks saved at depth D, fresh messages, the unfavoured value (bpgd.cpp:501-523)
            restore = R_BK; set_ctx = C_BK; set_pos = bk_pos; set_val = bk_val;
            mode = M_TREE2; d = Dp + 1; iter = 0; first = true; adv = ADV_NONE;
        }
        if (adv == ADV_SIDE) { // the next side thread that was handed a snapshot (bpgd.cpp:527-570)
            if (ROLE == 1) break; // (owner: the side threads are tasks, queued when the main thread handed them their snapshots)
            for (++jside; jside < NS; ++jside) if (G.alt_depth[jside] == Dp + jside + 1) break;
            if (jside >= NS) break;
            ++sides_run;
            restore = R_SIDE; rslot = jside; set_ctx = C_SIDE0; set_pos = (int)G.dec_vn[jside]; set_val = (int)G.dec_val[jside];
            mode = M_SIDE; d = G.alt_depth[jside]; iter = 0; first = true; adv = ADV_NONE;
        }
        if (restore != R_NONE) {
            const uint8_t *src = restore == R_FORK ? fork0 + (int64_t)rslot * forkb : (restore == R_BK ? snap_b + rec : snap_b + (int64_t)(2 + rslot) * rec);
            if (ROLE == 2 && restore == R_CTX) src = rslot >= 0 ? ec->fork + (int64_t)rslot * forkb : ec->fork + (int64_t)nforks * forkb + (int64_t)(-1 - rslot) * rec16;
            gdg_snap_load<NT>(g, s, G, src);
            if (restore == R_FORK || (restore == R_CTX && rslot >= 0)) gdg_msg_load<NT, VF, DM>(s, st, nmsg, src + rec16); // (an unfavoured guess does not re-initialise the messages)
            restore = R_NONE;
        }
        if (set_ctx != C_NONE) { // vn_set_value + peel
            __syncthreads();
            if (tid < 64) {
                bool bad = (set_pos == NONE) ? true : gdg_set_value_wave(g, s, G.pos_lv[set_pos], set_val);
                if (!bad) bad = peel_wave<false, NT>(g, s);
                if (tid == 0) s.scal[1] = bad ? 1 : 0;
            }
            __syncthreads();
            const int ctx = set_ctx;
            set_ctx = C_NONE;
            if (s.scal[1] != 0) { // the threads that made this decimation end here
                if (ctx == C_LEAF) { dead_depth = d; adv = ADV_LEAF; }
                else if (ctx == C_DESC) { if (prev_with_main) want_main_end = true; dead_depth = d; adv = ADV_LEAF; }
                else if (ctx == C_MAIN) { want_main_end = true; adv = ADV_LEAF; }
                else if (ctx == C_TREE1) adv = saved ? ADV_REPLAY : ADV_LEAF;
                else if (ctx == C_BK || ctx == C_TREE2) adv = ADV_LEAF;
                else adv = ADV_SIDE;
                continue;
            }
        }
        EPT(3);
        // loop bounds of the thread bodies
        if (mode == M_DESC) {
            if (p == 0 && main_alive && d >= P.max_step) { // (max_step <= D: the main thread's loop is over, the tree threads go on)
                if (!main_conv) { gdg_store_err<NT>(s, G, new_n, ROLE == 1 ? (uint32_t *)(ec->vec + (int64_t)ec->nh * ec->vecb) : main_fb); ag_publish_barrier(); have_main_fb = true; }
                main_alive = false;
            }
        } else if (mode == M_MAIN) {
            if (d >= P.max_step) { want_main_end = true; adv = ADV_LEAF; continue; }
        } else if (mode == M_TREE1) {
            if (d >= P.max_tree_branch_step + Dp + 1) { adv = saved ? ADV_REPLAY : ADV_LEAF; continue; }
        } else if (mode == M_TREE2) {
            if (iter >= P.max_tree_branch_step) { adv = ADV_LEAF; continue; }
        } else if (iter >= P.max_side_branch_step) { adv = ADV_SIDE; continue; }
        // ---- one step: the BP block ...
        const bool desc = mode == M_DESC;
        const bool with_main = mode == M_MAIN || (desc && p == 0 && main_alive);
        const int ntree = desc ? (1 << (Dp - d)) - (p == 0 ? 1 : 0) : 0; // tree threads that share this node
        const int share = desc ? ntree + (with_main ? 1 : 0) : 1;
        int cv;
        {
            const int nlive = gdg_caches<NT, VF, DM, KG>(g, s, G, st, vc, cn, cmap);
            if (first) { bp_init<VF, DM>(s, vc); __syncthreads(); }
            first = false;
            EPT(0);
            cv = bp_run<NT, VF, DM, KG, false, false, false, !std::is_same_v<ST, GdgNoStatic>>(g, P, s, P.max_iter_per_step, nlive, vc, cn, hist_b, it, P.gdg_factor, dead_unsat, nullptr, h4, with_main);
            blocks += share; R.post_it += it * share; // counted once per thread of the ensemble that runs this block
            EPT(1);
        }
        // ... the scan (the main thread scans BEFORE its convergence test, bpgd.cpp:630-633; the others only go on when the block failed)
        double pm_t = 0.0;
        if (cv && desc && with_main) { // the tree threads below this node offer the vector as it is now
            pm_t = gdg_get_pm<NT>(g, s, G);
            if (ROLE == 0) { gdg_store_err<NT>(s, G, new_n, stash); ag_publish_barrier(); }
            else if (ntree > 0) emit(1, ntree, pm_t);
        }
        int gpos = NONE, favor = 0, rc = 0;
        if (cv && it < P.max_iter_per_step) { // (an early exit leaves the parity words re-armed with the syndrome bits by the iteration that noticed it:
            for (int l = tid; l < m; l += NT) s.par[l] = 0u; // what the scan below counts is the checks the last iteration left unmet -- none)
            __syncthreads();
        }
        if (!cv || with_main) {
            const bool sidethr = !(mode == M_MAIN || (desc && p == 0)); // thresholds (0, -10) once an unfavoured branch has been taken
            rc = gdg_select_core<NT>(g, P, s, G, hist_b, sidethr ? 0.0 : -3.0, sidethr ? -10.0 : (d == 0 ? -16.0 : -12.0), d, gpos, favor, st, h4);
        }
        EPT(2);
        if (cv) { // ---- a converged block: the strict-< offers, in the order main, tree threads by id, side threads
            const double pm = gdg_get_pm<NT>(g, s, G);
            int who = 0, count = 1;
            bool offer = true;
            if (desc) { who = with_main ? 0 : (p == 0 ? 1 : p); count = with_main ? 1 : ntree; }
            else if (mode == M_TREE1) { who = p; own_pm = pm; }
            else if (mode == M_TREE2) { who = p; offer = !(pm > own_pm); }
            else if (mode == M_SIDE) who = 1 + T + jside;
            if (with_main) main_conv = true;
            if (ROLE != 0) { if (offer) emit(who, count, pm); }
            else if (offer) {
                if (pm < best) {
                    best = pm; winner = who; ties = 0;
                    for (int j = tid; j < new_n; j += NT) G.best_err[j] = s.hard[G.pos_lv[j]];
                    __syncthreads();
                } else if (pm == best) { // same metric: a different vector makes the reference's answer timing dependent
                    bool diff = false;
                    for (int j = tid; j < new_n; j += NT) diff |= (G.best_err[j] != s.hard[G.pos_lv[j]]);
                    if (block_any<NT>(diff, s)) ties += count;
                }
            }
            if (ROLE == 0 && desc && with_main && ntree > 0) { // ... then the tree threads of the all-favoured path, with the stashed vector
                if (pm_t < best) {
                    best = pm_t; winner = 1; ties = 0;
                    for (int j = tid; j < new_n; j += NT) G.best_err[j] = vec_byte(stash, j);
                    __syncthreads();
                } else if (pm_t == best) {
                    bool diff = false;
                    for (int j = tid; j < new_n; j += NT) diff |= (G.best_err[j] != vec_byte(stash, j));
                    if (block_any<NT>(diff, s)) ties += ntree;
                }
            }
            if (with_main) main_alive = false; // (converged: nothing to keep for the fallback)
            if (desc) dead_depth = d;
            adv = mode == M_SIDE ? ADV_SIDE : ADV_LEAF;
            continue;
        }
        if (rc == -1 || gpos == NONE) { // the scan failed: the threads of this step end
            if (with_main) want_main_end = true;
            if (desc) dead_depth = d;
            adv = mode == M_SIDE ? ADV_SIDE : (mode == M_TREE1 && saved ? ADV_REPLAY : ADV_LEAF);
            continue;
        }
        // ---- the guess: save what later walks start from, then decide the favoured value (at the top of the next turn)
        {
            uint8_t *dst = nullptr;
            const bool to_ctx = ROLE == 1 && desc && d == Dp - 1;                               // (owner: the leaves below this fork are tasks)
            if (to_ctx) dst = ec->fork + (int64_t)(p >> 1) * forkb;
            else if (desc) dst = fork0 + (int64_t)d * forkb;                                   // fork: the unfavoured child starts here
            else if (mode == M_MAIN && d >= Dp && d < S)                                        // handed to side thread d - D
                dst = ROLE == 1 ? ec->fork + (int64_t)nforks * forkb + (int64_t)(d - Dp) * rec16 : snap_b + (int64_t)(2 + d - Dp) * rec;
            else if (mode == M_TREE1 && d == Dp) dst = snap_b + rec;                             // the tree thread's own way back
            if (dst) {
                gdg_snap_save<NT>(g, s, G, dst);
                if (desc) gdg_msg_save<NT, VF, DM>(s, st, nmsg, dst + rec16);
                if (tid == 0) {
                    if (to_ctx) { ag_st(&ec->node[2 * (p >> 1)], (uint32_t)gpos); ag_st(&ec->node[2 * (p >> 1) + 1], (uint32_t)favor); }
                    else if (desc) { nd_gpos[d] = gpos; nd_fav[d] = favor; }
                    else if (mode == M_MAIN && ROLE == 1) { uint32_t *sd = ec->node + 2 * nforks + 3 * (d - Dp); ag_st(&sd[0], (uint32_t)gpos); ag_st(&sd[1], (uint32_t)(1 - favor)); ag_st(&sd[2], (uint32_t)(d + 1)); }
                    else if (mode == M_MAIN) { const int j = d - Dp; G.dec_vn[j] = (int16_t)gpos; G.dec_val[j] = (int8_t)(1 - favor); G.alt_depth[j] = (int16_t)(d + 1); }
                }
                if (mode == M_TREE1) { bk_pos = gpos; bk_val = 1 - favor; saved = true; }
                ag_publish_barrier();
                {
                    if (ROLE == 1 && to_ctx && tid == 0) { // one token and one work item per tree thread below the fork (leaf 0 is the main thread: the owner goes on)
                        const int lo = p == 0 ? 1 : p, cnt = p == 0 ? 1 : 2;
                        __hip_atomic_fetch_add(&ec->hdr[0], (uint32_t)cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        for (int k = 0; k < cnt; ++k) ring_push(gp->q, gp->qmask, item_side(ec->id, lo + k));
                    }
                    if (ROLE == 1 && mode == M_MAIN && tid == 0) { // ... and one per side thread that is handed its snapshot
                        __hip_atomic_fetch_add(&ec->hdr[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ring_push(gp->q, gp->qmask, item_side(ec->id, 1 + T + d - Dp));
                    }
                }
            }
        }
        set_pos = gpos; set_val = favor; prev_with_main = with_main;
        set_ctx = desc ? C_DESC : (mode == M_MAIN ? C_MAIN : (mode == M_TREE1 ? C_TREE1 : (mode == M_TREE2 ? C_TREE2 : C_SIDE)));
        if (mode == M_TREE2 || mode == M_SIDE) ++iter;
        ++d;
        if (desc && d == Dp) { // the shared part of this path ends: the leaf's thread goes on alone
            if (p == 0) {
                if (main_alive) mode = M_MAIN;
                else { set_ctx = C_NONE; adv = ADV_LEAF; } // (no tree thread 0: nobody walks the all-favoured leaf)
            } else if (ROLE == 1) { set_ctx = C_NONE; adv = ADV_LEAF; } // (owner: both leaves below this fork are tasks)
            else { mode = M_TREE1; saved = false; own_pm = 10000.0; }
        }
    }
    __syncthreads();
    if (ROLE != 0) { // hand in: statistics, then this role's token; the last token queues FINAL
        ag_publish_barrier();
        if (tid == 0) {
            __hip_atomic_fetch_add(&ec->hdr[8], (uint32_t)blocks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&ec->hdr[9], (uint32_t)R.post_it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (sides_run) __hip_atomic_fetch_add(&ec->hdr[10], (uint32_t)sides_run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ROLE == 1) ag_st(&ec->hdr[11], have_main_fb ? 1u : 0u);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const uint32_t before = __hip_atomic_fetch_sub(&ec->hdr[0], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            if (before == 1u) ring_push(gp->q, gp->qmask, item_final(ec->id));
        }
        __syncthreads();
        return;
    }
    if (!(best < 9999.0) && have_main_fb) { // nothing converged: the main thread's vector as it left its loop (:677-682)
        for (int j = tid; j < new_n; j += NT) G.best_err[j] = vec_byte(main_fb, j);
        __syncthreads();
    }
    R.conv = best < 9999.0; R.pm = best;
    R.live_vn = 1 + T + sides_run; R.live_cn = blocks; R.live_e = winner; R.osd_rowadds = ties;
}

// FINAL item of a parked ensemble: the offers of every hypothesis in the oracle's order (main, tree threads by id, side threads),
// strict "<", ties of the winning metric with a different vector counted; fills R and s.hard like the ensemble on one workgroup.
template <int NT>
__device__ __forceinline__ void gdg_ens_finalize(const SwdGraphDev &g, const SwdDecodeParams &P, Lds &s, const EnsCtx &c, WinResult &R) {
    const int tid = threadIdx.x, n = g.n, new_n = g.new_n;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    R = WinResult{};
    R.exit_class = SWD_EXIT_POST;
    double best = 10000.0;
    int winner = -1, ties = 0;
    auto byte_of = [&](int who, int j) { const uint32_t *v = (const uint32_t *)(c.vec + (int64_t)who * c.vecb); return (uint8_t)((ag_ld(&v[j >> 2]) >> (8 * (j & 3))) & 0xFFu); };
    for (int who = 0; who < c.nh; ++who) {
        const uint32_t w0 = ag_ld(&c.etab[4 * who]);
        if (!(w0 & 1u)) continue;
        const int count = (int)(w0 >> 8);
        const double pm = __longlong_as_double((long long)((unsigned long long)ag_ld(&c.etab[4 * who + 2]) | ((unsigned long long)ag_ld(&c.etab[4 * who + 3]) << 32)));
        if (pm < best) { best = pm; winner = who; ties = 0; }
        else if (pm == best) {
            bool diff = false;
            for (int j = tid; j < new_n; j += NT) diff |= (byte_of(who, j) != byte_of(winner, j));
            if (block_any<NT>(diff, s)) ties += count;
        }
    }
    const int src = winner >= 0 ? winner : (ag_ld(&c.hdr[11]) ? c.nh : -1); // nothing converged: the main thread's vector as it left its loop
    for (int v = tid; v < n; v += NT) s.hard[v] = 0;
    __syncthreads();
    if (src >= 0)
        for (int j = tid; j < new_n; j += NT)
            if (byte_of(src, j)) { const uint32_t pw = ag_ld(&c.pos[j >> 1]); s.hard[(j & 1) ? (pw >> 16) : (pw & 0xFFFFu)] = 1; }
    __syncthreads();
    const int T = (1 << P.max_tree_depth) - 1;
    R.conv = best < 9999.0; R.pm = best;
    R.pre_it = (int)ag_ld(&c.hdr[12]); R.post_it = (int)ag_ld(&c.hdr[9]); R.total_it = R.pre_it + R.post_it;
    R.live_vn = 1 + T + (int)ag_ld(&c.hdr[10]); R.live_cn = (int)ag_ld(&c.hdr[8]); R.live_e = winner; R.osd_rowadds = ties;
}

// bpgdg_decoder.decode / bpgd_decoder.decode / bp_history_decoder for one syndrome.  On return
// s.hard[0..n) is the returned vector.
// par != nullptr (parallel form): a tree with side branches is parked in a context and its side branches are queued;
// then R.exit_class = -2 on return and the result arrives later as a FINAL item (gdg_finalize).
#ifdef SWD_GDGPROF // diagnostic build (scripts/gdg_phase_profile.py): 100 MHz ticks per kind of work inside a tree walk
#define GPT0() long long gp_t_ = wall_clock64()
#define GPT(k) do { const long long gp_n_ = wall_clock64(); R.gp[k] += gp_n_ - gp_t_; gp_t_ = gp_n_; } while (0)
#else
#define GPT0() do { } while (0)
#define GPT(k) do { } while (0)
#endif
// The window's edge columns (row_col) and check order (perm) staged in LDS, in the slot-list region the guessing decoders no
// longer use: the reset after the sort chases slot -> column for every position of every check and the peeling sweeps look both
// up for every check they fire -- from L2 that was one memory latency per look-up.  gl: the graph descriptor the tree walk uses
// from here on (pointers into LDS; generic loads).  The caller's next barrier publishes the copies.
template <int NT>
__device__ __forceinline__ void gdg_stage_graph(SwdGraphDev &gl, const SwdGraphDev &g, Lds &s) {
    const int E = g.E, m = g.m, Ee = (E + 1) & ~1;
    if ((Ee + m) * 2 <= g.K * m * 2) { // (uniform; E <= K m always, the check order needs m entries more)
        uint16_t *rc = s.lslot, *pm = rc + Ee;
        for (int e = threadIdx.x; e < E; e += NT) rc[e] = g.row_col[e];
        for (int l = threadIdx.x; l < m; l += NT) pm[l] = g.perm[l];
        gl.row_col = rc; gl.perm = pm;
    }
}

// the descriptor gdg_stage_graph set up, again (the copies are in LDS already)
__device__ __forceinline__ void gdg_staged_graph(SwdGraphDev &gl, const SwdGraphDev &g, const Lds &s) {
    const int Ee = (g.E + 1) & ~1;
    if ((Ee + g.m) * 2 <= g.K * g.m * 2) { gl.row_col = s.lslot; gl.perm = s.lslot + Ee; }
}

// ENS (kernel kind 7): bpgdg_decoder(multi_thread=True) -- the post-processing is the reference's threaded ensemble
// (gdg_ensemble_tree, run by the caller) instead of gdg()'s tree walk.
template <int NT, int VF, int DM, int KG, bool ENS = false, int VFP = VF>
__device__ __forceinline__ void decode_window_gdg(const SwdGraphDev &g_in, const SwdLdsLayout &L, const SwdDecodeParams &P, Lds &s,
                                                  const uint8_t *synd, double *hist_b, uint8_t *snap_b, WinResult &R,
                                                  const SwdPipeArgs *par = nullptr, uint32_t *acc = nullptr, int wi = 0, int b = 0) {
    SwdGraphDev g_loc = g_in; // (row_col / perm move into LDS for the tree walk: gdg_stage_graph)
    const SwdGraphDev &g = g_loc;
    const int tid = threadIdx.x, m = g.m, n = g.n, new_n = g.new_n;
    GdgLds G;
    gdg_bind(G, gdg_lds_base(s, L), L, n, new_n);
#pragma unroll
    for (int i = 0; i < 9; ++i) R.t[i] = 0;
#ifdef SWD_GDGPROF
    for (int i = 0; i < 5; ++i) R.gp[i] = 0;
#endif
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
    GdgVC<VF, DM> vc;
    GdgCC<KG> cn;
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
    // (serial form only: 16384 shots 1.70 -> 1.76 M windows/s; in the parallel form the owner runs the main branch alone and parks
    // the tree -- staging 13 KB for that costs more than it saves, 1.21 -> 1.20 M at 4096 shots -- and neither do the tasks repay it)
    if (!par || ENS) gdg_stage_graph<NT>(g_loc, g_in, s); // (the ensemble's owner walks the shared steps, the main thread and the side threads: it repays the staging; gdg_staged_graph)
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
    // Only the first new_n sorted positions are ever looked at (the rest is decided 0): select the new_n smallest (key, index)
    // pairs -- ties to the lowest indices, like the stable sort -- and sort those alone (57 -> ~25 us per window that goes into
    // the tree walk; the full sort remains when the scratch region cannot hold the second pair of arrays).
    int npad2 = 2;
    while (npad2 < new_n) npad2 <<= 1;
    if (new_n < n && L.off_aux + 3072 + npad2 * 10 <= (g.E + 1) * 8) {
        select_smallest<NT>(key, n, new_n, (int *)s.aux, s); // vn_val[v] = 0 for everything but the new_n smallest; ends with a barrier
        uint64_t *key2 = (uint64_t *)(s.aux + 3072);
        uint16_t *idx2 = (uint16_t *)(key2 + npad2);
        const int ch = (n + NT - 1) / NT;
        const int v0 = tid * ch, v1 = min(n, v0 + ch);
        int cnt = 0;
        for (int v = v0; v < v1; ++v) cnt += (s.vn_val[v] == -1) ? 1 : 0;
        int tot;
        int pos = block_exscan<NT>(cnt, s, tot);
        for (int v = v0; v < v1; ++v) {
            if (s.vn_val[v] == -1) { key2[pos] = key[v]; idx2[pos] = (uint16_t)v; ++pos; }
            else { s.hard[v] = 0; G.bp_hard[v] = 0; }
        }
        for (int i = new_n + tid; i < npad2; i += NT) { key2[i] = ~0ull; idx2[i] = 0xFFFF; }
        __syncthreads();
        sort_pairs<NT>(key2, idx2, npad2);
        for (int i = tid; i < new_n; i += NT) G.pos_lv[i] = idx2[i];
    } else {
        sort_pairs<NT>(key, idx, L.npad);
        for (int i = tid; i < n; i += NT) {
            const int v = idx[i];
            if (i < new_n) G.pos_lv[i] = (uint16_t)v;
            else { s.vn_val[v] = 0; s.hard[v] = 0; G.bp_hard[v] = 0; }
        }
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
        const bool bad = peel_wave<false, NT>(g, s);
        if (tid == 0) s.scal[1] = bad ? 1 : 0;
    }
    __syncthreads();
    if (s.scal[1]) { // BPGD::reset failed: decode returns the BP vector with cols[new_n:] zeroed
        // (the threaded ensemble returns its zero-initialised min_pm_error instead: bp_guessing_decoder.pyx:247-251 copy it
        // over the first new_n sorted columns whatever happened, bpgd.cpp:619-625 return before anything is written to it)
        for (int v = tid; v < n; v += NT) s.hard[v] = ENS ? (uint8_t)0 : G.bp_hard[v];
        __syncthreads();
        R.exit_class = SWD_EXIT_FAIL_PEEL;
        return;
    }
    if constexpr (ENS) { // bpgdg_decoder(multi_thread=True): the caller (pipeline_kernel) runs the threaded ensemble from here -- the walk
        // (gdg_ensemble_tree) is inlined at ONE place of the kernel, where units and the tasks of parked ensembles meet
        for (int j = tid; j < SWD_GDG_MAXGUESS; j += NT) G.alt_depth[j] = -1;
        __syncthreads();
        R.exit_class = -3; R.live_vn = dead_unsat ? 1 : 0;
        return;
    }
    GdgVC<VFP, DM> vcp; // the shortened graph's cache (the full graph's is dead from here on)
    using ST = GdgStatic<VFP, DM>;
    constexpr bool SPARSE = !std::is_same_v<ST, GdgNoStatic>;
    ST vst;
    gdg_static_init<NT, VFP, DM>(g, s, G, vst);
    double h4[VFP][4] = {}; // history of the last four iterations of a block, for this thread's positions (static cache only)
    const GdgCnMap cmap = gdg_cn_map<NT, KG>(g, s, G); // (the degrees after reset + peeling bound every later state of this window)
    int nlive = gdg_caches<NT, VFP, DM, KG>(g, s, G, vst, vcp, cn, cmap);
    bp_init<VFP, DM>(s, vcp);
    __syncthreads();

    double min_pm = 10000.0;
    int used_guess = 0, min_converge_depth = P.max_step, converge = 0, blocks = 0;
    const bool gdg = (P.kind == 1);
    // parallel form: the tree gets a context (snapshots, records) if one is free; without one it is walked serially
    int ctxid = -1;
    GdgCtx ctx{};
    if (gdg && par && !ENS) {
        __syncthreads();
        if (tid == 0) { uint32_t id = 0; acc[1] = ring_pop(par->gdgp.fq, par->gdgp.fmask, &id) ? id : 0xFFFFFFFFu; }
        __syncthreads();
        ctxid = (int)acc[1];
        if (ctxid >= 0) { ctx = gdg_ctx(par->gdgp, ctxid); snap_b = ctx.snap; }
#ifdef SWD_GDG_CHECKS
        if (ctxid >= 0 && tid == 0) {
            uint32_t *chk_status = par->gdgp.chk_status;
            GDG_CHECK(ctxid < par->gdgp.nctx, 13);
            const uint32_t st = atomicExch(&ctx.hdr[1], 3u);
            GDG_CHECK(st != 1u && st != 3u, 14); // a context handed out while a parked tree or another main branch owns it
        }
#endif
    }
    // ---- phase 1: main branch
    GPT0();
    for (int depth = 0; depth < P.max_step; ++depth) {
        if (depth > 0) nlive = gdg_caches<NT, VFP, DM, KG>(g, s, G, vst, vcp, cn, cmap);
        GPT(0);
        const int cv = bp_run<NT, VFP, DM, KG, false, false, false, SPARSE>(g, P, s, P.max_iter_per_step, nlive, vcp, cn, hist_b, it, P.gdg_factor, dead_unsat, nullptr, h4);
        GPT(1);
        ++blocks; R.post_it += it;
        if (cv) {
            converge = 1; min_converge_depth = depth;
            min_pm = gdg_get_pm<NT>(g, s, G);
            for (int j = tid; j < new_n; j += NT) G.best_err[j] = s.hard[G.pos_lv[j]];
            GPT(4);
            break;
        }
        const int rc = gdg ? gdg_select_vn<NT>(g, P, s, G, hist_b, false, depth, min_converge_depth, used_guess, snap_b, nullptr, vst, h4)
                           : gdg_decimate_reliable<NT>(g, s, G, hist_b, vst, h4);
        GPT(2);
        if (rc == -1) break;
    }
    if (!converge)
        for (int j = tid; j < new_n; j += NT) G.best_err[j] = s.hard[G.pos_lv[j]];
    __syncthreads();
    // ---- phase 2, parallel form: park the tree, queue its side branches, go on with other work items
    const bool replayed = false;
    if (ctxid >= 0) {
        const SwdGdgPar &gp = par->gdgp;
        int ninit = 0;
        for (int i = 0; i < used_guess; ++i) ninit += (G.alt_depth[i] <= min_converge_depth) ? 1 : 0;
        if (ninit > 0 && P.max_side_branch_step > 0) {
            uint32_t *h = ctx.hdr;
            if (tid == 0) {
                ag_st(&h[0], 0u); ag_st(&h[1], 1u); ag_st(&h[2], (uint32_t)used_guess); ag_st(&h[3], 0u);
                ag_st(&h[4], (uint32_t)wi); ag_st(&h[5], (uint32_t)b); ag_st(&h[6], dead_unsat ? 1u : 0u);
                ag_st(&h[7], (uint32_t)min_converge_depth);
                ag_st(&h[8], 0u); ag_st(&h[9], (uint32_t)used_guess); ag_st(&h[10], (uint32_t)used_guess);
                ag_st(&h[11], (uint32_t)min_converge_depth); ag_st(&h[12], (uint32_t)converge); ag_st(&h[13], (uint32_t)blocks);
                ag_st(&h[14], (uint32_t)R.post_it); ag_st(&h[15], (uint32_t)SWD_GDG_SLOTS);
                const unsigned long long pb = (unsigned long long)__double_as_longlong(min_pm);
                ag_st(&h[16], (uint32_t)pb); ag_st(&h[17], (uint32_t)(pb >> 32));
                ag_st(&h[18], 0u); ag_st(&h[19], 0u); ag_st(&h[20], 0u); ag_st(&h[21], 0u); ag_st(&h[22], 0u);
                ag_st(&h[23], (uint32_t)min_converge_depth); ag_st(&h[24], (uint32_t)R.pre_it); ag_st(&h[25], (uint32_t)used_guess);
            }
            for (int k = tid; k < SWD_GDG_SLOTS / 4; k += NT) // serial stack order so far: the main branch's snapshots
                ag_st(&h[26 + k], (uint32_t)(4 * k) | ((uint32_t)(4 * k + 1) << 8) | ((uint32_t)(4 * k + 2) << 16) | ((uint32_t)(4 * k + 3) << 24));
            for (int i = tid; i < (new_n + 1) / 2; i += NT)
                ag_st(&ctx.pos[i], (uint32_t)G.pos_lv[2 * i] | ((2 * i + 1 < new_n) ? ((uint32_t)G.pos_lv[2 * i + 1] << 16) : 0u));
            for (int i = tid; i < used_guess; i += NT) {
                uint32_t *r = gdg_rec(ctx, i);
                ag_st(&r[0], (uint32_t)(uint8_t)G.alt_depth[i] | ((uint32_t)(uint8_t)G.dec_val[i] << 8) | ((uint32_t)(uint16_t)(G.dec_vn[i] + 1) << 16));
                ag_st(&r[1], 0u);
            }
            {   // the main branch's own result
                uint32_t *dst = ctx.err + (int64_t)SWD_GDG_SLOTS * ctx.err_words;
                for (int i = tid; i < (new_n + 3) / 4; i += NT) {
                    uint32_t x = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) { const int q = 4 * i + k; if (q < new_n) x |= (uint32_t)(G.best_err[q] ? 1 : 0) << (8 * k); }
                    ag_st(&dst[i], x);
                }
            }
            ag_publish_barrier();
            if (tid == 0) gdg_sched(gp, P, ctx, ctxid, false); // queues the first side branches
            __syncthreads();
            R.exit_class = -2;
            return;
        }
    }
    // ---- phase 2, serial form: side branches in stack order (pyx:301-335)
    for (int i = 0; gdg && !replayed && i < used_guess; ++i) {
        int depth = G.alt_depth[i];
        if (depth > min_converge_depth) continue;
        gdg_snap_load<NT>(g, s, G, snap_b + (int64_t)i * gdg_snap_bytes(m, new_n));
        if (tid < 64) {
            const int gp = G.dec_vn[i];
            bool bad = (gp < 0) ? true : gdg_set_value_wave(g, s, G.pos_lv[gp], G.dec_val[i]);
            if (!bad) bad = peel_wave<false, NT>(g, s);
            if (tid == 0) s.scal[1] = bad ? 1 : 0;
        }
        __syncthreads();
        GPT(3);
        if (s.scal[1]) continue;
        for (int j = 0; j < P.max_side_branch_step; ++j) {
            depth = G.alt_depth[i] + j;
            nlive = gdg_caches<NT, VFP, DM, KG>(g, s, G, vst, vcp, cn, cmap);
            if (j == 0) { bp_init<VFP, DM>(s, vcp); __syncthreads(); } // set_masks re-initialises the messages
            GPT(0);
            const int cv = bp_run<NT, VFP, DM, KG, false, false, false, SPARSE>(g, P, s, P.max_iter_per_step, nlive, vcp, cn, hist_b, it, P.gdg_factor, dead_unsat, nullptr, h4);
            GPT(1);
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
            const int rc2 = gdg_select_vn<NT>(g, P, s, G, hist_b, true, depth, min_converge_depth, used_guess, snap_b, nullptr, vst, h4);
            GPT(2);
            if (rc2 == -1) break;
        }
        __syncthreads();
        GPT(4);
    }
    __syncthreads();
    if (!replayed) {
        for (int v = tid; v < n; v += NT) s.hard[v] = 0;
        __syncthreads();
        for (int j = tid; j < new_n; j += NT) s.hard[G.pos_lv[j]] = G.best_err[j];
        __syncthreads();
    }
#ifdef SWD_GDG_CHECKS
    if (ctxid >= 0 && tid == 0) { uint32_t *chk_status = par->gdgp.chk_status; GDG_CHECK(atomicExch(&ctx.hdr[1], 0u) == 3u, 15); }
#endif
    if (ctxid >= 0 && tid == 0) ring_push(par->gdgp.fq, par->gdgp.fmask, (uint32_t)ctxid); // walked here after all: the context goes back
    R.conv = converge; R.pm = min_pm; R.total_it = R.pre_it + R.post_it;
    R.live_vn = used_guess; R.live_cn = blocks; R.live_e = min_converge_depth;
    R.exit_class = SWD_EXIT_POST;
    R.t[5] = wall_clock64();
}
