// One-workgroup-per-shot decode of a window: the whole of osd_window.decode
// (/root/reference/src/osd_window.pyx:158-199) for one syndrome --
//   pre-processing min-sum BP  -> LLR-history sort -> shortening (decimate + peel)
//   -> post-processing masked BP -> OSD on the ordered matrix --
// with the messages of the shot living in LDS for its whole lifetime.
//
// Numerics: fp64, no FMA contraction (build with -ffp-contract=off), every sum in the
// reference's order, so hard decisions, iteration counts and min_pm are bit-identical.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "swd.h"
#include "swd_graph.h"

#define SWD_DMAX 10 // largest column degree any kernel variant of this build supports

// Translation units of the osd_window kernels (swd_kernels_k0 / k3) set SWD_OSDW_TUNED: their kernels of up to 256 threads
// keep the BP register caches packed (VnCache / CnCache, P16) and -- variants with at most twelve groups of check
// positions -- are built for three waves per SIMD (168 VGPRs); together with the LDS diet of the osd_window layout
// (decided-node bits, 48-bit live masks, parity bytes, residual syndrome of the window's rows only; DIET below) three
// workgroups of the <256, 7, 6, 9> kernel fit a CU (53 600 B of LDS each; the hardware's limit is 53 760, not the
// 54 592 the occupancy API accepts: scripts/residency_check.py).
#ifndef SWD_OSDW_TUNED
#define SWD_OSDW_TUNED 0
#endif
#ifndef SWD_TUNED_NT // largest workgroup of the tuned osd_window kernels (16-bit LDS offsets, LDS diet); the host uses the same bound
#define SWD_TUNED_NT 256
#endif
#define SWD_P16(NT) (SWD_OSDW_TUNED && (NT) <= SWD_TUNED_NT)
// experiment builds only (scripts/devbuild.sh -DSWD_POST_RENUM=1 + SWD_POST_RENUM=1 in the environment): the shortened graph's message
// cells renumbered one column per live variable node in the tuned kernels too (the large-graph kernels always do it)
#ifndef SWD_POST_RENUM
#define SWD_POST_RENUM 0
#endif
#ifndef SWD_POST_DEPTH2
#define SWD_POST_DEPTH2 0
#endif
#ifndef SWD_BP_HARD_DEFER
#define SWD_BP_HARD_DEFER 1
#endif
#ifndef SWD_BP_PAR_INC
#define SWD_BP_PAR_INC 1
#endif
// Round-5 experiments on the iteration loop's dependent LDS round trips (bp_run):
//   SWD_BP_XARG_TRACK        1: the sign of a check's first-minimum position comes out of the sign shift registers (one compare-select
//                               more per position) instead of a re-read of the message before the write phase
//   SWD_BP_FLAG_MERGE_MAXVF  the convergence flags of block_any are read together with the first node's messages in caches of at
//                            most this depth (0: never; deep caches pay for the twelve registers held across the exit test with spills)
// Both measured on the headline (gpurun_out/r05f: 10.08 ms per launch without, 10.14 with the merge, 10.20 with the tracking, 10.17 with
// both) and left off: the loops are bound by the CU's LDS pipeline, not by these round trips.
#ifndef SWD_BP_XARG_TRACK
#define SWD_BP_XARG_TRACK 0
#endif
#ifndef SWD_BP_FLAG_MERGE_MAXVF
#define SWD_BP_FLAG_MERGE_MAXVF 0
#endif
// SWD_POST_SORTED (tuned osd_window kernels, round 5): the shortened graph's live nodes are listed by decreasing live degree, a node's
// live edges are moved to the front of its cache, the message cells are renumbered one column per listed node (cell(k, i) = k x nlive + i)
// and a wave's variable-node pass skips the positions beyond the largest live degree among its nodes (in steps of two) -- the
// iterations are LDS-bound and the live degree averages 2.65 of 6 positions.
// (bit 2, the large-graph kernels: measured on the 936 x 8784 model, gpurun_out/r05q -- 3.54 against 3.67 us per iteration of the shortened
// graph, paid for by 26 us more in the shortening step with its degree bytes and lists in HBM: no net gain, off)
#ifndef SWD_POST_SORTED
#define SWD_POST_SORTED 3
#endif
// SWD_FULL_SORTED (osd_window kernels, round 5): the full-graph phase serves its nodes in the graph's listed order (SwdGraphDev::vperm:
// degree tiers of two, heaviest first) and its variable-node pass is tiered like the shortened graph's.
// Measured on the headline (gpurun_out/r05l): bit-exact, 30 % fewer message positions in the full-graph variable-node pass (7296 of
// 10368 per iteration of the [[144,12,12]] mid window) -- and 10.01 ms per launch against 9.80: the three bodies per cache row, the row
// caps and the node numbers cost the check pass six more instructions per group of four.  Off.
// (bit 1, the large-graph kernels, whose full-graph messages live in HBM: 40.1 against 33.9 us per iteration of the 936 x 8784 model --
// the listed order scatters what were neighbouring 8-byte accesses; off too)
#ifndef SWD_FULL_SORTED
#define SWD_FULL_SORTED 0
#endif
// SWD_CN_HALF (round 5): the check pass of the SHORTENED graph walks its positions in groups of four whose second half is skipped when
// no lane of the wave has a position there (cn_assign bounds a thread's walk by T = 3, 4, 6, 8, 12 ...: T = 6 costs 6 reads and writes
// instead of 8).  Measured on the headline (gpurun_out/r05m): bit-exact and 10.45 ms per launch against 9.95 -- the scalar branch between
// the two halves keeps the scheduler from issuing a group's four reads together; off.
#ifndef SWD_CN_HALF
#define SWD_CN_HALF 0
#endif
#ifndef SWD_TIER_STEP // positions per tier of the sorted form (tiers of one position -- six bodies of the pass per cache row, a second prefix sum for the order -- measured slower: 10.0 against 9.84 ms per headline launch)
#define SWD_TIER_STEP(DM) 2
#endif
#ifndef SWD_POST_KGP // groups of four check positions in the shortened graph's register cache (0: as many as for the full graph)
#define SWD_POST_KGP 0
#endif

struct SwdLdsLayout {
    int32_t off_livemask, off_par, off_lv, off_jptr, off_lslot, off_cnval, off_cndeg, off_cndeg0, off_vnval, off_hard,
        off_misc, total;
    int32_t npad;      // power of two >= n (bitonic sort size)
    int32_t off_idx;   // inside scratch: u16 idx[npad] after u64 key[npad]
    int32_t off_aux;   // inside scratch: first byte after the sort arrays
    int32_t off_cs;    // inside scratch: arrays of the higher-order OSD sweep
    int32_t cs_par;    // threads that evaluate OSD candidates concurrently (a multiple of 64, <= threads per shot)
    int32_t off_gdg;   // guessing decoders: persistent per-shot arrays (outside scratch), -1 if unused
    int32_t off_cord;  // inside scratch: degree histogram + check order of the post phase (clear of lslot and the keys' tail)
    int32_t off_rc;    // inside scratch: u16 row_col[E] staged for the shortening step
    int32_t off_bak;   // inside scratch: state backup of the parallel peel (10 m + 2 n bytes)
    int32_t off_hs;    // inside scratch: f64 hs[n], summed posterior history of the live VNs after a failed post phase (HACC kernels)
    int32_t off_oslot; // inside scratch (tail of the sort keys): exchange slots of the column-form elimination (osd0_cols), -1 if unused
    int32_t off_oring; // inside scratch (behind the OSD-0 arrays): ring of row operations of the four-wave elimination (osd0_quad), -1 if unused
    int32_t off_owide, owide_ring; // large-graph kernels: LDS offset of the ring / masks / control words of osd0_colsw and its entries, -1 if unused
    // Large graphs (kernels instantiated with BIG, swd_kernels_k5.hip): the scratch region -- messages, sort keys, staged lists,
    // OSD arrays -- lives in HBM, big_scratch bytes per workgroup, and every other offset above (off_livemask ... off_misc, total)
    // is relative to the workgroup's LDS, which then only holds the per-check / per-variable-node state.  0: everything in LDS.
    int32_t big_scratch;
    // ... and what still fits beside that state goes back into LDS: off_pmsg >= 0 = a region of pmsg_bytes for
    //   post_lds  the messages of the shortened graph, renumbered one column of cells per live variable node
    //             (cell(k, i) = k * nlive + i for edge position k of the i-th live node: (column weight) x new_n cells)
    //   osd_lds   the arrays of the OSD phase that start at off_aux (transform matrix, pivots, ordered lists)
    int32_t off_pmsg, pmsg_bytes, post_lds, osd_lds;
};

// decoder parameters shared by every window of a launch (osd_window.pyx:10-16)
struct SwdDecodeParams {
    int32_t pre_iter, post_iter, osd_method, osd_order;
    double alpha;
    int32_t hist_is_state; // history buffer carries state in and out (single-shot decode())
    int32_t zero_hist;     // start every decode from a zero history
    int32_t record_all;    // store the posterior of every iteration (history is an output)
    // guessing decoders (bp_guessing_decoder.pyx): kind 0 = osd_window, 1 = bpgdg (single-thread gdg),
    // 2 = bpgd, 3 = plain bp_history_decoder.  pre_iter / alpha double as max_iter / ms_scaling_factor.
    int32_t kind;
    int32_t max_iter_per_step, max_step, max_tree_depth, max_side_depth, max_side_branch_step, low_error_mode;
    int32_t max_guess;
    double gdg_factor;
    int32_t max_tree_branch_step; // threaded ensemble (kernel kind 7): steps of a tree thread after its last split
    int32_t pad_;
};

// one window of the sliding-window plan: its graph + where it sits in the global DEM
struct SwdWindowDev {
    SwdGraphDev g;
    SwdLdsLayout L;
    int32_t row0;   // first detector row        (anchors[t].row, osd.py:139)
    int32_t col0;   // first global fault column (anchors[t].col)
    int32_t commit; // leading columns committed after decoding (osd.py:140,170-173)
    int32_t pad;
    const uint32_t *cn_map; // nullable [NT]: static check-to-thread map of the full-graph phase (check | rank << 16 | threads << 18, 0xFFFF = none)
};



// Parallel form of the guessing decoder's tree search (swd_gdg_kernel.h): task queue + per-owner contexts in HBM
struct SwdGdgPar {
    uint32_t *q;          // null: serial form.  Work-item ring: [0] head [1] tail [2] window units completed [3] -, then u64 entries
    uint32_t qmask;       // ring capacity - 1
    uint32_t *fq;         // ring of free context ids
    uint32_t fmask;
    int32_t ensemble;     // every hypothesis counts (multi_thread=True): no pruning / capacity in the replay
    int32_t inflight_max; // side branches of one tree queued or running at a time
    uint8_t *ctx;         // [nctx][ctx_stride] contexts of parked trees
    int64_t ctx_stride;
    int32_t off_pos, off_rec, off_err, err_stride;
    int32_t off_node;     // threaded ensemble (kind 7): table of the forks at depth D - 1 (guess, favoured value)
    int32_t ens_hyps;     // ... and the number of hypotheses 1 + (2^D - 1) + (S - D) its offer table holds
    uint8_t *csnap;       // [nctx][csnap_stride] their snapshot areas
    int64_t csnap_stride;
    int32_t nctx;
    int32_t shots_inflight; // shots admitted at a time (a finished shot admits the next)
    int32_t static_bound; // diagnostics (SWD_GDG_STATIC_BOUND): side branches prune against the main branch's min_converge_depth only
    uint32_t *chk_status; // diagnostic builds (SWD_GDG_CHECKS): invariant violations, bits 8..
};

struct SwdPipeArgs {
    const SwdWindowDev *wins;
    int32_t W, B;
    SwdDecodeParams P;
    const uint8_t *det;       // [B][det_stride] detector bits (window 0 reads rows row0..)
    int64_t det_stride;
    int32_t num_det;
    int32_t off_det;          // LDS offset of the residual-syndrome bytes
    uint8_t *total;           // nullable [B][total_stride] committed faults (total_e_hat)
    int64_t total_stride;
    const uint32_t *chk_colptr; // CSC of the global check matrix (residual update, osd.py:178)
    const uint16_t *chk_rows;
    uint8_t *win_out;         // nullable [B][win_out_stride]: full estimate of window W-1
    int64_t win_out_stride;
    int32_t *stats;           // nullable [B][W][SWD_STAT_WORDS]
    double *min_pm;           // nullable [B][W]
    double *hist;             // [B][4][nmax]
    int64_t hist_stride;      // doubles per shot
    uint8_t *osd0;            // nullable [B][n] (single-window use)
    uint8_t *bp_dec;          // nullable [B][n] (single-window use): BP hard decisions when the OSD takes over (property bp_decoding)
    uint8_t *snap;            // guessing decoders: [B][snap_stride] snapshot stack in HBM
    int64_t snap_stride;
    const uint32_t *obs_mask; // nullable [num_col]: bit k set if fault flips observable k (obs matrix)
    int32_t *shot_result;     // nullable [B][2]: predicted observable flips, residual syndrome != 0
    int64_t *prof;            // nullable [B][W][8]: 100 MHz ticks per phase (diagnostics only)
    // work-unit scheduling (one workgroup = one window of one shot, see pipeline_kernel)
    uint32_t *sched;          // [2 + B]: ticket counter, then per shot the number of windows finished, then the fault word of THIS launch (all zeroed per launch)
    uint32_t *status;         // one word owned by the decoder, never reset by a launch: bit 0 = a window gave up waiting for its predecessor (mirrored in sched[B + 1])
    uint8_t *state;           // [B][state_stride]: residual syndrome + accumulators handed to the next window
    int64_t state_stride;
    int32_t slot_scratch;     // hist / snap are private to the workgroup (indexed by blockIdx.x), not to the shot
    uint8_t *big;             // BIG kernels: [gridDim.x][big_stride] scratch regions of the workgroups in HBM
    int64_t big_stride;
    const uint32_t *order;    // nullable [B]: the shots in the order they are started (heaviest syndrome first: shot_order_kernel)
    SwdGdgPar gdgp;
};

namespace swd {

struct Lds {
    double *msg;        // scratch region start
    char *scratch;
    char *aux;          // scratch + off_aux (BIG kernels: possibly the LDS region of the layout instead)
    uint64_t *livemask; // [m]  (diet form: u32 [m] + u16 [m], see lm_get)
    int lm_m;           // m (diet form)
    uint32_t *par;      // [m]
    uint16_t *lv;       // [new_n]
    uint16_t *jptr;     // [K+1]
    uint16_t *lslot;    // [K*m] post phase: lslot[k*m + l] = k-th live edge slot of check lane l
    int8_t *cn_val;     // [m]  residual check value, -1 = cleared
    uint8_t *cn_deg;    // [m]  live degree
    uint8_t *cn_deg0;   // [m]  original degree
    int8_t *vn_val;     // [n]  -1 live / decided value
    uint8_t *hard;      // [n]  bp_decoding
    int *flags;         // [32]
    int *scal;          // [32]
    double *dbl;        // [24]
    int *iaux;          // [32]
    int fpar;
    // large-graph kernels, full-graph phase (SWD_BIG_HYBRID): message cells at byte offsets >= msg_lo live in LDS -- at msg_alt +
    // (offset - msg_lo) -- the others in the HBM scratch region as ever
    char *msg_alt;
    uint32_t msg_lo;
    // ... and (SWD_BIG_REC) the check-to-bit messages are not stored at all: a check leaves ONE record of 32 bytes in LDS -- the two
    // magnitudes, the sign bits of its positions (flip applied), the position of the first minimum -- and a variable node rebuilds
    // the message of an edge from the record of the edge's check; the message array only ever holds bit-to-check messages
    __attribute__((address_space(3))) char *rec; // (an LDS pointer by type: ds_read / ds_write, not flat accesses)
    // work assignment ids (== threadIdx.x up to a permutation of the waves, see swd_wave_roles): ctid picks
    // the check a thread serves, vtid its variable nodes
    int ctid, vtid;
};

// The CN pass of a wave costs its heaviest check, and checks are dealt to the waves heaviest first, so
// wave "role 0" is always the slowest.  Two shots share a CU; if both put role 0 on the same SIMD that
// SIMD is the bottleneck while the others idle.  Roles are therefore taken from the hardware SIMD the
// wave sits on, shifted by the wave slot the workgroup's first wave got (co-resident workgroups get
// different slots), and the VN work is dealt in the opposite order.  Any bijection is correct; if the
// four waves do not sit on four different SIMDs the identity is used.  word: 2 ints of LDS.
template <int NT>
__device__ __forceinline__ void swd_wave_roles(Lds &s, uint32_t *word) {
    const int tid = threadIdx.x;
    s.ctid = s.vtid = tid;
    if constexpr (NT == 256) {
        const int wave = tid >> 6, lane = tid & 63;
        const uint32_t simd = __builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4); // HW_ID.SIMD_ID
        const uint32_t slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4); // HW_ID.WAVE_ID
        if (tid == 0) { word[0] = 0; word[1] = slot; }
        __syncthreads();
        if (lane == 0) atomicOr(&word[0], 1u << simd);
        __syncthreads();
        if (word[0] == 0xFu) {
            const int role = (int)((simd + 2u * (word[1] & 1u)) & 3u);
            s.ctid = role * 64 + lane;
            s.vtid = (3 - role) * 64 + lane;
        }
        (void)wave;
        __syncthreads();
    }
}

template <int NT>
__device__ __forceinline__ bool block_any(bool p, Lds &s) {
    constexpr int NW = NT / 64;
    const int par = (s.fpar++) & 1;
    unsigned long long b = __ballot(p);
    if ((threadIdx.x & 63) == 0) s.flags[par * 16 + (threadIdx.x >> 6)] = (b != 0ull);
    __syncthreads();
    int r = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) r |= s.flags[par * 16 + w];
#ifdef SWD_NO_SCALAR_ANY
    return r != 0;
#else
    return __builtin_amdgcn_readfirstlane(r) != 0; // the same on every lane: callers branch on it with scalar branches
#endif
}

// exclusive prefix sum over the block; `total` = sum of all. Two barriers.
template <int NT>
__device__ __forceinline__ int block_exscan(int x, Lds &s, int &total) {
    constexpr int NW = NT / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int y = __shfl_up(incl, d, 64);
        if (lane >= d) incl += y;
    }
    int *wsum = s.scal + 8; // [NW]
    __syncthreads();
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        int c = wsum[w];
        if (w < wave) off += c;
        tot += c;
    }
    total = tot;
    return off + incl - x;
}

__device__ __forceinline__ uint64_t f2key(double x) {
    x = x + 0.0; // -0.0 -> +0.0 so that equal doubles get equal keys
    uint64_t u = (uint64_t)__double_as_longlong(x);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

__device__ __forceinline__ int wave_max(int x) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) x = max(x, __shfl_xor(x, d, 64));
    return x;
}

__device__ __forceinline__ void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------------------------------
// Masked flooding min-sum (osd_window.pyx:381-485; unmasked special case =
// bp_guessing_decoder.pyx:48-139).  Per iteration:
//   CN pass  lane per check: reads the b2c of its live edges, overwrites them in place with
//            c2b = (min over the other edges) * (+-alpha)          (osd_window.pyx:393-439)
//   VN pass  thread per live VN: posterior / hard decision / new b2c = prefix + suffix in
//            row order (osd_window.pyx:442-471); hard decisions are folded into the
//            per-check parity words with LDS atomics, which is the convergence test
//            H*e == s of osd_window.pyx:474-483 restricted to live checks.
// The parity words of iteration i are inspected at the start of the CN pass of iteration
// i+1, so an iteration costs two barriers.
// ------------------------------------------------------------------------------------------
// Per-thread register cache of the variable nodes a thread owns during one BP phase: thread t owns
// entries t, t+NT, ... of the VN list (all VNs in the pre phase, the compacted live list in the post
// phase).  Edge metadata and priors are loaded once per phase, so the iteration loop touches only LDS.
// DM = exact bound on the column degree of this kernel variant.
//
// The inner loops are VALU-issue bound (scripts/ubench/valu_indep.hip: ~4.5 cycles per wave
// instruction and SIMD, whatever the occupancy), so they are written to need no per-edge predicate:
//   * positions of a VN without a live edge point at the wave's "zero" slot Z_w, which holds +0.0
//     when it is read (x + 0.0 == x for every x a prefix / suffix sum can take here: they start from
//     the prior resp. +0.0 and so are never -0.0) and is re-armed after the VN's writes;
//   * positions of a check without a live edge point at the wave's "far" slot D_w, which holds a
//     positive value >= 50 when it is read: it clips to 50, never beats a real magnitude (all
//     <= 50) to first or second minimum of a check with >= 2 live edges, and never counts as
//     negative; checks with one live edge get their "minimum of nothing" (1e308) patched in after
//     the loop.  D_w / Z_w are only touched by their own wave, whose LDS operations execute in order.
// ed = byte offset of the edge's message slot (a ready LDS address: nothing to recompute or to keep a
// second copy of per iteration); par = index into par[] of the edge's check, two per word, m for
// dead positions.
// P16 (kernels whose LDS offsets fit 16 bits: the osd_window kernels of up to 256 threads): two offsets per register,
// unpacked where they are used (sub-dword operand selects, no instruction) behind an asm statement that keeps the
// unpacking inside the iteration loop -- hoisted out of it, the 32-bit copies would take the registers back.  With the
// packed caches the <256, 7, 6, 9> kernel needs no spill at 248 VGPRs and one reload per BP loop at the 168 of three
// waves per SIMD.
// Two offsets per register, unpacked where they are used.  SH = 0: byte offsets (kernels of up to 256 threads, whose LDS offsets fit
// 16 bits, PB: one parity byte per check); SH = 3: slot numbers, shifted where they are used (the 1024-thread osd_window
// kernels: fewer registers in their 128-VGPR budget, classic parity words).
template <int VF, int DM, int SH = 0, bool PB = true>
struct VnCacheP {
    static_assert(DM % 2 == 0, "edge offsets are packed in pairs");
    static constexpr bool par_bytes = PB;
    // the halves of par[][] hold the check's lane number, or -- byte-offset caches with parity words (SH == 0, !PB: round 5) -- the
    // byte offset of its parity word (lane << 2): the flip's address is then one add, like a message address
    static constexpr int par_shift = (SH == 0 && !PB) ? 2 : 0;
    double llr[VF];
    uint32_t edp[VF][DM / 2];
    __device__ __forceinline__ void set_ed(int i, int k, uint32_t v) { v >>= SH; edp[i][k >> 1] = (k & 1) ? ((edp[i][k >> 1] & 0xFFFFu) | (v << 16)) : ((edp[i][k >> 1] & 0xFFFF0000u) | v); }
    __device__ __forceinline__ void get_ed(int i, uint32_t (&ad)[DM]) const {
#pragma unroll
        for (int k = 0; k < DM; k += 2) {
            uint32_t w = edp[i][k >> 1];
            asm volatile("" : "+v"(w));
            ad[k] = (w & 0xFFFFu) << SH; ad[k + 1] = (w >> 16) << SH;
        }
    }
    uint32_t par[VF][(DM + 1) / 2];
};

__device__ __forceinline__ int swd_slot_far(const SwdGraphDev &g) { return g.E + 1 + (int)(threadIdx.x >> 6); }
template <int NT>
__device__ __forceinline__ int swd_slot_zero(const SwdGraphDev &g) { return g.E + 1 + NT / 64 + (int)(threadIdx.x >> 6); }

// remap != nullptr (shortened graph, BIG kernels with the post-phase messages in LDS): the messages are renumbered one column of
// cells per live variable node -- cell(k, i) = k * nlive + i for edge position k of the i-th live node, so that the threads of a
// wave touch consecutive cells in the variable-node pass -- and remap[old slot] = cell for the check side (g.E = D * nlive here).
// All global loads are issued unconditionally and before anything depends on them (the edge table is padded with SWD_PAD_EDGE
// beyond a column's degree, rows beyond g.D do not exist: clamped): one memory latency per call instead of two per variable node.
// Full graph, in two halves: vn_cache_issue starts every load (raw edge words + priors), vn_cache_pack builds the cache from them --
// whatever the caller does in between runs while the loads are in flight.
template <int NT, int VF, int DM>
struct VnRaw { uint32_t ev[VF][DM]; double llr[VF]; };
template <int NT, int VF, int DM, bool SORTED = false> // SORTED: entry idx = the idx-th node of the graph's listed order (vperm)
__device__ __forceinline__ void vn_cache_issue(const SwdGraphDev &g, const Lds &s, VnRaw<NT, VF, DM> &r) {
    const int n = g.n, D = g.D;
    const uint32_t *et = SORTED ? g.vn_edge_s : g.vn_edge;
    const double *lt = SORTED ? g.llr_s : g.llr;
#pragma unroll
    for (int i = 0; i < VF; ++i) {
        const int idx = s.vtid + i * NT;
        const int v = (idx < n) ? idx : 0;
        r.llr[i] = (n > 0) ? lt[v] : 0.0;
#pragma unroll
        for (int k = 0; k < DM; ++k) r.ev[i][k] = (n > 0) ? et[max(min(k, D - 1), 0) * n + v] : SWD_PAD_EDGE;
    }
}
// largest degree among the nodes this wave serves in cache row i (wave-uniform), from the raw edge words of the row
template <int NT, int VF, int DM>
__device__ __forceinline__ void vn_row_caps(const SwdGraphDev &g, const Lds &s, const uint32_t (&ev)[VF][DM], int (&kc)[VF]) {
#pragma unroll
    for (int i = 0; i < VF; ++i) {
        const bool valid = s.vtid + i * NT < g.n;
        int d = 0;
#pragma unroll
        for (int k = 0; k < DM; ++k) d += (valid && k < g.D && ev[i][k] != SWD_PAD_EDGE) ? 1 : 0;
        kc[i] = __builtin_amdgcn_readfirstlane(wave_max(d));
    }
}
template <int NT, int VF, int DM, int SH, bool PB>
__device__ __forceinline__ void vn_cache_pack(const SwdGraphDev &g, const Lds &s, const VnRaw<NT, VF, DM> &r, VnCacheP<VF, DM, SH, PB> &c) {
    const int n = g.n, D = g.D;
    const uint32_t dead = (uint32_t)swd_slot_zero<NT>(g) << 3;
#pragma unroll
    for (int i = 0; i < VF; ++i) {
        const int idx = s.vtid + i * NT;
        const bool valid = idx < n;
        c.llr[i] = valid ? r.llr[i] : 0.0;
#pragma unroll
        for (int k = 0; k < DM; ++k) c.set_ed(i, k, dead);
        constexpr int PS = VnCacheP<VF, DM, SH, PB>::par_shift;
#pragma unroll
        for (int k = 0; k < (DM + 1) / 2; ++k) c.par[i][k] = ((uint32_t)g.m << PS) * 0x10001u;
#pragma unroll
        for (int k = 0; k < DM; ++k) {
            const uint32_t e = r.ev[i][k];
            if (valid && k < D && e != SWD_PAD_EDGE) {
                c.set_ed(i, k, swd_edge_slot(e) << 3);
                c.par[i][k >> 1] = (k & 1) ? ((c.par[i][k >> 1] & 0xFFFFu) | ((swd_edge_lane(e) << PS) << 16))
                                           : ((c.par[i][k >> 1] & 0xFFFF0000u) | (swd_edge_lane(e) << PS));
            }
        }
    }
}

// ALLEDGES (with !FULL): the listed nodes with every edge of theirs, whatever the state of the checks.
// SORTED (with FULL): entry idx = the idx-th node of the graph's listed order (vperm); kc (optional): the rows' degree caps (vn_row_caps)
// JPACK: the halves of par[][] carry the edge's position inside its check's row in bits 10..15 (lane numbers need ten: bp_run<..., REC>)
template <int NT, int VF, int DM, bool FULL, bool ALLEDGES = false, bool SORTED = false, bool JPACK = false, int SH, bool PB>
__device__ __forceinline__ void vn_cache_load(const SwdGraphDev &g, Lds &s, int nlive, VnCacheP<VF, DM, SH, PB> &c, uint16_t *remap = nullptr, int (*kc)[VF] = nullptr) {
    static_assert(!SORTED || FULL, "the listed order belongs to the full graph");
    const int n = g.n, cnt = FULL ? n : nlive;
    const uint32_t dead = (uint32_t)swd_slot_zero<NT>(g) << 3;
    const int D = g.D;
    const uint32_t *et = SORTED ? g.vn_edge_s : g.vn_edge;
    const double *lt = SORTED ? g.llr_s : g.llr;
    uint32_t ev[VF][DM];
#pragma unroll
    for (int i = 0; i < VF; ++i) {
        const int idx = s.vtid + i * NT;
        const int v = (idx < cnt) ? (FULL ? idx : (int)s.lv[idx]) : 0;
        c.llr[i] = (n > 0) ? lt[v] : 0.0; // (n, D: uniform)
#pragma unroll
        for (int k = 0; k < DM; ++k) ev[i][k] = (n > 0) ? et[max(min(k, D - 1), 0) * n + v] : SWD_PAD_EDGE;
    }
    if constexpr (SORTED) { if (kc) vn_row_caps<NT, VF, DM>(g, s, ev, *kc); }
#pragma unroll
    for (int i = 0; i < VF; ++i) {
        const int idx = s.vtid + i * NT;
        const bool valid = idx < cnt;
        if (!valid) c.llr[i] = 0.0;
#pragma unroll
        for (int k = 0; k < DM; ++k) c.set_ed(i, k, dead);
        constexpr int PS = VnCacheP<VF, DM, SH, PB>::par_shift;
#pragma unroll
        for (int k = 0; k < (DM + 1) / 2; ++k) c.par[i][k] = ((uint32_t)g.m << PS) * 0x10001u;
#pragma unroll
        for (int k = 0; k < DM; ++k) {
            const uint32_t e = ev[i][k];
            if (valid && k < D && e != SWD_PAD_EDGE) {
                if (FULL || ALLEDGES || s.cn_val[swd_edge_lane(e)] >= 0) {
                    uint32_t slot = swd_edge_slot(e);
                    if (!FULL && remap) { const uint32_t cell = (uint32_t)(k * nlive + idx); remap[slot] = (uint16_t)cell; slot = cell; }
                    c.set_ed(i, k, slot << 3);
                    const uint32_t ph = (swd_edge_lane(e) << PS) | (JPACK ? (swd_edge_j(e) << 10) : 0u);
                    c.par[i][k >> 1] = (k & 1) ? ((c.par[i][k >> 1] & 0xFFFFu) | (ph << 16))
                                               : ((c.par[i][k >> 1] & 0xFFFF0000u) | ph);
                }
            }
        }
    }
}


// Shortened graph, sorted form (SWD_POST_SORTED): the listed nodes' LIVE edges at the front of the cache in their row order (the
// prefix / suffix sums of a node run over its live edges in that order either way; a dead position contributed + 0.0), cells
// renumbered cell(kk, i) = kk x nlive + i for the kk-th live edge of the i-th listed node, remap[old slot] = cell for the check side.
// kc[i]: the largest number of live edges among the nodes this wave serves in cache row i (wave-uniform).
template <int NT, int VF, int DM, int SH, bool PB>
__device__ __forceinline__ void vn_cache_load_compact(const SwdGraphDev &g, Lds &s, int nlive, VnCacheP<VF, DM, SH, PB> &c, uint16_t *remap, int (&kc)[VF]) {
    const int n = g.n, D = g.D;
    constexpr int PS = VnCacheP<VF, DM, SH, PB>::par_shift;
    const uint32_t dead = (uint32_t)swd_slot_zero<NT>(g) << 3;
    uint32_t ev[VF][DM];
#pragma unroll
    for (int i = 0; i < VF; ++i) {
        const int idx = s.vtid + i * NT;
        const int v = (idx < nlive) ? (int)s.lv[idx] : 0;
        c.llr[i] = (n > 0) ? g.llr[v] : 0.0;
#pragma unroll
        for (int k = 0; k < DM; ++k) ev[i][k] = (n > 0) ? g.vn_edge[max(min(k, D - 1), 0) * n + v] : SWD_PAD_EDGE;
    }
#pragma unroll
    for (int i = 0; i < VF; ++i) {
        const int idx = s.vtid + i * NT;
        const bool valid = idx < nlive;
        if (!valid) c.llr[i] = 0.0;
        uint32_t edl[DM], pl[DM]; // cell byte offsets / parity halves of the live edges, front-compacted
#pragma unroll
        for (int q = 0; q < DM; ++q) { edl[q] = dead; pl[q] = (uint32_t)g.m << PS; }
        int kk = 0;
#pragma unroll
        for (int k = 0; k < DM; ++k) {
            const uint32_t e = ev[i][k];
            const bool live = valid && k < D && e != SWD_PAD_EDGE && s.cn_val[swd_edge_lane(e)] >= 0;
            if (live) {
                const uint32_t cell = (uint32_t)(kk * nlive + idx);
                remap[swd_edge_slot(e)] = (uint16_t)cell;
#pragma unroll
                for (int q = 0; q <= k; ++q) // (kk <= k)
                    if (q == kk) { edl[q] = cell << 3; pl[q] = swd_edge_lane(e) << PS; }
                ++kk;
            }
        }
#pragma unroll
        for (int q = 0; q < DM; ++q) c.set_ed(i, q, edl[q]);
#pragma unroll
        for (int q = 0; q < (DM + 1) / 2; ++q) c.par[i][q] = pl[2 * q] | ((2 * q + 1 < DM ? pl[2 * q + 1] : ((uint32_t)g.m << PS)) << 16);
        kc[i] = __builtin_amdgcn_readfirstlane(wave_max(kk));
    }
}

__device__ __forceinline__ double &swd_msg_at(Lds &s, uint32_t ed) { return *(double *)((char *)s.msg + ed); }
typedef uint32_t swd_u32x4r __attribute__((ext_vector_type(4)));
// Hybrid store, explicit form: a FLAT access is worked through the texture addresser lane by lane whichever memory it ends in -- and the
// scattered 8-byte accesses of the variable-node pass are bound by exactly that unit -- so the two memories get an instruction each
// (ds_read / ds_write for the lanes in LDS, global_load / global_store for the rest, disjoint exec masks).
#ifndef SWD_BIG_HYBRID_SPLIT
#define SWD_BIG_HYBRID_SPLIT 1
#endif
#define SWD_AS3 __attribute__((address_space(3)))
#define SWD_AS1 __attribute__((address_space(1)))
template <bool HYB>
__device__ __forceinline__ double swd_msg_ld(Lds &s, uint32_t ed) {
    if constexpr (HYB && SWD_BIG_HYBRID_SPLIT) {
        double v;
        if (ed >= s.msg_lo) v = *(const SWD_AS3 double *)((SWD_AS3 char *)s.msg_alt + (ed - s.msg_lo));
        else v = *(const SWD_AS1 double *)((SWD_AS1 char *)s.msg + ed);
        return v;
    } else if constexpr (HYB) {
        const bool up = ed >= s.msg_lo;
        return *(const double *)((up ? s.msg_alt : (char *)s.msg) + (up ? ed - s.msg_lo : ed));
    } else return *(const double *)((char *)s.msg + ed);
}
template <bool HYB>
__device__ __forceinline__ void swd_msg_st(Lds &s, uint32_t ed, double v) {
    if constexpr (HYB && SWD_BIG_HYBRID_SPLIT) {
        if (ed >= s.msg_lo) *(SWD_AS3 double *)((SWD_AS3 char *)s.msg_alt + (ed - s.msg_lo)) = v;
        else *(SWD_AS1 double *)((SWD_AS1 char *)s.msg + ed) = v;
    } else if constexpr (HYB) {
        const bool up = ed >= s.msg_lo;
        *(double *)((up ? s.msg_alt : (char *)s.msg) + (up ? ed - s.msg_lo : ed)) = v;
    } else *(double *)((char *)s.msg + ed) = v;
}
template <bool HYB>
__device__ __forceinline__ double &swd_msg_h(Lds &s, uint32_t ed) {
    if constexpr (HYB) { // (no pre-subtracted base: arithmetic that leaves the LDS block is folded into its 32-bit offset and wraps)
        const bool up = ed >= s.msg_lo;
        return *(double *)((up ? s.msg_alt : (char *)s.msg) + (up ? ed - s.msg_lo : ed));
    }
    else return *(double *)((char *)s.msg + ed);
}

// bp_init for the table form of the full graph (bp_run<..., TBL>)
template <int NT, int VF, int DM, bool HYB>
__device__ __forceinline__ void bp_init_tbl(const SwdGraphDev &g, Lds &s) {
    const int n = g.n;
#pragma unroll 1
    for (int i = 0; i < VF; ++i) {
        const int idx = s.vtid + i * NT;
        if (idx < n) {
            const double l = g.llr[idx];
            for (int k = 0; k < g.D; ++k) {
                const uint32_t e = g.vn_edge[k * n + idx];
                if (e != SWD_PAD_EDGE) swd_msg_st<HYB>(s, swd_edge_slot(e) << 3, l);
            }
        }
    }
}

// bp_init (osd_window.pyx:370-379): b2c <- prior on every live edge of every live VN
template <int VF, int DM, bool HYB = false, int SH, bool PB>
__device__ __forceinline__ void bp_init(Lds &s, const VnCacheP<VF, DM, SH, PB> &c) {
#pragma unroll
    for (int i = 0; i < VF; ++i) {
        uint32_t ad[DM];
        c.get_ed(i, ad);
#pragma unroll
        for (int k = 0; k < DM; ++k) swd_msg_st<HYB>(s, ad[k], c.llr[i]); // dead positions land in Z_w (re-armed by bp_run)
    }
}


// Per-thread register cache of the check a lane owns during one BP phase: the LDS slots of its
// edges (u16, two per register) in walk order; unused / dead positions hold the wave's far slot.
// KG = groups of four positions.
// (offsets packed two per register, SH as in VnCacheP)
template <int KG, int SH = 0>
struct CnCacheP {
    uint32_t slp[KG * 2];
    __device__ __forceinline__ void set_slot(int k, int slot) { const uint32_t v = ((uint32_t)slot << 3) >> SH; slp[k >> 1] = (k & 1) ? ((slp[k >> 1] & 0xFFFFu) | (v << 16)) : ((slp[k >> 1] & 0xFFFF0000u) | v); }
    __device__ __forceinline__ void group(int gq, uint32_t (&ad)[4]) const { // byte offsets of positions 4 gq .. 4 gq + 3
        uint32_t w0 = slp[2 * gq], w1 = slp[2 * gq + 1];
        asm volatile("" : "+v"(w0), "+v"(w1));
        ad[0] = (w0 & 0xFFFFu) << SH; ad[1] = (w0 >> 16) << SH; ad[2] = (w1 & 0xFFFFu) << SH; ad[3] = (w1 >> 16) << SH;
    }
    int cnt, live, l, sub, grp;
};

// Live-position masks of the checks: 64 bits per check, or -- tuned osd_window kernels (D) on graphs with row weight <= 48
// (s.lm_m != 0) -- 32 + 16 bits in two arrays.
template <bool D = false> __device__ __forceinline__ uint64_t lm_get(const Lds &s, int l) {
    if constexpr (D) {
        if (s.lm_m) {
            const uint32_t *lo = (const uint32_t *)s.livemask; const uint16_t *hi = (const uint16_t *)(lo + s.lm_m);
            return (uint64_t)lo[l] | ((uint64_t)hi[l] << 32);
        }
    }
    return s.livemask[l];
}
template <bool D = false> __device__ __forceinline__ void lm_set(Lds &s, int l, uint64_t v) {
    if constexpr (D) {
        if (s.lm_m) {
            uint32_t *lo = (uint32_t *)s.livemask; uint16_t *hi = (uint16_t *)(lo + s.lm_m);
            lo[l] = (uint32_t)v; hi[l] = (uint16_t)(v >> 32);
            return;
        }
    }
    s.livemask[l] = v;
}

// grp = 1, 2 or 4 threads share a check (adjacent lanes of a quad): thread `sub` walks positions sub, sub + grp, ...
// (tuned kernels: 48-bit live masks, original degrees from the graph)
template <int NT, int KG, bool FULL, int SH>
__device__ __forceinline__ void cn_cache_load(const SwdGraphDev &g, Lds &s, bool uselist, int lc, int sub, int grp, CnCacheP<KG, SH> &cc) {
    constexpr bool DIET = SWD_P16(NT);
    const int m = g.m, dummy = swd_slot_far(g);
    const bool act = (lc >= 0) && (lc < m) && (s.cn_val[lc >= 0 ? lc : 0] >= 0);
    const int l = act ? lc : 0;
    cc.l = act ? lc : -1;
    cc.sub = sub;
    cc.grp = grp;
    // list mode walks the compacted live edges, otherwise all original positions (dead ones skipped)
    const bool bylist = !FULL && uselist;
    // (tuned kernels keep no copy of the original degrees in LDS)
    const int cnt = act ? ((FULL || bylist) ? (int)s.cn_deg[l] : (DIET ? (int)g.row_deg[l] : (int)s.cn_deg0[l])) : 0;
    const uint64_t lmask = (FULL || bylist || !act) ? ~0ull : lm_get(s, l);
    cc.cnt = (cnt > sub) ? (cnt - sub + grp - 1) / grp : 0;
    cc.live = act ? (int)s.cn_deg[l] : 0;
#pragma unroll
    for (int kk = 0; kk < KG * 4; ++kk) {
        const int k = kk * grp + sub;
        int sv = dummy;
        if (k < cnt && ((lmask >> (k & 63)) & 1ull))
            sv = bylist ? (int)s.lslot[k * m + l] : (int)s.jptr[k] + l;
        cc.set_slot(kk, sv);
    }
}

// value of the lane `lane ^ X` (X = 1 or 2) of the same quad
template <int X>
__device__ __forceinline__ int quad_xor(int v) {
    return __builtin_amdgcn_mov_dpp(v, X == 1 ? 0xB1 : 0x4E, 0xF, 0xF, true);
}
template <int X>
__device__ __forceinline__ double quad_xor(double v) {
    const long long b = __double_as_longlong(v);
    const uint32_t lo = (uint32_t)quad_xor<X>((int)(uint32_t)b), hi = (uint32_t)quad_xor<X>((int)(uint32_t)(b >> 32));
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}

// fp64 min / max without the canonicalising v_max x,x the compiler puts in front of fmin/fmax
__device__ __forceinline__ double vmin64(double a, double b) { double d; asm("v_min_f64 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ double vmax64(double a, double b) { double d; asm("v_max_f64 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ double vminabs64(double x, double c) { double d; asm("v_min_f64 %0, |%1|, %2" : "=v"(d) : "v"(x), "v"(c)); return d; }
// w = 2*w + (x <= 0): shift register of "is negative" bits (osd_window.pyx:404-409 counts <= 0 as negative)
__device__ __forceinline__ void neg_shift_in(uint32_t &w, double x) {
    asm("v_cmp_ge_f64 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(w) : "v"(x) : "vcc");
}

// Deals the live checks to the threads for a shortened-graph BP phase: by decreasing live degree (counting
// sort; a wave's CN pass costs its largest degree and after decimation the degrees are very uneven), and
// checks heavier than T are shared by 2 or 4 adjacent threads so that no thread walks more than T positions
// (T = smallest value for which one thread per check of degree <= T, two up to 2T and four up to 4T fit the
// workgroup; sorted by decreasing degree the quads come first, then the pairs, so quads / pairs stay aligned).
// dhist: 66 ints, cord: m u16 of LDS.  counted: dhist already holds the degree histogram of the live checks.
// (The guessing decoders rebuild their caches every few iterations; there the sort costs more than it saves.)
// Out: the check this thread serves (-1 for none), its rank among the check's threads and their number.
template <int NT, int KG>
__device__ __forceinline__ void cn_assign(const SwdGraphDev &g, Lds &s, int *dhist, uint16_t *cord, bool split, bool counted,
                                          int nlive_cn, int &lc, int &sub, int &grp) {
    const int tid = threadIdx.x, m = g.m;
    if (!counted) {
        for (int i = tid; i < 65; i += NT) dhist[i] = 0;
        __syncthreads();
        for (int l = tid; l < m; l += NT)
            if (s.cn_val[l] >= 0) atomicAdd(&dhist[max(1, min((int)s.cn_deg[l], 64))], 1); // edgeless live checks still own a parity word
        __syncthreads();
    }
    if (tid < 64) { // bin d = 64 - lane: exclusive prefix in order of decreasing degree
        const int d = 64 - tid;
        const int c0 = dhist[d];
        int T = 64, nq = 0, np = 0, nl = c0;
#pragma unroll
        for (int dd = 32; dd > 0; dd >>= 1) nl += __shfl_xor(nl, dd, 64);
        if (split) {
            const int cand[8] = {3, 4, 6, 8, 12, 16, 24, 32};
#pragma unroll 1
            for (int ci = 0; ci < 8; ++ci) {
                const int Tc = cand[ci];
#ifdef SWD_CN_MIN_T // experiment: no walk bound below this (fewer, fuller waves in the check pass of the shortened graph)
                if (Tc < SWD_CN_MIN_T) continue;
#endif
                int need = c0 * (d <= Tc ? 1 : (d <= 2 * Tc ? 2 : 4));
                int q = (d > 2 * Tc) ? c0 : 0, pr = (d > Tc && d <= 2 * Tc) ? c0 : 0;
#pragma unroll
                for (int dd = 32; dd > 0; dd >>= 1) {
                    need += __shfl_xor(need, dd, 64); q += __shfl_xor(q, dd, 64); pr += __shfl_xor(pr, dd, 64);
                }
                const bool toobig = __ballot(c0 > 0 && d > 4 * Tc) != 0ull;
                if (!toobig && need <= NT && Tc <= KG * 4) { T = Tc; nq = q; np = pr; break; }
            }
        }
        if (tid == 0) { s.iaux[0] = T; s.iaux[1] = nq; s.iaux[2] = np; s.iaux[3] = nl; }
        int incl = c0;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
            const int y = __shfl_up(incl, dd, 64);
            if (tid >= dd) incl += y;
        }
        dhist[d] = incl - c0;
    }
    __syncthreads();
    for (int l = tid; l < m; l += NT)
        if (s.cn_val[l] >= 0) cord[atomicAdd(&dhist[max(1, min((int)s.cn_deg[l], 64))], 1)] = (uint16_t)l;
    __syncthreads();
    const int nq = s.iaux[1], np = s.iaux[2], ct = s.ctid;
    const int nl = (nlive_cn >= 0) ? nlive_cn : s.iaux[3];
    int cidx;
    if (ct < 4 * nq) { cidx = ct >> 2; sub = ct & 3; grp = 4; }
    else if (ct < 4 * nq + 2 * np) { const int t2 = ct - 4 * nq; cidx = nq + (t2 >> 1); sub = t2 & 1; grp = 2; }
    else { cidx = nq + np + (ct - 4 * nq - 2 * np); sub = 0; grp = 1; }
    lc = (cidx < nl) ? (int)cord[cidx] : -1;
}

// ACC: the posterior history is only ever consumed as ((h0 + h1) + h2) + h3 in slot order after a run of ALL
// max_iter iterations, and with max_iter a multiple of four slot order is chronological order of the last four
// iterations: the sum is accumulated in registers (hs[i] for the i-th variable node of the thread) in exactly
// that order and the 4 x n ring in HBM is neither written nor read.
// One routine for every kernel: the register caches keep LDS offsets packed two per register (unpacked where they are used);
// PB: one parity byte per check, flipped by word atomics (tuned kernels of up to 256 threads), else one parity word.
// SPARSE (with !FULL): the node list may hold n = "no node here" (the guessing decoders keep a position's thread for the whole
// tree walk instead of compacting the live nodes after every decimation), and the posterior history goes to registers of the
// node's thread -- h4[i][slot] for its i-th node -- instead of the ring in HBM: the thread that runs a position's node is the one
// that classifies the position afterwards, and without the stores the iteration barriers no longer wait for HBM.
// TIER (sorted form of the shortened graph, vn_cache_load_compact): kcap[i] = largest number of live edges among the nodes the wave
// serves in cache row i; the variable-node pass runs over the first 2, 4, ... DM positions only (a wave-uniform choice per row).
// TBL (large-graph kernels, full graph): no register cache of the variable nodes -- a row's edge words and prior come from the graph's
// tables (coalesced, L2-resident, shared by every workgroup), asked for one row ahead; nine rows of cache do not fit that kernel's 128
// registers beside the check cache and came back from scratch, row by row, inside the iterations.
template <int NT, int VF, int DM, int KG, bool FULL, bool SF = false, bool ACC = false, bool SPARSE = false, bool TIER = false, bool HYB = false, bool REC = false, bool TBL = false, int SH, bool PB>
__device__ __forceinline__ int bp_run(const SwdGraphDev &g, const SwdDecodeParams &P, Lds &s, int max_iter, int nlive,
                      const VnCacheP<VF, DM, SH, PB> &c, const CnCacheP<KG, SH> &cn, double *hist_b, int &iters_done,
                      double alpha, bool force_unsat = false, double *hs = nullptr, double (*h4)[4] = nullptr, bool rec_early = false,
                      const int *kcap = nullptr) {
    const int tid = threadIdx.x, m = g.m, n = g.n;
    const int vcnt = FULL ? n : nlive;
    // rec_early: the caller reads the history even when the run converges before its last four iterations (the threaded
    // ensemble's main thread scans BEFORE its convergence test, bpgd.cpp:630-633)
    const bool record_all = P.record_all != 0 || rec_early;
    const int l = cn.l >= 0 ? cn.l : 0;      // NT >= m: at most one check per thread
    const int cv = (cn.l >= 0) ? (int)s.cn_val[l] : -1;
    const int cnt = cn.cnt;
#ifdef SWD_NO_SCALAR_WMAX
    const int wmax = wave_max(cnt);
#else
    const int wmax = __builtin_amdgcn_readfirstlane(wave_max(cnt)); // (scalar: the groups of four positions are skipped by scalar branches)
#endif
    // largest number of threads sharing a check in this wave (uniform): waves whose checks all have a thread of their own skip the merge
    [[maybe_unused]] const int gmax = __builtin_amdgcn_readfirstlane(wave_max(cn.grp));
    const int farslot = swd_slot_far(g), zeroslot = swd_slot_zero<NT>(g);
    constexpr int K4 = KG * 4;
    constexpr int NR = (K4 + 31) / 32;       // sign shift registers
    iters_done = 0;
    if (max_iter <= 0) return 0;
    // VNs this wave walks (wave-uniform): entries vtid + i*NT < vcnt for some lane
    const int wbase = s.vtid & ~63;
    const int nch = __builtin_amdgcn_readfirstlane((vcnt > wbase) ? min(VF, (vcnt - wbase + NT - 1) / NT) : 0);
#ifdef SWD_BPPROF
    if (!FULL && (tid & 63) == 0) ((uint8_t *)&s.scal[28])[tid >> 6] = (uint8_t)wmax;
#endif
    swd_msg_st<HYB>(s, (uint32_t)farslot << 3, 64.0);
    swd_msg_st<HYB>(s, (uint32_t)zeroslot << 3, 0.0);
    char *const parb = (char *)s.par;
    // (shortened graph) the node a list entry names does not change during the run: read once, not once per iteration
    // (full graph in tiers: the listed order of the graph, SwdGraphDev::vperm)
    constexpr bool kNodeList = !FULL || TIER;
    [[maybe_unused]] int vnode[kNodeList ? VF : 1];
    if constexpr (kNodeList) {
#pragma unroll
        for (int i = 0; i < VF; ++i) { const int idx = s.vtid + i * NT; vnode[i] = (idx < vcnt) ? (FULL ? (int)g.vperm[idx] : (int)s.lv[idx]) : n; }
    }
#ifdef SWD_BPPROF
    long long tc0, tc1, tc2, tc3;
    long long acc_cn = 0, acc_any = 0, acc_vn = 0, acc_bar = 0;
#define BPT(x) x = clock64()
#else
#define BPT(x)
#endif
    static_assert(!TBL || (FULL && !TIER && !REC && !SPARSE && !PB), "table form: the full graph of the large-graph kernels");
    [[maybe_unused]] uint32_t tbe[3][DM]; // edge words of rows i, i + 1, i + 2 (asked for two rows ahead)
    [[maybe_unused]] double tbl[3];
    [[maybe_unused]] auto tb_addr = [&](int row, const uint32_t (&e)[DM], uint32_t (&ad_)[DM]) {
        const bool ok = s.vtid + row * NT < n;
        const uint32_t dead_ = (uint32_t)zeroslot << 3;
#pragma unroll
        for (int k = 0; k < DM; ++k) ad_[k] = (ok && k < g.D && e[k] != SWD_PAD_EDGE) ? (swd_edge_slot(e[k]) << 3) : dead_;
    };
    [[maybe_unused]] auto tb_load = [&](int row, uint32_t (&e)[DM], double &lv_) {
        const int idx = s.vtid + row * NT;
        const int v = (idx < n) ? idx : 0;
        lv_ = g.llr[v];
#pragma unroll
        for (int k = 0; k < DM; ++k) e[k] = g.vn_edge[max(min(k, g.D - 1), 0) * n + v];
    };
    // Round 6: the hard decisions of a variable-node pass stay in a register (one bit per cache row) and go to LDS once, when the run
    // ends -- nothing reads s.hard while the iterations run (convergence is tested through the parity words), and the byte store per row
    // and iteration was one LDS instruction in twenty-five of the loop (SWD_BP_HARD_DEFER=0: the store per row of rounds 1-5)
    constexpr bool kHardDefer = SWD_BP_HARD_DEFER != 0 && VF <= 32 && !TBL;
    constexpr bool kParInc = kHardDefer && SWD_BP_PAR_INC != 0;
    [[maybe_unused]] uint32_t hdbits = 0, hdprev = 0;
    [[maybe_unused]] auto hard_flush = [&]() {
        if constexpr (kHardDefer) {
#pragma unroll
            for (int i = 0; i < VF; ++i)
                if (i < nch) { // wave-uniform
                    const int idx = s.vtid + i * NT;
                    int v;
                    if constexpr (!kNodeList) v = (idx < vcnt) ? idx : n; else v = vnode[i];
                    ((bool *)s.hard)[v] = ((hdbits >> i) & 1u) != 0u;
                }
        }
    };
    for (int it = 0; it < max_iter; ++it) {
        bool unsat = force_unsat; // a check without any selected column but syndrome 1 can never be met
        BPT(tc0);
        {
            if (cv >= 0 && cn.sub == 0) {
                // kParInc (round 6): the parity word of a check is set once per run and then kept up to date by the nodes whose decision
                // CHANGED in a pass (the previous pass's decisions are the bits of hdbits) -- instead of being reset here and flipped by
                // every node that decides 1 in every pass: after a few iterations few decisions move, and a wave without one skips
                // the atomics of the row altogether.  Same value in front of every test (residual check value ^ parity of the decisions).
                if constexpr (PB) { // tuned kernels: one parity byte per check, flipped by word atomics (bit 8 (l & 3) of word l >> 2)
                    if (it > 0 && ((const uint8_t *)s.par)[l] != 0) unsat = true;
                    if (!kParInc || it == 0) ((uint8_t *)s.par)[l] = (uint8_t)cv;
                } else {
                    if (it > 0 && s.par[l] != 0u) unsat = true;
                    if (!kParInc || it == 0) s.par[l] = (uint32_t)cv;
                }
            }
            // CN pass (osd_window.pyx:393-439).  Slots come from registers, so the message reads of a
            // group of four are independent.  The two-minimum update
            //   min2 = min(min2, max(min1, a)); min1 = min(min1, a)
            // equals the reference's left/right running minima; |clip(x, -50, 50)| = min(|x|, 50).
            // (A hand-made software pipeline over the groups was faster with the default machine scheduler and
            // is slower with iterative-ilp, which overlaps the reads of the next group by itself.)
            double min1 = 1e308, min2 = 1e308;
            uint32_t argslot = (uint32_t)farslot << 3; // byte offset of the first position holding the minimum
            uint32_t argk = 0;                          // ... and its position number: its sign comes out of the sign registers below
                                                        // (round 4 re-read the message: one more dependent LDS round trip per iteration)
            uint32_t neg[NR];
#pragma unroll
            for (int r = 0; r < NR; ++r) neg[r] = 0;
            constexpr bool kHalf = SWD_CN_HALF != 0 && !FULL;
#pragma unroll
            for (int gq = 0; gq < KG; ++gq) {
                if (gq * 4 < wmax) { // wave-uniform
                    double xs[4];
                    uint32_t ad[4];
                    cn.group(gq, ad);
                    const bool hi = !kHalf || gq * 4 + 2 < wmax; // (wave-uniform) does any lane walk the group's second half?
#pragma unroll
                    for (int u = 0; u < 2; ++u) xs[u] = swd_msg_ld<HYB>(s, ad[u]);
                    if (hi) {
#pragma unroll
                        for (int u = 2; u < 4; ++u) xs[u] = swd_msg_ld<HYB>(s, ad[u]);
                    }
                    auto one = [&](auto u_tag) {
                        constexpr int u = decltype(u_tag)::value;
                        const int k = gq * 4 + u;
                        const double ax = vminabs64(xs[u], 50.0);
                        if constexpr (SWD_BP_XARG_TRACK || REC) argk = (ax < min1) ? (uint32_t)k : argk;
                        argslot = (ax < min1) ? ad[u] : argslot; // (as sign-of-difference mask + v_bfi instead of v_cmp + v_cndmask: 10.1 against 10.0 ms, round 4)
                        min2 = vmin64(min2, vmax64(min1, ax));
                        min1 = vmin64(min1, ax);
                        neg_shift_in(neg[k >> 5], xs[u]);
                    };
                    one(std::integral_constant<int, 0>{}); one(std::integral_constant<int, 1>{});
                    if (hi) { one(std::integral_constant<int, 2>{}); one(std::integral_constant<int, 3>{}); }
                    else {
#pragma unroll
                        for (int u = 2; u < 4; ++u) neg[(gq * 4 + u) >> 5] <<= 1;
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) neg[(gq * 4 + u) >> 5] <<= 1; // keep position k at bit (31 - k % 32) ...
                }
            }
            // ... after this alignment of a partly filled last register
            if (K4 & 31) neg[NR - 1] <<= (32 - (K4 & 31));
            int npar = (cn.sub == 0) ? cv : 0;
#pragma unroll
            for (int r = 0; r < NR; ++r) npar += __popc(neg[r]);
            // "is negative" bit of the first-minimum position (position k sits at bit 31 - k % 32 of neg[k / 32]; no live position:
            // position 0 = the far slot, positive)
            uint32_t argneg = 0;
            if constexpr (SWD_BP_XARG_TRACK) {
                uint32_t wsel = neg[0];
#pragma unroll
                for (int r = 1; r < NR; ++r) wsel = ((argk >> 5) == (uint32_t)r) ? neg[r] : wsel;
                argneg = (wsel << (argk & 31u)) >> 31;
            }
            if constexpr (!FULL || SF) {
                // merge the partial results of the check's threads (butterfly inside the quad).  On a tie
                // of the minima the second minimum equals the first, so which side's position is kept
                // as "first minimum" does not change any value written below.
                if (gmax >= 2) { // (scalar branch)
                    const double o1 = quad_xor<1>(min1), o2 = quad_xor<1>(min2);
                    const uint32_t oa = (uint32_t)quad_xor<1>((int)argslot); const int op = quad_xor<1>(npar);
                    const uint32_t on = (uint32_t)quad_xor<1>((int)argneg);
                    if (cn.grp >= 2) {
                        npar += op;
                        argneg = (o1 < min1) ? on : argneg;
                        argslot = (o1 < min1) ? oa : argslot;
                        min2 = vmin64(vmax64(min1, o1), vmin64(min2, o2));
                        min1 = vmin64(min1, o1);
                    }
                }
                if (gmax == 4) {
                    const double o1 = quad_xor<2>(min1), o2 = quad_xor<2>(min2);
                    const uint32_t oa = (uint32_t)quad_xor<2>((int)argslot); const int op = quad_xor<2>(npar);
                    const uint32_t on = (uint32_t)quad_xor<2>((int)argneg);
                    if (cn.grp == 4) {
                        npar += op;
                        argneg = (o1 < min1) ? on : argneg;
                        argslot = (o1 < min1) ? oa : argslot;
                        min2 = vmin64(vmax64(min1, o1), vmin64(min2, o2));
                        min1 = vmin64(min1, o1);
                    }
                }
            }
            const uint32_t flip = (npar & 1) ? 0xFFFFFFFFu : 0u;
            // the first position holding the minimum gets the second minimum (ties: both equal); its own sign: argneg
            if constexpr (!SWD_BP_XARG_TRACK && !REC) argneg = (swd_msg_ld<HYB>(s, argslot) <= 0) ? 1u : 0u; // (re-read before the slots are overwritten)
            if (cn.live == 1) min1 = min2 = 1e308; // minimum over no other edge (the far slot may have come first)
            const double p1 = min1 * alpha, p2 = min2 * alpha;
            const uint32_t p1lo = (uint32_t)__double_as_longlong(p1), p1hi = (uint32_t)(__double_as_longlong(p1) >> 32);
            if constexpr (REC) { // one record per check instead of a message per edge (position k's sign bit: bit 31 - k % 32 of word k / 32)
                static_assert(FULL && !SF && NR <= 2, "records: one thread per check, at most 64 positions");
                if (cn.l >= 0) {
                    swd_u32x4r q0, q1;
                    q0.x = p1lo; q0.y = p1hi; q0.z = (uint32_t)__double_as_longlong(p2); q0.w = (uint32_t)(__double_as_longlong(p2) >> 32);
                    q1.x = neg[0] ^ flip; q1.y = (NR > 1 ? neg[NR - 1] : 0u) ^ flip; q1.z = argk; q1.w = 0u;
                    *(SWD_AS3 swd_u32x4r *)(s.rec + l * 32) = q0;
                    *(SWD_AS3 swd_u32x4r *)(s.rec + l * 32 + 16) = q1;
                }
                swd_msg_st<HYB>(s, (uint32_t)farslot << 3, 64.0); // (nothing overwrote it; kept for symmetry with the message form)
            } else {
#pragma unroll
            for (int gq = 0; gq < KG; ++gq) {
                if (gq * 4 < wmax) {
                    uint32_t ad[4];
                    cn.group(gq, ad);
                    auto put = [&](auto u_tag) {
                        constexpr int u = decltype(u_tag)::value;
                        const int k = gq * 4 + u;
                        const uint32_t sb = ((neg[k >> 5] ^ flip) << (k & 31)) & 0x80000000u;
                        const uint32_t hi = sb | p1hi; // p1 >= +0: value * (+-alpha) is the magnitude with this sign
                        swd_msg_st<HYB>(s, ad[u], __longlong_as_double((long long)(((uint64_t)hi << 32) | p1lo)));
                    };
                    put(std::integral_constant<int, 0>{}); put(std::integral_constant<int, 1>{});
                    if (!kHalf || gq * 4 + 2 < wmax) { put(std::integral_constant<int, 2>{}); put(std::integral_constant<int, 3>{}); }
                }
            }
            {
                const uint32_t sb = ((0u - argneg) ^ flip) & 0x80000000u;
                const uint64_t b2 = (uint64_t)__double_as_longlong(p2) | ((uint64_t)sb << 32);
                swd_msg_st<HYB>(s, argslot, __longlong_as_double((long long)b2));
                swd_msg_st<HYB>(s, (uint32_t)farslot << 3, 64.0); // re-arm
            }
            } // (message form)
        }
        BPT(tc1);
        // block_any, cut in two: the per-wave flags go out before the barrier; behind it they are read TOGETHER with the first
        // node's messages -- one LDS round trip instead of two in front of the variable-node pass (a converged run has loaded
        // those messages for nothing)
        int anyr = 0;
        {
            constexpr int NW = NT / 64;
            const int fp_ = (s.fpar++) & 1;
            const unsigned long long bu = __ballot(unsat);
            if ((tid & 63) == 0) s.flags[fp_ * 16 + (tid >> 6)] = (bu != 0ull);
            if constexpr (TBL) { // (in flight across the barrier)
                if (nch > 0) tb_load(0, tbe[0], tbl[0]);
                if (nch > 1) tb_load(1, tbe[1], tbl[1]);
            }
            __syncthreads();
#pragma unroll
            for (int w = 0; w < NW; ++w) anyr |= s.flags[fp_ * 16 + w];
        }
        constexpr bool kFlagMerge = VF <= SWD_BP_FLAG_MERGE_MAXVF;
        [[maybe_unused]] double cc0[DM];
        [[maybe_unused]] uint32_t ad0[DM];
        if (kFlagMerge && nch > 0) { // (wave-uniform)
            c.get_ed(0, ad0);
#pragma unroll
            for (int k = 0; k < DM; ++k) cc0[k] = swd_msg_ld<HYB>(s, ad0[k]);
#pragma unroll
            for (int k = 0; k < DM; ++k) asm volatile("" : "+v"(cc0[k])); // (loaded here, in front of the exit test)
        }
#ifdef SWD_NO_SCALAR_ANY
        const bool any = anyr != 0;
#else
        const bool any = __builtin_amdgcn_readfirstlane(anyr) != 0; // the same on every lane: a scalar branch
#endif
        BPT(tc2);
#ifdef SWD_BPPROF
        acc_cn += tc1 - tc0; acc_any += tc2 - tc1;
        if (it > 0 && !any && !FULL && tid == 0) { s.scal[24] += (int)acc_cn; s.scal[25] += (int)acc_any; s.scal[26] += (int)acc_vn; s.scal[27] += (int)acc_bar; }
#endif
        if (it > 0 && !any) {
            if constexpr (kHardDefer) { hard_flush(); __syncthreads(); } // (callers read other threads' decisions)
            iters_done = it;
            return 1;
        }
        if constexpr (kHardDefer) { hdprev = hdbits; hdbits = 0; }

        const int slot_h = it & 3;
        const bool record = record_all || it >= max_iter - 4;
        // VN pass (osd_window.pyx:442-471).  (Reading the next VN's messages before this one's are written
        // was tried and is slower.)
        static_assert(!(TIER && kFlagMerge), "the tiered pass reads its own messages");
        // one node of the pass over its first KD positions (KD = DM unless TIER)
        auto vn_one = [&](auto kd_tag, auto i_tag) {
                constexpr int KD = decltype(kd_tag)::value, i = decltype(i_tag)::value;
                const int idx = s.vtid + i * NT;
                bool valid = idx < vcnt;
                int v;
                if constexpr (!kNodeList) v = valid ? idx : n; else v = vnode[i];
                if constexpr (SPARSE) valid = v < n;
                double cc[KD], pre[KD];
                uint32_t ad[DM];
                if (kFlagMerge && i == 0) {
#pragma unroll
                    for (int k = 0; k < KD; ++k) { ad[k] = ad0[k]; cc[k] = cc0[k]; }
                } else if constexpr (TBL) {
                    // (the messages of row i + 1 asked for here as well, one row ahead: 41.6 -> 51.0 us per iteration -- the pass is bound by the
                    //  texture addresser's rate for scattered accesses, not by their latency)
                    if (i + 2 < nch) tb_load(i + 2, tbe[(i + 2) % 3], tbl[(i + 2) % 3]);      // (uniform) edge words, two rows ahead
                    tb_addr(i, tbe[i % 3], ad);
#pragma unroll
                    for (int k = 0; k < KD; ++k) cc[k] = swd_msg_ld<HYB>(s, ad[k]);
                } else {
                    c.get_ed(i, ad);
                    if constexpr (REC) { // the check-to-bit message of edge k from the record of its check (lane in bits 0..9, position in bits 10..15)
#pragma unroll
                        for (int k = 0; k < KD; ++k) {
                            uint32_t pw = c.par[i][k >> 1];
                            asm volatile("" : "+v"(pw));
                            const uint32_t ph = (k & 1) ? (pw >> 16) : (pw & 0xFFFFu), ln_ = ph & 0x3FFu, j_ = ph >> 10;
                            const swd_u32x4r q0 = *(const SWD_AS3 swd_u32x4r *)(s.rec + ln_ * 32);
                            const swd_u32x4r q1 = *(const SWD_AS3 swd_u32x4r *)(s.rec + ln_ * 32 + 16);
                            const bool isarg = j_ == q1.z;
                            const uint32_t mlo = isarg ? q0.z : q0.x, mhi = isarg ? q0.w : q0.y;
                            const uint32_t sb = (((j_ < 32u) ? q1.x : q1.y) << (j_ & 31u)) & 0x80000000u;
                            cc[k] = __longlong_as_double((long long)(((uint64_t)(mhi | sb) << 32) | mlo));
                        }
                    } else {
#pragma unroll
                    for (int k = 0; k < KD; ++k) cc[k] = swd_msg_ld<HYB>(s, ad[k]);
                    }
                }
                double temp;
                if constexpr (TBL) temp = valid ? tbl[i % 3] : 0.0; else temp = c.llr[i];
#pragma unroll
                for (int k = 0; k < KD; ++k) { pre[k] = temp; temp = temp + cc[k]; }
                if constexpr (ACC) {
                    if (it >= max_iter - 4) hs[i] = (it == max_iter - 4) ? temp : hs[i] + temp; // wave-uniform conditions
                } else if constexpr (SPARSE) {
                    if (record) { // (uniform; the slot too)
                        if (slot_h == 0) h4[i][0] = temp; else if (slot_h == 1) h4[i][1] = temp; else if (slot_h == 2) h4[i][2] = temp; else h4[i][3] = temp;
                    }
                } else {
                    if (record && valid) hist_b[slot_h * n + v] = temp;
                }
                const bool hd = valid && (temp <= 0);
                if constexpr (kHardDefer) hdbits |= hd ? (1u << i) : 0u;
                else ((bool *)s.hard)[v] = hd; // a bool store is not a character-type access: it does not fence the double loads / stores around it
                double suf = 0.0;
#pragma unroll
                for (int k = KD - 1; k >= 0; --k) {
                    swd_msg_st<HYB>(s, ad[k], pre[k] + suf);
                    suf = suf + cc[k];
                }
                swd_msg_st<HYB>(s, (uint32_t)zeroslot << 3, 0.0); // re-arm
#ifdef SWD_EXP_EXTRA_LDS // experiment: N more LDS reads per node and iteration (is the LDS pipeline what the iterations queue for?)
                {
                    double dx_[SWD_EXP_EXTRA_LDS];
#pragma unroll
                    for (int e_ = 0; e_ < SWD_EXP_EXTRA_LDS; ++e_) dx_[e_] = swd_msg_ld<HYB>(s, (uint32_t)farslot << 3);
#pragma unroll
                    for (int e_ = 0; e_ < SWD_EXP_EXTRA_LDS; ++e_) asm volatile("" :: "v"(dx_[e_]));
                }
#endif
#ifdef SWD_EXP_EXTRA_VALU // experiment: N more vector instructions per node and iteration
                {
                    uint32_t vx_ = (uint32_t)v;
#pragma unroll
                    for (int e_ = 0; e_ < SWD_EXP_EXTRA_VALU; ++e_) asm volatile("v_add_u32 %0, %0, %0" : "+v"(vx_));
                }
#endif
                if constexpr (TBL) {
                    if (hd) {
#pragma unroll
                        for (int k = 0; k < KD; ++k) {
                            const uint32_t e = tbe[i % 3][k];
                            if (k < g.D && e != SWD_PAD_EDGE) atomicXor((uint32_t *)(parb + (swd_edge_lane(e) << 2)), 1u);
                        }
                    }
                } else
                if (kParInc ? (hd != (((hdprev >> i) & 1u) != 0u)) : hd) {
#pragma unroll
                    for (int k2 = 0; k2 < (KD + 1) / 2; ++k2) {
                        uint32_t pw = c.par[i][k2];
                        asm volatile("" : "+v"(pw));
                        if constexpr (PB) {
                            atomicXor((uint32_t *)(parb + (pw & 0xFFFCu)), 1u << ((pw & 3u) << 3));
                            if (2 * k2 + 1 < KD) atomicXor((uint32_t *)(parb + ((pw >> 16) & 0xFFFCu)), 1u << (((pw >> 16) & 3u) << 3));
                        } else {
                            constexpr int PS2 = 2 - VnCacheP<VF, DM, SH, PB>::par_shift; // (0: the halves are byte offsets already)
                            constexpr uint32_t LM = REC ? 0x3FFu : 0xFFFFu; // (REC: the halves carry the position above the lane number)
                            atomicXor((uint32_t *)(parb + ((pw & LM) << PS2)), 1u);
                            if (2 * k2 + 1 < KD) atomicXor((uint32_t *)(parb + (((pw >> 16) & LM) << PS2)), 1u);
                        }
                    }
                }
        };
        // TIER: the smallest multiple KD of the tier step >= the wave's largest live degree in this cache row (scalar branches)
        auto vn_tier = [&](auto self, auto kd_tag, auto i_tag, int kcv) -> void {
            constexpr int KD = decltype(kd_tag)::value;
            if constexpr (KD >= DM) vn_one(std::integral_constant<int, DM>{}, i_tag);
            else {
                if (kcv <= KD) vn_one(kd_tag, i_tag);
                else self(self, std::integral_constant<int, KD + SWD_TIER_STEP(DM)>{}, i_tag, kcv);
            }
        };
        auto vn_rows = [&](auto self, auto i_tag) -> void {
            constexpr int i = decltype(i_tag)::value;
            if constexpr (i < VF) {
                if (i < nch) { // wave-uniform
                    if constexpr (TIER) vn_tier(vn_tier, std::integral_constant<int, SWD_TIER_STEP(DM)>{}, i_tag, kcap[i]);
                    else vn_one(std::integral_constant<int, DM>{}, i_tag);
                }
                self(self, std::integral_constant<int, i + 1>{});
            }
        };
        vn_rows(vn_rows, std::integral_constant<int, 0>{});
        BPT(tc3);
        __syncthreads();
#ifdef SWD_BPPROF
        acc_vn += tc3 - tc2; acc_bar += clock64() - tc3;
#endif
    }
#ifdef SWD_BPPROF
    if (!FULL && tid == 0) { s.scal[24] += (int)acc_cn; s.scal[25] += (int)acc_any; s.scal[26] += (int)acc_vn; s.scal[27] += (int)acc_bar; }
#endif
    hard_flush(); // (visible to the other threads behind block_any's barrier)
    bool unsat = force_unsat;
    for (int l = tid; l < m; l += NT)
        if (s.cn_val[l] >= 0 && (PB ? ((const uint8_t *)s.par)[l] != 0 : s.par[l] != 0u)) unsat = true;
    const bool any = block_any<NT>(unsat, s);
    iters_done = max_iter;
    return any ? 0 : 1;
}

// Bitonic sort of (key, idx) pairs, ascending lexicographic == the reference's stable
// ascending argsort (index_sort, src/include/bpgd.cpp:384-389).
template <int NT>
__device__ __forceinline__ void sort_pairs(uint64_t *key, uint16_t *idx, int npad) {
    const int half = npad >> 1;
    for (int k = 2; k <= npad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < half; i += NT) {
                const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                const int hi = lo + j;
                const bool up = ((lo & k) == 0);
                const uint64_t ka = key[lo], kb = key[hi];
                const uint16_t ia = idx[lo], ib = idx[hi];
                const bool gt = (ka > kb) || (ka == kb && ia > ib);
                if (gt == up) { key[lo] = kb; key[hi] = ka; idx[lo] = ib; idx[hi] = ia; }
            }
            __syncthreads();
        }
    }
}

// Decided variable nodes.  Byte form (D = false: every kernel but the tuned osd_window ones): vn_val[v] = -1 live / decided
// value.  Bit form (D = true: 1.5 KB of LDS less per [[144]] window): one "decided" bit per node in the same array, the
// decided value is what hard[v] holds -- nothing writes hard[v] of a decided node afterwards (the post phase only
// stores decisions of live nodes).
template <bool D = false> __device__ __forceinline__ bool vn_decided(const Lds &s, int v) {
    if constexpr (D) return (((const uint32_t *)s.vn_val)[v >> 5] >> (v & 31)) & 1u;
    else return s.vn_val[v] >= 0;
}
template <bool D = false> __device__ __forceinline__ int vn_value(const Lds &s, int v) {
    if constexpr (D) return vn_decided<true>(s, v) ? (int)s.hard[v] : -1;
    else return s.vn_val[v];
}
template <bool D = false> __device__ __forceinline__ void vn_mark0(Lds &s, int v) { // decided 0; (bit form) hard[v] follows before anybody asks for the value
    if constexpr (D) atomicOr(&((uint32_t *)s.vn_val)[v >> 5], 1u << (v & 31));
    else s.vn_val[v] = 0;
}
template <bool D = false> __device__ __forceinline__ void vn_decide(Lds &s, int v, int val) {
    if constexpr (D) atomicOr(&((uint32_t *)s.vn_val)[v >> 5], 1u << (v & 31));
    else s.vn_val[v] = (int8_t)val;
    s.hard[v] = (uint8_t)val;
}
template <int NT, bool D = false> __device__ __forceinline__ void vn_reset(Lds &s, int n) {
    if constexpr (D) {
        for (int i = threadIdx.x; i < (n + 31) / 32; i += NT) ((uint32_t *)s.vn_val)[i] = 0u;
        for (int v = threadIdx.x; v < n; v += NT) s.hard[v] = 0;
    } else {
        for (int v = threadIdx.x; v < n; v += NT) { s.vn_val[v] = -1; s.hard[v] = 0; }
    }
}
template <int NT, bool D = false> __device__ __forceinline__ void vn_backup(const Lds &s, int n, char *bak) { // bak: n bytes for the marks, n for hard
    if constexpr (D) {
        for (int i = threadIdx.x; i < (n + 31) / 32; i += NT) ((uint32_t *)bak)[i] = ((const uint32_t *)s.vn_val)[i];
        for (int i = threadIdx.x; i < n; i += NT) bak[n + i] = (char)s.hard[i];
    } else {
        for (int i = threadIdx.x; i < n; i += NT) { bak[i] = (char)s.vn_val[i]; bak[n + i] = (char)s.hard[i]; }
    }
}
template <int NT, bool D = false> __device__ __forceinline__ void vn_restore(Lds &s, int n, const char *bak) {
    if constexpr (D) {
        for (int i = threadIdx.x; i < (n + 31) / 32; i += NT) ((uint32_t *)s.vn_val)[i] = ((const uint32_t *)bak)[i];
        for (int i = threadIdx.x; i < n; i += NT) s.hard[i] = (uint8_t)bak[n + i];
    } else {
        for (int i = threadIdx.x; i < n; i += NT) { s.vn_val[i] = (int8_t)bak[i]; s.hard[i] = (uint8_t)bak[n + i]; }
    }
}

// Marks everything but the `keep` smallest (key, index) pairs: vn_val[v] = 0 for the columns the
// reference's stable argsort (index_sort, bpgd.cpp:384-389) puts at positions keep.. (osd_window.pyx:
// 178-183 only uses that tail as a set).  MSB-first radix select on the 64-bit keys, 8 bits per
// pass; every wave scans the 256-bin histogram itself, so a pass costs one barrier.  Ties on the
// boundary key go to the smallest indices, which is what a stable sort does.
// hist: 3 x 256 ints of scratch.  Ends with a barrier.
template <int NT>
__device__ __forceinline__ void select_smallest(const uint64_t *key, int n, int keep, int *hist, Lds &s) {
    constexpr bool DIET = SWD_P16(NT);
    const int tid = threadIdx.x, lane = tid & 63;
    uint64_t prefix = 0, pmask = 0;
    int need = keep;
    bool exact = false; // boundary falls between two bins: no tie to break
    for (int b = tid; b < 256; b += NT) hist[b] = 0;
    __syncthreads();
#pragma unroll 1
    for (int pass = 0; pass < 8; ++pass) {
        const int shift = 56 - 8 * pass;
        int *h = hist + (pass % 3) * 256, *hz = hist + ((pass + 1) % 3) * 256; // hz: last read two passes ago
        for (int b = tid; b < 256; b += NT) hz[b] = 0;
        for (int v = tid; v < n; v += NT) {
            const uint64_t k = key[v];
            if ((k & pmask) == prefix) atomicAdd(&h[(int)(k >> shift) & 255], 1);
        }
        __syncthreads();
        const int4 c4 = *(const int4 *)(h + 4 * lane);
        const int tot = c4.x + c4.y + c4.z + c4.w;
        int incl = tot;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int y = __shfl_up(incl, d, 64);
            if (lane >= d) incl += y;
        }
        const unsigned long long hit = __ballot(incl >= need);
        const int bl = __ffsll((long long)hit) - 1; // first lane whose bins reach the rank
        int before = __shfl(incl - tot, bl, 64);
        const int cx = __shfl(c4.x, bl, 64), cy = __shfl(c4.y, bl, 64), cz = __shfl(c4.z, bl, 64), cw = __shfl(c4.w, bl, 64);
        int bin = 4 * bl, cnt = cx;
        if (before + cnt < need) { before += cnt; ++bin; cnt = cy; }
        if (before + cnt < need) { before += cnt; ++bin; cnt = cz; }
        if (before + cnt < need) { before += cnt; ++bin; cnt = cw; }
        prefix |= (uint64_t)bin << shift;
        pmask |= 0xFFull << shift;
        need -= before;
        if (cnt == need) { exact = true; break; } // the whole bin is kept
    }
    // keep: (k & pmask) < prefix, and of the keys == prefix the first `need` in index order
    const int ch = (n + NT - 1) / NT;
    const int v0 = tid * ch, v1 = min(n, v0 + ch);
    if (exact) {
        for (int v = tid; v < n; v += NT)
            if ((key[v] & pmask) > prefix) vn_mark0<DIET>(s, v);
        __syncthreads();
        return;
    }
    int eq = 0;
    for (int v = v0; v < v1; ++v) eq += (key[v] == prefix) ? 1 : 0;
    int tot;
    int rank = block_exscan<NT>(eq, s, tot);
    for (int v = v0; v < v1; ++v) {
        const uint64_t k = key[v];
        if (k > prefix) vn_mark0<DIET>(s, v);
        else if (k == prefix) { if (rank >= need) vn_mark0<DIET>(s, v); ++rank; }
    }
    __syncthreads();
}

// the same behind a function boundary (its own register allocation; the Lds descriptor travels by value)
template <int NT>
__device__ __attribute__((noinline)) void select_smallest_call(const uint64_t *key, int n, int keep, int *hist, Lds s) { select_smallest<NT>(key, n, keep, hist, s); }

// llr[list[0]] + llr[list[1]] + ... added one by one in list order (what the reference's loops do), by one wave: the loads of 64
// entries are in flight together, the additions walk the lanes.  Every lane returns the sum.
__device__ __forceinline__ double ordered_llr_sum_wave(const double *llr, const uint16_t *list, int total) {
    const int lane = threadIdx.x & 63;
    double pm = 0.0;
    for (int base = 0; base < total; base += 64) {
        const int i = base + lane;
        const long long xb = __double_as_longlong((i < total) ? llr[list[i]] : 0.0);
        const int cnt = min(64, total - base);
        for (int k = 0; k < cnt; ++k) { // k wave-uniform
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)xb, k), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(xb >> 32), k);
            pm += __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
        }
    }
    return pm;
}

// sum of llr[v] over hard[v]==1 in ascending v (min_pm, osd_window.pyx:168-169 / 233-235).
// `list` must hold n u16.  Result valid on every thread.
template <int NT>
__device__ __forceinline__ double ordered_pm(const SwdGraphDev &g, Lds &s, uint16_t *list) {
    const int n = g.n;
    const int ch = (n + NT - 1) / NT;
    const int v0 = threadIdx.x * ch, v1 = min(n, v0 + ch);
    int cnt = 0;
    for (int v = v0; v < v1; ++v) cnt += s.hard[v] ? 1 : 0;
    int total;
    int pos = block_exscan<NT>(cnt, s, total);
    for (int v = v0; v < v1; ++v)
        if (s.hard[v]) list[pos++] = (uint16_t)v;
    __syncthreads();
    double *dres = s.dbl;
    if (threadIdx.x < 64) {
        const double pm = ordered_llr_sum_wave(g.llr, list, total);
        if (threadIdx.x == 0) *dres = pm;
    }
    __syncthreads();
    return *dres;
}

// vn_set_value (osd_window.pyx:340-368) executed by wave 0: lane k takes the k-th neighbour
// check of vn (distinct checks, so the updates are independent).  Returns true on
// contradiction.
template <bool D = false>
__device__ __forceinline__ bool vn_set_value_wave(const SwdGraphDev &g, Lds &s, int vn, int value) {
    const int lane = threadIdx.x & 63;
    // (the edge table is padded beyond a column's degree: no load of the degree in front of the edge loads)
    const uint32_t e = (lane < g.D) ? g.vn_edge[lane * g.n + vn] : SWD_PAD_EDGE;
    if (lane == 0) vn_decide<D>(s, vn, value);
    bool bad = false;
    if (e != SWD_PAD_EDGE) {
        const int l = swd_edge_lane(e), j = swd_edge_j(e);
        int cv = s.cn_val[l];
        if (cv >= 0) {
            const int d = (int)s.cn_deg[l] - 1;
            if (value) cv ^= 1;
            lm_set<D>(s, l, lm_get<D>(s, l) & ~(1ull << j));
            if (d == 0) {
                if (cv != 0) bad = true;
                cv = -1;
            }
            s.cn_val[l] = (int8_t)cv;
            s.cn_deg[l] = (uint8_t)d;
        }
    }
    wave_fence();
    return __ballot(bad) != 0ull;
}

// peel (osd_window.pyx:306-338) on wave 0, reproducing the reference's sweep order: the next
// check handled is the lowest original index >= sweep pointer with live degree 1, wrapping to
// a new sweep when the current one is exhausted.  Returns true on contradiction.
template <bool D = false, int NT = SWD_MAX_M> // NT: threads of the workgroup (>= m: bounds the checks per lane)
__device__ __forceinline__ bool peel_wave(const SwdGraphDev &g, Lds &s) {
    const int lane = threadIdx.x & 63;
    int ptr = 0;
    // original index << 16 | lane number of the checks this lane scans: loaded when the first sweep finds something to do
    // (most calls find nothing), kept for the later sweeps of the call; the minimum over the packed word brings the lane
    // number along with the index
    constexpr int PW = NT / 64;
    uint32_t pk[PW];
    bool loaded = false;
    for (;;) {
        uint32_t ones = 0; // bit k: check lane + 64 k is live with degree 1
#pragma unroll
        for (int k = 0; k < PW; ++k) {
            const int l = lane + 64 * k;
            if (l < g.m && s.cn_val[l] >= 0 && s.cn_deg[l] == 1) ones |= 1u << k;
        }
        if (__ballot(ones != 0u) == 0ull) return false;
        if (!loaded) {
#pragma unroll
            for (int k = 0; k < PW; ++k) {
                const int l = lane + 64 * k;
                pk[k] = (64 * k < g.m) ? (((uint32_t)g.perm[min(l, g.m - 1)] << 16) | (uint32_t)l) : 0u; // (64 k < m: wave-uniform)
            }
            loaded = true;
        }
        uint32_t best_ge = 0xFFFFFFFFu, best_all = 0xFFFFFFFFu;
#pragma unroll
        for (int k = 0; k < PW; ++k) {
            if ((ones >> k) & 1u) {
                best_all = min(best_all, pk[k]);
                if ((int)(pk[k] >> 16) >= ptr) best_ge = min(best_ge, pk[k]);
            }
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            best_ge = min(best_ge, (uint32_t)__shfl_xor((int)best_ge, d, 64));
            best_all = min(best_all, (uint32_t)__shfl_xor((int)best_all, d, 64));
        }
        const uint32_t pick = (best_ge != 0xFFFFFFFFu) ? best_ge : best_all;
        const int c = (int)(pick >> 16), l = (int)(pick & 0xFFFFu);
        const uint64_t mk = lm_get<D>(s, l);
        const int j = __ffsll((long long)mk) - 1;
        const int vn = g.row_col[s.jptr[j] + l];
        const int val = s.cn_val[l];
        if (vn_set_value_wave<D>(g, s, vn, val)) return true;
        ptr = c + 1;
    }
}

// OSD-0 on wave 0 (osd_window.pyx:201-240).  GF(2) elimination in "transform" form: T starts
// as the m x m identity and accumulates the row operations; a sorted column (<= D ones) is
// reduced by XOR-ing the <= D columns of T it selects, so the work per scanned column does
// not depend on n.  Pivot rule = mod2sparse_decomp_osd (mod2sparse_extra.cpp:186-195): first
// column in priority order with a 1 in a not-yet-pivoted row, lowest such row.  Row
// operations are applied to all rows (Gauss-Jordan) which gives the same solution as the
// reference's LU + forward/backward substitution (mod2sparse_extra.cpp:78-106) because both
// solve the same invertible pivot-row x pivot-column system with zeros elsewhere.
//
// T is stored word-major, Tw[w*m + j] = word w of column j (bit r of it = T[64w + r][j]), so that
// lanes walking consecutive columns hit consecutive LDS words.  The wave evaluates 64/WG sorted
// columns per step (WG = words per column rounded up to a power of two; lane = column*WG + word):
// every column of the step that reduces to zero on the unpivoted rows is dependent and stays so,
// which lets whole runs of dependent columns be skipped at once -- the scan typically walks ~900
// columns for 216 pivots.  crows[p*DM + k] = original row of the k-th one of the p-th sorted column
// (0xFFFF pad), staged in LDS by the whole workgroup for p < nst.
__device__ __forceinline__ int osd_tidx(int row, int w, int m) { return w * m + row; }

template <int NT, int DM>
__device__ __forceinline__ int osd0_block(const SwdGraphDev &g, Lds &s, const uint16_t *order, uint64_t *Tw,
                          uint64_t *Sbuf, uint16_t *piv_col, uint16_t *piv_row, const uint8_t *synd_b,
                          const uint16_t *crows, int nst, int *npiv_out) {
    // General m (here: m > 256): wave 0 evaluates 64/WG sorted columns per step against T in LDS and picks the
    // pivot; the row operation on T (m columns x wm words) is shared by ALL threads of the workgroup -- with
    // one wave it was 13.6k cycles per pivot for the [[288,12,18]] windows (m = 576, wm = 9).  Two barriers
    // per step; the step outcome travels through ctl[] (found | done << 1, pivot row).  Every thread calls.
    const int tid = threadIdx.x, lane = tid & 63;
    const int m = g.m, n = g.n, wm = g.wm, rank = g.rank;
    int WG = 1;
    while (WG < wm) WG <<= 1;
    const int NB = 64 / WG;            // columns per step
    const int c = lane / WG, w = lane % WG;
    const bool wact = w < wm;
    const int wl = wact ? w : 0;
    int *ctl = s.iaux + 8;
    uint64_t Pw = 0;                    // word w of the pivoted-row mask (replicated per column group), wave 0
    int npiv = 0, rowadds = 0, p = 0;
#ifdef SWD_OSDPROF // diagnostic build: cycles of the pick (wave 0), the barrier waits and the row operation, per elimination
    long long tp_ = 0, tb_ = 0, tu_ = 0, q0_, q1_, q2_, q3_;
    int nsteps_ = 0;
#endif
    for (;;) {
#ifdef SWD_OSDPROF
        q0_ = clock64();
#endif
        if (tid < 64) {
            int found = 0;
            while (!found && p < n && npiv < rank) { // T only changes at a pivot: runs of dependent columns need no barrier
                const int pc = p + c;
                const bool cval = wact && pc < n;
                int rows[DM];
                if (p + NB <= nst) { // wave-uniform: whole step inside the staged prefix
#pragma unroll
                    for (int k = 0; k < DM; ++k) rows[k] = crows[pc * DM + k];
                } else {             // beyond the staged prefix (rare): straight from the graph
                    const int v = cval ? (int)order[pc] : 0;
                    const int deg = cval ? (int)g.col_deg[v] : 0;
#pragma unroll
                    for (int k = 0; k < DM; ++k) rows[k] = (k < deg) ? (int)g.vn_row[k * n + v] : 0xFFFF;
                }
                uint64_t tw[DM];
#pragma unroll
                for (int k = 0; k < DM; ++k) tw[k] = Tw[osd_tidx(rows[k] == 0xFFFF ? 0 : rows[k], wl, m)];
                uint64_t red = 0;
#pragma unroll
                for (int k = 0; k < DM; ++k) red ^= (rows[k] == 0xFFFF) ? 0ull : tw[k];
                const uint64_t cand = cval ? (red & ~Pw) : 0ull;
                const unsigned long long bal = __ballot(cand != 0ull);
                if (bal == 0ull) {
                    p += NB;
                } else {
                    const int fl = __ffsll((long long)bal) - 1; // first column with a usable 1, its lowest word
                    const int cs = fl / WG, ws = fl % WG;
                    const uint64_t cw = __shfl(cand, fl, 64);
                    const int bit = __ffsll((long long)cw) - 1;
                    const int r = ws * 64 + bit;
                    {   // row additions the reference's LU would apply: unpivoted rows with a 1 in this column
                        uint64_t un = (c == cs) ? cand : 0ull;
                        if (lane == fl) un &= ~(1ull << bit);
                        rowadds += __popcll(un);
                    }
                    if (c == cs && wact) Sbuf[w] = (w == ws) ? (red & ~(1ull << bit)) : red;
                    if (w == ws) Pw |= 1ull << bit;
                    if (lane == 0) { piv_col[npiv] = order[p + cs]; piv_row[npiv] = (uint16_t)r; ctl[1] = r; }
                    ++npiv;
                    p += cs + 1;
                    found = 1;
                }
            }
            if (lane == 0) ctl[0] = found | ((p < n && npiv < rank) ? 0 : 2);
        }
#ifdef SWD_OSDPROF
        q1_ = clock64();
#endif
        __syncthreads();
#ifdef SWD_OSDPROF
        q2_ = clock64();
#endif
        const int c0 = ctl[0];
        if (c0 & 1) { // T[:, j] ^= S for every column j of T with T[r][j] = 1
            const int r = ctl[1];
            const int ws = r >> 6;
            const uint64_t rb = 1ull << (r & 63);
            for (int j = tid; j < m; j += NT) {
                if (Tw[osd_tidx(j, ws, m)] & rb) {
                    for (int x = 0; x < wm; ++x) Tw[osd_tidx(j, x, m)] ^= Sbuf[x];
                }
            }
        }
        __syncthreads();
#ifdef SWD_OSDPROF
        q3_ = clock64();
        tp_ += q1_ - q0_; tb_ += q2_ - q1_; tu_ += q3_ - q2_; ++nsteps_;
        if ((c0 & 2) && (tid == 0 || tid == 512) && (blockIdx.x & 63) == 0)
            printf("osdprof thread %d: eliminations %d  pick %lld  barrier wait %lld  row operation + barrier %lld cycles\n", tid, nsteps_, tp_, tb_, tu_);
#endif
        if (c0 & 2) break;
    }
    if (tid < 64) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) rowadds += __shfl_xor(rowadds, d, 64);
        // y = T * s  (s in original row order): lanes over the columns of T, then a wave XOR-reduction
        for (int x = 0; x < wm; ++x) {
            uint64_t acc = 0;
            for (int j = lane; j < m; j += 64)
                if (synd_b[j]) acc ^= Tw[osd_tidx(j, x, m)];
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) acc ^= __shfl_xor(acc, d, 64);
            if (lane == 0) Sbuf[x] = acc;
        }
        wave_fence();
        for (int i = lane; i < npiv; i += 64) {
            const int r = piv_row[i];
            s.hard[piv_col[i]] = (uint8_t)((Sbuf[r >> 6] >> (r & 63)) & 1ull);
        }
        wave_fence();
        *npiv_out = npiv;
    }
    return rowadds;
}

__device__ __forceinline__ uint64_t wave_read64(uint64_t v, int srclane) { // srclane wave-uniform
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, srclane);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), srclane);
    return ((uint64_t)hi << 32) | lo;
}

// General m in column form (256 < m <= 64 * WMC, 1024 threads).  The transform matrix lives in registers, one column
// per thread of nine "column waves"; LDS holds a mirror that only the column evaluations read.  Wave 0 (the resolver)
// evaluates a batch of 64 sorted columns against the mirror, one per lane, and then resolves every pivot among them on
// its own, without a barrier: first column with a one in an unpivoted row, lowest such row r, S = its reduced vector
// without bit r; the batch's later columns get the row operation "rows i != r with u[i] = 1 get row r added" right
// there (y ^= S if y[r] -- what re-evaluating them against the updated T would give; columns before the pivot column
// were dependent and stay so), and (S, r) goes into a ring in LDS.  The column waves follow the ring at their own pace
// (col ^= S if col[r]); two barriers per batch of 64 columns bracket the refresh of the mirror.  The resolver and the
// column waves sit on different SIMDs (waves 1-3, 5-7, 9-11 hold the columns), so the serial part -- about 70 vector
// instructions per pivot -- overlaps the 9 x 40 of the update.  (osd0_block below: 3.4k cycles per pivot, wave 0
// evaluating against T in LDS while 15 waves wait, then all threads rewriting T in LDS, two barriers per pivot.)
#define SWD_LDS_AS __attribute__((address_space(3)))
#ifndef SWD_OSD_QUAD
#define SWD_OSD_QUAD 1 // m <= 256 on at least four waves: the elimination with the transform matrix on the other waves (osd0_quad)
#endif
#ifndef SWD_WIDE_EVAL_DS
#define SWD_WIDE_EVAL_DS 0
#endif
#ifndef SWD_BIG_TBL
#define SWD_BIG_TBL 0 // experiment (round 5, bit-exact): large-graph kernels, full-graph phase: the variable-node pass reads the graph's tables (one row
                      // ahead) instead of a register cache that does not fit: set-up 34 -> 24 us, 40.0 -> 41.6 us per iteration, 160 -> 163 k decodes/s: neutral, off
#endif
#ifndef SWD_BIG_REC
#define SWD_BIG_REC 0 // experiment (round 5, bit-exact): large-graph kernels, full-graph phase: one record per check in LDS instead of a check-to-bit
                      // message per edge in HBM -- half the phase's traffic through L2 gone, 40.0 -> 38.8 us per iteration (the phase is bound by its
                      // own instruction stream and its register spills, not by that traffic), and the shortened graph's iterations of the same
                      // kernel 3.38 -> 3.78 us: 160 -> 156 k decodes/s, off
#endif
#ifndef SWD_BIG_HYBRID
#define SWD_BIG_HYBRID 1 // large-graph kernels: the top of the full graph's message array in the LDS region that idles during that phase
#endif
#ifndef SWD_OSD_WIDE
#define SWD_OSD_WIDE 1 // large-graph kernels: the column-form elimination on fifteen column waves (osd0_colsw) instead of osd0_block
#endif
#ifndef SWD_SPEC_LOAD
#define SWD_SPEC_LOAD 0 // experiment (round 5): the tuned osd_window kernels ask for the next unit's variable-node cache before they draw the
                        // ticket -- the loads are in flight during the three round trips of ticket, counter and state record: 9.37 -> 9.44 ms per
                        // launch, no gain (two other workgroups on the CU already run during a workgroup's waits)
#endif
#ifndef SWD_SERIAL_PRIO
#define SWD_SERIAL_PRIO 0 // s_setprio of a wave the rest of its workgroup waits for (single-wave eliminations, the column form's resolver)
#endif
#ifndef SWD_OSD_RING
#define SWD_OSD_RING 128 // row operations the column-form elimination can publish per round (ring entries in the exchange region)
#endif
// bit `rbit` of word `rw` (both wave-uniform) of a register-resident bit vector: a scalar branch per word instead of a select chain
template <int WMC>
__device__ __forceinline__ uint32_t osd_vec_bit(const uint64_t (&v)[WMC], int rw, int rbit) {
    uint32_t h = 0;
    switch (rw) {
#define SWD_CASE(x) case x: if constexpr (x < WMC) h = (rbit < 32) ? (uint32_t)v[x < WMC ? x : 0] : (uint32_t)(v[x < WMC ? x : 0] >> 32); break;
        SWD_CASE(0) SWD_CASE(1) SWD_CASE(2) SWD_CASE(3) SWD_CASE(4) SWD_CASE(5) SWD_CASE(6) SWD_CASE(7) SWD_CASE(8)
        SWD_CASE(9) SWD_CASE(10) SWD_CASE(11) SWD_CASE(12) SWD_CASE(13) SWD_CASE(14) SWD_CASE(15)
#undef SWD_CASE
    default: break;
    }
    return (h >> (rbit & 31)) & 1u;
}

template <int WMC>
__device__ __forceinline__ void osd_vec_setbit(uint64_t (&v)[WMC], int rw, int rbit) { // rw, rbit wave-uniform
    switch (rw) {
#define SWD_CASE(x) case x: if constexpr (x < WMC) v[x < WMC ? x : 0] |= 1ull << rbit; break;
        SWD_CASE(0) SWD_CASE(1) SWD_CASE(2) SWD_CASE(3) SWD_CASE(4) SWD_CASE(5) SWD_CASE(6) SWD_CASE(7) SWD_CASE(8)
        SWD_CASE(9) SWD_CASE(10) SWD_CASE(11) SWD_CASE(12) SWD_CASE(13) SWD_CASE(14) SWD_CASE(15)
#undef SWD_CASE
    default: break;
    }
}

#ifdef SWD_OSD_COLS_V1 // round 2-4 form (one batch of 64 sorted columns per pair of barriers), kept for A/B builds
template <int NT, int DM, int WMC>
__device__ __forceinline__ int osd0_cols(const SwdGraphDev &g, Lds &s, const uint16_t *order, uint64_t *Tw, uint64_t *Sbuf,
                                         uint16_t *piv_col, uint16_t *piv_row, const uint8_t *synd_b, const uint16_t *crows, int nst,
                                         int *npiv_out, char *slotmem) {
    constexpr int NBC = 64;
    static_assert(NT >= 768 && WMC <= 16, "wave 0 resolves, nine of the waves 1..11 hold the columns");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = g.m, n = g.n, wm = g.wm, rank = g.rank;
    constexpr int ES = (WMC + 2) & ~1; // words per ring entry: S, then pivot row | column within the batch << 16 (16-byte multiples)
    SWD_LDS_AS uint64_t *ringS = (SWD_LDS_AS uint64_t *)slotmem;                  // [NBC][ES] the batch's row operations
    SWD_LDS_AS uint64_t *Pl = ringS + NBC * ES;                                   // [WMC] pivoted rows
    SWD_LDS_AS int *ctl = (SWD_LDS_AS int *)(Pl + WMC + 1); // 0: published, 1: batch closed, 2: elimination finished, 3: pivots so far, 4: row additions
    volatile SWD_LDS_AS int *vctl = ctl;
    const bool colwave = (wave & 3) != 0 && wave < 12;
    const int jc = colwave ? (wave - 1 - (wave >> 2)) * 64 + lane : m; // the column of T this thread keeps
    uint64_t col[WMC];
#pragma unroll
    for (int x = 0; x < WMC; ++x) col[x] = (jc < m && x == (jc >> 6)) ? (1ull << (jc & 63)) : 0ull;
    if (tid == 0) { ctl[0] = 0; ctl[1] = 0; ctl[2] = 0; ctl[3] = 0; ctl[4] = 0; }
    if (tid < WMC) Pl[tid] = 0ull;
    __syncthreads();
    // resolver state: lane x < WMC keeps word x of the pivoted-row mask and counts the unpivoted ones of the pivot columns there
    // (a single wave issues one instruction of any kind per four cycles: the pivot search runs word-per-lane, not as scalar code)
    uint64_t Pmine = 0;
    int racc = 0;
    int npiv = 0, p = 0;
#ifdef SWD_OSDPROF // diagnostic build: cycles of the resolver (evaluation, pivots), of a column wave (applying, waiting) and between the barriers
    long long q_eval = 0, q_res = 0, q_app = 0, q_wait = 0, q_sync = 0, q0_;
    int q_batches = 0;
#endif
    for (;;) {
        const int npiv0 = npiv, p0 = p;
#ifdef SWD_OSDPROF
        q0_ = clock64(); ++q_batches;
#endif
        if (wave == 0) {
            const int pc = p + lane;
            const bool cval = pc < n;
            int rows[DM];
            if (p + NBC <= nst) { // uniform: whole batch inside the staged prefix
#pragma unroll
                for (int kk = 0; kk < DM; ++kk) rows[kk] = crows[pc * DM + kk];
            } else {
                const int v = cval ? (int)order[pc] : 0;
                const int deg = cval ? (int)g.col_deg[v] : 0;
#pragma unroll
                for (int kk = 0; kk < DM; ++kk) rows[kk] = (kk < deg) ? (int)g.vn_row[kk * n + v] : 0xFFFF;
            }
            uint64_t red[WMC];
#pragma unroll
            for (int x = 0; x < WMC; ++x) {
                red[x] = 0ull;
                if (x < wm) { // uniform
#pragma unroll
                    for (int kk = 0; kk < DM; ++kk) red[x] ^= (rows[kk] == 0xFFFF) ? 0ull : Tw[osd_tidx(rows[kk] == 0xFFFF ? 0 : rows[kk], x, m)];
                }
            }
            bool alive = cval;
            int nb = 0; // pivots of this batch
#ifdef SWD_OSDPROF
            q_eval += clock64() - q0_; q0_ = clock64();
#endif
            uint64_t pb[WMC]; // the pivoted-row mask, every word in every lane (read back after each pivot, ahead of its use)
#pragma unroll
            for (int x = 0; x < WMC; ++x) pb[x] = Pl[x];
            while (npiv < rank) {
                uint32_t nz = 0;
#pragma unroll
                for (int x = 0; x < WMC; ++x) {
                    nz |= (uint32_t)red[x] & ~(uint32_t)pb[x];
                    nz |= (uint32_t)(red[x] >> 32) & ~(uint32_t)(pb[x] >> 32);
                }
                const unsigned long long bal = __ballot(alive && nz != 0u);
                if (bal == 0ull) break; // every remaining column of the batch is dependent
                const int cs = __ffsll((long long)bal) - 1;
                uint32_t lz = 0;
                asm volatile("" : "+v"(lz)); // the lane number, recomputed here: kept across the loop it is spilled and reloaded per pivot
                const int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, lz));
                SWD_LDS_AS uint64_t *ent = ringS + nb * ES;
                if (ln == cs) { // the pivot column's lane publishes its reduced vector (LDS operations of a wave execute in order)
#pragma unroll
                    for (int x = 0; x < WMC; ++x) ent[x] = red[x];
                }
                asm volatile("" ::: "memory");
                // one round trip: the vector a word per lane (pivot search) and every word in every lane (the update below)
                const uint64_t wv = ent[ln < WMC ? ln : 0];
                uint64_t S[WMC];
#pragma unroll
                for (int x = 0; x < WMC; ++x) S[x] = ent[x];
                const uint64_t c = (ln < WMC) ? (wv & ~Pmine) : 0ull; // its ones in unpivoted rows
                const unsigned long long balc = __ballot(c != 0ull);
                const int fx = __ffsll((long long)balc) - 1;
                const int bit = __builtin_amdgcn_readlane(__ffsll((long long)c) - 1, fx);
                racc += __popcll(c); // row additions the reference's LU would apply: unpivoted rows with a one in this column (the pivot itself is taken off at the end)
                if (ln == fx) {
                    __hip_atomic_fetch_and(&ent[fx], ~(1ull << bit), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_or(&Pl[fx], 1ull << bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    Pmine |= 1ull << bit;
                }
                if (ln == 0) ent[WMC] = (uint64_t)(uint32_t)((fx * 64 + bit) | (cs << 16));
                asm volatile("" ::: "memory"); // a wave's LDS operations execute in order: the count follows the entry
                if (ln == 0) ctl[0] = nb + 1;
#pragma unroll
                for (int x = 0; x < WMC; ++x) pb[x] = Pl[x]; // for the next pivot
                // the batch's later columns under the same row operation: y ^= S if y[r]; S here still has bit r, which y keeps
                const uint32_t ybit = osd_vec_bit<WMC>(red, fx, bit);
                if (ln <= cs) alive = false;
                else if (alive && ybit) {
#pragma unroll
                    for (int x = 0; x < WMC; ++x) red[x] ^= S[x];
                    osd_vec_setbit<WMC>(red, fx, bit);
                }
                ++npiv; ++nb;
            }
            p += NBC;
#ifdef SWD_OSDPROF
            q_res += clock64() - q0_; q0_ = clock64();
#endif
            int rsum = racc;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) rsum += __shfl_xor(rsum, d, 64);
            if (lane == 0) {
                ctl[3] = npiv; ctl[4] = rsum - npiv;
                ctl[2] = (p < n && npiv < rank) ? 0 : 1;
            }
            asm volatile("" ::: "memory");
            if (lane == 0) ctl[1] = 1;
        } else if (colwave) {
            int done_ops = 0;
            for (;;) {
                const int closed = vctl[1]; // read before the count: a closed batch's count is final
                const int avail = vctl[0];
#ifdef SWD_OSDPROF
                q_wait += clock64() - q0_; q0_ = clock64();
#endif
                while (done_ops < avail) {
                    SWD_LDS_AS const uint64_t *ent = ringS + done_ops * ES;
                    uint64_t S[WMC];
#pragma unroll
                    for (int x = 0; x < WMC; ++x) S[x] = ent[x];
                    const int r = __builtin_amdgcn_readfirstlane((int)(uint32_t)ent[WMC]) & 0xFFFF;
                    const int rw = r >> 6, rbit = r & 63;
                    if (osd_vec_bit<WMC>(col, rw, rbit)) {
#pragma unroll
                        for (int x = 0; x < WMC; ++x) col[x] ^= S[x];
                    }
                    ++done_ops;
                }
#ifdef SWD_OSDPROF
                q_app += clock64() - q0_; q0_ = clock64();
#endif
                if (closed) break;
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads(); // the batch's row operations are in every column; ctl[] is final
        const int fin = ctl[2];
        npiv = ctl[3];
        p = p0 + NBC;
        if (tid == 0) { ctl[0] = 0; ctl[1] = 0; } // nobody reads these two between the barriers
        if (tid >= 64 && tid - 64 < npiv - npiv0) { // the batch's pivots (another wave than the resolver looks the columns up)
            const int e = (int)(uint32_t)ringS[(tid - 64) * ES + WMC];
            piv_col[npiv0 + tid - 64] = order[p0 + (e >> 16)];
            piv_row[npiv0 + tid - 64] = (uint16_t)(e & 0xFFFF);
        }
        if (jc < m) {
#pragma unroll
            for (int x = 0; x < WMC; ++x)
                if (x < wm) Tw[osd_tidx(jc, x, m)] = col[x]; // the mirror (after the last batch: what the higher-order sweep reads)
        }
        if (tid < wm) Sbuf[tid] = 0ull; // (used after the last batch)
        __syncthreads();
#ifdef SWD_OSDPROF
        q_sync += clock64() - q0_;
        if (fin && (tid == 0 || tid == 64) && (blockIdx.x & 63) == 0)
            printf("osdprof cols thread %d: batches %d pivots %d | resolver: evaluation %lld resolve %lld | column wave: apply %lld wait %lld | rest of the batch %lld cycles\n",
                   tid, q_batches, npiv, q_eval, q_res, q_app, q_wait, q_sync);
#endif
        if (fin) break;
    }
    // y = T * s (s in original row order)
    const bool on = jc < m && synd_b[jc < m ? jc : 0] != 0;
    if (colwave) {
#pragma unroll
        for (int x = 0; x < WMC; ++x) {
            if (x < wm) { // uniform
                uint64_t acc = on ? col[x] : 0ull;
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) acc ^= __shfl_xor(acc, d, 64);
                if (lane == 0 && acc) atomicXor((unsigned long long *)&Sbuf[x], (unsigned long long)acc);
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < npiv; i += NT) {
        const int r = piv_row[i];
        s.hard[piv_col[i]] = (uint8_t)((Sbuf[r >> 6] >> (r & 63)) & 1ull);
    }
    *npiv_out = npiv;
    return ctl[4];
}

#else
// Round 5: ROUNDS of up to four batches per pair of barriers.  What the instrumented build showed for the form above on the [[288,12,18]]
// windows: of ~1.0 M cycles per elimination the resolver spends ~0.4 M evaluating its batches (54 LDS reads per lane whose
// latencies the 128-VGPR allocation serialises) and another share in 47 x 2 workgroup barriers -- the scan walks ~3000 sorted columns
// for its 576 pivots, and the ~2400 columns behind the first ~600 hold two or three pivots per batch.  Now three HELPER waves
// (13, 14, 15: one on each SIMD the resolver does not use) evaluate the next three batches against the same mirror while the
// resolver works on the first, follow the ring like the column waves do (a candidate column takes a row operation exactly like a
// column of T: y ^= S if y[r]), and hand their reduced vectors over through LDS when the resolver reaches their batch: the
// resolver evaluates only the first batch of a round, and the mirror refresh + two barriers come once per round.  A round ends
// after four batches, when the ring (RING operations) could overflow in the next batch, or with the elimination.
template <int NT, int DM, int WMC>
__device__ __forceinline__ int osd0_cols(const SwdGraphDev &g, Lds &s, const uint16_t *order, uint64_t *Tw, uint64_t *Sbuf,
                                         uint16_t *piv_col, uint16_t *piv_row, const uint8_t *synd_b, const uint16_t *crows, int nst,
                                         int *npiv_out, char *slotmem) {
    constexpr int NBC = 64;      // columns per batch
    constexpr int RING = SWD_OSD_RING; // row operations a round can publish (make_layout sizes the exchange region with the same constant)
    constexpr int NHELP = 3;     // helper waves
    static_assert(NT >= 1024 && WMC <= 16, "wave 0 resolves, nine of the waves 1..11 hold the columns, waves 13..15 help");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = g.m, n = g.n, wm = g.wm, rank = g.rank;
    constexpr int ES = (WMC + 2) & ~1; // words per ring entry: S, then pivot row | column within the round << 16 (16-byte multiples)
    SWD_LDS_AS uint64_t *ringS = (SWD_LDS_AS uint64_t *)slotmem;                  // [RING][ES] the round's row operations
    SWD_LDS_AS uint64_t *hbuf = ringS + RING * ES;                                // [WMC][64] hand-over of a helper's reduced vectors (word-major)
    SWD_LDS_AS uint64_t *Pl = hbuf + WMC * NBC;                                   // [WMC] pivoted rows
    // 0: published, 1: round closed, 2: elimination finished, 3: pivots so far, 4: row additions, 5: batch the resolver works on,
    // 6: batch whose vectors lie in hbuf, 7: batches of the round, 8..11: operations published when batch b began
    SWD_LDS_AS int *ctl = (SWD_LDS_AS int *)(Pl + WMC + 1);
    volatile SWD_LDS_AS int *vctl = ctl;
    const bool colwave = (wave & 3) != 0 && wave < 12;
    const int helper = (wave >= 13 && wave <= 15) ? wave - 12 : 0; // helper h evaluates batch h of a round
    const int jc = colwave ? (wave - 1 - (wave >> 2)) * 64 + lane : m; // the column of T this thread keeps
    uint64_t col[WMC];
#pragma unroll
    for (int x = 0; x < WMC; ++x) col[x] = (jc < m && x == (jc >> 6)) ? (1ull << (jc & 63)) : 0ull;
    if (tid < 16) ctl[tid] = 0;
    if (tid < WMC) Pl[tid] = 0ull;
    __syncthreads();
    // the reduced vector of sorted column pc against the mirror (resolver: first batch of a round; helpers: the next ones)
    auto evaluate = [&](int pb, uint64_t (&red)[WMC]) {
        const int pc = pb + lane;
        const bool cval = pc < n;
        int rows[DM];
        if (pb + NBC <= nst) { // uniform: whole batch inside the staged prefix
#pragma unroll
            for (int kk = 0; kk < DM; ++kk) rows[kk] = crows[pc * DM + kk];
        } else {
            const int v = cval ? (int)order[pc] : 0;
            const int deg = cval ? (int)g.col_deg[v] : 0;
#pragma unroll
            for (int kk = 0; kk < DM; ++kk) rows[kk] = (kk < deg) ? (int)g.vn_row[kk * n + v] : 0xFFFF;
        }
#pragma unroll
        for (int x = 0; x < WMC; ++x) {
            red[x] = 0ull;
            if (x < wm) { // uniform
#pragma unroll
                for (int kk = 0; kk < DM; ++kk) red[x] ^= (rows[kk] == 0xFFFF) ? 0ull : Tw[osd_tidx(rows[kk] == 0xFFFF ? 0 : rows[kk], x, m)];
            }
        }
    };
    // resolver state: lane x < WMC keeps word x of the pivoted-row mask and counts the unpivoted ones of the pivot columns there
    // (a single wave issues one instruction of any kind per four cycles: the pivot search runs word-per-lane, not as scalar code)
    uint64_t Pmine = 0;
    int racc = 0;
    int npiv = 0, p = 0;
#ifdef SWD_OSDPROF // diagnostic build: cycles of the resolver (evaluation, hand-over waits, pivots), of a column wave (applying, waiting) and between the barriers
    long long q_eval = 0, q_res = 0, q_app = 0, q_wait = 0, q_sync = 0, q_hand = 0, q0_;
    int q_rounds = 0, q_batches = 0;
#endif
    for (;;) {
        const int npiv0 = npiv, p0 = p;
#ifdef SWD_OSDPROF
        q0_ = clock64(); ++q_rounds;
#endif
        if (wave == 0) {
#if SWD_SERIAL_PRIO
            __builtin_amdgcn_s_setprio(SWD_SERIAL_PRIO);
#endif
            int used = 0, nbatch = 0; // operations published this round, batches begun
            bool fin = false;
            for (;;) {
                uint64_t red[WMC];
                if (nbatch == 0) evaluate(p, red);
                else { // the helper's vectors, every operation of the round so far applied
                    if (lane == 0) { ctl[8 + nbatch] = used; }
                    asm volatile("" ::: "memory");
                    if (lane == 0) ctl[5] = nbatch;
                    while (vctl[6] != nbatch) __builtin_amdgcn_s_sleep(1);
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int x = 0; x < WMC; ++x) red[x] = (x < wm) ? hbuf[x * NBC + lane] : 0ull;
                }
                bool alive = p + lane < n;
#ifdef SWD_OSDPROF
                if (nbatch == 0) q_eval += clock64() - q0_; else q_hand += clock64() - q0_;
                q0_ = clock64(); ++q_batches;
#endif
                uint64_t pb[WMC]; // the pivoted-row mask, every word in every lane (read back after each pivot, ahead of its use)
#pragma unroll
                for (int x = 0; x < WMC; ++x) pb[x] = Pl[x];
                while (npiv < rank) {
                    uint32_t nz = 0;
#pragma unroll
                    for (int x = 0; x < WMC; ++x) {
                        nz |= (uint32_t)red[x] & ~(uint32_t)pb[x];
                        nz |= (uint32_t)(red[x] >> 32) & ~(uint32_t)(pb[x] >> 32);
                    }
                    const unsigned long long bal = __ballot(alive && nz != 0u);
                    if (bal == 0ull) break; // every remaining column of the batch is dependent
                    const int cs = __ffsll((long long)bal) - 1;
                    uint32_t lz = 0;
                    asm volatile("" : "+v"(lz)); // the lane number, recomputed here: kept across the loop it is spilled and reloaded per pivot
                    const int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, lz));
                    SWD_LDS_AS uint64_t *ent = ringS + used * ES;
                    if (ln == cs) { // the pivot column's lane publishes its reduced vector (LDS operations of a wave execute in order)
#pragma unroll
                        for (int x = 0; x < WMC; ++x) ent[x] = red[x];
                    }
                    asm volatile("" ::: "memory");
                    // one round trip: the vector a word per lane (pivot search) and every word in every lane (the update below)
                    const uint64_t wv = ent[ln < WMC ? ln : 0];
                    uint64_t S[WMC];
#pragma unroll
                    for (int x = 0; x < WMC; ++x) S[x] = ent[x];
                    const uint64_t c = (ln < WMC) ? (wv & ~Pmine) : 0ull; // its ones in unpivoted rows
                    const unsigned long long balc = __ballot(c != 0ull);
                    const int fx = __ffsll((long long)balc) - 1;
                    const int bit = __builtin_amdgcn_readlane(__ffsll((long long)c) - 1, fx);
                    racc += __popcll(c); // row additions the reference's LU would apply: unpivoted rows with a one in this column (the pivot itself is taken off at the end)
                    if (ln == fx) {
                        __hip_atomic_fetch_and(&ent[fx], ~(1ull << bit), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_or(&Pl[fx], 1ull << bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        Pmine |= 1ull << bit;
                    }
                    if (ln == 0) ent[WMC] = (uint64_t)(uint32_t)((fx * 64 + bit) | ((nbatch * NBC + cs) << 16));
                    asm volatile("" ::: "memory"); // a wave's LDS operations execute in order: the count follows the entry
                    if (ln == 0) ctl[0] = used + 1;
#pragma unroll
                    for (int x = 0; x < WMC; ++x) pb[x] = Pl[x]; // for the next pivot
                    // the batch's later columns under the same row operation: y ^= S if y[r]; S here still has bit r, which y keeps
                    const uint32_t ybit = osd_vec_bit<WMC>(red, fx, bit);
                    if (ln <= cs) alive = false;
                    else if (alive && ybit) {
#pragma unroll
                        for (int x = 0; x < WMC; ++x) red[x] ^= S[x];
                        osd_vec_setbit<WMC>(red, fx, bit);
                    }
                    ++npiv; ++used;
                }
                p += NBC; ++nbatch;
#ifdef SWD_OSDPROF
                q_res += clock64() - q0_; q0_ = clock64();
#endif
                fin = !(p < n && npiv < rank);
                if (fin || nbatch > NHELP || used + NBC > RING) break;
            }
            int rsum = racc;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) rsum += __shfl_xor(rsum, d, 64);
            if (lane == 0) {
                ctl[3] = npiv; ctl[4] = rsum - npiv; ctl[7] = nbatch;
                ctl[2] = fin ? 1 : 0;
            }
            asm volatile("" ::: "memory");
            if (lane == 0) ctl[1] = 1;
#if SWD_SERIAL_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
        } else if (colwave) {
            int done_ops = 0;
            for (;;) {
                const int closed = vctl[1]; // read before the count: a closed round's count is final
                const int avail = vctl[0];
#ifdef SWD_OSDPROF
                q_wait += clock64() - q0_; q0_ = clock64();
#endif
                while (done_ops < avail) {
                    SWD_LDS_AS const uint64_t *ent = ringS + done_ops * ES;
                    uint64_t S[WMC];
#pragma unroll
                    for (int x = 0; x < WMC; ++x) S[x] = ent[x];
                    const int r = __builtin_amdgcn_readfirstlane((int)(uint32_t)ent[WMC]) & 0xFFFF;
                    const int rw = r >> 6, rbit = r & 63;
                    if (osd_vec_bit<WMC>(col, rw, rbit)) {
#pragma unroll
                        for (int x = 0; x < WMC; ++x) col[x] ^= S[x];
                    }
                    ++done_ops;
                }
#ifdef SWD_OSDPROF
                q_app += clock64() - q0_; q0_ = clock64();
#endif
                if (closed) break;
                __builtin_amdgcn_s_sleep(1);
            }
        } else if (helper && p + helper * NBC < n) {
            // batch `helper` of this round: evaluated against the mirror of the round's start, then every operation the resolver
            // publishes before it reaches this batch; handed over when it does; dropped when the round ends first
            uint64_t red[WMC];
            evaluate(p + helper * NBC, red);
            int done_ops = 0;
            for (;;) {
                const int closed = vctl[1];
                const int cur = vctl[5];     // (read before the bound: bst is written first)
                const int target = (cur >= helper) ? vctl[8 + helper] : vctl[0];
                asm volatile("" ::: "memory");
                while (done_ops < target) {
                    SWD_LDS_AS const uint64_t *ent = ringS + done_ops * ES;
                    uint64_t S[WMC];
#pragma unroll
                    for (int x = 0; x < WMC; ++x) S[x] = ent[x];
                    const int r = __builtin_amdgcn_readfirstlane((int)(uint32_t)ent[WMC]) & 0xFFFF;
                    const int rw = r >> 6, rbit = r & 63;
                    if (osd_vec_bit<WMC>(red, rw, rbit)) {
#pragma unroll
                        for (int x = 0; x < WMC; ++x) red[x] ^= S[x];
                    }
                    ++done_ops;
                }
                if (cur >= helper) { // (cur >= helper implies target = the operations of the batches before this one, all applied now)
#pragma unroll
                    for (int x = 0; x < WMC; ++x)
                        if (x < wm) hbuf[x * NBC + lane] = red[x];
                    asm volatile("" ::: "memory");
                    if (lane == 0) ctl[6] = helper; // (a wave's LDS operations execute in order)
                    break;
                }
                if (closed) break;
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads(); // the round's row operations are in every column; ctl[] is final
        const int fin = ctl[2];
        npiv = ctl[3];
        p = p0 + NBC * ctl[7];
        if (tid == 0) { ctl[0] = 0; ctl[1] = 0; ctl[5] = 0; ctl[6] = 0; } // nobody reads these four between the barriers
        if (tid >= 64 && tid - 64 < npiv - npiv0) { // the round's pivots (another wave than the resolver looks the columns up)
            const int e = (int)(uint32_t)ringS[(tid - 64) * ES + WMC];
            piv_col[npiv0 + tid - 64] = order[p0 + (int)((uint32_t)e >> 16)];
            piv_row[npiv0 + tid - 64] = (uint16_t)(e & 0xFFFF);
        }
        if (jc < m) {
#pragma unroll
            for (int x = 0; x < WMC; ++x)
                if (x < wm) Tw[osd_tidx(jc, x, m)] = col[x]; // the mirror (after the last round: what the higher-order sweep reads)
        }
        if (tid < wm) Sbuf[tid] = 0ull; // (used after the last round)
        __syncthreads();
#ifdef SWD_OSDPROF
        q_sync += clock64() - q0_;
        if (fin && (tid == 0 || tid == 64 || tid == 13 * 64) && (blockIdx.x & 63) == 0)
            printf("osdprof cols2 thread %d: rounds %d batches %d pivots %d | resolver: evaluation %lld hand-over %lld resolve %lld | column wave: apply %lld wait %lld | rest of the round %lld cycles\n",
                   tid, q_rounds, q_batches, npiv, q_eval, q_hand, q_res, q_app, q_wait, q_sync);
#endif
        if (fin) break;
    }
    // y = T * s (s in original row order)
    const bool on = jc < m && synd_b[jc < m ? jc : 0] != 0;
    if (colwave) {
#pragma unroll
        for (int x = 0; x < WMC; ++x) {
            if (x < wm) { // uniform
                uint64_t acc = on ? col[x] : 0ull;
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) acc ^= __shfl_xor(acc, d, 64);
                if (lane == 0 && acc) atomicXor((unsigned long long *)&Sbuf[x], (unsigned long long)acc);
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < npiv; i += NT) {
        const int r = piv_row[i];
        s.hard[piv_col[i]] = (uint8_t)((Sbuf[r >> 6] >> (r & 63)) & 1ull);
    }
    *npiv_out = npiv;
    return ctl[4];
}
#endif // SWD_OSD_COLS_V1

// Large-graph kernels (scratch region in HBM), 256 < m <= 960: the column form on ALL fifteen waves behind the resolver -- wave w holds
// columns 64 (w - 1) .. 64 w - 1 of the transform matrix, one per lane, fifteen words each.  One batch of 64 sorted columns per pair
// of barriers (the round-2 protocol: the resolver publishes (S, r) into a ring, the column waves follow it at their own pace,
// col ^= S if col[r]; the barriers bracket the refresh of the mirror the evaluations read); the ring, the pivoted-row mask and the
// control words are LDS (`ringmem`: the layout's off_owide, `ring` entries -- a batch that fills the ring closes early and the next
// one starts behind its last pivot column); the mirror is wherever the OSD arrays are (LDS when the layout says osd_lds, else HBM).
// Replaces osd0_block for these graphs: 6500 cycles per pivot there -- wave 0 evaluating against T in memory, every thread
// rewriting T, two barriers per PIVOT -- on the 936-check model of IBM.ipynb:119.
template <int NT, int DM, int WMC>
__device__ __forceinline__ int osd0_colsw(const SwdGraphDev &g, Lds &s, const uint16_t *order, uint64_t *Tw, uint64_t *Sbuf,
                                          uint16_t *piv_col, uint16_t *piv_row, const uint8_t *synd_b, const uint16_t *crows, int nst,
                                          int *npiv_out, char *ringmem, int ring, bool lds_mirror) {
    constexpr int NBC = 64;
    static_assert(NT == 1024 && WMC <= 15, "wave 0 resolves, waves 1..15 hold the columns");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = g.m, n = g.n, wm = g.wm, rank = g.rank;
    constexpr int ES = 16; // words per ring entry: S, then pivot row | column within the batch << 16
    SWD_LDS_AS uint64_t *ringS = (SWD_LDS_AS uint64_t *)ringmem;                  // [ring][ES] the batch's row operations
    SWD_LDS_AS uint64_t *Pl = ringS + ring * ES;                                  // [16] pivoted rows
    SWD_LDS_AS int *ctl = (SWD_LDS_AS int *)(Pl + 16); // 0: published, 1: batch closed, 2: elimination finished, 3: pivots so far, 4: row additions, 5: first column of the next batch, 6: of the closing one
    volatile SWD_LDS_AS int *vctl = ctl;
    const bool colwave = wave != 0;
    const int jc = colwave ? (wave - 1) * 64 + lane : m; // the column of T this thread keeps
    uint64_t col[WMC];
#pragma unroll
    for (int x = 0; x < WMC; ++x) col[x] = (jc < m && x == (jc >> 6)) ? (1ull << (jc & 63)) : 0ull;
    if (tid < 8) ctl[tid] = 0;
    if (tid < 16) Pl[tid] = 0ull;
    __syncthreads();
    // resolver state: lane x < WMC keeps word x of the pivoted-row mask and counts the unpivoted ones of the pivot columns there
    uint64_t Pmine = 0;
    int racc = 0;
    int npiv = 0, p = 0;
#ifdef SWD_OSDPROF
    long long q_eval = 0, q_res = 0, q_app = 0, q_wait = 0, q_sync = 0, q0_;
    int q_batches = 0;
#endif
    for (;;) {
        const int npiv0 = npiv;
        int p0 = p;
#ifdef SWD_OSDPROF
        q0_ = clock64(); ++q_batches;
#endif
        if (wave == 0) {
          for (;;) { // batches, until one finds a pivot (the others change nothing anybody else would have to see)
            p0 = p;
#ifdef SWD_OSDPROF
            q0_ = clock64();
#endif
            const int pc = p + lane;
            const bool cval = pc < n;
            int rows[DM];
            if (p + NBC <= nst) { // uniform: whole batch inside the staged prefix
#pragma unroll
                for (int kk = 0; kk < DM; ++kk) rows[kk] = crows[pc * DM + kk];
            } else {
                const int v = cval ? (int)order[pc] : 0;
                const int deg = cval ? (int)g.col_deg[v] : 0;
#pragma unroll
                for (int kk = 0; kk < DM; ++kk) rows[kk] = (kk < deg) ? (int)g.vn_row[kk * n + v] : 0xFFFF;
            }
            uint64_t (&red)[WMC] = col; // (the resolver keeps no column of T: its batch vectors take those registers -- a second array of
                                        //  thirty does not fit this kernel's 128 and sent both to scratch)
            if (SWD_WIDE_EVAL_DS && lds_mirror) { // (uniform) ds_read instead of flat loads
                SWD_LDS_AS const uint64_t *TwL = (SWD_LDS_AS const uint64_t *)Tw;
#pragma unroll
                for (int x = 0; x < WMC; ++x) {
                    red[x] = 0ull;
                    if (x < wm) { // uniform
#pragma unroll
                        for (int kk = 0; kk < DM; ++kk) red[x] ^= (rows[kk] == 0xFFFF) ? 0ull : TwL[osd_tidx(rows[kk] == 0xFFFF ? 0 : rows[kk], x, m)];
                    }
                }
            } else {
#pragma unroll
                for (int x = 0; x < WMC; ++x) {
                    red[x] = 0ull;
                    if (x < wm) { // uniform
#pragma unroll
                        for (int kk = 0; kk < DM; ++kk) red[x] ^= (rows[kk] == 0xFFFF) ? 0ull : Tw[osd_tidx(rows[kk] == 0xFFFF ? 0 : rows[kk], x, m)];
                    }
                }
            }
            bool alive = cval;
            int nb = 0, last_cs = NBC - 1; // pivots of this batch, its last pivot column
#ifdef SWD_OSDPROF
            q_eval += clock64() - q0_; q0_ = clock64();
#endif
            while (npiv < rank && nb < ring) {
                uint32_t nz = 0; // (the pivoted-row mask is read where it is used: fifteen more registers do not fit the 128 of this kernel)
#pragma unroll
                for (int x = 0; x < WMC; ++x) {
                    const uint64_t pbx = Pl[x];
                    nz |= (uint32_t)red[x] & ~(uint32_t)pbx;
                    nz |= (uint32_t)(red[x] >> 32) & ~(uint32_t)(pbx >> 32);
                }
                const unsigned long long bal = __ballot(alive && nz != 0u);
                if (bal == 0ull) break; // every remaining column of the batch is dependent
                const int cs = __ffsll((long long)bal) - 1;
                last_cs = cs;
                uint32_t lz = 0;
                asm volatile("" : "+v"(lz)); // the lane number, recomputed here: kept across the loop it is spilled and reloaded per pivot
                const int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, lz));
                SWD_LDS_AS uint64_t *ent = ringS + nb * ES;
                if (ln == cs) { // the pivot column's lane publishes its reduced vector (LDS operations of a wave execute in order)
#pragma unroll
                    for (int x = 0; x < WMC; ++x) ent[x] = red[x];
                }
                asm volatile("" ::: "memory");
                const uint64_t wv = ent[ln < WMC ? ln : 0]; // the vector a word per lane: pivot search
                const uint64_t c = (ln < WMC) ? (wv & ~Pmine) : 0ull; // its ones in unpivoted rows
                const unsigned long long balc = __ballot(c != 0ull);
                const int fx = __ffsll((long long)balc) - 1;
                const int bit = __builtin_amdgcn_readlane(__ffsll((long long)c) - 1, fx);
                racc += __popcll(c); // row additions the reference's LU would apply: unpivoted rows with a one in this column (the pivot itself is taken off at the end)
                if (ln == fx) {
                    __hip_atomic_fetch_and(&ent[fx], ~(1ull << bit), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_or(&Pl[fx], 1ull << bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    Pmine |= 1ull << bit;
                }
                if (ln == 0) ent[WMC] = (uint64_t)(uint32_t)((fx * 64 + bit) | (cs << 16));
                asm volatile("" ::: "memory"); // a wave's LDS operations execute in order: the count follows the entry
                if (ln == 0) ctl[0] = nb + 1;
                // the batch's later columns under the same row operation: y ^= S if y[r] (S without bit r by now: y keeps its own)
                const uint32_t ybit = osd_vec_bit<WMC>(red, fx, bit);
                if (ln <= cs) alive = false;
                else if (alive && ybit) {
#pragma unroll
                    for (int x = 0; x < WMC; ++x) red[x] ^= ent[x];
                }
                ++npiv; ++nb;
            }
            p = (nb >= ring && npiv < rank) ? p0 + last_cs + 1 : p0 + NBC; // (ring full: the columns behind the last pivot are evaluated again)
#ifdef SWD_OSDPROF
            q_res += clock64() - q0_; q0_ = clock64();
#endif
            if (nb > 0 || !(p < n && npiv < rank)) break;
          }
            int rsum = racc;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) rsum += __shfl_xor(rsum, d, 64);
            if (lane == 0) {
                ctl[3] = npiv; ctl[4] = rsum - npiv; ctl[5] = p; ctl[6] = p0;
                ctl[2] = (p < n && npiv < rank) ? 0 : 1;
            }
            asm volatile("" ::: "memory");
            if (lane == 0) ctl[1] = 1;
        } else {
            int done_ops = 0;
            for (;;) {
                const int closed = vctl[1]; // read before the count: a closed batch's count is final
                const int avail = vctl[0];
#ifdef SWD_OSDPROF
                q_wait += clock64() - q0_; q0_ = clock64();
#endif
                while (done_ops < avail) {
                    SWD_LDS_AS const uint64_t *ent = ringS + done_ops * ES;
                    const int r = __builtin_amdgcn_readfirstlane((int)(uint32_t)ent[WMC]) & 0xFFFF;
                    const int rw = r >> 6, rbit = r & 63;
                    if (osd_vec_bit<WMC>(col, rw, rbit)) {
#pragma unroll
                        for (int x = 0; x < WMC; ++x) col[x] ^= ent[x];
                    }
                    ++done_ops;
                }
#ifdef SWD_OSDPROF
                q_app += clock64() - q0_; q0_ = clock64();
#endif
                if (closed) break;
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads(); // the batch's row operations are in every column; ctl[] is final
        const int fin = ctl[2];
        npiv = ctl[3];
        p = ctl[5];
        p0 = ctl[6];
        if (tid >= 64 && tid - 64 < npiv - npiv0) { // the batch's pivots (another wave than the resolver looks the columns up)
            const int e = (int)(uint32_t)ringS[(tid - 64) * ES + WMC];
            piv_col[npiv0 + tid - 64] = order[p0 + (e >> 16)];
            piv_row[npiv0 + tid - 64] = (uint16_t)(e & 0xFFFF);
        }
        if (jc < m && npiv != npiv0) { // the mirror (after the last batch: what the higher-order sweep reads)
            if (lds_mirror) {
                SWD_LDS_AS uint64_t *TwL = (SWD_LDS_AS uint64_t *)Tw;
#pragma unroll
                for (int x = 0; x < WMC; ++x)
                    if (x < wm) TwL[osd_tidx(jc, x, m)] = col[x];
            } else {
#pragma unroll
                for (int x = 0; x < WMC; ++x)
                    if (x < wm) Tw[osd_tidx(jc, x, m)] = col[x];
            }
        }
        if (tid < wm) Sbuf[tid] = 0ull; // (used after the last batch)
        if (tid == 0) { ctl[0] = 0; ctl[1] = 0; } // nobody reads these two between the barriers
        __syncthreads();
#ifdef SWD_OSDPROF
        q_sync += clock64() - q0_;
        if (fin && (tid == 0 || tid == 64) && (blockIdx.x & 63) == 0)
            printf("osdprof colsw thread %d: batches %d pivots %d | resolver: evaluation %lld resolve %lld | column wave: apply %lld wait %lld | rest of the batch %lld cycles\n",
                   tid, q_batches, npiv, q_eval, q_res, q_app, q_wait, q_sync);
#endif
        if (fin) break;
    }
    // y = T * s (s in original row order)
    const bool on = jc < m && synd_b[jc < m ? jc : 0] != 0;
    if (colwave) {
#pragma unroll
        for (int x = 0; x < WMC; ++x) {
            if (x < wm) { // uniform
                uint64_t acc = on ? col[x] : 0ull;
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) acc ^= __shfl_xor(acc, d, 64);
                if (lane == 0 && acc) atomicXor((unsigned long long *)&Sbuf[x], (unsigned long long)acc);
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < npiv; i += NT) {
        const int r = piv_row[i];
        s.hard[piv_col[i]] = (uint8_t)((Sbuf[r >> 6] >> (r & 63)) & 1ull);
    }
    *npiv_out = npiv;
    return ctl[4];
}

// osd0_wave for m <= 256 (wm <= 4): the transform matrix lives in registers -- lane l owns columns
// l, l+64, l+128, l+192 of T, four words each -- and LDS only holds a mirror that the column
// evaluations read.  A step evaluates 16 sorted columns against the mirror and then resolves every
// pivot among them without touching LDS: after a pivot (row r, reduced column u) the row operation
// "rows i != r with u[i] = 1 get row r added" is applied to the owned columns of T and to the reduced
// vectors of the step's later columns (y ^= S if y[r], S = u without bit r), which is exactly what
// re-evaluating them against the updated T would give.  The mirror is refreshed once per step.
// word WS of the calling lane's quad (lane = column * 4 + word)
template <int WS>
__device__ __forceinline__ uint64_t quad_word(uint64_t v) {
    constexpr int ctrl = WS | (WS << 2) | (WS << 4) | (WS << 6); // quad_perm broadcast of lane WS
    const uint32_t lo = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)v, ctrl, 0xF, 0xF, true);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)(v >> 32), ctrl, 0xF, 0xF, true);
    return ((uint64_t)hi << 32) | lo;
}

template <int WS>
__device__ __forceinline__ void osd_treg_update(uint64_t (&t)[4][4], int bit, const uint64_t (&S)[4], uint64_t &red, uint64_t Sw, bool later) {
    // the step's later columns under the same row operation: y ^= S if y[r]
    if (later && ((quad_word<WS>(red) >> bit) & 1ull)) red ^= Sw;
    // 32-bit halves: the selector is one sign-extended bit of the pivot row's word, S is wave-uniform
    uint32_t Sl[4], Sh[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) { Sl[x] = (uint32_t)S[x]; Sh[x] = (uint32_t)(S[x] >> 32); }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t word = (bit < 32) ? (uint32_t)t[q][WS] : (uint32_t)(t[q][WS] >> 32); // wave-uniform choice
        const uint32_t msk = (uint32_t)__builtin_amdgcn_sbfe((int)word, (uint32_t)(bit & 31), 1u);
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const uint32_t lo = (uint32_t)t[q][x] ^ (Sl[x] & msk), hi = (uint32_t)(t[q][x] >> 32) ^ (Sh[x] & msk);
            t[q][x] = ((uint64_t)hi << 32) | lo;
        }
    }
}

template <int DM>
__device__ __forceinline__ int osd0_wave_reg(const SwdGraphDev &g, Lds &s, const uint16_t *order, uint64_t *Tw,
                             uint64_t *Sbuf, uint16_t *piv_col, uint16_t *piv_row, const uint8_t *synd_b,
                             const uint16_t *crows, int nst, int *npiv_out) {
    const int lane = threadIdx.x & 63;
    const int m = g.m, n = g.n, wm = g.wm, rank = g.rank;
    constexpr int NB = 16;              // columns per step, four lanes (words) each
    const int c = lane >> 2, w = lane & 3;
    const bool wact = w < wm;
    const int wl = wact ? w : 0;
    uint64_t t[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int x = 0; x < 4; ++x) t[q][x] = (x == q && q * 64 + lane < m) ? (1ull << lane) : 0ull;
    uint64_t Pw = 0;                    // word w of the pivoted-row mask (replicated per column group)
    int npiv = 0, rowadds = 0, p = 0;
#if SWD_SERIAL_PRIO
    __builtin_amdgcn_s_setprio(SWD_SERIAL_PRIO); // the workgroup's other waves wait for this one: ahead of the other workgroups' waves on its SIMD
#endif
#ifdef SWD_BPPROF
    long long acc_scan = 0, acc_upd = 0, acc_f1 = 0; int nscan = 0;
#endif
    while (p < n && npiv < rank) {
#ifdef SWD_BPPROF
        long long t0 = clock64(); ++nscan;
#endif
        const int pc = p + c;
        const bool cval = wact && pc < n;
        int rows[DM];
        if (p + NB <= nst) { // wave-uniform: whole step inside the staged prefix
#pragma unroll
            for (int k = 0; k < DM; ++k) rows[k] = crows[pc * DM + k];
        } else {             // beyond the staged prefix (rare): straight from the graph
            const int v = cval ? (int)order[pc] : 0;
            const int deg = cval ? (int)g.col_deg[v] : 0;
#pragma unroll
            for (int k = 0; k < DM; ++k) rows[k] = (k < deg) ? (int)g.vn_row[k * n + v] : 0xFFFF;
        }
        uint64_t tw[DM];
#pragma unroll
        for (int k = 0; k < DM; ++k) tw[k] = Tw[osd_tidx(rows[k] == 0xFFFF ? 0 : rows[k], wl, m)];
        uint64_t red = 0;
#pragma unroll
        for (int k = 0; k < DM; ++k) red ^= (rows[k] == 0xFFFF) ? 0ull : tw[k];
        if (!cval) red = 0ull;
        uint64_t cand = red & ~Pw;
        bool found = false;
#ifdef SWD_BPPROF
        long long tA = clock64();
        acc_scan += tA - t0;
#endif
        for (;;) {
            const unsigned long long bal = __ballot(cand != 0ull);
            if (bal == 0ull) break;
            found = true;
            const int fl = __ffsll((long long)bal) - 1; // first column with a usable 1, its lowest word
            const int cs = fl >> 2, ws = fl & 3;
            const int bit = __ffsll((long long)wave_read64(cand, fl)) - 1;
            const int r = ws * 64 + bit;
            {   // row additions the reference's LU would apply: unpivoted rows with a 1 in this column
                uint64_t un = (c == cs) ? cand : 0ull;
                if (lane == fl) un &= ~(1ull << bit);
                rowadds += __popcll(un);
            }
            const uint64_t redc = (lane == fl) ? (red & ~(1ull << bit)) : red;
            uint64_t S[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) S[x] = wave_read64(redc, cs * 4 + x);
            if (w == ws) Pw |= 1ull << bit;
            if (lane == 0) { piv_col[npiv] = (uint16_t)(p + cs); piv_row[npiv] = (uint16_t)r; } // position, translated below
            ++npiv;
            const uint64_t Sw = (w == 0) ? S[0] : (w == 1) ? S[1] : (w == 2) ? S[2] : S[3];
            switch (ws) { // wave-uniform
            case 0: osd_treg_update<0>(t, bit, S, red, Sw, c > cs); break;
            case 1: osd_treg_update<1>(t, bit, S, red, Sw, c > cs); break;
            case 2: osd_treg_update<2>(t, bit, S, red, Sw, c > cs); break;
            default: osd_treg_update<3>(t, bit, S, red, Sw, c > cs); break;
            }
            if (npiv >= rank) break;
            cand = (c > cs) ? (red & ~Pw) : 0ull;
        }
        if (found) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = q * 64 + lane;
#pragma unroll
                for (int x = 0; x < 4; ++x)
                    if (j < m && x < wm) Tw[osd_tidx(j, x, m)] = t[q][x];
            }
            wave_fence();
        }
        p += NB;
#ifdef SWD_BPPROF
        acc_upd += clock64() - tA;
#endif
    }
#ifdef SWD_BPPROF
    if (lane == 0) { s.scal[20] = nscan; s.scal[21] = (int)(acc_scan >> 4); s.scal[22] = (int)(acc_upd >> 4); s.scal[23] = (int)(acc_f1 >> 4); }
#endif
#if SWD_SERIAL_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) rowadds += __shfl_xor(rowadds, d, 64);
    // y = T * s  (s in original row order): XOR of the owned columns the syndrome selects, reduced over the wave
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        uint64_t acc = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = q * 64 + lane;
            if (j < m && synd_b[j]) acc ^= t[q][x];
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) acc ^= __shfl_xor(acc, d, 64);
        if (lane == 0 && x < wm) Sbuf[x] = acc;
    }
    wave_fence();
    for (int i = lane; i < npiv; i += 64) {
        const int r = piv_row[i];
        const uint16_t col = order[piv_col[i]];
        piv_col[i] = col;
        s.hard[col] = (uint8_t)((Sbuf[r >> 6] >> (r & 63)) & 1ull);
    }
    wave_fence();
    *npiv_out = npiv;
    return rowadds;
}

// Round 5, m <= 256 on workgroups of at least four waves: the same elimination with the transform matrix OFF the resolving wave.
// What osd0_wave_reg costs per pivot is its own instruction stream -- ~130 instructions, 40 of them the update of the lane's four
// columns of T, 1500 cycles whatever else runs on the CU (raising the wave's priority changes nothing) -- while the three other
// waves of the workgroup wait at a barrier.  Here wave 0 (the resolver) keeps only what the pivot search needs: the step's 16
// reduced columns, one word per lane, and the mask of unpivoted rows.  Per pivot it publishes (S, r) -- the pivot column's reduced
// vector without bit r -- into a ring in LDS and applies the row operation to the step's later columns; the other waves (the
// followers) hold the columns of T, one or two per lane, and apply the ring's operations at their own pace: col ^= S if col[r].
// An entry is valid when its tag carries the current generation (a wave's LDS operations execute in order: S first, then the
// tag).  A generation ends -- CLOSE entry, the followers' refresh of the mirror in LDS that the evaluations read, one barrier --
// with every step that found a pivot (SWD_QUAD_LAZY = 0) or when SWD_QUAD_LAZY operations are pending: until then the resolver
// evaluates its next 16 columns against the mirror as it stands and applies the pending operations to them itself (y ^= S if
// y[r], ~25 instructions each; no gain measured, see SWD_QUAD_LAZY).
#ifndef SWD_QUAD_LAZY
#define SWD_QUAD_LAZY 0 // (measured on the [[144,12,12]] windows: 8 moves 41 k cycles per elimination out of the closes and 48 k into the evaluations)
#endif
#define SWD_QUAD_RING (SWD_QUAD_LAZY + 18) // entries: pending operations + the up to 16 of a step + the CLOSE entry
#define SWD_QUAD_RING_BYTES (SWD_QUAD_RING * 48 + 32)
typedef uint32_t swd_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t swd_u32x8 __attribute__((ext_vector_type(8)));
// tagpos: pivot row | kind << 8 (1: operation, 2: CLOSE) | generation << 10, sorted position of the pivot column << 32
struct __attribute__((aligned(16))) QuadEnt { uint64_t S[4]; uint64_t tagpos; uint64_t pad; };

template <int NT, int DM>
__device__ __forceinline__ int osd0_quad(const SwdGraphDev &g, Lds &s, const uint16_t *order, uint64_t *Tw, uint64_t *Sbuf,
                                         uint16_t *piv_col, uint16_t *piv_row, const uint8_t *synd_b, const uint16_t *crows, int nst,
                                         int *npiv_out, char *ringmem) {
    static_assert(NT >= 256 && NT % 64 == 0, "wave 0 resolves, the other waves hold the columns of T");
    static_assert(SWD_QUAD_RING <= 64, "the pending pivot rows are kept one per lane");
    constexpr int NB = 16;                 // columns per step, four lanes (words) each
    constexpr int NF = NT / 64 - 1;        // followers
    constexpr int CPL = (NF * 64 >= 256) ? 1 : 2; // columns of T per follower lane
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = g.m, n = g.n, wm = g.wm, rank = g.rank;
    SWD_LDS_AS QuadEnt *ring = (SWD_LDS_AS QuadEnt *)ringmem;
    SWD_LDS_AS int *ctl = (SWD_LDS_AS int *)(ring + SWD_QUAD_RING); // 0: operations of the generation, 1: finished, 3: pivots so far, 4: row additions
    // follower state: the columns jc (and jc2) of T as eight 32-bit words
    const int jc = wave > 0 ? (wave - 1) * 64 + lane : m;
    const int jc2 = (CPL == 2 && wave > 0) ? NF * 64 + jc : m;
    const bool has2 = CPL == 2 && __builtin_amdgcn_readfirstlane((wave > 0 && NF * 64 + (wave - 1) * 64 < m) ? 1 : 0) != 0;
    swd_u32x8 ca, cb;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        ca[d] = (jc < m && d == (jc >> 5)) ? (1u << (jc & 31)) : 0u;
        cb[d] = (jc2 < m && d == (jc2 >> 5)) ? (1u << (jc2 & 31)) : 0u;
    }
    // resolver state
    const int c = lane >> 2, w = lane & 3;
    const bool wact = w < wm;
    const int wl = wact ? w : 0;
    uint64_t U = ~0ull;  // word w of the unpivoted-row mask (replicated per column group)
    int racc = 0;        // ones of the pivot columns in unpivoted rows, pivot included
    int rv = 0;          // lane e: pivot row of the generation's e-th operation
    int npiv = 0, p = 0;
    if (tid < SWD_QUAD_RING) ring[tid].tagpos = 0ull;
    __syncthreads();
#ifdef SWD_BPPROF
    long long q_eval = 0, q_chain = 0, q_sync = 0, q0_ = clock64(); int q_steps = 0;
#endif
    for (int gen = 1;; ++gen) {
#ifdef SWD_BPPROF
        if (gen > 1) q_sync += clock64() - q0_;
#endif
        if (wave == 0) {
            int used = 0;
            bool fin = false;
            for (;;) {
#ifdef SWD_BPPROF
                q0_ = clock64(); ++q_steps;
#endif
                const int pc = p + c;
                const bool cval = wact && pc < n;
                int rows[DM];
                if (p + NB <= nst) { // wave-uniform: whole step inside the staged prefix
#pragma unroll
                    for (int k = 0; k < DM; ++k) rows[k] = crows[pc * DM + k];
                } else {             // beyond the staged prefix (rare): straight from the graph
                    const int v = cval ? (int)order[pc] : 0;
                    const int deg = cval ? (int)g.col_deg[v] : 0;
#pragma unroll
                    for (int k = 0; k < DM; ++k) rows[k] = (k < deg) ? (int)g.vn_row[k * n + v] : 0xFFFF;
                }
                uint64_t tw[DM];
#pragma unroll
                for (int k = 0; k < DM; ++k) tw[k] = Tw[osd_tidx(rows[k] == 0xFFFF ? 0 : rows[k], wl, m)];
                uint64_t red = 0;
#pragma unroll
                for (int k = 0; k < DM; ++k) red ^= (rows[k] == 0xFFFF) ? 0ull : tw[k];
                if (!cval) red = 0ull;
                // the generation's operations so far, which the mirror does not have yet: y ^= S if y[r]
                if (used > 0) {
                    uint64_t Sn = ring[0].S[w];
                    for (int e = 0; e < used; ++e) {
                        const uint64_t Sw = Sn;
                        if (e + 1 < used) Sn = ring[e + 1].S[w];
                        const int r = __builtin_amdgcn_readlane(rv, e), ws = r >> 6, bit = r & 63;
                        const uint32_t half = (bit < 32) ? (uint32_t)red : (uint32_t)(red >> 32);
                        uint32_t yb = (w == ws) ? ((half >> (bit & 31)) & 1u) : 0u;
                        yb |= (uint32_t)__builtin_amdgcn_mov_dpp((int)yb, 0xB1, 0xF, 0xF, true); // quad_perm [1,0,3,2]
                        yb |= (uint32_t)__builtin_amdgcn_mov_dpp((int)yb, 0x4E, 0xF, 0xF, true); // quad_perm [2,3,0,1]
                        if (yb) red ^= Sw;
                    }
                }
                uint64_t cand = red & U;
#ifdef SWD_BPPROF
                q_eval += clock64() - q0_; q0_ = clock64();
#endif
                for (;;) {
                    const unsigned long long bal = __ballot(cand != 0ull);
                    if (bal == 0ull) break;
                    const int fl = __ffsll((long long)bal) - 1; // first column with a usable 1, its lowest word
                    const int cs = fl >> 2, ws = fl & 3;
                    // word w of the pivot column's vector from its quad (asked for first: the answer takes ~100 cycles)
                    const int src = (cs * 4 + w) << 2;
                    const uint32_t slo = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(uint32_t)red);
                    const uint32_t shi = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(uint32_t)(red >> 32));
                    const int bit = __ffsll((long long)wave_read64(cand, fl)) - 1;
                    const int r = ws * 64 + bit;
                    const uint64_t mask = (w == ws) ? (1ull << bit) : 0ull; // bit r in the lanes that hold word ws
                    if (c == cs) { // the pivot column's quad publishes S = its vector without bit r
                        racc += __popcll(cand);
                        SWD_LDS_AS QuadEnt *ent = ring + used;
                        ent->S[w] = red & ~mask;
                        asm volatile("" ::: "memory"); // (a wave's LDS operations execute in order: the tag follows the vector)
                        ent->tagpos = (uint64_t)((uint32_t)r | (1u << 8) | ((uint32_t)gen << 10)) | ((uint64_t)(uint32_t)(p + cs) << 32);
                    }
                    rv = (lane == used) ? r : rv;
                    // bit r of this lane's column: the quad's lane ws has it
                    const uint32_t half = (bit < 32) ? (uint32_t)red : (uint32_t)(red >> 32);
                    uint32_t yb = (w == ws) ? ((half >> (bit & 31)) & 1u) : 0u;
                    yb |= (uint32_t)__builtin_amdgcn_mov_dpp((int)yb, 0xB1, 0xF, 0xF, true);
                    yb |= (uint32_t)__builtin_amdgcn_mov_dpp((int)yb, 0x4E, 0xF, 0xF, true);
                    const bool later = c > cs;
                    // rows i != r with u[i] = 1 get row r added: y ^= S if y[r] (y keeps its bit r)
                    if (later && yb) red ^= (((uint64_t)shi << 32) | slo) & ~mask;
                    U &= ~mask;
                    ++npiv; ++used;
                    if (npiv >= rank) break;
                    cand = later ? (red & U) : 0ull;
                }
                p += NB;
#ifdef SWD_BPPROF
                q_chain += clock64() - q0_;
#endif
                fin = !(p < n && npiv < rank);
                if (used >= (SWD_QUAD_LAZY > 0 ? SWD_QUAD_LAZY : 1) || fin) break;
            }
#ifdef SWD_BPPROF
            q0_ = clock64();
#endif
            int rsum = 0;
            if (fin) {
                rsum = racc;
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) rsum += __shfl_xor(rsum, d, 64);
            }
            if (lane == 0) {
                ctl[0] = used; ctl[1] = fin ? 1 : 0; ctl[3] = npiv; ctl[4] = rsum - npiv;
                asm volatile("" ::: "memory");
                ring[used].tagpos = (uint64_t)((2u << 8) | ((uint32_t)gen << 10)); // CLOSE
            }
        } else {
            int pos = 0;
            for (;;) {
                SWD_LDS_AS QuadEnt *ent = ring + pos;
                const uint32_t tag = (uint32_t)__builtin_amdgcn_readfirstlane((int)*(volatile SWD_LDS_AS uint32_t *)&ent->tagpos);
                asm volatile("" ::: "memory");
                const swd_u32x4 s01 = *(SWD_LDS_AS const swd_u32x4 *)&ent->S[0];
                const swd_u32x4 s23 = *(SWD_LDS_AS const swd_u32x4 *)&ent->S[2];
                asm volatile("" ::: "memory");
                if ((int)(tag >> 10) != gen) { __builtin_amdgcn_s_sleep(1); continue; }
                if (((tag >> 8) & 3u) != 1u) break; // CLOSE
                const int r = (int)(tag & 0xFFu), d = r >> 5, b = r & 31;
                const uint32_t S[8] = {s01[0], s01[1], s01[2], s01[3], s23[0], s23[1], s23[2], s23[3]};
                {
                    const uint32_t msk = (uint32_t)__builtin_amdgcn_sbfe((int)ca[d], (uint32_t)b, 1u);
#pragma unroll
                    for (int x = 0; x < 8; ++x) ca[x] ^= S[x] & msk;
                }
                if (has2) {
                    const uint32_t msk = (uint32_t)__builtin_amdgcn_sbfe((int)cb[d], (uint32_t)b, 1u);
#pragma unroll
                    for (int x = 0; x < 8; ++x) cb[x] ^= S[x] & msk;
                }
                ++pos;
            }
        }
        // A follower that has seen the CLOSE entry has every operation of the generation in its columns and finds ctl[] final
        // (written before the entry); nobody reads the mirror until the resolver's next evaluation, behind the barrier.
        int fin;
        if (wave == 0) {
            fin = ctl[1];
            if (tid < wm) Sbuf[tid] = 0ull; // (used after the last generation)
        } else {
            const int nops = ctl[0], npiv0 = ctl[3] - nops;
            fin = ctl[1];
            npiv = ctl[3];
            if (tid - 64 < nops) { // the generation's pivots (another wave than the resolver looks the columns up)
                const uint64_t e = ring[tid - 64].tagpos;
                piv_col[npiv0 + tid - 64] = order[(int)(e >> 32)];
                piv_row[npiv0 + tid - 64] = (uint16_t)((uint32_t)e & 0xFFu);
            }
            if (nops) { // the mirror (after the last generation: what the higher-order sweep reads)
                if (jc < m) {
#pragma unroll
                    for (int x = 0; x < 4; ++x)
                        if (x < wm) Tw[osd_tidx(jc, x, m)] = ((uint64_t)ca[2 * x + 1] << 32) | ca[2 * x];
                }
                if (jc2 < m) {
#pragma unroll
                    for (int x = 0; x < 4; ++x)
                        if (x < wm) Tw[osd_tidx(jc2, x, m)] = ((uint64_t)cb[2 * x + 1] << 32) | cb[2 * x];
                }
            }
        }
        __syncthreads();
        if (fin) break;
    }
#ifdef SWD_BPPROF // steps, cycles / 16 of the evaluations (pending operations included), of the pivot loops, of the closes (CLOSE entry to the second barrier)
    if (tid == 0) { q_sync += clock64() - q0_; s.scal[20] = q_steps; s.scal[21] = (int)(q_eval >> 4); s.scal[22] = (int)(q_chain >> 4); s.scal[23] = (int)(q_sync >> 4); }
#endif
    // y = T * s (s in original row order)
    if (wave > 0) {
        const bool ona = jc < m && synd_b[jc < m ? jc : 0] != 0, onb = jc2 < m && synd_b[jc2 < m ? jc2 : 0] != 0;
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            if (x < wm) { // uniform
                uint64_t acc = (ona ? (((uint64_t)ca[2 * x + 1] << 32) | ca[2 * x]) : 0ull) ^ (onb ? (((uint64_t)cb[2 * x + 1] << 32) | cb[2 * x]) : 0ull);
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) acc ^= __shfl_xor(acc, d, 64);
                if (lane == 0 && acc) atomicXor((unsigned long long *)&Sbuf[x], (unsigned long long)acc);
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < npiv; i += NT) {
        const int r = piv_row[i];
        s.hard[piv_col[i]] = (uint8_t)((Sbuf[r >> 6] >> (r & 63)) & 1ull);
    }
    *npiv_out = npiv;
    return ctl[4];
}

// Higher-order OSD sweep (osd_window.pyx:242-284): after OSD-0, try the osd_cs candidates (every
// weight-1 pattern on the k = new_n - rank non-pivot columns of the first new_n sorted columns, then
// the weight-2 patterns on the first `order` of them, osd_cs_setup :134-155) or the 2^order osd_e
// patterns (osd_e_setup :128-132); a candidate x gives y = pivot solution of (s ^ Ht x) with x on its
// own columns, scored by pm = sum of llr over y in ascending column order; the first strictly smaller
// pm wins (:276).  With T the accumulated row transform of osd0_wave, the pivot solution of s ^ Ht x
// is y0 ^ XOR_c T*H[:,c]: no second elimination.  One candidate per thread; the ordered sum is a
// merge of the pivots sorted by column with the candidate's own columns.
struct CsBest { double pm; int l; };

__device__ __forceinline__ bool cs_better(double pa, int la, double pb, int lb) {
    return (pa < pb) || (pa == pb && la < lb);
}

// The ordered sum is the expensive part: one term per pivot (216 for the [[144]] windows, 576 for [[288]]) for each of the ~260 /
// ~620 candidates, although a candidate's solution has a few dozen ones.  So the rows of T -- and with them every candidate's y -- are
// first PERMUTED into the order the sum walks: bit i of a permuted word is the pivot variable of rank i among the pivot columns
// sorted by column (each thread permutes its own column of T in place: read it, zero it, OR the bits back at their ranks; rows
// without a pivot are dropped -- they are no variables).  A candidate's sum then visits the SET bits of y in ascending order and adds
// exactly the terms the term-by-term loop added (it added +0.0 for the others: x + 0.0 == x).  rank_of_row: m u16 of scratch.
template <int NT>
__device__ __forceinline__ double osd_sweep(const SwdGraphDev &g, const SwdLdsLayout &L, const SwdDecodeParams &P, Lds &s,
                            const uint16_t *idx, uint64_t *Tc, const uint64_t *y0,
                            const uint16_t *piv_col, const uint16_t *piv_row, int npiv, double pm0, uint16_t *rank_of_row) {
    const int tid = threadIdx.x, n = g.n, wm = g.wm, kset = g.new_n - g.rank, CP = L.cs_par;
    const int order = P.osd_order;
    char *cs = s.scratch + L.off_cs;
    uint16_t *prow_of = (uint16_t *)cs;          // [n]
    uint16_t *Ht = prow_of + n;                  // [kset]
    uint16_t *pcs = Ht + kset;                   // [rank] pivot columns ascending
    uint16_t *prs = pcs + g.rank;                // [rank] their pivot rows
    uint16_t *hts_col = prs + g.rank;            // [16]  osd_e: first `order` Ht columns sorted by column
    uint16_t *hts_bit = hts_col + 16;            // [16]
    double *pllr = (double *)(((uintptr_t)(hts_bit + 16) + 7) & ~(uintptr_t)7); // [rank]
    uint64_t *ybuf = (uint64_t *)(pllr + g.rank); // [wm * CP]

    for (int v = tid; v < n; v += NT) prow_of[v] = 0xFFFF;
    __syncthreads();
    for (int i = tid; i < npiv; i += NT) prow_of[piv_col[i]] = piv_row[i];
    __syncthreads();
    {   // Ht_cols: non-pivot columns among the first new_n sorted positions, in sorted order
        const int ch = (g.new_n + NT - 1) / NT;
        const int p0 = tid * ch, p1 = min(g.new_n, p0 + ch);
        int cnt = 0;
        for (int p = p0; p < p1; ++p) cnt += (prow_of[idx[p]] == 0xFFFF) ? 1 : 0;
        int tot;
        int pos = block_exscan<NT>(cnt, s, tot);
        for (int p = p0; p < p1; ++p)
            if (prow_of[idx[p]] == 0xFFFF) { if (pos < kset) Ht[pos] = idx[p]; ++pos; }
    }
    {   // pivots sorted by column
        const int ch = (n + NT - 1) / NT;
        const int v0 = tid * ch, v1 = min(n, v0 + ch);
        int cnt = 0;
        for (int v = v0; v < v1; ++v) cnt += (prow_of[v] != 0xFFFF) ? 1 : 0;
        int tot;
        int pos = block_exscan<NT>(cnt, s, tot);
        for (int v = v0; v < v1; ++v)
            if (prow_of[v] != 0xFFFF) { pcs[pos] = (uint16_t)v; prs[pos] = prow_of[v]; pllr[pos] = g.llr[v]; ++pos; }
    }
    __syncthreads();
    uint64_t *y0p = ybuf + 48; // y0 in rank order (the candidates no longer use ybuf; its first words serve the arg-min and the winner)
    {
        const int m = g.m;
        for (int r = tid; r < m; r += NT) rank_of_row[r] = 0xFFFF;
        for (int w = tid; w < wm; w += NT) y0p[w] = 0ull;
        __syncthreads();
        for (int i = tid; i < npiv; i += NT) rank_of_row[prs[i]] = (uint16_t)i;
        __syncthreads();
        for (int r = tid; r < m; r += NT) {
            const int rk = rank_of_row[r];
            if (rk != 0xFFFF && ((y0[r >> 6] >> (r & 63)) & 1ull)) atomicOr((unsigned long long *)&y0p[rk >> 6], 1ull << (rk & 63));
        }
        for (int j = tid; j < m; j += NT) { // column j of T: bits by pivot row -> bits by rank, in place (the column is this thread's alone)
            uint64_t col[16];
#pragma unroll
            for (int w = 0; w < 16; ++w) { col[w] = (w < wm) ? Tc[osd_tidx(j, w, m)] : 0ull; }
#pragma unroll
            for (int w = 0; w < 16; ++w) if (w < wm) Tc[osd_tidx(j, w, m)] = 0ull;
#pragma unroll
            for (int w = 0; w < 16; ++w) {
                if (w < wm) { // (uniform)
                    uint64_t x = col[w];
                    while (x) {
                        const int b = __ffsll((long long)x) - 1;
                        x &= x - 1;
                        const int rk = rank_of_row[64 * w + b];
                        if (rk != 0xFFFF) atomicOr((unsigned long long *)&Tc[osd_tidx(j, rk >> 6, m)], 1ull << (rk & 63));
                    }
                }
            }
        }
    }
    __syncthreads();
    const bool exhaustive = (P.osd_method == 1);
    if (exhaustive && tid == 0) { // first `order` Ht columns by ascending column (insertion sort, <= 15)
        for (int i = 0; i < order; ++i) {
            const uint16_t c = Ht[i];
            int j = i;
            while (j > 0 && hts_col[j - 1] > c) { hts_col[j] = hts_col[j - 1]; hts_bit[j] = hts_bit[j - 1]; --j; }
            hts_col[j] = c; hts_bit[j] = (uint16_t)i;
        }
    }
    __syncthreads();
    const long ncand = exhaustive ? (1L << order) : (long)kset + (long)order * (order - 1) / 2;
    double best_pm = pm0;
    int best_l = -1;
    const int INF = 0x7fffffff;
    for (long base = 0; base < ncand; base += CP) {
        const long l = base + tid;
        if (tid < CP && l < ncand) {
            double pm = 0.0;
            if (!exhaustive) {
                // osd_cs: one or two extra columns.  Their row lists and priors are fetched once (not per word of y), the ordered
                // sum runs in three stretches of the pivots (sorted by column) with the candidate's columns added in between.
                int c1, c2 = INF;
                if (l < kset) c1 = Ht[l];
                else {
                    int rem = (int)(l - kset), i = 0;
                    while (rem >= order - 1 - i) { rem -= order - 1 - i; ++i; }
                    c1 = Ht[i]; c2 = Ht[i + 1 + rem];
                }
                const int d1 = g.col_deg[c1], d2 = (c2 != INF) ? (int)g.col_deg[c2] : 0;
                int rw[2 * SWD_DMAX];
#pragma unroll
                for (int k = 0; k < SWD_DMAX; ++k) {
                    rw[k] = (k < d1) ? (int)g.vn_row[k * n + c1] : -1;
                    rw[SWD_DMAX + k] = (k < d2) ? (int)g.vn_row[k * n + c2] : -1;
                }
                const int cA = min(c1, c2), cB = max(c1, c2); // cB = INF for a single column
                const double llrA = g.llr[cA], llrB = (cB != INF) ? g.llr[cB] : 0.0;
                // number of pivot columns below cA / cB (pcs ascending): the candidate's own columns enter the sum there
                int posA = 0, posB = npiv;
                { int lo = 0, hi = npiv; while (lo < hi) { const int mid = (lo + hi) >> 1; if ((int)pcs[mid] < cA) lo = mid + 1; else hi = mid; } posA = lo; }
                if (cB != INF) { int lo = posA, hi = npiv; while (lo < hi) { const int mid = (lo + hi) >> 1; if ((int)pcs[mid] < cB) lo = mid + 1; else hi = mid; } posB = lo; }
                bool a_done = false, b_done = (cB == INF);
                for (int w = 0; w < wm; ++w) {
                    uint64_t yw = y0p[w];
#pragma unroll
                    for (int k = 0; k < 2 * SWD_DMAX; ++k)
                        if (rw[k] >= 0) yw ^= Tc[osd_tidx(rw[k], w, g.m)];
                    while (yw) { // the set pivot variables of this word, in column order
                        const int i = 64 * w + __ffsll((long long)yw) - 1;
                        yw &= yw - 1;
                        if (!a_done && i >= posA) { pm += llrA; a_done = true; }
                        if (!b_done && i >= posB) { pm += llrB; b_done = true; }
                        pm += pllr[i];
                    }
                }
                if (!a_done) pm += llrA;
                if (!b_done) pm += llrB;
            } else {
                int q = 0;
                for (int w = 0; w < wm; ++w) {
                    uint64_t yw = y0p[w];
                    for (int i = 0; i < order; ++i)
                        if ((l >> i) & 1) {
                            const int c = Ht[i], dc = g.col_deg[c];
                            for (int k = 0; k < dc; ++k) yw ^= Tc[osd_tidx((int)g.vn_row[k * n + c], w, g.m)];
                        }
                    while (yw) {
                        const int i = 64 * w + __ffsll((long long)yw) - 1;
                        yw &= yw - 1;
                        const int pc = pcs[i];
                        while (q < order && (int)hts_col[q] < pc) { // the pattern's own columns that come before this pivot column
                            if ((l >> hts_bit[q]) & 1) pm += g.llr[hts_col[q]];
                            ++q;
                        }
                        pm += pllr[i];
                    }
                }
                while (q < order) {
                    if ((l >> hts_bit[q]) & 1) pm += g.llr[hts_col[q]];
                    ++q;
                }
            }
            if (pm < best_pm) { best_pm = pm; best_l = (int)l; }
        }
    }
    // block arg-min; ties go to the earliest candidate, OSD-0 itself (l = -1) first
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const double opm = __shfl_xor(best_pm, d, 64);
        const int ol = __shfl_xor(best_l, d, 64);
        if (cs_better(opm, ol, best_pm, best_l)) { best_pm = opm; best_l = ol; }
    }
    double *wpm = (double *)ybuf; // candidates are done with ybuf
    int *wl = (int *)(wpm + 16);
    __syncthreads();
    if ((tid & 63) == 0) { wpm[tid >> 6] = best_pm; wl[tid >> 6] = best_l; }
    __syncthreads();
    for (int w = 0; w < NT / 64; ++w)
        if (cs_better(wpm[w], wl[w], best_pm, best_l)) { best_pm = wpm[w]; best_l = wl[w]; }
    __syncthreads();
    if (best_l < 0) return pm0;
    // rebuild the winner into s.hard
    const long l = best_l;
    for (int v = tid; v < n; v += NT) s.hard[v] = 0;
    for (int w = tid; w < wm; w += NT) {
        uint64_t yw = y0p[w];
        if (!exhaustive) {
            int c1, c2 = INF;
            if (l < kset) c1 = Ht[l];
            else {
                int rem = (int)(l - kset), i = 0;
                while (rem >= order - 1 - i) { rem -= order - 1 - i; ++i; }
                c1 = Ht[i]; c2 = Ht[i + 1 + rem];
            }
            for (int k = 0; k < g.col_deg[c1]; ++k) yw ^= Tc[osd_tidx((int)g.vn_row[k * n + c1], w, g.m)];
            if (c2 != INF)
                for (int k = 0; k < g.col_deg[c2]; ++k) yw ^= Tc[osd_tidx((int)g.vn_row[k * n + c2], w, g.m)];
        } else {
            for (int i = 0; i < order; ++i)
                if ((l >> i) & 1) {
                    const int c = Ht[i];
                    for (int k = 0; k < g.col_deg[c]; ++k) yw ^= Tc[osd_tidx((int)g.vn_row[k * n + c], w, g.m)];
                }
        }
        ybuf[32 + w] = yw; // wpm/wl occupy the first 24 words
    }
    __syncthreads();
    for (int i = tid; i < npiv; i += NT) s.hard[pcs[i]] = (uint8_t)((ybuf[32 + (i >> 6)] >> (i & 63)) & 1ull); // (bit i = the pivot variable of rank i)
    __syncthreads();
    if (tid == 0) {
        if (!exhaustive) {
            if (l < kset) s.hard[Ht[l]] = 1;
            else {
                int rem = (int)(l - kset), i = 0;
                while (rem >= order - 1 - i) { rem -= order - 1 - i; ++i; }
                s.hard[Ht[i]] = 1; s.hard[Ht[i + 1 + rem]] = 1;
            }
        } else {
            for (int i = 0; i < order; ++i)
                if ((l >> i) & 1) s.hard[Ht[i]] = 1;
        }
    }
    __syncthreads();
    return best_pm;
}

// OSD on the ordered matrix (osd_window.pyx:215-284 / bp4_osd.pyx:297-366): the caller has filled the
// sort keys (key[v], idx[v] = v, padding ~0 / 0xFFFF) at the start of the scratch region.  Sorts, runs the
// OSD-0 elimination on wave 0, then the higher-order sweep.  On return s.hard[0..n) holds the OSD
// solution; the return value is its path metric (sum of g.llr over the solution in column order).
// QUAD: the scratch region is LDS (osd0_quad addresses its ring there)
// WIDE: large-graph kernel (osd0_colsw when the layout has its LDS block)
#ifdef SWD_OSD_NOINLINE // experiment (round 6): the OSD of the shots that need one as a function of its own
#define SWD_OSD_FN __device__ __attribute__((noinline))
#else
#define SWD_OSD_FN __device__ __forceinline__
#endif
template <int NT, int DM, bool COLFORM = false, bool QUAD = COLFORM, bool WIDE = false> // COLFORM: the caller's kernel can afford osd0_cols (osd_window kernels of 1024 threads)
SWD_OSD_FN double osd_run(const SwdGraphDev &g, const SwdLdsLayout &L, const SwdDecodeParams &P, Lds &s,
                                          const uint8_t *synd, uint8_t *osd0_b, int &rowadds, long long &t_sorted,
                                          long long &t_elim, bool presorted = false) {
    const int tid = threadIdx.x, m = g.m, n = g.n;
    uint64_t *key = (uint64_t *)s.scratch;
    uint16_t *idx = (uint16_t *)(s.scratch + L.off_idx);
    if (!presorted) sort_pairs<NT>(key, idx, L.npad);
    t_sorted = wall_clock64();
    uint64_t *Tc = (uint64_t *)s.aux;
    uint64_t *Sbuf = Tc + m * g.wm;
    uint16_t *piv_col = (uint16_t *)(Sbuf + g.wm);
    uint16_t *piv_row = piv_col + g.rank;
    uint16_t *list1 = (uint16_t *)(s.scratch + L.off_aux + (size_t)(m * g.wm + g.wm) * 8 + (size_t)g.rank * 4); // (its place in the scratch region, wherever s.aux points)
    for (int i = tid; i < m * g.wm; i += NT) { // word-major identity: Tw[w*m + j]
        const int w = i / m, j = i - w * m;
        Tc[i] = (w == (j >> 6)) ? (1ull << (j & 63)) : 0ull;
    }
    for (int v = tid; v < n; v += NT) s.hard[v] = 0;
    // stage the row lists of the leading sorted columns over the (dead) sort keys
    uint16_t *crows = (uint16_t *)s.scratch;
    bool cols = false; // column-form elimination: 256 < m <= 576 on the 1024-thread variants
    // (not in the column-weight-10 variant: its windows -- SHYPS twelve-round, 252 checks -- never need it, and the mere presence
    // of the routine in that kernel cost its BP loops registers: 0.61 -> 0.51 M windows/s)
    constexpr bool kColForm = COLFORM && NT >= 1024 && DM <= 8;
    if constexpr (kColForm) cols = g.wm > 4 && g.wm <= 9 && L.off_oslot >= 0;
    const int nst = min(n, (cols ? L.off_oslot : L.npad * 8) / (DM * 2));
    for (int p = tid; p < nst; p += NT) {
        const int v = idx[p];
        const int deg = g.col_deg[v];
#pragma unroll
        for (int k = 0; k < DM; ++k) crows[p * DM + k] = (k < deg) ? g.vn_row[k * n + v] : (uint16_t)0xFFFF;
    }
    __syncthreads();
    constexpr bool kQuad = QUAD && NT >= 256 && SWD_OSD_QUAD;
    bool quad = false;
    if constexpr (kQuad) quad = g.wm <= 4 && L.off_oring >= 0;
    if (quad) {
        if constexpr (kQuad) {
            int npiv = 0;
            const int ra = osd0_quad<NT, DM>(g, s, idx, Tc, Sbuf, piv_col, piv_row, synd, crows, nst, &npiv, s.scratch + L.off_oring);
            if (tid == 0) { s.scal[2] = ra; s.scal[3] = npiv; }
        }
    } else if (g.wm <= 4) {
        if (tid < 64) {
            int npiv;
            const int ra = osd0_wave_reg<DM>(g, s, idx, Tc, Sbuf, piv_col, piv_row, synd, crows, nst, &npiv);
            if (tid == 0) { s.scal[2] = ra; s.scal[3] = npiv; }
        }
    } else {
        int npiv = 0, ra = 0;
        bool done = false;
        if constexpr (kColForm) {
            if (cols) { ra = osd0_cols<NT, DM, 9>(g, s, idx, Tc, Sbuf, piv_col, piv_row, synd, crows, nst, &npiv, s.scratch + L.off_oslot); done = true; }
        }
        if constexpr (WIDE && NT == 1024 && DM <= 8 && SWD_OSD_WIDE) {
            if (g.wm <= 15 && L.off_owide >= 0) {
                ra = osd0_colsw<NT, DM, 15>(g, s, idx, Tc, Sbuf, piv_col, piv_row, synd, crows, nst, &npiv, (char *)s.hard - L.off_hard + L.off_owide, L.owide_ring, L.osd_lds != 0);
                done = true;
            }
        }
        if (!done) ra = osd0_block<NT, DM>(g, s, idx, Tc, Sbuf, piv_col, piv_row, synd, crows, nst, &npiv);
        if (tid == 0) { s.scal[2] = ra; s.scal[3] = npiv; }
    }
    __syncthreads();
    rowadds = s.scal[2];
    const int npiv = s.scal[3];
    t_elim = wall_clock64();
    if (osd0_b)
        for (int v = tid; v < n; v += NT) osd0_b[v] = s.hard[v];
    double pm = ordered_pm<NT>(g, s, list1);
    if (P.osd_order > 0) pm = osd_sweep<NT>(g, L, P, s, idx, Tc, Sbuf, piv_col, piv_row, npiv, pm, list1); // (list1: dead after ordered_pm)
    return pm;
}

struct WinResult {
    int exit_class, conv, total_it, pre_it, post_it, live_vn, live_cn, live_e, osd_rowadds;
    double pm;
    long long t[9]; // phase boundaries (wall_clock64 ticks): init, pre, sort, shorten, post, osd sort, elim, sweep
#ifdef SWD_GDGPROF
    long long gp[5]; // guessing decoders, diagnostic build: cache rebuilds, BP blocks, select_vn, branch starts, path metrics
#endif
};

// scratch: where the scratch region of this workgroup starts -- the LDS block itself, or (BIG kernels) its region in HBM, in which
// case the remaining offsets of the layout are LDS offsets as they stand
__device__ __forceinline__ void lds_bind(Lds &s, char *smem, const SwdLdsLayout &L, char *scratch) {
    s.scratch = scratch;
    s.aux = scratch + L.off_aux;
    s.msg = (double *)scratch;
    s.livemask = (uint64_t *)(smem + L.off_livemask);
    s.par = (uint32_t *)(smem + L.off_par);
    s.lv = (uint16_t *)(smem + L.off_lv);
    s.jptr = (uint16_t *)(smem + L.off_jptr);
    s.lslot = (uint16_t *)((L.off_lslot > 0 ? smem : scratch) + (L.off_lslot >= 0 ? L.off_lslot : 0)); // 0: staged at the start of the scratch region
    s.cn_val = (int8_t *)(smem + L.off_cnval);
    s.cn_deg = (uint8_t *)(smem + L.off_cndeg);
    s.cn_deg0 = (uint8_t *)(smem + L.off_cndeg0);
    s.vn_val = (int8_t *)(smem + L.off_vnval);
    s.hard = (uint8_t *)(smem + L.off_hard);
    s.flags = (int *)(smem + L.off_misc);
    s.scal = s.flags + 32;
    s.dbl = (double *)(s.scal + 32);
    s.iaux = (int *)(s.dbl + 24);
}

// osd_window.decode (osd_window.pyx:158-199) for one syndrome `synd` (LDS bytes, original check
// order).  On return s.hard[0..n) is the vector decode() returns.
// vraw (kernels that split the cache load): where the raw edge words and priors of the variable nodes go; pre_valid: the caller has
// issued the loads for this graph already
template <int NT, int VF, int DM, int KG, bool SF, bool HACC, bool BIG = false>
__device__ __forceinline__ void decode_window(const SwdGraphDev &g, const SwdLdsLayout &L, const SwdDecodeParams &P, Lds &s,
                              const uint8_t *synd, double *hist_b, uint8_t *osd0_b, uint8_t *bpdec_b, WinResult &R, const uint32_t *cn_map,
                              VnRaw<NT, (NT <= SWD_TUNED_NT) ? VF : 1, DM> &vraw, bool pre_valid = false) {
    constexpr bool DIET = SWD_P16(NT); // the tuned kernels' LDS forms (decided-node bits, 48-bit live masks, no copy of the check degrees)
    if constexpr (DIET) s.lm_m = (L.off_par - L.off_livemask < 8 * g.m) ? g.m : 0; // m if the masks are stored in the 48-bit form
    // Tuned kernels: the parity WORDS of the iterations lie over the live masks, which nothing reads while bp_run is running (the full-
    // graph caches never look at them, the shortening step rebuilds them from scratch, the shortened graph's caches are loaded before
    // its iterations start and the OSD does not use them): (m + 1) x 4 <= m x 6 bytes.  Round 4 kept one parity BYTE per check to fit
    // three workgroups per CU, which cost ~5 address / mask instructions per flip in the hottest loop (a flip is now one add + ds_xor).
    if constexpr (DIET) s.par = (uint32_t *)s.livemask;
    const int tid = threadIdx.x;
    const int m = g.m, n = g.n;
#pragma unroll
    for (int i = 0; i < 9; ++i) R.t[i] = 0;
    R.t[0] = wall_clock64();

    // kernels of up to 256 threads: the cache's loads are in flight during the reset loops (headline 9.95 -> 9.84 ms per launch at
    // order 0; the 1024-thread kernels lose by it -- [[288]] 63.2 -> 64.0 ms -- and load where they always did)
    constexpr bool kSplitLoad = NT <= SWD_TUNED_NT;
    // the full-graph phase in the graph's listed order, tiered variable-node pass (bit 0: the LDS-resident kernels, bit 1: the large-graph ones)
    constexpr bool kFullSorted = (SWD_FULL_SORTED & (BIG ? 2 : 1)) != 0;
    [[maybe_unused]] int kcf[VF];
    // (memory loads return in order: what the reset loops need from the graph is asked for BEFORE the cache's 7 x DM + 7 loads, so
    // that the loops run while those are in flight instead of behind them)
    [[maybe_unused]] int d_first = 0, p_first = 0, j_first = 0;
    if constexpr (kSplitLoad) {
        if (tid < m) { d_first = g.row_deg[tid]; p_first = g.perm[tid]; }
        if (tid <= g.K) j_first = g.jptr[tid];
        __builtin_amdgcn_sched_barrier(0); // (issued here)
        if (!pre_valid) vn_cache_issue<NT, VF, DM, kFullSorted>(g, s, vraw); // (uniform)
    }
    // reset (osd_window.pyx:288-303)
    for (int l = tid; l < m; l += NT) {
        const int d = (kSplitLoad && l == tid) ? d_first : (int)g.row_deg[l];
        s.cn_val[l] = (int8_t)(synd[(kSplitLoad && l == tid) ? p_first : (int)g.perm[l]] ? 1 : 0);
        s.cn_deg[l] = (uint8_t)d;
        if constexpr (!DIET) s.cn_deg0[l] = (uint8_t)d;
        lm_set<DIET>(s, l, (d >= 64) ? ~0ull : ((1ull << d) - 1ull));
    }
    vn_reset<NT, DIET>(s, n);
    for (int j = tid; j <= g.K; j += NT) s.jptr[j] = (kSplitLoad && j == tid) ? (uint16_t)j_first : g.jptr[j];
    if (P.zero_hist)
        for (int i = tid; i < 4 * n; i += NT) hist_b[i] = 0.0;
    // the tuned kernels use the packed caches and their overloads of the BP routines: byte offsets + parity bytes up to 256
    // threads, slot numbers + parity words in the 1024-thread kernels
    std::conditional_t<SWD_P16(NT), VnCacheP<VF, DM, 0, false>, VnCacheP<VF, DM, 3, false>> vc;
#ifdef SWD_INITPROF
    const long long ip0 = wall_clock64();
#endif
    // large graphs: check-to-bit messages as one record per check in LDS (bp_run<..., REC>) when the layout's LDS block has room for them
    constexpr bool kTbl = BIG && SWD_BIG_TBL && !kFullSorted && !kSplitLoad; // the full graph's variable-node pass from the graph's tables (bp_run<..., TBL>)
    constexpr bool kRec = BIG && SWD_BIG_REC && !kTbl && !kFullSorted && !kSplitLoad;
    bool rec_on = false;
    if constexpr (kRec) rec_on = L.off_pmsg >= 0 && m <= 1023 && L.pmsg_bytes >= (m + 1) * 32 + 4096;
    if constexpr (kSplitLoad) {
        vn_cache_pack<NT, VF, DM>(g, s, vraw, vc);
        if constexpr (kFullSorted) vn_row_caps<NT, VF, DM>(g, s, vraw.ev, kcf);
    } else if constexpr (kTbl) { // (no cache)
    } else {
        if (kRec && rec_on) { if constexpr (kRec) vn_cache_load<NT, VF, DM, true, false, false, true>(g, s, n, vc); }
        else vn_cache_load<NT, VF, DM, true, false, kFullSorted>(g, s, n, vc, nullptr, kFullSorted ? &kcf : nullptr);
    }
#ifdef SWD_INITPROF
    asm volatile("" : "+v"(vc.edp[VF - 1][0]), "+v"(vc.llr[0]));
    const long long ip1 = wall_clock64();
#endif
    // the check state and jptr written above are read below by OTHER threads (a check is served by the thread
    // whose ctid equals its lane number, which need not be the thread that initialised it)
    __syncthreads();
#ifdef SWD_INITPROF
    const long long ip2 = wall_clock64();
#endif
    // large graphs: the LDS region that will hold the shortened graph's messages / the OSD arrays is idle in this phase -- the TOP of the
    // message array (the far and zero slots included) lives there, one select per access (flat addresses reach both memories)
    constexpr bool kHyb = BIG && SWD_BIG_HYBRID;
    const int rec_b = (kRec && rec_on) ? (((m + 1) * 32 + 15) & ~15) : 0;
    if constexpr (kRec) {
        if (rec_on) {
            s.rec = (SWD_AS3 char *)((char *)s.hard - L.off_hard + L.off_pmsg);
            if (tid < 8) ((SWD_AS3 uint32_t *)(s.rec + m * 32))[tid] = 0u; // the record dead positions point at: + 0.0
        }
    }
    if constexpr (kHyb) {
        const int cells = g.E + 1 + 2 * (NT / 64);
        const int cap = (L.off_pmsg >= 0) ? (L.pmsg_bytes - rec_b) / 8 : 0;
        const int lo = max(cells - cap, 0);
        s.msg_lo = (cap > 0) ? (uint32_t)lo << 3 : 0xFFFFFFFFu;
        s.msg_alt = (char *)s.hard - L.off_hard + (L.off_pmsg >= 0 ? L.off_pmsg + rec_b : 0);
    }
    if constexpr (kTbl) bp_init_tbl<NT, VF, DM, kHyb>(g, s);
    else bp_init<VF, DM, kHyb>(s, vc);
#ifdef SWD_INITPROF
    const long long ip3 = wall_clock64();
#endif
    std::conditional_t<SWD_P16(NT), CnCacheP<KG>, CnCacheP<KG, 3>> cn;
    if constexpr (SF) { // heavy checks are shared by 2 or 4 threads in the full-graph phase too (host-built map)
        const uint32_t e = cn_map[s.ctid];
        cn_cache_load<NT, KG, true>(g, s, false, (e & 0xFFFFu) == 0xFFFFu ? -1 : (int)(e & 0xFFFFu), (int)((e >> 16) & 3u), (int)(e >> 18), cn);
    } else {
        cn_cache_load<NT, KG, true>(g, s, false, s.ctid < m ? s.ctid : -1, 0, 1, cn);
    }
    __syncthreads();

    int it = 0;
    R.conv = 0; R.pm = 0.0; R.pre_it = R.post_it = 0;
    R.live_vn = n; R.live_cn = m; R.live_e = g.E; R.osd_rowadds = 0;
    uint16_t *list0 = (uint16_t *)s.scratch;
    R.t[1] = wall_clock64();
#ifdef SWD_INITPROF // diagnostic build: where the set-up of a window goes (reset loops | cache loads | barrier | bp_init | check caches)
    if (tid == 0) { s.scal[20] = (int)(ip0 - R.t[0]); s.scal[21] = (int)(ip1 - ip0); s.scal[22] = (int)(ip2 - ip1); s.scal[23] = (int)(ip3 - ip2); s.scal[24] = (int)(R.t[1] - ip3); }
#endif

    double hs[VF]; // HACC: summed posterior history of this thread's variable nodes
#pragma unroll
    for (int i = 0; i < VF; ++i) hs[i] = 0.0;
    if constexpr (kTbl) R.conv = bp_run<NT, VF, DM, KG, true, SF, HACC, false, false, kHyb, false, true>(g, P, s, P.pre_iter, n, vc, cn, hist_b, it, P.alpha, false, hs);
    else
    if (kRec && rec_on) {
        if constexpr (kRec) R.conv = bp_run<NT, VF, DM, KG, true, SF, HACC, false, false, kHyb, true>(g, P, s, P.pre_iter, n, vc, cn, hist_b, it, P.alpha, false, hs);
    } else
    R.conv = bp_run<NT, VF, DM, KG, true, SF, HACC, false, kFullSorted, kHyb>(g, P, s, P.pre_iter, n, vc, cn, hist_b, it, P.alpha, false, hs, nullptr, false,
                                                                        kFullSorted ? kcf : nullptr);
    R.pre_it = it;
    R.t[2] = wall_clock64();
    if (R.conv) {
        R.exit_class = SWD_EXIT_PRE;
        R.pm = ordered_pm<NT>(g, s, list0);
        R.total_it = R.pre_it;
        return;
    }
    // ---- order columns by summed LLR history (osd_window.pyx:172-176); only the set cols[new_n:]
    // matters unless a decimation fails, so the common path selects instead of sorting
    uint64_t *key = (uint64_t *)s.scratch;
    uint16_t *idx = (uint16_t *)(s.scratch + L.off_idx);
    __syncthreads();
    if constexpr (HACC) {
#pragma unroll
        for (int i = 0; i < VF; ++i) { // the keys come straight from the accumulators of the owning thread
            const int li = s.vtid + i * NT;
            if (li < n) { const int v = kFullSorted ? (int)g.vperm[li] : li; key[v] = f2key(hs[i]); idx[v] = (uint16_t)v; }
        }
        for (int v = n + tid; v < L.npad; v += NT) { key[v] = ~0ull; idx[v] = 0xFFFF; }
    } else {
        for (int v = tid; v < L.npad; v += NT) {
            if (v < n) {
                const double sum = ((hist_b[v] + hist_b[n + v]) + hist_b[2 * n + v]) + hist_b[3 * n + v];
                key[v] = f2key(sum);
                idx[v] = (uint16_t)v;
            } else { key[v] = ~0ull; idx[v] = 0xFFFF; }
        }
    }
    // columns of the edges, by slot: the shortening step walks them several times
    uint16_t *rc = (uint16_t *)(s.scratch + L.off_rc);
    for (int e = tid; e < g.E; e += NT) rc[e] = g.row_col[e];
    __syncthreads();
    // ---- shortening: decide cols[new_n:] = 0 (osd_window.pyx:178-183)
    if (g.new_n < n) {
        // BIG kernels: s.aux is HBM there, and the select's eight passes of histogram atomics cost 187 us per decode of the 936 x 8784
        // model.  The live masks (8 m bytes of LDS) are rewritten by the shortening step below and are not read before it: the three
        // histograms go there (47 us)
        if constexpr (BIG) select_smallest_call<NT>(key, n, g.new_n, (m * 8 >= 3 * 256 * 4) ? (int *)s.livemask : (int *)s.aux, s);
        else select_smallest<NT>(key, n, g.new_n, (int *)s.aux, s);
    }
    R.t[3] = wall_clock64();
    bool contra = false;
    for (int l = tid; l < m; l += NT) {
        const int d = g.row_deg[l];
        uint64_t mk = 0;
        int cntl = 0;
        for (int j = 0; j < d; ++j) {
            const int v = rc[s.jptr[j] + l];
            if (!vn_decided<DIET>(s, v)) { mk |= 1ull << j; ++cntl; }
        }
        lm_set<DIET>(s, l, mk);
        s.cn_deg[l] = (uint8_t)cntl;
        if (cntl == 0) {
            if (s.cn_val[l] != 0) contra = true;
            else s.cn_val[l] = -1;
        }
    }
    const bool any_contra = block_any<NT>(contra, s);
#ifdef SWD_SHPROF
    long long sh0 = wall_clock64();
#endif
    if (any_contra) {
        // "setting vn failed" (osd_window.pyx:179-181): the reference stops at the first decimation
        // that empties an unsatisfied check; only VNs up to that sorted position were zeroed.
        sort_pairs<NT>(key, idx, L.npad);
        uint16_t *pos = (uint16_t *)s.aux;
        for (int i = tid; i < n; i += NT) pos[idx[i]] = (uint16_t)i;
        if (tid == 0) s.scal[0] = 0x7fffffff;
        __syncthreads();
        for (int l = tid; l < m; l += NT) {
            if (s.cn_deg[l] == 0 && s.cn_val[l] > 0) {
                int mx = 0;
                const int d = g.row_deg[l];
                for (int j = 0; j < d; ++j) mx = max(mx, (int)pos[g.row_col[s.jptr[j] + l]]);
                atomicMin(&s.scal[0], mx);
            }
        }
        __syncthreads();
        const int kstop = s.scal[0];
        for (int i = g.new_n + tid; i <= kstop && i < n; i += NT) s.hard[idx[i]] = 0;
        __syncthreads();
        R.exit_class = SWD_EXIT_FAIL_SET;
        R.total_it = R.pre_it;
        return;
    }
    for (int v = tid; v < n; v += NT)
        if (vn_decided<DIET>(s, v)) s.hard[v] = 0; // (every decided node so far was decided 0)
    __syncthreads();
    // ---- peel (osd_window.pyx:184-186).  Degree-1 checks force their last VN; the closure of these
    // forced values does not depend on the order they are applied in, and a contradiction shows up
    // in every order.  So all degree-1 checks fire at once, round by round; only when a contradiction
    // appears is the saved state restored and the reference's sweep order replayed by one wave (the
    // partial result the reference leaves behind depends on that order).
    {
        char *bak = s.scratch + L.off_bak;
        vn_backup<NT, DIET>(s, n, bak);
        for (int i = tid; i < m; i += NT) {
            bak[2 * n + i] = (char)s.cn_val[i]; bak[2 * n + m + i] = (char)s.cn_deg[i];
            ((uint64_t *)(bak + ((2 * n + 2 * m + 7) & ~7)))[i] = lm_get<DIET>(s, i);
        }
        bool bad = false;
        for (;;) {
            bool fired = false;
            for (int l = tid; l < m; l += NT) {
                if (s.cn_val[l] >= 0 && s.cn_deg[l] == 1) {
                    const int j = __ffsll((long long)lm_get<DIET>(s, l)) - 1;
                    const int v = rc[s.jptr[j] + l];
                    const int8_t val = s.cn_val[l];
                    vn_decide<DIET>(s, v, val);
                    fired = true;
                }
            }
            if (!block_any<NT>(fired, s)) break;
            for (int l = tid; l < m; l += NT) {
                int cv = s.cn_val[l];
                if (cv < 0) continue;
                uint64_t mk = lm_get<DIET>(s, l), left = mk;
                int deg = s.cn_deg[l];
                while (left) {
                    const int j = __ffsll((long long)left) - 1;
                    left &= left - 1;
                    const int vv = vn_value<DIET>(s, rc[s.jptr[j] + l]);
                    if (vv >= 0) { mk &= ~(1ull << j); --deg; cv ^= vv; }
                }
                if (deg == 0) { if (cv != 0) bad = true; cv = -1; }
                lm_set<DIET>(s, l, mk); s.cn_deg[l] = (uint8_t)deg; s.cn_val[l] = (int8_t)cv;
            }
            if (block_any<NT>(bad, s)) { bad = true; break; }
        }
        if (bad) {
            vn_restore<NT, DIET>(s, n, bak);
            for (int i = tid; i < m; i += NT) {
                s.cn_val[i] = (int8_t)bak[2 * n + i]; s.cn_deg[i] = (uint8_t)bak[2 * n + m + i];
                lm_set<DIET>(s, i, ((uint64_t *)(bak + ((2 * n + 2 * m + 7) & ~7)))[i]);
            }
            __syncthreads();
            if (tid < 64) {
                const bool b2 = peel_wave<DIET, NT>(g, s);
                if (tid == 0) s.scal[1] = b2 ? 1 : 0;
            }
        } else if (tid == 0) s.scal[1] = 0;
        if (tid == 0) { s.scal[2] = 0; s.scal[3] = 0; }
    }
    __syncthreads();
#ifdef SWD_SHPROF
    long long sh1 = wall_clock64();
#endif
    if (s.scal[1]) {
        R.exit_class = SWD_EXIT_FAIL_PEEL;
        R.total_it = R.pre_it;
        return;
    }
    // ---- compact the live VNs, re-initialise their messages (osd_window.pyx:187)
    const bool uselist = L.off_lslot >= 0;
    int nlive;
    // live checks are dealt to the threads in order of decreasing live degree (counting sort): a wave's
    // CN pass costs its largest degree, and after shortening the degrees are very uneven
    int *dhist = (int *)(s.scratch + L.off_cord);      // [65]
    uint16_t *cord = (uint16_t *)(dhist + 66);          // [m]
    // Sorted form of the shortened graph (tuned kernels, SWD_POST_SORTED): the live nodes are listed by decreasing live degree -- in
    // tiers of two positions, index order inside a tier -- so that the nodes a wave serves in one cache row need about the same number
    // of positions; the live degrees are counted from the check side (one LDS atomic per live edge) into a byte per node that lies
    // over the dead backup of the peeling step.
    // (SWD_POST_SORTED bit 0: the tuned kernels of up to 256 threads, bit 1: the LDS-resident 1024-thread kernels, bit 2: the large-graph
    // kernels, whose shortened graph lives in the LDS region post_lds names)
    constexpr bool kSorted = (SWD_POST_SORTED & (BIG ? 4 : (SWD_P16(NT) ? 1 : (SWD_OSDW_TUNED && NT >= 512 ? 2 : 0)))) != 0 && VF > 2;
    bool sorted = false;
    if constexpr (kSorted) sorted = L.post_lds != 0 && uselist && g.new_n <= 2 * NT;
    uint32_t *deg32 = (uint32_t *)(s.scratch + L.off_bak); // [n] bytes, four per word
    for (int i = tid; i < 65; i += NT) dhist[i] = 0;
    if constexpr (kSorted) {
        if (sorted)
            for (int i = tid; i < (n + 3) / 4; i += NT) deg32[i] = 0u;
    }
    __syncthreads(); // the histogram (and the degree bytes) are zero before the first atomic below
    {
        int lc = 0, le = 0;
        for (int l = tid; l < m; l += NT)
            if (s.cn_val[l] >= 0) {
                ++lc;
                le += __popcll(lm_get<DIET>(s, l));
                atomicAdd(&dhist[min((int)s.cn_deg[l], 64)], 1);
                // compact list of the live edge slots of this check (post-phase CN pass)
                uint64_t mk = uselist ? lm_get<DIET>(s, l) : 0ull;
                int k = 0;
                while (mk) {
                    const int j = __ffsll((long long)mk) - 1;
                    mk &= mk - 1;
                    const int slot = s.jptr[j] + l;
                    s.lslot[k * m + l] = (uint16_t)slot;
                    if constexpr (kSorted) {
                        if (sorted) { const int v = rc[slot]; atomicAdd(&deg32[v >> 2], 1u << (8 * (v & 3))); }
                    }
                    ++k;
                }
            }
        if (lc) { atomicAdd(&s.scal[2], lc); atomicAdd(&s.scal[3], le); }
        const int ch = (n + NT - 1) / NT;
        const int v0 = tid * ch, v1 = min(n, v0 + ch);
        bool plain = true;
        if constexpr (kSorted) {
            if (sorted) {
                plain = false;
                __syncthreads(); // the degrees are complete
                // tiers of SWD_TIER_STEP(DM) positions: tier t = 1 .. NTIER holds the nodes whose live degree rounds up to t x step; the tier
                // counts share prefix sums, FB bits each (a tier holds at most new_n <= 2 NT nodes), FPI of them per int
                constexpr int TS = SWD_TIER_STEP(DM), NTIER = (DM + TS - 1) / TS;
                constexpr int FB = (2 * NT < 1024) ? 10 : 12, FPI = (2 * NT < 1024) ? 3 : 2, FM = (1 << FB) - 1, NI = (NTIER + FPI - 1) / FPI;
                const uint8_t *deg8 = (const uint8_t *)deg32;
                auto tier_q = [&](int v) { const int d = deg8[v]; const int t = min(max((d + TS - 1) / TS, 1), NTIER); return NTIER - t; }; // 0 = the heaviest tier
                int cnt[NI], pos[NI], tot[NI];
#pragma unroll
                for (int j = 0; j < NI; ++j) cnt[j] = 0;
                for (int v = v0; v < v1; ++v)
                    if (!vn_decided<DIET>(s, v)) {
                        const int q = tier_q(v);
#pragma unroll
                        for (int j = 0; j < NI; ++j) cnt[j] += (q / FPI == j) ? (1 << (FB * (q % FPI))) : 0;
                    }
#pragma unroll
                for (int j = 0; j < NI; ++j) pos[j] = block_exscan<NT>(cnt[j], s, tot[j]);
                int base[NTIER], run = 0;
#pragma unroll
                for (int q = 0; q < NTIER; ++q) { base[q] = run; run += (tot[q / FPI] >> (FB * (q % FPI))) & FM; }
                nlive = run;
                for (int v = v0; v < v1; ++v)
                    if (!vn_decided<DIET>(s, v)) {
                        const int q = tier_q(v);
                        int p = 0;
#pragma unroll
                        for (int qq = 0; qq < NTIER; ++qq)
                            if (qq == q) { p = base[qq] + ((pos[qq / FPI] >> (FB * (qq % FPI))) & FM); pos[qq / FPI] += 1 << (FB * (qq % FPI)); }
                        s.lv[p] = (uint16_t)v;
                    }
            }
        }
        if (plain) {
            int cnt = 0;
            for (int v = v0; v < v1; ++v) cnt += vn_decided<DIET>(s, v) ? 0 : 1;
            int pos = block_exscan<NT>(cnt, s, nlive);
            for (int v = v0; v < v1; ++v)
                if (!vn_decided<DIET>(s, v)) s.lv[pos++] = (uint16_t)v;
        }
    }
    __syncthreads();
#ifdef SWD_SHPROF
    long long sh2 = wall_clock64();
#endif
    R.live_vn = nlive; R.live_cn = s.scal[2]; R.live_e = s.scal[3];
    int clc, csub, cgrp;
    cn_assign<NT, KG>(g, s, dhist, cord, uselist, true, R.live_cn, clc, csub, cgrp);
    // BIG kernels: the shortened graph's messages move into LDS (layout flag post_lds), renumbered one column of cells per live
    // variable node; the staged column table of the shortening step (dead now) becomes the old-slot -> cell table
    // (tuned kernels of up to 256 threads, experiment SWD_POST_RENUM: the same renumbering inside their LDS scratch region --
    // the variable-node pass of the shortened graph then touches consecutive cells instead of scattered slots)
    SwdGraphDev gp = g;
    bool renum = false;
    constexpr bool kRenum = BIG || (SWD_POST_RENUM && SWD_P16(NT));
    if constexpr (kRenum) renum = L.post_lds != 0 && uselist;
    // the post phase proper, for a register cache of any depth: caches, (re)initialised messages, the iterations
    // (renum_tag: compile-time twin of `renum`, so that a BIG kernel's post-phase message pointer is an LDS pointer on every path
    // that reaches the iterations -- ds_read / ds_write instead of flat accesses)
    auto run_post = [&](auto &vcx, auto &cnx, double *hsx, auto vfx, auto kgx, auto renum_tag, auto sorted_tag) {
        constexpr int VFX = decltype(vfx)::value, KGX = decltype(kgx)::value;
        if constexpr (decltype(sorted_tag)::value) { // sorted form: compacted caches, renumbered cells, tiered variable-node pass
            int kc[VFX];
            uint16_t *remap = rc;
            gp.E = g.D * nlive;
            if constexpr (BIG) s.msg = (double *)(s.hard - L.off_hard + L.off_pmsg); // the LDS block starts off_hard bytes below s.hard
            vn_cache_load_compact<NT, VFX, DM>(gp, s, nlive, vcx, remap, kc);
            __syncthreads();
            for (int l = tid; l < m; l += NT)
                if (s.cn_val[l] >= 0) {
                    const int d = s.cn_deg[l];
                    for (int k = 0; k < d; ++k) s.lslot[k * m + l] = remap[s.lslot[k * m + l]];
                }
            __syncthreads();
            cn_cache_load<NT, KGX, false>(gp, s, uselist, clc, csub, cgrp, cnx);
            __syncthreads(); // every lane has read its slot list before the messages are re-initialised
            bp_init<VFX, DM>(s, vcx);
            __syncthreads();
            R.t[4] = wall_clock64();
            return bp_run<NT, VFX, DM, KGX, false, false, HACC, false, true>(gp, P, s, P.post_iter, nlive, vcx, cnx, hist_b, it, P.alpha, false, hsx, nullptr, false, kc);
        } else
        if constexpr (kRenum && decltype(renum_tag)::value) {
            {
                uint16_t *remap = rc;
                gp.E = g.D * nlive;
                if constexpr (BIG) s.msg = (double *)(s.hard - L.off_hard + L.off_pmsg); // the LDS block starts off_hard bytes below s.hard
                vn_cache_load<NT, VFX, DM, false>(gp, s, nlive, vcx, remap);
                __syncthreads();
                for (int l = tid; l < m; l += NT)
                    if (s.cn_val[l] >= 0) {
                        const int d = s.cn_deg[l];
                        for (int k = 0; k < d; ++k) s.lslot[k * m + l] = remap[s.lslot[k * m + l]];
                    }
                __syncthreads();
            }
        } else vn_cache_load<NT, VFX, DM, false>(gp, s, nlive, vcx);
        cn_cache_load<NT, KGX, false>(gp, s, uselist, clc, csub, cgrp, cnx);
        __syncthreads(); // every lane has read its slot list before the messages are re-initialised
        bp_init<VFX, DM>(s, vcx);
        __syncthreads();
        R.t[4] = wall_clock64();
        return bp_run<NT, VFX, DM, KGX, false, false, HACC>(gp, P, s, P.post_iter, nlive, vcx, cnx, hist_b, it, P.alpha, false, hsx);
    };
    // The shortened graph needs less of both register caches than the full graph: at most new_n (normally <= 2 NT) live nodes,
    // and -- heavy checks being shared by up to four threads -- at most T positions per thread, T = what cn_assign just chose
    // (s.iaux[0]).  With depth 2 / KGP groups of four positions the iteration loop keeps 16 + 2 KGP registers of cache instead of
    // 8 VF + 2 KG, its check pass is unrolled KGP times instead of KG times, and no reload sits in it.
    constexpr int KGP = (SWD_POST_KGP > 0 && SWD_POST_KGP < KG) ? SWD_POST_KGP : KG;
    const bool small_ok = g.new_n <= 2 * NT && (KGP == KG || s.iaux[0] <= 4 * KGP);
    using VC2 = std::conditional_t<SWD_P16(NT), VnCacheP<2, DM, 0, false>, VnCacheP<2, DM, 3, false>>;
    using CC2 = std::conditional_t<SWD_P16(NT), CnCacheP<KGP>, CnCacheP<KGP, 3>>;
    // (BIG kernels: the host sets post_lds only for new_n <= 2 NT, so the renumbered form always runs at depth 2)
    bool post_done = false;
    if constexpr (kSorted) {
        if (sorted && small_ok) {
            VC2 vc2; CC2 cn2;
            double hs2[2] = {0.0, 0.0};
            R.conv = run_post(vc2, cn2, hs2, std::integral_constant<int, 2>{}, std::integral_constant<int, KGP>{}, std::false_type{}, std::true_type{});
            hs[0] = hs2[0]; hs[1] = hs2[1];
            post_done = true;
        }
    }
    if (post_done) {
    } else
    if constexpr (BIG && VF > 2) {
        if (renum && small_ok) {
            VC2 vc2; CC2 cn2;
            double hs2[2] = {0.0, 0.0};
            R.conv = run_post(vc2, cn2, hs2, std::integral_constant<int, 2>{}, std::integral_constant<int, KGP>{}, std::true_type{}, std::false_type{});
            hs[0] = hs2[0]; hs[1] = hs2[1];
            post_done = true;
        }
    } else if constexpr (kRenum) {
        if (renum) { R.conv = run_post(vc, cn, hs, std::integral_constant<int, VF>{}, std::integral_constant<int, KG>{}, std::true_type{}, std::false_type{}); post_done = true; }
    } else if constexpr ((SWD_POST_DEPTH2 & (NT >= 1024 ? 1 : 2)) != 0 && VF > 2) {
        // the LDS-resident kernels: bit 0 = the 1024-thread ones ([[288]] (4,1): 61.9 -> 60.5 ms per launch), bit 1 = those of up to
        // 256 threads (headline, order 0: 10.25 -> 9.7 ms per launch); both on in the production build
        if (small_ok) {
            VC2 vc2; CC2 cn2;
            double hs2[2] = {0.0, 0.0};
            R.conv = run_post(vc2, cn2, hs2, std::integral_constant<int, 2>{}, std::integral_constant<int, KGP>{}, std::false_type{}, std::false_type{});
            hs[0] = hs2[0]; hs[1] = hs2[1];
            post_done = true;
        }
    }
    if (!post_done) R.conv = run_post(vc, cn, hs, std::integral_constant<int, VF>{}, std::integral_constant<int, KG>{}, std::false_type{}, std::false_type{});
#ifdef SWD_SHPROF
    if (tid == 0) { s.scal[20] = (int)(sh0 - R.t[3]); s.scal[21] = (int)(sh1 - sh0); s.scal[22] = (int)(sh2 - sh1); s.scal[23] = (int)(R.t[4] - sh2); }
#endif
    if constexpr (BIG) {
        s.msg = (double *)s.scratch;
        if (L.osd_lds) s.aux = (char *)s.hard - L.off_hard + L.off_pmsg; // the arrays of the OSD phase go where the post-phase messages were
    }
    R.post_it = it;
    R.t[5] = wall_clock64();
    R.total_it = R.pre_it + R.post_it;
    if (R.conv) {
        R.exit_class = SWD_EXIT_POST;
        R.pm = ordered_pm<NT>(g, s, list0);
        return;
    }
    if (P.osd_order < 0) { R.exit_class = SWD_EXIT_NO_OSD; return; }
    double *hsl = (double *)(s.aux + (L.off_hs - L.off_aux)); // HACC: history sums of the live VNs by column (the messages are dead now)
    if constexpr (HACC) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < VF; ++i) {
            const int li = s.vtid + i * NT;
            if (li < nlive) hsl[s.lv[li]] = hs[i];
        }
    }
    // osd_window.bp_decoding after an OSD exit = the post-phase BP decisions incl. the decided values (osd_window.pyx:499-501)
    if (bpdec_b)
        for (int v = tid; v < n; v += NT) bpdec_b[v] = s.hard[v];
    // ---- OSD (osd_window.pyx:201-284): keys -1000 / +1000 / history sum, stable ascending order.
    // The decided-0 columns (most of the window after shortening) all carry +1000 and therefore form one
    // block in index order; only the rest (decided-1 and live columns, <= new_n) needs sorting:
    //   order = [rest with key < 1000, sorted] ++ [decided-0 by index] ++ [rest with key > 1000, sorted].
    // A live column whose sum is exactly 1000.0 would interleave with the block by index: then the full sort runs.
    __syncthreads();
    bool presorted = false;
    {
        const uint64_t k1000 = f2key(1000.0);
        uint16_t *zlist = (uint16_t *)s.aux; // [n], free until the transform matrix is set up
        const int ch = (n + NT - 1) / NT;
        const int v0 = tid * ch, v1 = min(n, v0 + ch);
        int cnt = 0; // rest count | zero count << 16
        for (int v = v0; v < v1; ++v) cnt += (vn_value<DIET>(s, v) == 0) ? 0x10000 : 1;
        int tot;
        const int pos = block_exscan<NT>(cnt, s, tot);
        int ps = pos & 0xFFFF, pz = pos >> 16, nlt = 0;
        bool eq = false;
        for (int v = v0; v < v1; ++v) {
            const int vv = vn_value<DIET>(s, v);
            if (vv == 0) { zlist[pz++] = (uint16_t)v; continue; }
            const uint64_t k = f2key(vv == 1 ? -1000.0 : (HACC ? hsl[v] : ((hist_b[v] + hist_b[n + v]) + hist_b[2 * n + v]) + hist_b[3 * n + v]));
            key[ps] = k; idx[ps] = (uint16_t)v; ++ps;
            nlt += (k < k1000) ? 1 : 0;
            eq |= (k == k1000);
        }
        const int totS = tot & 0xFFFF, totZ = tot >> 16;
        int npadS = 2;
        while (npadS < totS) npadS <<= 1;
        for (int i = totS + tid; i < npadS; i += NT) { key[i] = ~0ull; idx[i] = 0xFFFF; }
        if (tid == 0) s.iaux[3] = 0;
        const bool any_eq = block_any<NT>(eq, s);
        if (!any_eq) {
            if (nlt) atomicAdd(&s.iaux[3], nlt);
            sort_pairs<NT>(key, idx, npadS);
            const int c_lt = s.iaux[3];
            // the (rare) tail of the rest with keys > 1000 moves behind the decided-0 block
            uint16_t tail[VF];
#pragma unroll
            for (int k = 0; k < VF; ++k) { const int i = c_lt + tid + k * NT; tail[k] = (i < totS) ? idx[i] : (uint16_t)0; }
            __syncthreads();
            for (int i = tid; i < totZ; i += NT) idx[c_lt + i] = zlist[i];
#pragma unroll
            for (int k = 0; k < VF; ++k) { const int i = c_lt + tid + k * NT; if (i < totS) idx[totZ + i] = tail[k]; }
            __syncthreads();
            presorted = true;
        }
    }
    if (!presorted) {
        for (int v = tid; v < L.npad; v += NT) {
            if (v < n) {
                double sum;
                const int vv = vn_value<DIET>(s, v);
                if (vv == 1) sum = -1000.0;
                else if (vv == 0) sum = 1000.0;
                else sum = HACC ? hsl[v] : ((hist_b[v] + hist_b[n + v]) + hist_b[2 * n + v]) + hist_b[3 * n + v];
                key[v] = f2key(sum);
                idx[v] = (uint16_t)v;
            } else { key[v] = ~0ull; idx[v] = 0xFFFF; }
        }
        __syncthreads();
    }
    R.pm = osd_run<NT, DM, !BIG, !BIG, BIG>(g, L, P, s, synd, osd0_b, R.osd_rowadds, R.t[6], R.t[7], presorted); // (the column-form elimination addresses LDS explicitly;
    // behind a function boundary -- its own register allocation -- the headline launch took 12.3 instead of 10.5 ms: round 4)
    R.exit_class = SWD_EXIT_OSD;
}

#include "swd_gdg_kernel.h"

// The window loop of the reference harness (/root/reference/osd.py:130-179) as work units on a persistent
// grid: one unit = one window of one shot -- decode window t on the shot's residual syndrome, commit the
// leading `commit` columns into total_e_hat, fold the committed faults back into the residual syndrome
// (osd.py:178, done sparsely on the LDS copy), hand the residual syndrome + observable accumulator to the
// unit of window t+1 through HBM.  W = 1 with commit = 0 is the plain batched osd_window.decode.
//
// Invariants of the hand-over (regression: scripts/stress_handoff.py, tests/test_gpu_scheduler.py):
//   1. units are drawn in ticket order, window-major; unit (t, b) waits only for unit (t-1, b), whose ticket
//      is older, so its workgroup is already resident and running: the wait cannot deadlock under any
//      dispatch order and needs no co-residency beyond the grid the launch itself sized;
//   2. producer: every state word is written with an agent-scope atomic store (sc1, write-through past the
//      XCD's L2), every wave waits for the acknowledgement of its stores (s_waitcnt vmcnt(0), in an asm
//      statement the compiler cannot drop or move), the workgroup barrier orders all waves' acknowledgements
//      before thread 0's agent-scope store of the progress counter;
//   3. consumer: thread 0 polls the counter with agent-scope atomic loads (sc1: served past L1), a workgroup
//      barrier, then every state word is read with an agent-scope atomic load issued after the poll returned
//      (a wave's vector-memory loads return in order), so no L1- or L2-resident stale line can be read;
//   4. state words are never accessed with plain loads or stores, and scratch that one unit writes and reads
//      back (history ring, snapshots) is indexed by workgroup, never shared between units.
// On top of that thread 0 issues an agent-scope release fence before the counter store and an agent-scope acquire
// fence after its poll, which makes the hand-over correct by the memory model alone (fence-fence synchronisation
// through the relaxed counter, workgroup barriers on both sides, every state access an agent-scope atomic): measured
// +0.4 % per launch (11.82 vs 11.78 ms).  -DSWD_HANDOFF_NOFENCE builds the variant that relies only on items 2-3
// (gfx950's sc1 accesses; the guide lists "sc1 payload -> asm vmcnt(0) -> sc1 flag" as a valid form).
// A wait that exceeds its 10 s bound sets bit 0 of *status, records exit class SWD_EXIT_SCHED_FAULT for the unit
// and commits nothing for it: the caller sees the fault (swd_pipeline_status) instead of a plausible wrong answer.
// BIG (large graphs, osd_window only): the scratch region of the window's layout -- fp64 messages, sort keys, OSD arrays --
// is the workgroup's region of a.big in HBM (served by L2 / the memory-side cache); LDS keeps the per-check and per-node state.
// VFP (guessing decoders): depth of the register cache for the shortened graph, 2 or VF (swd_gdg_kernel.h)
// Order in which a launch starts its shots: by decreasing syndrome weight (counting sort, ties in any order).  Decoding time
// grows with the weight, and what a persistent grid loses at the end of a launch is the time its last-started shots still
// need: the light ones go last.  A shot's result does not depend on when it is decoded.
// wt_order: [2 B] words -- weights (scratch), then the order.
template <int NT>
__global__ void __launch_bounds__(NT) shot_weight_kernel(const uint8_t *det, int64_t det_stride, int num_det, int B, uint32_t *wt) {
    const int b = (int)blockIdx.x * (NT / 64) + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    const uint8_t *d = det + (int64_t)b * det_stride;
    int c = 0;
    for (int r = lane; r < num_det; r += 64) c += d[r] ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if (lane == 0) wt[b] = (uint32_t)min(c, 1023);
}
template <int NT>
__global__ void __launch_bounds__(NT) shot_order_kernel(const uint32_t *wt, int B, uint32_t *order) {
    static_assert(NT == 1024, "one bin per thread");
    __shared__ uint32_t bin[2][1024];
    const int t = threadIdx.x;
    bin[0][t] = 0;
    __syncthreads();
    // (eight weights asked for before the first is used: one memory round trip per eight entries instead of one per entry -- a 65 536-entry
    //  order, the quaternary decoder's batches, took 40 us in this kernel, 64 dependent loads per thread in each of its two passes)
    for (int b0 = t; b0 < B; b0 += 8 * NT) {
        uint32_t w8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) w8[u] = (b0 + u * NT < B) ? wt[b0 + u * NT] : 0xFFFFFFFFu;
#pragma unroll
        for (int u = 0; u < 8; ++u) if (w8[u] != 0xFFFFFFFFu) atomicAdd(&bin[0][1023 - w8[u]], 1u);
    }
    __syncthreads();
    const uint32_t own = bin[0][t];
    int cur = 0;
    for (int o = 1; o < 1024; o <<= 1) { // inclusive scan
        bin[cur ^ 1][t] = bin[cur][t] + (t >= o ? bin[cur][t - o] : 0u);
        cur ^= 1;
        __syncthreads();
    }
    const uint32_t excl = bin[cur][t] - own;
    __syncthreads();
    bin[0][t] = excl;
    __syncthreads();
    for (int b0 = t; b0 < B; b0 += 8 * NT) {
        uint32_t w8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) w8[u] = (b0 + u * NT < B) ? wt[b0 + u * NT] : 0xFFFFFFFFu;
#pragma unroll
        for (int u = 0; u < 8; ++u) if (w8[u] != 0xFFFFFFFFu) order[atomicAdd(&bin[0][1023 - w8[u]], 1u)] = (uint32_t)(b0 + u * NT);
    }
}

template <int NT, int VF, int DM, int KG, int KIND, bool SF = false, bool BIG = false, int VFP = VF>
__global__ void __launch_bounds__(NT, (NT >= 1024 ? 4 : (NT == 640 ? 3 : ((SWD_OSDW_TUNED && NT == 512 && SWD_TUNED_NT >= 512 && (KIND == 0 || KIND == 3)) ? 6 : ((SWD_OSDW_TUNED && NT == 256 && KG <= 12 && (KIND == 0 || KIND == 3)) ? 3 : 2))))) pipeline_kernel(const SwdPipeArgs a) {
    static_assert(!BIG || KIND == 0 || KIND == 3 || KIND == 1 || KIND == 7, "the HBM-resident scratch region exists for the osd_window kernels and for the guessing decoders' serial walk / ticket-scheduled ensemble");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    // One workgroup decodes ONE window of one shot.  Units are handed out by an atomic ticket in
    // window-major order (all shots' window 0, then window 1, ...): a shot's windows are sequential
    // (commit -> residual syndrome), and shot times are heavy-tailed (a shot with several OSD windows
    // takes 5x the mean), so whole shots per workgroup left a quarter of the launch to stragglers;
    // level by level every shot advances together and the tail is one window long.  Unit t depends on
    // unit t - B, whose ticket was drawn earlier by a workgroup that is already running, so the wait
    // below cannot deadlock whatever order the hardware dispatches workgroups in.
    uint32_t *acc = (uint32_t *)(smem + a.off_det) - 4; // 16 bytes below sdet are reserved by the host
    // The grid is only as large as the device can hold at once; every workgroup keeps drawing tickets
    // (starting a fresh workgroup per unit left a fifth of the slots empty at any time).
    const uint32_t nunits = (uint32_t)a.B * (uint32_t)a.W;
    Lds s;
    swd_wave_roles<NT>(s, acc + 2);
#ifdef SWD_RESIDENCY // diagnostics: how many workgroups of this launch are resident at the same time? (status words 4: now, 5: most, 6: with >= 1 unit)
    if (tid == 0) { const uint32_t now = atomicAdd(&a.status[4], 1u) + 1u; atomicMax(&a.status[5], now); }
    uint32_t res_units = 0;
#endif
    bool queued = false; // parallel form of the guessing decoders: work items instead of window-major tickets
    constexpr bool kQueue = KIND == 2 || KIND == 7; // (kind 7, the threaded ensemble: UNIT items only -- its units run from 50 us to several ms, and a
                                                    //  workgroup that draws a ticket whose predecessor is such a unit would idle for that long)
    if constexpr (kQueue) queued = a.gdgp.q != nullptr;
    // parallel form: every shot is admitted through the counter a.sched[0] -- by a workgroup that finds nothing unclaimed in the
    // ring, or by the one that finishes a shot -- so progress never depends on a workgroup that is not resident yet
    constexpr uint32_t shots0 = 0;
    // The tuned osd_window kernels ask for the next unit's variable-node cache (7 x DM edge words + the priors per thread, from the
    // graph's tables in L2) BEFORE they draw the ticket: consecutive tickets name the same window -- or a window that shares its
    // graph -- nearly always, and the ticket, the predecessor's counter and the state record are three more dependent round trips to
    // memory during which those loads can be in flight.  A unit of another graph asks again (decode_window).
    constexpr bool kSpecLoad = SWD_SPEC_LOAD && (KIND == 0 || KIND == 3) && !BIG && NT <= SWD_TUNED_NT && !((SWD_FULL_SORTED & 1) != 0);
    [[maybe_unused]] VnRaw<NT, kSpecLoad ? VF : 1, DM> vspec;
    [[maybe_unused]] int wi_spec = 0;
    [[maybe_unused]] const uint32_t *spec_tab = nullptr;
    for (;;) {
    const long long t_unit0 = wall_clock64();
    int wi, b, final_ctx = -1;
    if constexpr (kSpecLoad) {
        const SwdGraphDev &gs = a.wins[wi_spec].g;
        vn_cache_issue<NT, VF, DM, false>(gs, s, vspec);
        spec_tab = gs.vn_edge;
        __builtin_amdgcn_sched_barrier(0);
    }
    [[maybe_unused]] bool ens_task = false;      // kind 7: this turn runs a tree thread of a parked ensemble, not a unit
    [[maybe_unused]] uint32_t ens_item = 0;
#ifdef SWD_GDG_DEBUG
    long long dbg_ta = 0, dbg_tb = 0, dbg_tc = 0;
#endif
    if (queued) {
        // Work items (swd_gdg_kernel.h) from one FIFO ring: windows of admitted shots (a window is queued when its
        // predecessor has committed), side branches and results of parked decimation trees.  No item waits for
        // another one, so a hard tree never blocks a workgroup that could do other work; the number of shots in
        // flight is bounded so that a continuation never queues behind the whole batch.
        uint32_t item = 0;
        if constexpr (kQueue) {
            __syncthreads();
            if (tid == 0) {
                // nothing unclaimed in the ring: admit one more shot instead of waiting (the number of shots under way grows
                // to what keeps the grid busy and no further -- every queued item waits behind the whole ring)
                if ((int32_t)(ag_ld(&a.gdgp.q[1]) - ag_ld(&a.gdgp.q[0])) <= 0) {
                    const uint32_t nb = shots0 + atomicAdd(a.sched, 1u);
                    if (nb < (uint32_t)a.B) ring_push(a.gdgp.q, a.gdgp.qmask, item_unit(a.order ? (int)a.order[nb] : (int)nb, 0));
                }
                acc[2] = ring_pop_wait(a.gdgp.q, a.gdgp.qmask, a.status, &a.sched[a.B + 1]); acc[3] = 0u;
            }
            __syncthreads();
            item = acc[2];
        }
#ifdef SWD_GDG_DEBUG
        asm volatile("" ::: "memory"); dbg_ta = wall_clock64(); asm volatile("" ::: "memory");
        if (tid == 0) { uint32_t *dbg_status = a.gdgp.chk_status; GDG_COUNT(7, 1); GDG_COUNT(8, dbg_ta - t_unit0); if (dbg_ta - t_unit0 > 5000) { GDG_COUNT(6, 1); GDG_COUNT(15, dbg_ta - t_unit0); } }
#endif
        if (item == SWD_ITEM_EXIT) break;
        const uint32_t type = item >> 30;
        if (type == SWD_ITEM_SIDE) {
            if constexpr (KIND == 2) gdg_run_task<NT, VF, DM, KG, VFP>(a, smem, item, s.ctid, s.vtid);
            if constexpr (KIND != 7) continue;
            // (kind 7: a tree thread of a parked ensemble -- it meets the units at the one place where the walk is inlined, below)
            ens_task = true; ens_item = item;
            wi = (int)ag_ld(&((const uint32_t *)(a.gdgp.ctx + (int64_t)((item >> 8) & 0x3FFFFFu) * a.gdgp.ctx_stride))[4]); b = 0;
        }
        else if (type == SWD_ITEM_FINAL) {
            final_ctx = (int)(item & 0x3FFFFFFFu);
            uint32_t *h = (uint32_t *)(a.gdgp.ctx + (int64_t)final_ctx * a.gdgp.ctx_stride);
            wi = (int)__hip_atomic_load(&h[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            b = (int)__hip_atomic_load(&h[5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else { wi = (int)((item >> 22) & 0xFFu); b = (int)(item & 0x3FFFFFu); }
    } else {
    __syncthreads();
    if (tid == 0) {
        const uint32_t t = atomicAdd(a.sched, 1u);
        acc[2] = t; acc[3] = 0u;
        if (a.order && t < nunits) acc[1] = a.order[t % (uint32_t)a.B]; // (every round of tickets walks the shots in the same order)
    }
    __syncthreads();
    const uint32_t ticket = acc[2];
    if (ticket >= nunits) break;
    wi = (int)(ticket / (uint32_t)a.B); b = a.order ? (int)acc[1] : (int)(ticket % (uint32_t)a.B);
    }
    // (wi / b as explicit scalars -- readfirstlane -- were tried: the kernel already spills 267 SGPRs, no gain)
    __syncthreads();
    // LDS holds the residual syndrome of the window's own rows only (whole words: [dbase, dbase + dlen)); the rest of
    // the shot's residual syndrome stays in the state record in HBM
    uint8_t *sdet = (uint8_t *)(smem + a.off_det);
    uint8_t *state_b = a.state + (int64_t)b * a.state_stride;
    constexpr bool SLICE = SWD_P16(NT) && (KIND == 0 || KIND == 3); // (the other kernels keep the whole residual syndrome in LDS: dbase = 0)
    const int dbase = SLICE ? (a.wins[wi].row0 & ~3) : 0;
    const int dlen = (SLICE ? min((a.wins[wi].row0 + a.wins[wi].g.m + 3) & ~3, (a.num_det + 3) & ~3) : ((a.num_det + 3) & ~3)) - dbase;
    if (ens_task) { // (a task of a parked ensemble has no syndrome of its own: its state comes from the context)
    } else if (wi == 0) {
        const uint8_t *det_b = a.det + (int64_t)b * a.det_stride;
        if constexpr (SLICE) {
            for (int r = tid; r < dlen; r += NT) sdet[r] = (dbase + r < a.num_det && det_b[dbase + r]) ? 1 : 0;
        } else {
            for (int r = tid; r < a.num_det; r += NT) sdet[r] = det_b[r] ? 1 : 0;
        }
        if (SLICE && a.W > 1) { // the rows of the later windows go into the state record right away
            uint32_t *st32 = (uint32_t *)state_b;
            for (int q = tid; q < (a.num_det + 3) / 4; q += NT) {
                if (4 * q >= dbase && 4 * q < dbase + dlen) continue;
                uint32_t x = 0;
                for (int k = 0; k < 4; ++k) x |= (4 * q + k < a.num_det && det_b[4 * q + k]) ? (1u << (8 * k)) : 0u;
                __hip_atomic_store(&st32[4 + q], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (tid == 0) { acc[0] = 0; acc[1] = 0; }
    } else {
        // Hand-over without cache maintenance: the state words and the progress counter are written and read
        // with agent-scope atomic accesses (coherent per access across the XCDs' L2s); a release / acquire
        // FENCE at agent scope would write back / invalidate a whole L2 per window and costs more than the
        // window's own tail.  Order: data stores complete (workgroup fence = wait for their acknowledgement),
        // then the counter; the reader sees the counter, then loads the data the same way.
        // (work items of the parallel guessing-decoder form are only queued once their predecessor has committed)
        if (!queued && tid == 0) {
            const long long t_wait0 = wall_clock64();
            while (__hip_atomic_load(&a.sched[1 + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (uint32_t)wi) {
                __builtin_amdgcn_s_sleep(8);
                // cannot happen by construction (the awaited ticket is older and running); a bound keeps a bug
                // from hanging the device: flag the launch and go on (wall_clock64 ticks at 100 MHz -> 10 s)
                if (wall_clock64() - t_wait0 > 1000000000ll) { atomicOr(a.status, 1u); atomicOr(&a.sched[a.B + 1], 1u); acc[3] = 1u; break; } // (the decoder's sticky word and this launch's own)
            }
#ifndef SWD_HANDOFF_NOFENCE
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
        }
        __syncthreads();
        const uint32_t *st32 = (const uint32_t *)state_b;
        uint32_t *sdet32 = (uint32_t *)sdet;
        if constexpr (SLICE) {
            for (int r = tid; r < dlen / 4; r += NT)
                sdet32[r] = __hip_atomic_load(&st32[4 + dbase / 4 + r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            for (int r = tid; r < (a.num_det + 3) / 4; r += NT)
                sdet32[r] = __hip_atomic_load(&st32[4 + r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid == 0) { acc[0] = __hip_atomic_load(&st32[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); acc[1] = 0; }
    }
    // Scratch that one window writes and reads back lives with the workgroup, not with the shot: consecutive
    // windows of a shot run on different XCDs, whose L2s do not see each other's ordinary stores.
    const int sidx = a.slot_scratch ? (int)blockIdx.x : b;
    double *hist_b = a.hist + (int64_t)sidx * a.hist_stride;
    s.fpar = 0;
    __syncthreads();
#ifdef SWD_TSPROF // diagnostic build: when the unit started (a parked tree's commit happens in another iteration of this loop)
    if (a.prof && tid == 0 && final_ctx < 0 && !ens_task) a.prof[((int64_t)b * a.W + wi) * 8 + 4] = wall_clock64();
#endif
#ifdef SWD_GDG_DEBUG
    asm volatile("" ::: "memory"); dbg_tb = wall_clock64(); asm volatile("" ::: "memory");
    if (queued && tid == 0) { uint32_t *dbg_status = a.gdgp.chk_status; GDG_COUNT(13, dbg_tb - t_unit0); GDG_COUNT(14, 1); }
#endif
    {
        const SwdWindowDev &w = a.wins[wi];
        const SwdGraphDev &g = w.g;
        const SwdLdsLayout &L = w.L;
        lds_bind(s, smem, L, BIG ? (char *)(a.big + (int64_t)blockIdx.x * a.big_stride) : smem);
        WinResult R;
#ifdef SWD_SELPROF
        if (tid < 8) s.scal[20 + tid] = 0;
        __syncthreads();
#endif
#ifdef SWD_BPPROF
        if (tid == 0) { s.scal[24] = s.scal[25] = s.scal[26] = s.scal[27] = 0; s.scal[20] = s.scal[21] = s.scal[22] = 0; }
#endif
        if (acc[3]) { // the predecessor never arrived: nothing is decoded or committed for this unit
            for (int v = tid; v < g.n; v += NT) s.hard[v] = 0;
            R = WinResult{};
            R.exit_class = SWD_EXIT_SCHED_FAULT;
        } else if constexpr (KIND == 0 || KIND == 3) // 3: osd_window with the posterior history accumulated in registers (bp_run, ACC)
        {
            if constexpr (kSpecLoad) {
                const bool hit = g.vn_edge == spec_tab && g.llr == a.wins[wi_spec].g.llr && g.n == a.wins[wi_spec].g.n;
                wi_spec = wi;
                decode_window<NT, VF, DM, KG, SF, KIND == 3, BIG>(g, L, a.P, s, sdet + (w.row0 - dbase), hist_b, a.osd0 ? a.osd0 + (int64_t)b * g.n : nullptr,
                                                             a.bp_dec ? a.bp_dec + (int64_t)b * g.n : nullptr, R, w.cn_map, vspec, hit);
            } else {
                VnRaw<NT, (NT <= SWD_TUNED_NT) ? VF : 1, DM> vraw;
                decode_window<NT, VF, DM, KG, SF, KIND == 3, BIG>(g, L, a.P, s, sdet + (w.row0 - dbase), hist_b, a.osd0 ? a.osd0 + (int64_t)b * g.n : nullptr,
                                                             a.bp_dec ? a.bp_dec + (int64_t)b * g.n : nullptr, R, w.cn_map, vraw);
            }
        }
        else {
            uint8_t *snap_b = a.snap + (int64_t)sidx * a.snap_stride;
            bool redo = false;
            if (ens_task) { // kind 7: nothing to decode first -- the walk below starts from the context
            } else if (final_ctx >= 0 && KIND == 7) { // the result of a parked ensemble: every hypothesis has written its offer
                if constexpr (KIND == 7) {
                    const EnsCtx ec = gdg_ens_ctx(a.gdgp, final_ctx);
                    gdg_ens_finalize<NT>(g, a.P, s, ec, R);
                    if (tid == 0) ring_push(a.gdgp.fq, a.gdgp.fmask, (uint32_t)final_ctx);
                    __syncthreads();
                }
            } else
            if (final_ctx >= 0) { // the result of a parked tree
                const GdgCtx c = gdg_ctx(a.gdgp, final_ctx);
#ifdef SWD_GDG_CHECKS
                if (tid == 0) {
                    uint32_t *chk_status = a.gdgp.chk_status;
                    GDG_CHECK(final_ctx < a.gdgp.nctx, 9);
                    GDG_CHECK(atomicExch(&c.hdr[1], 2u) == 1u, 10);   // exactly one FINAL per parked tree
                    GDG_CHECK(ag_ld(&c.hdr[22]) == 1u && ag_ld(&c.hdr[18]) == ag_ld(&c.hdr[19]), 11);
                    GDG_CHECK(wi < a.W && b < a.B, 12);
                }
#endif
                redo = !gdg_finalize<NT>(g, a.P, s, c, R); // its snapshot area overflowed: the whole unit again, serially
                if (tid == 0) {
                    while (ag_ld(&c.hdr[0]) != 0u) __builtin_amdgcn_s_sleep(2); // the scheduler run that queued this item has left
                    ring_push(a.gdgp.fq, a.gdgp.fmask, (uint32_t)final_ctx);
                }
                __syncthreads();
            }
            if (!ens_task && (final_ctx < 0 || redo)) {
#ifdef SWD_GDG_DEBUG
                asm volatile("" ::: "memory"); dbg_tc = wall_clock64(); asm volatile("" ::: "memory");
#endif
                s.fpar = 0;
                decode_window_gdg<NT, VF, DM, KG, KIND == 7, VFP>(g, L, a.P, s, sdet + (w.row0 - dbase), hist_b, snap_b, R, (queued && !redo) ? &a : nullptr, acc, wi, b);
#ifdef SWD_GDG_DEBUG
                if (tid == 0 && queued) { uint32_t *dbg_status = a.gdgp.chk_status; GDG_COUNT(R.exit_class == -2 ? 9 : 10, 1); GDG_COUNT(R.exit_class == -2 ? 11 : 12, wall_clock64() - t_unit0); }
#endif
                if (R.exit_class == -2) continue; // parked: a FINAL item brings the result
            }
            if constexpr (KIND == 7) {
                // The threaded ensemble (swd_gdg_kernel.h, gdg_ensemble_tree), inlined HERE and nowhere else: a unit whose pre-processing
                // BP failed arrives with exit class -3 after BPGD::reset and the peeling (decode_window_gdg), a task of a parked
                // ensemble arrives from the queue.  Roles: 1 = the unit's owner (a context is free and the tree has a level to share:
                // the tree threads become tasks, a FINAL item brings the result), 0 = the whole ensemble here, 2 = one tree thread.
                if (ens_task || R.exit_class == -3) {
                    GdgLds G;
                    gdg_bind(G, gdg_lds_base(s, L), L, g.n, g.new_n);
                    SwdGraphDev g_ens = g;
                    int role = 0, hyp = 0;
                    bool dead_unsat = false;
                    EnsCtx ec{};
                    if (ens_task) {
                        role = 2; hyp = (int)(ens_item & 0xFFu);
                        ec = gdg_ens_ctx(a.gdgp, (int)((ens_item >> 8) & 0x3FFFFFu));
                        dead_unsat = ag_ld(&ec.hdr[6]) != 0u;
                        s.fpar = 0;
                        __syncthreads();
                        for (int l = tid; l < g.m; l += NT) s.cn_deg0[l] = g.row_deg[l];
                        for (int v = tid; v < g.n; v += NT) { s.vn_val[v] = 0; s.hard[v] = 0; }
                        for (int j = tid; j <= g.K; j += NT) s.jptr[j] = g.jptr[j];
                        for (int i = tid; i < (g.new_n + 1) / 2; i += NT) {
                            const uint32_t x = ag_ld(&ec.pos[i]);
                            G.pos_lv[2 * i] = (uint16_t)x;
                            if (2 * i + 1 < g.new_n) G.pos_lv[2 * i + 1] = (uint16_t)(x >> 16);
                        }
                        if (a.P.max_iter_per_step < 4)
                            for (int i = a.P.max_iter_per_step * g.n + tid; i < 4 * g.n; i += NT) hist_b[i] = 0.0;
                        __syncthreads();
                        // the masks of the thread's fork decide which checks are live: the check-to-thread map of the walk comes from them
                        {
                            const int64_t fkb = gdg_ens_fork_bytes(g.m, g.new_n, gdg_ens_fork_cells(g.E + 1 + 2 * (NT / 64), NT, VFP, DM));
                            const int Tt = (1 << a.P.max_tree_depth) - 1, nfk = 1 << (a.P.max_tree_depth - 1);
                            gdg_snap_load<NT>(g, s, G, hyp <= Tt ? ec.fork + (int64_t)(hyp >> 1) * fkb
                                                                  : ec.fork + (int64_t)nfk * fkb + (int64_t)(hyp - Tt - 1) * ((gdg_snap_bytes(g.m, g.new_n) + 15) & ~(int64_t)15));
                        }
                    } else {
                        dead_unsat = R.live_vn != 0;
                        gdg_staged_graph(g_ens, g, s); // (decode_window_gdg staged the window's edge columns and check order in LDS)
                        if (queued && a.gdgp.ctx && a.P.max_tree_depth >= 1) {
                            __syncthreads();
                            if (tid == 0) { uint32_t id = 0; acc[1] = ring_pop(a.gdgp.fq, a.gdgp.fmask, &id) ? id : 0xFFFFFFFFu; }
                            __syncthreads();
                            const int ectx_id = (int)acc[1];
                            if (ectx_id >= 0) {
                                role = 1;
                                ec = gdg_ens_ctx(a.gdgp, ectx_id);
                                if (tid == 0) {
                                    ag_st(&ec.hdr[0], 1u); ag_st(&ec.hdr[4], (uint32_t)wi); ag_st(&ec.hdr[5], (uint32_t)b); ag_st(&ec.hdr[6], dead_unsat ? 1u : 0u);
                                    ag_st(&ec.hdr[8], 0u); ag_st(&ec.hdr[9], 0u); ag_st(&ec.hdr[10], 0u); ag_st(&ec.hdr[11], 0u); ag_st(&ec.hdr[12], (uint32_t)R.pre_it);
                                }
                                for (int k = tid; k < ec.nh; k += NT) ag_st(&ec.etab[4 * k], 0u);
                                for (int i = tid; i < (g.new_n + 1) / 2; i += NT)
                                    ag_st(&ec.pos[i], (uint32_t)G.pos_lv[2 * i] | ((2 * i + 1 < g.new_n) ? ((uint32_t)G.pos_lv[2 * i + 1] << 16) : 0u));
                                ag_publish_barrier();
                            }
                        }
                        R.live_vn = g.n; R.post_it = 0;
                    }
                    gdg_ensemble_tree<NT, VFP, DM, KG>(role, &g_ens, &a.P, s, G, hist_b, snap_b, &R, dead_unsat ? 1 : 0, &a.gdgp, &ec, hyp);
                    if (role != 0) continue; // a task is done; an owner's unit is parked: a FINAL item brings the result
                    for (int v = tid; v < g.n; v += NT) s.hard[v] = 0;
                    __syncthreads();
                    for (int j = tid; j < g.new_n; j += NT) s.hard[G.pos_lv[j]] = G.best_err[j];
                    __syncthreads();
                    R.total_it = R.pre_it + R.post_it;
                    R.exit_class = SWD_EXIT_POST;
                    R.t[5] = wall_clock64();
                }
            }
        }
        __syncthreads();
#ifdef SWD_GDG_DEBUG // diagnostic build: checksums of the unit's input syndrome and of its result vector (statistics words 5, 6)
        if (tid == 0) {
            uint32_t hi = 2166136261u, ho = 2166136261u;
            for (int r = 0; r < g.m; ++r) hi = (hi ^ sdet[w.row0 - dbase + r]) * 16777619u;
            for (int v = 0; v < g.n; ++v) ho = (ho ^ s.hard[v]) * 16777619u;
            R.live_cn = (int)hi; R.live_e = (int)ho;
        }
        __syncthreads();
#endif
        if (a.total) {
            uint8_t *tot_b = a.total + (int64_t)b * a.total_stride + w.col0;
            uint32_t *sdet_w = (uint32_t *)sdet;
            for (int i = tid; i < w.commit; i += NT) {
                const uint8_t hv = s.hard[i];
                tot_b[i] = hv;
                if (hv) {
                    const int c = w.col0 + i;
                    if (a.obs_mask) { const uint32_t om = a.obs_mask[c]; if (om) atomicXor(&acc[0], om); }
                    for (uint32_t e = a.chk_colptr[c]; e < a.chk_colptr[c + 1]; ++e) {
                        const int r = a.chk_rows[e];
                        if constexpr (!SLICE) { atomicXor(&sdet_w[r >> 2], 1u << ((r & 3) * 8)); continue; }
                        const int rl = r - dbase;
                        if (rl >= 0 && rl < dlen) atomicXor(&sdet_w[rl >> 2], 1u << ((r & 3) * 8));
                        else __hip_atomic_fetch_xor(&((uint32_t *)state_b)[4 + (r >> 2)], 1u << ((r & 3) * 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // a row of another window (not in the reference's circuits)
                    }
                }
            }
        }
        if (a.win_out && wi == a.W - 1) {
            uint8_t *o = a.win_out + (int64_t)b * a.win_out_stride;
            for (int v = tid; v < g.n; v += NT) o[v] = s.hard[v];
        }
        if (tid == 0) {
            if (a.stats) {
                int32_t *st = a.stats + ((int64_t)b * a.W + wi) * SWD_STAT_WORDS;
                st[0] = R.exit_class | (R.conv ? SWD_STATUS_CONVERGE : 0);
                st[1] = R.total_it; st[2] = R.pre_it; st[3] = R.post_it;
                st[4] = R.live_vn; st[5] = R.live_cn; st[6] = R.live_e;
#ifdef SWD_GDG_DEBUG // diagnostic build: word 7 counts how often the unit was committed (the caller zeroes the array)
                atomicAdd(&st[7], 1);
#else
                st[7] = R.osd_rowadds;
#endif
#ifdef SWD_BPPROF
                if (R.post_it > 0) st[7] = s.scal[28];
#endif
            }
            if (a.min_pm) a.min_pm[(int64_t)b * a.W + wi] = R.pm;
            if (a.prof) {
                int64_t *pr = a.prof + ((int64_t)b * a.W + wi) * 8;
                const long long tend = wall_clock64();
                long long prev = R.t[0];
#ifdef SWD_TSPROF
                const int64_t ts_start = pr[4];
#endif
#pragma unroll
                for (int i = 1; i <= 7; ++i) {
                    const long long cur = R.t[i] ? R.t[i] : prev;
                    pr[i - 1] = cur - prev;
                    prev = cur;
                }
                pr[7] = tend - prev;
#ifdef SWD_GDG_DEBUG
                pr[6] = R.t[0] - t_unit0; pr[2] = dbg_ta - t_unit0; pr[3] = dbg_tb - t_unit0; pr[4] = dbg_tc - t_unit0;
#else
                pr[0] += R.t[0] - t_unit0; // ticket, wait for the previous window, state load
#endif
#ifdef SWD_TSPROF
                pr[5] = t_unit0; pr[6] = tend; pr[4] = ts_start; // absolute ticks: start of the iteration that committed, commit, start of the unit
#endif
#ifdef SWD_GDGPROF
                pr[0] = R.t[2] - R.t[0]; pr[1] = (R.t[3] ? R.t[3] : R.t[2]) - R.t[2];
                pr[2] = R.gp[0]; pr[3] = R.gp[1]; pr[4] = R.gp[2]; pr[5] = R.gp[3]; pr[6] = R.gp[4]; pr[7] = tend - R.t[0];
#endif
#ifdef SWD_BPPROF
                pr[0] = s.scal[24]; pr[5] = s.scal[25]; pr[6] = s.scal[26]; pr[7] = s.scal[27];
                pr[1] = s.scal[20]; pr[2] = s.scal[21]; pr[3] = s.scal[22]; pr[4] = s.scal[23];
#endif
#ifdef SWD_INITPROF
                pr[0] = R.t[0] - t_unit0; pr[1] = s.scal[20]; pr[2] = s.scal[21]; pr[3] = s.scal[22]; pr[4] = s.scal[23]; pr[5] = s.scal[24];
                pr[6] = tend - R.t[1]; pr[7] = tend - (R.t[7] ? R.t[7] : tend);
#endif
#ifdef SWD_SELPROF
                for (int k = 0; k < 8; ++k) { pr[k] = s.scal[20 + k]; s.scal[20 + k] = 0; }
#endif
#ifdef SWD_SHPROF
                pr[1] = s.scal[20]; pr[2] = s.scal[21]; pr[3] = s.scal[22]; pr[4] = s.scal[23];
                if (tid == 0) { s.scal[20] = s.scal[21] = s.scal[22] = s.scal[23] = 0; }
#endif
            }
        }
        __syncthreads();
        if (wi + 1 < a.W) {
            // hand the shot to its next window: state, then an agent-scope release of the window count
            uint32_t *st32 = (uint32_t *)state_b;
            const uint32_t *sdet32 = (const uint32_t *)sdet;
            if constexpr (SLICE) {
                for (int r = tid; r < dlen / 4; r += NT)
                    __hip_atomic_store(&st32[4 + dbase / 4 + r], sdet32[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                for (int r = tid; r < (a.num_det + 3) / 4; r += NT)
                    __hip_atomic_store(&st32[4 + r], sdet32[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (tid == 0) __hip_atomic_store(&st32[0], acc[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the state stores are acknowledged ...
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();                                  // ... by every wave, before the counter moves
#ifndef SWD_HANDOFF_NOFENCE
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
#endif
            if (tid == 0) {
                __hip_atomic_store(&a.sched[1 + b], (uint32_t)(wi + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if constexpr (kQueue) { if (queued) ring_push(a.gdgp.q, a.gdgp.qmask, item_unit(b, wi + 1)); } // the shot's next window is ready
            }
        }
        if constexpr (kQueue) {
            if (queued && tid == 0) {
                if (wi == a.W - 1) { // the shot is finished: admit the next one
                    const uint32_t nb = shots0 + atomicAdd(a.sched, 1u);
                    if (nb < (uint32_t)a.B) ring_push(a.gdgp.q, a.gdgp.qmask, item_unit(a.order ? (int)a.order[nb] : (int)nb, 0));
                }
                if (atomicAdd(&a.gdgp.q[2], 1u) + 1u == nunits) // the launch's last unit: release every workgroup
                    for (uint32_t k = 0; k < gridDim.x; ++k) ring_push(a.gdgp.q, a.gdgp.qmask, SWD_ITEM_EXIT);
            }
        }
    }
    if (a.shot_result && wi == a.W - 1) {
        // osd.py:184-187: flagged = residual syndrome of the whole run non-zero; observable flips
        // predicted by the committed faults (compared with the sampled ones by the caller)
        bool nz = false;
        if constexpr (SLICE) { for (int r = tid; r < dlen; r += NT) nz |= (sdet[r] != 0); }
        else { for (int r = tid; r < a.num_det; r += NT) nz |= (sdet[r] != 0); }
        if (SLICE && a.W > 1) { // the rows of the earlier windows: from the state record
            const uint32_t *st32 = (const uint32_t *)state_b;
            for (int q = tid; q < (a.num_det + 3) / 4; q += NT)
                if (4 * q < dbase || 4 * q >= dbase + dlen) nz |= __hip_atomic_load(&st32[4 + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
        }
        const bool any = block_any<NT>(nz, s);
        if (tid == 0) { a.shot_result[2 * b] = (int32_t)acc[0]; a.shot_result[2 * b + 1] = any ? 1 : 0; }
    }
#ifdef SWD_RESIDENCY
    ++res_units;
#endif
    } // next unit
#ifdef SWD_RESIDENCY
    if (tid == 0) { atomicSub(&a.status[4], 1u); if (res_units) atomicAdd(&a.status[6], 1u); atomicMax(&a.status[7], res_units); }
#endif
}

} // namespace swd
