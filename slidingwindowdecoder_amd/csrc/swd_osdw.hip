// C-ABI host side (include/swd.h) of the window decoder:
//   swd_osdw_*      one window matrix, B syndromes per call  (replaces the reference's osd_window
//                   extension type, /root/reference/src/osd_window.pyx, behind a batch interface)
//   swd_pipeline_*  the whole (W,F) sliding-window loop of /root/reference/osd.py:130-179 for B
//                   shots in one launch
// Both run swd::pipeline_kernel (swd_osdw_kernel.h); a single window is a pipeline of length 1.
#include <thread>

#include "swd_plan.h"
#include "swd_variants.h"

namespace swd {

int make_layout(const Graph &g, int new_n, int nt, int kind, SwdLdsLayout &L, bool big, int lds_budget) {
    const int m = g.m, n = g.n, E = g.E, wm = g.wm;
    const int npad = std::max(next_pow2(n), 2);
    L.npad = npad;
    L.off_idx = npad * 8;
    L.off_aux = align_up(npad * 10, 16);
    int osd_bytes = align_up(L.off_aux + m * wm * 8 + wm * 8 + g.rank * 4 + n * 2 + 16, 16);
    const int osd_bytes_at_aux = osd_bytes; // end of the arrays that start at off_aux (the higher-order sweep's may follow)
    // ring of row operations of the four-wave elimination (osd0_quad: m <= 256, at least four waves, scratch region in LDS)
    L.off_oring = -1;
    if (wm <= 4 && nt >= 256 && !big) { L.off_oring = osd_bytes; osd_bytes += align_up(SWD_QUAD_RING_BYTES, 16); }
    // higher-order OSD arrays: over the dead sort keys when they fit there, else after the OSD-0 arrays
    // candidates evaluated concurrently: as many threads as still let the sweep's arrays lie over the dead sort keys (at least 256):
    // [[288]] (4,1) windows, OSD-CS 10: 621 candidates in one round of 640 threads instead of three rounds of 256
    const int kset = std::max(new_n - g.rank, 0);
    auto cs_need = [&](int cp) { return align_up(n * 2 + kset * 2 + g.rank * 4 + 64 + 8, 8) + g.rank * 8 + std::max(wm * cp, 64 + wm) * 8; };
    L.cs_par = nt;
    while (L.cs_par > std::min(nt, 256) && cs_need(L.cs_par) > npad * 8) L.cs_par -= 64;
    const int cs_bytes = cs_need(L.cs_par);
    if (cs_bytes <= npad * 8) L.off_cs = 0;
    else { L.off_cs = osd_bytes; osd_bytes += align_up(cs_bytes, 16); }
    const int rare_bytes = L.off_aux + std::max(n * 2, 3 * 256 * 4); // failed-decimation positions / select histograms
    // messages + slot E + one far and one zero slot per wave (swd_osdw_kernel.h, VnCache)
    int scratch = std::max(std::max((E + 1 + 2 * (nt / 64)) * 8, osd_bytes), std::max(rare_bytes, n * 2));
    // post-phase check order (degree histogram + order): behind the staged slot lists when those fit
    L.off_cord = align_up((g.K * m * 2 <= scratch) ? std::max(L.off_aux, g.K * m * 2) : L.off_aux, 16);
    L.off_rc = align_up(std::max(L.off_cord + 66 * 4 + m * 2, L.off_aux + 3 * 256 * 4), 16); // clear of the select histograms
    L.off_bak = align_up(L.off_rc + E * 2, 16);
    scratch = std::max(scratch, L.off_bak + 10 * m + 2 * n + 8);
    // exchange slots of the column-form elimination (osd0_cols: 1024 threads, 256 < m <= 576): tail of the sort keys,
    // the staged row lists of the sorted columns end before it
    L.off_oslot = -1;
    if (wm > 4 && wm <= 9 && nt >= 1024) {
        // ring of SWD_OSD_RING row operations (10 words each), hand-over block of one batch (9 x 64 words), pivoted-row mask, control words
        const int sb = align_up(SWD_OSD_RING * 10 * 8 + 9 * 64 * 8 + 10 * 8 + 16 * 4, 16);
        if (sb <= npad * 4) L.off_oslot = npad * 8 - sb;
    }
    L.off_hs = align_up(L.off_aux + n * 2, 16); // behind the decided-0 list of the OSD ordering, used before the elimination sets up
    scratch = std::max(scratch, L.off_hs + n * 8);
    scratch = align_up(scratch, 16);
    // large graphs: the scratch region goes to HBM; what follows is laid out from LDS offset 0
    L.big_scratch = big ? scratch : 0;
    int o = big ? 0 : scratch;
    // The tuned osd_window kernels (up to 256 threads, swd_osdw_kernel.h: SWD_P16 / DIET) keep a smaller state: 32 + 16 bits of
    // live mask per check (row weight <= 48), one parity byte per check, no copy of the original check degrees, one
    // "decided" bit per variable node.  Every other kernel: the classic arrays.
    const bool diet = kind == 0 && nt <= SWD_TUNED_NT;
    L.off_livemask = o; o += (diet && g.K <= 48) ? align_up(m * 6, 4) : m * 8;
    L.off_par = o; o += diet ? align_up(m + 1, 4) : (m + 1) * 4; // parities; index m: sink for the dead positions' flips
    L.off_lv = o; o = align_up(o + new_n * 2, 4);
    L.off_jptr = o; o = align_up(o + (g.K + 1) * 2, 4);
    // live-slot lists: osd_window stages them in the (then dead) scratch region; the guessing decoders
    // keep their messages alive across decimation steps and get a dedicated region
    L.off_lslot = (g.K * m * 2 <= scratch) ? 0 : -1;
    L.off_gdg = -1;
    if (kind != 0) {
        L.off_gdg = o;
        o = align_up(o + new_n * 2 * 2 + SWD_GDG_MAXGUESS * 2 * 2 + new_n + n + SWD_GDG_MAXGUESS + new_n, 8);
        L.off_lslot = o;
        o = align_up(o + g.K * m * 2, 8);
    }
    L.off_cnval = o; o += m;
    L.off_cndeg = o; o += m;
    L.off_cndeg0 = o; o += diet ? 0 : m;
    L.off_vnval = o = align_up(o, 4); o += diet ? ((n + 31) / 32) * 4 : n;
    L.off_hard = o; o = align_up(o + n + 1, 16); // +1: sink for threads without a VN
    L.off_misc = o; o += 640; // flags[32] scal[32] dbl[24] iaux[32]
    o = align_up(o, 16);
    // large graphs: what fits beside the state goes back into LDS -- the messages of the shortened graph (one column of cells per
    // live variable node: D x new_n cells + the sink, far and zero slots) and the arrays of the OSD phase that start at off_aux
    L.off_pmsg = -1; L.pmsg_bytes = 0; L.post_lds = 0; L.osd_lds = 0;
    if (big && kind == 0 && !getenv("SWD_BIG_NO_LDS")) {
        const int post_b = align_up((g.D * new_n + 1 + 2 * (nt / 64)) * 8, 16);
        // (transform matrix, pivots; the ordered list behind them stays in the scratch region -- osd_run)
        const int aux_b = align_up(std::max(std::max(osd_bytes_at_aux - n * 2, rare_bytes) - L.off_aux, L.off_hs + n * 8 - L.off_aux), 16);
        const int avail = lds_budget - o;
        const bool post_ok = g.D * new_n <= 65535 && post_b <= avail && L.off_lslot == 0 && new_n <= 2 * nt, aux_ok = aux_b <= avail;
        if (post_ok || aux_ok) {
            L.off_pmsg = o;
            L.post_lds = post_ok ? 1 : 0; L.osd_lds = aux_ok ? 1 : 0;
            L.pmsg_bytes = std::max(post_ok ? post_b : 0, aux_ok ? aux_b : 0);
            o += L.pmsg_bytes;
        }
    }
    // large graphs, 256 < m <= 960: ring (64 or 32 entries of 16 words), pivoted-row mask and control words of osd0_colsw
    L.off_owide = -1; L.owide_ring = 0;
    if (big && kind == 0 && nt == 1024 && wm > 4 && wm <= 15 && !getenv("SWD_NO_OSD_WIDE")) {
        const char *rq = getenv("SWD_OWIDE_RING"); // (tests: a smaller ring, so that batches close early)
        for (int ring = rq ? std::max(atoi(rq), 4) : 64; ring >= (rq ? 4 : 32); ring >>= 1) {
            const int bytes = align_up(ring * 16 * 8 + 16 * 8 + 64, 16);
            if (o + bytes <= lds_budget) { L.off_owide = o; L.owide_ring = ring; o += bytes; break; }
        }
    }
    // experiment (SWD_POST_RENUM=1, tuned osd_window kernels of up to 256 threads): renumber the shortened graph's message cells
    // one column per live variable node inside the scratch region; the old-slot -> cell table takes the staged column table's place
    // (round 5, SWD_POST_SORTED: the production form of the tuned kernels' post phase whenever the renumbered cells fit in front of the
    // staged column table, which becomes the old-slot -> cell table; SWD_NO_POST_SORTED=1 in the environment keeps the round-4 form)
    // (diet kernels keep cell BYTE offsets in 16 bits, the 1024-thread osd_window kernels cell numbers)
    if (kind == 0 && !big && !getenv("SWD_NO_POST_SORTED") && L.off_lslot == 0 && (g.D * new_n + 1 + 2 * (nt / 64)) * 8 <= L.off_rc &&
        g.D * new_n + 1 + 2 * (nt / 64) <= (diet ? 8191 : 65535) && (diet || nt >= 512))
        L.post_lds = 1;
    L.total = align_up(o, 16);
    return 0;
}

// OSD-only layout of a graph (used by the quaternary decoder): npad / off_idx / off_aux / off_cs / cs_par,
// off_livemask = bytes of scratch the OSD phase needs
int make_layout_for_osd(const Graph &g, int nt, SwdLdsLayout &L) { return make_layout(g, g.n, nt, 0, L, false, 0); }


// Static check-to-thread map of the full-graph BP phase for variants that share heavy checks among threads
// (same rule as cn_assign on the device: smallest T <= cap such that one thread per check of degree <= T,
// two up to 2T, four up to 4T fit nt threads; lanes are already sorted by decreasing degree).
bool split_map(const Graph &g, int nt, int cap, std::vector<uint32_t> *map) {
    static const int cand[8] = {3, 4, 6, 8, 12, 16, 24, 32};
    for (int ci = 0; ci < 8; ++ci) {
        const int T = cand[ci];
        if (T > cap) break;
        long need = 0;
        bool ok = true;
        for (int l = 0; l < g.m; ++l) {
            const int d = g.row_deg[l];
            if (d > 4 * T) { ok = false; break; }
            need += d <= T ? 1 : (d <= 2 * T ? 2 : 4);
        }
        if (!ok || need > nt) continue;
        if (map) {
            map->assign(nt, 0xFFFFu);
            int t = 0;
            for (int pass = 4; pass >= 1; pass >>= 1) // quads first, then pairs, then singles (degrees are sorted)
                for (int l = 0; l < g.m; ++l) {
                    const int d = g.row_deg[l];
                    const int grp = d <= T ? 1 : (d <= 2 * T ? 2 : 4);
                    if (grp != pass) continue;
                    for (int r = 0; r < grp; ++r) (*map)[t++] = (uint32_t)l | ((uint32_t)r << 16) | ((uint32_t)grp << 18);
                }
        }
        return true;
    }
    return false;
}

#define SWD_IF_0(...)
#define SWD_IF_1(...) __VA_ARGS__
#define SWD_IF(c, ...) SWD_IF_##c(__VA_ARGS__)
#define SWD_PTR_0(kind, nt, vf, dm, kg, sf) nullptr
#define SWD_PTR_1(kind, nt, vf, dm, kg, sf) SWD_LAUNCHER_NAME(kind, nt, vf, dm, kg, sf)
#define X(nt, vf, dm, kg, sf, k1, k2, k3) SWD_DECLARE_LAUNCHER(0, nt, vf, dm, kg, sf) SWD_IF(k1, SWD_DECLARE_LAUNCHER(1, nt, vf, dm, kg, sf)) SWD_IF(k2, SWD_DECLARE_LAUNCHER(2, nt, vf, dm, kg, sf)) SWD_IF(k3, SWD_DECLARE_LAUNCHER(3, nt, vf, dm, kg, sf)) SWD_IF(k1, SWD_DECLARE_LAUNCHER(7, nt, vf, dm, kg, sf))
SWD_VARIANTS(X)
#undef X
static const Variant kVariants[] = {
#define X(nt, vf, dm, kg, sf, k1, k2, k3) {nt, vf, dm, kg, sf, SWD_PTR_1(0, nt, vf, dm, kg, sf), SWD_PTR_##k1(1, nt, vf, dm, kg, sf), SWD_PTR_##k2(2, nt, vf, dm, kg, sf), SWD_PTR_##k3(3, nt, vf, dm, kg, sf), SWD_PTR_##k1(7, nt, vf, dm, kg, sf)},
    SWD_VARIANTS(X)
#undef X
};

#define X(nt, vf, dm, kg) SWD_DECLARE_LAUNCHER(5, nt, vf, dm, kg, 0) SWD_DECLARE_LAUNCHER(6, nt, vf, dm, kg, 0) SWD_DECLARE_LAUNCHER(8, nt, vf, dm, kg, 0) SWD_DECLARE_LAUNCHER(9, nt, vf, dm, kg, 0)
SWD_BIG_VARIANTS(X)
#undef X
static const Variant kBigVariants[] = {
#define X(nt, vf, dm, kg) {nt, vf, dm, kg, 0, SWD_LAUNCHER_NAME(5, nt, vf, dm, kg, 0), SWD_LAUNCHER_NAME(8, nt, vf, dm, kg, 0), nullptr, SWD_LAUNCHER_NAME(6, nt, vf, dm, kg, 0), SWD_LAUNCHER_NAME(9, nt, vf, dm, kg, 0)},
    SWD_BIG_VARIANTS(X)
#undef X
};

const Variant *select_big_variant(int mmax, int nmax, int dm, int kmax, int kind) {
    for (const Variant &v : kBigVariants) {
        if (kind != 0 && !v.launch_gdg) continue;
        if (v.nt >= mmax && v.nt * v.vf >= nmax && v.dm >= dm && 4 * v.kg >= kmax) return &v;
    }
    return nullptr;
}

const Variant *select_variant(const std::vector<WindowHost> &wins, int mmax, int nmax, int dm, int kmax, int kind) {
    for (const Variant &v : kVariants) {
        if (!(kind == 0 ? v.launch != nullptr : v.launch_gdg != nullptr)) continue;
        if (!(v.nt >= mmax && v.nt * v.vf >= nmax && v.dm >= dm)) continue;
        if (kind == 0 && v.nt <= SWD_TUNED_NT) { // these kernels keep LDS byte offsets in 16 bits (swd_osdw_kernel.h, P16)
            bool fits = true;
            for (auto &w : wins) {
                SwdLdsLayout L{};
                make_layout(*w.g, w.new_n, v.nt, kind, L);
                if (L.total > 65536) { fits = false; break; }
            }
            if (!fits) continue;
        }
        if (!v.sf) { if (4 * v.kg >= kmax) return &v; continue; }
        bool ok = true;
        for (auto &w : wins) {
            SwdLdsLayout L{};
            make_layout(*w.g, w.new_n, v.nt, kind, L);
            if (L.off_lslot < 0 || !split_map(*w.g, v.nt, 4 * v.kg, nullptr)) { ok = false; break; } // the post phase must walk lists
        }
        if (ok) return &v;
    }
    return nullptr;
}

static int launch(Plan *d, const SwdPipeArgs &a0, hipStream_t st) {
    // work-unit scheduling state: ticket counter + per-shot progress (zeroed per launch), hand-over buffer
    SwdPipeArgs a = a0;
    std::lock_guard<std::recursive_mutex> lk(d->mu);
    Plan::LaunchSlot &sl = d->slot[d->next_slot];
    d->next_slot = (d->next_slot + 1) % Plan::kSlots;
    d->cur = &sl;
    if (!sl.done) SWD_HIP(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    else SWD_HIP(hipStreamWaitEvent(st, sl.done, 0)); // the slot's previous launch (possibly on another stream) is done first
    a.state_stride = 16 + align_up(d->num_det, 16);
    // [ ticket counter | per-shot progress [B] | this LAUNCH's fault word ]: the kernel raises a scheduling fault in the decoder's
    // sticky word (swd_pipeline_status) and in sched[B + 1], which belongs to this launch alone (the stream lanes read it)
    if (sl.sched.reserve((size_t)(a.B + 2) * 4) || sl.state.reserve((size_t)a.B * a.state_stride)) return -1;
    a.sched = sl.sched.as<uint32_t>(); a.state = sl.state.as<uint8_t>();
    a.status = d->status.as<uint32_t>();
    SWD_HIP(hipMemsetAsync(a.sched, 0, (size_t)(a.B + 2) * 4, st));
    if (d->timing) {
        if (!d->ev0) { SWD_HIP(hipEventCreate(&d->ev0)); SWD_HIP(hipEventCreate(&d->ev1)); }
        SWD_HIP(hipEventRecord(d->ev0, st));
    }
    int rc;
    // guessing decoders: the parallel form (side branches as work items) shortens the critical path of a batch that
    // cannot fill the device with whole shots; large batches keep the serial walk (no speculation, no queue traffic)
    static const int par_max_shots = getenv("SWD_GDG_PAR_MAX_SHOTS") ? atoi(getenv("SWD_GDG_PAR_MAX_SHOTS")) : 5120; // measured, [[144]] GDG windows (work items with 64 contexts vs serial, round 6): 4096 shots 36.9 vs 41.3 ms, 5120 shots 44.2 vs 45.2, 6144 shots 50.3 vs 47.5, 8192 shots 66.6 vs 55.9 (round 5, every context handed out: 4608)
    const int stream_serial_min = getenv("SWD_GDG_STREAM_SERIAL_MIN") ? atoi(getenv("SWD_GDG_STREAM_SERIAL_MIN")) : 3072; // (read per launch: the tests switch it)
    // ... and so do the batches of a stream object from 3072 shots (round 6): the work-item form shortens ONE launch's critical path at the price of
    // queue traffic and idle polling; with two batches in flight the next launch's grid fills the tail the serial walk leaves --
    // [[144]] GDG windows, 4096 shots per batch: 1.19 M windows/s one launch at a time (work items), 1.21 M streamed with work items,
    // 1.56 M streamed with the serial walk; streamed, serial against work items: 1024 shots 14.6 / 12.4 ms per batch, 2048 26.1 / 21.9,
    // 3072 25.3 / 29.6, 4096 28.9 / 37.3 (profiles/r06_gdg_stream.log)
    // (a caller that alternates streams of its own is in the same position: the previous launch of this handle still runs on another stream)
    bool overlapped = false;
    if (d->kind == 1 && !d->stream_push && d->last_done && d->last_stream != st) {
        overlapped = hipEventQuery(d->last_done) == hipErrorNotReady;
        (void)hipGetLastError(); // ("not ready" is an answer, not an error)
    }
    d->stream_serial = (d->stream_push || overlapped) && a.B >= stream_serial_min; // (the threaded ensemble's launcher reads it too: swd_plan.h, KIND 7)
    const bool par = d->kind == 1 && d->gdg_parallel && !a.hist && d->variant->launch_par && a.B <= par_max_shots && a.B < SWD_GDG_ITEM_MAX_SHOTS &&
                     !d->stream_serial;
    // osd_window: when the posterior history is only consumed as its slot-order sum (no history in or out, both
    // iteration caps multiples of four) the kernel that accumulates the sum in registers runs: no 4 x n ring in HBM
    const bool acc = d->kind == 0 && d->variant->launch_acc && !a.hist && !a.P.record_all && !a.P.hist_is_state && !a.P.zero_hist &&
                     a.P.pre_iter >= 4 && a.P.pre_iter % 4 == 0 && a.P.post_iter >= 4 && a.P.post_iter % 4 == 0 && !getenv("SWD_NO_HACC");
    const bool ens = d->kind == 1 && d->gp.multi_thread == 1; // the reference's threaded ensemble: its own kernel (kind 7)
    rc = (d->kind == 0) ? (acc ? d->variant->launch_acc(d, a, st) : d->variant->launch(d, a, st))
                        : (ens ? d->variant->launch_ens(d, a, st) : (par ? d->variant->launch_par(d, a, st) : d->variant->launch_gdg(d, a, st)));
    if (rc) return rc;
    SWD_HIP(hipEventRecord(sl.done, st));
    d->last_done = sl.done; d->last_stream = st;
    if (d->timing) {
        SWD_HIP(hipEventRecord(d->ev1, st));
        SWD_HIP(hipEventSynchronize(d->ev1));
        float ms = 0;
        SWD_HIP(hipEventElapsedTime(&ms, d->ev0, d->ev1));
        d->t_total_ms += ms;
        d->t_launches += 1;
    }
    return 0;
}

static int check_device(int device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("no HIP device available: the MI355X decoder has no CPU fallback");
        return -1;
    }
    if (device < 0 || device >= ndev) { set_error("device %d out of range (%d devices)", device, ndev); return -1; }
    if (hipSetDevice(device) != hipSuccess) { set_error("hipSetDevice(%d) failed", device); return -1; }
    return 0;
}

static int check_params(swd_osdw_params &p) {
    if (p.osd_method == 0) p.osd_order = 0; // osd_window.pyx:69-71
    if (p.osd_method < 0 || p.osd_method > 2) { set_error("ERROR: OSD method '%d' invalid.", p.osd_method); return -1; }
    return 0;
}

static void fill_params(const Plan *d, SwdDecodeParams &P, bool hist_is_state, bool hist_is_output) {
    P.pre_iter = d->p.pre_max_iter; P.post_iter = d->p.post_max_iter;
    P.osd_method = d->p.osd_method; P.osd_order = d->p.osd_order; P.alpha = d->p.ms_scaling_factor;
    P.hist_is_state = hist_is_state ? 1 : 0;
    P.record_all = (hist_is_state || hist_is_output) ? 1 : 0;
    // a fresh reference object has an all-zero history; only observable when fewer than four
    // iterations ran or when the history is returned
    P.zero_hist = (!hist_is_state && (hist_is_output || d->p.pre_max_iter < 4)) ? 1 : 0;
    P.kind = d->kind;
    if (d->kind != 0) {
        P.pre_iter = d->gp.max_iter; P.alpha = d->gp.ms_scaling_factor; P.post_iter = 0;
        P.osd_method = 0; P.osd_order = -1;
        P.max_iter_per_step = d->gp.max_iter_per_step; P.max_step = d->gp.max_step;
        P.max_tree_depth = d->gp.max_tree_depth; P.max_side_depth = d->gp.max_side_depth;
        P.max_side_branch_step = d->gp.max_side_branch_step; P.low_error_mode = d->gp.low_error_mode;
        P.max_guess = d->max_guess; P.gdg_factor = d->gp.gdg_factor; P.max_tree_branch_step = d->gp.max_tree_branch_step;
        P.zero_hist = (!hist_is_state && (hist_is_output || d->gp.max_iter < 4)) ? 1 : 0;
    }
}

} // namespace swd

using namespace swd;

// ------------------------------------------------------------------------------------------
// single window
// ------------------------------------------------------------------------------------------
extern "C" swd_osdw *swd_osdw_create(const swd_graph_desc *g, const swd_osdw_params *p, int device) {
    if (!p || !g) { set_error("null argument"); return nullptr; }
    if (check_device(device)) return nullptr;
    Plan *d = new Plan();
    d->device = device;
    d->p = *p;
    std::map<std::string, std::shared_ptr<Graph>> cache;
    if (check_params(d->p)) { delete d; return nullptr; }
    if (getenv("SWD_FORCE_HUGE") /* tests: the general form on graphs a variant would take */ || d->add_window(g, 0, 0, 0, cache) || d->finalize(nullptr)) {
        // no kernel variant takes this graph (more than 1024 checks, 9216 columns, 65 535 edges, row weight 64 or column weight 10):
        // the general form, every array in HBM (swd_huge.hip) -- the reference's mod2sparse has no size limit
        d->wins.clear();
        d->huge.reset(huge_create(g, &d->p, device));
        if (!d->huge) { delete d; return nullptr; }
    }
    return (swd_osdw *)d;
}

extern "C" void swd_osdw_destroy(swd_osdw *h) {
    Plan *d = (Plan *)h;
    if (!d) return;
    (void)hipSetDevice(d->device);
    delete d;
}

extern "C" int swd_osdw_info(const swd_osdw *h, int32_t *m, int32_t *n, int32_t *new_n, int32_t *rank) {
    const Plan *d = (const Plan *)h;
    if (!d) { set_error("null decoder"); return -1; }
    if (d->huge) {
        if (m) *m = d->huge->m;
        if (n) *n = d->huge->n;
        if (new_n) *new_n = d->huge->new_n;
        if (rank) *rank = d->huge->rank;
        return 0;
    }
    const WindowHost &w = d->wins[0];
    if (m) *m = w.g->m;
    if (n) *n = w.g->n;
    if (new_n) *new_n = w.new_n;
    if (rank) *rank = w.g->rank;
    return 0;
}

extern "C" int swd_osdw_set_timing(swd_osdw *h, int32_t on) {
    Plan *d = (Plan *)h;
    if (!d) { set_error("null decoder"); return -1; }
    d->timing = on != 0; d->t_total_ms = 0; d->t_launches = 0;
    return 0;
}

extern "C" int swd_osdw_get_timing(swd_osdw *h, double *total_ms, int64_t *launches) {
    Plan *d = (Plan *)h;
    if (!d) { set_error("null decoder"); return -1; }
    if (total_ms) *total_ms = d->t_total_ms;
    if (launches) *launches = d->t_launches;
    d->t_total_ms = 0; d->t_launches = 0;
    return 0;
}

extern "C" int swd_osdw_decode_batch_dev(swd_osdw *h, int32_t B, const uint8_t *synd, int64_t synd_stride,
                                         uint8_t *out, int64_t out_stride, int32_t *stats, double *min_pm,
                                         double *hist, int32_t hist_is_state, uint8_t *osd0, uint8_t *bp_dec, void *stream) {
    Plan *d = (Plan *)h;
    if (!d) { set_error("null decoder"); return -1; }
    if (B <= 0) return 0;
    if (!synd || !out) { set_error("null output/input pointer"); return -1; }
    SWD_HIP(hipSetDevice(d->device));
    if (!hist && hist_is_state) { set_error("hist_is_state requires a caller-provided history buffer"); return -1; }
    if (d->huge) return d->huge->decode_dev(B, synd, synd_stride, out, out_stride, stats, min_pm, hist, hist_is_state, osd0, bp_dec, stream);
    const int n = d->wins[0].g->n, m = d->wins[0].g->m;
    const bool hist_out = hist != nullptr;
    SwdPipeArgs a{};
    a.wins = d->d_wins.as<SwdWindowDev>(); a.W = 1; a.B = B;
    fill_params(d, a.P, hist_is_state != 0, hist_out);
    a.det = synd; a.det_stride = synd_stride ? synd_stride : m; a.num_det = m; a.off_det = d->off_det;
    a.total = nullptr; a.win_out = out; a.win_out_stride = out_stride ? out_stride : n;
    a.stats = stats; a.min_pm = min_pm; a.hist = hist; a.hist_stride = 4 * (int64_t)n; a.osd0 = osd0; a.bp_dec = bp_dec;
    return launch(d, a, (hipStream_t)stream); // a null history / the snapshot stack come from the launch slot
}

// large batches: one synchronous copy per array straight from / to the caller's buffers
static int osdw_decode_batch_direct(swd_osdw *h, int32_t B, const uint8_t *synd, uint8_t *out, int32_t *stats,
                                     double *min_pm, double *hist, int32_t hist_is_state, uint8_t *osd0, uint8_t *bp_dec) {
    Plan *d = (Plan *)h;
    if (!d) { set_error("null decoder"); return -1; }
    if (B <= 0) return 0;
    if (!synd || !out || !stats || !min_pm) { set_error("null output/input pointer"); return -1; }
    std::lock_guard<std::recursive_mutex> lk(d->mu);
    SWD_HIP(hipSetDevice(d->device));
    const size_t m = d->m0(), n = d->n0();
    const size_t hbytes = (size_t)B * 4 * n * 8;
    if (d->synd.reserve(B * m) || d->out.reserve(B * n) || d->stats.reserve((size_t)B * SWD_STAT_WORDS * 4) ||
        d->pm.reserve(B * 8) || (hist && d->hist.reserve(hbytes)))
        return -1;
    if ((osd0 || bp_dec) && d->osd0.reserve(2 * B * n)) return -1; // [ osd0 | bp_dec ]
    SWD_HIP(hipMemcpy(d->synd.p, synd, B * m, hipMemcpyHostToDevice));
    if (hist && hist_is_state) SWD_HIP(hipMemcpy(d->hist.p, hist, hbytes, hipMemcpyHostToDevice));
    if (osd0 || bp_dec) SWD_HIP(hipMemset(d->osd0.p, 0, 2 * B * n));
    int rc = swd_osdw_decode_batch_dev(h, B, d->synd.as<uint8_t>(), 0, d->out.as<uint8_t>(), 0, d->stats.as<int32_t>(),
                                       d->pm.as<double>(), hist ? d->hist.as<double>() : nullptr,
                                       (hist && hist_is_state) ? 1 : 0, osd0 ? d->osd0.as<uint8_t>() : nullptr,
                                       bp_dec ? d->osd0.as<uint8_t>() + B * n : nullptr, nullptr);
    if (rc) return rc;
    SWD_HIP(hipDeviceSynchronize());
    SWD_HIP(hipMemcpy(out, d->out.p, B * n, hipMemcpyDeviceToHost));
    SWD_HIP(hipMemcpy(stats, d->stats.p, (size_t)B * SWD_STAT_WORDS * 4, hipMemcpyDeviceToHost));
    SWD_HIP(hipMemcpy(min_pm, d->pm.p, B * 8, hipMemcpyDeviceToHost));
    if (hist) SWD_HIP(hipMemcpy(hist, d->hist.p, hbytes, hipMemcpyDeviceToHost));
    if (osd0) SWD_HIP(hipMemcpy(osd0, d->osd0.p, B * n, hipMemcpyDeviceToHost));
    if (bp_dec) SWD_HIP(hipMemcpy(bp_dec, d->osd0.as<uint8_t>() + B * n, B * n, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int swd_osdw_decode_batch(swd_osdw *h, int32_t B, const uint8_t *synd, uint8_t *out, int32_t *stats,
                                     double *min_pm, double *hist, int32_t hist_is_state, uint8_t *osd0, uint8_t *bp_dec) {
    Plan *d = (Plan *)h;
    if (!d) { set_error("null decoder"); return -1; }
    if (B <= 0) return 0;
    if (!synd || !out || !stats || !min_pm) { set_error("null output/input pointer"); return -1; }
    std::lock_guard<std::recursive_mutex> lk(d->mu);
    SWD_HIP(hipSetDevice(d->device));
    const size_t m = d->m0(), n = d->n0();
    // one packed device buffer mirrored by a pinned host buffer: [ syndromes | history ] travel in,
    // [ history | vectors | statistics | path metrics | OSD-0 vectors ] travel out, one copy each way
    const bool hist_in = hist && hist_is_state;
    const size_t hbytes = hist ? (size_t)B * 4 * n * 8 : 0;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_synd = 0, o_hist = al((size_t)B * m), o_out = o_hist + al(hbytes), o_stats = o_out + al((size_t)B * n),
                 o_pm = o_stats + al((size_t)B * SWD_STAT_WORDS * 4), o_osd0 = o_pm + al((size_t)B * 8),
                 o_bpd = o_osd0 + (osd0 ? al((size_t)B * n) : 0), total = o_bpd + (bp_dec ? al((size_t)B * n) : 0);
    if (total > SWD_STAGE_MAX) return osdw_decode_batch_direct(h, B, synd, out, stats, min_pm, hist, hist_is_state, osd0, bp_dec); // copy time dominates there
    if (d->io.reserve(total) || d->stage.reserve(total)) return -1;
    char *hs = (char *)d->stage.p, *ds = (char *)d->io.p;
    memcpy(hs + o_synd, synd, (size_t)B * m);
    if (hist_in) memcpy(hs + o_hist, hist, hbytes);
    hipStream_t st = nullptr;
    SWD_HIP(hipMemcpyAsync(ds, hs, hist_in ? o_hist + hbytes : (size_t)B * m, hipMemcpyHostToDevice, st));
    if (osd0 || bp_dec) SWD_HIP(hipMemsetAsync(ds + o_osd0, 0, total - o_osd0, st));
    int rc = swd_osdw_decode_batch_dev(h, B, (const uint8_t *)(ds + o_synd), 0, (uint8_t *)(ds + o_out), 0, (int32_t *)(ds + o_stats),
                                       (double *)(ds + o_pm), hist ? (double *)(ds + o_hist) : nullptr, hist_in ? 1 : 0,
                                       osd0 ? (uint8_t *)(ds + o_osd0) : nullptr, bp_dec ? (uint8_t *)(ds + o_bpd) : nullptr, st);
    if (rc) return rc;
    const size_t first = hist ? o_hist : o_out;
    SWD_HIP(hipMemcpyAsync(hs + first, ds + first, total - first, hipMemcpyDeviceToHost, st));
    SWD_HIP(hipStreamSynchronize(st));
    memcpy(out, hs + o_out, (size_t)B * n);
    memcpy(stats, hs + o_stats, (size_t)B * SWD_STAT_WORDS * 4);
    memcpy(min_pm, hs + o_pm, (size_t)B * 8);
    if (hist) memcpy(hist, hs + o_hist, hbytes);
    if (osd0) memcpy(osd0, hs + o_osd0, (size_t)B * n);
    if (bp_dec) memcpy(bp_dec, hs + o_bpd, (size_t)B * n);
    return 0;
}

// ------------------------------------------------------------------------------------------
// sliding-window pipeline
// ------------------------------------------------------------------------------------------
extern "C" swd_pipeline *swd_pipeline_create(int32_t num_windows, const swd_window_desc *wins,
                                             const swd_graph_desc *chk, const swd_osdw_params *p, int device) {
    if (!p || !wins || !chk || num_windows <= 0) { set_error("null argument"); return nullptr; }
    if (check_device(device)) return nullptr;
    Plan *d = new Plan();
    d->device = device;
    d->p = *p;
    std::map<std::string, std::shared_ptr<Graph>> cache;
    if (check_params(d->p)) { delete d; return nullptr; }
    for (int i = 0; i < num_windows; ++i)
        if (d->add_window(&wins[i].graph, wins[i].row0, wins[i].col0, wins[i].commit, cache)) { delete d; return nullptr; }
    if (d->finalize(chk)) { delete d; return nullptr; }
    return (swd_pipeline *)d;
}

extern "C" void swd_pipeline_destroy(swd_pipeline *h) { swd_osdw_destroy((swd_osdw *)h); }

extern "C" int swd_pipeline_info(const swd_pipeline *h, int32_t *num_windows, int32_t *num_det, int32_t *num_col,
                                 int32_t *lds_bytes, int32_t *threads) {
    const Plan *d = (const Plan *)h;
    if (!d) { set_error("null pipeline"); return -1; }
    if (num_windows) *num_windows = (int32_t)d->wins.size();
    if (num_det) *num_det = d->num_det;
    if (num_col) *num_col = d->num_col;
    if (lds_bytes) *lds_bytes = d->lds_total;
    if (threads) *threads = d->nt;
    return 0;
}

extern "C" int swd_pipeline_set_timing(swd_pipeline *h, int32_t on) { return swd_osdw_set_timing((swd_osdw *)h, on); }
extern "C" int swd_pipeline_get_timing(swd_pipeline *h, double *total_ms, int64_t *launches) {
    return swd_osdw_get_timing((swd_osdw *)h, total_ms, launches);
}

extern "C" int swd_pipeline_set_observables(swd_pipeline *h, const swd_graph_desc *obs) {
    Plan *d = (Plan *)h;
    if (!d || !obs) { set_error("null argument"); return -1; }
    if (obs->m > 32) { set_error("at most 32 observables are supported (got %d)", obs->m); return -1; }
    if (obs->n != d->num_col) { set_error("observable matrix has %d columns, expected %d", obs->n, d->num_col); return -1; }
    SWD_HIP(hipSetDevice(d->device));
    std::vector<uint32_t> mask(d->num_col, 0);
    for (int k = 0; k < obs->m; ++k)
        for (int e = obs->row_ptr[k]; e < obs->row_ptr[k + 1]; ++e) {
            if (obs->col_idx[e] < 0 || obs->col_idx[e] >= obs->n) { set_error("observable matrix: column out of range"); return -1; }
            mask[obs->col_idx[e]] ^= 1u << k;
        }
    if (d->d_obs.reserve(mask.size() * 4)) return -1;
    SWD_HIP(hipMemcpy(d->d_obs.p, mask.data(), mask.size() * 4, hipMemcpyHostToDevice));
    return 0;
}

extern "C" int swd_pipeline_decode_dev(swd_pipeline *h, int32_t B, const uint8_t *det, int64_t det_stride,
                                       uint8_t *total, int64_t total_stride, int32_t *stats, double *min_pm,
                                       int32_t *shot_result, void *stream) {
    Plan *d = (Plan *)h;
    if (!d) { set_error("null pipeline"); return -1; }
    if (B <= 0) return 0;
    if (!det || !total) { set_error("null output/input pointer"); return -1; }
    SWD_HIP(hipSetDevice(d->device));
    SwdPipeArgs a{};
    a.wins = d->d_wins.as<SwdWindowDev>(); a.W = (int)d->wins.size(); a.B = B;
    a.slot_scratch = a.W > 1 ? 1 : 0;
    fill_params(d, a.P, false, false);
    a.det = det; a.det_stride = det_stride ? det_stride : d->num_det; a.num_det = d->num_det; a.off_det = d->off_det;
    a.total = total; a.total_stride = total_stride ? total_stride : d->num_col;
    a.chk_colptr = d->d_colptr; a.chk_rows = d->d_rows;
    a.win_out = nullptr; a.stats = stats; a.min_pm = min_pm;
    a.hist = nullptr; a.hist_stride = 4 * (int64_t)d->nmax; a.osd0 = nullptr;
    a.obs_mask = d->d_obs.p ? d->d_obs.as<uint32_t>() : nullptr;
    a.shot_result = shot_result;
    if (d->profiling) {
        if (d->prof.reserve((size_t)B * a.W * 8 * sizeof(int64_t))) return -1;
        a.prof = d->prof.as<int64_t>();
    }
    return launch(d, a, (hipStream_t)stream);
}

extern "C" int swd_pipeline_status(swd_pipeline *h, uint32_t *flags) {
    Plan *d = (Plan *)h;
    if (!d || !flags) { set_error("null argument"); return -1; }
    std::lock_guard<std::recursive_mutex> lk(d->mu);
    SWD_HIP(hipSetDevice(d->device));
    SWD_HIP(hipDeviceSynchronize());
    SWD_HIP(hipMemcpy(flags, d->status.p, 4, hipMemcpyDeviceToHost));
    if (*flags) SWD_HIP(hipMemset(d->status.p, 0, 4)); // read-and-clear
    return 0;
}

// diagnostics (development builds with -DSWD_GDG_DEBUG): the 16 status words, read and cleared
extern "C" int swd_pipeline_debug_counters(swd_pipeline *h, uint32_t *out16) {
    Plan *d = (Plan *)h;
    if (!d || !out16) { set_error("null argument"); return -1; }
    std::lock_guard<std::recursive_mutex> lk(d->mu);
    SWD_HIP(hipSetDevice(d->device));
    SWD_HIP(hipDeviceSynchronize());
    SWD_HIP(hipMemcpy(out16, d->status.p, 64, hipMemcpyDeviceToHost));
    SWD_HIP(hipMemset((char *)d->status.p + 4, 0, 60));
    return 0;
}

// ------------------------------------------------------------------------------------------
// streaming form (include/swd.h: swd_pipeline_stream_*) and the host-buffer entry points built on it
// ------------------------------------------------------------------------------------------
namespace swd {

// total_e_hat bytes -> bits: thread = one output byte (columns 8 j .. 8 j + 7 of one shot), bit k = column 8 j + k
__global__ void __launch_bounds__(256) pack_bits_kernel(const uint8_t *total, int64_t stride, int num_col, int B, uint8_t *bits, int row_bytes) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)B * row_bytes) return;
    const int b = (int)(t / row_bytes), j = (int)(t - (long long)b * row_bytes);
    const uint8_t *src = total + (int64_t)b * stride + 8 * j;
    uint32_t x = 0;
    if (8 * j + 8 <= num_col && (((uintptr_t)src) & 7) == 0) {
        const uint64_t w = *(const uint64_t *)src; // bytes 0/1 -> bits: gather the low bit of every byte
        x = (uint32_t)((((w & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56) & 0xFFu);
    } else {
        for (int k = 0; k < 8 && 8 * j + k < num_col; ++k) x |= (src[k] ? 1u : 0u) << k;
    }
    bits[t] = (uint8_t)x;
}

// bits -> bytes on the host: one 8-byte store per packed byte; rows dealt to a few threads when the output is large
static void unpack_rows(const uint8_t *bits, size_t row_bytes, int num_col, uint8_t *out, size_t r0, size_t r1) {
    static uint64_t lut[256];
    static std::once_flag once;
    std::call_once(once, [] {
        for (int v = 0; v < 256; ++v) {
            uint64_t w = 0;
            for (int k = 0; k < 8; ++k) w |= (uint64_t)((v >> k) & 1) << (8 * k);
            lut[v] = w;
        }
    });
    const int full = num_col / 8;
    for (size_t r = r0; r < r1; ++r) {
        const uint8_t *src = bits + r * row_bytes;
        uint8_t *dst = out + r * (size_t)num_col;
        for (int j = 0; j < full; ++j) memcpy(dst + 8 * j, &lut[src[j]], 8);
        for (int c = full * 8; c < num_col; ++c) dst[c] = (src[c >> 3] >> (c & 7)) & 1;
    }
}
static void unpack_bits_host(const uint8_t *bits, size_t B, int num_col, uint8_t *out) {
    const size_t row_bytes = ((size_t)num_col + 7) / 8;
    unsigned nthr = 1;
    if (B * (size_t)num_col >= (8u << 20)) nthr = std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
    if (nthr <= 1) { unpack_rows(bits, row_bytes, num_col, out, 0, B); return; }
    std::vector<std::thread> th;
    for (unsigned i = 1; i < nthr; ++i) th.emplace_back(unpack_rows, bits, row_bytes, num_col, out, B * i / nthr, B * (i + 1) / nthr);
    unpack_rows(bits, row_bytes, num_col, out, 0, B / nthr);
    for (auto &t : th) t.join();
}

static int stream_alloc_lane(HostStream *hs, StreamLane &l) {
    Plan *d = hs->plan;
    const size_t B = (size_t)hs->max_shots, W = d->wins.size();
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t row_bytes = ((size_t)d->num_col + 7) / 8;
    const bool stats = !(hs->flags & SWD_STREAM_NO_STATS);
    // device: det | total bytes | then the block that travels back in one copy: bits | stats | min_pm | shot_result | status
    l.o_total = al(B * d->num_det);
    l.o_bits = l.o_total + al(B * (size_t)d->num_col);
    l.h_stats = al(B * row_bytes);
    l.h_pm = l.h_stats + (stats ? al(B * W * SWD_STAT_WORDS * 4) : 0);
    l.h_shot = l.h_pm + (stats ? al(B * W * 8) : 0);
    l.h_status = l.h_shot + al(B * 8);
    l.out_bytes = l.h_status + 256;
    l.o_stats = l.o_bits + l.h_stats; l.o_pm = l.o_bits + l.h_pm; l.o_shot = l.o_bits + l.h_shot; l.o_status = l.o_bits + l.h_status;
    l.dev_bytes = l.o_bits + l.out_bytes;
    l.allocated = false;
    if (l.dev.reserve(l.dev_bytes) || l.hin.reserve(B * d->num_det) || l.hout.reserve(l.out_bytes)) return -1;
    l.allocated = true;
    return 0;
}

// The second lane is a high-priority stream: the runtime deals streams of ONE priority onto a few hardware queues per device and lets
// later streams share them (GPU_MAX_HW_QUEUES, default 4; which queue a stream shares is not ours to choose), and two lanes on one
// queue run their launches back to back -- seen with pairs of plain streams (scripts/hwq_probe.py: 1.83 instead of 1.12 ms per step).
// Queues are pooled per priority, so lanes of different priority never share one.  The lanes' launches are persistent grids that
// take the slots the other launch leaves free either way; the priority only decides which of them gets a freed slot first.
static int stream_lane_init(StreamLane &l, bool high_priority) {
    if (!l.st) {
        int least = 0, greatest = 0;
        if (high_priority && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest < least && !getenv("SWD_STREAM_SAME_PRIORITY"))
            SWD_HIP(hipStreamCreateWithPriority(&l.st, hipStreamNonBlocking, greatest));
        else SWD_HIP(hipStreamCreateWithFlags(&l.st, hipStreamNonBlocking));
    }
    if (!l.done) SWD_HIP(hipEventCreateWithFlags(&l.done, hipEventDisableTiming));
    if (!l.ready) SWD_HIP(hipEventCreateWithFlags(&l.ready, hipEventDisableTiming));
    return 0;
}

static HostStream *stream_new(Plan *d, int max_shots, int flags) {
    if (d->wins.empty() || d->num_col <= 0) { set_error("streaming needs a sliding-window pipeline"); return nullptr; }
    if (max_shots <= 0) { set_error("max_shots must be positive"); return nullptr; }
    if (hipSetDevice(d->device) != hipSuccess) { set_error("hipSetDevice(%d) failed", d->device); return nullptr; }
    HostStream *hs = new HostStream();
    hs->plan = d; hs->max_shots = max_shots; hs->flags = flags; hs->device = d->device;
    for (int i = 0; i < 2; ++i)
        if (stream_lane_init(hs->lane[i], i == 1)) { delete hs; return nullptr; }
    return hs;
}

// the plan a stream call works on; NULL (with a message) once the pipeline has been destroyed
static Plan *stream_plan(HostStream *hs) {
    if (!hs->plan) set_error("the pipeline of this stream object has been destroyed");
    return hs->plan;
}

static int stream_push(HostStream *hs, int B, const uint8_t *det) {
    std::lock_guard<std::mutex> lk(hs->mu);
    Plan *d = stream_plan(hs);
    if (!d) return -1;
    if (B <= 0 || B > hs->max_shots) { set_error("stream push: %d shots, the stream was created for 1..%d", B, hs->max_shots); return -1; }
    if (!det) { set_error("null input pointer"); return -1; }
    StreamLane &l = hs->lane[hs->npush & 1];
    if (l.busy) { set_error("stream push: two batches are in flight, pop the oldest first"); return -1; }
    SWD_HIP(hipSetDevice(d->device));
    if (!l.allocated && stream_alloc_lane(hs, l)) return -1;
    const size_t W = d->wins.size(), row_bytes = ((size_t)d->num_col + 7) / 8;
    const bool stats = !(hs->flags & SWD_STREAM_NO_STATS);
    char *dv = (char *)l.dev.p;
    memcpy(l.hin.p, det, (size_t)B * d->num_det);
    SWD_HIP(hipMemcpyAsync(dv, l.hin.p, (size_t)B * d->num_det, hipMemcpyHostToDevice, l.st));
    const uint32_t *fault = nullptr; // this launch's own fault word (launch(): sched[B + 1] of its launch slot)
    {
        std::lock_guard<std::recursive_mutex> lkp(d->mu);
        d->stream_push = hs->overlapping; // (a synchronous host call in one part has nothing in flight beside it)
        int rc = swd_pipeline_decode_dev((swd_pipeline *)d, B, (const uint8_t *)dv, 0, (uint8_t *)(dv + l.o_total), 0,
                                         stats ? (int32_t *)(dv + l.o_stats) : nullptr, stats ? (double *)(dv + l.o_pm) : nullptr,
                                         (int32_t *)(dv + l.o_shot), l.st);
        d->stream_push = false;
        if (rc) return rc;
        fault = d->cur->sched.as<uint32_t>() + B + 1;
        // the fault word of THIS launch (batches in flight on the other lane have their own).  The copy reads the launch slot, so it
        // belongs to the slot's use: still under the plan's lock -- nobody else can have taken the slot yet -- the slot's `done` event
        // is recorded again behind it.  A later launch that re-uses the slot from any stream or stream object waits for that event
        // before its hipMemsetAsync clears the word.
        SWD_HIP(hipMemcpyAsync(dv + l.o_status, fault, 4, hipMemcpyDeviceToDevice, l.st));
        SWD_HIP(hipEventRecord(d->cur->done, l.st));
    }
    const long long nb = (long long)B * (long long)row_bytes;
    hipLaunchKernelGGL(pack_bits_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, l.st, (const uint8_t *)(dv + l.o_total),
                       (int64_t)d->num_col, d->num_col, B, (uint8_t *)(dv + l.o_bits), (int)row_bytes);
    SWD_HIP(hipGetLastError());
    // what travels back: only the used prefix of every array (four copies into the page-locked block)
    char *ho = (char *)l.hout.p;
    SWD_HIP(hipMemcpyAsync(ho, dv + l.o_bits, (size_t)nb, hipMemcpyDeviceToHost, l.st));
    if (stats) {
        SWD_HIP(hipMemcpyAsync(ho + l.h_stats, dv + l.o_stats, (size_t)B * W * SWD_STAT_WORDS * 4, hipMemcpyDeviceToHost, l.st));
        SWD_HIP(hipMemcpyAsync(ho + l.h_pm, dv + l.o_pm, (size_t)B * W * 8, hipMemcpyDeviceToHost, l.st));
    }
    SWD_HIP(hipMemcpyAsync(ho + l.h_shot, dv + l.o_shot, (size_t)B * 8, hipMemcpyDeviceToHost, l.st));
    SWD_HIP(hipMemcpyAsync(ho + l.h_status, dv + l.o_status, 4, hipMemcpyDeviceToHost, l.st));
    SWD_HIP(hipEventRecord(l.done, l.st));
    l.B = B; l.busy = true;
    hs->npush++;
    return 0;
}

// packed_out: the caller's `total` takes the packed rows whatever the stream's flag says (swd_pipeline_decode_packed)
static int stream_pop(HostStream *hs, uint8_t *total, int32_t *stats, double *min_pm, int32_t *shot_result, int packed_out) {
    std::lock_guard<std::mutex> lk(hs->mu);
    Plan *d = stream_plan(hs);
    if (!d) return -1;
    StreamLane &l = hs->lane[hs->npop & 1];
    if (!l.busy) { set_error("stream pop: no batch in flight"); return -1; }
    // the batch leaves the stream whatever happens below (a caller draining `pending` must terminate on a persistent device error)
    l.busy = false;
    hs->npop++;
    SWD_HIP(hipSetDevice(d->device));
    SWD_HIP(hipEventSynchronize(l.done));
    const size_t B = (size_t)l.B, W = d->wins.size(), row_bytes = ((size_t)d->num_col + 7) / 8;
    const char *ho = (const char *)l.hout.p;
    const uint32_t err = *(const uint32_t *)(ho + l.h_status);
    if (err) { // (the decoder's sticky word keeps the flag for swd_pipeline_status)
        set_error("internal: a window waited more than 10 s for its predecessor (scheduling fault, flags 0x%x)", err);
        return -1;
    }
    if ((stats || min_pm) && (hs->flags & SWD_STREAM_NO_STATS)) { set_error("stream pop: the stream was created with SWD_STREAM_NO_STATS"); return -1; }
    if (total) {
        if (packed_out || (hs->flags & SWD_STREAM_PACKED)) memcpy(total, ho, B * row_bytes);
        else unpack_bits_host((const uint8_t *)ho, B, d->num_col, total);
    }
    if (stats) memcpy(stats, ho + l.h_stats, B * W * SWD_STAT_WORDS * 4);
    if (min_pm) memcpy(min_pm, ho + l.h_pm, B * W * 8);
    if (shot_result) memcpy(shot_result, ho + l.h_shot, B * 8);
    return (int)B;
}

// One host-buffer call on the plan's own stream object.  A very large batch is cut into parts that go through the two lanes one
// after the other, two in flight: the copy-out and the unpacking of part k overlap the launch of part k + 1 (whose grid also fills the
// tail of part k), so only the last part's unpacking is left after the device has finished.
static int pipeline_decode_host(Plan *d, int32_t B, const uint8_t *det, uint8_t *total, int32_t *stats, double *min_pm,
                                int32_t *shot_result, int packed) {
    std::lock_guard<std::recursive_mutex> lk(d->mu);
    SWD_HIP(hipSetDevice(d->device));
    static const int split_min = getenv("SWD_HOST_SPLIT_MIN") ? atoi(getenv("SWD_HOST_SPLIT_MIN")) : 2048; // shots from which a call is cut
    // (measured, 4096 shots of the headline workload, OSD-CS 10: one part 13.8 ms per call, two 17.5, four 22.1, eight 27.1 -- a launch of
    // 1024 shots is bound by its longest shot's chain of eleven windows, not by the device; only calls of 16384 shots and more are cut)
    static const int want_parts = getenv("SWD_HOST_PARTS") ? std::max(1, atoi(getenv("SWD_HOST_PARTS"))) : 0;
    int parts = want_parts ? ((B >= split_min) ? want_parts : 1) : std::max(1, B / 8192);
    while (parts > 1 && B / parts < 512) --parts; // (a part keeps the device busy)
    const int per = (B + parts - 1) / parts;
    if (!d->hstream || d->hstream->max_shots < per) {
        d->hstream.reset(stream_new(d, std::max(per, 256), 0));
        if (!d->hstream) return -1;
        d->hstream->owned_by_plan = true;
    }
    HostStream *hs = d->hstream.get();
    hs->overlapping = parts > 1;
    while (hs->npop < hs->npush) (void)stream_pop(hs, nullptr, nullptr, nullptr, nullptr, 0); // (a failed earlier call left batches behind)
    const size_t W = d->wins.size(), row_bytes = ((size_t)d->num_col + 7) / 8;
    auto push = [&](int k) { const int lo = k * per, n = std::min(per, B - lo); return n > 0 ? stream_push(hs, n, det + (size_t)lo * d->num_det) : 0; };
    auto pop = [&](int k) {
        const size_t lo = (size_t)k * per;
        if ((int)lo >= B) return 0;
        return stream_pop(hs, total + lo * (packed ? row_bytes : (size_t)d->num_col), stats ? stats + lo * W * SWD_STAT_WORDS : nullptr,
                          min_pm ? min_pm + lo * W : nullptr, shot_result ? shot_result + lo * 2 : nullptr, packed) < 0 ? -1 : 0;
    };
    int rc = 0;
    for (int k = 0; k < parts && k < 2; ++k) if (push(k)) return -1;
    for (int k = 0; k < parts; ++k) {
        if (pop(k)) rc = -1;
        if (k + 2 < parts && push(k + 2)) return -1;
    }
    return rc;
}

} // namespace swd

extern "C" int swd_pipeline_decode(swd_pipeline *h, int32_t B, const uint8_t *det, uint8_t *total, int32_t *stats,
                                   double *min_pm, int32_t *shot_result) {
    Plan *d = (Plan *)h;
    if (!d) { set_error("null pipeline"); return -1; }
    if (B <= 0) return 0;
    if (!det || !total) { set_error("null output/input pointer"); return -1; }
    return pipeline_decode_host(d, B, det, total, stats, min_pm, shot_result, 0);
}

extern "C" int swd_pipeline_decode_packed(swd_pipeline *h, int32_t B, const uint8_t *det, uint8_t *total_bits, int32_t *stats,
                                          double *min_pm, int32_t *shot_result) {
    Plan *d = (Plan *)h;
    if (!d) { set_error("null pipeline"); return -1; }
    if (B <= 0) return 0;
    if (!det || !total_bits) { set_error("null output/input pointer"); return -1; }
    return pipeline_decode_host(d, B, det, total_bits, stats, min_pm, shot_result, 1);
}

extern "C" swd_stream *swd_pipeline_stream_create(swd_pipeline *h, int32_t max_shots, int32_t flags) {
    Plan *d = (Plan *)h;
    if (!d) { set_error("null pipeline"); return nullptr; }
    HostStream *hs = stream_new(d, max_shots, flags);
    if (hs) { std::lock_guard<std::recursive_mutex> lk(d->mu); d->streams.push_back(hs); }
    return (swd_stream *)hs;
}

extern "C" void swd_pipeline_stream_destroy(swd_stream *s) {
    HostStream *hs = (HostStream *)s;
    if (!hs) return;
    (void)hipSetDevice(hs->device); // also for a detached stream: its lanes live on the device of the pipeline it was made for
    if (Plan *d = hs->plan) { // (a stream whose pipeline went first was detached by ~Plan and only frees its own lanes)
        std::lock_guard<std::recursive_mutex> lk(d->mu);
        d->streams.erase(std::remove(d->streams.begin(), d->streams.end(), hs), d->streams.end());
    }
    delete hs;
}

extern "C" int swd_pipeline_stream_push(swd_stream *s, int32_t B, const uint8_t *det) {
    HostStream *hs = (HostStream *)s;
    if (!hs) { set_error("null stream"); return -1; }
    return stream_push(hs, B, det);
}

extern "C" int swd_pipeline_stream_pop(swd_stream *s, uint8_t *total, int32_t *stats, double *min_pm, int32_t *shot_result) {
    HostStream *hs = (HostStream *)s;
    if (!hs) { set_error("null stream"); return -1; }
    return stream_pop(hs, total, stats, min_pm, shot_result, 0);
}

extern "C" int swd_pipeline_stream_pending(swd_stream *s) {
    HostStream *hs = (HostStream *)s;
    if (!hs) { set_error("null stream"); return -1; }
    std::lock_guard<std::mutex> lk(hs->mu);
    return (int)(hs->npush - hs->npop);
}

extern "C" int swd_pipeline_stream_push_dev(swd_stream *s, int32_t B, const uint8_t *det, int64_t det_stride, uint8_t *total,
                                            int64_t total_stride, int32_t *stats, double *min_pm, int32_t *shot_result, void *after) {
    HostStream *hs = (HostStream *)s;
    if (!hs) { set_error("null stream"); return -1; }
    std::lock_guard<std::mutex> lk(hs->mu);
    if (hs->npush != hs->npop) { set_error("stream push_dev: host batches are in flight on this stream object"); return -1; }
    Plan *d = stream_plan(hs);
    if (!d) return -1;
    SWD_HIP(hipSetDevice(d->device));
    StreamLane &l = hs->lane[hs->npush & 1];
    // The producer of the inputs (and whoever still reads the lane's output buffers) comes first.  The lanes are non-blocking
    // streams: they do not synchronise with the legacy default stream by themselves, so after == NULL -- what a framework hands
    // over for "the default stream" -- is an event on that stream like any other; only the sentinel skips the wait.
    if (after != SWD_STREAM_NO_DEPENDENCY) {
        SWD_HIP(hipEventRecord(l.ready, (hipStream_t)after));
        SWD_HIP(hipStreamWaitEvent(l.st, l.ready, 0));
    }
    int rc;
    {
        std::lock_guard<std::recursive_mutex> lkp(d->mu);
        d->stream_push = true;
        rc = swd_pipeline_decode_dev((swd_pipeline *)d, B, det, det_stride, total, total_stride, stats, min_pm, shot_result, l.st);
        d->stream_push = false;
    }
    if (rc) return rc;
    SWD_HIP(hipEventRecord(l.done, l.st));
    hs->npush++; hs->npop++; // (nothing to pop: the results are the caller's device buffers)
    return 0;
}

extern "C" int swd_pipeline_stream_wait(swd_stream *s, void *stream) {
    HostStream *hs = (HostStream *)s;
    if (!hs) { set_error("null stream"); return -1; }
    std::lock_guard<std::mutex> lk(hs->mu);
    if (!stream_plan(hs)) return -1;
    SWD_HIP(hipSetDevice(hs->plan->device));
    for (auto &l : hs->lane) {
        if (stream) {
            SWD_HIP(hipEventRecord(l.ready, l.st));
            SWD_HIP(hipStreamWaitEvent((hipStream_t)stream, l.ready, 0));
        } else SWD_HIP(hipStreamSynchronize(l.st));
    }
    return 0;
}

// diagnostics: per-phase device timers of the last pipeline launch (100 MHz ticks), [B][W][8]
extern "C" int swd_pipeline_set_profiling(swd_pipeline *h, int32_t on) {
    Plan *d = (Plan *)h;
    if (!d) { set_error("null pipeline"); return -1; }
    d->profiling = on != 0;
    return 0;
}
extern "C" int swd_pipeline_get_profile(swd_pipeline *h, int32_t B, int64_t *out) {
    Plan *d = (Plan *)h;
    if (!d || !out) { set_error("null argument"); return -1; }
    SWD_HIP(hipSetDevice(d->device));
    SWD_HIP(hipDeviceSynchronize());
    SWD_HIP(hipMemcpy(out, d->prof.p, (size_t)B * d->wins.size() * 8 * sizeof(int64_t), hipMemcpyDeviceToHost));
    return 0;
}

// ------------------------------------------------------------------------------------------
// guessing decoders (bpgdg_decoder / bpgd_decoder / bp_history_decoder)
// ------------------------------------------------------------------------------------------
static int check_gdg_params(const swd_gdg_params &gp) {
    if (gp.mode < 0 || gp.mode > 2) { set_error("gdg mode %d invalid (0 bpgdg, 1 bpgd, 2 bp_history)", gp.mode); return -1; }
    if (gp.multi_thread < 0 || gp.multi_thread > 2) { set_error("invalid guessing-decoder parameters: multi_thread"); return -1; }
    if (gp.max_iter_per_step < 0 || gp.max_step < 0 || gp.max_tree_depth < 0 || gp.max_tree_depth > 6) {
        set_error("invalid guessing-decoder parameters"); return -1;
    }
    return 0;
}

extern "C" swd_gdg *swd_gdg_create(const swd_graph_desc *g, const swd_gdg_params *gp, int device) {
    if (!gp || !g) { set_error("null argument"); return nullptr; }
    if (check_device(device) || check_gdg_params(*gp)) return nullptr;
    Plan *d = new Plan();
    d->device = device;
    d->gp = *gp;
    d->kind = gp->mode + 1;
    d->p.pre_max_iter = gp->max_iter; d->p.osd_order = -1; d->p.ms_scaling_factor = gp->ms_scaling_factor;
    std::map<std::string, std::shared_ptr<Graph>> cache;
    if (d->add_window(g, 0, 0, 0, cache) || d->finalize(nullptr)) { delete d; return nullptr; }
    return (swd_gdg *)d;
}

extern "C" void swd_gdg_destroy(swd_gdg *h) { swd_osdw_destroy((swd_osdw *)h); }

extern "C" int swd_gdg_decode_batch(swd_gdg *h, int32_t B, const uint8_t *synd, uint8_t *out, int32_t *stats,
                                    double *min_pm, double *hist, int32_t hist_is_state) {
    return swd_osdw_decode_batch((swd_osdw *)h, B, synd, out, stats, min_pm, hist, hist_is_state, nullptr, nullptr);
}

extern "C" int swd_gdg_decode_batch_dev(swd_gdg *h, int32_t B, const uint8_t *synd, int64_t synd_stride, uint8_t *out,
                                        int64_t out_stride, int32_t *stats, double *min_pm, void *stream) {
    return swd_osdw_decode_batch_dev((swd_osdw *)h, B, synd, synd_stride, out, out_stride, stats, min_pm, nullptr, 0,
                                     nullptr, nullptr, stream);
}

extern "C" swd_pipeline *swd_pipeline_create_gdg(int32_t num_windows, const swd_window_desc *wins,
                                                 const swd_graph_desc *chk, const swd_gdg_params *gp, int device) {
    if (!gp || !wins || !chk || num_windows <= 0) { set_error("null argument"); return nullptr; }
    if (check_device(device) || check_gdg_params(*gp)) return nullptr;
    Plan *d = new Plan();
    d->device = device;
    d->gp = *gp;
    d->kind = gp->mode + 1;
    d->p.pre_max_iter = gp->max_iter; d->p.osd_order = -1; d->p.ms_scaling_factor = gp->ms_scaling_factor;
    std::map<std::string, std::shared_ptr<Graph>> cache;
    for (int i = 0; i < num_windows; ++i)
        if (d->add_window(&wins[i].graph, wins[i].row0, wins[i].col0, wins[i].commit, cache)) { delete d; return nullptr; }
    if (d->finalize(chk)) { delete d; return nullptr; }
    return (swd_pipeline *)d;
}
