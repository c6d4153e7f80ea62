// Host-side plan of a decode launch (shared by the translation units that instantiate the kernels):
// the windows of a plan, the launch-slot ring, and launch_nt, which sizes the persistent grid and launches one
// instantiation of swd::pipeline_kernel.  The kernels are instantiated in swd_kernels_*.hip (one file per kind, so
// that they compile in parallel); swd_osdw.hip holds the C ABI and the variant table.
#pragma once
#include <string.h>

#include <map>
#include <memory>
#include <mutex>

#include <stdlib.h>

#include "swd_host.h"
#include "swd_osdw_kernel.h"

namespace swd {

inline int next_pow2(int x) { int p = 1; while (p < x) p <<= 1; return p; }
inline int align_up(int x, int a) { return (x + a - 1) / a * a; }

int make_layout(const Graph &g, int new_n, int nt, int kind, SwdLdsLayout &L, bool big = false, int lds_budget = 0);

struct WindowHost {
    std::shared_ptr<Graph> g;
    int new_n = 0, row0 = 0, col0 = 0, commit = 0;
    SwdLdsLayout L{};
};

struct Plan;
// Streaming form of the window loop (include/swd.h: swd_pipeline_stream_*): two lanes, each with its own HIP stream, device
// buffers and page-locked staging, so that the copies and the host-side unpacking of batch k overlap the launch of batch k + 1 and
// the persistent grid of k + 1 fills the workgroup slots the tail of k leaves empty.
struct StreamLane {
    hipStream_t st = nullptr;
    hipEvent_t done = nullptr, ready = nullptr;
    int B = 0;
    bool busy = false;       // a host batch is in flight on this lane (pushed, not popped)
    bool allocated = false;  // dev / hin / hout all hold what stream_alloc_lane sized (set after the three reserves succeeded)
    DevBuf dev;              // [ det | total bytes | bits | stats | min_pm | shot_result | status copy ]
    PinnedBuf hin, hout;     // det in; [ bits | stats | min_pm | shot_result | status ] out
    size_t o_total = 0, o_bits = 0, o_stats = 0, o_pm = 0, o_shot = 0, o_status = 0, dev_bytes = 0; // offsets inside dev
    size_t h_stats = 0, h_pm = 0, h_shot = 0, h_status = 0, out_bytes = 0;                            // offsets inside hout
};
struct HostStream {
    Plan *plan = nullptr;    // nulled by ~Plan when the pipeline is destroyed first: every later call on the stream fails cleanly
    int max_shots = 0, flags = 0;
    int device = 0;          // the plan's device, kept here so that a stream detached from its pipeline still frees its lanes on it
    StreamLane lane[2];
    bool overlapping = true; // batches of this object overlap (false: the plan's own object while it serves a host call in ONE part)
    long long npush = 0, npop = 0;
    bool owned_by_plan = false;
    std::mutex mu;
    ~HostStream() {
        (void)hipSetDevice(device); // (streams, events and the lanes' buffers belong to this device whatever is current in a multi-GPU process)
        for (auto &l : lane) {
            if (l.st) { (void)hipStreamSynchronize(l.st); (void)hipStreamDestroy(l.st); }
            if (l.done) (void)hipEventDestroy(l.done);
            if (l.ready) (void)hipEventDestroy(l.ready);
        }
    }
};
// Kernel variants: threads per shot, VNs per thread, column-degree bound, groups of four row positions.
// A plan uses the first variant with NT >= m, NT*VF >= n, DM >= D, 4*KG >= K over all its windows.
struct Variant {
    int nt, vf, dm, kg;
    int sf; // full-graph phase shares heavy checks among threads (needs the host-built map): 4 * kg may be < K
    int (*launch)(Plan *, const SwdPipeArgs &, hipStream_t);      // osd_window kernels (kind 0)
    int (*launch_gdg)(Plan *, const SwdPipeArgs &, hipStream_t);  // guessing decoders, serial tree walk (kind 1)
    int (*launch_par)(Plan *, const SwdPipeArgs &, hipStream_t);  // guessing decoders, side branches as work items (kind 2)
    int (*launch_acc)(Plan *, const SwdPipeArgs &, hipStream_t);  // osd_window, posterior history accumulated in registers (kind 3)
    int (*launch_ens)(Plan *, const SwdPipeArgs &, hipStream_t);  // bpgdg_decoder(multi_thread=True): the reference's threaded ensemble (kind 7)
};
const Variant *select_variant(const std::vector<WindowHost> &wins, int mmax, int nmax, int dm, int kmax, int kind);
const Variant *select_big_variant(int mmax, int nmax, int dm, int kmax, int kind = 0); // graphs beyond one CU's LDS: scratch region in HBM
bool split_map(const Graph &g, int nt, int cap, std::vector<uint32_t> *map);



// The general form of osd_window for graphs beyond every kernel variant (swd_huge.hip): every array in HBM, several checks / nodes
// per thread.  A Plan that carries one has no windows; the single-window entry points route to it.
struct HugeIface {
    int m = 0, n = 0, new_n = 0, rank = 0;
    virtual ~HugeIface() {}
    virtual int decode_dev(int32_t B, const uint8_t *synd, int64_t synd_stride, uint8_t *out, int64_t out_stride, int32_t *stats,
                           double *min_pm, double *hist, int32_t hist_is_state, uint8_t *osd0, uint8_t *bp_dec, void *stream) = 0;
};
HugeIface *huge_create(const swd_graph_desc *g, const swd_osdw_params *p, int device);

struct Plan {
    std::unique_ptr<HugeIface> huge;
    int m0() const { return huge ? huge->m : wins[0].g->m; }
    int n0() const { return huge ? huge->n : wins[0].g->n; }
    std::vector<WindowHost> wins;
    swd_osdw_params p{};
    swd_gdg_params gp{};
    int kind = 0;           // 0 osd_window, 1 bpgdg, 2 bpgd, 3 bp_history
    bool gdg_parallel = false; // bpgdg: side branches of the decimation tree run as work items on the persistent grid
    bool stream_push = false;  // (under mu) the launch being prepared comes from a stream object: another batch follows or is in flight
    bool stream_serial = false; // (under mu) ... and is large enough for the guessing decoders' ticket-scheduled forms (launch())
    hipEvent_t last_done = nullptr; // (under mu) end of the most recent launch and its stream: a launch that finds it still running on
    hipStream_t last_stream = nullptr; // another stream is a stream batch in all but name
    int new_n_max = 0;
    int max_guess = 0;
    int64_t snap_stride = 0;
    DevBuf snap;
    int device = 0, nt = 256, vf = 7, dm = 8;
    const Variant *variant = nullptr;
    int num_det = 0, num_col = 0, nmax = 0, off_det = 0, lds_total = 0;
    bool post_depth2 = false;  // guessing decoders: every window keeps new_n <= 2 nt columns -> depth-2 cache for the shortened graph
    bool big = false;          // large graphs: the layouts' scratch region lives in HBM (big_stride bytes per workgroup)
    int64_t big_stride = 0;
    DevBuf d_wins, d_chk, d_obs, d_cnmap;
    DevBuf shot;
    const uint32_t *d_colptr = nullptr;
    const uint16_t *d_rows = nullptr;
    // host-pointer staging (the host-buffer entry points hold `mu` for their whole duration)
    DevBuf synd, out, stats, pm, hist, osd0, total;
    DevBuf prof, io;
    PinnedBuf stage;
    // Scratch a launch writes and reads back -- ticket counter + per-shot progress, the window hand-over
    // records, the per-workgroup history ring and snapshot stack -- comes from a small ring of launch slots, so
    // launches of one decoder on different streams (or from different host threads) never share it: a launch
    // that re-uses a slot first makes its stream wait for the slot's previous launch (hipStreamWaitEvent).
    struct LaunchSlot {
        DevBuf sched, state, hist, snap, gq, gfq, gctx, gsnap, big, order;
        DevBuf gfree_tmpl;     // a full free-context ring, copied into the slot's ring per launch (per slot: a launch on another
        int gfree_n = 0;       // stream may still be copying from the template of ITS slot while this one is rebuilt)
        int gfree_avail = -1;  // contexts the template hands out (KIND 2: depends on the batch size)
        hipEvent_t done = nullptr;
    };
    static constexpr int kSlots = 4;
    LaunchSlot slot[kSlots];
    LaunchSlot *cur = nullptr; // slot of the launch being prepared (valid under mu)
    int next_slot = 0;
    std::recursive_mutex mu;
    DevBuf status;             // one word, never reset by a launch: scheduling faults (swd_pipeline_status)
    bool profiling = false;
    bool timing = false;
    double t_total_ms = 0;
    int64_t t_launches = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::unique_ptr<HostStream> hstream; // the host-buffer entry point swd_pipeline_decode runs on its own two-lane stream object
    std::vector<HostStream *> streams;   // live stream objects of the caller (swd_pipeline_stream_create), under mu

    ~Plan() {
        (void)hipSetDevice(device);
        hstream.reset();
        // stream objects that outlive their pipeline: drain their lanes (they read this plan's buffers) and detach them.  The list is
        // taken out under the plan's lock, so a swd_pipeline_stream_destroy running on another thread either removed its object
        // before (it is not in `live`) or finds the list empty afterwards.
        std::vector<HostStream *> live;
        {
            std::lock_guard<std::recursive_mutex> lkp(mu);
            live.swap(streams);
        }
        for (HostStream *hs : live) {
            std::lock_guard<std::mutex> lk(hs->mu);
            for (auto &l : hs->lane) { if (l.st) (void)hipStreamSynchronize(l.st); l.busy = false; }
            hs->npop = hs->npush;
            hs->plan = nullptr;
        }
        if (ev0) { (void)hipEventDestroy(ev0); (void)hipEventDestroy(ev1); }
        for (auto &sl : slot) if (sl.done) (void)hipEventDestroy(sl.done);
    }

    int add_window(const swd_graph_desc *gd, int row0, int col0, int commit,
                   std::map<std::string, std::shared_ptr<Graph>> &cache) {
        // identical window matrices (the mid windows of a memory experiment are translates of one
        // another) share one device graph so the hot read-only data stays small in L2
        std::string key((const char *)gd->row_ptr, (size_t)(gd->m + 1) * 4);
        key.append((const char *)gd->col_idx, (size_t)gd->nnz * 4);
        key.append((const char *)gd->channel_probs, (size_t)gd->n * 8);
        WindowHost w;
        auto itc = cache.find(key);
        if (itc != cache.end()) w.g = itc->second;
        else {
            w.g = std::make_shared<Graph>();
            if (w.g->build(gd)) return -1;
            if (w.g->D > SWD_DMAX) { set_error("column weight %d exceeds this build's bound %d", w.g->D, SWD_DMAX); return -1; }
            if (w.g->upload()) return -1;
            cache[key] = w.g;
        }
        const int m = w.g->m, n = w.g->n;
        const int req_new_n = (kind == 0) ? p.new_n : gp.new_n;
        w.new_n = (req_new_n <= 0) ? std::min(n, 2 * m) : std::min(req_new_n, n); // osd_window.pyx:60-63
        if (kind == 0 && p.osd_order > w.new_n - w.g->rank) {                              // osd_window.pyx:88-92
            set_error("For this code, the OSD order should be set in the range 0<=osd_oder<=%d.", w.new_n - w.g->rank);
            return -1;
        }
        if (commit < 0 || commit > n || row0 < 0 || col0 < 0) { // osd.py:140,170-173: e_hat[:commit] is a slice of the window's estimate
            set_error("invalid window placement: row0 %d col0 %d commit %d for a window with %d columns", row0, col0, commit, n);
            return -1;
        }
        w.row0 = row0; w.col0 = col0; w.commit = commit;
        wins.push_back(w);
        return 0;
    }

    // bytes of snapshot records per workgroup of a guessing-decoder launch, for the form that will run (gdg_parallel decided)
    void size_snapshots() {
        snap_stride = 0;
        for (auto &w : wins) {
            const int64_t rec = ((w.new_n + 2 * w.g->m + 7) & ~7) + 8 * (int64_t)w.g->m;
            // (the threaded-ensemble form keeps: the state after reset, one tree thread's saved masks, one record per side thread)
            const int ens_slots = (kind == 1 && gp.multi_thread == 1) ? 2 + std::max(gp.max_side_depth - gp.max_tree_depth, 0) : 0;
            snap_stride = std::max(snap_stride, rec * (gdg_parallel ? SWD_GDG_SLOTS : std::max(std::max(max_guess, 1), ens_slots)));
        }
        snap_stride = (snap_stride + 15) & ~(int64_t)15;
    }

    int finalize(const swd_graph_desc *chk) {
        if (kind == 0 && p.osd_method == 1 && p.osd_order > 15) { set_error("osd_e supports osd_order <= 15 on the device"); return -1; }
        if (kind != 0) {
            max_guess = ((1 << gp.max_tree_depth) - 1) * 2 + gp.max_side_depth - gp.max_tree_depth; // bp_guessing_decoder.pyx:181
            if (max_guess < 0) max_guess = 0;
            if (max_guess > SWD_GDG_MAXGUESS) { set_error("max_guess=%d exceeds the device limit of %d snapshots", max_guess, SWD_GDG_MAXGUESS); return -1; }
            // parallel form of the tree search (swd_gdg_kernel.h): needs its record formats to hold the parameters
            // (and its work items to hold the unit: item_unit packs the window in 8 bits and the shot in 22, swd_gdg_kernel.h)
            if (gp.multi_thread == 1 && (gp.max_side_depth - gp.max_tree_depth > SWD_GDG_MAXGUESS - 2 || gp.max_tree_depth > 6)) {
                set_error("multi_thread: at most %d side threads and tree depth 6 on the device", SWD_GDG_MAXGUESS - 2); return -1;
            }
            // (... and a tree of at most SWD_GDG_SLOTS snapshots: deeper trees take the serial walk)
            gdg_parallel = kind == 1 && gp.multi_thread != 1 && max_guess <= SWD_GDG_SLOTS && gp.max_side_branch_step <= SWD_GDG_MAXSTEP && gp.max_step < 200 && gp.max_side_depth < 200 &&
                           wins.size() <= SWD_GDG_ITEM_MAX_WINDOWS && !getenv("SWD_GDG_SERIAL");
            for (auto &w : wins) new_n_max = std::max(new_n_max, w.new_n);
            size_snapshots();
        }
        nmax = 0;
        int lmax = 0, mmax = 0;
        dm = 0;
        for (auto &w : wins) { nmax = std::max(nmax, w.g->n); dm = std::max(dm, w.g->D); }
        int kmax = 0;
        mmax = 0;
        for (auto &w : wins) { kmax = std::max(kmax, w.g->K); mmax = std::max(mmax, w.g->m); }
        const int mtop = mmax, ktop = kmax;
        int mtop_rows = 0;
        for (auto &w : wins) mtop_rows = std::max(mtop_rows, w.row0 + w.g->m);
        // the LDS-resident kernels first; graphs beyond them (no variant, or more than a CU's 160 KB of LDS per shot) take the
        // large-graph form of the osd_window kernels, whose scratch region lives in HBM
        for (int attempt = 0; attempt < 2; ++attempt) {
            big = attempt == 1;
            // (large graphs under a guessing decoder: the serial tree walk / the ticket-scheduled ensemble with every message in HBM --
            // a functional path for graphs such as the reference's [[288,12,18]] (4,1) windows, not a tuned one)
            variant = big ? select_big_variant(mtop, nmax, dm, ktop, kind) : select_variant(wins, mtop, nmax, dm, ktop, kind);
            if (variant && big && kind == 1 && gp.multi_thread == 1 && !variant->launch_ens) variant = nullptr;
            if (!variant) {
                set_error("no kernel variant for m=%d n=%d column weight %d row weight %d%s", mtop, nmax, dm, ktop,
                          big ? " (large-graph kernels: up to 1024 checks, 9216 columns, column weight 10, row weight 64)" : "");
                continue;
            }
            if (big && kind != 0) {
                bool d2 = true;
                for (auto &w : wins) d2 = d2 && w.new_n <= 2 * variant->nt;
                if (!d2) { set_error("guessing decoders on large graphs keep at most %d columns (new_n <= 2 x threads)", 2 * variant->nt); variant = nullptr; continue; }
                gdg_parallel = false; // (no work items: the serial tree walk)
                size_snapshots();     // ... whose snapshot area holds max_guess records per workgroup, not the parallel form's slots
            }
            nt = variant->nt; vf = variant->vf;
            mmax = 0; lmax = 0; big_stride = 0;
            // (large graphs: LDS left for the post-phase messages / OSD arrays after the state, the accumulators and the syndrome bytes)
            const int det_est = chk ? chk->m : mtop_rows;
            for (auto &w : wins) {
                make_layout(*w.g, w.new_n, nt, kind, w.L, big, 160 * 1024 - 64 - align_up(det_est, 16));
                lmax = std::max(lmax, w.L.total); mmax = std::max(mmax, w.row0 + w.g->m);
                big_stride = std::max<int64_t>(big_stride, align_up(w.L.big_scratch, 256));
            }
            if (chk) { num_det = chk->m; num_col = chk->n; } else { num_det = mmax; num_col = 0; }
            if (mmax > num_det) { set_error("window rows exceed the global check matrix (%d > %d)", mmax, num_det); return -1; }
            off_det = align_up(lmax, 16) + 16; // 16 bytes below the syndrome bytes: per-shot accumulators
            int dmax = num_det; // tuned osd_window kernels (up to 256 threads): LDS keeps the residual syndrome of one window's rows (whole words), swd_osdw_kernel.h
            if (kind == 0 && nt <= SWD_TUNED_NT) {
                dmax = 0;
                for (auto &w : wins) dmax = std::max(dmax, std::min((w.row0 + w.g->m + 3) & ~3, (num_det + 3) & ~3) - (w.row0 & ~3));
            }
            lds_total = off_det + align_up(dmax, 16);
            if (lds_total > 160 * 1024) {
                set_error("window graph needs %d bytes of LDS per shot (> 163840)%s", lds_total, big ? " even with its messages in HBM" : "");
                variant = nullptr;
                continue;
            }
            break;
        }
        if (!variant) return -1;
        post_depth2 = kind != 0 && !getenv("SWD_GDG_NO_DEPTH2");
        for (auto &w : wins) post_depth2 = post_depth2 && w.new_n <= 2 * nt;
        if (kind == 1 && gp.multi_thread == 1) {
            // prefix-tree walk of the ensemble (swd_gdg_kernel.h, gdg_ensemble_tree): behind the 2 + (S - D) mask records come D fork
            // records (masks + message cells: every cell of the window, or -- static tree-walk cache -- VF x DM cells per thread),
            // the stash vector and the main thread's exit vector
            const int ens_slots = 2 + std::max(gp.max_side_depth - gp.max_tree_depth, 0);
            const int vfp = (post_depth2 && vf > 2) ? 2 : vf;
            for (auto &w : wins) {
                const int64_t rec = ((w.new_n + 2 * w.g->m + 7) & ~7) + 8 * (int64_t)w.g->m;
                const int64_t cells = std::max<int64_t>(w.g->E + 1 + 2 * (nt / 64), (int64_t)vfp * variant->dm * nt);
                const int64_t forkb = ((rec + 15) & ~(int64_t)15) + ((cells * 8 + 15) & ~(int64_t)15);
                snap_stride = std::max(snap_stride, ((rec * ens_slots + 15) & ~(int64_t)15) + gp.max_tree_depth * forkb + 2 * (((int64_t)w.new_n + 15) & ~(int64_t)15) + 16);
            }
            snap_stride = (snap_stride + 15) & ~(int64_t)15;
        }
        if (status.reserve(64)) return -1; // word 0: fault flags; words 1..15: counters of diagnostic builds
        SWD_HIP(hipMemset(status.p, 0, 64));
        std::vector<SwdWindowDev> hw(wins.size());
        if (variant->sf) {
            std::vector<uint32_t> all, one;
            for (auto &w : wins) { split_map(*w.g, nt, 4 * variant->kg, &one); all.insert(all.end(), one.begin(), one.end()); }
            if (d_cnmap.reserve(all.size() * 4)) return -1;
            SWD_HIP(hipMemcpy(d_cnmap.p, all.data(), all.size() * 4, hipMemcpyHostToDevice));
        }
        for (size_t i = 0; i < wins.size(); ++i) {
            hw[i].cn_map = variant->sf ? d_cnmap.as<uint32_t>() + i * (size_t)nt : nullptr;
            hw[i].g = wins[i].g->d;
            hw[i].g.new_n = wins[i].new_n;
            hw[i].L = wins[i].L;
            hw[i].row0 = wins[i].row0; hw[i].col0 = wins[i].col0; hw[i].commit = wins[i].commit; hw[i].pad = 0;
        }
        if (d_wins.reserve(hw.size() * sizeof(SwdWindowDev))) return -1;
        SWD_HIP(hipMemcpy(d_wins.p, hw.data(), hw.size() * sizeof(SwdWindowDev), hipMemcpyHostToDevice));
        if (chk) {
            // CSC of the global check matrix for the residual-syndrome update (osd.py:178)
            if (chk->m > 65535) { set_error("more than 65535 detectors"); return -1; }
            std::vector<uint32_t> cp(chk->n + 1, 0);
            for (int e = 0; e < chk->nnz; ++e) {
                if (chk->col_idx[e] < 0 || chk->col_idx[e] >= chk->n) { set_error("global check matrix: column out of range"); return -1; }
                cp[chk->col_idx[e] + 1]++;
            }
            for (int c = 0; c < chk->n; ++c) cp[c + 1] += cp[c];
            std::vector<uint16_t> rows(chk->nnz);
            std::vector<uint32_t> fill(cp.begin(), cp.end() - 1);
            for (int r = 0; r < chk->m; ++r)
                for (int e = chk->row_ptr[r]; e < chk->row_ptr[r + 1]; ++e) rows[fill[chk->col_idx[e]]++] = (uint16_t)r;
            size_t o_rows = align_up((int)(cp.size() * 4), 256);
            if (d_chk.reserve(o_rows + rows.size() * 2)) return -1;
            SWD_HIP(hipMemcpy(d_chk.p, cp.data(), cp.size() * 4, hipMemcpyHostToDevice));
            SWD_HIP(hipMemcpy((char *)d_chk.p + o_rows, rows.data(), rows.size() * 2, hipMemcpyHostToDevice));
            d_colptr = (const uint32_t *)d_chk.p;
            d_rows = (const uint16_t *)((char *)d_chk.p + o_rows);
            for (auto &w : wins)
                if (w.col0 + w.commit > num_col) { set_error("commit range exceeds the global column count"); return -1; }
            for (auto &w : wins)
                if (w.row0 + w.g->m > chk->m) { set_error("invalid window placement: rows %d..%d exceed the %d detectors", w.row0, w.row0 + w.g->m, chk->m); return -1; }
        }
        return 0;
    }
};



template <int NT, int VF, int DM, int KG, int KIND, bool SF = false, bool BIG = false, int VFP = VF>
int launch_nt(Plan *d, const SwdPipeArgs &a0, hipStream_t st) {
    SwdPipeArgs a = a0;
    static std::mutex fn_mu; // the attribute and the occupancy answer belong to the function, not to a decoder
    std::lock_guard<std::mutex> fn_lock(fn_mu);
    static int lds_limit[64] = {0}; // per device, monotone
    if (d->lds_total > lds_limit[d->device & 63]) {
        SWD_HIP(hipFuncSetAttribute((const void *)pipeline_kernel<NT, VF, DM, KG, KIND, SF, BIG, VFP>, hipFuncAttributeMaxDynamicSharedMemorySize, d->lds_total));
        lds_limit[d->device & 63] = d->lds_total;
    }
    // persistent grid: as many workgroups as fit the device at once (they draw work units until none is left)
    static int slots[64] = {0};
    static int slots_lds[64] = {0};
    if (!slots[d->device & 63] || slots_lds[d->device & 63] != d->lds_total) {
        int per_cu = 0, cus = 0;
        SWD_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pipeline_kernel<NT, VF, DM, KG, KIND, SF, BIG, VFP>, NT, (size_t)d->lds_total));
        SWD_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, d->device));
        // the occupancy API accepts 54 592 B of LDS for three workgroups per CU; the hardware placed a third one up to 53 552 B
        // and not at 54 032 B (scripts/residency_check.py): count LDS in granules of 1280 B
        while (per_cu > 1 && (long long)per_cu * align_up(d->lds_total, 1280) > 160 * 1024) --per_cu;
        slots[d->device & 63] = std::max(1, per_cu) * std::max(1, cus);
        slots_lds[d->device & 63] = d->lds_total;
        if (getenv("SWD_DEBUG")) fprintf(stderr, "[swd] pipeline_kernel<%d,%d,%d,%d,%d>: %d workgroups per CU x %d CUs, %d B LDS\n", NT, VF, DM, KG, KIND, per_cu, cus, d->lds_total);
    }
    const long long units = (long long)a.B * a.W;
    static const int grid_pct = getenv("SWD_GRID_PCT") ? std::max(1, atoi(getenv("SWD_GRID_PCT"))) : 100; // diagnostics: how does the launch scale with the workgroups per CU?
    const unsigned grid = (unsigned)std::min<long long>(units, std::max(1, slots[d->device & 63] * grid_pct / 100));
    if constexpr (KIND == 7) {
        // The threaded ensemble on the work-item ring (sliding-window launches): UNIT items (a window is queued when its predecessor has
        // committed), one TASK item per tree thread of a parked ensemble, FINAL (swd_gdg_kernel.h, gdg_ensemble_tree ROLE 1 / 2).
        // Context: header | position list | forks at depth D - 1 | offer table | one vector per hypothesis + the main thread's exit
        // vector; its fork records (masks + message cells, 2^(D-1) of them) in the csnap area.
        // (large-graph form: window-major tickets, every thread body on the unit's workgroup)
        const int Dp = d->gp.max_tree_depth, NS = std::max(d->gp.max_side_depth - d->gp.max_tree_depth, 0);
        // (a stream object's batches of 3072 shots or more: tickets, every thread body on the unit's workgroup -- the ring's workgroups poll
        // until the launch ends, the ticket-scheduled ones leave their slots to the next batch: 64 hypotheses, 4096 shots per batch 77.9 ->
        // 75.7 ms per step, 16 384 shots 0.62 -> 0.68 M windows/s; one launch at a time the ring wins, 80 against 119 ms)
        if (!BIG && a.W > 1 && a.B < SWD_GDG_ITEM_MAX_SHOTS && a.W <= SWD_GDG_ITEM_MAX_WINDOWS && !getenv("SWD_ENS_TICKETS") && !d->stream_serial) {
            a.slot_scratch = 1;
            const bool tasks = Dp >= 1 && !getenv("SWD_ENS_NO_TASKS");
            unsigned nctx = 64;
            while (nctx < 2 * grid) nctx <<= 1;
            const int nh = 1 + ((1 << Dp) - 1) + NS, nforks = Dp >= 1 ? (1 << (Dp - 1)) : 0;
            unsigned cap = 1024;
            while (cap < (unsigned)a.B + (tasks ? nctx * (unsigned)((1 << Dp) + 2 + NS) : 0u) + 2u * grid + 1024u) cap <<= 1;
            const size_t qbytes = 16 + (size_t)cap * 8, fbytes = 16 + (size_t)nctx * 8;
            const int pos_b = align_up(d->new_n_max * 2, 16), vec_b = align_up(d->new_n_max, 16);
            a.gdgp.off_pos = SWD_GDG_HDR_BYTES; a.gdgp.off_node = a.gdgp.off_pos + pos_b; a.gdgp.off_rec = a.gdgp.off_node + align_up(std::max(nforks, 1) * 8 + NS * 12, 16); // (forks: guess, favoured value; side threads: guess, value, depth)
            a.gdgp.off_err = a.gdgp.off_rec + align_up(nh * 16, 16); a.gdgp.err_stride = vec_b; a.gdgp.ens_hyps = nh;
            a.gdgp.ctx_stride = align_up(a.gdgp.off_err + (nh + 1) * vec_b, 256);
            int64_t forkb = 0, rec16 = 0;
            const int vfp = (d->post_depth2 && d->vf > 2) ? 2 : d->vf;
            for (auto &w : d->wins) {
                const int64_t rec = ((w.new_n + 2 * w.g->m + 7) & ~7) + 8 * (int64_t)w.g->m;
                const int64_t cells = std::max<int64_t>(w.g->E + 1 + 2 * (NT / 64), (int64_t)vfp * DM * NT);
                forkb = std::max(forkb, ((rec + 15) & ~(int64_t)15) + ((cells * 8 + 15) & ~(int64_t)15));
                rec16 = std::max(rec16, (rec + 15) & ~(int64_t)15);
            }
            a.gdgp.csnap_stride = ((int64_t)nforks * forkb + (int64_t)NS * rec16 + 255) & ~(int64_t)255; // fork records, then the side threads' masks
            Plan::LaunchSlot &sl = *d->cur;
            if (sl.gq.reserve(qbytes)) return -1;
            a.gdgp.q = sl.gq.as<uint32_t>(); a.gdgp.qmask = cap - 1;
            a.gdgp.shots_inflight = (int)std::min<long long>(a.B, grid);
            SWD_HIP(hipMemsetAsync(a.gdgp.q, 0, qbytes, st));
            if (tasks) {
                if (sl.gfq.reserve(fbytes) || sl.gctx.reserve((size_t)nctx * a.gdgp.ctx_stride) || sl.gsnap.reserve((size_t)nctx * a.gdgp.csnap_stride + 256)) return -1;
                if (sl.gfree_n != (int)nctx) { // template of the full free ring: ids 0..nctx-1 in order
                    std::vector<uint64_t> tmpl(2 + nctx);
                    tmpl[0] = (uint64_t)nctx << 32; tmpl[1] = 0; // head 0, tail = contexts available
                    for (unsigned t = 0; t < nctx; ++t) tmpl[2 + t] = ((uint64_t)(t + 1) << 32) | t;
                    if (sl.gfree_tmpl.reserve(fbytes)) return -1;
                    SWD_HIP(hipMemcpy(sl.gfree_tmpl.p, tmpl.data(), fbytes, hipMemcpyHostToDevice));
                    sl.gfree_n = (int)nctx;
                }
                a.gdgp.fq = sl.gfq.as<uint32_t>(); a.gdgp.fmask = nctx - 1;
                a.gdgp.ctx = sl.gctx.as<uint8_t>(); a.gdgp.csnap = sl.gsnap.as<uint8_t>();
                a.gdgp.nctx = (int)nctx;
                SWD_HIP(hipMemcpyAsync(a.gdgp.fq, sl.gfree_tmpl.p, fbytes, hipMemcpyDeviceToDevice, st));
            }
        }
    }
    if constexpr (KIND == 2) { // work-item ring + contexts of parked trees (swd_gdg_kernel.h); items need per-workgroup scratch
        a.slot_scratch = 1;
        static const int inflight = getenv("SWD_GDG_INFLIGHT") ? std::max(1, atoi(getenv("SWD_GDG_INFLIGHT"))) : 4; // (4096 shots: 3, 4: 1.202 M windows/s; 6: 1.191; 8: 1.179; 12: 1.175 -- every branch is queued anyway while workgroups wait)
        unsigned nctx = 64;
        static const unsigned nctx_mult = getenv("SWD_GDG_NCTX_MULT") ? (unsigned)std::max(1, atoi(getenv("SWD_GDG_NCTX_MULT"))) : 2u; // diagnostics
        while (nctx < nctx_mult * grid) nctx <<= 1;
        // shots admitted at a time, in percent of the workgroups of the grid (a finished shot admits the next): every queued item
        // waits behind the whole ring, so a shot's chain of dependent items (unit -> side branches -> final -> next window)
        // finishes the sooner the fewer other shots are under way -- as long as the side branches keep the grid busy
        // (measured, [[144]] GDG windows: 2048 shots 43.1 ms at 400 %, 37.1 at 100 %, 36.2 at 75 %, 39.6 at 50 %, 65 at 25 %;
        // 4096 shots 70.8 ms at 100 %, 68.1 at 400 %)
        static const int shots_pct_env = getenv("SWD_GDG_SHOTS_PCT") ? std::max(1, atoi(getenv("SWD_GDG_SHOTS_PCT"))) : 0;
        const int shots_pct = shots_pct_env ? shots_pct_env : (a.B <= 4 * grid ? 100 : 400);
        a.gdgp.shots_inflight = (int)std::min<long long>(a.B, std::max<long long>(1, (long long)grid * shots_pct / 100));
        unsigned cap = 1024;
        while (cap < (unsigned)a.B + nctx * (unsigned)(SWD_GDG_SLOTS + 2) + 2u * grid + 1024u) cap <<= 1; // more than can ever be queued at once (idle workgroups admit further shots)
        const size_t qbytes = 16 + (size_t)cap * 8, fbytes = 16 + (size_t)nctx * 8;
        const int pos_b = align_up(d->new_n_max * 2, 16), err_b = align_up(d->new_n_max, 16);
        a.gdgp.off_pos = SWD_GDG_HDR_BYTES; a.gdgp.off_rec = a.gdgp.off_pos + pos_b; a.gdgp.off_err = a.gdgp.off_rec + SWD_GDG_SLOTS * SWD_GDG_REC_BYTES;
        a.gdgp.err_stride = err_b;
        a.gdgp.ctx_stride = align_up(a.gdgp.off_err + (SWD_GDG_SLOTS + 1) * err_b, 256);
        a.gdgp.csnap_stride = (d->snap_stride + 255) & ~(int64_t)255;
        Plan::LaunchSlot &sl = *d->cur;
        if (sl.gq.reserve(qbytes) || sl.gfq.reserve(fbytes) || sl.gctx.reserve((size_t)nctx * a.gdgp.ctx_stride) ||
            sl.gsnap.reserve((size_t)nctx * a.gdgp.csnap_stride))
            return -1;
        // Contexts handed out at a time: every one of them up to 3072 shots; 64 beyond -- a batch that fills the grid with units for
        // most of the launch walks most trees serially (no speculative branches, no queue traffic) and spreads only a few at a time
        // (measured, [[144]] GDG windows, ms per launch with 16 / 64 / 256 / all contexts: 256 shots 10.7 / 9.2 / 8.0 / 8.1, 1024 shots
        // 16.2 / 15.0 / 14.0 / 13.4, 2048 shots 25.1 / 24.0 / 24.0 / 22.9, 4096 shots 36.5 / 36.5 / 37.3 / 38.7; profiles/r06_gdg_stream.log)
        const unsigned navail = getenv("SWD_GDG_NCTX") ? std::min<unsigned>(nctx, (unsigned)atoi(getenv("SWD_GDG_NCTX"))) : (a.B > 3072 ? std::min(nctx, 64u) : nctx);
        if (sl.gfree_n != (int)nctx || sl.gfree_avail != (int)navail) { // template of the free ring: ids 0..navail-1 in order
            std::vector<uint64_t> tmpl(2 + nctx, 0);
            tmpl[0] = (uint64_t)navail << 32; tmpl[1] = 0; // head 0, tail = contexts available
            for (unsigned t = 0; t < navail; ++t) tmpl[2 + t] = ((uint64_t)(t + 1) << 32) | t;
            if (sl.gfree_tmpl.reserve(fbytes)) return -1;
            if (sl.done) SWD_HIP(hipEventSynchronize(sl.done)); // (the slot's previous launch may still be copying from the template)
            SWD_HIP(hipMemcpy(sl.gfree_tmpl.p, tmpl.data(), fbytes, hipMemcpyHostToDevice));
            sl.gfree_n = (int)nctx; sl.gfree_avail = (int)navail;
        }
        a.gdgp.q = sl.gq.as<uint32_t>(); a.gdgp.qmask = cap - 1;
        a.gdgp.fq = sl.gfq.as<uint32_t>(); a.gdgp.fmask = nctx - 1;
        a.gdgp.ctx = sl.gctx.as<uint8_t>(); a.gdgp.csnap = sl.gsnap.as<uint8_t>();
        a.gdgp.ensemble = d->gp.multi_thread == 2 ? 1 : 0;
        static const bool adaptive = !(getenv("SWD_GDG_ADAPTIVE") && atoi(getenv("SWD_GDG_ADAPTIVE")) == 0);
        a.gdgp.inflight_max = adaptive ? inflight : -inflight;
        a.gdgp.nctx = (int)nctx; a.gdgp.chk_status = d->status.as<uint32_t>();
        a.gdgp.static_bound = getenv("SWD_GDG_STATIC_BOUND") ? 1 : 0;
        SWD_HIP(hipMemsetAsync(a.gdgp.q, 0, qbytes, st));
        SWD_HIP(hipMemcpyAsync(a.gdgp.fq, sl.gfree_tmpl.p, fbytes, hipMemcpyDeviceToDevice, st));
    }
    // history ring and (guessing decoders) snapshot stack: per workgroup for sliding-window plans, per shot
    // otherwise; a caller-provided history buffer (single-window calls) is used as it is
    const size_t nscr = a.slot_scratch ? (size_t)grid : (size_t)a.B;
    if constexpr (KIND != 3) { // (kind 3 accumulates the history sum in registers and never touches the ring)
        if (!a.hist) {
            if (d->cur->hist.reserve(nscr * a.hist_stride * sizeof(double))) return -1;
            a.hist = d->cur->hist.as<double>();
        }
    }
    if constexpr (BIG) { // the workgroups' scratch regions in HBM (messages, sort keys, OSD arrays)
        if (d->cur->big.reserve((size_t)grid * d->big_stride)) return -1;
        a.big = d->cur->big.as<uint8_t>(); a.big_stride = d->big_stride;
    }
    if (d->kind != 0) {
        if (d->cur->snap.reserve(nscr * d->snap_stride + 8)) return -1;
        a.snap = d->cur->snap.as<uint8_t>(); a.snap_stride = d->snap_stride;
    }
    // Guessing decoders start their shots heaviest syndrome first (sliding-window launches whose grid does not hold every shot at
    // once): [[144]] GDG windows, 4096 shots 0.79 -> 0.92 M windows/s, 2048 shots 0.71 -> 0.76 M, 16384 shots (serial form) +1.5 %.
    // The osd_window kernels gain nothing from it (their launches end with the last ROUND of windows, not with the last shots;
    // ordering that round by the weight of the last window's rows: no gain either) and keep the natural order.
    static const int order_env = getenv("SWD_SHOT_ORDER") ? atoi(getenv("SWD_SHOT_ORDER")) : -1; // diagnostics: 0 off, 1 on for every kernel kind
    const bool want_order = order_env >= 0 ? order_env != 0 : (KIND == 1 || KIND == 2 || KIND == 7);
    a.order = nullptr;
    if (want_order && a.W > 1 && (unsigned)a.B > grid && a.det) {
        if (d->cur->order.reserve((size_t)a.B * 8)) return -1;
        uint32_t *wo = d->cur->order.as<uint32_t>();
        hipLaunchKernelGGL((shot_weight_kernel<256>), dim3((a.B + 3) / 4), dim3(256), 0, st, a.det, a.det_stride, a.num_det, a.B, wo);
        hipLaunchKernelGGL((shot_order_kernel<1024>), dim3(1), dim3(1024), 0, st, (const uint32_t *)wo, a.B, wo + a.B);
        a.order = wo + a.B;
    }
    hipLaunchKernelGGL((pipeline_kernel<NT, VF, DM, KG, KIND, SF, BIG, VFP>), dim3(grid), dim3(NT), d->lds_total, st, a);
    SWD_HIP(hipGetLastError());
    return 0;
}

// one launcher per kernel instantiation, defined in swd_kernels_*.hip
#define SWD_LAUNCHER_NAME(kind, nt, vf, dm, kg, sf) swd_launch_k##kind##_##nt##_##vf##_##dm##_##kg##_##sf
#define SWD_DECLARE_LAUNCHER(kind, nt, vf, dm, kg, sf) int SWD_LAUNCHER_NAME(kind, nt, vf, dm, kg, sf)(Plan *, const SwdPipeArgs &, hipStream_t);
// large-graph form of the osd_window kernels: kind 5 = kind 0, kind 6 = kind 3 (history sum in registers), scratch region in HBM
#define SWD_DEFINE_BIG_LAUNCHER(kind, nt, vf, dm, kg) \
    int SWD_LAUNCHER_NAME(kind, nt, vf, dm, kg, 0)(Plan *d, const SwdPipeArgs &a, hipStream_t st) { return launch_nt<nt, vf, dm, kg, (kind) == 6 ? 3 : 0, false, true>(d, a, st); }
#define SWD_DEFINE_LAUNCHER(kind, nt, vf, dm, kg, sf) \
    int SWD_LAUNCHER_NAME(kind, nt, vf, dm, kg, sf)(Plan *d, const SwdPipeArgs &a, hipStream_t st) { return launch_nt<nt, vf, dm, kg, kind, (sf) != 0>(d, a, st); }
// large-graph form of the guessing decoders' kernels: kind 8 = kind 1 (serial tree walk), kind 9 = kind 7 (threaded ensemble, tickets);
// depth-2 register cache for the kept columns (Plan::finalize admits only plans with new_n <= 2 nt)
#define SWD_DEFINE_BIG_GDG_LAUNCHER(kind, nt, vf, dm, kg) \
    int SWD_LAUNCHER_NAME(kind, nt, vf, dm, kg, 0)(Plan *d, const SwdPipeArgs &a, hipStream_t st) { return launch_nt<nt, vf, dm, kg, (kind) == 9 ? 7 : 1, false, true, 2>(d, a, st); }
// guessing decoders: a second instantiation with a depth-2 register cache for the shortened graph, taken when every window of the
// plan keeps new_n <= 2 nt columns (Plan::post_depth2)
#define SWD_DEFINE_GDG_LAUNCHER(kind, nt, vf, dm, kg, sf)                                                                  \
    int SWD_LAUNCHER_NAME(kind, nt, vf, dm, kg, sf)(Plan *d, const SwdPipeArgs &a, hipStream_t st) {                        \
        if constexpr ((vf) > 2) { if (d->post_depth2) return launch_nt<nt, vf, dm, kg, kind, (sf) != 0, false, 2>(d, a, st); } \
        return launch_nt<nt, vf, dm, kg, kind, (sf) != 0>(d, a, st);                                                       \
    }

} // namespace swd
