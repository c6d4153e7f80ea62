// C-ABI host side of the quaternary decoder (include/swd.h: swd_bp4_*), replacing the reference's
// bp4_osd extension type (/root/reference/src/bp4_osd.pyx).
#include <math.h>
#include <string.h>

#include <mutex>

#include "swd_host.h"
#include "swd_bp4_kernel.h"

namespace swd {
int make_layout_for_osd(const Graph &g, int nt, SwdLdsLayout &L); // swd_osdw.hip

struct Bp4 {
    Graph gx, gz;
    swd_bp4_params p{};
    int device = 0, nt = 256, nt_osd = 256, dm = 4, n = 0;
    int nt_split = 0; // > 0: threads of the launch with two threads per qubit (the specialised instantiation; swd_bp4_kernel.h) // nt: threads of the BP kernel (a launch parameter), nt_osd: of the OSD kernel (its layouts)
    SwdLdsLayout Lx{}, Lz{};
    SwdBp4Layout L{};
    DevBuf llr, sx, sz, out, osd0, stats, pm, bpd, io, lpr; // (sx .. lpr: staging of the host-buffer entry points)
    PinnedBuf stage;
    const double *d_llr_x = nullptr, *d_llr_y = nullptr, *d_llr_z = nullptr;
    // What a launch writes and reads back besides the caller's arrays -- the queue of unconverged decodes between the BP and the OSD
    // kernel, the posterior buffer when the caller passes none, the four runs of camel_decode -- comes from a ring of launch slots
    // (as Plan::LaunchSlot does for the window decoders): launches of one handle on different streams never share it, and a launch
    // that re-uses a slot first makes its stream wait for the slot's previous launch.
    struct LaunchSlot {
        DevBuf osd_q, lpr, cdec, cpm, cst;
        hipEvent_t done = nullptr;
    };
    static constexpr int kSlots = 4;
    LaunchSlot slot[kSlots];
    int next_slot = 0;
    hipEvent_t last_done = nullptr; // (under mu) end of the most recent decode launch and the stream it ran on
    hipStream_t last_stream = nullptr;
    bool overlapped = false;        // (under mu) the launch being prepared found the previous one still running on another stream
    std::mutex mu;
    ~Bp4() { for (auto &sl : slot) if (sl.done) (void)hipEventDestroy(sl.done); }
    // the next slot, ordered behind its previous launch on `st` (call under mu)
    int take_slot(hipStream_t st, LaunchSlot **out) {
        LaunchSlot &sl = slot[next_slot];
        next_slot = (next_slot + 1) % kSlots;
        if (!sl.done) SWD_HIP(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
        else SWD_HIP(hipStreamWaitEvent(st, sl.done, 0));
        *out = &sl;
        return 0;
    }
};

template <int WMAX, int NTO, int DM, bool FAST, bool LAZY>
static int bp4_launch(Bp4 *d, const SwdBp4Args &a0, hipStream_t st, int nt) {
    SwdBp4Args a = a0;
    a.split = (FAST && nt == d->nt_split) ? 1 : 0;
    static std::mutex fn_mu; // the attributes and occupancy answers belong to the functions, not to a handle
    std::lock_guard<std::mutex> fn_lock(fn_mu);
    static int lds_limit[64] = {0};
    if (d->L.total > lds_limit[d->device & 63]) {
        SWD_HIP(hipFuncSetAttribute((const void *)bp4_kernel<WMAX, DM, FAST, LAZY>, hipFuncAttributeMaxDynamicSharedMemorySize, d->L.total));
        lds_limit[d->device & 63] = d->L.total;
    }
    // persistent grid: as many workgroups as the device holds at once, each walks its share of the units
    static int slots[64] = {0}, slots_lds[64] = {0}, slots_nt[64] = {0};
    if (!slots[d->device & 63] || slots_lds[d->device & 63] != d->L.total || slots_nt[d->device & 63] != nt) {
        int per_cu = 0, cus = 0;
        SWD_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, bp4_kernel<WMAX, DM, FAST, LAZY>, nt, (size_t)d->L.total));
        SWD_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, d->device));
        while (per_cu > 1 && (long long)per_cu * ((d->L.total + 2048 + 1279) / 1280 * 1280) > 160 * 1024) --per_cu; // LDS is granted in granules of 1280 B (swd_plan.h); + the kernel's 2 KB exp table
        slots[d->device & 63] = std::max(1, per_cu) * std::max(1, cus);
        slots_lds[d->device & 63] = d->L.total; slots_nt[d->device & 63] = nt;
    }
    const int units = a.camel ? 4 * a.B : a.B;
    const bool with_osd = !a.camel && a.osd_order >= 0;
    if (with_osd || a.ticket) SWD_HIP(hipMemsetAsync(a.osd_count, 0, 4 * sizeof(uint32_t), st)); // (queue counter, ticket counter)
    hipLaunchKernelGGL((bp4_kernel<WMAX, DM, FAST, LAZY>), dim3(std::min(units, slots[d->device & 63])), dim3(nt), d->L.total, st, a);
    SWD_HIP(hipGetLastError());
    if (with_osd) { // the queue of unconverged decodes (often empty: its workgroups then read the count and leave)
        static int lds_limit2[64] = {0}, slots2[64] = {0}, slots2_lds[64] = {0};
        if (d->L.total > lds_limit2[d->device & 63]) {
            SWD_HIP(hipFuncSetAttribute((const void *)bp4_osd_kernel<NTO, DM>, hipFuncAttributeMaxDynamicSharedMemorySize, d->L.total));
            lds_limit2[d->device & 63] = d->L.total;
        }
        if (!slots2[d->device & 63] || slots2_lds[d->device & 63] != d->L.total) {
            int per_cu = 0, cus = 0;
            SWD_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, bp4_osd_kernel<NTO, DM>, NTO, (size_t)d->L.total));
            SWD_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, d->device));
            while (per_cu > 1 && (long long)per_cu * ((d->L.total + 1279) / 1280 * 1280) > 160 * 1024) --per_cu;
            slots2[d->device & 63] = std::max(1, per_cu) * std::max(1, cus);
            slots2_lds[d->device & 63] = d->L.total;
        }
        hipLaunchKernelGGL((bp4_osd_kernel<NTO, DM>), dim3(std::min(a.B, slots2[d->device & 63])), dim3(NTO), d->L.total, st, a);
        SWD_HIP(hipGetLastError());
    }
    return 0;
}

template <int WMAX, int NTO, bool FAST, bool LAZY = false>
static int bp4_dispatch_dm(Bp4 *d, const SwdBp4Args &a, hipStream_t st, int nt) {
    return d->dm == 4 ? bp4_launch<WMAX, NTO, 4, FAST, LAZY>(d, a, st, nt)
                      : (d->dm == 8 ? bp4_launch<WMAX, NTO, 8, FAST, LAZY>(d, a, st, nt) : bp4_launch<WMAX, NTO, SWD_DMAX, FAST, LAZY>(d, a, st, nt));
}
template <bool FAST>
static int bp4_dispatch_nt(Bp4 *d, const SwdBp4Args &a, hipStream_t st, int nt) {
    // BP kernel: one thread per qubit (two in the specialised launch) while the waves fit a workgroup; OSD kernel: the workgroups its
    // layouts were made for.  The two-half node update (LAZY, swd_bp4_kernel.h) where it was measured to pay: every launch with one thread
    // per qubit (profiles/r06_bp4_lazy.log; the two-threads-per-qubit launches of the small codes keep the fused update); SWD_BP4_NO_LAZY: never
    static const bool lazy_ok = getenv("SWD_BP4_NO_LAZY") == nullptr;
    if constexpr (FAST) {
        // (two threads per qubit: the fused form one launch at a time, the two-half form when another launch is in flight beside this one --
        // SHYPS r = 3, two streams: 33.2 -> 40.8 M decodes/s, [[72]] 70.7 / 70.0; one at a time it loses 5-7 %)
        if (lazy_ok && nt <= 256 && (nt != d->nt_split || d->overlapped)) return bp4_dispatch_dm<4, 256, true, true>(d, a, st, nt);
        if (lazy_ok && nt > 256 && nt <= 512) return bp4_dispatch_dm<8, 256, true, true>(d, a, st, nt);
        if (lazy_ok && nt > 512) return d->nt_osd == 256 ? bp4_dispatch_dm<16, 256, true, true>(d, a, st, nt) : bp4_dispatch_dm<16, 1024, true, true>(d, a, st, nt);
    }
    if (nt <= 256) return bp4_dispatch_dm<4, 256, FAST>(d, a, st, nt); // (n <= 3072: the OSD layouts are those of 256 threads)
    if (nt <= 512) return bp4_dispatch_dm<8, 256, FAST>(d, a, st, nt); // (still six waves per SIMD: three workgroups of up to eight waves per CU)
    return d->nt_osd == 256 ? bp4_dispatch_dm<16, 256, FAST>(d, a, st, nt) : bp4_dispatch_dm<16, 1024, FAST>(d, a, st, nt);
}
static int bp4_dispatch(Bp4 *d, const SwdBp4Args &a, hipStream_t st) {
    // the specialised instantiation: a thread per qubit and per check, no camel run (swd_bp4_kernel.h); two threads per qubit for
    // small codes (swd_bp4_create; SWD_BP4_NOSPLIT: the one-thread form)
    const bool fast = !a.camel && d->n <= d->nt && d->gx.m + d->gz.m <= d->nt && !getenv("SWD_BP4_GENERIC");
    if (fast) return bp4_dispatch_nt<true>(d, a, st, d->nt_split ? d->nt_split : d->nt);
    return bp4_dispatch_nt<false>(d, a, st, d->nt);
}
} // namespace swd

using namespace swd;

extern "C" swd_bp4 *swd_bp4_create(const swd_graph_desc *hx, const swd_graph_desc *hz, const double *px, const double *py,
                                   const double *pz, const swd_bp4_params *p, int device) {
    if (!hx || !hz || !px || !py || !pz || !p) { set_error("null argument"); return nullptr; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device available: the MI355X decoder has no CPU fallback"); return nullptr; }
    if (device < 0 || device >= ndev || hipSetDevice(device) != hipSuccess) { set_error("device %d not usable", device); return nullptr; }
    if (hx->n != hz->n) { set_error("Hx, Hz blocklength does not match!"); return nullptr; }
    Bp4 *d = new Bp4();
    d->device = device; d->p = *p; d->n = hx->n;
    const int n = hx->n;
    if (d->p.osd_method == 0) d->p.osd_order = 0;
    if (d->p.osd_method < 0 || d->p.osd_method > 2) { set_error("ERROR: OSD method '%d' invalid.", d->p.osd_method); delete d; return nullptr; }
    // OSD path metrics use prior_llr_x = log((1-(px+py))/(px+py)), prior_llr_z = log((1-(pz+py))/(pz+py)) (bp4_osd.pyx:134-137)
    std::vector<double> qx(n), qz(n);
    for (int v = 0; v < n; ++v) { qx[v] = px[v] + py[v]; qz[v] = pz[v] + py[v]; }
    swd_graph_desc dx = *hx, dz = *hz;
    dx.channel_probs = qx.data(); dz.channel_probs = qz.data();
    if (d->gx.build(&dx) || d->gz.build(&dz)) { delete d; return nullptr; }
    const int kmin = std::min(n - d->gx.rank, n - d->gz.rank);
    if (d->p.osd_order > kmin) { // bp4_osd.pyx:92-98
        set_error("For this code, the OSD order should be set in the range 0<=osd_oder<=%d.", kmin);
        delete d; return nullptr;
    }
    if (d->p.osd_order > 0 && d->gx.rank < d->gz.rank) {
        // the reference sizes BOTH sweeps with kx = n - rank_x (bp4_osd.pyx:103-104, :284); with rank(Hx) < rank(Hz) its z-basis sweep
        // reads n - rank_x columns behind the rank_z pivots of an n-long array, i.e. past its end
        set_error("higher-order OSD with rank(Hx) < rank(Hz) is undefined in the reference (kz = n - rank_x: it reads past its column array) and not supported");
        delete d; return nullptr;
    }
    if (d->p.osd_method == 1 && d->p.osd_order > 15) { set_error("osd_e supports osd_order <= 15 on the device"); delete d; return nullptr; }
    const int D = std::max(d->gx.D, d->gz.D);
    if (D > SWD_DMAX) { set_error("column weight %d exceeds this build's bound %d", D, SWD_DMAX); delete d; return nullptr; }
    d->dm = D <= 4 ? 4 : (D <= 8 ? 8 : SWD_DMAX); // SHYPS stabiliser matrices reach column weight 9
    d->nt_osd = n <= 3072 ? 256 : 1024;
    d->nt = n <= 1024 ? std::max(64, (n + 63) / 64 * 64) : 1024;
    if (const char *e = getenv("SWD_BP4_NT")) { const int v = atoi(e); if (v >= 64 && v <= 1024 && v % 64 == 0) d->nt = v; } // (diagnostics)
    // two threads per qubit while the workgroup stays within four waves (measured, profiles/r06_bp4_split.log: [[72]] 43.8 -> 46.9 M
    // decodes/s, SHYPS r = 3 23.9 -> 24.9 M; [[144]] on five waves 35.9 -> 28.0 M, so larger codes keep one thread per qubit;
    // SWD_BP4_SPLIT_MAX overrides the bound on 2 n for experiments)
    const int split_max = getenv("SWD_BP4_SPLIT_MAX") ? atoi(getenv("SWD_BP4_SPLIT_MAX")) : 256;
    d->nt_split = (2 * n <= std::min(split_max, 512) && !getenv("SWD_BP4_NOSPLIT")) ? std::max(64, (2 * n + 63) / 64 * 64) : 0;
    if (d->gx.upload() || d->gz.upload()) { delete d; return nullptr; }
    d->gx.d.new_n = n; d->gz.d.new_n = n;
    // rank(Hx) > rank(Hz): the z-basis sweep walks the first kx = n - rank_x non-pivot columns like the reference's (not the n - rank_z
    // that exist).  The sweep takes k = new_n - rank candidate columns among the first new_n sorted ones, and the first k non-pivot
    // columns always lie among the first k + rank: new_n = kx + rank_z is the same set.
    if (d->p.osd_order > 0 && d->gx.rank > d->gz.rank) d->gz.d.new_n = n - d->gx.rank + d->gz.rank;
    make_layout_for_osd(d->gx, d->nt_osd, d->Lx);
    make_layout_for_osd(d->gz, d->nt_osd, d->Lz);
    auto al = [](int x, int a) { return (x + a - 1) / a * a; };
    const int msgx_b = al((d->gx.E + 1) * 8, 16), msgz_b = al((d->gz.E + 1) * 8, 16);
    int scratch = std::max(msgx_b + msgz_b, std::max(d->Lx.off_livemask, d->Lz.off_livemask));
    SwdBp4Layout &L = d->L;
    L.off_msgz = msgx_b;
    int o = al(scratch, 16);
    const int mx = d->gx.m, mz = d->gz.m;
    L.off_parx = o; o += mx * 4;
    L.off_parz = o; o += mz * 4;
    L.off_jptrx = o; o = al(o + (d->gx.K + 1) * 2, 4);
    L.off_jptrz = o; o = al(o + (d->gz.K + 1) * 2, 4);
    L.off_cnx = o; o += mx;
    L.off_cnz = o; o += mz;
    L.off_sxo = o; o += mx;
    L.off_szo = o; o += mz;
    L.off_decx = o; o += n;
    L.off_decz = o; o += n;
    L.off_hard = o; o = al(o + n, 16);
    L.off_misc = o; o += 640;
    L.total = al(o, 16);
    if (L.total > 160 * 1024) { set_error("code needs %d bytes of LDS per shot (> 163840)", L.total); delete d; return nullptr; }
    // channel LLRs (bp4_osd.pyx:127-133)
    std::vector<double> llr(3 * (size_t)n);
    for (int v = 0; v < n; ++v) {
        double num = px[v] + py[v] + pz[v];
        num = 1.0 - num;
        llr[v] = log(num / px[v]); llr[n + v] = log(num / py[v]); llr[2 * n + v] = log(num / pz[v]);
    }
    if (d->llr.reserve(llr.size() * 8)) { delete d; return nullptr; }
    if (hipMemcpy(d->llr.p, llr.data(), llr.size() * 8, hipMemcpyHostToDevice) != hipSuccess) { set_error("hipMemcpy failed"); delete d; return nullptr; }
    d->d_llr_x = d->llr.as<double>(); d->d_llr_y = d->d_llr_x + n; d->d_llr_z = d->d_llr_y + n;
    return (swd_bp4 *)d;
}

extern "C" void swd_bp4_destroy(swd_bp4 *h) {
    Bp4 *d = (Bp4 *)h;
    if (!d) return;
    (void)hipSetDevice(d->device);
    delete d;
}

extern "C" int swd_bp4_info(const swd_bp4 *h, int32_t *mx, int32_t *mz, int32_t *n, int32_t *rank_x, int32_t *rank_z) {
    const Bp4 *d = (const Bp4 *)h;
    if (!d) { set_error("null decoder"); return -1; }
    if (mx) *mx = d->gx.m;
    if (mz) *mz = d->gz.m;
    if (n) *n = d->n;
    if (rank_x) *rank_x = d->gx.rank;
    if (rank_z) *rank_z = d->gz.rank;
    return 0;
}

extern "C" int swd_bp4_decode_batch_dev(swd_bp4 *h, int32_t B, const uint8_t *sx, const uint8_t *sz, uint8_t *out,
                                        int32_t *stats, double *lpr, uint8_t *osd0, uint8_t *bp_dec, void *stream) {
    Bp4 *d = (Bp4 *)h;
    if (!d) { set_error("null decoder"); return -1; }
    if (B <= 0) return 0;
    if (!sx || !sz || !out || !stats) { set_error("null output/input pointer"); return -1; }
    SWD_HIP(hipSetDevice(d->device));
    hipStream_t st = (hipStream_t)stream;
    std::lock_guard<std::mutex> lk(d->mu);
    Bp4::LaunchSlot *sl = nullptr;
    if (d->take_slot(st, &sl)) return -1;
    const bool lpr_wanted = lpr != nullptr;
    if (!lpr) {
        // every slot's buffer at the first launch of this size, not one per launch: the allocations (226 MB each for 65 536 decodes of
        // a 144-qubit code) would otherwise fall into the first kSlots launches one by one -- 5-7 ms of host time each, behind a warm-up
        for (auto &s2 : d->slot) if (s2.lpr.reserve((size_t)B * 3 * d->n * 8)) return -1;
        lpr = sl->lpr.as<double>();
    }
    SwdBp4Args a{};
    a.gx = d->gx.d; a.gz = d->gz.d; a.Lx = d->Lx; a.Lz = d->Lz; a.L = d->L;
    a.llr_x = d->d_llr_x; a.llr_y = d->d_llr_y; a.llr_z = d->d_llr_z;
    a.max_iter = d->p.max_iter; a.osd_method = d->p.osd_method; a.osd_order = d->p.osd_order; a.alpha = d->p.ms_scaling_factor;
    a.B = B; a.sx = sx; a.sz = sz; a.out = out; a.osd0 = osd0; a.bp_dec = bp_dec; a.stats = stats; a.lpr = lpr;
    a.lpr_wanted = lpr_wanted ? 1 : 0;
    // [ OSD queue counter | ticket counter | - | - | OSD queue [B] | weights [B] | start order [B] ]
    for (auto &s2 : d->slot) if (s2.osd_q.reserve((size_t)B * 12 + 16)) return -1;
    a.osd_count = sl->osd_q.as<uint32_t>(); a.osd_list = sl->osd_q.as<int32_t>() + 4;
    static const bool static_units = getenv("SWD_BP4_STATIC") != nullptr; // (diagnostics: the static shares of rounds 4-5)
    d->overlapped = false;
    if (!static_units) {
        a.ticket = a.osd_count + 1;
        // start order: heaviest syndrome first.  The decodes that run all max_iter iterations are NOT the heaviest ones ([[144]]: weights 5-25
        // around a median of 10, spread over ranks 0.07-0.89 of the order: profiles/r06_bp4_heavy.log), so the order does not move the tail of
        // a launch; it still pays where the iteration count follows the weight -- SHYPS r = 3 18.8 -> 24.9 M decodes/s, [[72]] 40.0 -> 45.9,
        // [[144]] 35.1 -> 36.2; [[288]] .. [[756]] lose 2-3 % to the two small kernels (profiles/r06_bp4_order.log).  SWD_BP4_NO_ORDER: off
        // ... and not when another launch of this handle is still running on a DIFFERENT stream (a caller that keeps two launches in
        // flight): whatever tail this launch has, the other launch's grid fills it, and the two small kernels are 6 % of a step
        // ([[144]], two streams in turn: 55.4 -> 58.7 M decodes/s without them)
        static const bool by_weight = getenv("SWD_BP4_NO_ORDER") == nullptr;
        static const bool always = getenv("SWD_BP4_ORDER_ALWAYS") != nullptr;
        bool overlapped = false;
        if (!always && d->last_done && d->last_stream != st) {
            overlapped = hipEventQuery(d->last_done) == hipErrorNotReady;
            (void)hipGetLastError(); // ("not ready" is an answer, not an error: it must not be what the launch checks below pick up)
        }
        d->overlapped = overlapped;
        if (by_weight && !overlapped) {
            uint32_t *wt = sl->osd_q.as<uint32_t>() + 4 + B, *ord = wt + B;
            hipLaunchKernelGGL(bp4_weight_kernel, dim3((B + 3) / 4), dim3(256), 0, st, sx, sz, d->gx.m, d->gz.m, B, wt);
            hipLaunchKernelGGL((shot_order_kernel<1024>), dim3(1), dim3(1024), 0, st, (const uint32_t *)wt, B, ord);
            SWD_HIP(hipGetLastError());
            a.order = ord;
        }
    }
    if (bp4_dispatch(d, a, st)) return -1;
    SWD_HIP(hipEventRecord(sl->done, st));
    d->last_done = sl->done; d->last_stream = st;
    return 0;
}

extern "C" int swd_bp4_camel_decode_batch_dev(swd_bp4 *h, int32_t B, const uint8_t *sx, const uint8_t *sz, uint8_t *out,
                                              int32_t *stats, double *min_pm, void *stream) {
    Bp4 *d = (Bp4 *)h;
    if (!d) { set_error("null decoder"); return -1; }
    if (B <= 0) return 0;
    if (!sx || !sz || !out || !stats) { set_error("null output/input pointer"); return -1; }
    SWD_HIP(hipSetDevice(d->device));
    const size_t n = d->n;
    hipStream_t st = (hipStream_t)stream;
    std::lock_guard<std::mutex> lk(d->mu);
    Bp4::LaunchSlot *sl = nullptr;
    if (d->take_slot(st, &sl)) return -1;
    if (sl->lpr.reserve((size_t)4 * B * 3 * n * 8) || sl->cdec.reserve((size_t)4 * B * 2 * n) || sl->cpm.reserve((size_t)4 * B * 8) ||
        sl->cst.reserve((size_t)4 * B * 2 * 4))
        return -1;
    SwdBp4Args a{};
    a.gx = d->gx.d; a.gz = d->gz.d; a.Lx = d->Lx; a.Lz = d->Lz; a.L = d->L;
    a.llr_x = d->d_llr_x; a.llr_y = d->d_llr_y; a.llr_z = d->d_llr_z;
    a.max_iter = d->p.max_iter; a.osd_method = d->p.osd_method; a.osd_order = d->p.osd_order; a.alpha = d->p.ms_scaling_factor;
    a.B = B; a.sx = sx; a.sz = sz; a.out = nullptr; a.osd0 = nullptr; a.stats = nullptr; a.lpr = sl->lpr.as<double>();
    a.camel = 1; a.camel_dec = sl->cdec.as<uint8_t>(); a.camel_pm = sl->cpm.as<double>(); a.camel_st = sl->cst.as<int32_t>();
    int rc;
    rc = bp4_dispatch(d, a, st);
    if (rc) return rc;
    hipLaunchKernelGGL(bp4_camel_select, dim3(B), dim3(256), 0, st, (int)n, a.camel_dec, a.camel_pm, a.camel_st, out, stats, min_pm);
    SWD_HIP(hipGetLastError());
    SWD_HIP(hipEventRecord(sl->done, st));
    return 0;
}

extern "C" int swd_bp4_camel_decode_batch(swd_bp4 *h, int32_t B, const uint8_t *sx, const uint8_t *sz, uint8_t *out,
                                          int32_t *stats, double *min_pm) {
    Bp4 *d = (Bp4 *)h;
    if (!d) { set_error("null decoder"); return -1; }
    if (B <= 0) return 0;
    if (!sx || !sz || !out || !stats) { set_error("null output/input pointer"); return -1; }
    SWD_HIP(hipSetDevice(d->device));
    const size_t n = d->n, mx = d->gx.m, mz = d->gz.m;
    if (d->sx.reserve(B * mx) || d->sz.reserve(B * mz) || d->out.reserve(B * 2 * n) || d->stats.reserve((size_t)B * SWD_STAT_WORDS * 4) ||
        d->pm.reserve((size_t)B * 8))
        return -1;
    SWD_HIP(hipMemcpy(d->sx.p, sx, B * mx, hipMemcpyHostToDevice));
    SWD_HIP(hipMemcpy(d->sz.p, sz, B * mz, hipMemcpyHostToDevice));
    int rc = swd_bp4_camel_decode_batch_dev(h, B, d->sx.as<uint8_t>(), d->sz.as<uint8_t>(), d->out.as<uint8_t>(), d->stats.as<int32_t>(),
                                            d->pm.as<double>(), nullptr);
    if (rc) return rc;
    SWD_HIP(hipDeviceSynchronize());
    SWD_HIP(hipMemcpy(out, d->out.p, B * 2 * n, hipMemcpyDeviceToHost));
    SWD_HIP(hipMemcpy(stats, d->stats.p, (size_t)B * SWD_STAT_WORDS * 4, hipMemcpyDeviceToHost));
    if (min_pm) SWD_HIP(hipMemcpy(min_pm, d->pm.p, (size_t)B * 8, hipMemcpyDeviceToHost));
    return 0;
}

// large batches: one synchronous copy per array straight from / to the caller's buffers
static int bp4_decode_batch_direct(swd_bp4 *h, int32_t B, const uint8_t *sx, const uint8_t *sz, uint8_t *out,
                                    int32_t *stats, double *lpr, uint8_t *osd0, uint8_t *bp_dec) {
    Bp4 *d = (Bp4 *)h;
    if (!d) { set_error("null decoder"); return -1; }
    if (B <= 0) return 0;
    if (!sx || !sz || !out || !stats) { set_error("null output/input pointer"); return -1; }
    SWD_HIP(hipSetDevice(d->device));
    const size_t n = d->n, mx = d->gx.m, mz = d->gz.m;
    if (d->sx.reserve(B * mx) || d->sz.reserve(B * mz) || d->out.reserve(B * 2 * n) || d->stats.reserve((size_t)B * SWD_STAT_WORDS * 4) ||
        d->lpr.reserve((size_t)B * 3 * n * 8) || d->osd0.reserve(B * 2 * n) || (bp_dec && d->bpd.reserve(B * 2 * n)))
        return -1;
    SWD_HIP(hipMemcpy(d->sx.p, sx, B * mx, hipMemcpyHostToDevice));
    SWD_HIP(hipMemcpy(d->sz.p, sz, B * mz, hipMemcpyHostToDevice));
    SWD_HIP(hipMemset(d->osd0.p, 0, B * 2 * n));
    int rc = swd_bp4_decode_batch_dev(h, B, d->sx.as<uint8_t>(), d->sz.as<uint8_t>(), d->out.as<uint8_t>(), d->stats.as<int32_t>(),
                                      d->lpr.as<double>(), d->osd0.as<uint8_t>(), bp_dec ? d->bpd.as<uint8_t>() : nullptr, nullptr);
    if (rc) return rc;
    SWD_HIP(hipDeviceSynchronize());
    SWD_HIP(hipMemcpy(out, d->out.p, B * 2 * n, hipMemcpyDeviceToHost));
    SWD_HIP(hipMemcpy(stats, d->stats.p, (size_t)B * SWD_STAT_WORDS * 4, hipMemcpyDeviceToHost));
    if (lpr) SWD_HIP(hipMemcpy(lpr, d->lpr.p, (size_t)B * 3 * n * 8, hipMemcpyDeviceToHost));
    if (osd0) SWD_HIP(hipMemcpy(osd0, d->osd0.p, B * 2 * n, hipMemcpyDeviceToHost));
    if (bp_dec) SWD_HIP(hipMemcpy(bp_dec, d->bpd.p, B * 2 * n, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int swd_bp4_decode_batch(swd_bp4 *h, int32_t B, const uint8_t *sx, const uint8_t *sz, uint8_t *out,
                                    int32_t *stats, double *lpr, uint8_t *osd0, uint8_t *bp_dec) {
    Bp4 *d = (Bp4 *)h;
    if (!d) { set_error("null decoder"); return -1; }
    if (B <= 0) return 0;
    if (!sx || !sz || !out || !stats) { set_error("null output/input pointer"); return -1; }
    SWD_HIP(hipSetDevice(d->device));
    const size_t n = d->n, mx = d->gx.m, mz = d->gz.m;
    // packed device buffer + pinned mirror: [ sx | sz ] in, [ out | stats | lpr | osd0 | bp_dec ] out, one copy each way
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_sx = 0, o_sz = al(B * mx), o_out = o_sz + al(B * mz), o_stats = o_out + al(B * 2 * n),
                 o_lpr = o_stats + al((size_t)B * SWD_STAT_WORDS * 4), o_osd0 = o_lpr + al((size_t)B * 3 * n * 8),
                 o_bpd = o_osd0 + al(B * 2 * n), total = o_bpd + al(B * 2 * n);
    if (total > SWD_STAGE_MAX) return bp4_decode_batch_direct(h, B, sx, sz, out, stats, lpr, osd0, bp_dec); // large batches: copy time dominates, no second host copy
    if (d->io.reserve(total) || d->stage.reserve(total)) return -1;
    char *hs = (char *)d->stage.p, *ds = (char *)d->io.p;
    memcpy(hs + o_sx, sx, B * mx);
    memcpy(hs + o_sz, sz, B * mz);
    hipStream_t st = nullptr;
    SWD_HIP(hipMemcpyAsync(ds, hs, o_sz + B * mz, hipMemcpyHostToDevice, st));
    SWD_HIP(hipMemsetAsync(ds + o_osd0, 0, B * 2 * n, st));
    int rc = swd_bp4_decode_batch_dev(h, B, (const uint8_t *)(ds + o_sx), (const uint8_t *)(ds + o_sz), (uint8_t *)(ds + o_out),
                                      (int32_t *)(ds + o_stats), (double *)(ds + o_lpr), (uint8_t *)(ds + o_osd0),
                                      (uint8_t *)(ds + o_bpd), st);
    if (rc) return rc;
    // only what the caller asked for travels back (the posteriors are 24 n bytes per decode against 2 n of decisions)
    const size_t o_end = bp_dec ? total : (osd0 ? o_bpd : (lpr ? o_osd0 : o_lpr));
    SWD_HIP(hipMemcpyAsync(hs + o_out, ds + o_out, o_end - o_out, hipMemcpyDeviceToHost, st));
    SWD_HIP(hipStreamSynchronize(st));
    memcpy(out, hs + o_out, B * 2 * n);
    memcpy(stats, hs + o_stats, (size_t)B * SWD_STAT_WORDS * 4);
    if (lpr) memcpy(lpr, hs + o_lpr, (size_t)B * 3 * n * 8);
    if (osd0) memcpy(osd0, hs + o_osd0, B * 2 * n);
    if (bp_dec) memcpy(bp_dec, hs + o_bpd, B * 2 * n);
    return 0;
}
