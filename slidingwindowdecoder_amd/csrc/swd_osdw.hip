// C-ABI for the batched osd_window decoder (include/swd.h) -- host side of
// swd_osdw_kernel.h.  Replaces the reference's osd_window extension type
// (/root/reference/src/osd_window.pyx) behind a batch interface.
#include <string.h>

#include "swd_host.h"
#include "swd_osdw_kernel.h"

namespace swd {

static int next_pow2(int x) { int p = 1; while (p < x) p <<= 1; return p; }
static int align_up(int x, int a) { return (x + a - 1) / a * a; }

struct Osdw {
    Graph g;
    swd_osdw_params p{};
    int device = 0;
    int new_n = 0;
    int nt = 256;
    SwdLdsLayout L{};
    DevBuf synd, out, status, iters, pm, hist, osd0;
    bool timing = false;
    double t_total_ms = 0;
    int64_t t_launches = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;

    int layout() {
        const int m = g.m, n = g.n, E = g.E, wm = g.wm;
        const int npad = std::max(next_pow2(n), 2);
        L.npad = npad;
        L.off_idx = npad * 8;
        L.off_aux = align_up(npad * 10, 16);
        const int osd_bytes = L.off_aux + m * wm * 8 + wm * 8 + g.rank * 4 + n * 2 + 16;
        const int rare_bytes = L.off_aux + n * 2;
        int scratch = std::max(std::max(E * 8, osd_bytes), std::max(rare_bytes, n * 2));
        scratch = align_up(scratch, 16);
        int o = scratch;
        L.off_livemask = o; o += m * 8;
        L.off_par = o; o += m * 4;
        L.off_lv = o; o = align_up(o + new_n * 2, 4);
        L.off_jptr = o; o = align_up(o + (g.K + 1) * 2, 4);
        L.off_cnval = o; o += m;
        L.off_cndeg = o; o += m;
        L.off_vnval = o; o += n;
        L.off_hard = o; o = align_up(o + n, 16);
        L.off_misc = o; o += 64 * 4;
        L.total = align_up(o, 16);
        if (L.total > 160 * 1024) {
            set_error("window graph needs %d bytes of LDS per shot (> 163840): m=%d n=%d nnz=%d", L.total, m, n, E);
            return -1;
        }
        return 0;
    }
};

template <int NT>
static int launch_nt(Osdw *d, const SwdOsdwArgs &a, hipStream_t st) {
    static int lds_limit[64] = {0}; // per device, monotone: the attribute is per function, not per decoder
    if (d->L.total > lds_limit[d->device & 63]) {
        SWD_HIP(hipFuncSetAttribute((const void *)osdw_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, d->L.total));
        lds_limit[d->device & 63] = d->L.total;
    }
    hipLaunchKernelGGL(osdw_kernel<NT>, dim3(a.B), dim3(NT), d->L.total, st, a);
    SWD_HIP(hipGetLastError());
    return 0;
}

static int launch(Osdw *d, const SwdOsdwArgs &a, hipStream_t st) {
    if (d->timing) {
        if (!d->ev0) { SWD_HIP(hipEventCreate(&d->ev0)); SWD_HIP(hipEventCreate(&d->ev1)); }
        SWD_HIP(hipEventRecord(d->ev0, st));
    }
    int rc;
    switch (d->nt) {
    case 64: rc = launch_nt<64>(d, a, st); break;
    case 128: rc = launch_nt<128>(d, a, st); break;
    case 256: rc = launch_nt<256>(d, a, st); break;
    case 512: rc = launch_nt<512>(d, a, st); break;
    default: rc = launch_nt<1024>(d, a, st); break;
    }
    if (rc) return rc;
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

} // namespace swd

using namespace swd;

extern "C" swd_osdw *swd_osdw_create(const swd_graph_desc *g, const swd_osdw_params *p, int device) {
    if (!p) { set_error("null params"); return nullptr; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("no HIP device available: the MI355X decoder has no CPU fallback");
        return nullptr;
    }
    if (device < 0 || device >= ndev) { set_error("device %d out of range (%d devices)", device, ndev); return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { set_error("hipSetDevice(%d) failed", device); return nullptr; }
    Osdw *d = new Osdw();
    d->device = device;
    d->p = *p;
    if (d->g.build(g)) { delete d; return nullptr; }
    const int m = d->g.m, n = d->g.n;
    if (d->g.D > SWD_DMAX) { set_error("column weight %d exceeds this build's bound %d", d->g.D, SWD_DMAX); delete d; return nullptr; }
    d->new_n = (p->new_n <= 0) ? std::min(n, 2 * m) : std::min(p->new_n, n); // osd_window.pyx:60-63
    if (d->p.osd_method == 0) d->p.osd_order = 0;                             // osd_window.pyx:69-71
    if (d->p.osd_method < 0 || d->p.osd_method > 2) { set_error("ERROR: OSD method '%d' invalid.", d->p.osd_method); delete d; return nullptr; }
    if (d->p.osd_order > d->new_n - d->g.rank) {                              // osd_window.pyx:88-92
        set_error("For this code, the OSD order should be set in the range 0<=osd_oder<=%d.", d->new_n - d->g.rank);
        delete d; return nullptr;
    }
    if (d->p.osd_order > 0) { set_error("osd_order > 0 is not available in this build of the device OSD"); delete d; return nullptr; }
    if (d->g.upload()) { delete d; return nullptr; }
    d->g.d.new_n = d->new_n;
    d->nt = n <= 192 ? 64 : n <= 768 ? 128 : n <= 3072 ? 256 : 1024;
    if (d->layout()) { delete d; return nullptr; }
    return (swd_osdw *)d;
}

extern "C" void swd_osdw_destroy(swd_osdw *h) {
    Osdw *d = (Osdw *)h;
    if (!d) return;
    (void)hipSetDevice(d->device);
    if (d->ev0) { (void)hipEventDestroy(d->ev0); (void)hipEventDestroy(d->ev1); }
    delete d;
}

extern "C" int swd_osdw_info(const swd_osdw *h, int32_t *m, int32_t *n, int32_t *new_n, int32_t *rank) {
    const Osdw *d = (const Osdw *)h;
    if (!d) { set_error("null decoder"); return -1; }
    if (m) *m = d->g.m;
    if (n) *n = d->g.n;
    if (new_n) *new_n = d->new_n;
    if (rank) *rank = d->g.rank;
    return 0;
}

extern "C" int swd_osdw_set_timing(swd_osdw *h, int32_t on) {
    Osdw *d = (Osdw *)h;
    if (!d) { set_error("null decoder"); return -1; }
    d->timing = on != 0; d->t_total_ms = 0; d->t_launches = 0;
    return 0;
}

extern "C" int swd_osdw_get_timing(swd_osdw *h, double *total_ms, int64_t *launches) {
    Osdw *d = (Osdw *)h;
    if (!d) { set_error("null decoder"); return -1; }
    if (total_ms) *total_ms = d->t_total_ms;
    if (launches) *launches = d->t_launches;
    d->t_total_ms = 0; d->t_launches = 0;
    return 0;
}

extern "C" int swd_osdw_decode_batch_dev(swd_osdw *h, int32_t B, const uint8_t *synd, int64_t synd_stride,
                                         uint8_t *out, int64_t out_stride, int32_t *status, int32_t *iters,
                                         double *min_pm, double *hist, int32_t hist_is_state, uint8_t *osd0,
                                         void *stream) {
    Osdw *d = (Osdw *)h;
    if (!d) { set_error("null decoder"); return -1; }
    if (B <= 0) return 0;
    if (!synd || !out || !status || !iters || !min_pm) { set_error("null output/input pointer"); return -1; }
    SWD_HIP(hipSetDevice(d->device));
    if (!hist) {
        if (hist_is_state) { set_error("hist_is_state requires a caller-provided history buffer"); return -1; }
        if (d->hist.reserve((size_t)B * 4 * d->g.n * sizeof(double))) return -1;
        hist = d->hist.as<double>();
    }
    SwdOsdwArgs a{};
    a.g = d->g.d; a.L = d->L;
    a.pre_iter = d->p.pre_max_iter; a.post_iter = d->p.post_max_iter;
    a.osd_method = d->p.osd_method; a.osd_order = d->p.osd_order; a.alpha = d->p.ms_scaling_factor;
    a.B = B; a.hist_is_state = hist_is_state;
    a.synd = synd; a.synd_stride = synd_stride ? synd_stride : d->g.m;
    a.out = out; a.out_stride = out_stride ? out_stride : d->g.n;
    a.status = status; a.iters = iters; a.min_pm = min_pm; a.hist = hist; a.osd0 = osd0;
    return launch(d, a, (hipStream_t)stream);
}

extern "C" int swd_osdw_decode_batch(swd_osdw *h, int32_t B, const uint8_t *synd, uint8_t *out, int32_t *status,
                                     int32_t *iters, double *min_pm, double *hist, int32_t hist_is_state,
                                     uint8_t *osd0) {
    Osdw *d = (Osdw *)h;
    if (!d) { set_error("null decoder"); return -1; }
    if (B <= 0) return 0;
    SWD_HIP(hipSetDevice(d->device));
    const size_t m = d->g.m, n = d->g.n;
    if (d->synd.reserve(B * m) || d->out.reserve(B * n) || d->status.reserve(B * 4) || d->iters.reserve(B * 4) ||
        d->pm.reserve(B * 8) || d->hist.reserve((size_t)B * 4 * n * 8))
        return -1;
    if (osd0 && d->osd0.reserve(B * n)) return -1;
    SWD_HIP(hipMemcpy(d->synd.p, synd, B * m, hipMemcpyHostToDevice));
    if (hist && hist_is_state) SWD_HIP(hipMemcpy(d->hist.p, hist, (size_t)B * 4 * n * 8, hipMemcpyHostToDevice));
    if (osd0) SWD_HIP(hipMemset(d->osd0.p, 0, B * n));
    int rc = swd_osdw_decode_batch_dev(h, B, d->synd.as<uint8_t>(), 0, d->out.as<uint8_t>(), 0, d->status.as<int32_t>(),
                                       d->iters.as<int32_t>(), d->pm.as<double>(), d->hist.as<double>(),
                                       (hist && hist_is_state) ? 1 : 0, osd0 ? d->osd0.as<uint8_t>() : nullptr, nullptr);
    if (rc) return rc;
    SWD_HIP(hipDeviceSynchronize());
    SWD_HIP(hipMemcpy(out, d->out.p, B * n, hipMemcpyDeviceToHost));
    SWD_HIP(hipMemcpy(status, d->status.p, B * 4, hipMemcpyDeviceToHost));
    SWD_HIP(hipMemcpy(iters, d->iters.p, B * 4, hipMemcpyDeviceToHost));
    SWD_HIP(hipMemcpy(min_pm, d->pm.p, B * 8, hipMemcpyDeviceToHost));
    if (hist) SWD_HIP(hipMemcpy(hist, d->hist.p, (size_t)B * 4 * n * 8, hipMemcpyDeviceToHost));
    if (osd0) SWD_HIP(hipMemcpy(osd0, d->osd0.p, B * n, hipMemcpyDeviceToHost));
    return 0;
}
