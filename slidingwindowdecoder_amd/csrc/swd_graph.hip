// CSR -> device Tanner-graph layout (see swd_graph.h) and host-side GF(2) rank.
// Replaces numpy2mod2sparse / spmatrix2mod2sparse (/root/reference/src/mod2sparse.pyx:5-31) and
// mod2sparse_rank (/root/reference/src/include/mod2sparse_extra.cpp:32-76, value only).
#include <math.h>
#include <string.h>

#include <mutex>

#include "swd_host.h"

namespace swd {

static thread_local std::string g_err;

void set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}

const char *last_error() { return g_err.c_str(); }

int gf2_rank(int m, int n, const std::vector<int32_t> &row_ptr, const std::vector<int32_t> &col_idx) {
    const int W = (n + 63) / 64;
    std::vector<uint64_t> rows((size_t)m * W, 0);
    for (int r = 0; r < m; ++r)
        for (int e = row_ptr[r]; e < row_ptr[r + 1]; ++e) rows[(size_t)r * W + (col_idx[e] >> 6)] ^= 1ull << (col_idx[e] & 63);
    int rank = 0;
    std::vector<char> used(m, 0);
    for (int c = 0; c < n && rank < m; ++c) {
        int pr = -1;
        for (int r = 0; r < m; ++r)
            if (!used[r] && ((rows[(size_t)r * W + (c >> 6)] >> (c & 63)) & 1)) { pr = r; break; }
        if (pr < 0) continue;
        used[pr] = 1;
        ++rank;
        for (int r = 0; r < m; ++r)
            if (!used[r] && ((rows[(size_t)r * W + (c >> 6)] >> (c & 63)) & 1))
                for (int w = c >> 6; w < W; ++w) rows[(size_t)r * W + w] ^= rows[(size_t)pr * W + w];
    }
    return rank;
}

int Graph::build(const swd_graph_desc *g) {
    if (!g || !g->row_ptr || !g->col_idx || !g->channel_probs) { set_error("null graph description"); return -1; }
    m = g->m; n = g->n; E = g->nnz;
    if (m <= 0 || n <= 0 || E <= 0) { set_error("empty check matrix (m=%d n=%d nnz=%d)", m, n, E); return -1; }
    if (g->row_ptr[0] != 0 || g->row_ptr[m] != E) { set_error("row_ptr does not span nnz"); return -1; }
    if (m > SWD_MAX_M) { set_error("m=%d exceeds this build's limit of %d checks", m, SWD_MAX_M); return -1; }
    if (n > 65534) { set_error("n=%d exceeds this build's limit of 65534 columns", n); return -1; }
    if (E > SWD_MAX_E) { set_error("nnz=%d exceeds this build's limit of %d edges", E, SWD_MAX_E); return -1; }
    row_ptr.assign(g->row_ptr, g->row_ptr + m + 1);
    col_idx.assign(g->col_idx, g->col_idx + E);
    for (int r = 0; r < m; ++r) {
        if (row_ptr[r + 1] < row_ptr[r]) { set_error("row_ptr not monotone at row %d", r); return -1; }
        std::sort(col_idx.begin() + row_ptr[r], col_idx.begin() + row_ptr[r + 1]);
        for (int e = row_ptr[r]; e < row_ptr[r + 1]; ++e) {
            if (col_idx[e] < 0 || col_idx[e] >= n) { set_error("column index out of range in row %d", r); return -1; }
            if (e > row_ptr[r] && col_idx[e] == col_idx[e - 1]) { set_error("duplicate entry in row %d", r); return -1; }
        }
    }
    // degrees
    row_deg.assign(m, 0);
    col_deg.assign(n, 0);
    K = 0; D = 0;
    std::vector<int> cdeg(n, 0);
    for (int r = 0; r < m; ++r) {
        int d = row_ptr[r + 1] - row_ptr[r];
        if (d > SWD_MAX_ROW_DEG) { set_error("row %d has weight %d > %d", r, d, SWD_MAX_ROW_DEG); return -1; }
        if (d == 0) { set_error("row %d is empty: every check needs degree > 0 (osd_window.pyx:117)", r); return -1; }
        K = std::max(K, d);
        for (int e = row_ptr[r]; e < row_ptr[r + 1]; ++e) cdeg[col_idx[e]]++;
    }
    for (int v = 0; v < n; ++v) {
        if (cdeg[v] > SWD_MAX_COL_DEG) { set_error("column %d has weight %d > %d", v, cdeg[v], SWD_MAX_COL_DEG); return -1; }
        D = std::max(D, cdeg[v]);
        col_deg[v] = (uint8_t)cdeg[v];
    }
    // lanes: checks by degree descending (stable in original index)
    perm.resize(m); iperm.resize(m);
    std::vector<int> order(m);
    for (int r = 0; r < m; ++r) order[r] = r;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
        return (row_ptr[a + 1] - row_ptr[a]) > (row_ptr[b + 1] - row_ptr[b]);
    });
    for (int l = 0; l < m; ++l) { perm[l] = (uint16_t)order[l]; iperm[order[l]] = (uint16_t)l; row_deg[l] = (uint8_t)(row_ptr[order[l] + 1] - row_ptr[order[l]]); }
    jptr.assign(K + 1, 0);
    for (int j = 0; j < K; ++j) {
        int cnt = 0;
        for (int l = 0; l < m; ++l) cnt += (row_deg[l] > j);
        jptr[j + 1] = (uint16_t)(jptr[j] + cnt);
    }
    row_col.assign(E, 0);
    // CSC with row-ascending entries
    col_ptr.assign(n + 1, 0);
    for (int v = 0; v < n; ++v) col_ptr[v + 1] = col_ptr[v] + cdeg[v];
    row_idx.assign(E, 0);
    vn_edge.assign((size_t)std::max(D, 1) * n, SWD_PAD_EDGE);
    vn_row.assign((size_t)std::max(D, 1) * n, 0xFFFF);
    std::vector<int> fill(n, 0);
    for (int r = 0; r < m; ++r) {
        const int l = iperm[r];
        for (int e = row_ptr[r]; e < row_ptr[r + 1]; ++e) {
            const int j = e - row_ptr[r];
            const int v = col_idx[e];
            const int slot = jptr[j] + l;
            row_col[slot] = (uint16_t)v;
            const int k = fill[v]++;
            row_idx[col_ptr[v] + k] = r;
            vn_edge[(size_t)k * n + v] = (uint32_t)slot | ((uint32_t)l << 16) | ((uint32_t)j << 26);
            vn_row[(size_t)k * n + v] = (uint16_t)r;
        }
    }
    llr.resize(n);
    for (int v = 0; v < n; ++v) {
        const double p = g->channel_probs[v];
        llr[v] = log((1 - p) / p); // osd_window.pyx:113
    }
    // listed order of the nodes for the tiered full-graph pass: degree tiers of two positions, heaviest first, stable in the index
    // (a window's column types -- runs of consecutive columns of one degree -- stay runs: neighbouring lanes keep neighbouring slots)
    {
        std::vector<int> ord(n);
        for (int v = 0; v < n; ++v) ord[v] = v;
        std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return (cdeg[a] + 1) / 2 > (cdeg[b] + 1) / 2; });
        vperm.resize(n); llr_s.resize(n);
        vn_edge_s.assign((size_t)std::max(D, 1) * n, SWD_PAD_EDGE);
        for (int i = 0; i < n; ++i) {
            vperm[i] = (uint16_t)ord[i];
            llr_s[i] = llr[ord[i]];
            for (int k = 0; k < D; ++k) vn_edge_s[(size_t)k * n + i] = vn_edge[(size_t)k * n + ord[i]];
        }
    }
    rank = gf2_rank(m, n, row_ptr, col_idx);
    wm = (m + 63) / 64;
    return 0;
}

int Graph::upload() {
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t o_jptr = 0;
    size_t o_rowcol = al(o_jptr + jptr.size() * 2);
    size_t o_rowdeg = al(o_rowcol + row_col.size() * 2);
    size_t o_perm = al(o_rowdeg + row_deg.size());
    size_t o_iperm = al(o_perm + perm.size() * 2);
    size_t o_vnedge = al(o_iperm + iperm.size() * 2);
    size_t o_vnrow = al(o_vnedge + vn_edge.size() * 4);
    size_t o_coldeg = al(o_vnrow + vn_row.size() * 2);
    size_t o_llr = al(o_coldeg + col_deg.size());
    size_t o_vperm = al(o_llr + llr.size() * 8);
    size_t o_vnedge_s = al(o_vperm + vperm.size() * 2);
    size_t o_llr_s = al(o_vnedge_s + vn_edge_s.size() * 4);
    size_t total = al(o_llr_s + llr_s.size() * 8);
    std::vector<char> h(total, 0);
    memcpy(&h[o_jptr], jptr.data(), jptr.size() * 2);
    memcpy(&h[o_rowcol], row_col.data(), row_col.size() * 2);
    memcpy(&h[o_rowdeg], row_deg.data(), row_deg.size());
    memcpy(&h[o_perm], perm.data(), perm.size() * 2);
    memcpy(&h[o_iperm], iperm.data(), iperm.size() * 2);
    memcpy(&h[o_vnedge], vn_edge.data(), vn_edge.size() * 4);
    memcpy(&h[o_vnrow], vn_row.data(), vn_row.size() * 2);
    memcpy(&h[o_coldeg], col_deg.data(), col_deg.size());
    memcpy(&h[o_llr], llr.data(), llr.size() * 8);
    memcpy(&h[o_vperm], vperm.data(), vperm.size() * 2);
    memcpy(&h[o_vnedge_s], vn_edge_s.data(), vn_edge_s.size() * 4);
    memcpy(&h[o_llr_s], llr_s.data(), llr_s.size() * 8);
    if (dev.reserve(total)) return -1;
    SWD_HIP(hipMemcpy(dev.p, h.data(), total, hipMemcpyHostToDevice));
    char *b = (char *)dev.p;
    d.m = m; d.n = n; d.E = E; d.K = K; d.D = D; d.rank = rank; d.wm = wm; d.new_n = 0;
    d.jptr = (const uint16_t *)(b + o_jptr);
    d.row_col = (const uint16_t *)(b + o_rowcol);
    d.row_deg = (const uint8_t *)(b + o_rowdeg);
    d.perm = (const uint16_t *)(b + o_perm);
    d.iperm = (const uint16_t *)(b + o_iperm);
    d.vn_edge = (const uint32_t *)(b + o_vnedge);
    d.vn_row = (const uint16_t *)(b + o_vnrow);
    d.col_deg = (const uint8_t *)(b + o_coldeg);
    d.llr = (const double *)(b + o_llr);
    d.vperm = (const uint16_t *)(b + o_vperm);
    d.vn_edge_s = (const uint32_t *)(b + o_vnedge_s);
    d.llr_s = (const double *)(b + o_llr_s);
    return 0;
}

} // namespace swd

extern "C" const char *swd_last_error(void) { return swd::last_error(); }
extern "C" int swd_abi_version(void) { return SWD_ABI_VERSION; }
extern "C" int swd_device_count(void) {
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) return 0;
    return c;
}
