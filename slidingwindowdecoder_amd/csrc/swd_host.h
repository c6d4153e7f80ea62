// Host-side helpers shared by the C-ABI translation units: error reporting, device buffers,
// and the CSR -> device-graph builder.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include <algorithm>
#include <string>
#include <vector>

#include "swd.h"
#include "swd_graph.h"

namespace swd {

void set_error(const char *fmt, ...);

#define SWD_HIP(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            swd::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,   \
                           __LINE__);                                                         \
            return -1;                                                                        \
        }                                                                                     \
    } while (0)

// Device buffer that only grows.  Growing never calls hipFree on the launch path: hipFree synchronises the whole device (and the old
// allocation may still be read by a launch in flight on another stream), so the outgrown allocation is retired and released with the
// buffer.  Growth is geometric (at least 1.5x), which bounds the retired bytes by twice the final size.
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    std::vector<void *> retired;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() {
        if (p) (void)hipFree(p);
        for (void *q : retired) (void)hipFree(q);
    }
    int reserve(size_t bytes) {
        if (bytes <= cap) return 0;
        const size_t want = std::max(bytes, cap + cap / 2);
        void *q = nullptr;
        hipError_t e = hipMalloc(&q, want);
        size_t got = want;
        if (e != hipSuccess && want > bytes) { (void)hipGetLastError(); e = hipMalloc(&q, bytes); got = bytes; }
        if (e != hipSuccess) {
            swd::set_error("hipMalloc(%zu bytes) failed: %s (%s:%d)", bytes, hipGetErrorString(e), __FILE__, __LINE__);
            return -1;
        }
        if (p) retired.push_back(p);
        p = q; cap = got;
        return 0;
    }
    template <class T> T *as() { return (T *)p; }
};

// Page-locked host staging area: one asynchronous copy in and one out per host-buffer call instead of a
// synchronous pageable copy per array (a one-syndrome decode() spent most of its time in those).
constexpr size_t SWD_STAGE_MAX = 4u << 20; // host-buffer calls moving more than this copy array by array instead
struct PinnedBuf {
    void *p = nullptr;
    size_t cap = 0;
    ~PinnedBuf() { if (p) (void)hipHostFree(p); }
    int reserve(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        SWD_HIP(hipHostMalloc(&p, bytes, hipHostMallocDefault));
        cap = bytes;
        return 0;
    }
};

// Host copy of the device graph arrays + the device allocation holding them.
struct Graph {
    int m = 0, n = 0, E = 0, K = 0, D = 0, rank = 0, wm = 0;
    std::vector<uint16_t> jptr, row_col, perm, iperm, vn_row;
    std::vector<uint8_t> row_deg, col_deg;
    std::vector<uint32_t> vn_edge, vn_edge_s;
    std::vector<uint16_t> vperm;
    std::vector<double> llr, llr_s;
    // original CSR/CSC kept for host-side use (rank, OSD-CS setup)
    std::vector<int32_t> row_ptr, col_idx, col_ptr, row_idx;
    DevBuf dev;
    SwdGraphDev d{};

    int build(const swd_graph_desc *g);   // validates, fills host arrays, computes rank
    int upload();                          // hipMalloc + copy, fills d
};

int gf2_rank(int m, int n, const std::vector<int32_t> &row_ptr, const std::vector<int32_t> &col_idx);

} // namespace swd
