// Device-side layout of one window Tanner graph (replaces the doubly-linked mod2sparse
// nodes of /root/reference/src/include/mod2sparse.h:46-82 on the hot path).
//
// HBM (read-only, shared by every shot; ~70 KB for the [[144,12,12]] (3,1) window so it stays
// L2-resident):
//   * checks are renumbered by DEGREE-DESCENDING order: "lane" l handles original check
//     perm[l].  Edge slots use the jagged-diagonal numbering slot(j, l) = jptr[j] + l
//     (j = position inside the row), so for a fixed j consecutive lanes touch consecutive
//     8-byte LDS words (conflict-free ds_read_b64 / ds_write_b64) and the slot count is
//     exactly nnz -- no ELL padding, which is what lets the fp64 messages of a 5976-edge
//     window fit the LDS budget;
//   * row_col[slot]  = column of that edge (coalesced for a fixed j);
//   * vn_edge[k*n+v] = k-th edge of column v in row-ascending order (the order the
//     reference's variable-node update sums in, osd_window.pyx:446-471), packed
//     slot | lane << 16 | j << 26; k-major so consecutive v are coalesced;
//   * vn_row[k*n+v]  = original row index of that edge (OSD works in original row order).
#pragma once
#include <stdint.h>

#define SWD_MAX_ROW_DEG 64   // livemask is one 64-bit word per check
#define SWD_MAX_COL_DEG 16
#define SWD_MAX_M 1024       // 10-bit lane field in vn_edge
#define SWD_MAX_E 65535      // 16-bit slot field
#define SWD_PAD_EDGE 0xFFFFFFFFu

struct SwdGraphDev {
    int32_t m, n, E, K, D;   // K = max row degree, D = max column degree
    int32_t new_n, rank, wm; // wm = ceil(m / 64)
    const uint16_t *jptr;    // [K+1]
    const uint16_t *row_col; // [E]
    const uint8_t *row_deg;  // [m] by lane
    const uint16_t *perm;    // [m] lane -> original check
    const uint16_t *iperm;   // [m] original check -> lane
    const uint32_t *vn_edge; // [D*n]
    const uint16_t *vn_row;  // [D*n]
    const uint8_t *col_deg;  // [n]
    const double *llr;       // [n]
    // The full-graph variable-node pass in TIERS (round 5): the nodes listed by decreasing degree in steps of two (stable in the
    // column index), so that the 64 nodes a wave serves in one cache row need the same number of message positions and the pass
    // skips the rest -- vperm[i] = the i-th listed node, vn_edge_s / llr_s = vn_edge / llr in listed order (no dependent load).
    const uint16_t *vperm;     // [n]
    const uint32_t *vn_edge_s; // [D*n]
    const double *llr_s;       // [n]
};

__host__ __device__ inline uint32_t swd_edge_slot(uint32_t e) { return e & 0xFFFFu; }
__host__ __device__ inline uint32_t swd_edge_lane(uint32_t e) { return (e >> 16) & 0x3FFu; }
__host__ __device__ inline uint32_t swd_edge_j(uint32_t e) { return e >> 26; }
