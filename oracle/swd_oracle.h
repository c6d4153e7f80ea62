/*
 * swd_oracle.h -- CPU restatement of the reference decoders (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle for the MI355X sliding-window decoder.  It restates, in plain
 * C over CSR/CSC index arrays, the algorithms of gongaa/SlidingWindowDecoder:
 *   osd_window            /root/reference/src/osd_window.pyx
 *   bp_history_decoder    /root/reference/src/bp_guessing_decoder.pyx:5-158
 *   bpgdg_decoder (gdg)   /root/reference/src/bp_guessing_decoder.pyx:160-442 + src/include/bpgd.cpp
 *   bpgd_decoder  (gd)    /root/reference/src/bp_guessing_decoder.pyx:473-571 + src/include/bpgd.cpp:258-286
 *   LU / OSD helpers      /root/reference/src/include/mod2sparse_extra.cpp:78-376
 *
 * It is pinned against golden vectors produced by the reference's own compiled Cython
 * extension (tests/golden/, generator script tests/golden/make_golden.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product (slidingwindowdecoder_amd) never links, imports or calls it.
 */
#ifndef SWD_ORACLE_H
#define SWD_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct swo_graph swo_graph;

/* CSR of an m x n GF(2) matrix (column indices ascending inside each row) + per-column
 * fault probabilities; llr = log((1-p)/p) as osd_window.pyx:113. */
swo_graph *swo_graph_create(int m, int n, const int32_t *row_ptr, const int32_t *col_idx,
                            const double *channel_probs);
void swo_graph_free(swo_graph *g);
int swo_graph_rank(const swo_graph *g); /* mod2sparse_extra.cpp:32-76 (value only) */

/* exit classes */
enum {
    SWO_EXIT_PRE = 0,      /* pre-processing BP converged   (osd_window.pyx:166-170) */
    SWO_EXIT_POST = 1,     /* post-processing BP converged  (osd_window.pyx:188-192) */
    SWO_EXIT_OSD = 2,      /* OSD ran                        (osd_window.pyx:193-195) */
    SWO_EXIT_FAIL_SET = 3, /* "setting vn failed"            (osd_window.pyx:179-181) */
    SWO_EXIT_FAIL_PEEL = 4,/* "peeling failed"               (osd_window.pyx:184-186) */
    SWO_EXIT_NO_OSD = 5    /* post BP failed and osd_order == -1 (osd_window.pyx:199) */
};

typedef struct {
    int32_t pre_max_iter;
    int32_t post_max_iter;
    double ms_scaling_factor;
    int32_t new_n;      /* <=0 : min(n, 2m) */
    int32_t osd_method; /* 0 osd_0, 1 osd_e, 2 osd_cs */
    int32_t osd_order;
} swo_osdw_params;

typedef struct {
    int32_t converge;
    int32_t bp_iteration;
    int32_t exit_class;
    int32_t reserved;
    double min_pm;
} swo_result;

/* One osd_window object: holds the persistent LLR history exactly like the reference
 * object does (osd_window.pyx:53-56; slots are never cleared between decodes). */
typedef struct swo_osdw swo_osdw;
swo_osdw *swo_osdw_create(const swo_graph *g, const swo_osdw_params *p);
void swo_osdw_free(swo_osdw *d);
void swo_osdw_clear_history(swo_osdw *d);
/* decode one syndrome; out[n] receives the returned vector (bp/osdw decoding). */
int swo_osdw_decode(swo_osdw *d, const uint8_t *synd, uint8_t *out, swo_result *res);
const double *swo_osdw_history(const swo_osdw *d);  /* n x 4 */
const uint8_t *swo_osdw_osd0(const swo_osdw *d);    /* n */
const uint8_t *swo_osdw_bp(const swo_osdw *d);      /* n */
/* batch helper for the CPU baseline: B syndromes, fresh history for every shot */
int swo_osdw_decode_batch(swo_osdw *d, int B, const uint8_t *synd, uint8_t *out,
                          swo_result *res);

/* ---- GDG / BPGD (bp_guessing_decoder.pyx) ---- */
typedef struct {
    int32_t max_iter;
    double ms_scaling_factor;
    int32_t max_iter_per_step;
    int32_t max_step;
    int32_t max_tree_depth;
    int32_t max_side_depth;
    int32_t max_tree_branch_step;
    int32_t max_side_branch_step;
    double gdg_factor;
    int32_t new_n; /* <=0 : min(n, 2m) */
    int32_t low_error_mode;
} swo_gdg_params;

typedef struct swo_gdg swo_gdg;
swo_gdg *swo_gdg_create(const swo_graph *g, const swo_gdg_params *p);
void swo_gdg_free(swo_gdg *d);
void swo_gdg_clear_history(swo_gdg *d);
/* mode 0: bpgdg_decoder.decode (single thread gdg); mode 1: bpgd_decoder.decode (gd);
 * mode 2: bp_history_decoder (plain BP only) */
/* mode 3: bpgdg_decoder.decode with multi_thread=True -- the threaded ensemble of bpgd.cpp:419-688 with the thread bodies run
 * in a fixed order (main, tree threads by id, side threads by index; ties of the path metric to the earliest) */
int swo_gdg_decode(swo_gdg *d, int mode, const uint8_t *synd, uint8_t *out, swo_result *res);
/* after a mode-3 decode: path metric per hypothesis (10000.0: not converged) -> number of hypotheses; the winner's index (-1:
 * none) and how many converged hypotheses share the winning metric with a different vector (> 0: the reference's own answer
 * depends on thread timing) */
int swo_gdg_ensemble_info(const swo_gdg *d, double *pm, int cap, int32_t *winner, int32_t *ties);
void swo_gdg_ensemble_blocks(const swo_gdg *d, int32_t *total, int32_t *prefix, double *prefix_unique);
const int *swo_gdg_cols(const swo_gdg *d); /* column order of the last post-processing (index_sort of the history sums) */
const double *swo_gdg_history(const swo_gdg *d);

/* ---- bp4_osd (src/bp4_osd.pyx): quaternary min-sum BP on (Hx, Hz) + one OSD per basis ---- */
typedef struct {
    int32_t max_iter;
    double ms_scaling_factor;
    int32_t osd_method; /* 0 osd_0, 1 osd_e, 2 osd_cs */
    int32_t osd_order;  /* -1: no OSD */
} swo_bp4_params;

typedef struct swo_bp4 swo_bp4;
/* Hx: mx x n CSR, Hz: mz x n CSR; px/py/pz: n Pauli error probabilities */
swo_bp4 *swo_bp4_create(int mx, int mz, int n, const int32_t *rpx, const int32_t *cix, const int32_t *rpz,
                        const int32_t *ciz, const double *px, const double *py, const double *pz,
                        const swo_bp4_params *p);
void swo_bp4_free(swo_bp4 *d);
/* out_x[n], out_z[n] = rows 0 and 1 of the array decode() returns; lpr[n*3] nullable = log_prob_ratios */
int swo_bp4_decode(swo_bp4 *d, const uint8_t *sx, const uint8_t *sz, uint8_t *out_x, uint8_t *out_z,
                   swo_result *res, double *lpr, uint8_t *osd0_x, uint8_t *osd0_z);
/* bp4_osd.camel_decode (src/bp4_osd.pyx:223-247); res->min_pm = the object's min_pm afterwards */
int swo_bp4_camel_decode(swo_bp4 *d, const uint8_t *sx, const uint8_t *sz, uint8_t *out_x, uint8_t *out_z, swo_result *res);
const signed char *swo_bp4_bp_decoding(const swo_bp4 *d, int z); /* bp_decoding_x / bp_decoding_z after the last call */
int swo_bp4_ranks(const swo_bp4 *d, int32_t *rank_x, int32_t *rank_z);

#ifdef __cplusplus
}
#endif
#endif
