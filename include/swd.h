/*
 * swd.h -- C ABI of the MI355X sliding-window QLDPC decoder (libswd_hip.so).
 *
 * The reference has no FFI layer: its boundary is the Cython extension-type surface
 * re-exported by /root/reference/src/__init__.py:2-4 and called one syndrome at a time from
 * Python loops (/root/reference/osd.py:152-167, guessing.py:160-197).  Each entry point below
 * names the reference interface it replaces.  Plain pointers and sizes only; integer return
 * codes (0 = ok, <0 = error, text via swd_last_error()); no exceptions, no exit().
 *
 * Conventions
 *   - check matrices are passed as CSR over GF(2): row_ptr[m+1], col_idx[nnz] (columns need
 *     not be sorted; duplicates are not allowed);
 *   - syndromes / error vectors are one byte per bit (0/1), row-major [shot][bit];
 *   - "_dev" entry points take DEVICE pointers (e.g. torch tensors' data_ptr()) and a
 *     hipStream_t passed as void*; the others take HOST pointers and copy.
 */
#ifndef SWD_H
#define SWD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SWD_ABI_VERSION 4

/* exit class of one window decode, low byte of status[]; bit 8 = converge flag */
enum {
    SWD_EXIT_PRE = 0,       /* pre-processing BP converged    osd_window.pyx:166-170 */
    SWD_EXIT_POST = 1,      /* post-processing BP converged   osd_window.pyx:188-192 */
    SWD_EXIT_OSD = 2,       /* OSD produced the answer        osd_window.pyx:193-195 */
    SWD_EXIT_FAIL_SET = 3,  /* "setting vn failed"            osd_window.pyx:179-181 */
    SWD_EXIT_FAIL_PEEL = 4, /* "peeling failed"               osd_window.pyx:184-186 */
    SWD_EXIT_NO_OSD = 5,    /* BP failed and osd_order == -1  osd_window.pyx:199     */
    SWD_EXIT_SCHED_FAULT = 6 /* pipeline only: the window's predecessor never finished (see swd_pipeline_status);
                                nothing was decoded or committed for this window */
};
#define SWD_STATUS_CONVERGE 0x100

typedef struct swd_graph_desc {
    int32_t m, n, nnz;
    const int32_t *row_ptr;      /* m+1 */
    const int32_t *col_idx;      /* nnz */
    const double *channel_probs; /* n; llr = log((1-p)/p) as osd_window.pyx:113 */
} swd_graph_desc;

/* kwargs of osd_window.__cinit__ (osd_window.pyx:10-16) */
typedef struct swd_osdw_params {
    int32_t pre_max_iter;  /* default 8   */
    int32_t post_max_iter; /* default 100 */
    double ms_scaling_factor;
    int32_t new_n;      /* <=0: min(n, 2m) (osd_window.pyx:60-63) */
    int32_t osd_method; /* 0 osd_0, 1 osd_e, 2 osd_cs (osd_window.pyx:69-79) */
    int32_t osd_order;  /* -1 disables OSD */
} swd_osdw_params;

typedef struct swd_osdw swd_osdw;

const char *swd_last_error(void);
int swd_abi_version(void);
int swd_device_count(void);

/* replaces osd_window(pcm, **kwargs)  (osd_window.pyx:8-126).  device = HIP ordinal. */
swd_osdw *swd_osdw_create(const swd_graph_desc *g, const swd_osdw_params *p, int device);
void swd_osdw_destroy(swd_osdw *d);
/* properties fixed at construction: rank (osd_window.pyx:87), new_n, m, n */
int swd_osdw_info(const swd_osdw *d, int32_t *m, int32_t *n, int32_t *new_n, int32_t *rank);

/* per-decode record, SWD_STAT_WORDS int32 words:
 *   [0] exit class | SWD_STATUS_CONVERGE   (property `converge`)
 *   [1] property `bp_iteration` (pre + post iterations executed)
 *   [2] pre-processing iterations   [3] post-processing iterations
 *   [4] live variable nodes, [5] live checks, [6] live edges of the shortened graph (post phase)
 *   [7] GF(2) row additions applied by the OSD elimination
 * words 2..7 feed the algorithmic-bytes accounting of bench.py. */
#define SWD_STAT_WORDS 8

/* replaces the per-shot loop around osd_window.decode (osd.py:166-167): B independent
 * syndromes in one launch.  Host pointers.
 *   synd   [B*m]   in
 *   out    [B*n]   returned vector of decode() (bp_decoding or osdw_decoding)
 *   stats  [B*SWD_STAT_WORDS]
 *   min_pm [B]     property `min_pm`
 *   hist   [B*4*n] nullable; LLR history, layout [shot][slot][vn]  (property log_prob_ratios is
 *                  its transpose).  If hist_is_state != 0 the buffer is read as the initial
 *                  history too (the reference object keeps it between decodes); otherwise
 *                  every shot starts from a zero history like a freshly built object.
 *   osd0   [B*n]   nullable; property osd0_decoding (only written for SWD_EXIT_OSD shots)
 *   bp_dec [B*n]   nullable; for SWD_EXIT_OSD shots the BP hard decisions the OSD started from (property
 *                  bp_decoding after such a decode, osd_window.pyx:499-501); for every other exit class
 *                  bp_decoding is `out` itself and bp_dec is left untouched                               */
int swd_osdw_decode_batch(swd_osdw *d, int32_t B, const uint8_t *synd, uint8_t *out,
                          int32_t *stats, double *min_pm, double *hist, int32_t hist_is_state,
                          uint8_t *osd0, uint8_t *bp_dec);

/* same, device-resident buffers, asynchronous on `stream` (hipStream_t).  stats / min_pm / hist /
 * osd0 / bp_dec may be NULL; strides are in bytes between consecutive shots (0 = dense). */
int swd_osdw_decode_batch_dev(swd_osdw *d, int32_t B, const uint8_t *synd, int64_t synd_stride,
                              uint8_t *out, int64_t out_stride, int32_t *stats, double *min_pm,
                              double *hist, int32_t hist_is_state, uint8_t *osd0, uint8_t *bp_dec, void *stream);

/* average duration (ms) of the decode kernel launches since the last call, measured with HIP
 * events on the launch stream when timing was enabled with swd_osdw_set_timing(d, 1) */
int swd_osdw_set_timing(swd_osdw *d, int32_t on);
int swd_osdw_get_timing(swd_osdw *d, double *total_ms, int64_t *launches);

/* ---- guessing decoders --------------------------------------------------------------------
 * Replace bpgdg_decoder (single-thread gdg(), the deterministic path), bpgd_decoder and
 * bp_history_decoder of /root/reference/src/bp_guessing_decoder.pyx.  kwargs as in
 * bp_guessing_decoder.pyx:7-9, 162-171, 475-478.  stats words for these decoders:
 *   [0] exit class | SWD_STATUS_CONVERGE (property `converge`), [1] BP iterations in total,
 *   [2] pre-processing iterations, [3] iterations inside decimation steps, [4] snapshots pushed,
 *   [5] BP blocks run, [6] min_converge_depth, [7] scheduling diagnostic, not part of the result:
 *   snapshots of the main branch that ran as work items (0 whenever the launch took the serial
 *   tree walk: large batches, batches of a stream object, posterior history requested).          */
typedef struct swd_gdg_params {
    int32_t max_iter;            /* pre-processing BP iterations (default 50; 8 in the notebooks) */
    double ms_scaling_factor;
    int32_t max_iter_per_step;   /* 6  */
    int32_t max_step;            /* 25 */
    int32_t max_tree_depth;      /* 3  */
    int32_t max_side_depth;      /* 10 */
    int32_t max_tree_branch_step;/* 10 (multi-thread ensemble only; unused by gdg()) */
    int32_t max_side_branch_step;/* 10 */
    double gdg_factor;           /* gdg_factor / gd_factor */
    int32_t new_n;               /* <=0: min(n, 2m) */
    int32_t low_error_mode;
    int32_t mode;                /* 0 bpgdg_decoder, 1 bpgd_decoder, 2 bp_history_decoder */
    int32_t multi_thread;        /* 0: the single-thread gdg() (bit-exact; side branches of one shot run concurrently on different
                                    workgroups).  1: bpgdg_decoder(multi_thread=True), the threaded ensemble of bpgd.cpp:419-688
                                    (main thread, 2^D - 1 tree threads, S - D side threads; kernel kind 7) with the thread
                                    bodies in a fixed order -- identical to the reference on every syndrome whose winning path
                                    metric is not shared by a second, different vector (statistics word 7 counts those; the
                                    reference's own answer depends on thread timing there).  stats for this mode: [4]
                                    hypotheses run, [5] BP blocks, [6] winner (0 main, 1.. tree ids, then side threads; -1 none),
                                    [7] tied hypotheses with a different vector.  Every decode has the state of a NEWLY BUILT
                                    reference object: min_pm_error starts as zeros (when BPGD::reset fails the zero vector comes back; a
                                    reference object that is re-used returns its PREVIOUS decode's vector there, bpgd.cpp:583, 619-625)
                                    and each thread's 4-slot posterior history starts as zeros (a re-used reference thread keeps its
                                    own stale slots when max_iter_per_step < 4).  The oracle and oracle/ref_shim.cpp build a fresh
                                    object per decode too.  2: NOT a reference mode -- every leaf of gdg()'s
                                    decimation tree counts (no min_converge_depth pruning, no snapshot cap), smallest path metric
                                    wins, ties to the earliest in stack order. */
} swd_gdg_params;

typedef struct swd_gdg swd_gdg;
swd_gdg *swd_gdg_create(const swd_graph_desc *g, const swd_gdg_params *p, int device);
void swd_gdg_destroy(swd_gdg *d);
/* host pointers; hist [B*4*n] nullable as in swd_osdw_decode_batch (pre-processing BP history) */
int swd_gdg_decode_batch(swd_gdg *d, int32_t B, const uint8_t *synd, uint8_t *out, int32_t *stats,
                         double *min_pm, double *hist, int32_t hist_is_state);
int swd_gdg_decode_batch_dev(swd_gdg *d, int32_t B, const uint8_t *synd, int64_t synd_stride,
                             uint8_t *out, int64_t out_stride, int32_t *stats, double *min_pm,
                             void *stream);

/* ---- quaternary BP + OSD -------------------------------------------------------------------
 * Replaces bp4_osd(Hx, Hz, channel_probs_x/y/z, max_iter, ms_scaling_factor, osd_method, osd_order)
 * (/root/reference/src/bp4_osd.pyx:8-140) and its decode(sx, sz) (bp4_osd.pyx:197-221).  The
 * variable-node update uses exp/log1p; the device evaluates them with the algorithms of the C library the
 * reference links (glibc >= 2.28: table-driven exp in its FMA build, fdlibm log1p -- csrc/swd_libm.h), so
 * posterior LLRs, decisions and OSD orderings are bit-identical to a reference run on an FMA-capable x86-64.
 * Restrictions (swd_bp4_create fails with a message): column weight of Hx / Hz <= 10; osd_order > 0 needs
 * rank(Hx) >= rank(Hz) -- the reference sizes BOTH sweeps with kx = n - rank_x (bp4_osd.pyx:103-104, :284): with
 * rank(Hx) > rank(Hz) its z-basis sweep walks fewer candidate columns than exist, which is reproduced (pinned by
 * tests/golden/bp4_unequal_ranks.npz); with rank(Hx) < rank(Hz) it reads past its column array, which is not. */
typedef struct swd_bp4_params {
    int32_t max_iter;          /* default 32 */
    double ms_scaling_factor;
    int32_t osd_method;        /* 0 osd_0, 1 osd_e, 2 osd_cs */
    int32_t osd_order;         /* -1 disables OSD */
} swd_bp4_params;
typedef struct swd_bp4 swd_bp4;
swd_bp4 *swd_bp4_create(const swd_graph_desc *hx, const swd_graph_desc *hz, const double *px, const double *py,
                        const double *pz, const swd_bp4_params *p, int device); /* channel_probs of hx/hz ignored */
void swd_bp4_destroy(swd_bp4 *d);
int swd_bp4_info(const swd_bp4 *d, int32_t *mx, int32_t *mz, int32_t *n, int32_t *rank_x, int32_t *rank_z);
/* sx [B*mx], sz [B*mz] -> out [B*2*n] (row 0: X string, row 1: Z string, like the (2, n) array decode()
 * returns); stats [B*SWD_STAT_WORDS] ([0] exit|converge, [1] bp_iteration); lpr [B*3*n] nullable posterior
 * LLRs laid out [shot][x|y|z][vn] (property log_prob_ratios transposed); osd0 [B*2*n] nullable;
 * bp_dec [B*2*n] nullable: BP hard decisions at exit (properties bp_decoding_x / bp_decoding_z). */
int swd_bp4_decode_batch(swd_bp4 *d, int32_t B, const uint8_t *sx, const uint8_t *sz, uint8_t *out,
                         int32_t *stats, double *lpr, uint8_t *osd0, uint8_t *bp_dec);
int swd_bp4_decode_batch_dev(swd_bp4 *d, int32_t B, const uint8_t *sx, const uint8_t *sz, uint8_t *out,
                             int32_t *stats, double *lpr, uint8_t *osd0, uint8_t *bp_dec, void *stream);
/* bp4_osd.camel_decode (/root/reference/src/bp4_osd.pyx:223-247, called by Misc.ipynb): the last qubit is fixed to
 * I, X, Z, Y in turn, plain BP4 decodes the rest, the converged run of smallest path metric wins (ties: the
 * earliest).  out [B*2*n]; stats [B*SWD_STAT_WORDS] ([0] converge flag, [1] bp_iteration of the last run,
 * [5] winning Pauli or -1); min_pm [B] nullable (10000.0 when no run converged).  Every shot has the state of a
 * newly constructed reference object: without a converged run the zero vectors are returned. */
int swd_bp4_camel_decode_batch(swd_bp4 *d, int32_t B, const uint8_t *sx, const uint8_t *sz, uint8_t *out,
                               int32_t *stats, double *min_pm);
int swd_bp4_camel_decode_batch_dev(swd_bp4 *d, int32_t B, const uint8_t *sx, const uint8_t *sz, uint8_t *out,
                                   int32_t *stats, double *min_pm, void *stream);

/* ---- DEM sampler -----------------------------------------------------------------------------
 * Replaces `dem.compile_sampler().sample(shots)` of the reference harness (/root/reference/osd.py:124-125,
 * guessing.py:129-130; Stim is not available here): per shot, faults e ~ Bernoulli(priors) over the
 * columns of the detector error model, det = chk e, observable flips = obs e over GF(2).
 * chk: num_det x num_col CSR with channel_probs = priors; obs: (<= 32) x num_col CSR or NULL.
 * Stream: Philox4x32-10 keyed by `seed`, counter (shot, column / 4); fault iff x < round(p 2^32); shot b of
 * a call is shot number first_shot + b, so a result does not depend on batching or on the rank. */
typedef struct swd_sampler swd_sampler;
swd_sampler *swd_sampler_create(const swd_graph_desc *chk, const swd_graph_desc *obs, int device);
void swd_sampler_destroy(swd_sampler *s);
int swd_sampler_info(const swd_sampler *s, int32_t *num_det, int32_t *num_col, int32_t *num_obs);
/* det [B*num_det] u8; obs_flips [B] bit masks, nullable; faults [B*num_col] u8, nullable.  Host pointers. */
int swd_sampler_sample(swd_sampler *s, int32_t B, uint64_t seed, uint64_t first_shot, uint8_t *det,
                       uint32_t *obs_flips, uint8_t *faults);
/* device pointers (strides in bytes per shot, 0 = dense), asynchronous on `stream` */
int swd_sampler_sample_dev(swd_sampler *s, int32_t B, uint64_t seed, uint64_t first_shot, uint8_t *det,
                           int64_t det_stride, uint32_t *obs_flips, uint8_t *faults, int64_t faults_stride,
                           void *stream);

/* ---- sliding-window pipeline ---------------------------------------------------------------
 * Replaces the window loop of the reference harness (/root/reference/osd.py:130-179, identical in
 * guessing.py:135-214 and the notebooks): for every shot, decode window t on the residual
 * syndrome, commit the first `commit` columns of the estimate into total_e_hat, update the
 * residual syndrome det ^ chk * total_e_hat, go to window t+1.  One launch for B shots. */
typedef struct swd_window_desc {
    swd_graph_desc graph; /* window matrix incl. the merged noisy-syndrome identity (osd.py:103-113) */
    int32_t row0;         /* first detector row of the window            (anchors[top_left][0]) */
    int32_t col0;         /* first global column of the window           (anchors[top_left][1]) */
    int32_t commit;       /* committed leading columns (osd.py:140,170-173) */
    int32_t reserved;
} swd_window_desc;

typedef struct swd_pipeline swd_pipeline;

/* chk = region-permuted global check matrix [num_det x num_col] (CSR), used for the residual
 * update (osd.py:178).  All windows share the decoder parameters p (osd.py:152-161). */
swd_pipeline *swd_pipeline_create(int32_t num_windows, const swd_window_desc *wins,
                                  const swd_graph_desc *chk, const swd_osdw_params *p, int device);
/* same window loop with a guessing decoder in every window (guessing.py:135-214) */
swd_pipeline *swd_pipeline_create_gdg(int32_t num_windows, const swd_window_desc *wins,
                                      const swd_graph_desc *chk, const swd_gdg_params *p, int device);
void swd_pipeline_destroy(swd_pipeline *pl);
int swd_pipeline_info(const swd_pipeline *pl, int32_t *num_windows, int32_t *num_det,
                      int32_t *num_col, int32_t *lds_bytes, int32_t *threads);
/* optional: observables matrix obs [num_obs x num_col] (CSR, num_obs <= 32) for the on-device
 * logical accounting of osd.py:184-187 */
int swd_pipeline_set_observables(swd_pipeline *pl, const swd_graph_desc *obs);
/* det [B*num_det] in; total [B*num_col] out (total_e_hat); stats [B*W*SWD_STAT_WORDS], min_pm
 * [B*W] and shot_result [B*2] nullable.  shot_result[2b] = bit mask of the observables the
 * committed faults flip (obs @ total_e_hat, needs swd_pipeline_set_observables), shot_result[2b+1]
 * = 1 if the residual syndrome det ^ chk @ total_e_hat is non-zero ("flagged").  Host pointers. */
int swd_pipeline_decode(swd_pipeline *pl, int32_t B, const uint8_t *det, uint8_t *total,
                        int32_t *stats, double *min_pm, int32_t *shot_result);
/* device pointers, asynchronous on `stream` */
int swd_pipeline_decode_dev(swd_pipeline *pl, int32_t B, const uint8_t *det, int64_t det_stride,
                            uint8_t *total, int64_t total_stride, int32_t *stats, double *min_pm,
                            int32_t *shot_result, void *stream);
/* Scheduling-fault flags accumulated by every launch of this pipeline since the last call, read and cleared
 * (0 = none; bit 0: a window waited more than 10 s for its predecessor window of the same shot, which cannot
 * happen by construction -- its statistics record SWD_EXIT_SCHED_FAULT; bit 1: a workgroup of the guessing decoders'
 * work-item loop found nothing to do for 20 s while units were still outstanding and left -- results incomplete).  Synchronises the device, so call it
 * after the asynchronous swd_pipeline_decode_dev launches it should cover; swd_pipeline_decode checks it itself. */
int swd_pipeline_status(swd_pipeline *pl, uint32_t *flags);

/* total_e_hat bit-packed: total_bits [B * ((num_col + 7) / 8)], bit (c & 7) of byte (c >> 3) of a shot's row = column c
 * (numpy: np.unpackbits(bits, axis=1, count=num_col, bitorder="little")).  1098 B per shot instead of 8784 for the [[144,12,12]]
 * experiment -- the per-shot output SURVEY section 8(e) counts.  Host pointers; otherwise as swd_pipeline_decode. */
int swd_pipeline_decode_packed(swd_pipeline *pl, int32_t B, const uint8_t *det, uint8_t *total_bits,
                               int32_t *stats, double *min_pm, int32_t *shot_result);

/* ---- streaming form of the window loop -------------------------------------------------------
 * Consecutive batches through ONE pipeline, two in flight (what a deployment, or the shots loop of the reference harness
 * /root/reference/osd.py:130-191 cut into batches, does): a stream object owns two lanes, each with its own HIP stream, launch
 * slot, device buffers and page-locked staging.  Batch k + 1 is copied in and launched while batch k still runs: its persistent
 * grid takes the workgroup slots that the tail of batch k leaves empty, and the copy-out / unpacking of batch k overlaps the
 * launch of k + 1.  Results are those of swd_pipeline_decode, batch by batch, in push order.  (Guessing decoders: a stream batch of
 * 3072 shots or more walks each decimation tree serially instead of spreading its side branches over the grid as work items --
 * the next batch fills the tail the serial walk leaves; same results, statistics word 7 aside.  A caller that alternates streams of
 * its own with swd_pipeline_decode_dev gets the same choice: a launch that finds the handle's previous launch still running on
 * another stream is treated as a stream batch.)
 *   flags  SWD_STREAM_PACKED    total_e_hat is returned bit-packed (layout of swd_pipeline_decode_packed)
 *          SWD_STREAM_NO_STATS  per-window stats / min_pm are not copied back (pop takes NULL for them)
 * Host form: push(det [B*num_det]) enqueues copy-in + launch + copy-out on the next lane and returns at once; pop() waits for
 * the OLDEST batch in flight and fills the caller's arrays (total [B*num_col] bytes, or [B*((num_col+7)/8)] with
 * SWD_STREAM_PACKED; stats / min_pm / shot_result nullable), returns its B.  At most two batches in flight: a third push without a
 * pop fails.  A scheduling fault of the batch (see swd_pipeline_status) makes its pop fail.
 * Device form: push_dev launches on the next lane with the caller's device buffers (the caller keeps one set of output buffers
 * per lane, i.e. alternates between two); `after` (hipStream_t) = a stream whose work so far must precede the launch: the
 * producer of det and whoever still reads the lane's output buffers.  NULL means the legacy default stream, as everywhere in HIP
 * (the lanes are non-blocking streams and do not synchronise with it by themselves); SWD_STREAM_NO_DEPENDENCY skips the wait
 * (inputs and output buffers already synchronised by the caller).  wait(stream): `stream` waits for both lanes (device-side), or
 * with NULL the host does.
 * Lifetime: destroy the stream objects of a pipeline before the pipeline.  The other order is tolerated -- swd_pipeline_destroy
 * drains and detaches the live stream objects, every later call on them fails with a message, swd_pipeline_stream_destroy still
 * frees them -- but must not race with calls on those stream objects from other threads.
 * A scheduling fault (see swd_pipeline_status) is reported by the pop of the batch it happened in: every launch has a fault word
 * of its own next to the decoder's sticky one. */
typedef struct swd_stream swd_stream;
#define SWD_STREAM_PACKED 1
#define SWD_STREAM_NO_STATS 2
#define SWD_STREAM_NO_DEPENDENCY ((void *)(intptr_t)-1)
swd_stream *swd_pipeline_stream_create(swd_pipeline *pl, int32_t max_shots, int32_t flags);
void swd_pipeline_stream_destroy(swd_stream *s);
int swd_pipeline_stream_push(swd_stream *s, int32_t B, const uint8_t *det);
int swd_pipeline_stream_pop(swd_stream *s, uint8_t *total, int32_t *stats, double *min_pm, int32_t *shot_result);
int swd_pipeline_stream_pending(swd_stream *s); /* host batches pushed and not yet popped (0..2) */
int swd_pipeline_stream_push_dev(swd_stream *s, int32_t B, const uint8_t *det, int64_t det_stride, uint8_t *total,
                                 int64_t total_stride, int32_t *stats, double *min_pm, int32_t *shot_result, void *after);
int swd_pipeline_stream_wait(swd_stream *s, void *stream);

/* Threading and streams: every entry point may be called from any host thread.  Launches of ONE decoder /
 * pipeline handle are serialised on the host while they are prepared; on the device, launches on different streams
 * run concurrently -- each launch takes its scheduling scratch from a ring of four launch slots, and a fifth launch
 * in flight makes its stream wait for the first (hipStreamWaitEvent).  The same holds for swd_bp4 handles (the queue
 * of unconverged decodes between the BP and the OSD kernel, the internal posterior buffer and the camel_decode
 * scratch belong to the launch slot).  Host-buffer entry points of one handle are mutually exclusive for their
 * whole duration (they share staging buffers). */

/* diagnostics: device-side phase timers (100 MHz ticks) of the last launch, out [B*W*8]:
 * init, pre BP, sort, shorten+peel, post BP, OSD sort, OSD elimination, OSD sweep + epilogue */
int swd_pipeline_set_profiling(swd_pipeline *pl, int32_t on);
int swd_pipeline_get_profile(swd_pipeline *pl, int32_t B, int64_t *out);
int swd_pipeline_set_timing(swd_pipeline *pl, int32_t on);
int swd_pipeline_get_timing(swd_pipeline *pl, double *total_ms, int64_t *launches);

/* diagnostics: occupies `blocks` workgroups of `threads` threads with `lds_bytes` of LDS each for `microseconds` (bounded: at most
 * 2 s) on `stream` -- a foreign, long-running kernel for tests of the persistent grids' forward progress under reduced
 * residency (tests/test_gpu_forward_progress.py: blocks that each take more than half a CU's LDS hold one CU apiece). */
int swd_diag_occupy(int device, int32_t blocks, int32_t threads, int32_t lds_bytes, int32_t microseconds, void *stream);

#ifdef __cplusplus
}
#endif
#endif
