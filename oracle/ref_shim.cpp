// Thin extern "C" driver over the reference's own C/C++ sources (compiled where they lie under
// /root/reference/src/include by `make ref`): mod2sparse.c, mod2sparse_extra.cpp, bpgd.cpp.
// TEST INFRASTRUCTURE ONLY (validates oracle/swd_oracle.c); never linked into the product.
// Only the pieces of the path that exist as C/C++ in the reference are reachable this way: the
// sparse GF(2) LU used by OSD (mod2sparse_extra.cpp:78-376), index_sort (bpgd.cpp:384-389) and the
// BPGD worker class (bpgd.hpp:12-48).  The BP loop of osd_window / bp4_osd lives in Cython.
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include "bpgd.hpp"
#include "mod2sparse_extra.hpp"

static mod2sparse *from_csr(int m, int n, const int32_t *rp, const int32_t *ci) {
    mod2sparse *H = mod2sparse_allocate(m, n);
    for (int i = 0; i < m; i++)
        for (int k = rp[i]; k < rp[i + 1]; k++) mod2sparse_insert(H, i, ci[k]);
    return H;
}

extern "C" {

void *ref_pcm_new(int m, int n, const int32_t *rp, const int32_t *ci) { return from_csr(m, n, rp, ci); }
void ref_pcm_free(void *H) { mod2sparse_free((mod2sparse *)H); free(H); }
int ref_rank(void *H) { return mod2sparse_rank((mod2sparse *)H); }
void ref_index_sort(double *v, int32_t *cols, int n) { index_sort(v, (int *)cols, n); }

// OSD-0 for a given column priority order: cols[] in (priority order), out (solution over the n columns
// of H), cols[] out (order after the decomposition: pivots first).  Call shape of osd_window.pyx:220-229.
int ref_osd0(void *Hv, int rank, int32_t *cols, const uint8_t *synd, uint8_t *out) {
    mod2sparse *H = (mod2sparse *)Hv;
    const int m = mod2sparse_rows(H), n = mod2sparse_cols(H);
    mod2sparse *L = mod2sparse_allocate(m, rank), *U = mod2sparse_allocate(rank, n);
    int *rows = (int *)calloc(m, sizeof(int));
    char *z = (char *)calloc(m, 1), *x = (char *)calloc(n, 1);
    for (int i = 0; i < m; i++) z[i] = (char)synd[i];
    int rc = mod2sparse_decomp_osd(H, rank, L, U, rows, (int *)cols);
    LU_forward_backward_solve(L, U, rows, (int *)cols, z, x);
    for (int j = 0; j < n; j++) out[j] = (uint8_t)x[j];
    mod2sparse_free(L); free(L); mod2sparse_free(U); free(U); free(rows); free(z); free(x);
    return rc;
}

void *ref_bpgd_new(int m, int n, int num_iter, int low_error_mode, double factor) {
    return new BPGD(m, n, num_iter, low_error_mode, factor);
}
void ref_bpgd_free(void *b) { delete (BPGD *)b; }
int ref_bpgd_reset(void *b, void *H, int32_t *cols, double *llr_prior, const uint8_t *synd) {
    BPGD *p = (BPGD *)b;
    char *s = (char *)malloc(p->m);
    for (int i = 0; i < p->m; i++) s[i] = (char)synd[i];
    int rc = p->reset((mod2sparse *)H, (int *)cols, llr_prior, s);
    free(s);
    return rc;
}
int ref_bpgd_min_sum_log(void *b) { return ((BPGD *)b)->min_sum_log(); }
int ref_bpgd_decimate_vn_reliable(void *b, int depth, double fraction) {
    return ((BPGD *)b)->decimate_vn_reliable(depth, fraction);
}
int ref_bpgd_select_vn(void *b, int depth, int32_t *guess_vn) {
    int g = -1;
    int rc = ((BPGD *)b)->select_vn(depth, g);
    *guess_vn = g;
    return rc;
}
int ref_bpgd_vn_set_value(void *b, int vn, int value) { return ((BPGD *)b)->vn_set_value(vn, (char)value); }
int ref_bpgd_peel(void *b) { return ((BPGD *)b)->peel(); }
double ref_bpgd_get_pm(void *b) { return ((BPGD *)b)->get_pm(); }
int ref_bpgd_num_active_vn(void *b) { return ((BPGD *)b)->num_active_vn; }
void ref_bpgd_error(void *b, uint8_t *out) {
    BPGD *p = (BPGD *)b;
    for (int v = 0; v < p->n; v++) out[v] = (uint8_t)p->error[v];
}
void ref_bpgd_llr_posterior(void *b, double *out) {  // n x 4
    BPGD *p = (BPGD *)b;
    for (int v = 0; v < p->n; v++) memcpy(out + 4 * v, p->llr_posterior[v], 4 * sizeof(double));
}

// The reference's threaded ensemble itself (BPGD_main_thread::do_work, bpgd.cpp:591-688, spawns its tree and side threads):
// one decode of the post-processing for a given column order.  Besides the shared result (min_pm, min_pm_error) the
// per-thread path metrics are read back from the thread objects -- each is a pure function of the inputs, only the strict-<
// update of the shared result depends on thread timing (on exact ties) -- so the oracle's restatement can be held to every
// thread's metric bit for bit.  pms: [num_tree + num_side], 10000.0 = not converged.  Returns num_tree | num_side << 16.
int ref_gdg_multi(void *Hv, int m, int new_n, int num_iter, int max_step, int max_tree_depth, int max_side_depth, int max_tree_step,
                  int max_side_step, int low_error_mode, double factor, int32_t *cols, double *llr, const uint8_t *synd,
                  uint8_t *min_pm_error, double *min_pm, double *pms) {
    BPGD_main_thread mt(m, new_n, num_iter, max_step, max_tree_depth, max_side_depth, max_tree_step, max_side_step, low_error_mode, factor);
    char *s = (char *)malloc(m);
    for (int i = 0; i < m; i++) s[i] = (char)synd[i];
    mt.do_work((mod2sparse *)Hv, (int *)cols, llr, s);
    free(s);
    for (int v = 0; v < new_n; v++) min_pm_error[v] = (uint8_t)mt.min_pm_error[v];
    *min_pm = mt.min_pm;
    int k = 0;
    for (auto &t : mt.bpgd_tree_vec) pms[k++] = t->min_pm;
    for (auto &t : mt.bpgd_side_vec) pms[k++] = t->min_pm;
    return mt.num_tree_threads | (mt.num_side_threads << 16);
}

// The same, on ONE BPGD_main_thread object that the caller keeps across decodes -- what the reference's bpgdg_decoder does
// (bp_guessing_decoder.pyx:238-251 calls do_work on the object built in __cinit__).  do_work resets min_pm but not min_pm_error
// (bpgd.cpp:599), and returns early when BPGD::reset fails (:619-625): a re-used object then hands back its PREVIOUS decode's
// vector, where a fresh object (ref_gdg_multi above, the oracle, the device) hands back zeros.  tests/test_oracle_vs_ref.py
// measures exactly that difference.
void *ref_gdg_multi_new(int m, int new_n, int num_iter, int max_step, int max_tree_depth, int max_side_depth, int max_tree_step,
                        int max_side_step, int low_error_mode, double factor) {
    return new BPGD_main_thread(m, new_n, num_iter, max_step, max_tree_depth, max_side_depth, max_tree_step, max_side_step, low_error_mode, factor);
}
void ref_gdg_multi_free(void *o) { delete (BPGD_main_thread *)o; }
int ref_gdg_multi_decode(void *o, void *Hv, int m, int new_n, int32_t *cols, double *llr, const uint8_t *synd, uint8_t *min_pm_error,
                         double *min_pm, double *pms) {
    BPGD_main_thread &mt = *(BPGD_main_thread *)o;
    char *s = (char *)malloc(m);
    for (int i = 0; i < m; i++) s[i] = (char)synd[i];
    mt.do_work((mod2sparse *)Hv, (int *)cols, llr, s);
    free(s);
    for (int v = 0; v < new_n; v++) min_pm_error[v] = (uint8_t)mt.min_pm_error[v];
    *min_pm = mt.min_pm;
    int k = 0;
    for (auto &t : mt.bpgd_tree_vec) pms[k++] = t->min_pm;
    for (auto &t : mt.bpgd_side_vec) pms[k++] = t->min_pm;
    return mt.num_tree_threads | (mt.num_side_threads << 16);
}

}  // extern "C"
