/*
 * swd_oracle.c -- CPU restatement of the reference decoders.  TEST INFRASTRUCTURE ONLY:
 * see swd_oracle.h for who may use it.  Every routine cites the reference lines it follows
 * (paths relative to /root/reference).
 *
 * Storage differs from the reference on purpose (flat CSR/CSC arrays instead of the
 * doubly-linked mod2sparse nodes, bit-packed dense rows for the LU) but every floating
 * point operation is issued in the reference's order, so results are bit-identical; that
 * claim is what tests/test_oracle_golden.py checks against vectors emitted by the
 * reference's own compiled extension.
 */
#include "swd_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------ */
/* Tanner graph with message storage (stands in for mod2sparse + mod2entry,              */
/* src/include/mod2sparse.h:46-82)                                                        */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    int m, n, nnz;
    int *row_ptr, *col_idx; /* CSR, columns ascending in a row            */
    int *col_ptr, *row_idx; /* CSC, rows ascending in a column            */
    int *c2r;               /* CSC position -> CSR edge id                */
    double *b2c, *c2b;      /* bit_to_check / check_to_bit per CSR edge   */
    int *sgn;               /* mod2entry.sgn                              */
} tanner;

static void *xcalloc(size_t n, size_t s) {
    void *p = calloc(n ? n : 1, s);
    if (!p) { fprintf(stderr, "swd_oracle: out of memory\n"); abort(); }
    return p;
}

static void tanner_finish(tanner *t) {
    /* build CSC from CSR */
    int m = t->m, n = t->n, nnz = t->nnz;
    t->col_ptr = xcalloc(n + 1, sizeof(int));
    t->row_idx = xcalloc(nnz, sizeof(int));
    t->c2r = xcalloc(nnz, sizeof(int));
    for (int e = 0; e < nnz; e++) t->col_ptr[t->col_idx[e] + 1]++;
    for (int v = 0; v < n; v++) t->col_ptr[v + 1] += t->col_ptr[v];
    int *fill = xcalloc(n, sizeof(int));
    for (int c = 0; c < m; c++)
        for (int e = t->row_ptr[c]; e < t->row_ptr[c + 1]; e++) {
            int v = t->col_idx[e];
            int k = t->col_ptr[v] + fill[v]++;
            t->row_idx[k] = c;
            t->c2r[k] = e;
        }
    free(fill);
    t->b2c = xcalloc(nnz, sizeof(double));
    t->c2b = xcalloc(nnz, sizeof(double));
    t->sgn = xcalloc(nnz, sizeof(int));
}

static void tanner_init_csr(tanner *t, int m, int n, const int32_t *row_ptr, const int32_t *col_idx) {
    memset(t, 0, sizeof(*t));
    t->m = m; t->n = n; t->nnz = row_ptr[m];
    t->row_ptr = xcalloc(m + 1, sizeof(int));
    t->col_idx = xcalloc(t->nnz, sizeof(int));
    memcpy(t->row_ptr, row_ptr, (m + 1) * sizeof(int));
    memcpy(t->col_idx, col_idx, t->nnz * sizeof(int));
    tanner_finish(t);
}

static void tanner_free(tanner *t) {
    free(t->row_ptr); free(t->col_idx); free(t->col_ptr); free(t->row_idx); free(t->c2r);
    free(t->b2c); free(t->c2b); free(t->sgn);
    memset(t, 0, sizeof(*t));
}

/* sub-matrix of the columns cols[0..nsub) of src, new column j = src column cols[j]
 * (mod2sparse_copycols, src/include/mod2sparse.c:239-272, as used by BPGD::reset
 * src/include/bpgd.cpp:200-202) */
static void tanner_copycols(const tanner *src, const int *cols, int nsub, tanner *dst) {
    int m = src->m;
    memset(dst, 0, sizeof(*dst));
    dst->m = m; dst->n = nsub;
    int *newid = xcalloc(src->n, sizeof(int));
    for (int v = 0; v < src->n; v++) newid[v] = -1;
    for (int j = 0; j < nsub; j++) newid[cols[j]] = j;
    dst->row_ptr = xcalloc(m + 1, sizeof(int));
    int nnz = 0;
    for (int c = 0; c < m; c++) {
        for (int e = src->row_ptr[c]; e < src->row_ptr[c + 1]; e++)
            if (newid[src->col_idx[e]] >= 0) nnz++;
        dst->row_ptr[c + 1] = nnz;
    }
    dst->nnz = nnz;
    dst->col_idx = xcalloc(nnz, sizeof(int));
    for (int c = 0; c < m; c++) {
        int k = dst->row_ptr[c];
        for (int e = src->row_ptr[c]; e < src->row_ptr[c + 1]; e++) {
            int j = newid[src->col_idx[e]];
            if (j >= 0) dst->col_idx[k++] = j;
        }
        /* keep the row ordered by new column id like mod2sparse_insert does */
        int lo = dst->row_ptr[c], hi = dst->row_ptr[c + 1];
        for (int a = lo + 1; a < hi; a++) {
            int x = dst->col_idx[a], b = a - 1;
            while (b >= lo && dst->col_idx[b] > x) { dst->col_idx[b + 1] = dst->col_idx[b]; b--; }
            dst->col_idx[b + 1] = x;
        }
    }
    free(newid);
    tanner_finish(dst);
}

/* ------------------------------------------------------------------------------------ */
struct swo_graph {
    tanner t;
    double *llr;
    int rank;
};

static int gf2_rank_csr(const tanner *t);

swo_graph *swo_graph_create(int m, int n, const int32_t *row_ptr, const int32_t *col_idx,
                            const double *channel_probs) {
    swo_graph *g = xcalloc(1, sizeof(*g));
    tanner_init_csr(&g->t, m, n, row_ptr, col_idx);
    g->llr = xcalloc(n, sizeof(double));
    for (int v = 0; v < n; v++) /* osd_window.pyx:113 */
        g->llr[v] = log((1 - channel_probs[v]) / channel_probs[v]);
    g->rank = gf2_rank_csr(&g->t);
    return g;
}

void swo_graph_free(swo_graph *g) {
    if (!g) return;
    tanner_free(&g->t);
    free(g->llr);
    free(g);
}

int swo_graph_rank(const swo_graph *g) { return g->rank; }

/* ------------------------------------------------------------------------------------ */
/* masked min-sum: one flooding iteration                                                */
/*   osd_window.pyx:392-471  ==  bp_guessing_decoder.pyx:64-126 (all-live)               */
/*   ==  src/include/bpgd.cpp:103-182                                                     */
/* vn_mask[v] == -1 : live VN.  cn_mask[c] == -1 : cleared CN, else residual check value */
/* ------------------------------------------------------------------------------------ */
static void minsum_iteration(tanner *t, const signed char *vn_mask, const signed char *cn_mask,
                             const double *llr, double alpha, double *hist /* n x 4 */,
                             int slot, signed char *hard) {
    const int m = t->m, n = t->n;
    for (int cn = 0; cn < m; cn++) {
        if (cn_mask[cn] == -1) continue;
        double temp = 1e308;
        int sgn = (cn_mask[cn] == 1) ? 1 : 0;
        for (int e = t->row_ptr[cn]; e < t->row_ptr[cn + 1]; e++) { /* left to right */
            if (vn_mask[t->col_idx[e]] != -1) continue;
            t->c2b[e] = temp;
            t->sgn[e] = sgn;
            if (t->b2c[e] > 50.0) t->b2c[e] = 50.0;
            else if (t->b2c[e] < -50.0) t->b2c[e] = -50.0;
            if (fabs(t->b2c[e]) < temp) temp = fabs(t->b2c[e]);
            if (t->b2c[e] <= 0) sgn = 1 - sgn;
        }
        temp = 1e308;
        sgn = 0;
        for (int e = t->row_ptr[cn + 1] - 1; e >= t->row_ptr[cn]; e--) { /* right to left */
            if (vn_mask[t->col_idx[e]] != -1) continue;
            if (temp < t->c2b[e]) t->c2b[e] = temp;
            t->sgn[e] += sgn;
            t->c2b[e] *= ((t->sgn[e] % 2 == 0) ? 1.0 : -1.0) * alpha;
            if (fabs(t->b2c[e]) < temp) temp = fabs(t->b2c[e]);
            if (t->b2c[e] <= 0) sgn = 1 - sgn;
        }
    }
    for (int vn = 0; vn < n; vn++) {
        if (vn_mask[vn] != -1) continue;
        double temp = llr[vn];
        for (int k = t->col_ptr[vn]; k < t->col_ptr[vn + 1]; k++) { /* top to bottom */
            if (cn_mask[t->row_idx[k]] == -1) continue;
            int e = t->c2r[k];
            t->b2c[e] = temp;
            temp += t->c2b[e];
        }
        hist[(size_t)vn * 4 + slot] = temp;
        hard[vn] = (temp <= 0) ? 1 : 0;
        temp = 0.0;
        for (int k = t->col_ptr[vn + 1] - 1; k >= t->col_ptr[vn]; k--) { /* bottom to top */
            if (cn_mask[t->row_idx[k]] == -1) continue;
            int e = t->c2r[k];
            t->b2c[e] += temp;
            temp += t->c2b[e];
        }
    }
}

/* H * u == s ?  (mod2sparse_mulvec, src/include/mod2sparse.c:655-678 + compare loop) */
static int syndrome_matches(const tanner *t, const signed char *u, const signed char *s,
                            signed char *scratch) {
    for (int c = 0; c < t->m; c++) scratch[c] = 0;
    for (int v = 0; v < t->n; v++)
        if (u[v])
            for (int k = t->col_ptr[v]; k < t->col_ptr[v + 1]; k++) scratch[t->row_idx[k]] ^= 1;
    for (int c = 0; c < t->m; c++)
        if (s[c] != scratch[c]) return 0;
    return 1;
}

/* bp_init: osd_window.pyx:370-379 / BPGD::init bpgd.cpp:82-95 */
static void bp_init(tanner *t, const signed char *vn_mask, const double *llr) {
    for (int vn = 0; vn < t->n; vn++) {
        if (vn_mask[vn] != -1) continue;
        for (int k = t->col_ptr[vn]; k < t->col_ptr[vn + 1]; k++) t->b2c[t->c2r[k]] = llr[vn];
    }
}

/* index_sort: stable ascending argsort (src/include/bpgd.cpp:384-389) -- merge sort */
static void index_sort(const double *v, int *cols, int n) {
    int *a = xcalloc(n, sizeof(int)), *b = xcalloc(n, sizeof(int));
    for (int i = 0; i < n; i++) a[i] = i;
    for (int w = 1; w < n; w *= 2) {
        for (int lo = 0; lo < n; lo += 2 * w) {
            int mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            int i = lo, j = mid, k = lo;
            while (i < mid && j < hi) {
                if (v[a[j]] < v[a[i]]) b[k++] = a[j++]; /* take right only if strictly less */
                else b[k++] = a[i++];
            }
            while (i < mid) b[k++] = a[i++];
            while (j < hi) b[k++] = a[j++];
        }
        int *tmp = a; a = b; b = tmp;
    }
    memcpy(cols, a, n * sizeof(int));
    free(a); free(b);
}

/* ------------------------------------------------------------------------------------ */
/* GF(2) LU with caller-supplied column order                                            */
/*   mod2sparse_decomp_osd   src/include/mod2sparse_extra.cpp:113-376 (strategy "first") */
/*   LU_forward_backward_solve   mod2sparse_extra.cpp:78-106 -> mod2sparse.c:1099-1215   */
/* Dense bit rows replace the sparse B/L/U; the pivot rule is the reference's:           */
/* step i takes the first k>=i whose column cols[k] has a 1 in a not-yet-pivoted row and  */
/* pivots on the lowest such row index (column entries are row-ascending), swaps          */
/* cols[i]<->cols[k], and adds the pivot row to the later rows only.                      */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    int m, n, R, W;     /* W = 64-bit words per row */
    uint64_t *B;        /* m x W working copy (rows keep their original index)            */
    int *prow;          /* prow[i]  = original row pivoted at step i  (rows[i])           */
    unsigned char *piv; /* piv[r]   = 1 once row r was used as pivot                       */
    uint64_t *Lcol;     /* R x Wm bit columns of L below the diagonal: rows that got pivot i added */
    int Wm;
    int nnf;
} lu_t;

static inline int bit_get(const uint64_t *row, int j) { return (int)((row[j >> 6] >> (j & 63)) & 1); }
static inline void bit_flip(uint64_t *row, int j) { row[j >> 6] ^= (uint64_t)1 << (j & 63); }

static void lu_init(lu_t *lu, const tanner *t, int R) {
    lu->m = t->m; lu->n = t->n; lu->R = R;
    lu->W = (t->n + 63) / 64; lu->Wm = (t->m + 63) / 64;
    lu->B = xcalloc((size_t)lu->m * lu->W, sizeof(uint64_t));
    lu->prow = xcalloc(R, sizeof(int));
    lu->piv = xcalloc(lu->m, 1);
    lu->Lcol = xcalloc((size_t)R * lu->Wm, sizeof(uint64_t));
    lu->nnf = 0;
}
static void lu_free(lu_t *lu) { free(lu->B); free(lu->prow); free(lu->piv); free(lu->Lcol); }

/* cols[] (n long) in: priority order; out: pivot columns first (cols[0..R)) */
static void lu_decomp_osd(lu_t *lu, const tanner *t, int *cols) {
    const int m = lu->m, n = lu->n, W = lu->W;
    memset(lu->B, 0, (size_t)m * W * sizeof(uint64_t));
    memset(lu->piv, 0, m);
    memset(lu->Lcol, 0, (size_t)lu->R * lu->Wm * sizeof(uint64_t));
    lu->nnf = 0;
    for (int c = 0; c < m; c++)
        for (int e = t->row_ptr[c]; e < t->row_ptr[c + 1]; e++) bit_flip(lu->B + (size_t)c * W, t->col_idx[e]);
    for (int i = 0; i < lu->R; i++) {
        int found = 0, k, prow = -1;
        for (k = i; k < n; k++) {
            int col = cols[k];
            for (int r = 0; r < m; r++)
                if (!lu->piv[r] && bit_get(lu->B + (size_t)r * W, col)) { prow = r; found = 1; break; }
            if (found) break;
        }
        if (!found) { lu->nnf++; lu->prow[i] = -1; continue; }
        int pc = cols[k];
        cols[k] = cols[i];
        cols[i] = pc;
        lu->prow[i] = prow;
        lu->piv[prow] = 1;
        const uint64_t *P = lu->B + (size_t)prow * W;
        for (int r = 0; r < m; r++) {
            if (lu->piv[r]) continue;
            uint64_t *Rr = lu->B + (size_t)r * W;
            if (bit_get(Rr, pc)) {
                for (int w = 0; w < W; w++) Rr[w] ^= P[w];
                bit_flip(lu->Lcol + (size_t)i * lu->Wm, r);
            }
        }
    }
}

/* x (n long) <- solution with zeros outside the pivot columns; z = right-hand side (m) */
static void lu_solve(const lu_t *lu, const int *cols, const signed char *z, signed char *x,
                     signed char *work /* m */, signed char *y /* R */) {
    const int m = lu->m, n = lu->n, R = lu->R, W = lu->W;
    for (int j = 0; j < n; j++) x[j] = 0;
    for (int r = 0; r < m; r++) work[r] = z[r];
    /* forward substitution == replay of the row additions on the right-hand side */
    for (int i = 0; i < R; i++) {
        int pr = lu->prow[i];
        if (pr < 0) { y[i] = 0; continue; }
        y[i] = work[pr];
        if (y[i]) {
            const uint64_t *L = lu->Lcol + (size_t)i * lu->Wm;
            for (int r = 0; r < m; r++)
                if (bit_get(L, r)) work[r] ^= 1;
        }
    }
    /* backward substitution on U = pivot rows restricted to pivot columns */
    for (int i = R - 1; i >= 0; i--) {
        int pr = lu->prow[i];
        if (pr < 0) continue;
        const uint64_t *U = lu->B + (size_t)pr * W;
        int b = 0;
        for (int j = i + 1; j < R; j++)
            if (bit_get(U, cols[j])) b ^= x[cols[j]];
        x[cols[i]] = (signed char)(b ^ y[i]);
    }
}

static int gf2_rank_csr(const tanner *t) {
    int R = t->m < t->n ? t->m : t->n;
    lu_t lu;
    lu_init(&lu, t, R);
    int *cols = xcalloc(t->n, sizeof(int));
    for (int j = 0; j < t->n; j++) cols[j] = j;
    lu_decomp_osd(&lu, t, cols);
    int rank = R - lu.nnf;
    free(cols);
    lu_free(&lu);
    return rank;
}

/* ------------------------------------------------------------------------------------ */
/* osd_window                                                                            */
/* ------------------------------------------------------------------------------------ */
struct swo_osdw {
    tanner t; /* private copy: messages live in it */
    const swo_graph *g;
    swo_osdw_params p;
    int m, n, new_n, rank, k;
    signed char *synd, *scratch_m, *bp_decoding, *current_vn, *current_cn;
    signed char *osd0, *osdw, *y, *gvec, *Htx, *work_m, *yR;
    int *cn_degree, *cur_deg;
    double *hist, *llr_sum;
    int *cols, *orig_cols, *Ht_cols;
    long enc_count;
    signed char **enc; /* osdw_encoding_inputs */
    lu_t lu;
    int bp_iteration, converge;
    double min_pm;
};

static signed char *dec2bin_rev(long v, int k) { /* mod2sparse_extra.cpp:8-21 */
    signed char *b = xcalloc(k, 1);
    for (int i = 0; i < k; i++) { b[i] = (signed char)(v % 2); v /= 2; if (v == 0) break; }
    return b;
}

swo_osdw *swo_osdw_create(const swo_graph *g, const swo_osdw_params *p) {
    swo_osdw *d = xcalloc(1, sizeof(*d));
    d->g = g; d->p = *p;
    int m = g->t.m, n = g->t.n;
    d->m = m; d->n = n;
    tanner_init_csr(&d->t, m, n, g->t.row_ptr, g->t.col_idx);
    d->new_n = (p->new_n <= 0) ? (n < 2 * m ? n : 2 * m) : (p->new_n < n ? p->new_n : n); /* :60-63 */
    d->rank = g->rank;
    if (p->osd_method == 0) d->p.osd_order = 0; /* :69-71 */
    if (d->p.osd_order > d->new_n - d->rank) { swo_osdw_free(d); return NULL; } /* :88-92 */
    d->k = d->new_n - d->rank;
    d->synd = xcalloc(m, 1); d->scratch_m = xcalloc(m, 1); d->work_m = xcalloc(m, 1);
    d->bp_decoding = xcalloc(n, 1); d->current_vn = xcalloc(n, 1); d->current_cn = xcalloc(m, 1);
    d->osd0 = xcalloc(n, 1); d->osdw = xcalloc(n, 1); d->y = xcalloc(n, 1);
    d->gvec = xcalloc(m, 1); d->Htx = xcalloc(m, 1); d->yR = xcalloc(d->rank + 1, 1);
    d->cn_degree = xcalloc(m, sizeof(int)); d->cur_deg = xcalloc(m, sizeof(int));
    d->hist = xcalloc((size_t)n * 4, sizeof(double)); d->llr_sum = xcalloc(n, sizeof(double));
    d->cols = xcalloc(n, sizeof(int)); d->orig_cols = xcalloc(n, sizeof(int));
    d->Ht_cols = xcalloc(d->k + 1, sizeof(int));
    for (int c = 0; c < m; c++) d->cn_degree[c] = g->t.row_ptr[c + 1] - g->t.row_ptr[c]; /* :115-123 */
    if (d->p.osd_order > -1) lu_init(&d->lu, &d->t, d->rank);
    if (d->p.osd_order > 0 && d->p.osd_method == 1) { /* osd_e_setup :128-132 */
        d->enc_count = 1L << d->p.osd_order;
        d->enc = xcalloc(d->enc_count, sizeof(*d->enc));
        for (long i = 0; i < d->enc_count; i++) d->enc[i] = dec2bin_rev(i, d->k);
    } else if (d->p.osd_order > 0 && d->p.osd_method == 2) { /* osd_cs_setup :134-155 */
        int w = d->p.osd_order;
        d->enc_count = d->k + (long)w * (w - 1) / 2;
        d->enc = xcalloc(d->enc_count, sizeof(*d->enc));
        long c = 0;
        for (int i = 0; i < d->k; i++) { d->enc[c] = xcalloc(d->k, 1); d->enc[c][i] = 1; c++; }
        for (int i = 0; i < w; i++)
            for (int j = 0; j < w; j++)
                if (i < j) { d->enc[c] = xcalloc(d->k, 1); d->enc[c][i] = 1; d->enc[c][j] = 1; c++; }
    }
    return d;
}

void swo_osdw_free(swo_osdw *d) {
    if (!d) return;
    tanner_free(&d->t);
    free(d->synd); free(d->scratch_m); free(d->work_m); free(d->bp_decoding); free(d->current_vn);
    free(d->current_cn); free(d->osd0); free(d->osdw); free(d->y); free(d->gvec); free(d->Htx);
    free(d->yR); free(d->cn_degree); free(d->cur_deg); free(d->hist); free(d->llr_sum);
    free(d->cols); free(d->orig_cols); free(d->Ht_cols);
    if (d->enc) { for (long i = 0; i < d->enc_count; i++) free(d->enc[i]); free(d->enc); }
    if (d->lu.B) lu_free(&d->lu);
    free(d);
}

void swo_osdw_clear_history(swo_osdw *d) { memset(d->hist, 0, (size_t)d->n * 4 * sizeof(double)); }
const double *swo_osdw_history(const swo_osdw *d) { return d->hist; }
const uint8_t *swo_osdw_osd0(const swo_osdw *d) { return (const uint8_t *)d->osd0; }
const uint8_t *swo_osdw_bp(const swo_osdw *d) { return (const uint8_t *)d->bp_decoding; }

/* osd_window.pyx:340-368 */
static int osdw_vn_set_value(swo_osdw *d, int vn, int value) {
    if (d->current_vn[vn] != -1) return (d->current_vn[vn] == value) ? 0 : -1;
    d->current_vn[vn] = (signed char)value;
    d->bp_decoding[vn] = (signed char)value;
    const tanner *t = &d->t;
    for (int k = t->col_ptr[vn]; k < t->col_ptr[vn + 1]; k++) {
        int cn = t->row_idx[k];
        if (d->current_cn[cn] == -1) continue;
        int deg = d->cur_deg[cn] - 1;
        if (value) d->current_cn[cn] = (signed char)(1 - d->current_cn[cn]);
        if (deg == 0) {
            if (d->current_cn[cn] != 0) return -1;
            d->current_cn[cn] = -1;
        }
        d->cur_deg[cn] = deg;
    }
    return 0;
}

/* osd_window.pyx:306-338 */
static int osdw_peel(swo_osdw *d) {
    const tanner *t = &d->t;
    for (;;) {
        int degree_check = 1;
        for (int cn = 0; cn < d->m; cn++) {
            if (d->current_cn[cn] == -1) continue;
            if (d->cur_deg[cn] >= 2) continue;
            degree_check = 0;
            int vn = -1;
            for (int e = t->row_ptr[cn]; e < t->row_ptr[cn + 1]; e++) {
                if (d->current_vn[t->col_idx[e]] != -1) continue;
                vn = t->col_idx[e];
                break;
            }
            if (vn < 0) return -1; /* reference would index out of bounds; cannot happen with consistent degrees */
            if (osdw_vn_set_value(d, vn, d->current_cn[cn]) == -1) return -1;
        }
        if (degree_check) return 0;
    }
}

/* osd_window.pyx:381-485 */
static int osdw_bp(swo_osdw *d, int max_iter) {
    d->converge = 0;
    for (int it = 0; it < max_iter; it++) {
        d->bp_iteration += 1;
        minsum_iteration(&d->t, d->current_vn, d->current_cn, d->g->llr, d->p.ms_scaling_factor,
                         d->hist, it % 4, d->bp_decoding);
        if (syndrome_matches(&d->t, d->bp_decoding, d->synd, d->scratch_m)) { d->converge = 1; return 1; }
    }
    return 0;
}

/* osd_window.pyx:201-284 */
static void osdw_osd(swo_osdw *d) {
    const int n = d->n, m = d->m;
    const double *llr = d->g->llr;
    for (int vn = 0; vn < n; vn++) {
        if (d->current_vn[vn] == 1) d->llr_sum[vn] = -1000;
        else if (d->current_vn[vn] == 0) d->llr_sum[vn] = 1000;
        else { const double *h = d->hist + (size_t)vn * 4; d->llr_sum[vn] = h[0] + h[1] + h[2] + h[3]; }
    }
    index_sort(d->llr_sum, d->cols, n);
    for (int vn = 0; vn < n; vn++) d->orig_cols[vn] = d->cols[vn];
    lu_decomp_osd(&d->lu, &d->t, d->cols);
    lu_solve(&d->lu, d->cols, d->synd, d->osd0, d->work_m, d->yR);
    d->min_pm = 0.0;
    for (int vn = 0; vn < n; vn++) {
        if (d->osd0[vn]) d->min_pm += llr[vn];
        d->osdw[vn] = d->osd0[vn];
    }
    if (d->p.osd_order == 0) return;
    /* non-pivot columns among the first new_n of the sorted order (:243-256) */
    int counter = 0;
    for (int i = 0; i < d->new_n; i++) {
        int cn = d->orig_cols[i], in_pivot = 0;
        for (int j = 0; j < d->rank; j++)
            if (d->cols[j] == cn) { in_pivot = 1; break; }
        if (!in_pivot) { if (counter < d->k) d->Ht_cols[counter] = cn; counter++; }
    }
    const tanner *t = &d->t;
    for (long l = 0; l < d->enc_count; l++) {
        const signed char *x = d->enc[l];
        for (int c = 0; c < m; c++) d->Htx[c] = 0;
        for (int j = 0; j < d->k; j++)
            if (x[j]) {
                int v = d->Ht_cols[j];
                for (int k = t->col_ptr[v]; k < t->col_ptr[v + 1]; k++) d->Htx[t->row_idx[k]] ^= 1;
            }
        for (int c = 0; c < m; c++) d->gvec[c] = (signed char)(d->synd[c] ^ d->Htx[c]);
        lu_solve(&d->lu, d->cols, d->gvec, d->y, d->work_m, d->yR);
        for (int j = 0; j < d->k; j++) d->y[d->Ht_cols[j]] = x[j];
        double pm = 0.0;
        for (int vn = 0; vn < n; vn++)
            if (d->y[vn]) pm += llr[vn];
        if (pm < d->min_pm) {
            d->min_pm = pm;
            for (int vn = 0; vn < n; vn++) d->osdw[vn] = d->y[vn];
        }
    }
}

/* osd_window.pyx:158-199 (+ reset :288-303) */
int swo_osdw_decode(swo_osdw *d, const uint8_t *synd, uint8_t *out, swo_result *res) {
    const int n = d->n, m = d->m;
    const double *llr = d->g->llr;
    for (int c = 0; c < m; c++) d->synd[c] = (signed char)synd[c];
    d->bp_iteration = 0; d->min_pm = 0.0;
    for (int c = 0; c < m; c++) d->cur_deg[c] = d->cn_degree[c];
    for (int c = 0; c < m; c++) d->current_cn[c] = d->synd[c];
    for (int v = 0; v < n; v++) d->current_vn[v] = -1;
    for (int v = 0; v < n; v++) d->bp_decoding[v] = 0;
    int exit_class;
    const signed char *ret = d->bp_decoding;
    bp_init(&d->t, d->current_vn, llr);
    if (osdw_bp(d, d->p.pre_max_iter)) {
        d->converge = 1;
        for (int v = 0; v < n; v++) if (d->bp_decoding[v]) d->min_pm += llr[v];
        exit_class = SWO_EXIT_PRE;
    } else {
        for (int v = 0; v < n; v++) { const double *h = d->hist + (size_t)v * 4; d->llr_sum[v] = h[0] + h[1] + h[2] + h[3]; }
        index_sort(d->llr_sum, d->cols, n);
        exit_class = -1;
        for (int i = d->new_n; i < n; i++)
            if (osdw_vn_set_value(d, d->cols[i], 0) == -1) { exit_class = SWO_EXIT_FAIL_SET; break; }
        if (exit_class < 0) {
            for (int i = d->new_n; i < n; i++) d->bp_decoding[d->cols[i]] = 0;
            if (osdw_peel(d) == -1) exit_class = SWO_EXIT_FAIL_PEEL;
        }
        if (exit_class < 0) {
            bp_init(&d->t, d->current_vn, llr);
            if (osdw_bp(d, d->p.post_max_iter)) {
                d->converge = 1;
                for (int v = 0; v < n; v++) if (d->bp_decoding[v]) d->min_pm += llr[v];
                exit_class = SWO_EXIT_POST;
            } else if (d->p.osd_order > -1) {
                osdw_osd(d);
                ret = d->osdw;
                exit_class = SWO_EXIT_OSD;
            } else exit_class = SWO_EXIT_NO_OSD;
        }
    }
    for (int v = 0; v < n; v++) out[v] = (uint8_t)ret[v];
    if (res) {
        res->converge = d->converge; res->bp_iteration = d->bp_iteration;
        res->exit_class = exit_class; res->reserved = 0; res->min_pm = d->min_pm;
    }
    return 0;
}

int swo_osdw_decode_batch(swo_osdw *d, int B, const uint8_t *synd, uint8_t *out, swo_result *res) {
    for (int b = 0; b < B; b++) {
        swo_osdw_clear_history(d);
        swo_osdw_decode(d, synd + (size_t)b * d->m, out + (size_t)b * d->n, res ? res + b : NULL);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* bp_history_decoder / bpgdg_decoder / bpgd_decoder                                     */
/* ------------------------------------------------------------------------------------ */
typedef struct { /* BPGD, src/include/bpgd.hpp:12-48 */
    int m, n, num_iter, low_error_mode;
    double factor;
    tanner pcm; int have_pcm;
    double *llr_prior, *post; /* post: n x 4 */
    signed char *vn_mask, *cn_mask, *error, *syndrome, *temp_syndrome;
    int *vn_degree, *cn_degree;
} bpgd_t;

struct swo_gdg {
    tanner t;
    const swo_graph *g;
    swo_gdg_params p;
    int m, n, new_n, max_guess, used_guess, min_converge_depth;
    signed char *synd, *scratch_m, *bp_decoding, *all_live_vn, *bpgd_error;
    double *hist, *llr_sum;
    int *cols;
    bpgd_t b;
    signed char **vn_stack, **cn_stack;
    int **cn_degree_stack;
    signed char *decision_value_stack;
    int *decision_vn_stack, *alt_depth_stack;
    double min_pm;
    int converge, bp_iteration;
    /* threaded ensemble (mode 3): path metric of every hypothesis of the last decode, 10000.0 = did not converge:
     * [0] main thread, [1 .. T] tree threads by id, [T+1 .. T+S] side threads by index */
    double *ens_pm;
    int ens_count, ens_winner, ens_ties;
    int ens_blocks, ens_blocks_prefix; double ens_blocks_unique; /* statistics: BP blocks of the last ensemble, those at depth < D of the main / tree threads, and the latter counted once per distinct prefix */
};

static void bpgd_alloc(bpgd_t *b, int m, int n, int num_iter, int low_error_mode, double factor) {
    memset(b, 0, sizeof(*b));
    b->m = m; b->n = n; b->num_iter = num_iter; b->low_error_mode = low_error_mode; b->factor = factor;
    b->llr_prior = xcalloc(n, sizeof(double)); b->post = xcalloc((size_t)n * 4, sizeof(double));
    b->vn_mask = xcalloc(n, 1); b->cn_mask = xcalloc(m, 1); b->error = xcalloc(n, 1);
    b->syndrome = xcalloc(m, 1); b->temp_syndrome = xcalloc(m, 1);
    b->vn_degree = xcalloc(n, sizeof(int)); b->cn_degree = xcalloc(m, sizeof(int));
}
static void bpgd_release(bpgd_t *b) {
    if (b->have_pcm) tanner_free(&b->pcm);
    free(b->llr_prior); free(b->post); free(b->vn_mask); free(b->cn_mask); free(b->error);
    free(b->syndrome); free(b->temp_syndrome); free(b->vn_degree); free(b->cn_degree);
}

/* BPGD::vn_set_value bpgd.cpp:51-80 */
static int bpgd_vn_set_value(bpgd_t *b, int vn, int value) {
    if (b->vn_mask[vn] != -1) return (b->vn_mask[vn] == value) ? 0 : -1;
    b->vn_mask[vn] = (signed char)value;
    b->error[vn] = (signed char)value;
    const tanner *t = &b->pcm;
    for (int k = t->col_ptr[vn]; k < t->col_ptr[vn + 1]; k++) {
        int cn = t->row_idx[k];
        if (b->cn_mask[cn] == -1 || b->cn_degree[cn] == 0) return -1;
        int deg = b->cn_degree[cn] - 1;
        if (value) b->cn_mask[cn] = (signed char)(1 - b->cn_mask[cn]);
        b->cn_degree[cn] = deg;
        if (deg == 0) {
            if (b->cn_mask[cn] != 0) return -1;
            b->cn_mask[cn] = -1;
        }
    }
    return 0;
}

/* BPGD::peel bpgd.cpp:13-49 */
static int bpgd_peel(bpgd_t *b) {
    const tanner *t = &b->pcm;
    for (;;) {
        int degree_check = 1;
        for (int cn = 0; cn < b->m; cn++) {
            if (b->cn_mask[cn] == -1) continue;
            if (b->cn_degree[cn] >= 2) continue;
            if (b->cn_degree[cn] <= 0) { b->cn_mask[cn] = -1; continue; }
            degree_check = 0;
            int vn = -1;
            for (int e = t->row_ptr[cn]; e < t->row_ptr[cn + 1]; e++) {
                if (b->vn_mask[t->col_idx[e]] != -1) continue;
                vn = t->col_idx[e];
                break;
            }
            if (vn == -1) return -1;
            if (bpgd_vn_set_value(b, vn, b->cn_mask[cn]) == -1) return -1;
        }
        if (degree_check) return 0;
    }
}

/* BPGD::min_sum_log bpgd.cpp:97-197 */
static int bpgd_min_sum_log(bpgd_t *b) {
    for (int it = 0; it < b->num_iter; it++) {
        minsum_iteration(&b->pcm, b->vn_mask, b->cn_mask, b->llr_prior, b->factor, b->post, it % 4, b->error);
        if (syndrome_matches(&b->pcm, b->error, b->syndrome, b->temp_syndrome)) return 1;
    }
    return 0;
}

/* BPGD::reset bpgd.cpp:199-239 */
static int bpgd_reset(bpgd_t *b, const tanner *src, const int *cols, const double *src_llr,
                      const signed char *src_synd) {
    if (b->have_pcm) tanner_free(&b->pcm);
    tanner_copycols(src, cols, b->n, &b->pcm);
    b->have_pcm = 1;
    const tanner *t = &b->pcm;
    for (int v = 0; v < b->n; v++) b->llr_prior[v] = src_llr[cols[v]];
    for (int v = 0; v < b->n; v++) b->vn_mask[v] = -1;
    for (int c = 0; c < b->m; c++) b->cn_mask[c] = src_synd[c];
    for (int c = 0; c < b->m; c++) {
        int deg = t->row_ptr[c + 1] - t->row_ptr[c];
        b->cn_degree[c] = deg;
        if (deg == 0) b->cn_mask[c] = -1;
    }
    for (int v = 0; v < b->n; v++) b->vn_degree[v] = t->col_ptr[v + 1] - t->col_ptr[v];
    for (int v = 0; v < b->n; v++) b->error[v] = 0;
    for (int c = 0; c < b->m; c++) b->syndrome[c] = src_synd[c];
    /* temp_syndrome is not reset by the reference either */
    if (bpgd_peel(b) == -1) return -1;
    bp_init(&b->pcm, b->vn_mask, b->llr_prior);
    return 0;
}

/* BPGD::set_masks bpgd.cpp:241-248 */
static void bpgd_set_masks(bpgd_t *b, const signed char *vn, const signed char *cn, const int *deg) {
    for (int v = 0; v < b->n; v++) b->vn_mask[v] = vn[v];
    for (int v = 0; v < b->n; v++) b->error[v] = b->vn_mask[v];
    for (int c = 0; c < b->m; c++) b->cn_mask[c] = cn[c];
    for (int c = 0; c < b->m; c++) b->cn_degree[c] = deg[c];
    bp_init(&b->pcm, b->vn_mask, b->llr_prior);
}

static double bpgd_get_pm(const bpgd_t *b) { /* bpgd.cpp:250-256 */
    double pm = 0;
    for (int v = 0; v < b->n; v++) if (b->error[v]) pm += b->llr_prior[v];
    return pm;
}

/* BPGD::decimate_vn_reliable bpgd.cpp:258-286 */
static int bpgd_decimate_vn_reliable(bpgd_t *b) {
    int best = -1, best_sign = 0;
    double largest = 0.0;
    for (int v = 0; v < b->n; v++) {
        if (b->vn_mask[v] != -1) continue;
        double hs = b->post[(size_t)v * 4 + 3];
        if (fabs(hs) > largest) { largest = fabs(hs); best = v; best_sign = (hs > 0) ? 0 : 1; }
    }
    if (best < 0) return -1; /* reference indexes vn_mask[-1]; treat as failure */
    if (bpgd_vn_set_value(b, best, best_sign) == -1) return -1;
    if (bpgd_peel(b) == -1) return -1;
    return 0;
}

swo_gdg *swo_gdg_create(const swo_graph *g, const swo_gdg_params *p) {
    swo_gdg *d = xcalloc(1, sizeof(*d));
    d->g = g; d->p = *p;
    int m = g->t.m, n = g->t.n;
    d->m = m; d->n = n;
    tanner_init_csr(&d->t, m, n, g->t.row_ptr, g->t.col_idx);
    d->new_n = (p->new_n <= 0) ? (n < 2 * m ? n : 2 * m) : (p->new_n < n ? p->new_n : n);
    d->max_guess = ((1 << p->max_tree_depth) - 1) * 2 + p->max_side_depth - p->max_tree_depth; /* :181 */
    if (d->max_guess < 1) d->max_guess = 1;
    d->synd = xcalloc(m, 1); d->scratch_m = xcalloc(m, 1); d->bp_decoding = xcalloc(n, 1);
    d->all_live_vn = xcalloc(n, 1); memset(d->all_live_vn, -1, n);
    d->bpgd_error = xcalloc(d->new_n, 1);
    d->hist = xcalloc((size_t)n * 4, sizeof(double)); d->llr_sum = xcalloc(n, sizeof(double));
    d->cols = xcalloc(n, sizeof(int));
    bpgd_alloc(&d->b, m, d->new_n, p->max_iter_per_step, p->low_error_mode, p->gdg_factor);
    d->vn_stack = xcalloc(d->max_guess, sizeof(*d->vn_stack));
    d->cn_stack = xcalloc(d->max_guess, sizeof(*d->cn_stack));
    d->cn_degree_stack = xcalloc(d->max_guess, sizeof(*d->cn_degree_stack));
    for (int i = 0; i < d->max_guess; i++) {
        d->vn_stack[i] = xcalloc(d->new_n, 1); d->cn_stack[i] = xcalloc(m, 1);
        d->cn_degree_stack[i] = xcalloc(m, sizeof(int));
    }
    d->decision_value_stack = xcalloc(d->max_guess, 1);
    d->decision_vn_stack = xcalloc(d->max_guess, sizeof(int));
    d->alt_depth_stack = xcalloc(d->max_guess, sizeof(int));
    d->min_pm = 100000.0; d->min_converge_depth = 100;
    return d;
}

void swo_gdg_free(swo_gdg *d) {
    if (!d) return;
    tanner_free(&d->t);
    free(d->synd); free(d->scratch_m); free(d->bp_decoding); free(d->all_live_vn); free(d->bpgd_error);
    free(d->hist); free(d->llr_sum); free(d->cols);
    bpgd_release(&d->b);
    for (int i = 0; i < d->max_guess; i++) { free(d->vn_stack[i]); free(d->cn_stack[i]); free(d->cn_degree_stack[i]); }
    free(d->vn_stack); free(d->cn_stack); free(d->cn_degree_stack);
    free(d->decision_value_stack); free(d->decision_vn_stack); free(d->alt_depth_stack);
    free(d->ens_pm);
    free(d);
}
void swo_gdg_clear_history(swo_gdg *d) { memset(d->hist, 0, (size_t)d->n * 4 * sizeof(double)); }
const double *swo_gdg_history(const swo_gdg *d) { return d->hist; }

/* bp_history_decoder.bp_decode_llr bp_guessing_decoder.pyx:48-139 */
static int gdg_bp(swo_gdg *d) {
    bp_init(&d->t, d->all_live_vn, d->g->llr);
    for (int it = 0; it < d->p.max_iter; it++) {
        d->bp_iteration += 1;
        /* unmasked: CN sign seed is synd (:71), every node live */
        minsum_iteration(&d->t, d->all_live_vn, d->synd, d->g->llr, d->p.ms_scaling_factor, d->hist, it % 4, d->bp_decoding);
        if (syndrome_matches(&d->t, d->bp_decoding, d->synd, d->scratch_m)) return 1;
    }
    return 0;
}

/* bpgdg_decoder.select_vn bp_guessing_decoder.pyx:340-442 */
static int gdg_select_vn(swo_gdg *d, int side_branch, int current_depth) {
    bpgd_t *b = &d->b;
    const tanner *t = &b->pcm;
    double A = side_branch ? 0.0 : -3.0;
    double A_sum = side_branch ? -10.0 : -12.0;
    if (current_depth == 0) A_sum = -16.0;
    const double C = 30.0, D = 3.0;
    int sum_smallest_vn = -1, sum_smallest_all_neg_vn = -1, guess_vn;
    double sum_smallest = 10000, sum_smallest_all_neg = 10000;
    int favor, unfavor, guess = 1;
    for (int vn = 0; vn < d->new_n; vn++) {
        if (b->vn_mask[vn] != -1) continue;
        if (b->vn_degree[vn] <= 2) continue;
        int num_flip = 0;
        for (int k = t->col_ptr[vn]; k < t->col_ptr[vn + 1]; k++) {
            int cn = t->row_idx[k];
            if (b->cn_mask[cn] == -1) continue;
            if (b->syndrome[cn] != b->temp_syndrome[cn]) num_flip++;
        }
        const double *h = b->post + (size_t)vn * 4;
        int all_smaller_than_A = 1, all_negative = 1, all_larger_than_C = 1, all_larger_than_D = 1;
        double history_sum = 0.0;
        for (int i = 0; i < 4; i++) {
            double llr = h[i];
            history_sum += llr;
            if (llr < C) all_larger_than_C = 0;
            if (llr < D) all_larger_than_D = 0;
            if (llr > A) all_smaller_than_A = 0;
            if (llr > 0.0) all_negative = 0;
        }
        if (!d->p.low_error_mode && all_larger_than_C && current_depth < 4) {
            if (bpgd_vn_set_value(b, vn, 0) == -1) return -1;
        } else if (!d->p.low_error_mode && num_flip >= 3 && all_larger_than_D) {
            if (bpgd_vn_set_value(b, vn, 0) == -1) return -1;
        } else if (!d->p.low_error_mode && (all_smaller_than_A && history_sum < A_sum)) {
            if (bpgd_vn_set_value(b, vn, 1) == -1) return -1;
        } else {
            if (history_sum < sum_smallest) { sum_smallest = history_sum; sum_smallest_vn = vn; }
            if (all_negative && history_sum < sum_smallest_all_neg) { sum_smallest_all_neg = history_sum; sum_smallest_all_neg_vn = vn; }
        }
    }
    if (bpgd_peel(b) == -1) return -1;
    if (sum_smallest_all_neg_vn != -1) { guess_vn = sum_smallest_all_neg_vn; favor = 1; }
    else { guess_vn = sum_smallest_vn; favor = (sum_smallest > 0) ? 0 : 1; }
    unfavor = 1 - favor;
    if (current_depth > d->min_converge_depth) guess = 0;
    if (!side_branch && current_depth >= d->p.max_side_depth) guess = 0;
    if (side_branch && current_depth > d->p.max_tree_depth) guess = 0;
    if (guess && d->used_guess < d->max_guess) {
        int u = d->used_guess;
        d->decision_value_stack[u] = (signed char)unfavor;
        d->decision_vn_stack[u] = guess_vn;
        d->alt_depth_stack[u] = current_depth + 1;
        memcpy(d->vn_stack[u], b->vn_mask, d->new_n);
        memcpy(d->cn_stack[u], b->cn_mask, d->m);
        memcpy(d->cn_degree_stack[u], b->cn_degree, d->m * sizeof(int));
        d->used_guess = u + 1;
    }
    if (guess_vn < 0) return -1; /* reference would index vn_mask[-1]; no live candidate left */
    if (bpgd_vn_set_value(b, guess_vn, favor) == -1) return -1;
    if (bpgd_peel(b) == -1) return -1;
    return 0;
}

/* bpgdg_decoder.gdg bp_guessing_decoder.pyx:254-338 */
static void gdg_run(swo_gdg *d) {
    bpgd_t *b = &d->b;
    const int n = d->n, new_n = d->new_n;
    d->converge = 0;
    for (int v = 0; v < n; v++) { const double *h = d->hist + (size_t)v * 4; d->llr_sum[v] = h[0] + h[1] + h[2] + h[3]; }
    index_sort(d->llr_sum, d->cols, n);
    for (int v = new_n; v < n; v++) d->bp_decoding[d->cols[v]] = 0;
    if (bpgd_reset(b, &d->t, d->cols, d->g->llr, d->synd) == -1) return;
    d->min_pm = 10000.0; d->used_guess = 0; d->bp_iteration = 0; d->min_converge_depth = d->p.max_step;
    for (int depth = 0; depth < d->p.max_step; depth++) {
        if (bpgd_min_sum_log(b)) {
            d->converge = 1; d->min_converge_depth = depth;
            double pm = bpgd_get_pm(b);
            memcpy(d->bpgd_error, b->error, new_n);
            d->min_pm = pm;
            break;
        }
        if (gdg_select_vn(d, 0, depth) == -1) break;
    }
    if (!d->converge) memcpy(d->bpgd_error, b->error, new_n);
    for (int i = 0; i < d->used_guess; i++) {
        int depth = d->alt_depth_stack[i];
        if (depth > d->min_converge_depth) continue;
        bpgd_set_masks(b, d->vn_stack[i], d->cn_stack[i], d->cn_degree_stack[i]);
        if (bpgd_vn_set_value(b, d->decision_vn_stack[i], d->decision_value_stack[i]) == -1) continue;
        if (bpgd_peel(b) == -1) continue;
        for (int j = 0; j < d->p.max_side_branch_step; j++) {
            depth = d->alt_depth_stack[i] + j;
            if (bpgd_min_sum_log(b)) {
                d->converge = 1;
                double pm = bpgd_get_pm(b);
                if (pm < d->min_pm) {
                    if (depth < d->min_converge_depth) d->min_converge_depth = depth;
                    memcpy(d->bpgd_error, b->error, new_n);
                    d->min_pm = pm;
                }
                break;
            }
            if (depth > d->min_converge_depth + 2) break;
            if (gdg_select_vn(d, 1, depth) == -1) break;
        }
    }
    for (int v = 0; v < new_n; v++) d->bp_decoding[d->cols[v]] = d->bpgd_error[v];
}

/* bpgd_decoder.gd bp_guessing_decoder.pyx:517-560 */
static void gd_run(swo_gdg *d) {
    bpgd_t *b = &d->b;
    const int n = d->n, new_n = d->new_n;
    d->converge = 0;
    for (int v = 0; v < n; v++) { const double *h = d->hist + (size_t)v * 4; d->llr_sum[v] = h[0] + h[1] + h[2] + h[3]; }
    index_sort(d->llr_sum, d->cols, n);
    for (int v = new_n; v < n; v++) d->bp_decoding[d->cols[v]] = 0;
    if (bpgd_reset(b, &d->t, d->cols, d->g->llr, d->synd) == -1) return;
    d->min_pm = 10000.0; d->bp_iteration = 0; d->min_converge_depth = d->p.max_step;
    for (int depth = 0; depth < d->p.max_step; depth++) {
        if (bpgd_min_sum_log(b)) {
            d->converge = 1; d->min_converge_depth = depth;
            double pm = bpgd_get_pm(b);
            memcpy(d->bpgd_error, b->error, new_n);
            d->min_pm = pm;
            break;
        }
        if (bpgd_decimate_vn_reliable(b) == -1) break;
    }
    if (!d->converge) memcpy(d->bpgd_error, b->error, new_n);
    for (int v = 0; v < new_n; v++) d->bp_decoding[d->cols[v]] = d->bpgd_error[v];
}

/* BPGD::select_vn bpgd.cpp:288-355 -- the C++ selection routine the threaded ensemble uses (the single-thread gdg() has its own
 * in Cython, restated above as gdg_select_vn).  Thresholds are the class's int members (bpgd.hpp:15): A, A_sum, C = 30, D = 3.
 * Returns the favoured value and *guess_vn (may be -1), or -1 when an aggressive decimation or the peeling after it fails. */
static int bpgd_select_vn(bpgd_t *b, int A, int A_sum, int current_depth, int *guess_vn) {
    const tanner *t = &b->pcm;
    const int C = 30, D = 3;
    int sum_smallest_vn = -1, sum_smallest_all_neg_vn = -1;
    double sum_smallest = 10000.0, sum_smallest_all_neg = 10000.0;
    for (int vn = 0; vn < b->n; vn++) {
        if (b->vn_mask[vn] != -1) continue;
        if (b->vn_degree[vn] <= 2) continue;
        int num_flip = 0;
        for (int k = t->col_ptr[vn]; k < t->col_ptr[vn + 1]; k++) {
            int cn = t->row_idx[k];
            if (b->cn_mask[cn] == -1) continue;
            if (b->syndrome[cn] != b->temp_syndrome[cn]) num_flip++;
        }
        const double *h = b->post + (size_t)vn * 4;
        int all_smaller_than_A = 1, all_negative = 1, all_larger_than_C = 1, all_larger_than_D = 1;
        double history_sum = 0.0;
        for (int i = 0; i < 4; i++) {
            double llr = h[i];
            history_sum += llr;
            if (llr < C) all_larger_than_C = 0;
            if (llr < D) all_larger_than_D = 0;
            if (llr > A) all_smaller_than_A = 0;
            if (llr > 0) all_negative = 0;
        }
        if (!b->low_error_mode && all_larger_than_C && current_depth < 4) { if (bpgd_vn_set_value(b, vn, 0) == -1) return -1; }
        else if (!b->low_error_mode && num_flip >= 3 && all_larger_than_D) { if (bpgd_vn_set_value(b, vn, 0) == -1) return -1; }
        else if (!b->low_error_mode && all_smaller_than_A && history_sum < A_sum) { if (bpgd_vn_set_value(b, vn, 1) == -1) return -1; }
        else {
            if (history_sum < sum_smallest) { sum_smallest = history_sum; sum_smallest_vn = vn; }
            if (all_negative && history_sum < sum_smallest_all_neg) { sum_smallest_all_neg = history_sum; sum_smallest_all_neg_vn = vn; }
        }
    }
    if (bpgd_peel(b) == -1) return -1;
    if (sum_smallest_all_neg_vn != -1) { *guess_vn = sum_smallest_all_neg_vn; return 1; }
    *guess_vn = sum_smallest_vn;
    return (sum_smallest > 0) ? 0 : 1;
}

/* The threaded ensemble, bpgdg_decoder.gdg_multi_thread (bp_guessing_decoder.pyx:238-251) over BPGD_main_thread::do_work
 * (bpgd.cpp:591-688), BPGD_tree_thread::do_work (:435-525) and BPGD_side_thread::do_work (:527-570), as ONE thread running
 * the bodies in a fixed order: main, tree threads by id, side threads by index.  Every thread's work is a pure function of
 * (matrix, column order, priors, syndrome) and -- side threads -- of the snapshot the main thread hands over; the only thing
 * the reference's threads race for is the strict-< update of the shared best under store_mtx (:463-466, :552-556, :650-654),
 * whose outcome depends on the arrival order only when two converged hypotheses carry exactly the same path metric.  Here
 * ties go to the earliest in the order above; ens_ties counts the converged hypotheses that share the winning metric with a
 * DIFFERENT vector (several threads often reach the same vector), so a caller can tell the shots on which the reference's own
 * answer is timing dependent.  State is that of a newly built
 * object (min_pm_error zero-initialised, bpgd.cpp:583): when BPGD::reset fails the zero vector comes back. */
static void ens_offer(swo_gdg *d, int who, const bpgd_t *b, double pm, double *best, signed char *best_err) {
    d->ens_pm[who] = pm;
    if (pm < *best) { *best = pm; memcpy(best_err, b->error, d->new_n); d->ens_winner = who; d->ens_ties = 0; }
    else if (pm == *best && memcmp(best_err, b->error, d->new_n) != 0) d->ens_ties++; /* same metric, another vector */
}

static void gdg_multi_run(swo_gdg *d) {
    bpgd_t *b = &d->b;
    const int n = d->n, new_n = d->new_n, m = d->m;
    const int Dp = d->p.max_tree_depth, S = d->p.max_side_depth;
    const int T = (1 << Dp) - 1, NS = (S - Dp > 0) ? S - Dp : 0;
    for (int v = 0; v < n; v++) { const double *h = d->hist + (size_t)v * 4; d->llr_sum[v] = h[0] + h[1] + h[2] + h[3]; }
    index_sort(d->llr_sum, d->cols, n);
    d->ens_count = 1 + T + NS; d->ens_winner = -1; d->ens_ties = 0;
    d->ens_blocks = d->ens_blocks_prefix = 0; d->ens_blocks_unique = 0.0;
    free(d->ens_pm); d->ens_pm = xcalloc(d->ens_count, sizeof(double));
    for (int i = 0; i < d->ens_count; i++) d->ens_pm[i] = 10000.0;
    double best = 10000.0;
    signed char *best_err = xcalloc(new_n ? new_n : 1, 1);
    /* snapshots handed to the side threads (:655-667) */
    signed char *side_vn = xcalloc((size_t)(NS ? NS : 1) * new_n, 1), *side_cn = xcalloc((size_t)(NS ? NS : 1) * m, 1);
    int *side_deg = xcalloc((size_t)(NS ? NS : 1) * m, sizeof(int)), *side_status = xcalloc(NS ? NS : 1, sizeof(int));
    int *side_vnidx = xcalloc(NS ? NS : 1, sizeof(int)), *side_val = xcalloc(NS ? NS : 1, sizeof(int)), *side_depth = xcalloc(NS ? NS : 1, sizeof(int));
    signed char *bk_vn = xcalloc(new_n ? new_n : 1, 1), *bk_cn = xcalloc(m, 1);
    int *bk_deg = xcalloc(m, sizeof(int));
    int main_converge = 0;

    /* ---- main thread (:591-688): thresholds (-3, -16 at depth 0 / -12, 30, 3) */
    if (bpgd_reset(b, &d->t, d->cols, d->g->llr, d->synd) != -1) {
        for (int depth = 0; depth < d->p.max_step; depth++) {
            int conv = bpgd_min_sum_log(b);
            d->ens_blocks++; if (depth < Dp) { d->ens_blocks_prefix++; d->ens_blocks_unique += 1.0 / (double)(1 << (Dp - depth)); }
            int guess_vn = -1;
            int favor = bpgd_select_vn(b, -3, depth == 0 ? -16 : -12, depth, &guess_vn); /* BEFORE the convergence test (:630-633) */
            if (conv || favor == -1 || guess_vn == -1) {
                int j = depth - Dp; if (j < 0) j = 0;
                for (; j < NS; j++) side_status[j] = -1;
                if (!conv) break;
                main_converge = 1;
                ens_offer(d, 0, b, bpgd_get_pm(b), &best, best_err);
                break;
            }
            if (depth >= Dp && depth < S) {
                int j = depth - Dp;
                memcpy(side_vn + (size_t)j * new_n, b->vn_mask, new_n);
                memcpy(side_cn + (size_t)j * m, b->cn_mask, m);
                memcpy(side_deg + (size_t)j * m, b->cn_degree, m * sizeof(int));
                side_vnidx[j] = guess_vn; side_val[j] = 1 - favor; side_depth[j] = depth + 1; side_status[j] = 1;
            }
            if (bpgd_vn_set_value(b, guess_vn, favor) != -1 && bpgd_peel(b) != -1) continue;
            int j = depth + 1 - Dp; if (j < 0) j = 0;
            for (; j < NS; j++) side_status[j] = -1;
            break;
        }
        signed char *main_err = xcalloc(new_n ? new_n : 1, 1);
        memcpy(main_err, b->error, new_n);

        /* ---- tree threads id = 1 .. 2^D - 1 (:435-525) */
        for (int id = 1; id <= T; id++) {
            if (bpgd_reset(b, &d->t, d->cols, d->g->llr, d->synd) == -1) continue;
            int on_side = 0, saved = 0, A = -3, A_sum = -16, bk_vnidx = -1, bk_val = 0, done = 0;
            double own_pm = 10000.0;
            for (int depth = 0; depth < d->p.max_tree_branch_step + Dp + 1; depth++) {
                if (depth > 0 && !on_side) A_sum = -12;
                d->ens_blocks++; if (depth < Dp) { d->ens_blocks_prefix++; d->ens_blocks_unique += 1.0 / (double)(1 << (Dp - depth)); }
                if (bpgd_min_sum_log(b)) { own_pm = bpgd_get_pm(b); ens_offer(d, id, b, own_pm, &best, best_err); done = 1; break; }
                int guess_vn = -1;
                int favor = bpgd_select_vn(b, A, A_sum, depth, &guess_vn);
                if (favor == -1 || guess_vn == -1) break;
                if (depth < Dp) {
                    int dir = (id >> (Dp - 1 - depth)) & 1;
                    if (dir) { on_side = 1; A = 0; A_sum = -10; favor = 1 - favor; } /* no re-initialisation of the messages (:494-495) */
                } else if (depth == Dp) {
                    memcpy(bk_vn, b->vn_mask, new_n); memcpy(bk_cn, b->cn_mask, m); memcpy(bk_deg, b->cn_degree, m * sizeof(int));
                    bk_vnidx = guess_vn; bk_val = 1 - favor; saved = 1;
                }
                if (bpgd_vn_set_value(b, guess_vn, favor) == -1) break;
                if (bpgd_peel(b) == -1) break;
            }
            if (done || !saved) continue;
            bpgd_set_masks(b, bk_vn, bk_cn, bk_deg); /* :501-506: masks, error = vn_mask, init() */
            if (bpgd_vn_set_value(b, bk_vnidx, bk_val) == -1) continue;
            if (bpgd_peel(b) == -1) continue;
            int depth = Dp + 1;
            for (int i = 0; i < d->p.max_tree_branch_step; i++) {
                d->ens_blocks++;
                if (bpgd_min_sum_log(b)) {
                    double pm = bpgd_get_pm(b);
                    if (pm > own_pm) break;
                    ens_offer(d, id, b, pm, &best, best_err);
                    break;
                }
                int guess_vn = -1;
                int favor = bpgd_select_vn(b, A, A_sum, depth, &guess_vn);
                if (favor == -1 || guess_vn == -1) break;
                if (bpgd_vn_set_value(b, guess_vn, favor) == -1) break;
                if (bpgd_peel(b) == -1) break;
                depth++;
            }
        }

        /* ---- side threads (:527-570): thresholds (0, -10, 30, 3); their messages come from their own reset(), i.e. the priors */
        for (int j = 0; j < NS; j++) {
            if (side_status[j] != 1) continue;
            if (bpgd_reset(b, &d->t, d->cols, d->g->llr, d->synd) == -1) continue;
            memcpy(b->vn_mask, side_vn + (size_t)j * new_n, new_n);
            memcpy(b->cn_mask, side_cn + (size_t)j * m, m);
            memcpy(b->cn_degree, side_deg + (size_t)j * m, m * sizeof(int));
            for (int v = 0; v < new_n; v++) b->error[v] = b->vn_mask[v];
            if (bpgd_vn_set_value(b, side_vnidx[j], side_val[j]) == -1) continue;
            if (bpgd_peel(b) == -1) continue;
            int depth = side_depth[j];
            for (int i = 0; i < d->p.max_side_branch_step; i++) {
                d->ens_blocks++;
                if (bpgd_min_sum_log(b)) { ens_offer(d, 1 + T + j, b, bpgd_get_pm(b), &best, best_err); break; }
                int guess_vn = -1;
                int favor = bpgd_select_vn(b, 0, -10, depth, &guess_vn);
                if (favor == -1 || guess_vn == -1) break;
                if (bpgd_vn_set_value(b, guess_vn, favor) == -1) break;
                if (bpgd_peel(b) == -1) break;
                depth++;
            }
        }
        if (!main_converge && best > 10000.0 - 1.0) memcpy(best_err, main_err, new_n); /* :677-682 */
        free(main_err);
    }
    d->min_pm = best;
    d->converge = best < 9999.0;
    for (int v = 0; v < new_n; v++) d->bp_decoding[d->cols[v]] = best_err[v];
    for (int v = new_n; v < n; v++) d->bp_decoding[d->cols[v]] = 0;
    free(best_err); free(side_vn); free(side_cn); free(side_deg); free(side_status); free(side_vnidx); free(side_val); free(side_depth);
    free(bk_vn); free(bk_cn); free(bk_deg);
}

/* per-hypothesis path metrics of the last ensemble decode: returns their number, *winner, *ties */
int swo_gdg_ensemble_info(const swo_gdg *d, double *pm, int cap, int32_t *winner, int32_t *ties) {
    for (int i = 0; i < d->ens_count && i < cap; i++) pm[i] = d->ens_pm[i];
    if (winner) *winner = d->ens_winner;
    if (ties) *ties = d->ens_ties;
    return d->ens_count;
}
/* BP blocks (min_sum_log calls) of the last ensemble decode: all, those of the main / tree threads at depth < max_tree_depth, and the
 * latter with a block shared by every thread of one direction prefix counted once (what a walk of the prefix tree would run) */
void swo_gdg_ensemble_blocks(const swo_gdg *d, int32_t *total, int32_t *prefix, double *prefix_unique) {
    *total = d->ens_blocks; *prefix = d->ens_blocks_prefix; *prefix_unique = d->ens_blocks_unique;
}
const int *swo_gdg_cols(const swo_gdg *d) { return d->cols; }

int swo_gdg_decode(swo_gdg *d, int mode, const uint8_t *synd, uint8_t *out, swo_result *res) {
    for (int c = 0; c < d->m; c++) d->synd[c] = (signed char)synd[c];
    int exit_class = 0;
    if (gdg_bp(d)) { d->converge = 1; exit_class = SWO_EXIT_PRE; }
    else if (mode == 0) { gdg_run(d); exit_class = SWO_EXIT_POST; }
    else if (mode == 1) { gd_run(d); exit_class = SWO_EXIT_POST; }
    else if (mode == 3) { gdg_multi_run(d); exit_class = SWO_EXIT_POST; }
    else { d->converge = 0; exit_class = SWO_EXIT_NO_OSD; }
    for (int v = 0; v < d->n; v++) out[v] = (uint8_t)d->bp_decoding[v];
    if (res) {
        res->converge = d->converge; res->bp_iteration = d->bp_iteration; res->exit_class = exit_class;
        res->reserved = d->used_guess; res->min_pm = d->min_pm;
    }
    return 0;
}


/* ------------------------------------------------------------------------------------ */
/* bp4_osd  (src/bp4_osd.pyx)                                                            */
/* ------------------------------------------------------------------------------------ */
#include <float.h>
#ifndef M_LN2
#define M_LN2 0.693147180559945309417232121458176568 /* math.h value */
#endif

static double log1pexp_(double x) { /* src/include/bpgd.cpp:399-406 */
    if (x > -log(DBL_EPSILON)) return x + log1p(exp(-x));
    return log1p(exp(x));
}
static double logaddexp_(double x, double y) { /* src/include/bpgd.cpp:408-416 */
    double const tmp = x - y;
    if (x == y) return x + M_LN2;
    if (tmp > 0) return x + log1pexp_(-tmp);
    else if (tmp <= 0) return y + log1pexp_(tmp);
    return tmp;
}

struct swo_bp4 {
    tanner hx, hz;
    swo_bp4_params p;
    int mx, mz, n, rank_x, rank_z, kx;
    double *llr_x, *llr_y, *llr_z, *prior_x, *prior_z, *lpr_x, *lpr_y, *lpr_z, *llr_post;
    signed char *synd_x, *synd_z, *cn_x, *cn_z, *vn, *dec_x, *dec_z, *osd0_x, *osd0_z, *osdw_x, *osdw_z;
    signed char *scratch, *y, *g, *Htx, *work, *yR;
    int *cols, *orig_cols, *Ht_cols;
    lu_t lux, luz;
    int bp_iteration, converge;
};

int swo_bp4_ranks(const swo_bp4 *d, int32_t *rx, int32_t *rz) { *rx = d->rank_x; *rz = d->rank_z; return 0; }
const signed char *swo_bp4_bp_decoding(const swo_bp4 *d, int z) { return z ? d->dec_z : d->dec_x; }

swo_bp4 *swo_bp4_create(int mx, int mz, int n, const int32_t *rpx, const int32_t *cix, const int32_t *rpz,
                        const int32_t *ciz, const double *px, const double *py, const double *pz,
                        const swo_bp4_params *p) {
    swo_bp4 *d = xcalloc(1, sizeof(*d));
    d->p = *p; d->mx = mx; d->mz = mz; d->n = n;
    if (d->p.osd_method == 0) d->p.osd_order = 0;
    tanner_init_csr(&d->hx, mx, n, rpx, cix);
    tanner_init_csr(&d->hz, mz, n, rpz, ciz);
    d->rank_x = gf2_rank_csr(&d->hx); d->rank_z = gf2_rank_csr(&d->hz);
    int kmin = (n - d->rank_x) < (n - d->rank_z) ? (n - d->rank_x) : (n - d->rank_z);
    if (d->p.osd_order > kmin) { tanner_free(&d->hx); tanner_free(&d->hz); free(d); return NULL; } /* bp4_osd.pyx:92-98 */
    d->kx = n - d->rank_x; /* kz = kx in the reference (bp4_osd.pyx:103-104); both bases use kx (:284) */
    double **dd[] = {&d->llr_x, &d->llr_y, &d->llr_z, &d->prior_x, &d->prior_z, &d->lpr_x, &d->lpr_y, &d->lpr_z, &d->llr_post};
    for (unsigned i = 0; i < sizeof(dd) / sizeof(dd[0]); i++) *dd[i] = xcalloc(n, sizeof(double));
    int mm = mx > mz ? mx : mz;
    signed char **cn[] = {&d->vn, &d->dec_x, &d->dec_z, &d->osd0_x, &d->osd0_z, &d->osdw_x, &d->osdw_z, &d->y};
    for (unsigned i = 0; i < sizeof(cn) / sizeof(cn[0]); i++) *cn[i] = xcalloc(n, 1);
    signed char **cm[] = {&d->synd_x, &d->synd_z, &d->cn_x, &d->cn_z, &d->scratch, &d->g, &d->Htx, &d->work};
    for (unsigned i = 0; i < sizeof(cm) / sizeof(cm[0]); i++) *cm[i] = xcalloc(mm, 1);
    d->yR = xcalloc(mm + 1, 1);
    d->cols = xcalloc(n, sizeof(int)); d->orig_cols = xcalloc(n, sizeof(int)); d->Ht_cols = xcalloc(n + 1, sizeof(int));
    if (d->p.osd_order > -1) { lu_init(&d->lux, &d->hx, d->rank_x); lu_init(&d->luz, &d->hz, d->rank_z); }
    for (int v = 0; v < n; v++) { /* bp4_osd.pyx:127-137 */
        double num = px[v] + py[v] + pz[v];
        num = 1.0 - num;
        d->llr_x[v] = log(num / px[v]); d->llr_y[v] = log(num / py[v]); d->llr_z[v] = log(num / pz[v]);
        double denom = px[v] + py[v];
        d->prior_x[v] = log((1.0 - denom) / denom);
        denom = pz[v] + py[v];
        d->prior_z[v] = log((1.0 - denom) / denom);
    }
    return d;
}

void swo_bp4_free(swo_bp4 *d) {
    if (!d) return;
    tanner_free(&d->hx); tanner_free(&d->hz);
    free(d->llr_x); free(d->llr_y); free(d->llr_z); free(d->prior_x); free(d->prior_z); free(d->lpr_x); free(d->lpr_y);
    free(d->lpr_z); free(d->llr_post); free(d->vn); free(d->dec_x); free(d->dec_z); free(d->osd0_x); free(d->osd0_z);
    free(d->osdw_x); free(d->osdw_z); free(d->y); free(d->synd_x); free(d->synd_z); free(d->cn_x); free(d->cn_z);
    free(d->scratch); free(d->g); free(d->Htx); free(d->work); free(d->yR); free(d->cols); free(d->orig_cols); free(d->Ht_cols);
    if (d->lux.B) lu_free(&d->lux);
    if (d->luz.B) lu_free(&d->luz);
    free(d);
}

/* cn_update_all (bp4_osd.pyx:483-529): plain min-sum on every check, no VN mask */
static void bp4_cn_update(tanner *t, const signed char *cn_mask, double alpha) {
    for (int cn = 0; cn < t->m; cn++) {
        double temp = 1e308;
        int sgn = (cn_mask[cn] == 1) ? 1 : 0;
        for (int e = t->row_ptr[cn]; e < t->row_ptr[cn + 1]; e++) {
            t->c2b[e] = temp; t->sgn[e] = sgn;
            if (t->b2c[e] > 50.0) t->b2c[e] = 50.0;
            else if (t->b2c[e] < -50.0) t->b2c[e] = -50.0;
            if (fabs(t->b2c[e]) < temp) temp = fabs(t->b2c[e]);
            if (t->b2c[e] <= 0) sgn = 1 - sgn;
        }
        temp = 1e308; sgn = 0;
        for (int e = t->row_ptr[cn + 1] - 1; e >= t->row_ptr[cn]; e--) {
            if (temp < t->c2b[e]) t->c2b[e] = temp;
            t->sgn[e] += sgn;
            t->c2b[e] *= ((t->sgn[e] % 2 == 0) ? 1.0 : -1.0) * alpha;
            if (fabs(t->b2c[e]) < temp) temp = fabs(t->b2c[e]);
            if (t->b2c[e] <= 0) sgn = 1 - sgn;
        }
    }
}

/* vn_update (bp4_osd.pyx:533-589) */
static void bp4_vn_update(swo_bp4 *d, int vn) {
    tanner *hx = &d->hx, *hz = &d->hz;
    double llrx = d->llr_x[vn], llry = d->llr_y[vn], llrz = d->llr_z[vn];
    double llrx_hx = 0.0;
    for (int k = hz->col_ptr[vn]; k < hz->col_ptr[vn + 1]; k++) llrx_hx += hz->c2b[hz->c2r[k]];
    double llrz_hz = 0.0;
    for (int k = hx->col_ptr[vn]; k < hx->col_ptr[vn + 1]; k++) llrz_hz += hx->c2b[hx->c2r[k]];
    double llry_all = llrx_hx + llrz_hz + llry;
    llrx_hx = llrx_hx + llrx;
    llrz_hz = llrz_hz + llrz;
    d->lpr_x[vn] = llrx_hx; d->lpr_y[vn] = llry_all; d->lpr_z[vn] = llrz_hz;
    int idx;
    if (0 < llrx_hx && 0 < llry_all && 0 < llrz_hz) idx = 0;
    else if (llrx_hx < llry_all && llrx_hx < llrz_hz) idx = 1;
    else if (llry_all > llrz_hz) idx = 2;
    else idx = 3;
    d->dec_x[vn] = (signed char)(idx % 2);
    d->dec_z[vn] = (signed char)(idx / 2);
    double num_hx = log1pexp_(-1. * llrx_hx);
    for (int k = hx->col_ptr[vn]; k < hx->col_ptr[vn + 1]; k++) {
        int e = hx->c2r[k];
        double msg = hx->c2b[e];
        double a = llrz_hz - msg, b = llry_all - msg;
        hx->b2c[e] = num_hx - logaddexp_(-1. * a, -1. * b);
    }
    double num_hz = log1pexp_(-1. * llrz_hz);
    for (int k = hz->col_ptr[vn]; k < hz->col_ptr[vn + 1]; k++) {
        int e = hz->c2r[k];
        double msg = hz->c2b[e];
        double a = llrx_hx - msg, b = llry_all - msg;
        hz->b2c[e] = num_hz - logaddexp_(-1. * a, -1. * b);
    }
}

/* osd(basis) (bp4_osd.pyx:261-368) */
static void bp4_osd_basis(swo_bp4 *d, int is_x) {
    const int n = d->n;
    tanner *H = is_x ? &d->hx : &d->hz;
    lu_t *lu = is_x ? &d->lux : &d->luz;
    const int rank = is_x ? d->rank_x : d->rank_z, m = H->m;
    int k = d->kx;
    if (k > n - rank) k = n - rank; /* the reference would read past the pivot block; identical when rank_x == rank_z */
    const signed char *synd = is_x ? d->synd_x : d->synd_z;
    signed char *osd0 = is_x ? d->osd0_z : d->osd0_x, *osdw = is_x ? d->osdw_z : d->osdw_x;
    const double *prior = is_x ? d->prior_x : d->prior_z;
    for (int v = 0; v < n; v++)
        d->llr_post[v] = is_x ? log1pexp_(-1. * d->lpr_x[v]) - logaddexp_(-1. * d->lpr_y[v], -1. * d->lpr_z[v])
                              : log1pexp_(-1. * d->lpr_z[v]) - logaddexp_(-1. * d->lpr_y[v], -1. * d->lpr_x[v]);
    index_sort(d->llr_post, d->cols, n);
    for (int v = 0; v < n; v++) d->orig_cols[v] = d->cols[v];
    lu_decomp_osd(lu, H, d->cols);
    lu_solve(lu, d->cols, synd, osd0, d->work, d->yR);
    double min_pm = 0.0;
    for (int v = 0; v < n; v++) { if (osd0[v]) min_pm += prior[v]; osdw[v] = osd0[v]; }
    if (d->p.osd_order == 0) return;
    int counter = 0;
    for (int i = 0; i < n; i++) {
        int cn = d->orig_cols[i], in_pivot = 0;
        for (int j = 0; j < rank; j++) if (d->cols[j] == cn) { in_pivot = 1; break; }
        if (!in_pivot) { if (counter < k) d->Ht_cols[counter] = cn; counter++; }
    }
    const int w = d->p.osd_order;
    const int kset = n - rank; /* osd_cs_setup_x/z use n - rank_x / n - rank_z (bp4_osd.pyx:153,176) */
    long ncand = (d->p.osd_method == 1) ? (1L << w) : (long)kset + (long)w * (w - 1) / 2;
    signed char *x = xcalloc(k > kset ? k : kset + 1, 1);
    for (long l = 0; l < ncand; l++) {
        memset(x, 0, k > kset ? k : kset + 1);
        if (d->p.osd_method == 1) { long v = l; for (int i = 0; i < kset && v; i++) { x[i] = (signed char)(v % 2); v /= 2; } }
        else if (l < kset) x[l] = 1;
        else { long q = l - kset; int i = 0; while (q >= w - 1 - i) { q -= w - 1 - i; i++; } x[i] = 1; x[i + 1 + q] = 1; }
        for (int c = 0; c < m; c++) d->Htx[c] = 0;
        for (int j = 0; j < k; j++)
            if (x[j]) { int v = d->Ht_cols[j]; for (int q = H->col_ptr[v]; q < H->col_ptr[v + 1]; q++) d->Htx[H->row_idx[q]] ^= 1; }
        for (int c = 0; c < m; c++) d->g[c] = (signed char)(synd[c] ^ d->Htx[c]);
        lu_solve(lu, d->cols, d->g, d->y, d->work, d->yR);
        for (int j = 0; j < k; j++) d->y[d->Ht_cols[j]] = x[j];
        double pm = 0.0;
        for (int v = 0; v < n; v++) if (d->y[v]) pm += prior[v];
        if (pm < min_pm) { min_pm = pm; for (int v = 0; v < n; v++) osdw[v] = d->y[v]; }
    }
    free(x);
}

/* reset (bp4_osd.pyx:370-385) + bp_init (:425-442); every VN is undecided after reset */
static void bp4_reset_init(swo_bp4 *d) {
    const int n = d->n;
    for (int c = 0; c < d->mx; c++) d->cn_x[c] = d->synd_x[c];
    for (int c = 0; c < d->mz; c++) d->cn_z[c] = d->synd_z[c];
    d->bp_iteration = 0;
    for (int v = 0; v < n; v++) { d->vn[v] = -1; d->dec_x[v] = d->dec_z[v] = 0; }
    for (int v = 0; v < n; v++) {
        double llrx = d->llr_x[v], llry = d->llr_y[v], llrz = d->llr_z[v];
        double msg_x = log1pexp_(-1. * llrx) - logaddexp_(-1. * llry, -1. * llrz);
        for (int k = d->hx.col_ptr[v]; k < d->hx.col_ptr[v + 1]; k++) d->hx.b2c[d->hx.c2r[k]] = msg_x;
        double msg_z = log1pexp_(-1. * llrz) - logaddexp_(-1. * llry, -1. * llrz);
        for (int k = d->hz.col_ptr[v]; k < d->hz.col_ptr[v + 1]; k++) d->hz.b2c[d->hz.c2r[k]] = msg_z;
    }
}

/* bp4_decode_llr (bp4_osd.pyx:444-481): decided VNs keep their messages and decisions (:456-459), checks are
   never masked (:483-529) */
static int bp4_run(swo_bp4 *d) {
    d->converge = 0;
    for (int it = 0; it < d->p.max_iter; it++) {
        d->bp_iteration += 1;
        bp4_cn_update(&d->hx, d->cn_x, d->p.ms_scaling_factor);
        bp4_cn_update(&d->hz, d->cn_z, d->p.ms_scaling_factor);
        for (int v = 0; v < d->n; v++) if (d->vn[v] == -1) bp4_vn_update(d, v);
        if (syndrome_matches(&d->hx, d->dec_z, d->synd_x, d->scratch) &&
            syndrome_matches(&d->hz, d->dec_x, d->synd_z, d->scratch)) { d->converge = 1; return 1; }
    }
    return 0;
}

/* camel_decode (bp4_osd.pyx:223-247): the last qubit is fixed to I, X, Z, Y in turn (vn_set_value :388-423),
   plain BP4 runs on the rest, the converged run of smallest path metric (cal_pm :249-258) wins; strict <
   keeps the earliest.  The returned osd0 vectors and min_pm persist in the object like in the reference:
   when no run converges the previous call's vectors come back (zeros for a new object). */
int swo_bp4_camel_decode(swo_bp4 *d, const uint8_t *sx, const uint8_t *sz, uint8_t *out_x, uint8_t *out_z, swo_result *res) {
    const int n = d->n;
    for (int c = 0; c < d->mx; c++) d->synd_x[c] = (signed char)sx[c];
    for (int c = 0; c < d->mz; c++) d->synd_z[c] = (signed char)sz[c];
    double min_pm = 10000.0;
    for (int value = 0; value < 4; value++) {
        bp4_reset_init(d);
        const int vn = n - 1, x = value % 2, z = value / 2;
        d->vn[vn] = (signed char)value;
        d->dec_x[vn] = (signed char)x; d->dec_z[vn] = (signed char)z;
        if (z) for (int k = d->hx.col_ptr[vn]; k < d->hx.col_ptr[vn + 1]; k++) { int cn = d->hx.row_idx[k]; d->cn_x[cn] = (signed char)(1 - d->cn_x[cn]); }
        if (x) for (int k = d->hz.col_ptr[vn]; k < d->hz.col_ptr[vn + 1]; k++) { int cn = d->hz.row_idx[k]; d->cn_z[cn] = (signed char)(1 - d->cn_z[cn]); }
        if (bp4_run(d)) {
            double pm = 0.0;
            for (int v = 0; v < n; v++) {
                if (d->dec_x[v] && d->dec_z[v]) pm += d->llr_y[v];
                else if (d->dec_x[v]) pm += d->llr_x[v];
                else if (d->dec_z[v]) pm += d->llr_z[v];
            }
            if (pm < min_pm) {
                min_pm = pm;
                for (int v = 0; v < n; v++) { d->osd0_x[v] = d->dec_x[v]; d->osd0_z[v] = d->dec_z[v]; }
            }
        }
    }
    if (min_pm < 9999.0) d->converge = 1;
    for (int v = 0; v < n; v++) { out_x[v] = (uint8_t)d->osd0_x[v]; out_z[v] = (uint8_t)d->osd0_z[v]; }
    if (res) { res->converge = d->converge; res->bp_iteration = d->bp_iteration; res->exit_class = SWO_EXIT_PRE; res->reserved = 0; res->min_pm = min_pm; }
    return 0;
}

int swo_bp4_decode(swo_bp4 *d, const uint8_t *sx, const uint8_t *sz, uint8_t *out_x, uint8_t *out_z,
                   swo_result *res, double *lpr, uint8_t *osd0_x, uint8_t *osd0_z) {
    const int n = d->n;
    for (int c = 0; c < d->mx; c++) d->synd_x[c] = (signed char)sx[c];
    for (int c = 0; c < d->mz; c++) d->synd_z[c] = (signed char)sz[c];
    bp4_reset_init(d);
    bp4_run(d);
    int exit_class;
    const signed char *rx = d->dec_x, *rz = d->dec_z;
    if (d->converge) {
        for (int v = 0; v < n; v++) { d->osd0_x[v] = d->dec_x[v]; d->osd0_z[v] = d->dec_z[v]; }
        exit_class = SWO_EXIT_PRE;
    } else if (d->p.osd_order > -1) {
        bp4_osd_basis(d, 1);
        bp4_osd_basis(d, 0);
        rx = d->osdw_x; rz = d->osdw_z;
        exit_class = SWO_EXIT_OSD;
    } else exit_class = SWO_EXIT_NO_OSD;
    for (int v = 0; v < n; v++) { out_x[v] = (uint8_t)rx[v]; out_z[v] = (uint8_t)rz[v]; }
    if (lpr) for (int v = 0; v < n; v++) { lpr[3 * v] = d->lpr_x[v]; lpr[3 * v + 1] = d->lpr_y[v]; lpr[3 * v + 2] = d->lpr_z[v]; }
    if (osd0_x) for (int v = 0; v < n; v++) { osd0_x[v] = (uint8_t)d->osd0_x[v]; osd0_z[v] = (uint8_t)d->osd0_z[v]; }
    if (res) { res->converge = d->converge; res->bp_iteration = d->bp_iteration; res->exit_class = exit_class; res->reserved = 0; res->min_pm = 0.0; }
    return 0;
}
