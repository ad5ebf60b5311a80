/*
 * smg_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).  See smg_oracle.h.
 *
 * Every function cites the reference file:line it restates (paths relative to the
 * reference checkout, HTDerekLiu/surface_multigrid_code).  Arithmetic order is the
 * reference's: Eigen column-major sparse kernels (per-row sums in ascending column
 * index), forward lexicographic Gauss-Seidel, `u += solve(B)` at the coarsest level.
 * Compile with -O2/-O3 and WITHOUT -ffast-math / -march=native (no FMA contraction),
 * like the reference's own CMake (03_mg_solver/CMakeLists.txt sets neither).
 *
 * PARITY UNPINNED (see header).
 */
#include "smg_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ utilities */

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void *xmalloc(size_t n)
{
    void *p = malloc(n ? n : 1);
    if (!p) { fprintf(stderr, "smg_oracle: out of memory\n"); abort(); }
    return p;
}
static void *xcalloc(size_t n, size_t s)
{
    void *p = calloc(n ? n : 1, s ? s : 1);
    if (!p) { fprintf(stderr, "smg_oracle: out of memory\n"); abort(); }
    return p;
}

void orc_csc_free(orc_csc *m)
{
    if (!m) return;
    free(m->colptr); free(m->rowidx); free(m->val);
    memset(m, 0, sizeof(*m));
}

static void csc_alloc(orc_csc *m, int nr, int nc, long nnz)
{
    m->n_rows = nr; m->n_cols = nc;
    m->colptr = (int *)xcalloc((size_t)nc + 1, sizeof(int));
    m->rowidx = (int *)xmalloc((size_t)nnz * sizeof(int));
    m->val = (double *)xmalloc((size_t)nnz * sizeof(double));
}

static void csc_copy(orc_csc *dst, const orc_csc *src)
{
    orc_csc_free(dst);
    long nnz = src->colptr ? src->colptr[src->n_cols] : 0;
    csc_alloc(dst, src->n_rows, src->n_cols, nnz);
    if (src->colptr) memcpy(dst->colptr, src->colptr, ((size_t)src->n_cols + 1) * sizeof(int));
    if (nnz) {
        memcpy(dst->rowidx, src->rowidx, (size_t)nnz * sizeof(int));
        memcpy(dst->val, src->val, (size_t)nnz * sizeof(double));
    }
}

static void csc_from_arrays(orc_csc *dst, int nr, int nc, const int *colptr, const int *rowidx,
                            const double *val)
{
    orc_csc_free(dst);
    long nnz = colptr[nc];
    csc_alloc(dst, nr, nc, nnz);
    memcpy(dst->colptr, colptr, ((size_t)nc + 1) * sizeof(int));
    memcpy(dst->rowidx, rowidx, (size_t)nnz * sizeof(int));
    memcpy(dst->val, val, (size_t)nnz * sizeof(double));
}

/* Eigen SparseMatrix::transpose() evaluated into a column-major matrix: a counting sort,
 * so row indices inside each output column come out ascending. */
static void csc_transpose(orc_csc *dst, const orc_csc *src)
{
    orc_csc t; memset(&t, 0, sizeof(t));
    long nnz = src->colptr[src->n_cols];
    csc_alloc(&t, src->n_cols, src->n_rows, nnz);
    for (long p = 0; p < nnz; p++) t.colptr[src->rowidx[p] + 1]++;
    for (int i = 0; i < src->n_rows; i++) t.colptr[i + 1] += t.colptr[i];
    int *next = (int *)xmalloc((size_t)src->n_rows * sizeof(int));
    memcpy(next, t.colptr, (size_t)src->n_rows * sizeof(int));
    for (int j = 0; j < src->n_cols; j++)
        for (int p = src->colptr[j]; p < src->colptr[j + 1]; p++) {
            int q = next[src->rowidx[p]]++;
            t.rowidx[q] = j;
            t.val[q] = src->val[p];
        }
    free(next);
    orc_csc_free(dst);
    *dst = t;
}

static int cmp_int(const void *a, const void *b)
{
    int x = *(const int *)a, y = *(const int *)b;
    return (x > y) - (x < y);
}

/* Eigen column-major sparse * sparse (conservative_sparse_sparse_product): for each output
 * column j, for each stored B(k,j) in ascending k, for each stored A(i,k): acc[i] += A(i,k)*B(k,j)
 * (first touch assigns).  No pruning: structural entries stay even when numerically zero.
 * Used for  mg[lv].A = mg[lv].PT * mg[lv-1].A * mg[lv].P  (min_quad_with_fixed_mg.cpp:25, :227),
 * which C++ evaluates left to right: (PT * A) * P. */
static void csc_mul(orc_csc *dst, const orc_csc *A, const orc_csc *B)
{
    int nr = A->n_rows, nc = B->n_cols;
    char *mask = (char *)xcalloc((size_t)nr, 1);
    double *acc = (double *)xmalloc((size_t)nr * sizeof(double));
    int *idx = (int *)xmalloc((size_t)nr * sizeof(int));
    /* pass 1: count */
    int *cp = (int *)xcalloc((size_t)nc + 1, sizeof(int));
    for (int j = 0; j < nc; j++) {
        int cnt = 0;
        for (int pb = B->colptr[j]; pb < B->colptr[j + 1]; pb++) {
            int k = B->rowidx[pb];
            for (int pa = A->colptr[k]; pa < A->colptr[k + 1]; pa++) {
                int i = A->rowidx[pa];
                if (!mask[i]) { mask[i] = 1; idx[cnt++] = i; }
            }
        }
        for (int t = 0; t < cnt; t++) mask[idx[t]] = 0;
        cp[j + 1] = cp[j] + cnt;
    }
    orc_csc C; memset(&C, 0, sizeof(C));
    csc_alloc(&C, nr, nc, cp[nc]);
    memcpy(C.colptr, cp, ((size_t)nc + 1) * sizeof(int));
    free(cp);
    /* pass 2: numeric */
    for (int j = 0; j < nc; j++) {
        int cnt = 0;
        for (int pb = B->colptr[j]; pb < B->colptr[j + 1]; pb++) {
            int k = B->rowidx[pb];
            double y = B->val[pb];
            for (int pa = A->colptr[k]; pa < A->colptr[k + 1]; pa++) {
                int i = A->rowidx[pa];
                double x = A->val[pa];
                if (!mask[i]) { mask[i] = 1; acc[i] = x * y; idx[cnt++] = i; }
                else acc[i] += x * y;
            }
        }
        qsort(idx, (size_t)cnt, sizeof(int), cmp_int);
        int base = C.colptr[j];
        for (int t = 0; t < cnt; t++) {
            C.rowidx[base + t] = idx[t];
            C.val[base + t] = acc[idx[t]];
            mask[idx[t]] = 0;
        }
    }
    free(mask); free(acc); free(idx);
    orc_csc_free(dst);
    *dst = C;
}

/* igl::slice(X, R, C, Y):  Y(i,j) = X(R[i], C[j]).  R == NULL / C == NULL mean "all"
 * (igl::slice(X,R,1,Y) and igl::slice(X,C,2,Y)).  Indices are assumed unique. */
typedef struct { int row, pos; } rowpos;
static int cmp_rowpos(const void *a, const void *b)
{
    int x = ((const rowpos *)a)->row, y = ((const rowpos *)b)->row;
    return (x > y) - (x < y);
}
static void csc_slice(orc_csc *dst, const orc_csc *X, const int *R, int nR, const int *C, int nC)
{
    int nr = R ? nR : X->n_rows, nc = C ? nC : X->n_cols;
    int *rmap = NULL; /* old row -> new row or -1 */
    if (R) {
        rmap = (int *)xmalloc((size_t)X->n_rows * sizeof(int));
        for (int i = 0; i < X->n_rows; i++) rmap[i] = -1;
        for (int i = 0; i < nR; i++) rmap[R[i]] = i;
    }
    int *cp = (int *)xcalloc((size_t)nc + 1, sizeof(int));
    for (int j = 0; j < nc; j++) {
        int cj = C ? C[j] : j, cnt = 0;
        for (int p = X->colptr[cj]; p < X->colptr[cj + 1]; p++)
            if (!rmap || rmap[X->rowidx[p]] >= 0) cnt++;
        cp[j + 1] = cp[j] + cnt;
    }
    orc_csc Y; memset(&Y, 0, sizeof(Y));
    csc_alloc(&Y, nr, nc, cp[nc]);
    memcpy(Y.colptr, cp, ((size_t)nc + 1) * sizeof(int));
    free(cp);
    rowpos *tmp = (rowpos *)xmalloc((size_t)(X->n_rows > 0 ? X->n_rows : 1) * sizeof(rowpos));
    for (int j = 0; j < nc; j++) {
        int cj = C ? C[j] : j, cnt = 0;
        for (int p = X->colptr[cj]; p < X->colptr[cj + 1]; p++) {
            int r = rmap ? rmap[X->rowidx[p]] : X->rowidx[p];
            if (r >= 0) { tmp[cnt].row = r; tmp[cnt].pos = p; cnt++; }
        }
        if (rmap) qsort(tmp, (size_t)cnt, sizeof(rowpos), cmp_rowpos);
        int base = Y.colptr[j];
        for (int t = 0; t < cnt; t++) {
            Y.rowidx[base + t] = tmp[t].row;
            Y.val[base + t] = X->val[tmp[t].pos];
        }
    }
    free(tmp); free(rmap);
    orc_csc_free(dst);
    *dst = Y;
}

/* Eigen col-major sparse * dense:  Y = A * X.  Y is zeroed, then for each dense column c,
 * for each sparse column j:  Y(i,c) += A(i,j) * X(j,c)  for the stored i of column j
 * (sparse_time_dense_product_impl, ColMajor lhs, alpha = 1).  => every Y(i,c) is a running
 * sum over ascending j. */
void orc_csc_times_dense(const orc_csc *A, const double *X, int ldx, int k, double *Y, int ldy)
{
    for (int c = 0; c < k; c++) {
        double *y = Y + (size_t)c * (size_t)ldy;
        const double *x = X + (size_t)c * (size_t)ldx;
        for (int i = 0; i < A->n_rows; i++) y[i] = 0.0;
        for (int j = 0; j < A->n_cols; j++) {
            double xj = x[j];
            for (int p = A->colptr[j]; p < A->colptr[j + 1]; p++)
                y[A->rowidx[p]] += A->val[p] * xj;
        }
    }
}

/* below this many rows a loop is not worth a fork/join */
#define ORC_PAR_MIN_ROWS 4096

/* all-core mode only: Y = A X computed row-wise from AT = A^T (column i of AT = row i of A, ascending j): every Y(i,c) is
 * the same running sum over ascending j as in orc_csc_times_dense, so the result is bit-identical. */
static void csc_times_dense_rows(const orc_csc *AT, const double *X, int ldx, int k, double *Y, int ldy)
{
    for (int c = 0; c < k; c++) {
        double *y = Y + (size_t)c * (size_t)ldy;
        const double *x = X + (size_t)c * (size_t)ldx;
#pragma omp parallel for schedule(static) if (AT->n_cols > ORC_PAR_MIN_ROWS)
        for (int i = 0; i < AT->n_cols; i++) {
            double s = 0.0;
            for (int p = AT->colptr[i]; p < AT->colptr[i + 1]; p++) s += AT->val[p] * x[AT->rowidx[p]];
            y[i] = s;
        }
    }
}

/* ------------------------------------------------------------------ sparse LDL^T */

static void ldlt_free(orc_ldlt *s)
{
    free(s->perm); free(s->first); free(s->rowoff); free(s->L); free(s->D); free(s->work);
    memset(s, 0, sizeof(*s));
}

/* reverse Cuthill-McKee on the pattern of a symmetric CSC matrix (fill-reducing ordering
 * standing in for Eigen's AMDOrdering). */
static void rcm_order(const orc_csc *A, int *perm)
{
    int n = A->n_rows;
    int *deg = (int *)xmalloc((size_t)n * sizeof(int));
    char *seen = (char *)xcalloc((size_t)n, 1);
    int *queue = (int *)xmalloc((size_t)n * sizeof(int));
    rowpos *nb = (rowpos *)xmalloc((size_t)n * sizeof(rowpos));
    for (int i = 0; i < n; i++) deg[i] = A->colptr[i + 1] - A->colptr[i];
    int filled = 0;
    while (filled < n) {
        /* start: unseen vertex of minimum degree */
        int s = -1;
        for (int i = 0; i < n; i++)
            if (!seen[i] && (s < 0 || deg[i] < deg[s])) s = i;
        int head = filled;
        queue[filled++] = s; seen[s] = 1;
        while (head < filled) {
            int v = queue[head++], cnt = 0;
            for (int p = A->colptr[v]; p < A->colptr[v + 1]; p++) {
                int w = A->rowidx[p];
                if (!seen[w]) { seen[w] = 1; nb[cnt].row = deg[w]; nb[cnt].pos = w; cnt++; }
            }
            qsort(nb, (size_t)cnt, sizeof(rowpos), cmp_rowpos);
            for (int t = 0; t < cnt; t++) queue[filled++] = nb[t].pos;
        }
    }
    for (int i = 0; i < n; i++) perm[i] = queue[n - 1 - i];
    free(deg); free(seen); free(queue); free(nb);
}

/* solver.compute(Ac)  (min_quad_with_fixed_mg.cpp:47-48, :253-254) */
static int ldlt_compute(orc_ldlt *s, const orc_csc *A)
{
    ldlt_free(s);
    int n = A->n_rows;
    s->n = n;
    s->perm = (int *)xmalloc((size_t)n * sizeof(int));
    rcm_order(A, s->perm);
    int *iperm = (int *)xmalloc((size_t)n * sizeof(int));
    for (int i = 0; i < n; i++) iperm[s->perm[i]] = i;
    s->first = (int *)xmalloc((size_t)n * sizeof(int));
    for (int i = 0; i < n; i++) s->first[i] = i;
    for (int j = 0; j < n; j++) {
        int pj = iperm[j];
        for (int p = A->colptr[j]; p < A->colptr[j + 1]; p++) {
            int pi = iperm[A->rowidx[p]];
            if (pj < pi && pj < s->first[pi]) s->first[pi] = pj;
        }
    }
    s->rowoff = (long *)xmalloc(((size_t)n + 1) * sizeof(long));
    s->rowoff[0] = 0;
    for (int i = 0; i < n; i++) s->rowoff[i + 1] = s->rowoff[i] + (i - s->first[i]);
    s->L = (double *)xcalloc((size_t)s->rowoff[n], sizeof(double));
    s->D = (double *)xcalloc((size_t)n, sizeof(double));
    s->work = (double *)xmalloc((size_t)n * sizeof(double));
    /* scatter the lower triangle of the permuted matrix */
    for (int j = 0; j < n; j++) {
        int pj = iperm[j];
        for (int p = A->colptr[j]; p < A->colptr[j + 1]; p++) {
            int pi = iperm[A->rowidx[p]];
            if (pi == pj) s->D[pi] = A->val[p];
            else if (pj < pi) s->L[s->rowoff[pi] + (pj - s->first[pi])] = A->val[p];
        }
    }
    free(iperm);
    /* skyline LDL^T, row by row; y_ij = L_ij * D_j kept in place until the row is finished */
    int ok = 1;
    for (int i = 0; i < n; i++) {
        int fi = s->first[i];
        double *Li = s->L + s->rowoff[i] - fi; /* Li[j] = entry (i,j) */
        for (int j = fi; j < i; j++) {
            int fj = s->first[j];
            const double *Lj = s->L + s->rowoff[j] - fj;
            int k0 = fi > fj ? fi : fj;
            double sum = Li[j];
            for (int k = k0; k < j; k++) sum -= Li[k] * Lj[k];
            Li[j] = sum; /* = y_ij */
        }
        double d = s->D[i];
        for (int j = fi; j < i; j++) {
            double y = Li[j];
            double l = y / s->D[j];
            d -= y * l;
            Li[j] = l;
        }
        s->D[i] = d;
        if (!(d != 0.0) || !isfinite(d)) ok = 0;
    }
    return ok ? 0 : -1;
}

/* x = solver.solve(b), one column */
static void ldlt_solve(const orc_ldlt *s, const double *b, double *x)
{
    int n = s->n;
    double *y = s->work;
    for (int i = 0; i < n; i++) y[i] = b[s->perm[i]];
    for (int i = 0; i < n; i++) {
        int fi = s->first[i];
        const double *Li = s->L + s->rowoff[i] - fi;
        double sum = y[i];
        for (int j = fi; j < i; j++) sum -= Li[j] * y[j];
        y[i] = sum;
    }
    for (int i = 0; i < n; i++) y[i] /= s->D[i];
    for (int i = n - 1; i >= 0; i--) {
        int fi = s->first[i];
        const double *Li = s->L + s->rowoff[i] - fi;
        double xi = y[i];
        for (int j = fi; j < i; j++) y[j] -= Li[j] * xi;
    }
    for (int i = 0; i < n; i++) x[s->perm[i]] = y[i];
}

/* ------------------------------------------------------------------ container */

orc_mg *orc_mg_create(int n_levels)
{
    if (n_levels < 1) return NULL;
    orc_mg *mg = (orc_mg *)xcalloc(1, sizeof(orc_mg));
    mg->n_levels = n_levels;
    mg->lv = (orc_level *)xcalloc((size_t)n_levels, sizeof(orc_level));
    return mg;
}

static void data_free(orc_mqwf_data *d)
{
    free(d->known); free(d->unknown);
    orc_csc_free(&d->LHS); orc_csc_free(&d->Auk);
    memset(d, 0, sizeof(*d));
}

void orc_mg_destroy(orc_mg *mg)
{
    if (!mg) return;
    for (int l = 0; l < mg->n_levels; l++) {
        orc_level *L = &mg->lv[l];
        orc_csc_free(&L->P_full); orc_csc_free(&L->A); orc_csc_free(&L->P); orc_csc_free(&L->PT);
        free(L->A_diag);
    }
    free(mg->lv);
    data_free(&mg->data);
    ldlt_free(&mg->solver);
    for (int l = 0; l < mg->n_levels; l++) {
        if (mg->color_ptr) free(mg->color_ptr[l]);
        if (mg->AT) orc_csc_free(&mg->AT[l]);
    }
    free(mg->color_ptr); free(mg->n_colors); free(mg->AT);
    free(mg->smoother); free(mg->omega);
    free(mg);
}

/* mg_precompute.cpp:71-77:  data.P = P; data.PT = P.transpose(); data.P_full = P; */
int orc_mg_set_prolong(orc_mg *mg, int lv, int n_rows, int n_cols, const int *colptr,
                       const int *rowidx, const double *val)
{
    if (!mg || lv < 1 || lv >= mg->n_levels) return -1;
    orc_level *L = &mg->lv[lv];
    csc_from_arrays(&L->P_full, n_rows, n_cols, colptr, rowidx, val);
    csc_copy(&L->P, &L->P_full);
    csc_transpose(&L->PT, &L->P);
    return 0;
}

/* ------------------------------------------------------------------ precompute */

/* shared tail of both precompute overloads:
 *   Galerkin  mg[lv].A = mg[lv].PT * mg[lv-1].A * mg[lv].P          (.cpp:22-26 / :223-228)
 *   coarsest  A(ii,ii) += 1e-12                                      (.cpp:32-36 / :236-241)
 *   A_diag = A.diagonal() on every level                             (.cpp:39-41 / :244-246)
 *   solver.compute(A_coarsest)                                       (.cpp:47-48 / :253-254) */
static int precompute_tail(orc_mg *mg, int recompute_PT)
{
    for (int lv = 1; lv < mg->n_levels; lv++) {
        orc_level *L = &mg->lv[lv];
        if (recompute_PT) csc_transpose(&L->PT, &L->P); /* .cpp:226 (known-overload only) */
        orc_csc tmp; memset(&tmp, 0, sizeof(tmp));
        csc_mul(&tmp, &L->PT, &mg->lv[lv - 1].A);
        csc_mul(&L->A, &tmp, &L->P);
        orc_csc_free(&tmp);
    }
    {
        orc_level *L = &mg->lv[mg->n_levels - 1];
        for (int ii = 0; ii < L->A.n_rows; ii++) {
            int found = 0;
            for (int p = L->A.colptr[ii]; p < L->A.colptr[ii + 1]; p++)
                if (L->A.rowidx[p] == ii) { L->A.val[p] += 1e-12; found = 1; break; }
            if (!found) return -2; /* coeffRef would insert; a Galerkin SPD operator always has it */
        }
    }
    for (int lv = 0; lv < mg->n_levels; lv++) {
        orc_level *L = &mg->lv[lv];
        free(L->A_diag);
        L->A_diag = (double *)xcalloc((size_t)L->A.n_rows, sizeof(double));
        for (int j = 0; j < L->A.n_cols; j++)
            for (int p = L->A.colptr[j]; p < L->A.colptr[j + 1]; p++)
                if (L->A.rowidx[p] == j) L->A_diag[j] = L->A.val[p];
    }
    /* all-core mode: the row-wise images belong to the previous matrices */
    if (mg->AT) for (int lv = 0; lv < mg->n_levels; lv++) {
        orc_csc_free(&mg->AT[lv]);
        if (mg->par) csc_transpose(&mg->AT[lv], &mg->lv[lv].A);
    }
    return ldlt_compute(&mg->solver, &mg->lv[mg->n_levels - 1].A);
}

/* min_quad_with_fixed_mg.cpp:3-51 */
int orc_precompute(orc_mg *mg, int n, const int *colptr, const int *rowidx, const double *val)
{
    if (!mg) return -1;
    data_free(&mg->data);
    csc_from_arrays(&mg->data.LHS, n, n, colptr, rowidx, val); /* :17 */
    mg->data.n = n;                                             /* :18 */
    csc_copy(&mg->lv[0].A, &mg->data.LHS);                      /* :22 */
    return precompute_tail(mg, 0);
}

/* min_quad_with_fixed_mg.cpp:137-257 */
int orc_precompute_known(orc_mg *mg, int n, const int *colptr, const int *rowidx,
                         const double *val, const int *known, int n_known)
{
    if (!mg || mg->n_levels < 2) return -1;
    data_free(&mg->data);
    orc_mqwf_data *d = &mg->data;
    orc_csc A; memset(&A, 0, sizeof(A));
    csc_from_arrays(&A, n, n, colptr, rowidx, val);
    /* unknown = setdiff(0..n-1, known): sorted ascending (:155-158) */
    char *isk = (char *)xcalloc((size_t)n, 1);
    for (int i = 0; i < n_known; i++) isk[known[i]] = 1;
    d->known = (int *)xmalloc((size_t)n_known * sizeof(int));
    memcpy(d->known, known, (size_t)n_known * sizeof(int));
    d->n_known = n_known;
    d->unknown = (int *)xmalloc((size_t)n * sizeof(int));
    d->n_unknown = 0;
    for (int i = 0; i < n; i++) if (!isk[i]) d->unknown[d->n_unknown++] = i;
    free(isk);
    csc_slice(&d->LHS, &A, d->unknown, d->n_unknown, d->unknown, d->n_unknown); /* :166-167,:175 */
    csc_slice(&d->Auk, &A, d->unknown, d->n_unknown, d->known, d->n_known);     /* :169-170,:176 */
    d->n = n;                                                                    /* :177 */
    orc_csc_free(&A);

    /* re-organise P so that it only contains unknowns (:185-220) */
    csc_slice(&mg->lv[1].P, &mg->lv[1].P_full, d->unknown, d->n_unknown, NULL, 0); /* :185 */
    for (int lv = 1; lv < mg->n_levels; lv++) {
        orc_csc *P = &mg->lv[lv].P;
        int *keep = (int *)xmalloc((size_t)(P->n_cols > 0 ? P->n_cols : 1) * sizeof(int));
        int nkeep = 0;
        for (int c = 0; c < P->n_cols; c++)
            for (int p = P->colptr[c]; p < P->colptr[c + 1]; p++)
                if (P->val[p] > 1e-15) { keep[nkeep++] = c; break; }          /* :195-202 */
        if (nkeep < P->n_cols) {                                              /* :206 */
            orc_csc Ptmp; memset(&Ptmp, 0, sizeof(Ptmp));
            csc_copy(&Ptmp, P);
            csc_slice(P, &Ptmp, NULL, 0, keep, nkeep);                        /* :210-211 */
            orc_csc_free(&Ptmp);
            if (lv < mg->n_levels - 1)
                csc_slice(&mg->lv[lv + 1].P, &mg->lv[lv + 1].P_full, keep, nkeep, NULL, 0); /* :213-214 */
            free(keep);
        } else {
            free(keep);
            break;                                                            /* :216-219 */
        }
    }
    csc_copy(&mg->lv[0].A, &d->LHS);                                          /* :223 */
    return precompute_tail(mg, 1);
}

/* ------------------------------------------------------------------ V-cycle pieces */

/* mg_VCycle.cpp:62-70   Au = mg[lv].A * u */
void orc_A(const orc_mg *mg, int lv, const double *u, int k, double *Au)
{
    const orc_csc *A = &mg->lv[lv].A;
    if (mg->par && mg->AT && mg->AT[lv].colptr) { csc_times_dense_rows(&mg->AT[lv], u, A->n_cols, k, Au, A->n_rows); return; }
    orc_csc_times_dense(A, u, A->n_cols, k, Au, A->n_rows);
}
/* mg_VCycle.cpp:72-81   Rx = mg[lv+1].PT * x */
void orc_restrict(const orc_mg *mg, int lv, const double *x, int k, double *Rx)
{
    const orc_csc *PT = &mg->lv[lv + 1].PT;
    if (mg->par) { csc_times_dense_rows(&mg->lv[lv + 1].P, x, PT->n_cols, k, Rx, PT->n_rows); return; }   /* P = (PT)^T */
    orc_csc_times_dense(PT, x, PT->n_cols, k, Rx, PT->n_rows);
}
/* mg_VCycle.cpp:83-92   Px = mg[lv+1].P * x */
void orc_prolong(const orc_mg *mg, int lv, const double *x, int k, double *Px)
{
    const orc_csc *P = &mg->lv[lv + 1].P;
    if (mg->par) { csc_times_dense_rows(&mg->lv[lv + 1].PT, x, P->n_cols, k, Px, P->n_rows); return; }   /* PT = P^T */
    orc_csc_times_dense(P, x, P->n_cols, k, Px, P->n_rows);
}

/* mg_VCycle.cpp:113-178: `iters` forward lexicographic Gauss-Seidel sweeps, in place.
 * Column colIdx of the (symmetric) CSC matrix is walked instead of row colIdx (:149-150);
 * the diagonal is skipped by index test (:153); division by the cached A_diag (:157).
 * dim > 1: the dense-column loop is outermost (:161-177) => k independent sweeps. */
/* Damped Jacobi (extension, see smg_oracle.h: orc_set_smoother): `iters` sweeps, every row from the old iterate. */
static void relax_jacobi(orc_mg *mg, int lv, const double *B, int k, int iters, double *u, double omega)
{
    const orc_csc *A = &mg->lv[lv].A;
    const double *diag = mg->lv[lv].A_diag;
    int n = A->n_rows;
    double *tmp = (double *)xmalloc((size_t)n * sizeof(double));
    for (int ri = 0; ri < k; ri++) {
        double *uc = u + (size_t)ri * (size_t)n;
        const double *bc = B + (size_t)ri * (size_t)n;
        for (int iter = 0; iter < iters; iter++) {
#pragma omp parallel for schedule(static) if (mg->par && n > ORC_PAR_MIN_ROWS)
            for (int colIdx = 0; colIdx < n; colIdx++) {
                double sum = 0;
                for (int p = A->colptr[colIdx]; p < A->colptr[colIdx + 1]; p++)
                    if (A->rowidx[p] != colIdx) sum += A->val[p] * uc[A->rowidx[p]];
                double t = (bc[colIdx] - sum) / diag[colIdx];
                tmp[colIdx] = uc[colIdx] + omega * (t - uc[colIdx]);
            }
            memcpy(uc, tmp, (size_t)n * sizeof(double));
        }
    }
    free(tmp);
}

/* Gershgorin bound of the spectrum of D^-1 A (extension, see smg_oracle.h) */
double orc_spectral_bound(const orc_mg *mg, int lv)
{
    const orc_csc *A = &mg->lv[lv].A;
    const double *diag = mg->lv[lv].A_diag;
    double lam = 0.0;
    for (int i = 0; i < A->n_cols; i++) {
        double sum = 0.0;
        for (int p = A->colptr[i]; p < A->colptr[i + 1]; p++) sum += fabs(A->val[p]);
        if (diag[i] > 0.0) { double r = sum / diag[i]; if (r > lam) lam = r; }
    }
    return lam;
}

/* Chebyshev-accelerated Jacobi (extension, see smg_oracle.h): one polynomial of degree iters + 1 */
static void relax_chebyshev(orc_mg *mg, int lv, const double *B, int k, int iters, double *u, double frac)
{
    if (iters <= 0) return;
    const orc_csc *A = &mg->lv[lv].A;
    const double *diag = mg->lv[lv].A_diag;
    int n = A->n_rows;
    const double lam = orc_spectral_bound(mg, lv);
    const double lmax = lam, lmin = lam * frac;
    const double theta = (lmax + lmin) / 2.0, delta = (lmax - lmin) / 2.0;
    const double sigma = theta / delta;
    double *tmp = (double *)xmalloc((size_t)n * sizeof(double));
    double *d = (double *)xcalloc((size_t)n, sizeof(double));
    for (int ri = 0; ri < k; ri++) {
        double *uc = u + (size_t)ri * (size_t)n;
        const double *bc = B + (size_t)ri * (size_t)n;
        double rho = 1.0 / sigma;
        for (int s = 0; s <= iters; s++) {
            double c1 = 0.0, c2 = 1.0 / theta;
            if (s > 0) {
                const double rho_new = 1.0 / (2.0 * sigma - rho);
                c1 = rho_new * rho;
                c2 = 2.0 * rho_new / delta;
                rho = rho_new;
            }
#pragma omp parallel for schedule(static) if (mg->par && n > ORC_PAR_MIN_ROWS)
            for (int colIdx = 0; colIdx < n; colIdx++) {
                double sum = 0;
                for (int p = A->colptr[colIdx]; p < A->colptr[colIdx + 1]; p++)
                    if (A->rowidx[p] != colIdx) sum += A->val[p] * uc[A->rowidx[p]];
                double t = (bc[colIdx] - sum) / diag[colIdx];
                double r = t - uc[colIdx];
                double dn = (s == 0) ? c2 * r : c1 * d[colIdx] + c2 * r;
                d[colIdx] = dn;
                tmp[colIdx] = uc[colIdx] + dn;
            }
            memcpy(uc, tmp, (size_t)n * sizeof(double));
        }
    }
    free(tmp); free(d);
}

int orc_set_smoother(orc_mg *mg, int lv, int kind, double omega)
{
    if (!mg || lv < 0 || lv >= mg->n_levels || (kind != ORC_SMOOTH_GS && kind != ORC_SMOOTH_JACOBI && kind != ORC_SMOOTH_CHEBY)) return -1;
    if (!mg->smoother) {
        mg->smoother = (int *)xcalloc((size_t)mg->n_levels, sizeof(int));
        mg->omega = (double *)xcalloc((size_t)mg->n_levels, sizeof(double));
    }
    mg->smoother[lv] = kind;
    mg->omega[lv] = omega;
    return 0;
}

void orc_relax(orc_mg *mg, int lv, const double *B, int k, int iters, double *u)
{
    double t0 = now_s();
    const orc_csc *A = &mg->lv[lv].A;
    const double *diag = mg->lv[lv].A_diag;
    int n = A->n_rows;
    if (mg->smoother && mg->smoother[lv] != ORC_SMOOTH_GS) {
        if (mg->smoother[lv] == ORC_SMOOTH_JACOBI) relax_jacobi(mg, lv, B, k, iters, u, mg->omega[lv]);
        else relax_chebyshev(mg, lv, B, k, iters, u, mg->omega[lv]);
        mg->t_relax += now_s() - t0; mg->c_relax++;
        return;
    }
    if (mg->par && mg->color_ptr && mg->color_ptr[lv]) {
        /* all-core mode: the same sweep, block by block; rows of a block do not reference each other, so sweeping a block
         * in parallel reads and writes exactly what the sequential loop below does */
        const int *cp = mg->color_ptr[lv];
        for (int iter = 0; iter < iters; iter++)
            for (int ri = 0; ri < k; ri++) {
                double *uc = u + (size_t)ri * (size_t)n;
                const double *bc = B + (size_t)ri * (size_t)n;
                for (int c = 0; c < mg->n_colors[lv]; c++) {
#pragma omp parallel for schedule(static) if (cp[c + 1] - cp[c] > ORC_PAR_MIN_ROWS)
                    for (int colIdx = cp[c]; colIdx < cp[c + 1]; colIdx++) {
                        double sum = 0;
                        for (int p = A->colptr[colIdx]; p < A->colptr[colIdx + 1]; p++)
                            if (A->rowidx[p] != colIdx) sum += A->val[p] * uc[A->rowidx[p]];
                        uc[colIdx] = (bc[colIdx] - sum) / diag[colIdx];
                    }
                }
            }
        mg->t_relax += now_s() - t0; mg->c_relax++;
        return;
    }
    for (int iter = 0; iter < iters; iter++)
        for (int ri = 0; ri < k; ri++) {
            double *uc = u + (size_t)ri * (size_t)n;
            const double *bc = B + (size_t)ri * (size_t)n;
            for (int colIdx = 0; colIdx < n; colIdx++) {
                double sum = 0;
                for (int p = A->colptr[colIdx]; p < A->colptr[colIdx + 1]; p++)
                    if (A->rowidx[p] != colIdx) sum += A->val[p] * uc[A->rowidx[p]];
                uc[colIdx] = (bc[colIdx] - sum) / diag[colIdx];
            }
        }
    mg->t_relax += now_s() - t0; mg->c_relax++;   /* PROFC_NODE("MG: relaxation"), :121 */
}

/* mg_VCycle.cpp:181-201   delta_u = solver.solve(B); u = u + delta_u  (no residual formed) */
void orc_coarse_solve(orc_mg *mg, int lv, const double *B, int k, double *u)
{
    int n = mg->lv[lv].A.n_rows;
    double *delta = (double *)xmalloc((size_t)n * sizeof(double));
    for (int c = 0; c < k; c++) {
        ldlt_solve(&mg->solver, B + (size_t)c * (size_t)n, delta);
        double *uc = u + (size_t)c * (size_t)n;
        for (int i = 0; i < n; i++) uc[i] = uc[i] + delta[i];
    }
    free(delta);
}

/* mg_VCycle.cpp:3-59 */
void orc_vcycle(orc_mg *mg, const double *B, int pre, int post, int lv, double *u, int k)
{
    if (lv == mg->n_levels - 1) {                      /* :28-33 */
        orc_coarse_solve(mg, lv, B, k, u);
        return;
    }
    int n = mg->lv[lv].A.n_rows;
    int nc = mg->lv[lv + 1].PT.n_rows;
    orc_relax(mg, lv, B, k, pre, u);                   /* :36 */
    double *Au = (double *)xmalloc((size_t)n * (size_t)k * sizeof(double));
    orc_A(mg, lv, u, k, Au);                           /* :40-41 */
    double *r = Au;                                    /* r = B - Au, :42 */
#pragma omp parallel for schedule(static) if (mg->par && n > ORC_PAR_MIN_ROWS)
    for (long t = 0; t < (long)n * (long)k; t++) r[t] = B[t] - Au[t];
    double *rc = (double *)xmalloc((size_t)nc * (size_t)k * sizeof(double));
    orc_restrict(mg, lv, r, k, rc);                    /* :43-44 */
    double *uc = (double *)xcalloc((size_t)nc * (size_t)k, sizeof(double)); /* :46-47 */
    orc_vcycle(mg, rc, pre, post, lv + 1, uc, k);      /* :48 */
    double *puc = r;                                   /* reuse the buffer */
    orc_prolong(mg, lv, uc, k, puc);                   /* :51-52 */
#pragma omp parallel for schedule(static) if (mg->par && n > ORC_PAR_MIN_ROWS)
    for (long t = 0; t < (long)n * (long)k; t++) u[t] = u[t] + puc[t]; /* :53 */
    free(rc); free(uc); free(Au);
    orc_relax(mg, lv, B, k, post, u);                  /* :57 */
}

/* Frobenius norm of RHS - A0*z  (Eigen .norm(): sqrt of the plain sum of squares,
 * accumulated column-major); min_quad_with_fixed_mg.cpp:110 / :332 */
static double residual_norm(const orc_mg *mg, const double *RHS, const double *z, int k, double *tmp)
{
    int n = mg->lv[0].A.n_rows;
    orc_A(mg, 0, z, k, tmp);
    double ss = 0.0;
    if (mg->par) {   /* all-core mode: a parallel reduction (summation order differs from Eigen's in the last digits) */
#pragma omp parallel for schedule(static) reduction(+ : ss)
        for (long t = 0; t < (long)n * (long)k; t++) {
            double d = RHS[t] - tmp[t];
            ss += d * d;
        }
        return sqrt(ss);
    }
    for (size_t t = 0; t < (size_t)n * (size_t)k; t++) {
        double d = RHS[t] - tmp[t];
        ss += d * d;
    }
    return sqrt(ss);
}

/* the loop shared by both solve overloads (.cpp:105-134 / :326-360); RHS, z: n0 x k, ld = n0 */
static int solve_loop(orc_mg *mg, const double *RHS, double *z, int k, double tol, int max_iter,
                      double *r_his, int *n_his)
{
    int n = mg->lv[0].A.n_rows;
    double *tmp = (double *)xmalloc((size_t)n * (size_t)k * sizeof(double));
    double residual = 0.0;
    int pre = 2, post = 2;                              /* :102-103 / :324-325 */
    int cnt = 0;
    for (int iter = 0; iter < max_iter; iter++) {
        residual = residual_norm(mg, RHS, z, k, tmp);   /* :110 */
        if (mg->verbose) printf("MG iteration: %d, residual: %g\n", iter, residual); /* :111 */
        r_his[cnt++] = residual;                        /* :112 */
        if (residual < tol) break;                      /* :113-116 */
        double t0 = now_s();
        orc_vcycle(mg, RHS, pre, post, 0, z, k);        /* :124 */
        mg->t_vcycle += now_s() - t0; mg->c_vcycle++;   /* PROFC_NODE("MG: total VCycle"), :123 */
    }
    if (mg->verbose && cnt) printf("residual norm: %g\n", r_his[cnt - 1]); /* :127 */
    free(tmp);
    *n_his = cnt;
    return (residual > tol) ? 0 : 1;                    /* :131-134 */
}

/* a level-0 vector copied into a fresh buffer; all-core mode: by the threads that sweep it, block by block (first touch, see csc_place) */
static void copy_level0(const orc_mg *mg, double *dst, const double *src, int n)
{
#ifdef _OPENMP
    if (mg->par && n > ORC_PAR_MIN_ROWS) {
        const int one[2] = {0, n};
        const int *cp = (mg->color_ptr && mg->color_ptr[0]) ? mg->color_ptr[0] : one;
        const int nc = (mg->color_ptr && mg->color_ptr[0]) ? mg->n_colors[0] : 1;
        for (int c = 0; c < nc; c++) {
#pragma omp parallel for schedule(static)
            for (int i = cp[c]; i < cp[c + 1]; i++) dst[i] = src[i];
        }
        return;
    }
#endif
    memcpy(dst, src, (size_t)n * sizeof(double));
}

/* min_quad_with_fixed_mg.cpp:80-135 */
int orc_solve(orc_mg *mg, const double *RHS, int ld_rhs, const double *z0, int ld_z0, int k,
              double tol, int max_iter, double *z, int ld_z, double *r_his, int *n_his)
{
    int n = mg->lv[0].A.n_rows;
    double *rhs = (double *)xmalloc((size_t)n * (size_t)k * sizeof(double));
    double *zz = (double *)xmalloc((size_t)n * (size_t)k * sizeof(double));
    for (int c = 0; c < k; c++) {
        copy_level0(mg, rhs + (size_t)c * n, RHS + (size_t)c * ld_rhs, n);
        copy_level0(mg, zz + (size_t)c * n, z0 + (size_t)c * ld_z0, n); /* z = z0, :97 */
    }
    int conv = solve_loop(mg, rhs, zz, k, tol, max_iter, r_his, n_his);
    for (int c = 0; c < k; c++) memcpy(z + (size_t)c * ld_z, zz + (size_t)c * n, (size_t)n * sizeof(double));
    free(rhs); free(zz);
    return conv;
}

/* min_quad_with_fixed_mg.cpp:288-361 */
int orc_solve_known(orc_mg *mg, const double *RHS, int ld_rhs, const double *known_val, int ld_kv,
                    const double *z0, int ld_z0, int k, double tol, int max_iter,
                    double *z, int ld_z, double *r_his, int *n_his)
{
    const orc_mqwf_data *d = &mg->data;
    int nu = d->n_unknown, nk = d->n_known;
    double *zu = (double *)xmalloc((size_t)nu * (size_t)k * sizeof(double));
    double *rhs = (double *)xmalloc((size_t)nu * (size_t)k * sizeof(double));
    double *akv = (double *)xmalloc((size_t)nu * (size_t)k * sizeof(double));
    double *kv = (double *)xcalloc((size_t)(nk > 0 ? nk : 1) * (size_t)k, sizeof(double));
    for (int c = 0; c < k; c++) {
        for (int i = 0; i < nu; i++) {
            zu[(size_t)c * nu + i] = z0[(size_t)c * ld_z0 + d->unknown[i]];     /* :310-311 */
            rhs[(size_t)c * nu + i] = RHS[(size_t)c * ld_rhs + d->unknown[i]];  /* :316-317 */
        }
        for (int i = 0; i < nk; i++) kv[(size_t)c * nk + i] = known_val[(size_t)c * ld_kv + i];
    }
    orc_csc_times_dense(&d->Auk, kv, nk, k, akv, nu);
    for (size_t t = 0; t < (size_t)nu * (size_t)k; t++) rhs[t] = rhs[t] - akv[t]; /* :318 */
    int conv = solve_loop(mg, rhs, zu, k, tol, max_iter, r_his, n_his);
    for (int c = 0; c < k; c++) {
        for (int i = 0; i < nu; i++) z[(size_t)c * ld_z + d->unknown[i]] = zu[(size_t)c * nu + i]; /* :354 */
        for (int i = 0; i < nk; i++) z[(size_t)c * ld_z + d->known[i]] = kv[(size_t)c * nk + i];   /* :355 */
    }
    free(zu); free(rhs); free(akv); free(kv);
    return conv;
}

/* ------------------------------------------------------------------ introspection */

int orc_level_rows(const orc_mg *mg, int lv) { return mg->lv[lv].A.n_rows; }
const orc_csc *orc_level_A(const orc_mg *mg, int lv) { return &mg->lv[lv].A; }
const orc_csc *orc_level_P(const orc_mg *mg, int lv) { return &mg->lv[lv].P; }
const orc_csc *orc_level_PT(const orc_mg *mg, int lv) { return &mg->lv[lv].PT; }
const double *orc_level_Adiag(const orc_mg *mg, int lv) { return mg->lv[lv].A_diag; }
const orc_csc *orc_data_LHS(const orc_mg *mg) { return &mg->data.LHS; }
const orc_csc *orc_data_Auk(const orc_mg *mg) { return &mg->data.Auk; }
int orc_data_unknown(const orc_mg *mg, const int **idx) { *idx = mg->data.unknown; return mg->data.n_unknown; }
void orc_profile(const orc_mg *mg, double *t_relax, long *c_relax, double *t_vcycle, long *c_vcycle)
{
    *t_relax = mg->t_relax; *c_relax = mg->c_relax; *t_vcycle = mg->t_vcycle; *c_vcycle = mg->c_vcycle;
}
/* ---- all-core mode (see header) */
int orc_set_parallel(orc_mg *mg, int lv, int n_colors, const int *color_ptr)
{
    if (!mg || lv < 0 || lv >= mg->n_levels || n_colors < 1 || !color_ptr) return -1;
    if (!mg->color_ptr) {
        mg->color_ptr = (int **)xcalloc((size_t)mg->n_levels, sizeof(int *));
        mg->n_colors = (int *)xcalloc((size_t)mg->n_levels, sizeof(int));
        mg->AT = (orc_csc *)xcalloc((size_t)mg->n_levels, sizeof(orc_csc));
    }
    const orc_csc *A = &mg->lv[lv].A;
    if (color_ptr[0] != 0 || color_ptr[n_colors] != A->n_rows) return -1;
    /* the blocks must really be independent sets, or the parallel sweep would not be the sequential one */
    for (int c = 0; c < n_colors; c++)
        for (int j = color_ptr[c]; j < color_ptr[c + 1]; j++)
            for (int p = A->colptr[j]; p < A->colptr[j + 1]; p++) {
                int i = A->rowidx[p];
                if (i != j && i >= color_ptr[c] && i < color_ptr[c + 1]) return -2;
            }
    free(mg->color_ptr[lv]);
    mg->color_ptr[lv] = (int *)xmalloc((size_t)(n_colors + 1) * sizeof(int));
    memcpy(mg->color_ptr[lv], color_ptr, (size_t)(n_colors + 1) * sizeof(int));
    mg->n_colors[lv] = n_colors;
    orc_csc_free(&mg->AT[lv]);
    csc_transpose(&mg->AT[lv], A);
    return 0;
}
/* all-core mode only: the arrays of a matrix re-allocated and copied by the threads that will stream them -- the same loop
 * structure and static schedule as the sweeps / products (cp: the nc + 1 offsets of the blocks the columns are visited by; one
 * block = all columns) -- so that on a multi-socket host every thread's share of the matrix lives in its own socket's memory
 * (first touch).  Built by one thread the whole hierarchy sits on one NUMA node and every other socket reads it over the fabric:
 * the thread sweep of bench.py's cpu_allcore leg got SLOWER beyond 16 threads on a 2 x 64-core host.  Values are untouched. */
static void csc_place(orc_csc *M, const int *cp, int nc)
{
#ifdef _OPENMP
    if (!M->colptr || M->n_cols <= ORC_PAR_MIN_ROWS) return;
    const int one[2] = {0, M->n_cols};
    if (!cp) { cp = one; nc = 1; }
    const long nnz = M->colptr[M->n_cols];
    int *ri = (int *)xmalloc((size_t)nnz * sizeof(int));
    double *va = (double *)xmalloc((size_t)nnz * sizeof(double));
    int *cptr = (int *)xmalloc((size_t)(M->n_cols + 1) * sizeof(int));
    for (int c = 0; c < nc; c++) {
#pragma omp parallel for schedule(static)
        for (int j = cp[c]; j < cp[c + 1]; j++) {
            cptr[j] = M->colptr[j];
            for (int p = M->colptr[j]; p < M->colptr[j + 1]; p++) { ri[p] = M->rowidx[p]; va[p] = M->val[p]; }
        }
    }
    cptr[M->n_cols] = M->colptr[M->n_cols];
    free(M->colptr); free(M->rowidx); free(M->val);
    M->colptr = cptr; M->rowidx = ri; M->val = va;
#else
    (void)M; (void)cp; (void)nc;
#endif
}

int orc_enable_parallel(orc_mg *mg, int on, int threads)
{
    mg->par = on ? 1 : 0;
    if (on) for (int lv = 0; lv < mg->n_levels; lv++) {   /* every level's A needs its row-wise image */
        if (!mg->AT) mg->AT = (orc_csc *)xcalloc((size_t)mg->n_levels, sizeof(orc_csc));
        if (!mg->AT[lv].colptr && mg->lv[lv].A.colptr) csc_transpose(&mg->AT[lv], &mg->lv[lv].A);
    }
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
    if (on) for (int lv = 0; lv < mg->n_levels; lv++) {
        /* (again at every call: the static schedule's shares depend on the thread count) */
        const int *cp = (mg->color_ptr && mg->color_ptr[lv]) ? mg->color_ptr[lv] : NULL;
        csc_place(&mg->lv[lv].A, cp, cp ? mg->n_colors[lv] : 0);   /* swept block by block (orc_relax) */
        csc_place(&mg->AT[lv], NULL, 0);                            /* row-wise products (orc_A) */
        csc_place(&mg->lv[lv].P, NULL, 0);
        csc_place(&mg->lv[lv].PT, NULL, 0);
        if (mg->lv[lv].A_diag && mg->lv[lv].A.n_cols > ORC_PAR_MIN_ROWS) {
            const int n = mg->lv[lv].A.n_cols;
            double *d = (double *)xmalloc((size_t)n * sizeof(double));
            const int one[2] = {0, n};
            const int *bp = cp ? cp : one;
            const int nb = cp ? mg->n_colors[lv] : 1;
            for (int c = 0; c < nb; c++) {
#pragma omp parallel for schedule(static)
                for (int j = bp[c]; j < bp[c + 1]; j++) d[j] = mg->lv[lv].A_diag[j];
            }
            free(mg->lv[lv].A_diag);
            mg->lv[lv].A_diag = d;
        }
    }
    return on ? omp_get_max_threads() : 1;
#else
    (void)threads;
    return 1;
#endif
}

void orc_profile_reset(orc_mg *mg) { mg->t_relax = mg->t_vcycle = 0.0; mg->c_relax = mg->c_vcycle = 0; }
