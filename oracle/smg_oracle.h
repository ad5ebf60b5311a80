/*
 * smg_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference's multigrid solve path
 * (HTDerekLiu/surface_multigrid_code: src/mg_VCycle.cpp, src/min_quad_with_fixed_mg.cpp,
 * src/mg_data.h), written from the published algorithm in the reference's own
 * operation order.  Only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg may link or call anything in oracle/.
 *
 * PARITY UNPINNED: the reference has no tests / golden vectors for this path and it cannot
 * be compiled in the build container (Eigen + libigl are un-vendored and absent), so this
 * oracle is pinned only against an independent scipy restatement (tests/golden/make_golden.py).
 *
 * Storage follows Eigen's default: compressed-column (CSC), int32 indices, fp64 values,
 * dense blocks column-major with an explicit leading dimension.
 */
#ifndef SMG_ORACLE_H
#define SMG_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int n_rows, n_cols;
    int *colptr;   /* n_cols + 1 */
    int *rowidx;   /* nnz, ascending inside each column */
    double *val;   /* nnz */
} orc_csc;

/* one element of std::vector<mg_data>  (reference src/mg_data.h:11-27; V,F and the dead
 * colouring fields S,SV,SVI,SC,SCS are not used by the solve path and are omitted) */
typedef struct {
    orc_csc P_full;  /* mg_data.h:15 */
    orc_csc A;       /* mg_data.h:16 */
    double *A_diag;  /* mg_data.h:17 */
    orc_csc P;       /* mg_data.h:18 */
    orc_csc PT;      /* mg_data.h:19 */
} orc_level;

/* stands in for Eigen::SimplicialLDLT<SparseMatrix<double>> (mg_VCycle.cpp:203):
 * sparse LDL^T with a fill-reducing symmetric permutation.  The reference uses AMD
 * ordering + supernode-free simplicial factorisation; here reverse Cuthill-McKee +
 * skyline (envelope) LDL^T -- same mathematics, different (permutation-dependent) rounding. */
typedef struct {
    int n;
    int *perm;      /* new -> old */
    int *first;     /* first stored column of skyline row i (permuted numbering) */
    long *rowoff;   /* offset of row i in L (entries first[i] .. i-1) */
    double *L;      /* strict lower skyline */
    double *D;      /* diagonal */
    double *work;   /* n doubles */
} orc_ldlt;

/* min_quad_with_fixed_mg_data (min_quad_with_fixed_mg.h:22-29) */
typedef struct {
    int n;
    int n_known, n_unknown;
    int *known, *unknown;
    orc_csc LHS, Auk;
} orc_mqwf_data;

typedef struct {
    int n_levels;
    orc_level *lv;          /* std::vector<mg_data> mg */
    orc_mqwf_data data;     /* caller-owned in the reference; kept in the handle here */
    orc_ldlt solver;        /* caller-owned in the reference; kept in the handle here */
    int verbose;            /* 0: silent (the reference prints unconditionally) */
    /* PROFC_NODE accumulators (profc.h): seconds + counts */
    double t_relax, t_vcycle; long c_relax, c_vcycle;
    /* "fair CPU" comparator, off by default: see orc_set_parallel */
    int par;                /* 1: OpenMP all-core mode */
    int **color_ptr;        /* per level: n_colors + 1 row offsets of the independent blocks, or NULL */
    int *n_colors;
    orc_csc *AT;            /* per level: A^T (the row-wise image of A), built by orc_set_parallel */
    /* per-level smoother selection (extension named by BASELINE.json's north_star, see orc_set_smoother) */
    int *smoother;          /* NULL or per level: ORC_SMOOTH_GS (the reference's relax()) / ORC_SMOOTH_JACOBI */
    double *omega;          /* per level: damping factor of the Jacobi sweep */
} orc_mg;

/* ---- hierarchy container ---- */
orc_mg *orc_mg_create(int n_levels);
void orc_mg_destroy(orc_mg *mg);
/* what mg_precompute stores per level (mg_precompute.cpp:71-77): P, PT = P^T, P_full = P.
 * Input is the CSC of P (#fine x #coarse) for level lv >= 1. */
int orc_mg_set_prolong(orc_mg *mg, int lv, int n_rows, int n_cols,
                       const int *colptr, const int *rowidx, const double *val);

/* ---- min_quad_with_fixed_mg_precompute ---- */
/* no constraints: min_quad_with_fixed_mg.cpp:3-51 */
int orc_precompute(orc_mg *mg, int n, const int *colptr, const int *rowidx, const double *val);
/* with `known`:   min_quad_with_fixed_mg.cpp:137-257 */
int orc_precompute_known(orc_mg *mg, int n, const int *colptr, const int *rowidx,
                         const double *val, const int *known, int n_known);

/* ---- min_quad_with_fixed_mg_solve ---- */
/* no constraints: min_quad_with_fixed_mg.cpp:80-135.  Dense blocks column-major.
 * Returns 1 if converged (!(residual > tol)), 0 otherwise.  r_his must hold max_iter doubles. */
int orc_solve(orc_mg *mg, const double *RHS, int ld_rhs, const double *z0, int ld_z0, int k,
              double tol, int max_iter, double *z, int ld_z, double *r_his, int *n_his);
/* with known_val: min_quad_with_fixed_mg.cpp:288-361.  RHS, z0, z are n x k (full size),
 * known_val is n_known x k. */
int orc_solve_known(orc_mg *mg, const double *RHS, int ld_rhs, const double *known_val, int ld_kv,
                    const double *z0, int ld_z0, int k, double tol, int max_iter,
                    double *z, int ld_z, double *r_his, int *n_his);

/* ---- mg_VCycle.cpp pieces (all dense blocks column-major, ld = rows of that level) ---- */
void orc_vcycle(orc_mg *mg, const double *B, int pre, int post, int lv, double *u, int k); /* :3-59 */
void orc_A(const orc_mg *mg, int lv, const double *u, int k, double *Au);        /* :62-70  */
void orc_restrict(const orc_mg *mg, int lv, const double *x, int k, double *Rx); /* :72-81  */
void orc_prolong(const orc_mg *mg, int lv, const double *x, int k, double *Px);  /* :83-92  */
void orc_relax(orc_mg *mg, int lv, const double *B, int k, int iters, double *u); /* :113-178 */
void orc_coarse_solve(orc_mg *mg, int lv, const double *B, int k, double *u);    /* :181-201 */

/* ---- damped-Jacobi smoother: NOT in the reference (its relax() is Gauss-Seidel only, mg_VCycle.cpp:113-178); it is the
 * "Gauss-Seidel/Jacobi smoothing" of BASELINE.json's north_star and fills the same slot.  Defined so that the HIP kernel and
 * this loop agree bit for bit.  One sweep, for every row i from the OLD iterate (all rows simultaneously):
 *     s = sum_{j != i, ascending j} A(j,i) * u_old[j]      (column i of the CSC matrix, like relax(), :149-155)
 *     t = (B[i] - s) / A_diag[i]
 *     u_new[i] = u_old[i] + omega * (t - u_old[i])
 * kind: ORC_SMOOTH_GS (default on every level) or ORC_SMOOTH_JACOBI.  orc_relax() dispatches on the level's setting. */
enum { ORC_SMOOTH_GS = 0, ORC_SMOOTH_JACOBI = 1, ORC_SMOOTH_CHEBY = 2 };
int orc_set_smoother(orc_mg *mg, int lv, int kind, double omega);
/* ORC_SMOOTH_CHEBY (extension, same slot): Chebyshev-accelerated Jacobi; `omega` is the interval fraction.  relax(iters) applies ONE
 * polynomial of degree iters + 1 in D^-1 A, optimal on [fraction * lam, lam], lam = orc_spectral_bound(lv):
 *   theta = (lam + fraction lam) / 2, delta = (lam - fraction lam) / 2, sigma = theta / delta, rho = 1 / sigma
 *   step s:  r_i = (B[i] - sum_{j != i, ascending j} A(j,i) u[j]) / A_diag[i] - u[i]          (all rows from the old u)
 *            s = 0: d = (1 / theta) r;   s >= 1: rho' = 1 / (2 sigma - rho), d = (rho' rho) d + (2 rho' / delta) r, rho = rho'
 *            u_new = u + d
 * lam = max_i (sum_j |A(j,i)|, ascending j) / A_diag[i]: the Gershgorin bound of the spectrum of D^-1 A. */
double orc_spectral_bound(const orc_mg *mg, int lv);

/* ---- introspection for tests ---- */
int orc_level_rows(const orc_mg *mg, int lv);
const orc_csc *orc_level_A(const orc_mg *mg, int lv);
const orc_csc *orc_level_P(const orc_mg *mg, int lv);
const orc_csc *orc_level_PT(const orc_mg *mg, int lv);
const double *orc_level_Adiag(const orc_mg *mg, int lv);
const orc_csc *orc_data_LHS(const orc_mg *mg);
const orc_csc *orc_data_Auk(const orc_mg *mg);
int orc_data_unknown(const orc_mg *mg, const int **idx);
void orc_profile(const orc_mg *mg, double *t_relax, long *c_relax, double *t_vcycle, long *c_vcycle);

/* ---- all-core mode: NOT a restatement of the reference (whose solve is single-threaded), but the comparator SURVEY.md
 * section 8d asks for beside it.  The caller hands in, per smoothed level, a partition of the rows into contiguous blocks
 * whose rows are mutually independent (a multi-colouring in whose order the system is numbered -- then the reference's
 * lexicographic sweep IS the multi-colour sweep and each block can be swept in parallel without changing a bit).  Sparse
 * products run row-wise over the stored transposes (same ascending-column accumulation per output element, same bits);
 * only the residual norm (parallel reduction) may differ in the last digits.  Call after orc_precompute*.  threads <= 0:
 * the OpenMP default.  Returns the number of threads in use (1 if built without OpenMP). */
int orc_set_parallel(orc_mg *mg, int lv, int n_colors, const int *color_ptr);
int orc_enable_parallel(orc_mg *mg, int on, int threads);
void orc_profile_reset(orc_mg *mg);

/* ---- generic sparse helpers (Eigen / libigl semantics), exported for tests ---- */
void orc_csc_free(orc_csc *m);
void orc_csc_times_dense(const orc_csc *A, const double *X, int ldx, int k, double *Y, int ldy);

#ifdef __cplusplus
}
#endif
#endif
