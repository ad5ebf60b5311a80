// smg_device.hpp -- launch interface of the hand-written gfx950 kernels (smg_device.hip).
//
// All dense blocks on the device use the INTERNAL layout: row-major n x k (the k right-hand-side columns
// of one vertex are contiguous, so every neighbour gather is one k*8-byte segment), rows in the level's
// colour-major internal numbering (smg_order.hpp).
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>

namespace smg {

// Solve-loop control block, resident in HBM.  The outer loop of min_quad_with_fixed_mg_solve
// (reference src/min_quad_with_fixed_mg.cpp:108-125) is enqueued without host round trips: the decide
// kernel appends to r_his and raises `done`; every later kernel of the stream starts with `if (done) return`.
struct Ctrl {
    int done;      // 1 once residual < tol (or non-finite) was observed
    int n_his;     // entries of r_his written
    int status;    // 0 ok, -1 non-finite residual
    int just_done; // set by the speculative decide when this very decision ended the loop
    double sumsq;  // sum of squares of the last residual (all-reduced across ranks when column-sharded)
    double tol;    // absolute tolerance of the break test (kept here so captured graphs do not bake it in)
    double* r_his; // residual history in HBM, his_cap entries (sized from max_iter at smg_solve_begin; read through this
    int his_cap;   // pointer at run time, so captured graphs survive a re-allocation)
    int pad_;
    double r_last, r_prev;   // the two most recent residuals (what the host's adaptive polling extrapolates from)
};

struct SellDev {
    int n_rows = 0, n_cols = 0, n_slices = 0;
    int C = 64;                      // slice height: 64 (one row per lane) or 128 (two adjacent rows per lane)
    const int* slice_row = nullptr;  // n_slices + 1
    const int* slice_off = nullptr;  // n_slices + 1 (units of C entries); = s * stride when stride > 0
    const int* slice_w = nullptr;    // n_slices: panel columns used by the slice
    int stride = 0;                  // > 0: fixed panel pitch, addressing needs no table
    int w_lo = 0;                    // columns requested before the slice's width is known, <= stride (0 when stride == 0)
    int w_max = 0;                   // widest slice (panel columns)
    const int* order = nullptr;      // optional launch order of the slices (region-major), whole-matrix kernels only
    const int* col = nullptr;
    const double* val = nullptr;
    const float* valf = nullptr;     // fp32 copy of val (same slots), only for the mixed-precision V-cycle
    // Long rows taken out of the panels (restriction operators of decimated hierarchies: a coarse vertex of the reference's construction
    // may gather from > 100 fine ones, and a panel row is one chain of dependent batches -- it set the launch's duration): in CSR,
    // ascending column order, served by launch_sell(SELL_AX) through a companion launch of one wave per (row, column); the panels hold
    // these rows empty.  Same products, same order of additions: bit-identical.
    int long_n = 0;
    const int* long_row = nullptr;   // long_n row numbers
    const int* long_ptr = nullptr;   // long_n + 1
    const int* long_col = nullptr;
    const double* long_val = nullptr;
    const float* long_valf = nullptr;
};

enum SellMode {
    SELL_AX = 0,        // y = A x                       (mg_VCycle.cpp:69 `A`, :91 `prolong`, :80 `restrict`)
    SELL_RESID = 1,     // y = b - A x                   (mg_VCycle.cpp:41-42)
    SELL_RESID_SS = 2,  // partial sums of |b - A x|^2   (min_quad_with_fixed_mg.cpp:110 / :332)
    SELL_ADD = 3,       // y = y + A x                   (mg_VCycle.cpp:51-53  u = u + P uc)
    SELL_GS = 4,        // y_i = (b_i - sum_{j != i} A_ij y_j) / A_ii on the given slice range (one colour)
    SELL_RESID_BOTH = 5,  // y = b - A x AND the partial sums of its squares (outer loop of the mixed-precision mode)
    SELL_JACOBI = 6,    // y_i = x_i + omega * ((b_i - sum_{j != i} A_ij x_j) / A_ii - x_i): one damped-Jacobi sweep from x into y (x != y),
                        // whole matrix in one launch (the "Jacobi" of BASELINE.json's north_star; same slot as SELL_GS, mg_VCycle.cpp:113-178)
    SELL_CHEBY = 7,     // one step of Chebyshev-accelerated Jacobi from x into y (x != y):  r_i = (b_i - sum_{j != i} A_ij x_j) / A_ii - x_i,
                        // d_i = c1 d_i + c2 r_i (first step, c1 == 0: d_i = c2 r_i, d not read),  y_i = x_i + d_i
    // ---- the outer loop's residual folded into the first smoothing launches of the V-cycle (level 0 only, fp64 only) ----
    // The reference computes |RHS - A z| (min_quad_with_fixed_mg.cpp:110), tests it, and then starts the cycle with relax() on the same
    // z: the first sweep streams the same matrix rows again.  These modes take the squared residual of the OLD iterate out of that
    // sweep -- per row a second accumulator, b_i - sum_j A_ij x_j over ALL stored entries in slot order, bit for bit what SELL_RESID_SS
    // forms -- and write the sweep's result OUT OF PLACE (y != x), so that the old iterate survives when the break test says stop.
    SELL_GS_OOP = 8,      // one colour of a Gauss-Seidel sweep from x into y: neighbours in earlier colours (rows below the first row of
                          // the launch's slice range: colour-major numbering) are read from y, the others from x.  Same values, same order
                          // as the in-place launch.
    SELL_GS_HEAD = 9,     // SELL_GS_OOP + partial sums of |b - A x|^2 over the rows of the colour
    SELL_JACOBI_HEAD = 10,  // SELL_JACOBI + partial sums of |b - A x|^2
    SELL_CHEBY_HEAD = 11,   // SELL_CHEBY + partial sums of |b - A x|^2
};
constexpr bool sell_is_gs(int m) { return m == SELL_GS || m == SELL_GS_OOP || m == SELL_GS_HEAD; }
constexpr bool sell_is_oop(int m) { return m == SELL_GS_OOP || m == SELL_GS_HEAD; }
constexpr bool sell_is_jacobi(int m) { return m == SELL_JACOBI || m == SELL_JACOBI_HEAD; }
constexpr bool sell_is_cheby(int m) { return m == SELL_CHEBY || m == SELL_CHEBY_HEAD; }
constexpr bool sell_is_head(int m) { return m == SELL_GS_HEAD || m == SELL_JACOBI_HEAD || m == SELL_CHEBY_HEAD; }
constexpr bool sell_has_ss(int m) { return sell_is_head(m) || m == SELL_RESID_SS || m == SELL_RESID_BOTH; }

// SELL_ADD: y = b + A x, where b is the iterate the correction is added to (b == nullptr: in place, b = y).
// y/x/b: internal layout, ld = number of columns k.  Slices [s_begin, s_end).  `ctrl` may be null (no
// early-exit test).  For SELL_RESID_SS, `partials` receives one double per launched block; the number of
// blocks is returned through *n_blocks.
// zero_rows (SELL_AX only, optional): an n_rows x k block that is set to +0.0 row by row alongside y (the restriction
// launch also performs `uc.setZero()`, mg_VCycle.cpp:46-47).
// first (with zero_rows, optional): the rows of the coarse level's first colour get, instead of +0.0, what the first colour
// launch of the first pre-smoothing sweep would leave there -- y_i / a_ii: with a zero initial guess every product of that
// launch is a_ij * 0, so (b_i - 0) / a_ii is bit for bit the sweep's value, and the cycle skips that launch.
struct FirstColour {
    const int* diag_slot = nullptr;   // per row of the first colour: slot of the diagonal in the sweep's SELL value array
    int n_first = 0;                  // rows of the first colour (they lead the colour-major numbering)
    const double* val = nullptr;      // the sweep's SELL values (A or A^T) ...
    const float* valf = nullptr;      // ... and their fp32 image
    int jacobi = 0;                   // 1: the coarse level is smoothed by damped Jacobi: n_first = all its rows, and they receive the
    double omega = 1.0;               //    first sweep from u = 0:  0 + omega * (y_i / a_ii - 0).  2: Chebyshev-Jacobi: d_i = omega * (y_i /
                                      //    a_ii - 0), u_i = 0 + d_i, both written (d / df below)
    double* d = nullptr;              // SELL_CHEBY launches and jacobi == 2: the update vector (n x k, internal layout) ...
    float* df = nullptr;              // ... its fp32 twin
    double c1 = 0.0;                  // SELL_CHEBY: coefficient of the old update (0 = first step, d is not read); `omega` is c2
};
hipError_t launch_sell(SellMode mode, const SellDev& A, int s_begin, int s_end, const double* x, const double* b,
                       double* y, int k, const Ctrl* ctrl, double* partials, int* n_blocks, hipStream_t st,
                       double* zero_rows = nullptr, const FirstColour* first = nullptr, double omega = 1.0);
hipError_t launch_sell_f32(SellMode mode, const SellDev& A, int s_begin, int s_end, const float* x, const float* b,
                           float* y, int k, const Ctrl* ctrl, hipStream_t st, float* zero_rows = nullptr,
                           const FirstColour* first = nullptr, double omega = 1.0);
// *out (device double) = max_i (sum_j |a_ij|) / a_ii over the rows of A: the Gershgorin bound of the spectrum of D^-1 A
hipError_t launch_gershgorin(const SellDev& A, double* out, hipStream_t st);
// ---- block (3 degrees of freedom per vertex) matrices: smg_bsr3.hpp (layout), smg_bsr3_device.hip (kernels) ------------------------
struct Bsr3Dev {
    int n_vert = 0, n_slices = 0, w_max = 0;
    const int* slice_row = nullptr;  // n_slices + 1 vertex offsets
    const int* slice_off = nullptr;  // n_slices + 1 panel-column offsets
    const int* slice_w = nullptr;    // n_slices
    const int* order = nullptr;      // optional region-major launch order, whole-matrix launches only
    const int* col = nullptr;        // block columns (vertices), -1 = padding
    const double* val = nullptr;     // nine planes per panel column
    const float* valf = nullptr;     // ... their fp32 image (mixed-precision cycle), or null
};
// modes SELL_AX, _RESID, _RESID_SS, _GS (one vertex colour = slices [s_begin, s_end), in place: y == x), _JACOBI, _CHEBY with the meaning
// they have per scalar row of the 3n x 3n matrix.  x / b / y / dvec: row-major (3 n_vert) x k.  omega, c1, dvec: as in launch_sell
// (Jacobi damping; Chebyshev: omega = c2, c1, the update vector).  partials / n_blocks: SELL_RESID_SS, one double per launched block.
hipError_t launch_bsr3(SellMode mode, const Bsr3Dev& A, int s_begin, int s_end, const double* x, const double* b, double* y, int k, const Ctrl* ctrl,
                       double* partials, int* n_blocks, hipStream_t st, double omega = 1.0, double c1 = 0.0, double* dvec = nullptr);
// the same on the fp32 image (SELL_AX, _RESID, _GS, _JACOBI, _CHEBY); SELL_RESID_BOTH (fp64 only, above): y = b - A x and the partial sums of its squares
hipError_t launch_bsr3_f32(SellMode mode, const Bsr3Dev& A, int s_begin, int s_end, const float* x, const float* b, float* y, int k, const Ctrl* ctrl, hipStream_t st,
                           double omega = 1.0, double c1 = 0.0, float* dvec = nullptr);
hipError_t launch_bsr3_gershgorin(const Bsr3Dev& A, double* out, hipStream_t st);
int bsr3_blocks(int n_slices);
// the image B (laid out on the host, Bsr3Buf::upload of a layout; panel_cols = its panel columns) filled on the device from the scalar CSR arrays of A
// in the caller's numbering, the pattern of its blocks (gptr / gcol) and the vertex numbering (perm new -> old, iperm): B = A(perm3, perm3), or its
// transpose (A structurally symmetric)
hipError_t launch_bsr3_fill(const int* ptr, const int* col, const double* val, const int* gptr, const int* gcol, const int* perm, const int* iperm, const Bsr3Dev& B,
                            size_t panel_cols, bool transposed, hipStream_t st);

// ---- Gauss-Seidel sweep for blocks of 64 right-hand-side columns: block-sequential order (smg_bgs.hpp plan, smg_bgs_device.hip kernel) ----
struct BgsDev {
    int n_blocks = 0, n_colors = 0, xrows = 0;   // xrows: rows of a block's LDS image (own rows + the level's largest rim, a multiple of 128)
    const int* hdr = nullptr;         // per block BGS_HDR ints: first unit, units, batches per row, first entry slot, local rows in use
    const int* xrow = nullptr;        // xrows per block: the row behind local index l (own rows, then the rim)
    const int* ugrow = nullptr;       // 16 per unit: row, ...
    const int* ulrow = nullptr;       //   ... its local index, ...
    const double* udiag = nullptr;    //   ... its diagonal entry
    const int* eidx = nullptr;        // 16 x 8 NB per unit: local index of the entry's column
    const double* eval = nullptr;     //   ... its value
};
// the blocks [b_begin, b_end) -- one block colour -- of one sweep, in place on u (row-major n x k, k a multiple of 16)
hipError_t launch_bgs(const BgsDev& P, int b_begin, int b_end, const double* b, double* u, int k, const Ctrl* ctrl, hipStream_t st);

// ---- Gauss-Seidel sweep of a Galerkin level of a decimated hierarchy: one wavefront per piece, one launch per piece colour (smg_wgs.hpp plan, smg_wgs_device.hip kernel) ----
struct WgsDev {
    int n_pieces = 0, n_colors = 0, rim_pitch = 0, nb_max = 0;   // nb_max: batches of 8 entry slots of the level's widest row
    const int* hdr = nullptr;         // per piece WGS_HDR ints: first entry slot, batches per row, rim rows, phases, ...
    const int* grow = nullptr;        // 64 per piece: the lane's row (-1: none), ...
    const int* meta = nullptr;        //   ... the phase it is updated in | batches its row needs << 16, ...
    const double* diag = nullptr;     //   ... its diagonal entry
    const int* rim = nullptr;         // rim_pitch per piece: the rows behind local indices 64, 65, ...
    const unsigned* eoff = nullptr;   // per piece 64 x 4 NB: byte offsets of two entry slots' columns in the one-column image, 16 + 16 bits
    const double* eval = nullptr;     // per piece 64 x 8 NB: the values
};
// the pieces [q_begin, q_end) -- one piece colour -- of one sweep, in place on u (row-major n x k, any k >= 1: column groups ride in the grid's second dimension)
hipError_t launch_wgs(const WgsDev& P, int q_begin, int q_end, const double* b, double* u, int k, const Ctrl* ctrl, hipStream_t st);

// ---- independent meshes in one handle (smg_hierarchy_create_union; kernels: smg_union_device.hip) ----
struct UnionDev {
    int m = 0, his_cap = 0, max_rows = 0;      // members; entries per member history; rows of the largest member on level 0
    const int* rows = nullptr;                 // level 0: internal row numbers grouped by member ...
    const int* rptr = nullptr;                 // ... m + 1 offsets
    const int* crow_member = nullptr;          // coarsest level: row -> member
    const long long* moff = nullptr;           // per member: offset of its inverse in the handle's d_Ainv
    const int* mlda = nullptr;                 // ... its leading dimension (rows padded to 64)
    const int* mrow0 = nullptr;                // ... its first row on the coarsest level
    double* ss = nullptr;                      // m: sum of squares of the member's residual
    double* his = nullptr;                     // m x his_cap: the members' residual histories
    int* nhis = nullptr;                       // m: entries written
    int* done = nullptr;                       // m: the member's loop has ended
    double* zsave = nullptr;                   // n_0 x k: the iterate before the cycle
};
// u[:, c] += blockdiag(Ainv_i) b[:, c] on the coarsest level (n rows in all)
hipError_t launch_blockdiag_gemv_add(const UnionDev& U, const double* Ainv, int n, const double* b, double* u, int k, const Ctrl* ctrl, hipStream_t st);
// per member: ss = |r|^2 over its rows, zsave = u there; then every member's break test and the handle's (ctrl): done when all members are
hipError_t launch_union_sumsq_decide(const UnionDev& U, const double* r, const double* u, int k, Ctrl* ctrl, hipStream_t st);
// after a cycle: members whose loop has ended get their rows of zsave back
hipError_t launch_union_restore(const UnionDev& U, double* u, int k, const Ctrl* ctrl, hipStream_t st);
hipError_t launch_scatter_dense(double* dense, const double* src, const long long* pos, int nnz, hipStream_t st);
hipError_t launch_dense_identity(double* dense, int np, int n, hipStream_t st);

// ---- relax(iters) of a latency-bound level in one launch: overlapped tiling (smg_tiled.hpp plan, smg_tiled_device.hip kernel) ------
struct TiledDev {
    int n_tiles = 0, nc = 0, P = 0, sweeps = 0, max_ext = 0, w_max = 0, threads = 512;
    const int* hdr = nullptr;
    const int* ext_rows = nullptr;
    const int* pcol = nullptr;
    const double* pval = nullptr;
    const int* prow = nullptr;
    const double* pdiag = nullptr;     // like prow: the diagonal entry of every panel row (its slot in the panel holds +0.0)
};
// y = relax(sweeps) of x with right-hand side b, x != y, row-major n x k blocks
hipError_t launch_tiled_gs(const TiledDev& T, const double* x, const double* b, double* y, int k, const Ctrl* ctrl, hipStream_t st);
hipError_t tiled_gs_prepare(int max_ext);

// ---- sparse coarse solver (smg_coarse.hpp): P A P^T = L L^T factored on the host, the triangular solves here --------------------------
struct SparseCholDev {
    int n = 0;
    const int* perm = nullptr;                       // new -> old
    const int *rptr = nullptr, *rcol = nullptr;      // strict lower triangle by rows
    const int *cptr = nullptr, *crow = nullptr;      // ... and by columns
    const double *rval = nullptr, *cval = nullptr, *diag = nullptr;
    double* work = nullptr;                          // 2 n KC (KC = columns per pass, sparse_coarse_work_cols): the forward solve's z, the backward solve's x
    int* err = nullptr;                              // [0] raised when a wait gave up; [1], [2]: ticket counters of the forward / backward launch
};
// u[:, c] += (L L^T)^-1 b[:, c] for the k columns of the row-major n x k blocks (caller numbering of the coarsest level)
hipError_t launch_sparse_coarse_solve(const SparseCholDev& F, const double* b, double* u, int k, const Ctrl* ctrl, hipStream_t st);
int sparse_coarse_work_cols(int k);   // F.work must hold 2 n sparse_coarse_work_cols(k) doubles for a solve with k columns

// ---- Schur-complement coarse solver (smg_schur.hpp): interior blocks of <= 64 rows eliminated exactly, the separator inverted densely ----
constexpr int SCHUR_M_MAX_DEV = 128;                 // = SCHUR_M_MAX (smg_schur.hpp): separator rows a block touches at most
struct SchurDev {
    int n = 0, nb = 0, ns = 0, ns_pad = 0;
    const int *irow = nullptr, *bsize = nullptr, *srow = nullptr, *sptr = nullptr, *sidx = nullptr, *aptr = nullptr, *ablk = nullptr, *apan = nullptr;
    double* arena = nullptr;                         // [D^-1 blocks][P^T panels][W^T panels][S^-1][products]: smg_schur.hpp
    float* arena32 = nullptr;                        // fp32 image of the arena (mixed-precision cycle), or null
    long long off_D = 0, off_P = 0, off_W = 0, off_S = 0, off_C = 0;
    const long long *coff = nullptr, *pos = nullptr, *pos2 = nullptr, *ones = nullptr, *rdst = nullptr, *rdst2 = nullptr, *rsrc = nullptr;
    const int* rptr = nullptr;
    int nnz = 0, n_ones = 0, n_red = 0;
    double *g = nullptr, *xs = nullptr;              // ns_pad x k each: the separator's right-hand side and solution
    float *g32 = nullptr, *xs32 = nullptr;
    double* sym_work = nullptr;                      // (ns_pad / 64)^2 x 64: the k = 1 product with S^-1 through its lower triangle
    double* gj_work = nullptr;                       // launch_spd_inverse's scratch for ns_pad
};
// arena <- the factorisation of the matrix whose values (CSR order of the plan's matrix) are vals
hipError_t launch_schur_factor(const SchurDev& F, const double* vals, hipStream_t st);
// *d_flag = 1 when the factorisation in the arena cannot be that of an SPD matrix (a non-positive or non-finite diagonal entry of an inverse), else 0
hipError_t launch_schur_check(const SchurDev& F, int* d_flag, hipStream_t st);
// u[:, c] += A^-1 b[:, c] for the k columns of the row-major n x k blocks (caller numbering of the coarsest level)
hipError_t launch_schur_solve(const SchurDev& F, const double* b, double* u, int k, const Ctrl* ctrl, hipStream_t st);
hipError_t launch_schur_solve_f32(const SchurDev& F, const float* b, float* u, int k, const Ctrl* ctrl, hipStream_t st);

int sell_blocks(int n_slices);  // 4 slices (waves) per 256-thread block
int sell_wide_blocks(int n_slices, int k);  // partial-sum slots the wide (k >= 8) path needs

// ctrl->sumsq = sum(partials[0..n)) in a fixed order (deterministic).  The buffer must have ss_partials_room() doubles of room behind the n partials
// (large n: a first launch leaves per-share sums there).
int ss_partials_room();
// ctrl->sumsq = sum(partials[0..n)) in a fixed order (deterministic).
hipError_t launch_ss_finalize(const double* partials, int n, Ctrl* ctrl, hipStream_t st, double* out = nullptr);   // out: where the sum goes (default ctrl->sumsq)
// r = sqrt(*sumsq); append to r_his; done = (r < ctrl->tol) or non-finite.  No-op when already done.
hipError_t launch_decide(Ctrl* ctrl, const double* sumsq, hipStream_t st);
// speculative split-phase iteration (the V-cycle overlaps the all-reduce): see smg.h, smg_solve_iter_cycle_speculative
hipError_t launch_decide_spec(Ctrl* ctrl, const double* sumsq, hipStream_t st);
hipError_t launch_copy_unless_done(double* dst, const double* src, size_t n, const Ctrl* ctrl, hipStream_t st);
hipError_t launch_restore_if_just_done(double* dst, const double* src, size_t n, const Ctrl* ctrl, hipStream_t st);
// both of the above in one launch (single-GPU path, no all-reduce in between)
hipError_t launch_ss_finalize_decide(const double* partials, int n, Ctrl* ctrl, hipStream_t st);

// u[i,:] += sum_j Ainv[i,j] * b[j,:]   (mg_VCycle.cpp:199-200 with the factorisation pre-inverted)
// sym_work (optional, (lda/64)^2 * 64 elements): with it, a single column (k = 1) is multiplied through the lower triangle of
// tiles only (the inverse is symmetric): half the bytes, two launches, deterministic per-row summation in block order.
// b, u: row-major n x ld blocks of which the first k columns take part (ld >= k)
hipError_t launch_dense_gemv_add(const double* Ainv, int n, int lda, const double* b, double* u, int k, int ld,
                                 const Ctrl* ctrl, hipStream_t st, double* sym_work = nullptr);
hipError_t launch_dense_gemv_add_f32(const float* Ainv, int n, int lda, const float* b, float* u, int k, int ld,
                                     const Ctrl* ctrl, hipStream_t st, float* sym_work = nullptr);
// first half of the symmetric k = 1 product alone: part[(I * (lda / 64) + J) * 64 + r] = the share of tile (I, J) in row 64 I + r of Ainv b; lda % 64 == 0
hipError_t launch_sym_gemv_tiles(const double* Ainv, int lda, const double* b, double* part, hipStream_t st);
hipError_t launch_sym_gemv_tiles_f32(const float* Ainv, int lda, const float* b, float* part, hipStream_t st);
// mixed precision glue
hipError_t launch_cvt_f64_f32(float* dst, const double* src, size_t n, hipStream_t st);
hipError_t launch_residual_to_f32(float* b32, float* u32, const double* r64, size_t n, const Ctrl* ctrl, hipStream_t st);
hipError_t launch_add_correction(double* z, const float* e, size_t n, const Ctrl* ctrl, hipStream_t st);
// In-place inversion of an SPD matrix (n x n, row-major, leading dimension lda, n % 64 == 0, lda == n) by
// blocked Gauss-Jordan elimination without pivoting.  work: 2*n*64 + 2*64*64 doubles.
hipError_t launch_spd_inverse(double* M, int n, double* work, hipStream_t st);

// value-only re-precompute (same sparsity as the last full precompute) -------------------------------------------
// out[e] = sum_t coef[t] * src[idx[t]]  (numeric Galerkin stage with a fixed recipe, smg_sparse.hpp)
hipError_t launch_recipe(int n_out, const int* ptr, const int* idx, const double* coef, const double* src, double* out, hipStream_t st);
// SELL panels of A(perm, perm) from A's CSR arrays (caller numbering, on the device): S.col / S.val (padded slots) are cleared and filled
// transposed: the image of A(perm, perm)^T instead -- A structurally symmetric (launch_bit_symmetric), values looked up by bisection
hipError_t launch_sell_fill(const int* ptr, const int* col, const double* val, const int* perm, const int* iperm, const SellDev& S, size_t padded, hipStream_t st,
                            bool transposed = false);
// slot[r] = index of a_rr in the value array of the filled square image S (-1: not stored); *first_missing (preset to n_rows by the caller) =
// the smallest row without one
hipError_t launch_sell_diag_slots(const SellDev& S, int* slot, int* first_missing, hipStream_t st);
// one empty launch: makes the runtime load this library's main code object now rather than inside the first real launch
hipError_t warm_device_code(hipStream_t st);
// map[slot] = index of the CSR entry slot holds in the image launch_sell_fill(..., transposed) builds (-1: padding): the gather map of
// the value-only re-precompute, for an image of that layout however it was built
hipError_t launch_sell_fill_map(const int* ptr, const int* col, const int* perm, const int* iperm, const SellDev& S, size_t padded, bool transposed, int* map, hipStream_t st);
// square CSR matrix (rows sorted) on the device: *differs = 0 when A == A^T bit for bit; bit 0: some value differs from its mirror image,
// bit 1: some entry has none (A is not structurally symmetric)
hipError_t launch_bit_symmetric(int n, const int* ptr, const int* col, const double* val, int* differs, hipStream_t st);
// dst[i] = map[i] >= 0 ? src[map[i]] : 0   (refresh of SELL value panels / LHS and Auk slices)
hipError_t launch_gather_vals(double* dst, const double* src, const int* map, size_t n, hipStream_t st);
// dense (np x np, row-major) = identity on the padding rows, zero elsewhere, then dense[pos[i]] = src[i]
hipError_t launch_dense_from_csr(double* dense, int np, int n, const double* src, const long long* pos, int nnz, hipStream_t st);
hipError_t launch_add_at(double* v, const int* where, int n, double c, hipStream_t st);

// operator assembly on the device for a fixed connectivity (row f-3): per-face terms, lumped mass, CSR values
// val = mass_coef * M + lap_coef * L  (L = igl::cotmatrix convention); Lval (optional) receives L alone.
hipError_t launch_assemble(int nV, int nF, int nnz, const double* V, const int* F, int voronoi, const int* l_ptr, const int* l_idx,
                           const signed char* l_sgn, const int* m_ptr, const int* m_idx, const int* diag_of, double* Qc, double* Qm,
                           double* Md, double mass_coef, double lap_coef, double* val, double* Lval, hipStream_t st);

// layout helpers ------------------------------------------------------------------------------------------
// dst[i*kin + c] = c < k ? src[map[i] + c*ld_src] : 0      (column-major caller block -> internal block of kin >= k columns per row)
hipError_t launch_gather_in(double* dst, const double* src, const int* map, int n, int k, int kin, int ld_src,
                            hipStream_t st);
// dst[map[i] + c*ld_dst] = src[i*kin + c], c < k
hipError_t launch_scatter_out(double* dst, const double* src, const int* map, int n, int k, int kin, int ld_dst,
                              hipStream_t st);
// dst[idx[i] + c*ld_dst] = src[i + c*ld_src]  (column-major -> column-major scatter, known values)
hipError_t launch_scatter_cm(double* dst, const double* src, const int* idx, int n, int k, int ld_src, int ld_dst,
                             hipStream_t st);
// dst[i + c*ld_dst] = src[idx[i] + c*ld_src]  (column-major row gather, igl::slice(X, idx, 1, Y))
hipError_t launch_gather_cm(double* dst, const double* src, const int* idx, int n, int k, int ld_src, int ld_dst,
                            hipStream_t st);
// y[i + c*ld] -= sum_p val[p] * x[col[p] + c*ldx]   (CSR, column-major blocks; RHS_u -= Auk * known_val,
// min_quad_with_fixed_mg.cpp:318.  The product is summed first, then subtracted.)
hipError_t launch_csr_sub(int n_rows, const int* ptr, const int* col, const double* val, const double* x, int ldx,
                          double* y, int ld, int k, hipStream_t st);

}  // namespace smg
