/*
 * smg.h -- C ABI of libsmg: the MI355X-native surface-multigrid solve path.
 *
 * Drop-in boundary for the `mg_VCycle` / `min_quad_with_fixed_mg_*` path of
 * HTDerekLiu/surface_multigrid_code.  The reference has no FFI layer: its boundary is the set of C++
 * free functions in src/min_quad_with_fixed_mg.h:32-113, src/mg_VCycle.h:22-76 and
 * src/mg_precompute.h:15-32 taking Eigen objects by reference.  Each entry point below names the
 * reference interface it replaces (paths relative to the reference checkout).  The C++ mirror with the
 * reference's own function names/argument order lives in surface_multigrid_code_amd/csrc/mg_api.hpp;
 * INTEGRATION.md shows the Eigen-side binding.
 *
 * Conventions
 *   - plain pointers and sizes only; fp64 values, int32 indices (Eigen's default StorageIndex);
 *   - sparse matrices are passed as CSR (rowptr, col, val).  For the symmetric system matrix these are the
 *     very arrays Eigen's column-major SparseMatrix holds (outerIndexPtr, innerIndexPtr, valuePtr); for the
 *     prolongation P a `_csc` twin takes Eigen's compressed-column arrays directly;
 *   - dense blocks are column-major n x k with a leading dimension, exactly Eigen::MatrixXd / VectorXd;
 *   - every function returns an int status: 0 = ok, < 0 = error (enum below); nothing throws across the ABI;
 *     `smg_last_error()` gives a thread-local message;
 *   - a handle owns its device memory and one HIP stream; a handle is not thread-safe, distinct handles are;
 *   - compute entry points REQUIRE a HIP device: without one they fail with SMG_ERR_NO_DEVICE -- there is no
 *     CPU fallback inside this library (the CPU oracle lives in oracle/ and is test infrastructure only).
 */
#ifndef SMG_H
#define SMG_H

#ifdef __cplusplus
extern "C" {
#endif

#define SMG_VERSION 501

enum {
    SMG_OK = 0,
    SMG_ERR_INVALID = -1,     /* bad argument / call order */
    SMG_ERR_NO_DEVICE = -2,   /* no usable HIP device */
    SMG_ERR_HIP = -3,         /* a HIP runtime call failed */
    SMG_ERR_NONFINITE = -4,   /* residual became NaN/Inf */
    SMG_ERR_ALLOC = -5,
    SMG_ERR_IO = -6,
    SMG_ERR_REDUCE = -7       /* the caller's reduction (smg_solve_sharded) reported a failure */
};

enum { SMG_HOST = 0, SMG_DEVICE = 1 };         /* where the dense blocks handed to smg_solve live */

/* decimation types of mg_precompute (src/mg_precompute.h:9: 0 qslim, 1 midpoint, 2 vertex removal) */
enum { SMG_DEC_QSLIM = 0, SMG_DEC_MIDPOINT = 1, SMG_DEC_VERTEX_REMOVAL = 2 };

/* smoother of the V-cycle (the slot of relax(), src/mg_VCycle.cpp:113-178).  The reference only has Gauss-Seidel; damped Jacobi is
 * the "Gauss-Seidel/Jacobi smoothing" of this build's brief: one whole-matrix launch per sweep instead of one launch per colour.
 *   GS      forward Gauss-Seidel on every level: the reference's sweep on the colour-major numbering (default)
 *   JACOBI  damped Jacobi on every level:  u_i <- u_i + omega * ((b_i - sum_{j != i} A(j,i) u_j) / A_diag_i - u_i), all rows from the old u
 *   HYBRID  Gauss-Seidel on the levels with more than `jacobi_max_rows` unknowns (bandwidth-bound: a sweep costs its bytes),
 *           Jacobi on the smaller ones (launch-latency-bound: a sweep costs its launches)
 *   CHEBYSHEV / HYBRID_CHEBYSHEV  the same two layouts with Chebyshev-accelerated Jacobi instead of damped Jacobi: relax(iters) is ONE
 *           polynomial of degree iters + 1 in D^-1 A (so V(2,2) smooths with degree 3: three whole-matrix launches where two Gauss-Seidel
 *           sweeps take two launches per colour), optimal on [cheby_fraction * lam, lam] with lam the Gershgorin bound
 *           max_i sum_j |A(j,i)| / A_diag_i.  Step s of the three-term recurrence (theta, delta = centre / half width of that interval,
 *           sigma = theta / delta, rho_0 = 1 / sigma):   r_i = (b_i - sum_{j != i} A(j,i) u_j) / A_diag_i - u_i,
 *           s = 0: d = r / theta;  s >= 1: rho_s = 1 / (2 sigma - rho_{s-1}), d = rho_s rho_{s-1} d + (2 rho_s / delta) r;   u += d.
 *           Measured: as many cycles as Gauss-Seidel everywhere where damped Jacobi needs 30-70 % more (anisotropic torus, C5). */
enum { SMG_SMOOTH_GS = 0, SMG_SMOOTH_JACOBI = 1, SMG_SMOOTH_HYBRID = 2, SMG_SMOOTH_CHEBYSHEV = 3, SMG_SMOOTH_HYBRID_CHEBYSHEV = 4 };

typedef struct smg_hierarchy smg_hierarchy;

/* Parameters the reference hard-codes or passes through default-argument overloads:
 *   tol = 1e-3 (min_quad_with_fixed_mg.cpp:63,270), max_iter = 20 (:77,:285), pre = post = 2 (:102-103,:324-325) */
typedef struct {
    double tol;
    int max_iter;
    int pre, post;
    int verbosity;     /* 0 silent; 1 prints "MG iteration: i, residual: r" lines like the reference (:111) */
    int check_every;   /* smg_solve looks at the device-side convergence flag every this many iterations; 0 (default): adaptive -- the
                          number of cycles still needed is extrapolated from the last two residuals, and all but the last of them are
                          enqueued before the next look.  Results do not depend on it (the break test runs on the device). */
    int use_graph;     /* 1: replay one captured hipGraph per outer iteration; 0: eager launches */
    int precision;     /* 0 (default): everything fp64, the reference arithmetic.  1: mixed -- the outer iterate and the
                          residual (hence r_his and the stopping test) stay fp64, the V-cycle runs on fp32 copies of the
                          operators as z += V32(RHS - A z): same iteration in exact arithmetic, fp64 accuracy at convergence,
                          ~2/3 of the bytes (BASELINE config 5: fp32 vs fp64) */
    int smoother;      /* SMG_SMOOTH_GS (default, the reference) / SMG_SMOOTH_JACOBI / SMG_SMOOTH_HYBRID */
    double omega;      /* Jacobi damping factor (default 0.8) */
    int jacobi_max_rows; /* HYBRID*: levels with at most this many unknowns are smoothed by (Chebyshev-)Jacobi (default 100000) */
    double cheby_fraction; /* lower end of the Chebyshev interval as a fraction of the Gershgorin bound (default 0.1) */
} smg_solve_opts;
void smg_solve_opts_default(smg_solve_opts *o);

int smg_version(void);
const char *smg_last_error(void);
int smg_device_count(void);
/* device memory currently held by all libsmg handles / assemblers of this process, in bytes (memory budget reporting) */
long long smg_device_bytes_live(void);
/* what ONE handle holds in HBM, by purpose: lines "name bytes" (and "total bytes") written into buf (memory budget reporting) */
int smg_debug_device_bytes(const smg_hierarchy *h, char *buf, int cap);

/* ---- std::vector<mg_data> (src/mg_data.h:11-27) ---------------------------------------------------------------- */
/* mg.reserve(nLvs) */
smg_hierarchy *smg_hierarchy_create(int n_levels);
void smg_hierarchy_destroy(smg_hierarchy *h);
int smg_hierarchy_levels(const smg_hierarchy *h);
/* Use an existing HIP stream (e.g. torch.cuda.current_stream().cuda_stream) instead of the handle's own. */
int smg_hierarchy_set_stream(smg_hierarchy *h, void *hip_stream);
/* Smoother used by smg_vcycle / smg_relax and the raw / bench entry points (which take no smg_solve_opts); smg_solve* set the same
 * state from their opts.  omega <= 0 keeps the current value, jacobi_max_rows < 0 likewise. */
int smg_hierarchy_set_smoother(smg_hierarchy *h, int smoother, double omega, int jacobi_max_rows);
int smg_hierarchy_set_chebyshev(smg_hierarchy *h, double cheby_fraction);     /* (0, 1); <= 0 keeps the current value */
/* the Gershgorin bound the Chebyshev smoother of level lv uses (0 before smg_precompute / on the coarsest level) */
double smg_level_spectral_bound(const smg_hierarchy *h, int lv);
/* mg[lv].P_full = mg[lv].P = P; mg[lv].PT = P^T  (what mg_precompute stores per level, src/mg_precompute.cpp:71-77).
 * P is #V_{lv-1} x #V_lv for lv = 1 .. n_levels-1 (the operator lives on the COARSER level, mg_VCycle.cpp:80,:91). */
int smg_level_set_prolong(smg_hierarchy *h, int lv, int n_fine, int n_coarse, const int *rowptr, const int *col,
                          const double *val);
int smg_level_set_prolong_csc(smg_hierarchy *h, int lv, int n_fine, int n_coarse, const int *colptr,
                              const int *rowidx, const double *val);
/* mg[lv].V / mg[lv].F (optional; not used by the solve) */
int smg_level_set_mesh(smg_hierarchy *h, int lv, const double *V, int nV, const int *F, int nF);
/* read them back (query sizes with NULL arrays first); V: nV x 3 row-major, F: nF x 3 */
int smg_level_get_mesh(const smg_hierarchy *h, int lv, int *nV, int *nF, double *V, int *F);

/* ---- mg_precompute (src/mg_precompute.h:26-32, src/mg_precompute.cpp:15-87) ------------------------------------ */
/* Builds the hierarchy from a triangle mesh: level count by the reference's float rule (:27-38), per level
 * tarF = round(#F * ratio) (:59), edge-collapse decimation + prolongation.  V: nV x 3 row-major, F: nF x 3.
 * dec_type (src/mg_precompute.cpp:10): 0 qslim (quadric error metric, merged vertex at the quadric's minimiser), 1 mid-point
 * (shortest edge first), 2 vertex removal (shortest edge first, merged vertex on an end point).
 * libsmg's own host implementation of the reference's construction (src/SSP_midpoint.cpp, src/SSP_collapse_edge.cpp, src/joint_lscm.cpp,
 * src/query_fine_to_coarse.cpp): boundary closed by a vertex at infinity, greedy shortest-edge collapse to the mid-point with libigl's
 * refuse / re-cost queue discipline, joint conformal flattening of the 1-rings before / after every collapse in the reference's three
 * cases (interior, one boundary end point, boundary edge with its snap candidates), the reference's validity and quality thresholds;
 * same P structure (3 stored entries per row, rows sum to 1).  Written from the formulation on own data structures; reproduces the
 * reference's checked-in 08_subdiv_remesh outputs point for point (tests/golden/bunny_remesh_500.npz).  Ties between exactly
 * equal-cost edges may be broken differently than libigl's edge numbering does (csrc/smg_decimate.cpp).
 * ONLY dec_type 1 (mid-point, the default of every reference caller: src/mg_precompute.cpp:94, 03_mg_solver/main.cpp:37) restates the reference and is
 * pinned against its outputs.  dec_type 0 and 2 run the same collapse machinery with LIBSMG'S OWN cost and placement (quadric error / end-point placement as
 * described above): they are NOT restatements of src/SSP_qslim.cpp / src/SSP_vertexRemoval.cpp and no parity with them is claimed. */
int smg_mg_precompute(const double *V, int nV, const int *F, int nF, float ratio, int nVCoarsest, int dec_type,
                      smg_hierarchy **out);
/* The same with an opt-in departure from the reference's plain greedy collapse order: a surviving vertex may stand for at most
 * absorption_cap x (#F / tarF) input vertices (edges that would exceed that wait; the bound doubles when nothing else can be collapsed).
 * Shortest-edge-first decimation coarsens densely sampled regions far beyond the requested ratio before it touches the rest, which the
 * V-cycle pays for (ogre.obj: factor 0.6 -> 0.3 with a cap of 2).  absorption_cap = 0 is smg_mg_precompute. */
int smg_mg_precompute_capped(const double *V, int nV, const int *F, int nF, float ratio, int nVCoarsest, int dec_type,
                             float absorption_cap, smg_hierarchy **out);
/* The same, keeping the record of every collapse (the reference's decInfo / decIM outputs of SSP_decimate, src/SSP_decimate.h:10-22;
 * host memory ~0.5 kB per collapse) when keep_log != 0: smg_query_coarse_to_fine needs it. */
int smg_mg_precompute_logged(const double *V, int nV, const int *F, int nF, float ratio, int nVCoarsest, int dec_type,
                             float absorption_cap, int keep_log, smg_hierarchy **out);
/* query_coarse_to_fine (src/query_coarse_to_fine.h; 08_subdiv_remesh/main.cpp:146): n points on the mesh of level lv (lv >= 1) --
 * face[i] a face of mg[lv].F, bary[3i..3i+2] barycentric coordinates with respect to its corners -- are carried through the bijection
 * of the successive self-parameterisation onto the mesh of level lv - 1 by undoing the collapses of that coarsening step, last to
 * first: out_face[i] a face of mg[lv-1].F, out_bary[3i..] coordinates there (>= 0, sum 1).  The level must have been built by
 * smg_mg_precompute_logged(keep_log = 1); SMG_ERR_INVALID otherwise (also after smg_hierarchy_load: the record is not part of the
 * .smgh file).  Host only (no GPU involved). */
int smg_query_coarse_to_fine(const smg_hierarchy *h, int lv, int n, const int *face, const double *bary, int *out_face,
                             double *out_bary);
/* query_fine_to_coarse (src/query_fine_to_coarse.h; what get_prolong does for the vertices, src/get_prolong.cpp:23-57): the other
 * direction -- points of level lv - 1's mesh onto level lv's mesh, collapses first to last.  A vertex of the fine mesh (one-hot
 * coordinates in any of its faces) arrives where its row of mg[lv].P_full says. */
int smg_query_fine_to_coarse(const smg_hierarchy *h, int lv, int n, const int *face, const double *bary, int *out_face,
                             double *out_bary);
/* Hierarchy of a mid-point-subdivided mesh: the n_sub finest transfer operators are the subdivision operators
 * (09_random_subdiv_remesh/main.cpp:46-140), levels below the base mesh come from smg_mg_precompute's decimator
 * (ratio, nVCoarsest applied to the base mesh; pass n_extra_levels = -1 for the float rule).  Outputs the fine
 * mesh through V_out/F_out (caller-allocated: nV_fine x 3, nF * 4^n_sub x 3) when non-null. */
int smg_mg_precompute_subdiv(const double *V, int nV, const int *F, int nF, int n_sub, float ratio, int nVCoarsest,
                             int n_extra_levels, smg_hierarchy **out, double *V_out, int *F_out);

/* mg_precompute_block (src/mg_precompute_block.h, src/mg_precompute_block.cpp:23-95; get_prolong_block,
 * src/get_prolong.cpp:59-115): same hierarchy with P (x) I_3 for 3-DOF-per-vertex systems, DOF index = 3*vertex + d.
 * To the V-cycle the block system is a plain sparse matrix. */
int smg_mg_precompute_block(const double *V, int nV, const int *F, int nF, float ratio, int nVCoarsest, int dec_type,
                            smg_hierarchy **out);
/* Block (3 x 3) kernels for such hierarchies.  The reference runs its scalar kernels on the 3n x 3n system its caller assembles
 * (06_example_balloon_sim/sim_utils/implicit_euler_mg_balloon.h:63-76: an elasticity Hessian, one 3 x 3 block per vertex pair).
 * smg_precompute recognises the structure -- every prolongation of the form Pv (x) I_3 (however the handle got them: this builder,
 * smg_level_set_prolong, a file), no constraints, n % 3 == 0 -- and then keeps the level matrices in 3 x 3 blocks (76 instead of
 * 108 bytes per block), numbers VERTICES colour-major and lets one lane update the three DOFs 3v, 3v+1, 3v+2 of its vertex in order:
 * the reference's lexicographic Gauss-Seidel sweep (src/mg_VCycle.cpp:146-160) on that numbering, with as many launches per sweep as
 * the vertex graph has colours; P (x) I_3 is applied from Pv.  Results: the reference algorithm on the scalar matrix, bit for bit
 * per kernel in the device numbering (smg_level_get_matrix(..., internal = 1) is the scalar matrix in it).
 * mode: -1 (default) decide at smg_precompute: structure present AND the blocks of A at least half full (kron(S, I_3) is better
 * served as three right-hand sides of a scalar problem); 0 never; 3 required (smg_precompute fails when the structure is absent).
 * Constraints: pinned VERTICES (all three DOFs 3v, 3v+1, 3v+2 in `known`) keep the structure -- the reference's slices and column drops
 * (src/min_quad_with_fixed_mg.cpp:137-257) are formed on the scalar matrices and factor as Pv' (x) I_3 again -- and stay on the block
 * kernels; constraints on single degrees of freedom select the scalar path (mode 3: smg_precompute fails and says so).
 * The mixed-precision cycle (smg_solve_opts.precision) is available: the V-cycle runs on an fp32 image of the 3 x 3-block panels. */
int smg_hierarchy_set_block_mode(smg_hierarchy *h, int mode);
int smg_hierarchy_block_size(const smg_hierarchy *h);   /* 1 or 3: what the last smg_precompute decided */
/* block image of A_lv (lv < n_levels - 1, block hierarchies only): stored 3 x 3 blocks, allocated block slots (SELL padding
 * included), vertex colours = launches per Gauss-Seidel sweep */
int smg_level_block_stats(const smg_hierarchy *h, int lv, long *n_blocks, long *n_block_slots, int *n_vertex_colors);
/* The block SELL image of A_lv as the device holds it (tests; built on the host, works without a GPU once the host half of
 * smg_precompute has run on a block hierarchy).  Slices of <= 64 vertices, one lane per vertex; panel column j of slice s: block column
 * col[(slice_off[s] + j) * 64 + lane] (-1 = padding) and value plane e = 3 * row_in_block + col_in_block at
 * val[((slice_off[s] + j) * 9 + e) * 64 + lane].  Query n_slices / n_panel_cols with NULL arrays first. */
int smg_level_get_block_image(const smg_hierarchy *h, int lv, int *n_slices, int *n_panel_cols, int *slice_row, int *slice_off, int *slice_w,
                              int *col, double *val);
/* relax() with MANY right-hand sides (reference k > 1 branch, src/mg_VCycle.cpp:161-177: k independent lexicographic sweeps).
 * With k a multiple of 16 (k >= 16) the levels of at least min_rows rows (default: never; SMG_BGS_MIN_ROWS; < 0: never; SMG_BGS=0: never)
 * can sweep BLOCK-wise: the level is cut into compact blocks of <= 64 rows, blocks are coloured, one launch per block colour; a wavefront
 * owns (block, 16 columns), reads the block's rows and its rim once into LDS and updates the block's rows there, <= 16 independent rows at a time.  That is the reference's lexicographic sweep on the numbering (block colour, block, vertex colour, row) -- per kernel bit for bit
 * what the oracle computes on that numbering, smg_level_get_block_gs_order -- and reads the iterate ~1.65 times per sweep instead of 3
 * (the multi-colour order of the wide kernels re-reads it once per colour).  Measured at 1 M rows: the fine-level sweep 8 - 10 % faster at
 * 64 columns, 22 % at 32; the plan costs a graph partition of the level at the first such solve, so it is an option for
 * callers that solve often, not the default.  It is a different, equally valid Gauss-Seidel order than the multi-colour one: iterates of
 * the two paths differ, converged solutions agree to the tolerance (same cycle counts measured).  The outer residual is then a launch of
 * its own (the multi-colour path folds it into its first sweep).  Changing min_rows takes effect at the next solve. */
int smg_hierarchy_set_block_gs(smg_hierarchy *h, int min_rows);
/* The block-sequential order of level lv as a solve with k columns would use it (after smg_precompute; builds the plan): *n_blocks,
 * *n_colors, color_ptr[n_colors + 1] (blocks per block colour), blk_ptr[n_blocks + 1] (positions per block), rows[n] (position -> row
 * in the INTERNAL numbering, smg_level_get_perm), stats[2] = {rows gathered per row beyond the iterate itself, share of the walk's row
 * slots that hold a row of their own (the rest repeat one)}.  Any pointer may be NULL.  Returns 1 when level lv sweeps block-sequentially for this k, 0 when
 * it does not (nothing is written then), < 0 on error. */
int smg_level_get_block_gs_order(smg_hierarchy *h, int lv, int k, int *n_blocks, int *n_colors, int *color_ptr, int *blk_ptr, int *rows, double *stats);
/* relax() on the Galerkin levels of the REFERENCE's own hierarchies (smg_mg_precompute = src/mg_precompute.cpp:15-87: A_l = PT A P with 18 - 30 entries
 * per row, 11 - 15 colours).  Such a level sweeps PIECE-wise: compact pieces of <= 64 rows, the piece graph coloured (4 - 6 colours whatever the rows'
 * degree), one launch per piece colour, one wavefront per piece (lane = row, the row in registers, the piece's rows and rim in LDS, rows updated phase by
 * phase there).  That is the reference's lexicographic sweep (src/mg_VCycle.cpp:146-160) on the numbering (piece colour, piece, local colour, row) -- per
 * kernel bit for bit what the oracle computes on that numbering, smg_level_get_wave_gs_order.  A different, equally valid Gauss-Seidel order than the
 * multi-colour one: iterates differ, converged solutions agree to the tolerance, cycle counts are the same (measured).
 * mode: -1 (default) automatic = Gauss-Seidel levels of 512 - 600 000 rows with more than 5 colours or rows of more than 12 entries that have no
 * one-launch relax() (overlapped tiling), any number of columns (the order of a level's sweep never depends on k: a column-sharded solve iterates bit for bit
 * like the fused one), fp64 cycles; 0 never (one launch per colour); 1 every Gauss-Seidel level in that size range, for every k (such a level then takes no
 * one-launch relax() either: that exists for k <= 7 only, and the order must not depend on k).
 * SMG_WGS=0 / 1 / 2 overrides (off / automatic / every level).  Takes effect at the next solve. */
int smg_hierarchy_set_wave_gs(smg_hierarchy *h, int mode);
/* Memory against speed (the reference has no counterpart: mg_data holds Eigen's compact CSC, src/mg_data.h:11-27).  By default every operator's SELL panels get a
 * FIXED pitch -- room for the level's widest slice (A_0 of a triangle mesh: 12 columns for 7 used) -- so that a wave addresses its panel from its slice number alone
 * and no launch waits for a table: 545 MB for the 171 MB of CSR operators of the 1 M-vertex benchmark hierarchy.  on = 1: compact panels + a slice-offset table
 * (what matrices with a few very wide slices get anyway): 421 MB there, the V(2,2) cycle 14 - 16 % slower (0.313 -> 0.362 ms: one more dependent load in front of
 * every launch), results bit-identical.  For many resident meshes per GPU.  Takes effect at the next smg_precompute, which is then a full one. */
int smg_hierarchy_set_memory_lean(smg_hierarchy *h, int on);
/* The piece-sequential order of level lv as a solve with k columns would use it (after smg_precompute; builds the plan): *n_pieces, *n_colors,
 * color_ptr[n_colors + 1] (pieces per piece colour), piece_ptr[n_pieces + 1] (positions per piece), rows[n] (position -> row in the INTERNAL numbering,
 * smg_level_get_perm), stats[3] = {rows gathered per row beyond the iterate itself, mean phases per piece, most phases of a piece}.  Any pointer may be
 * NULL.  Returns 1 when level lv sweeps piece-wise for this k, 0 when it does not (nothing is written then), < 0 on error. */
int smg_level_get_wave_gs_order(smg_hierarchy *h, int lv, int k, int *n_pieces, int *n_colors, int *color_ptr, int *piece_ptr, int *rows, double *stats);
/* ---- independent meshes in ONE handle (BASELINE north_star: "independent RHS columns / independent meshes shard") ----------------------------------
 * Across GPUs: one handle per device.  On ONE GPU separate handles do not overlap (a hipGraphLaunch of ~50 kernel nodes is enqueued under a process-wide
 * lock), so many small meshes go into one block-diagonal handle whose every launch serves all of them.
 * smg_hierarchy_create_union: members = m hierarchies with their prolongations set (smg_mg_precompute, smg_level_set_prolong, ...; same number of levels,
 * scalar); *out gets P_full_l = diag(P_full_l of the members), rows and columns in member order.  The members are only read and may be destroyed afterwards.
 * smg_precompute(out, A, ...) then takes the block-diagonal system (member i's rows are [first, first + count) of smg_union_member_rows; constraints in
 * that numbering).  What the reference does PER MESH stays per mesh: every member is its own min_quad_with_fixed_mg_solve loop
 * (src/min_quad_with_fixed_mg.cpp:105-134) -- its own residual norm, history and break test; a member whose test has passed keeps the iterate it had then
 * while the others go on (smg_union_get_history after smg_solve / smg_solve_end: its residual history, and the reference's return value for it) -- and
 * coarseSolve() uses the members' OWN dense inverses (sum n_i^2 entries, not (sum n_i)^2; every member's coarsest level must lie in the dense range).
 * The handle's own r_his is the norm over all members, `converged` = every member's loop ended below the tolerance.  Members are numerically isolated:
 * one whose residual stops being finite ends ITS loop as failed (its history ends with that value, its `converged` is 0, the handle's too) while the
 * others iterate on to their own tolerance, unaffected bit for bit; it no longer enters the handle's norm.  fp64 cycles only (precision = 1 is refused
 * before anything of the handle changes); no split-phase /
 * sharded iteration on a union.  Numberings and sweep orders are those of the union's matrices: a member's iterates agree with a stand-alone solve of the
 * same mesh to the tolerance, not bit for bit. */
int smg_hierarchy_create_union(const smg_hierarchy *const *members, int m, smg_hierarchy **out);
int smg_union_members(const smg_hierarchy *h);                                           /* 0: not a union */
int smg_union_member_rows(const smg_hierarchy *h, int member, int *first, int *count);   /* member's rows in the caller's numbering of level 0 */
int smg_union_get_history(smg_hierarchy *h, int member, double *r_his, int cap, int *n_his, int *converged);
/* The coarsest level's solver (coarseSolve(), src/mg_VCycle.cpp:181-201; solver.compute(Ac), src/min_quad_with_fixed_mg.cpp:47-48, :253-254).  Three of them,
 * chosen by size and by what the caller does (all: <= 1e-11 from LDL^T, deterministic):
 *  - DENSE INVERSE, up to n_max unknowns (smg_hierarchy_set_coarse_dense_max, default 16384, or SMG_COARSE_DENSE_MAX): the matrix is inverted on the device
 *    (blocked symmetric Gauss-Jordan on the matrix cores) and applied as a bandwidth-bound product (8 n^2 bytes of HBM, half of them streamed per cycle and
 *    column): 0.02 ms per cycle at 4 k unknowns, 0.2 ms at 16 k; the cheapest to apply below ~6 k unknowns, n^3 flops to build (2.4 ms at 4 k, 0.1 s at 16 k).
 *  - SCHUR COMPLEMENT (smg_hierarchy_set_coarse_schur below): one level of exact block elimination, only the separator (0.27 - 0.43 n rows) inverted densely.
 *  - SPARSE CHOLESKY P A P^T = L L^T -- what the reference's Eigen::SimplicialLDLT does -- computed on the host during smg_precompute (nested-dissection
 *    order), the two triangular solves on the device (one launch each, rows wait for the rows they read): O(n log n) memory, 2.6 ms per solve at 16 k
 *    unknowns, 16 ms at 63 k; for coarsest levels beyond the other two, or on request.  Not available with it: the mixed-precision cycle.
 * So mg_precompute's nVCoarsest may be anything the reference accepts, down to a 1-level call on the whole mesh.
 * smg_hierarchy_coarse_solver: 0 dense inverse / 1 sparse Cholesky / 2 Schur complement after a precompute; *factor_entries: n^2, the entries of L, resp. the
 * doubles the Schur solver keeps. */
int smg_hierarchy_set_coarse_dense_max(smg_hierarchy *h, int n_max);
int smg_hierarchy_coarse_solver(const smg_hierarchy *h, long *factor_entries);
/* The Schur-complement coarse solver (csrc/smg_schur.hpp): the rows are cut into compact blocks of <= 64, a vertex cover of the entries between blocks becomes
 * the separator, every block is inverted in LDS, and only the separator's Schur complement is inverted densely.  coarseSolve becomes g = b_S - sum W_i^T b_i,
 * x_S = S^-1 g, x_i = D_i^-1 b_i - W_i x_S: three launches instead of two, a quarter of the bytes (or less).  Same answer to rounding, bit-identical from
 * run to run; available in fp32 for the mixed-precision cycle.  For coarsest levels of n_min (default 2048, or SMG_COARSE_SCHUR_MIN; n_min < 0: unchanged)
 * to 65 536 (SMG_COARSE_SCHUR_MAX) unknowns:
 *   when = 0  never;
 *   when = 1  from the first smg_precompute on;
 *   when = 2  (default, or SMG_COARSE_SCHUR) THE CHOICE BY COST:
 *             - below 6 144 unknowns (SMG_COARSE_SCHUR_BIG) a handle that is factored once keeps the dense inverse (its cycle is ~4 us cheaper at 4 k unknowns);
 *               the first VALUE-ONLY re-precompute moves it to the Schur complement -- a caller that sends new values for an old pattern (05_example_mean_
 *               curvature_flow/main.cpp:74, 06: implicit_euler_mg_balloon.h:75) pays the factorisation at every step: 0.9 ms instead of 2.4 at 3 952 unknowns,
 *               value-only smg_precompute 3.1 -> 1.4 ms (the switch itself costs one plan on the host, ~ms, once);
 *             - from 6 144 unknowns on it is cheaper to build AND to apply: taken at the first precompute (15 804 unknowns: 38 us per solve and 182 MB
 *               against 204 us and 2 GB);
 *             - above the DEFAULT n_max (16 384) it stands in for the sparse factorisation (63 210 unknowns: 0.30 ms per solve against 15.8 ms, 2.6 GB
 *               against 0.1 GB).  A caller who SET n_max (smg_hierarchy_set_coarse_dense_max, SMG_COARSE_DENSE_MAX) gets the sparse factorisation above it --
 *               that call bounds memory, and the separator's inverse is dense -- unless when = 1 asks for the Schur solver explicitly.
 * A matrix whose blocks touch more than 128 separator rows each, or whose separator exceeds 0.7 n or 24 576 rows (its inverse is dense), or whose arena
 * would exceed SMG_SCHUR_ARENA_MAX_MB (default 6 144) or does not fit the device, keeps the dense inverse resp. the sparse factorisation (nothing half-built
 * stays behind).  After every factorisation the diagonals of the inverted blocks and of S^-1 are checked: a coarsest matrix that is not positive definite
 * fails smg_precompute with SMG_ERR_INVALID, as the sparse factorisation does. */
int smg_hierarchy_set_coarse_schur(smg_hierarchy *h, int when, int n_min);
/* On-disk hierarchy ({P_full_l}, optional V/F per level): build the expensive hierarchy once, ship it as a fixture.
 * Format (little endian): "SMGH" u32 version=1 i32 n_levels, then per level: i32 nV i32 nF f64 V[3nV] i32 F[3nF],
 * and for lv >= 1: i32 n_rows i32 n_cols i32 nnz i32 ptr[n_rows+1] i32 col[nnz] f64 val[nnz]. */
int smg_hierarchy_save(const smg_hierarchy *h, const char *path);
int smg_hierarchy_load(const char *path, smg_hierarchy **out);

/* ---- min_quad_with_fixed_mg_precompute (src/min_quad_with_fixed_mg.h:32-36 and :72-77) ------------------------- */
/* A: n x n symmetric, CSR == CSC.  known == NULL / n_known == 0 selects the no-constraint overload
 * (.cpp:3-51); otherwise the `known` overload (.cpp:137-257): unknown = setdiff, LHS/Auk slices, P re-organised
 * to unknowns with the all-(<=1e-15) column drop cascade, Galerkin A_l = PT_l A_{l-1} P_l, +1e-12 on the coarsest
 * diagonal, A_diag, coarsest factorisation (here: dense inverse computed on the device).  May be called again on
 * the same handle (new matrix every time step, 05_example_mean_curvature_flow/main.cpp:74); always restarts
 * from P_full.
 * A first (pattern-changing) call works on several host threads: the sparse algebra and the numberings on a thread of its own and a
 * process-wide pool of SMG_HOST_THREADS workers (default min(hardware threads, 32), kept for later calls), while the calling thread
 * brings the device up and builds the level images as they become ready -- 0.1 s for a million unknowns.  The arrays are read until
 * the call returns and not after; errors are reported on the calling thread as usual. */
int smg_precompute(smg_hierarchy *h, int n, const int *rowptr, const int *col, const double *val, const int *known,
                   int n_known);

/* Same sparsity / constraints / prolongations as the last smg_precompute on this handle, new VALUES already resident in
 * HBM (d_val: device pointer, the caller's CSR order): the whole re-precompute runs on the GPU (fixed-recipe Galerkin
 * products, SELL refresh, coarse inverse).  smg_precompute() takes the same path automatically when it is handed a
 * matrix with an unchanged pattern. */
int smg_precompute_values_device(smg_hierarchy *h, const double *d_val);

/* ---- operator assembly on the device for a fixed connectivity -------------------------
 * What the callers do with libigl around the solve every time step (05_example_mean_curvature_flow/main.cpp:66-69:
 * massmatrix(U), LHS = M - delta L, RHS = M U; 03_mg_solver/main.cpp:44-61) -- here as three kernels on new vertex
 * positions that never leave HBM.  Values are bit-identical to smg_mesh_cotmatrix / smg_mesh_massmatrix. */
typedef struct smg_assembler smg_assembler;
int smg_assembler_create(const int *F, int nF, int nV, smg_assembler **out);
void smg_assembler_destroy(smg_assembler *a);
/* CSR pattern of the assembled matrix (== the pattern of smg_mesh_cotmatrix); query nnz with NULL arrays first */
int smg_assembler_pattern(const smg_assembler *a, int *nnz, int *rowptr, int *col);
/* d_V: nV x 3 row-major (device).  d_val[nnz] = mass_coef * M + lap_coef * L with L the (negative semi-definite) cotangent
 * matrix and M the lumped mass matrix (voronoi != 0: mixed Voronoi areas, else barycentric); d_mass[nV] (optional) = diag M;
 * d_Lval[nnz] (optional) = L alone.  hip_stream: stream to enqueue on (NULL = default stream). */
int smg_assemble(smg_assembler *a, const double *d_V, int voronoi, double mass_coef, double lap_coef, double *d_val,
                 double *d_mass, double *d_Lval, void *hip_stream);

/* ---- min_quad_with_fixed_mg_solve (src/min_quad_with_fixed_mg.h:38-69 and :79-113) ------------------------------ */
/* RHS, z0, z: n x k column-major (n = full size incl. known rows); known_val: n_known x k (ignored without
 * constraints).  r_his must hold opts->max_iter doubles (any max_iter >= 0, as in the reference, .cpp:77); *n_his <= max_iter entries are written, one per loop
 * entry incl. the one that triggers the break (.cpp:108-116).  *converged = !(last measured residual > tol)
 * (.cpp:131-134).  memspace: SMG_HOST or SMG_DEVICE for RHS/known_val/z0/z (r_his is always host). */
int smg_solve(smg_hierarchy *h, const double *RHS, int ld_rhs, const double *known_val, int ld_kv, const double *z0,
              int ld_z0, int k, int memspace, const smg_solve_opts *opts, double *z, int ld_z, double *r_his,
              int *n_his, int *converged);

/* Split-phase form of the same loop for column-sharded multi-GPU runs: the caller owns
 * the all-reduce of the residual sum of squares between the two halves of an iteration.
 *   begin:     gathers RHS/z0 (column-major) into the handle, resets the control block.  SMG_DEVICE: the gathers are ENQUEUED on the
 *              handle's stream and the call returns (stream-ordered, like every other entry point of the split-phase API: do not
 *              overwrite RHS / z0 from another stream before that work has run); SMG_HOST: the host blocks are consumed on return;
 *   residual:  *d_sumsq (device double) = sum over the local columns of |RHS_u - A_0 z_u|^2   (.cpp:110/:332);
 *   cycle:     r = sqrt(*d_sumsq) -> r_his, break test, then one V-cycle (skipped on the device once done);
 *   end:       scatters z, copies r_his back, reports convergence.
 * residual and cycle run as cached graphs that write / read *d_sumsq in place (no staging copy): hand the SAME device buffer to
 * both, every iteration (another buffer is honoured, at the price of re-capturing the two graphs).  NULL = the handle's own word. */
int smg_solve_begin(smg_hierarchy *h, const double *RHS, int ld_rhs, const double *known_val, int ld_kv,
                    const double *z0, int ld_z0, int k, int memspace, const smg_solve_opts *opts);
int smg_solve_iter_residual(smg_hierarchy *h, double *d_sumsq);
int smg_solve_iter_cycle(smg_hierarchy *h, const double *d_sumsq);
/* Latency-hiding form of `cycle`: the V-cycle of iteration i does not wait for the all-reduce of residual i.
 *   cycle_speculative:  saves the iterate, runs the V-cycle in place (independent of the pending reduction);
 *   commit(d_sumsq):    r = sqrt(*d_sumsq) -> r_his, break test; if THIS test ends the loop the saved iterate is restored.
 * Results (z, r_his, converged) are bit-identical to residual / cycle.  Usage per iteration:
 *   residual(d) ; work = all_reduce(d, async) ; cycle_speculative() ; work.wait() ; commit(d)                        */
int smg_solve_iter_cycle_speculative(smg_hierarchy *h);
int smg_solve_iter_commit(smg_hierarchy *h, const double *d_sumsq);
int smg_solve_poll(smg_hierarchy *h, int *done, int *n_his);      /* synchronising read of the control block */
int smg_solve_end(smg_hierarchy *h, double *z, int ld_z, int memspace, double *r_his, int *n_his, int *converged);

/* The column-sharded solve as ONE call (BASELINE north_star: "C++ host code ... RCCL over xGMI only for the
 * residual-norm all-reduce"): this rank owns k_local of the k right-hand-side columns (RHS / z0 / z / known_val hold just those), the
 * hierarchy is replicated, and the library runs the reference's loop (src/min_quad_with_fixed_mg.cpp:108-125)
 *     for (iter < maxIter) { r = |RHS - A z|_F over ALL ranks' columns; push; if (r < tol) break; V-cycle }
 * itself: residual graph -> reduce(d_sumsq, 1, stream, ctx) -> cycle graph, the break test on the device from the reduced value, so
 * every rank records the same history and stops at the same iteration.  The only communication is `reduce`:
 *     int reduce(double *d_sumsq, int count, void *hip_stream, void *ctx)
 * must leave the sum over all ranks of the `count` device doubles at d_sumsq in place, ordered on hip_stream (enqueue it there --
 * ncclAllReduce(d, d, count, ncclDouble, ncclSum, comm, (hipStream_t)hip_stream) is the whole closure, examples/
 * 05_mean_curvature_flow_sharded.cpp -- or synchronise the stream and do it on the host), and return 0; any other value aborts the
 * solve with SMG_ERR_REDUCE.  It is called exactly once per loop entry, the same number of times on every rank.
 * k_local == 0 is legal (more ranks than columns): the rank contributes 0 to every reduction and follows the others' decision;
 * RHS / z0 / z may then be NULL.  Everything else (arguments, r_his / n_his / converged, memspace) as in smg_solve.
 * With world size 1 and a reduce that does nothing this IS smg_solve's loop (same graphs' kernels, same bits). */
typedef int (*smg_reduce_fn)(double *d_sumsq, int count, void *hip_stream, void *ctx);
int smg_solve_sharded(smg_hierarchy *h, const double *RHS, int ld_rhs, const double *known_val, int ld_kv, const double *z0,
                      int ld_z0, int k_local, int memspace, const smg_solve_opts *opts, smg_reduce_fn reduce, void *ctx,
                      double *z, int ld_z, double *r_his, int *n_his, int *converged);

/* ---- mg_VCycle.h pieces, host column-major blocks in the level's own (caller) numbering ------------------------ */
/* Level lv has smg_level_rows(h, lv) unknowns (after constraint elimination). */
int smg_level_rows(const smg_hierarchy *h, int lv);
int smg_vcycle(smg_hierarchy *h, const double *B, int pre, int post, int lv, double *u, int k);   /* mg_VCycle.h:22-30 */
int smg_apply_A(smg_hierarchy *h, int lv, const double *u, int k, double *Au);                    /* A()       :32-37 */
int smg_restrict(smg_hierarchy *h, int lv, const double *x, int k, double *Rx);                   /* restrict  :39-44 */
int smg_prolong(smg_hierarchy *h, int lv, const double *x, int k, double *Px);                    /* prolong   :46-51 */
int smg_relax(smg_hierarchy *h, int lv, const double *B, int k, int iters, double *u);            /* relax     :62-68 */
int smg_coarse_solve(smg_hierarchy *h, const double *B, int k, double *u);                        /* coarseSolve :70-76 */
int smg_residual_norm(smg_hierarchy *h, int lv, const double *B, const double *u, int k, double *norm);

/* ---- device-resident raw interface (internal numbering / internal row-major n x k layout) ---------------------- */
/* For benchmarks and callers that keep everything in HBM.  x, y, b: device pointers, rows in the level's
 * internal order (smg_level_get_perm), k columns interleaved.  mode: 0 y=Ax, 1 y=b-Ax, 3 y+=Ax. */
int smg_raw_spmv(smg_hierarchy *h, int lv, int mode, const double *x, const double *b, double *y, int k);
int smg_raw_relax(smg_hierarchy *h, int lv, const double *b, double *u, int k, int iters);
/* fp32 twin of smg_raw_spmv (mode 0 only): x, y are float device vectors; bytes per launch 8 nnz + 4 (n+1) + 8 n k */
int smg_raw_spmv_f32(smg_hierarchy *h, int lv, const float *x, float *y, int k);
int smg_raw_outer_iteration(smg_hierarchy *h, int n_iter);   /* residual + decide + V-cycle, n_iter times, on the
                                                                state loaded by smg_solve_begin; no host sync */
int smg_synchronize(smg_hierarchy *h);
/* diagnostic: average time (us, hipEvents on the handle's stream) of one graph-replayed V(pre,post) cycle started at level
 * lv on whatever the work vectors hold; k columns.  Used to see where a cycle's time goes level by level. */
int smg_bench_vcycle(smg_hierarchy *h, int lv, int k, int pre, int post, int reps, double *us_per_cycle);
/* diagnostic: average time (us) of `sweeps` graph-replayed Gauss-Seidel sweeps on level lv */
int smg_bench_relax(smg_hierarchy *h, int lv, int k, int sweeps, int reps, double *us_per_call);

/* ---- introspection (tests, tools) ------------------------------------------------------------------------------ */
/* which: 0 = A, 1 = P (lv >= 1), 2 = PT (lv >= 1), 3 = P_full (lv >= 1), 4 = Auk (lv == 0).
 * internal = 0: caller numbering; 1: the device numbering.  Query sizes with NULL arrays first. */
int smg_level_get_matrix(const smg_hierarchy *h, int lv, int which, int internal, int *n_rows, int *n_cols, int *nnz,
                         int *rowptr, int *col, double *val);
int smg_level_get_perm(const smg_hierarchy *h, int lv, int *perm);              /* internal -> caller, n entries */
int smg_level_get_colors(const smg_hierarchy *h, int lv, int *n_colors, int *color_ptr /* n_colors+1 or NULL */);
int smg_level_get_Adiag(const smg_hierarchy *h, int lv, double *diag);          /* mg[lv].A_diag, caller numbering */
int smg_get_unknown(const smg_hierarchy *h, int *n_unknown, int *unknown /* or NULL */);
int smg_level_sell_stats(const smg_hierarchy *h, int lv, int which, long *stored, long *padded, int *n_slices);
/* Rows of the first colour of level lv's Gauss-Seidel image whose diagonal slots the device knows (scalar hierarchies): > 0 means the
 * restriction launch of level lv - 1 can produce the first colour of this level's first sweep itself (one launch less per visit);
 * 0: not available (uncoloured / coarsest level, a row of the first colour without a stored diagonal). */
int smg_level_first_colour_rows(const smg_hierarchy *h, int lv);
/* algorithmic bytes of one y = A_lv x with k columns: 12 nnz + 4 (n+1) + 16 n k  (SURVEY.md section 8d); on a block hierarchy
 * 76 per 3 x 3 block (72 of values + 4 of block column) + 4 (n/3 + 1) + 16 n k */
long smg_level_spmv_bytes(const smg_hierarchy *h, int lv, int k);
/* algorithmic bytes of one V(pre,post) cycle incl. the outer residual evaluation, k columns -- of the cycle the handle is set to run:
 * a Chebyshev-Jacobi relax(iters) is iters + 1 passes over the level matrix, a Gauss-Seidel / Jacobi one iters passes */
long smg_vcycle_bytes(const smg_hierarchy *h, int k, int pre, int post);

/* ---- host-side self-checks (tests on boxes without a GPU; they return diagnostics only, never a solution: no CPU solve path) -------- */
/* The overlapped-tiling plan of relax(sweeps) on level lv (csrc/smg_tiled.hpp), executed on the HOST exactly as the kernel executes it
 * (tile by tile, phase by phase, tile-local numbering) on a deterministic test vector, against the plain colour-by-colour sweeps on the
 * level's matrix in the internal numbering: *max_abs_diff must be exactly 0.  Needs the host half of smg_precompute only.
 * *n_tiles = 0: the level does not qualify for tiling (too many colours, rows wider than 12 entries). */
int smg_debug_check_tiling_plan(smg_hierarchy *h, int lv, int sweeps, int tile_rows, int *n_tiles, int *max_ext_rows, double *redundancy,
                                double *max_abs_diff);
/* Test hook: builds the block Gauss-Seidel plan of level lv (smg_hierarchy_set_block_gs) on the host and executes it on the host the way the
 * kernel does, against the plain lexicographic sweep in the block order; *max_abs_diff must be 0.  Checks the plan's invariants on the way.
 * Needs no GPU (after the host half of smg_precompute).  *n_blocks = 0: the level does not qualify. */
int smg_debug_check_block_gs_plan(smg_hierarchy *h, int lv, int block_rows, int *n_blocks, int *n_colors, double *rim, double *fill, double *max_abs_diff);
/* Test hook: builds the wave Gauss-Seidel plan of level lv (smg_hierarchy_set_wave_gs; pieces_mode 0 compact pieces, 1 pieces along breadth-first level
 * sets) on the host and executes it on the host the way the kernel does, against the plain lexicographic sweep in the piece order; *max_abs_diff must be 0.
 * stats[3] as smg_level_get_wave_gs_order.  Checks the plan's invariants on the way.  Needs no GPU (after the host half of smg_precompute).
 * *n_pieces = 0: the level does not qualify (a row of more than 64 off-diagonal entries). */
int smg_debug_check_wave_gs_plan(smg_hierarchy *h, int lv, int piece_rows, int pieces_mode, int *n_pieces, int *n_colors, double *stats, double *max_abs_diff);
/* Test hook: raises the stall flag of the sparse triangular solves on the device, as a wait that gave up would (csrc/smg_coarse_device.hip).
 * The next solve's waits then give up at once, its coarse corrections are NaN, and the next synchronising entry point returns SMG_ERR_HIP
 * and clears the flag.  Fails unless the handle holds a sparse coarse factorisation. */
int smg_debug_raise_coarse_stall(smg_hierarchy *h);
/* Test hook: the plan of the Schur-complement coarse solver built for the SPD matrix (ptr, col, val; lower triangle counts) and executed ON THE HOST
 * the way the kernels read it: x = A^-1 b.  *n_blocks = 0: no plan for this matrix (x untouched).  Needs no GPU. */
int smg_debug_schur_solve_host(int n, const int *ptr, const int *col, const double *val, const double *b, double *x, int *n_blocks, int *n_sep);
/* Sparse Cholesky of the coarse solver (csrc/smg_coarse.hpp) on an SPD matrix given in CSR (both triangles): nested-dissection order,
 * factorisation, and the relative residual |b - A x| / |b| of a host solve with the factor for a deterministic right-hand side.
 * Returns SMG_ERR_INVALID when a pivot is not positive. */
int smg_debug_check_sparse_cholesky(int n, const int *rowptr, const int *col, const double *val, long *factor_entries, int *dependency_depth,
                                    double *rel_residual);

/* ---- profc.h mirror: named scopes accumulated with hipEvents (src/profc.h:9-13; mg_VCycle.cpp:121) ------------- */
int smg_prof_enable(smg_hierarchy *h, int on);     /* forces eager launches while on */
int smg_prof_reset(smg_hierarchy *h);
int smg_prof_count(smg_hierarchy *h);
int smg_prof_get(smg_hierarchy *h, int idx, char *name, int name_cap, long *count, double *total_ms);

/* ---- caller-side mesh numerics (host C++; libigl stand-ins used by the demos around the solve) ----------------- */
int smg_mesh_read(const char *path, double **V, int *nV, int **F, int *nF);   /* free with smg_free */
void smg_free(void *p);
int smg_mesh_normalize_unit_area(double *V, int nV, const int *F, int nF);        /* src/normalize_unit_area.cpp */
/* cotmatrix (negative semi-definite, igl convention).  Query nnz with NULL arrays, then fill. */
int smg_mesh_cotmatrix(const double *V, int nV, const int *F, int nF, int *nnz, int *rowptr, int *col, double *val);
int smg_mesh_massmatrix(const double *V, int nV, const int *F, int nF, int voronoi, double *diag);
int smg_mesh_boundary_loop(const int *F, int nF, int nV, int *loop, int *n_loop); /* longest loop; loop holds <= nV */
/* one mid-point upsampling step: S is (nV+nE) x nV in CSR (nV + 2 nE entries), NF is 4 nF x 3 */
int smg_mesh_midpoint_upsample(int nV, const int *F, int nF, int *nE, int *S_rowptr, int *S_col, double *S_val,
                               int *NF);
int smg_mesh_torus(int nu, int nv, double R, double r, double *V, int *F);    /* V: nu*nv x 3, F: 2*nu*nv x 3 */

#ifdef __cplusplus
}
#endif
#endif
