// mg_api.hpp -- C++ host mirror of the reference's operator API for the solve path, header-only over the C ABI
// (include/smg.h).  Same function names, argument order and return conventions as the reference:
//
//   mg_precompute(V, F, ratio, nVCoarsest, dec_type, mg)                     reference src/mg_precompute.h:26-32
//   min_quad_with_fixed_mg_precompute(A, [known,] data, mg, solver)          reference src/min_quad_with_fixed_mg.h:32-36, :72-77
//   min_quad_with_fixed_mg_solve(data, RHS, [known_val,] z0, solver, [tol, [maxIter,]] mg, z, r_his) -> bool
//                                                                            reference src/min_quad_with_fixed_mg.h:38-69, :79-113
//   mg_VCycle(solver, B, pre, post, lv, u, mg)                               reference src/mg_VCycle.h:22-30
//
// The reference passes Eigen objects; Eigen is not a dependency of libsmg, so this header uses the plain containers
// below, which have Eigen's memory layout (column-major dense, compressed-column sparse, int32 indices): an Eigen
// build maps them without copies (INTEGRATION.md shows the adapter).  `mg_data` keeps the reference's fields
// (src/mg_data.h:11-19); the device images live behind `solver` (the stand-in for Eigen::SimplicialLDLT), which
// like in the reference is a caller-owned object filled by ..._precompute and consumed by ..._solve / mg_VCycle.
#pragma once
#include <cstdint>
#include <cstdio>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/smg.h"

struct smgDense {  // Eigen::MatrixXd / VectorXd layout
    int rows = 0, cols = 0;
    std::vector<double> data;  // column-major
    smgDense() = default;
    smgDense(int r, int c, double v = 0.0) : rows(r), cols(c), data((size_t)r * c, v) {}
    void resize(int r, int c) { rows = r; cols = c; data.assign((size_t)r * c, 0.0); }
    void setZero() { data.assign(data.size(), 0.0); }
    double& operator()(int i, int j = 0) { return data[(size_t)j * rows + i]; }
    double operator()(int i, int j = 0) const { return data[(size_t)j * rows + i]; }
};
struct smgDenseI {  // Eigen::MatrixXi / VectorXi layout
    int rows = 0, cols = 0;
    std::vector<int> data;
    smgDenseI() = default;
    smgDenseI(int r, int c) : rows(r), cols(c), data((size_t)r * c, 0) {}
    void resize(int r, int c) { rows = r; cols = c; data.assign((size_t)r * c, 0); }
    int& operator()(int i, int j = 0) { return data[(size_t)j * rows + i]; }
    int operator()(int i, int j = 0) const { return data[(size_t)j * rows + i]; }
    int size() const { return rows * cols; }
};
struct smgSparse {  // Eigen::SparseMatrix<double> (ColMajor, int): outerIndexPtr / innerIndexPtr / valuePtr
    int rows = 0, cols = 0;
    std::vector<int> outer, inner;
    std::vector<double> values;
    int nonZeros() const { return outer.empty() ? 0 : outer.back(); }
};

struct mg_data {  // reference src/mg_data.h:11-19
    smgDense V;
    smgDenseI F;
    smgSparse P_full, A;
    std::vector<double> A_diag;
    smgSparse P, PT;
};

struct smgCoarseSolver {  // stands in for Eigen::SimplicialLDLT<Eigen::SparseMatrix<double>>
    std::shared_ptr<smg_hierarchy> h;
    // what the reference hard-codes in its solve (pre = post = 2, Gauss-Seidel) -- and the one place to opt into libsmg's extensions,
    // e.g.  coarseSolver.opts.smoother = SMG_SMOOTH_HYBRID_CHEBYSHEV; coarseSolver.opts.jacobi_max_rows = 300000;
    // tol / max_iter are overwritten by the solve overloads' arguments.
    smg_solve_opts opts;
    // Column-sharded multi-GPU runs: every rank calls the solve with ITS columns of RHS / z0 and sets `reduce` to
    // the all-reduce of the residual sum of squares (include/smg.h: smg_reduce_fn; ncclAllReduce on an ncclComm_t the caller created,
    // examples/05_mean_curvature_flow_sharded.cpp).  nullptr (default): the single-GPU loop.
    smg_reduce_fn reduce = nullptr;
    void* reduce_ctx = nullptr;
    smgCoarseSolver() { smg_solve_opts_default(&opts); }
};

struct min_quad_with_fixed_mg_data {  // reference src/min_quad_with_fixed_mg.h:22-29
    int n = 0;
    std::vector<int> known, unknown;
    smgSparse LHS, Auk;
};

namespace smg_detail {
inline void check(int rc, const char* what)
{
    if (rc != SMG_OK) throw std::runtime_error(std::string(what) + ": " + smg_last_error());
}
// CSR of M (what the C ABI returns) -> compressed columns, by reading the CSR of M^T
inline smgSparse fetch_csc(const smg_hierarchy* h, int lv, int which_T, int which, int rows, int cols)
{
    (void)which;
    smgSparse S;
    int nr = 0, nc = 0, nnz = 0;
    check(smg_level_get_matrix(h, lv, which_T, 0, &nr, &nc, &nnz, nullptr, nullptr, nullptr), "smg_level_get_matrix");
    S.rows = rows; S.cols = cols;
    S.outer.resize((size_t)nr + 1); S.inner.resize(nnz > 0 ? nnz : 1); S.values.resize(nnz > 0 ? nnz : 1);
    check(smg_level_get_matrix(h, lv, which_T, 0, nullptr, nullptr, nullptr, S.outer.data(), S.inner.data(), S.values.data()), "smg_level_get_matrix");
    S.inner.resize(nnz); S.values.resize(nnz);
    return S;
}
inline void sync_levels(const smg_hierarchy* h, std::vector<mg_data>& mg, bool matrices)
{
    const int L = smg_hierarchy_levels(h);
    mg.resize(L);
    for (int lv = 0; lv < L; lv++) {
        int nV = 0, nF = 0;
        smg_level_get_mesh(h, lv, &nV, &nF, nullptr, nullptr);
        if (nV > 0 && mg[lv].V.rows != nV) {
            std::vector<double> V((size_t)nV * 3); std::vector<int> F((size_t)nF * 3);
            smg_level_get_mesh(h, lv, nullptr, nullptr, V.data(), F.data());
            mg[lv].V.resize(nV, 3); mg[lv].F.resize(nF, 3);
            for (int i = 0; i < nV; i++) for (int c = 0; c < 3; c++) mg[lv].V(i, c) = V[3 * (size_t)i + c];
            for (int i = 0; i < nF; i++) for (int c = 0; c < 3; c++) mg[lv].F(i, c) = F[3 * (size_t)i + c];
        }
        if (lv >= 1) {
            int nr = 0, nc = 0;  // current (possibly constraint-sliced) shape of P
            smg_level_get_matrix(h, lv, 1, 0, &nr, &nc, nullptr, nullptr, nullptr, nullptr);
            // CSC of P_full == CSR of P_full^T: the handle keeps P (1) and PT (2) explicitly; before precompute
            // P == P_full, so PT's CSR arrays are P_full's CSC arrays
            mg[lv].P = fetch_csc(h, lv, 2, 1, nr, nc);
            mg[lv].PT = fetch_csc(h, lv, 1, 2, nc, nr);
            if (!matrices) mg[lv].P_full = mg[lv].P;
        }
        if (matrices) {
            const int n = smg_level_rows(h, lv);
            mg[lv].A = fetch_csc(h, lv, 0, 0, n, n);  // symmetric up to rounding: rows of A are what A*x uses
            mg[lv].A_diag.resize(n);
            smg_level_get_Adiag(h, lv, mg[lv].A_diag.data());
        }
    }
}
}  // namespace smg_detail

// ---- mg_precompute (reference src/mg_precompute.cpp:15-87; default-argument overloads :89-106) -----------------
inline void mg_precompute(const smgDense& Vf, const smgDenseI& Ff, const float& ratio, const int& nVCoarsest,
                          const int& dec_type, std::vector<mg_data>& mg)
{
    std::vector<double> V((size_t)Vf.rows * 3);
    std::vector<int> F((size_t)Ff.rows * 3);
    for (int i = 0; i < Vf.rows; i++) for (int c = 0; c < 3; c++) V[3 * (size_t)i + c] = Vf(i, c);
    for (int i = 0; i < Ff.rows; i++) for (int c = 0; c < 3; c++) F[3 * (size_t)i + c] = Ff(i, c);
    smg_hierarchy* h = nullptr;
    smg_detail::check(smg_mg_precompute(V.data(), Vf.rows, F.data(), Ff.rows, ratio, nVCoarsest, dec_type, &h), "mg_precompute");
    std::shared_ptr<smg_hierarchy> guard(h, smg_hierarchy_destroy);
    mg.clear();
    smg_detail::sync_levels(h, mg, false);
    std::printf("============\nMultigrid Info\n============\nnumLv: %d\n|V_coarsest|: %d\n", (int)mg.size(), mg.back().V.rows);
}
inline void mg_precompute(const smgDense& Vf, const smgDenseI& Ff, const int& dec_type, std::vector<mg_data>& mg)
{
    mg_precompute(Vf, Ff, 0.25f, 500, dec_type, mg);  // :104-105
}
inline void mg_precompute(const smgDense& Vf, const smgDenseI& Ff, std::vector<mg_data>& mg)
{
    mg_precompute(Vf, Ff, 1, mg);  // default: mid-point decimation (:94)
}

// ---- min_quad_with_fixed_mg_precompute --------------------------------------------------------------------------
namespace smg_detail {
inline void precompute_impl(const smgSparse& A, const int* known, int n_known, min_quad_with_fixed_mg_data& data,
                            std::vector<mg_data>& mg, smgCoarseSolver& solver)
{
    const int L = (int)mg.size();
    if (!solver.h || smg_hierarchy_levels(solver.h.get()) != L) {
        solver.h.reset(smg_hierarchy_create(L), smg_hierarchy_destroy);
        if (!solver.h) throw std::runtime_error(smg_last_error());
    }
    smg_hierarchy* h = solver.h.get();
    for (int lv = 1; lv < L; lv++) {
        const smgSparse& P = mg[lv].P_full;
        check(smg_level_set_prolong_csc(h, lv, P.rows, P.cols, P.outer.data(), P.inner.data(), P.values.data()), "smg_level_set_prolong_csc");
    }
    // A symmetric: its compressed-column arrays are its CSR arrays
    check(smg_precompute(h, A.rows, A.outer.data(), A.inner.data(), A.values.data(), known, n_known), "min_quad_with_fixed_mg_precompute");
    data.n = A.rows;
    data.known.assign(known, known + n_known);
    int nu = 0;
    smg_get_unknown(h, &nu, nullptr);
    data.unknown.resize(nu);
    smg_get_unknown(h, nullptr, data.unknown.data());
    sync_levels(h, mg, true);          // the reference mutates mg[l].A / A_diag / P / PT in place
    data.LHS = mg[0].A;
}
}  // namespace smg_detail

inline void min_quad_with_fixed_mg_precompute(const smgSparse& A, min_quad_with_fixed_mg_data& data, std::vector<mg_data>& mg,
                                              smgCoarseSolver& solver)
{
    smg_detail::precompute_impl(A, nullptr, 0, data, mg, solver);
}
inline void min_quad_with_fixed_mg_precompute(const smgSparse& A, const smgDenseI& known, min_quad_with_fixed_mg_data& data,
                                              std::vector<mg_data>& mg, smgCoarseSolver& solver)
{
    smg_detail::precompute_impl(A, known.data.data(), known.size(), data, mg, solver);
}

// ---- min_quad_with_fixed_mg_solve ---------------------------------------------------------------------------------
namespace smg_detail {
inline bool solve_impl(const smgDense& RHS, const smgDense* known_val, const smgDense& z0, const smgCoarseSolver& solver,
                       double tolerance, int maxIter, smgDense& z, std::vector<double>& r_his)
{
    smg_solve_opts o = solver.opts;
    o.tol = tolerance; o.max_iter = maxIter;
    z.resize(z0.rows, z0.cols);
    r_his.assign((size_t)(maxIter > 0 ? maxIter : 1), 0.0);
    int n_his = 0, conv = 0;
    if (solver.reduce)   // this rank's columns of a column-sharded solve; the library runs the loop and calls the reduction
        check(smg_solve_sharded(solver.h.get(), RHS.data.data(), RHS.rows, known_val ? known_val->data.data() : nullptr,
                                known_val ? known_val->rows : 0, z0.data.data(), z0.rows, RHS.cols, SMG_HOST, &o, solver.reduce, solver.reduce_ctx,
                                z.data.data(), z.rows, r_his.data(), &n_his, &conv), "min_quad_with_fixed_mg_solve (sharded)");
    else
    check(smg_solve(solver.h.get(), RHS.data.data(), RHS.rows, known_val ? known_val->data.data() : nullptr,
                    known_val ? known_val->rows : 0, z0.data.data(), z0.rows, RHS.cols, SMG_HOST, &o, z.data.data(), z.rows,
                    r_his.data(), &n_his, &conv), "min_quad_with_fixed_mg_solve");
    r_his.resize(n_his);
    for (int i = 0; i < n_his; i++) std::printf("MG iteration: %d, residual: %g\n", i, r_his[i]);  // .cpp:111
    if (n_his) std::printf("residual norm: %g\n", r_his.back());                                      // .cpp:127
    return conv != 0;
}
}  // namespace smg_detail

inline bool min_quad_with_fixed_mg_solve(const min_quad_with_fixed_mg_data&, const smgDense& RHS, const smgDense& z0,
                                         const smgCoarseSolver& solver, const double& tolerance, const int& maxIter,
                                         std::vector<mg_data>&, smgDense& z, std::vector<double>& r_his)
{
    return smg_detail::solve_impl(RHS, nullptr, z0, solver, tolerance, maxIter, z, r_his);
}
inline bool min_quad_with_fixed_mg_solve(const min_quad_with_fixed_mg_data& d, const smgDense& RHS, const smgDense& z0,
                                         const smgCoarseSolver& solver, const double& tolerance, std::vector<mg_data>& mg,
                                         smgDense& z, std::vector<double>& r_his)
{
    return min_quad_with_fixed_mg_solve(d, RHS, z0, solver, tolerance, 20, mg, z, r_his);  // .cpp:77
}
inline bool min_quad_with_fixed_mg_solve(const min_quad_with_fixed_mg_data& d, const smgDense& RHS, const smgDense& z0,
                                         const smgCoarseSolver& solver, std::vector<mg_data>& mg, smgDense& z,
                                         std::vector<double>& r_his)
{
    return min_quad_with_fixed_mg_solve(d, RHS, z0, solver, 1e-3, mg, z, r_his);  // .cpp:63
}
inline bool min_quad_with_fixed_mg_solve(const min_quad_with_fixed_mg_data&, const smgDense& RHS, const smgDense& known_val,
                                         const smgDense& z0, const smgCoarseSolver& solver, const double& tolerance,
                                         const int& maxIter, std::vector<mg_data>&, smgDense& z, std::vector<double>& r_his)
{
    return smg_detail::solve_impl(RHS, &known_val, z0, solver, tolerance, maxIter, z, r_his);
}
inline bool min_quad_with_fixed_mg_solve(const min_quad_with_fixed_mg_data& d, const smgDense& RHS, const smgDense& known_val,
                                         const smgDense& z0, const smgCoarseSolver& solver, const double& tolerance,
                                         std::vector<mg_data>& mg, smgDense& z, std::vector<double>& r_his)
{
    return min_quad_with_fixed_mg_solve(d, RHS, known_val, z0, solver, tolerance, 20, mg, z, r_his);  // .cpp:285
}
inline bool min_quad_with_fixed_mg_solve(const min_quad_with_fixed_mg_data& d, const smgDense& RHS, const smgDense& known_val,
                                         const smgDense& z0, const smgCoarseSolver& solver, std::vector<mg_data>& mg,
                                         smgDense& z, std::vector<double>& r_his)
{
    return min_quad_with_fixed_mg_solve(d, RHS, known_val, z0, solver, 1e-3, mg, z, r_his);  // .cpp:270
}

// ---- mg_VCycle (reference src/mg_VCycle.cpp:3-59) ----------------------------------------------------------------
inline void mg_VCycle(const smgCoarseSolver& solver, const smgDense& B, const int& preRelaxIter, const int& postRelaxIter,
                      const int lv, smgDense& u, std::vector<mg_data>&)
{
    smg_detail::check(smg_hierarchy_set_smoother(solver.h.get(), solver.opts.smoother, solver.opts.omega, solver.opts.jacobi_max_rows), "smg_hierarchy_set_smoother");
    smg_detail::check(smg_hierarchy_set_chebyshev(solver.h.get(), solver.opts.cheby_fraction), "smg_hierarchy_set_chebyshev");
    smg_detail::check(smg_vcycle(solver.h.get(), B.data.data(), preRelaxIter, postRelaxIter, lv, u.data.data(), B.cols), "mg_VCycle");
}
