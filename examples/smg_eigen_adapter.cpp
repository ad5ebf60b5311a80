// smg_eigen_adapter.cpp -- the reference's solve-path functions, with the reference's own Eigen signatures, on libsmg.
//
// Drop-in for the two translation units src/min_quad_with_fixed_mg.cpp and src/mg_VCycle.cpp of HTDerekLiu/surface_multigrid_code
// (INTEGRATION.md): compile this file instead of them, against the reference's own headers (-I<reference>/src, Eigen, libigl),
// and link libsmg.so.  Callers (03_mg_solver/main.cpp:71,75; 04_mg_solver_nobd/main.cpp:100,105;
// 05_example_mean_curvature_flow/main.cpp:74,76; 06_example_balloon_sim/implicit_euler_mg_balloon.h:75-76) stay as they are.
//
// What the reference's functions do to their arguments is reproduced, not only their results:
//   precompute  fills data.n / known / unknown / LHS / Auk (src/min_quad_with_fixed_mg.cpp:156-179) and MUTATES
//               mg[l].A, A_diag, P, PT (.cpp:22,25,40,185,211,214,226-227) -- read back from the handle;
//   solve       returns `!(residual > tol)`, fills r_his with one entry per loop entry, prints the reference's lines (.cpp:111,127).
// The device-resident state lives in an smg_hierarchy keyed by the address of the caller's std::vector<mg_data> (every function
// of the path receives it); smg_eigen_adapter_release(mg) drops it when the vector goes away.
//
// This image has no Eigen: -DSMG_ADAPTER_MOCK compiles the file against tests/mock_eigen (a stand-in with the members used
// here) -- a syntax / plumbing check of the adapter, NOT the reference compiled.
#ifdef SMG_ADAPTER_MOCK
#include "min_quad_with_fixed_mg.h"   // tests/mock_eigen/: the two structs only
#else
#include <min_quad_with_fixed_mg.h>   // the reference's own headers: signatures stay as they are
#include <mg_VCycle.h>
#endif
#include <smg.h>

#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

typedef Eigen::SimplicialLDLT<Eigen::SparseMatrix<double>> SmgLDLT;
typedef Eigen::SparseMatrix<double> SmgSpMat;

namespace {
std::map<const void*, std::shared_ptr<smg_hierarchy>>& handles()
{
    static std::map<const void*, std::shared_ptr<smg_hierarchy>> m;
    return m;
}
void check(int rc, const char* what)
{
    if (rc != SMG_OK) throw std::runtime_error(std::string(what) + ": " + smg_last_error());
}
smg_hierarchy* handle_of(const std::vector<mg_data>& mg)
{
    auto it = handles().find(&mg);
    if (it == handles().end()) throw std::runtime_error("min_quad_with_fixed_mg_precompute has not been called for this mg");
    return it->second.get();
}
// CSR of M^T as libsmg returns it == compressed columns of M
SmgSpMat fetch(const smg_hierarchy* h, int lv, int which_of_transpose, long rows, long cols)
{
    int nr = 0, nc = 0, nnz = 0;
    check(smg_level_get_matrix(h, lv, which_of_transpose, 0, &nr, &nc, &nnz, nullptr, nullptr, nullptr), "smg_level_get_matrix");
    std::vector<int> ptr((size_t)nr + 1), idx((size_t)(nnz > 0 ? nnz : 1));
    std::vector<double> val((size_t)(nnz > 0 ? nnz : 1));
    check(smg_level_get_matrix(h, lv, which_of_transpose, 0, nullptr, nullptr, nullptr, ptr.data(), idx.data(), val.data()), "smg_level_get_matrix");
    SmgSpMat M;
    M = Eigen::Map<const SmgSpMat>(rows, cols, nnz, ptr.data(), idx.data(), val.data());
    return M;
}

void precompute(const SmgSpMat& A_in, const int* known, int n_known, min_quad_with_fixed_mg_data& data, std::vector<mg_data>& mg)
{
    const int L = (int)mg.size();
    std::shared_ptr<smg_hierarchy>& slot = handles()[&mg];
    if (!slot || smg_hierarchy_levels(slot.get()) != L) {
        slot.reset(smg_hierarchy_create(L), smg_hierarchy_destroy);
        if (!slot) throw std::runtime_error(smg_last_error());
        for (int lv = 1; lv < L; lv++) {      // mg[lv].P_full as mg_precompute left it (src/mg_precompute.cpp:76): Eigen's CSC arrays go in as they are
            SmgSpMat P = mg[lv].P_full;
            P.makeCompressed();
            check(smg_level_set_prolong_csc(slot.get(), lv, (int)P.rows(), (int)P.cols(), P.outerIndexPtr(), P.innerIndexPtr(), P.valuePtr()),
                  "smg_level_set_prolong_csc");
        }
    }
    smg_hierarchy* h = slot.get();
    SmgSpMat A = A_in;
    A.makeCompressed();                        // symmetric: the compressed-column arrays are the CSR arrays
    check(smg_precompute(h, (int)A.rows(), A.outerIndexPtr(), A.innerIndexPtr(), A.valuePtr(), known, n_known), "smg_precompute");
    // ---- min_quad_with_fixed_mg_data (.cpp:17-22 / :156-179)
    data.n = (int)A.rows();
    int nu = 0;
    check(smg_get_unknown(h, &nu, nullptr), "smg_get_unknown");
    data.unknown.resize(nu);
    check(smg_get_unknown(h, nullptr, data.unknown.data()), "smg_get_unknown");
    data.known.resize(n_known);
    for (int i = 0; i < n_known; i++) data.known(i) = known[i];
    // ---- what precompute leaves in mg (.cpp:22-41 / :223-246): A, A_diag on every level, P / PT (constraint-sliced) for lv >= 1
    for (int lv = 0; lv < L; lv++) {
        const int n = smg_level_rows(h, lv);
        mg[lv].A = fetch(h, lv, 0, n, n);      // A_lv is symmetric up to rounding; its rows are what A * x uses
        mg[lv].A_diag.resize(n);
        check(smg_level_get_Adiag(h, lv, mg[lv].A_diag.data()), "smg_level_get_Adiag");
        if (lv >= 1) {
            const int nf = smg_level_rows(h, lv - 1);
            mg[lv].P = fetch(h, lv, 2, nf, n);     // CSC of P  == CSR of PT
            mg[lv].PT = fetch(h, lv, 1, n, nf);    // CSC of PT == CSR of P
        }
    }
    data.LHS = mg[0].A;                        // LHS = A(unknown, unknown) (.cpp:167 / :175)
    if (n_known > 0) {
        // Auk = A(unknown, known) (.cpp:170 / :176): the handle holds its CSR; compressed columns by a counting transpose
        int nr = 0, nc = 0, nnz = 0;
        check(smg_level_get_matrix(h, 0, 4, 0, &nr, &nc, &nnz, nullptr, nullptr, nullptr), "smg_level_get_matrix");
        std::vector<int> rp((size_t)nr + 1), ci((size_t)(nnz > 0 ? nnz : 1)), cp((size_t)nc + 1, 0), ri((size_t)(nnz > 0 ? nnz : 1));
        std::vector<double> v((size_t)(nnz > 0 ? nnz : 1)), cv((size_t)(nnz > 0 ? nnz : 1));
        check(smg_level_get_matrix(h, 0, 4, 0, nullptr, nullptr, nullptr, rp.data(), ci.data(), v.data()), "smg_level_get_matrix");
        for (int p = 0; p < nnz; p++) cp[(size_t)ci[p] + 1]++;
        for (int c = 0; c < nc; c++) cp[(size_t)c + 1] += cp[c];
        std::vector<int> fill(cp.begin(), cp.end() - 1);
        for (int r = 0; r < nr; r++)
            for (int p = rp[r]; p < rp[r + 1]; p++) { const int q = fill[ci[p]]++; ri[q] = r; cv[q] = v[p]; }
        data.Auk = Eigen::Map<const SmgSpMat>(nr, nc, nnz, cp.data(), ri.data(), cv.data());
    } else data.Auk = SmgSpMat();
}

template <typename DRHS, typename DZ0, typename DZ>
bool solve(const double* known_val, int ld_kv, const Eigen::PlainObjectBase<DRHS>& RHS, const Eigen::PlainObjectBase<DZ0>& z0,
           double tolerance, int maxIter, std::vector<mg_data>& mg, Eigen::PlainObjectBase<DZ>& z, std::vector<double>& r_his)
{
    smg_hierarchy* h = handle_of(mg);
    smg_solve_opts o;
    smg_solve_opts_default(&o);                // pre = post = 2 (.cpp:102-103), Gauss-Seidel: the reference's cycle
    o.tol = tolerance;
    o.max_iter = maxIter;
    o.verbosity = 1;                           // "MG iteration: i, residual: r" / "residual norm: r" (.cpp:111,127)
    z.resize(z0.rows(), z0.cols());
    r_his.assign((size_t)(maxIter > 0 ? maxIter : 1), 0.0);
    int n_his = 0, conv = 0;                   // MatrixXd / VectorXd are column-major: leading dimension = rows()
    check(smg_solve(h, RHS.derived().data(), (int)RHS.rows(), known_val, ld_kv, z0.derived().data(), (int)z0.rows(), (int)RHS.cols(), SMG_HOST, &o,
                    z.derived().data(), (int)z.rows(), r_his.data(), &n_his, &conv), "smg_solve");
    r_his.resize((size_t)n_his);               // one entry per loop entry incl. the one that breaks (.cpp:108-116)
    return conv != 0;                          // !(residual > tol) (.cpp:131-134)
}
}  // namespace

// drop the device-resident state of `mg` (the reference has nothing to release: its state lives in the caller's objects)
void smg_eigen_adapter_release(const std::vector<mg_data>& mg) { handles().erase(&mg); }

// ---------------------------------------------------------------- min_quad_with_fixed_mg.h:32-36 / :72-77
void min_quad_with_fixed_mg_precompute(const SmgSpMat& A, min_quad_with_fixed_mg_data& data, std::vector<mg_data>& mg, SmgLDLT&)
{
    precompute(A, nullptr, 0, data, mg);
}
void min_quad_with_fixed_mg_precompute(const SmgSpMat& A, const Eigen::VectorXi& known, min_quad_with_fixed_mg_data& data, std::vector<mg_data>& mg, SmgLDLT&)
{
    precompute(A, known.data(), (int)known.size(), data, mg);
}

// ---------------------------------------------------------------- min_quad_with_fixed_mg.h:38-69 (no constraints)
template <typename DerivedRHS, typename DerivedZ0, typename DerivedZ>
bool min_quad_with_fixed_mg_solve(const min_quad_with_fixed_mg_data&, const Eigen::PlainObjectBase<DerivedRHS>& RHS,
                                  const Eigen::PlainObjectBase<DerivedZ0>& z0, const SmgLDLT&, const double& tolerance, const int& maxIter,
                                  std::vector<mg_data>& mg, Eigen::PlainObjectBase<DerivedZ>& z, std::vector<double>& r_his)
{
    return solve(nullptr, 0, RHS, z0, tolerance, maxIter, mg, z, r_his);
}
template <typename DerivedRHS, typename DerivedZ0, typename DerivedZ>
bool min_quad_with_fixed_mg_solve(const min_quad_with_fixed_mg_data& data, const Eigen::PlainObjectBase<DerivedRHS>& RHS,
                                  const Eigen::PlainObjectBase<DerivedZ0>& z0, const SmgLDLT& solver, const double& tolerance,
                                  std::vector<mg_data>& mg, Eigen::PlainObjectBase<DerivedZ>& z, std::vector<double>& r_his)
{
    return min_quad_with_fixed_mg_solve(data, RHS, z0, solver, tolerance, 20, mg, z, r_his);     // maxIter = 20 (.cpp:77)
}
template <typename DerivedRHS, typename DerivedZ0, typename DerivedZ>
bool min_quad_with_fixed_mg_solve(const min_quad_with_fixed_mg_data& data, const Eigen::PlainObjectBase<DerivedRHS>& RHS,
                                  const Eigen::PlainObjectBase<DerivedZ0>& z0, const SmgLDLT& solver, std::vector<mg_data>& mg,
                                  Eigen::PlainObjectBase<DerivedZ>& z, std::vector<double>& r_his)
{
    return min_quad_with_fixed_mg_solve(data, RHS, z0, solver, 1e-3, mg, z, r_his);              // tolerance = 1e-3 (.cpp:63)
}

// ---------------------------------------------------------------- min_quad_with_fixed_mg.h:79-113 (with known_val)
template <typename DerivedRHS, typename DerivedKnownVal, typename DerivedZ0, typename DerivedZ>
bool min_quad_with_fixed_mg_solve(const min_quad_with_fixed_mg_data&, const Eigen::PlainObjectBase<DerivedRHS>& RHS,
                                  const Eigen::PlainObjectBase<DerivedKnownVal>& known_val, const Eigen::PlainObjectBase<DerivedZ0>& z0,
                                  const SmgLDLT&, const double& tolerance, const int& maxIter, std::vector<mg_data>& mg,
                                  Eigen::PlainObjectBase<DerivedZ>& z, std::vector<double>& r_his)
{
    return solve(known_val.derived().data(), (int)known_val.rows(), RHS, z0, tolerance, maxIter, mg, z, r_his);
}
template <typename DerivedRHS, typename DerivedKnownVal, typename DerivedZ0, typename DerivedZ>
bool min_quad_with_fixed_mg_solve(const min_quad_with_fixed_mg_data& data, const Eigen::PlainObjectBase<DerivedRHS>& RHS,
                                  const Eigen::PlainObjectBase<DerivedKnownVal>& known_val, const Eigen::PlainObjectBase<DerivedZ0>& z0,
                                  const SmgLDLT& solver, const double& tolerance, std::vector<mg_data>& mg,
                                  Eigen::PlainObjectBase<DerivedZ>& z, std::vector<double>& r_his)
{
    return min_quad_with_fixed_mg_solve(data, RHS, known_val, z0, solver, tolerance, 20, mg, z, r_his);   // maxIter = 20 (.cpp:285)
}
template <typename DerivedRHS, typename DerivedKnownVal, typename DerivedZ0, typename DerivedZ>
bool min_quad_with_fixed_mg_solve(const min_quad_with_fixed_mg_data& data, const Eigen::PlainObjectBase<DerivedRHS>& RHS,
                                  const Eigen::PlainObjectBase<DerivedKnownVal>& known_val, const Eigen::PlainObjectBase<DerivedZ0>& z0,
                                  const SmgLDLT& solver, std::vector<mg_data>& mg, Eigen::PlainObjectBase<DerivedZ>& z,
                                  std::vector<double>& r_his)
{
    return min_quad_with_fixed_mg_solve(data, RHS, known_val, z0, solver, 1e-3, mg, z, r_his);            // tolerance = 1e-3 (.cpp:270)
}

// ---------------------------------------------------------------- mg_VCycle.h:22-76
template <typename DerivedB, typename DerivedU>
void mg_VCycle(const SmgLDLT&, const Eigen::PlainObjectBase<DerivedB>& B, const int& preRelaxIter, const int& postRelaxIter, const int lv,
               Eigen::PlainObjectBase<DerivedU>& u, std::vector<mg_data>& mg)
{
    check(smg_vcycle(handle_of(mg), B.derived().data(), preRelaxIter, postRelaxIter, lv, u.derived().data(), (int)B.cols()), "smg_vcycle");
}
template <typename DerivedU, typename DerivedAU>
void A(const Eigen::PlainObjectBase<DerivedU>& u, const std::vector<mg_data>& mg, const int& lv, Eigen::PlainObjectBase<DerivedAU>& Au)
{
    Au.resize(u.rows(), u.cols());
    check(smg_apply_A(handle_of(mg), lv, u.derived().data(), (int)u.cols(), Au.derived().data()), "smg_apply_A");
}
template <typename DerivedX, typename DerivedRX>
void restrict(const Eigen::PlainObjectBase<DerivedX>& x, const std::vector<mg_data>& mg, const int& lv, Eigen::PlainObjectBase<DerivedRX>& Rx)
{
    smg_hierarchy* h = handle_of(mg);
    Rx.resize(smg_level_rows(h, lv + 1), x.cols());
    check(smg_restrict(h, lv, x.derived().data(), (int)x.cols(), Rx.derived().data()), "smg_restrict");
}
template <typename DerivedX, typename DerivedPX>
void prolong(const Eigen::PlainObjectBase<DerivedX>& x, const std::vector<mg_data>& mg, const int& lv, Eigen::PlainObjectBase<DerivedPX>& Px)
{
    smg_hierarchy* h = handle_of(mg);
    Px.resize(smg_level_rows(h, lv), x.cols());
    check(smg_prolong(h, lv, x.derived().data(), (int)x.cols(), Px.derived().data()), "smg_prolong");
}
template <typename DerivedB, typename DerivedU>
void relax(const Eigen::PlainObjectBase<DerivedB>& B, const int& lv, const int& iters, Eigen::PlainObjectBase<DerivedU>& u, std::vector<mg_data>& mg)
{
    check(smg_relax(handle_of(mg), lv, B.derived().data(), (int)B.cols(), iters, u.derived().data()), "smg_relax");
}
template <typename DerivedB, typename DerivedU>
void coarseSolve(const SmgLDLT&, const Eigen::PlainObjectBase<DerivedB>& B, const int&, Eigen::PlainObjectBase<DerivedU>& u, std::vector<mg_data>& mg)
{
    check(smg_coarse_solve(handle_of(mg), B.derived().data(), (int)B.cols(), u.derived().data()), "smg_coarse_solve");
}

// ---------------------------------------------------------------- explicit instantiations, as src/min_quad_with_fixed_mg.cpp:363-373 and
// src/mg_VCycle.cpp:203 make them: column vectors and dense column blocks
#define SMG_INST_SOLVE(M)                                                                                                                           \
    template bool min_quad_with_fixed_mg_solve<M, M, M>(const min_quad_with_fixed_mg_data&, const Eigen::PlainObjectBase<M>&, const Eigen::PlainObjectBase<M>&, \
                                                        const SmgLDLT&, std::vector<mg_data>&, Eigen::PlainObjectBase<M>&, std::vector<double>&);  \
    template bool min_quad_with_fixed_mg_solve<M, M, M>(const min_quad_with_fixed_mg_data&, const Eigen::PlainObjectBase<M>&, const Eigen::PlainObjectBase<M>&, \
                                                        const SmgLDLT&, const double&, std::vector<mg_data>&, Eigen::PlainObjectBase<M>&, std::vector<double>&); \
    template bool min_quad_with_fixed_mg_solve<M, M, M>(const min_quad_with_fixed_mg_data&, const Eigen::PlainObjectBase<M>&, const Eigen::PlainObjectBase<M>&, \
                                                        const SmgLDLT&, const double&, const int&, std::vector<mg_data>&, Eigen::PlainObjectBase<M>&,  \
                                                        std::vector<double>&);                                                                     \
    template bool min_quad_with_fixed_mg_solve<M, M, M, M>(const min_quad_with_fixed_mg_data&, const Eigen::PlainObjectBase<M>&, const Eigen::PlainObjectBase<M>&, \
                                                           const Eigen::PlainObjectBase<M>&, const SmgLDLT&, std::vector<mg_data>&,                \
                                                           Eigen::PlainObjectBase<M>&, std::vector<double>&);                                      \
    template bool min_quad_with_fixed_mg_solve<M, M, M, M>(const min_quad_with_fixed_mg_data&, const Eigen::PlainObjectBase<M>&, const Eigen::PlainObjectBase<M>&, \
                                                           const Eigen::PlainObjectBase<M>&, const SmgLDLT&, const double&, std::vector<mg_data>&, \
                                                           Eigen::PlainObjectBase<M>&, std::vector<double>&);                                      \
    template bool min_quad_with_fixed_mg_solve<M, M, M, M>(const min_quad_with_fixed_mg_data&, const Eigen::PlainObjectBase<M>&, const Eigen::PlainObjectBase<M>&, \
                                                           const Eigen::PlainObjectBase<M>&, const SmgLDLT&, const double&, const int&,            \
                                                           std::vector<mg_data>&, Eigen::PlainObjectBase<M>&, std::vector<double>&);               \
    template void mg_VCycle<M, M>(const SmgLDLT&, const Eigen::PlainObjectBase<M>&, const int&, const int&, const int, Eigen::PlainObjectBase<M>&,  \
                                  std::vector<mg_data>&);                                                                                          \
    template void A<M, M>(const Eigen::PlainObjectBase<M>&, const std::vector<mg_data>&, const int&, Eigen::PlainObjectBase<M>&);                   \
    template void restrict<M, M>(const Eigen::PlainObjectBase<M>&, const std::vector<mg_data>&, const int&, Eigen::PlainObjectBase<M>&);            \
    template void prolong<M, M>(const Eigen::PlainObjectBase<M>&, const std::vector<mg_data>&, const int&, Eigen::PlainObjectBase<M>&);             \
    template void relax<M, M>(const Eigen::PlainObjectBase<M>&, const int&, const int&, Eigen::PlainObjectBase<M>&, std::vector<mg_data>&);         \
    template void coarseSolve<M, M>(const SmgLDLT&, const Eigen::PlainObjectBase<M>&, const int&, Eigen::PlainObjectBase<M>&, std::vector<mg_data>&);
SMG_INST_SOLVE(Eigen::VectorXd)
SMG_INST_SOLVE(Eigen::MatrixXd)
#undef SMG_INST_SOLVE
