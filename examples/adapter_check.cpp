// examples/adapter_check.cpp -- drives examples/smg_eigen_adapter.cpp (built with -DSMG_ADAPTER_MOCK against tests/mock_eigen) the way
// the reference's 03_mg_solver/main.cpp:38-75 drives the reference: Poisson problem with the boundary loop pinned, through the
// reference's own function signatures; checks what the calls must leave in their arguments.  A plumbing check of the adapter on a
// real GPU -- the mock containers are not Eigen.
//
//   hipcc -std=c++17 -O2 -DSMG_ADAPTER_MOCK -Itests/mock_eigen -Iinclude examples/adapter_check.cpp examples/smg_eigen_adapter.cpp
//         -Lsurface_multigrid_code_amd/lib -lsmg -o examples/adapter_check
#include <cmath>
#include <cstdio>
#include <vector>

#include "min_quad_with_fixed_mg.h"   // tests/mock_eigen
#include <smg.h>

typedef Eigen::SimplicialLDLT<Eigen::SparseMatrix<double>> LDLT;
// the reference's declarations (src/min_quad_with_fixed_mg.h:32-36,72-77,38-69,79-113; src/mg_VCycle.h:22-68), defined by the adapter
void min_quad_with_fixed_mg_precompute(const Eigen::SparseMatrix<double>& A, const Eigen::VectorXi& known, min_quad_with_fixed_mg_data& data,
                                       std::vector<mg_data>& mg, LDLT& solver);
void min_quad_with_fixed_mg_precompute(const Eigen::SparseMatrix<double>& A, min_quad_with_fixed_mg_data& data, std::vector<mg_data>& mg, LDLT& solver);
template <typename DR, typename DK, typename DZ0, typename DZ>
bool min_quad_with_fixed_mg_solve(const min_quad_with_fixed_mg_data& data, const Eigen::PlainObjectBase<DR>& RHS, const Eigen::PlainObjectBase<DK>& known_val,
                                  const Eigen::PlainObjectBase<DZ0>& z0, const LDLT& solver, const double& tolerance, std::vector<mg_data>& mg,
                                  Eigen::PlainObjectBase<DZ>& z, std::vector<double>& r_his);
template <typename DR, typename DZ0, typename DZ>
bool min_quad_with_fixed_mg_solve(const min_quad_with_fixed_mg_data& data, const Eigen::PlainObjectBase<DR>& RHS, const Eigen::PlainObjectBase<DZ0>& z0,
                                  const LDLT& solver, const double& tolerance, const int& maxIter, std::vector<mg_data>& mg, Eigen::PlainObjectBase<DZ>& z,
                                  std::vector<double>& r_his);
template <typename DB, typename DU>
void mg_VCycle(const LDLT& solver, const Eigen::PlainObjectBase<DB>& B, const int& pre, const int& post, const int lv, Eigen::PlainObjectBase<DU>& u,
               std::vector<mg_data>& mg);
template <typename DU, typename DAU>
void A(const Eigen::PlainObjectBase<DU>& u, const std::vector<mg_data>& mg, const int& lv, Eigen::PlainObjectBase<DAU>& Au);
void smg_eigen_adapter_release(const std::vector<mg_data>& mg);

static Eigen::SparseMatrix<double> csc_from_csr(int nr, int nc, const std::vector<int>& rp, const std::vector<int>& ci, const std::vector<double>& v)
{
    std::vector<int> cp((size_t)nc + 1, 0), ri(ci.size());
    std::vector<double> cv(v.size());
    for (int c : ci) cp[(size_t)c + 1]++;
    for (int c = 0; c < nc; c++) cp[(size_t)c + 1] += cp[c];
    std::vector<int> fill(cp.begin(), cp.end() - 1);
    for (int r = 0; r < nr; r++)
        for (int p = rp[r]; p < rp[r + 1]; p++) { const int q = fill[ci[p]]++; ri[q] = r; cv[q] = v[p]; }
    Eigen::SparseMatrix<double> M;
    M = Eigen::Map<const Eigen::SparseMatrix<double>>(nr, nc, (long)ci.size(), cp.data(), ri.data(), cv.data());
    return M;
}

#define REQUIRE(c) do { if (!(c)) { std::printf("ADAPTER_CHECK FAILED: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

int main(int argc, char* argv[])
{
    const char* path = argc > 1 ? argv[1] : "tests/golden/meshes/bunny.smgm";
    double* Vp = nullptr; int* Fp = nullptr; int nV = 0, nF = 0;
    if (smg_mesh_read(path, &Vp, &nV, &Fp, &nF) != SMG_OK) { std::fprintf(stderr, "%s\n", smg_last_error()); return 1; }
    smg_mesh_normalize_unit_area(Vp, nV, Fp, nF);
    // mg_precompute: the reference's own builder in a real integration; here libsmg's, read back into std::vector<mg_data>
    smg_hierarchy* hb = nullptr;
    REQUIRE(smg_mg_precompute(Vp, nV, Fp, nF, 0.25f, 500, 1, &hb) == SMG_OK);
    const int L = smg_hierarchy_levels(hb);
    std::vector<mg_data> mg((size_t)L);
    for (int lv = 1; lv < L; lv++) {
        int nr, nc, nnz;
        smg_level_get_matrix(hb, lv, 3, 0, &nr, &nc, &nnz, nullptr, nullptr, nullptr);
        std::vector<int> rp((size_t)nr + 1), ci((size_t)nnz); std::vector<double> v((size_t)nnz);
        smg_level_get_matrix(hb, lv, 3, 0, nullptr, nullptr, nullptr, rp.data(), ci.data(), v.data());
        mg[lv].P_full = csc_from_csr(nr, nc, rp, ci, v);
        mg[lv].P = mg[lv].P_full;
    }
    smg_hierarchy_destroy(hb);
    // A = -cotmatrix (03_mg_solver/main.cpp:45-46); b = boundary loop, bval = 0 (:50-52); B = M * 1, B(b) = bval (:56-61)
    int nnz = 0;
    smg_mesh_cotmatrix(Vp, nV, Fp, nF, &nnz, nullptr, nullptr, nullptr);
    std::vector<int> ap((size_t)nV + 1), ai((size_t)nnz); std::vector<double> av((size_t)nnz);
    smg_mesh_cotmatrix(Vp, nV, Fp, nF, nullptr, ap.data(), ai.data(), av.data());
    for (double& x : av) x = -x;
    Eigen::SparseMatrix<double> Amat;
    Amat = Eigen::Map<const Eigen::SparseMatrix<double>>(nV, nV, nnz, ap.data(), ai.data(), av.data());   // symmetric: CSR arrays == CSC arrays
    std::vector<int> loop((size_t)nV); int nb = 0;
    smg_mesh_boundary_loop(Fp, nF, nV, loop.data(), &nb);
    REQUIRE(nb > 0);
    Eigen::VectorXi b(nb);
    for (int i = 0; i < nb; i++) b(i) = loop[i];
    Eigen::VectorXd bval(nb), B(nV), z0(nV), z;
    smg_mesh_massmatrix(Vp, nV, Fp, nF, 1, B.data());
    for (int i = 0; i < nb; i++) B(b(i)) = bval(i);

    min_quad_with_fixed_mg_data solverData;
    LDLT coarseSolver;
    min_quad_with_fixed_mg_precompute(Amat, b, solverData, mg, coarseSolver);                        // main.cpp:71
    // what precompute must leave behind (src/min_quad_with_fixed_mg.cpp:156-179, :223-246)
    REQUIRE(solverData.n == nV && solverData.known.size() == nb && solverData.unknown.size() == nV - nb);
    REQUIRE(solverData.LHS.rows() == nV - nb && solverData.Auk.rows() == nV - nb && solverData.Auk.cols() == nb && solverData.Auk.nonZeros() > 0);
    for (int lv = 0; lv < L; lv++) {
        REQUIRE(mg[lv].A.rows() == mg[lv].A.cols() && mg[lv].A_diag.size() == mg[lv].A.rows());
        if (lv >= 1) REQUIRE(mg[lv].P.rows() == mg[lv - 1].A.rows() && mg[lv].P.cols() == mg[lv].A.rows() && mg[lv].PT.rows() == mg[lv].A.rows());
    }
    REQUIRE(mg[0].A.rows() == nV - nb);
    std::vector<double> rHis;
    bool ok = min_quad_with_fixed_mg_solve(solverData, B, bval, z0, coarseSolver, 1e-8, mg, z, rHis);   // main.cpp:75
    REQUIRE(ok && !rHis.empty() && rHis.back() < 1e-8 && (int)rHis.size() <= 20);
    for (size_t i = 1; i < rHis.size(); i++) REQUIRE(rHis[i] < rHis[i - 1]);
    for (int i = 0; i < nb; i++) REQUIRE(z(b(i)) == bval(i));
    // the true residual of the returned z on the unknown rows, from the caller's own A
    {
        std::vector<char> isk((size_t)nV, 0);
        for (int i = 0; i < nb; i++) isk[b(i)] = 1;
        double ss = 0.0;
        for (int r = 0; r < nV; r++) {
            if (isk[r]) continue;
            double s = 0.0;
            for (int p = ap[r]; p < ap[r + 1]; p++) s += av[p] * z(ai[p]);
            ss += (B(r) - s) * (B(r) - s);
        }
        REQUIRE(std::sqrt(ss) < 2e-8);
        std::printf("true residual %.3e after %d iterations\n", std::sqrt(ss), (int)rHis.size());
    }
    // mg_VCycle / A() on the reduced system of level 1
    {
        const long n1 = mg[1].A.rows();
        Eigen::VectorXd B1(n1), u1(n1), Au;
        for (long i = 0; i < n1; i++) B1(i) = 1.0 / (1.0 + (double)(i % 7));
        mg_VCycle(coarseSolver, B1, 2, 2, 1, u1, mg);
        A(u1, mg, 1, Au);
        double r0 = 0.0, r1 = 0.0;
        for (long i = 0; i < n1; i++) { r0 += B1(i) * B1(i); r1 += (B1(i) - Au(i)) * (B1(i) - Au(i)); }
        REQUIRE(r1 < 0.25 * r0);
    }
    // the no-constraint overload on the same mg (re-precompute with another matrix: M + 0.01 (-L)-like shift keeps it SPD)
    {
        std::vector<double> av2(av);
        for (int r = 0; r < nV; r++) for (int p = ap[r]; p < ap[r + 1]; p++) if (ai[p] == r) av2[p] += 1.0;
        Eigen::SparseMatrix<double> A2;
        A2 = Eigen::Map<const Eigen::SparseMatrix<double>>(nV, nV, nnz, ap.data(), ai.data(), av2.data());
        min_quad_with_fixed_mg_precompute(A2, solverData, mg, coarseSolver);
        REQUIRE(solverData.unknown.size() == nV && solverData.known.size() == 0 && mg[0].A.rows() == nV);
        Eigen::MatrixXd R(nV, 3), Z0(nV, 3), Z;
        for (int i = 0; i < nV; i++) for (int c = 0; c < 3; c++) R(i, c) = Vp[3 * i + c];
        std::vector<double> rh2;
        bool ok2 = min_quad_with_fixed_mg_solve(solverData, R, Z0, coarseSolver, 1e-9, 30, mg, Z, rh2);
        REQUIRE(ok2 && Z.rows() == nV && Z.cols() == 3);
    }
    smg_eigen_adapter_release(mg);
    smg_free(Vp); smg_free(Fp);
    std::printf("ADAPTER_CHECK OK: %d levels, %d unknowns\n", L, nV - nb);
    return 0;
}
