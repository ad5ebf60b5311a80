// examples/05_mean_curvature_flow.cpp -- the reference's 05_example_mean_curvature_flow/main.cpp:57-79 on libsmg,
// without the viewer: implicit mean-curvature flow [Kazhdan et al. 2012], each step solves (M - delta L) U' = M U for the
// three coordinate columns with the surface multigrid V-cycle.  The matrix keeps its sparsity from step to step, so
// every min_quad_with_fixed_mg_precompute after the first runs its Galerkin products and coarse factorisation on the GPU.
//
//   ./05_mean_curvature_flow tests/golden/meshes/ogre_sim.smgm [steps]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../surface_multigrid_code_amd/csrc/mg_api.hpp"

int main(int argc, char* argv[])
{
    const char* path = argc > 1 ? argv[1] : "tests/golden/meshes/ogre_sim.smgm";
    const int steps = argc > 2 ? std::atoi(argv[2]) : 3;
    double* Vp = nullptr; int* Fp = nullptr; int nV = 0, nF = 0;
    if (smg_mesh_read(path, &Vp, &nV, &Fp, &nF) != SMG_OK) { std::fprintf(stderr, "%s\n", smg_last_error()); return 1; }
    smg_mesh_normalize_unit_area(Vp, nV, Fp, nF);
    smgDense V(nV, 3); smgDenseI F(nF, 3);
    for (int i = 0; i < nV; i++) for (int c = 0; c < 3; c++) V(i, c) = Vp[3 * i + c];
    for (int i = 0; i < nF; i++) for (int c = 0; c < 3; c++) F(i, c) = Fp[3 * i + c];
    std::printf("original mesh: |V| %d, |F|: %d\n", nV, nF);

    std::vector<mg_data> mg;
    mg_precompute(V, F, 0.25f, 100, 1, mg);

    // L = cotmatrix(V, F) of the ORIGINAL mesh (main.cpp:44), fixed for all steps
    smgSparse L;
    int nnz = 0;
    smg_mesh_cotmatrix(Vp, nV, Fp, nF, &nnz, nullptr, nullptr, nullptr);
    L.rows = L.cols = nV; L.outer.resize(nV + 1); L.inner.resize(nnz); L.values.resize(nnz);
    smg_mesh_cotmatrix(Vp, nV, Fp, nF, nullptr, L.outer.data(), L.inner.data(), L.values.data());

    const double delta = 0.01, mg_tol = 5e-7;
    smgDense U = V;
    std::vector<double> Urow((size_t)nV * 3), M(nV);
    min_quad_with_fixed_mg_data solverData;
    smgCoarseSolver coarseSolver;          // caller-owned, reused across steps like the handle it stands for
    for (int s = 0; s < steps; s++) {
        smgDense Upre = U;
        for (int i = 0; i < nV; i++) for (int c = 0; c < 3; c++) Urow[3 * (size_t)i + c] = U(i, c);
        smg_mesh_massmatrix(Urow.data(), nV, Fp, nF, /*voronoi=*/0, M.data());       // igl::massmatrix(U, F, BARYCENTRIC)
        smgSparse LHS = L;                                                              // LHS = M - delta * L
        for (int i = 0; i < nV; i++)
            for (int p = LHS.outer[i]; p < LHS.outer[i + 1]; p++) {
                const double x = delta * L.values[p];
                LHS.values[p] = (LHS.inner[p] == i) ? M[i] - x : -x;
            }
        smgDense RHS(nV, 3);                                                            // RHS = M * U
        for (int i = 0; i < nV; i++) for (int c = 0; c < 3; c++) RHS(i, c) = M[i] * U(i, c);
        min_quad_with_fixed_mg_precompute(LHS, solverData, mg, coarseSolver);
        std::vector<double> rHis;
        bool ok = min_quad_with_fixed_mg_solve(solverData, RHS, Upre, coarseSolver, mg_tol, mg, U, rHis);
        for (int i = 0; i < nV; i++) for (int c = 0; c < 3; c++) Urow[3 * (size_t)i + c] = U(i, c);
        smg_mesh_normalize_unit_area(Urow.data(), nV, Fp, nF);                          // rescale output (main.cpp:79)
        for (int i = 0; i < nV; i++) for (int c = 0; c < 3; c++) U(i, c) = Urow[3 * (size_t)i + c];
        double s2 = 0; for (double v : U.data) s2 += v * v;
        std::printf("step %d: converged %d in %d iterations, |U|^2 = %.15g\n", s, (int)ok, (int)rHis.size(), s2);
        if (!ok) return 2;
    }
    // one extra V-cycle through the mirror of mg_VCycle (reference src/mg_VCycle.h:22-30) on the last system
    smgDense B(nV, 1), u(nV, 1);
    for (int i = 0; i < nV; i++) B(i) = M[i];
    mg_VCycle(coarseSolver, B, 2, 2, 0, u, mg);
    double un = 0; for (double v : u.data) un += v * v;
    std::printf("mg_VCycle: |u|^2 = %.15g\n", un);
    smg_free(Vp); smg_free(Fp);
    return 0;
}
