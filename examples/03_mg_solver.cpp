// examples/03_mg_solver.cpp -- the reference's canonical caller (03_mg_solver/main.cpp:20-96) on libsmg, without the
// GLFW viewer: Poisson problem A z = B with the boundary loop pinned to zero, solved by the surface multigrid V-cycle
// on the GPU through the C++ mirror of the reference API (mg_api.hpp).
//
//   hipcc -std=c++17 -O2 examples/03_mg_solver.cpp -Lsurface_multigrid_code_amd/lib -lsmg -o 03_mg_solver
//   ./03_mg_solver tests/golden/meshes/bunny.smgm [tol]
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../surface_multigrid_code_amd/csrc/mg_api.hpp"

int main(int argc, char* argv[])
{
    const char* path = argc > 1 ? argv[1] : "tests/golden/meshes/bunny.smgm";
    const double tol = argc > 2 ? std::atof(argv[2]) : 1e-3;

    // load mesh, rescale to unit area
    double* Vp = nullptr; int* Fp = nullptr; int nV = 0, nF = 0;
    if (smg_mesh_read(path, &Vp, &nV, &Fp, &nF) != SMG_OK) { std::fprintf(stderr, "%s\n", smg_last_error()); return 1; }
    smg_mesh_normalize_unit_area(Vp, nV, Fp, nF);
    smgDense V(nV, 3); smgDenseI F(nF, 3);
    for (int i = 0; i < nV; i++) for (int c = 0; c < 3; c++) V(i, c) = Vp[3 * i + c];
    for (int i = 0; i < nF; i++) for (int c = 0; c < 3; c++) F(i, c) = Fp[3 * i + c];
    std::printf("original mesh: |V| %d, |F|: %d\n", nV, nF);

    // multigrid hierarchy
    int min_coarsest_nV = 500; float coarsening_ratio = 0.25f; int decimation_type = 1;
    std::vector<mg_data> mg;
    mg_precompute(V, F, coarsening_ratio, min_coarsest_nV, decimation_type, mg);

    // toy Poisson problem: A = -cotmatrix, z(b) = bval on the boundary loop
    smgSparse A;
    {
        int nnz = 0;
        smg_mesh_cotmatrix(Vp, nV, Fp, nF, &nnz, nullptr, nullptr, nullptr);
        A.rows = A.cols = nV; A.outer.resize(nV + 1); A.inner.resize(nnz); A.values.resize(nnz);
        smg_mesh_cotmatrix(Vp, nV, Fp, nF, nullptr, A.outer.data(), A.inner.data(), A.values.data());
        for (double& v : A.values) v = -v;
    }
    smgDenseI b(nV, 1); int nb = 0;
    smg_mesh_boundary_loop(Fp, nF, nV, b.data.data(), &nb);
    b.rows = nb; b.data.resize(nb);
    smgDense bval(nb, 1);
    smgDense B(nV, 1);
    smg_mesh_massmatrix(Vp, nV, Fp, nF, /*voronoi=*/1, B.data.data());   // B = M * ones
    for (int i = 0; i < nb; i++) B(b(i)) = bval(i);
    smgDense z0(nV, 1), z;

    min_quad_with_fixed_mg_data solverData;
    smgCoarseSolver coarseSolver;
    min_quad_with_fixed_mg_precompute(A, b, solverData, mg, coarseSolver);

    std::vector<double> rHis;
    bool ok = min_quad_with_fixed_mg_solve(solverData, B, bval, z0, coarseSolver, tol, mg, z, rHis);
    double zs = 0.0; for (double v : z.data) zs += v * v;
    std::printf("converged: %d  iterations: %d  |z|^2: %.17g  unknowns: %d\n", (int)ok, (int)rHis.size(), zs, (int)solverData.unknown.size());
    smg_free(Vp); smg_free(Fp);
    return ok ? 0 : 2;
}
