// examples/04_mg_solver_nobd.cpp -- the reference's second caller (04_mg_solver_nobd/main.cpp:60-105) on libsmg, without the
// GLFW viewer: Poisson problem on a CLOSED surface, a few hundred vertices pinned to zero (the reference pins the vertices listed
// in hilbert_cube_known.obj -- 346 of them; that mesh is not shipped, so here every (nV / 346)-th vertex is pinned), a random
// initial guess, tolerance 1e-10.
//
//   hipcc -std=c++17 -O2 examples/04_mg_solver_nobd.cpp -Lsurface_multigrid_code_amd/lib -lsmg -o 04_mg_solver_nobd
//   ./04_mg_solver_nobd tests/golden/meshes/bunny_15K_init.smgm [n_pins]
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../surface_multigrid_code_amd/csrc/mg_api.hpp"

int main(int argc, char* argv[])
{
    const char* path = argc > 1 ? argv[1] : "tests/golden/meshes/bunny_15K_init.smgm";
    const int n_pins = argc > 2 ? std::atoi(argv[2]) : 346;
    const bool fast_cycle = argc > 3 && std::atoi(argv[3]) != 0;   // opt into libsmg's hybrid Gauss-Seidel / Chebyshev-Jacobi cycle

    double* Vp = nullptr; int* Fp = nullptr; int nV = 0, nF = 0;
    if (smg_mesh_read(path, &Vp, &nV, &Fp, &nF) != SMG_OK) { std::fprintf(stderr, "%s\n", smg_last_error()); return 1; }
    smg_mesh_normalize_unit_area(Vp, nV, Fp, nF);
    smgDense V(nV, 3); smgDenseI F(nF, 3);
    for (int i = 0; i < nV; i++) for (int c = 0; c < 3; c++) V(i, c) = Vp[3 * i + c];
    for (int i = 0; i < nF; i++) for (int c = 0; c < 3; c++) F(i, c) = Fp[3 * i + c];
    std::printf("original mesh: |V| %d, |F|: %d\n", nV, nF);

    int min_coarsest_nV = 500; float coarsening_ratio = 0.25f; int decimation_type = 1;
    std::vector<mg_data> mg;
    mg_precompute(V, F, coarsening_ratio, min_coarsest_nV, decimation_type, mg);

    // A = -cotmatrix; z(b) = 0 on the pinned vertices
    smgSparse A;
    {
        int nnz = 0;
        smg_mesh_cotmatrix(Vp, nV, Fp, nF, &nnz, nullptr, nullptr, nullptr);
        A.rows = A.cols = nV; A.outer.resize(nV + 1); A.inner.resize(nnz); A.values.resize(nnz);
        smg_mesh_cotmatrix(Vp, nV, Fp, nF, nullptr, A.outer.data(), A.inner.data(), A.values.data());
        for (double& v : A.values) v = -v;
    }
    const int step = nV / n_pins;
    smgDenseI b(n_pins, 1);
    for (int i = 0; i < n_pins; i++) b(i) = i * step;
    smgDense bval(n_pins, 1);
    smgDense B(nV, 1);
    smg_mesh_massmatrix(Vp, nV, Fp, nF, /*voronoi=*/1, B.data.data());   // B = M * ones
    for (int i = 0; i < n_pins; i++) B(b(i)) = bval(i);

    // "random" initial guess in [-1, 1) (the reference seeds rand() with the time; here a fixed linear congruential sequence)
    smgDense z0(nV, 1), z;
    unsigned long long x = 12345;
    for (int i = 0; i < nV; i++) { x = (1103515245ull * x + 12345ull) % 2147483648ull; z0(i) = (double)x / 1073741824.0 - 1.0; }

    min_quad_with_fixed_mg_data solverData;
    smgCoarseSolver coarseSolver;
    if (fast_cycle) { coarseSolver.opts.smoother = SMG_SMOOTH_HYBRID_CHEBYSHEV; coarseSolver.opts.jacobi_max_rows = 300000; }
    min_quad_with_fixed_mg_precompute(A, b, solverData, mg, coarseSolver);

    std::vector<double> rHis;
    bool ok = min_quad_with_fixed_mg_solve(solverData, B, bval, z0, coarseSolver, 1e-10, mg, z, rHis);   // 04_mg_solver_nobd/main.cpp:105
    double zs = 0.0; for (double v : z.data) zs += v * v;
    std::printf("converged: %d  iterations: %d  |z|^2: %.17g  unknowns: %d\n", (int)ok, (int)rHis.size(), zs, (int)solverData.unknown.size());
    smg_free(Vp); smg_free(Fp);
    return ok ? 0 : 2;
}
