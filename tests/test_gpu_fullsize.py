"""GPU (-m gpu): BASELINE.json's full-size workload (C3: bunny_15K_init x3 mid-point subdivision, 1 011 330
vertices, 5 levels) -- size-independent properties plus one oracle comparison at full size."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c3(smg_mod):
    smg, mesh = smg_mod, smg_mod.mesh
    assert smg._lib.load().smg_device_count() > 0
    V, F = mesh.read_triangle_mesh("bunny_15K_init.smgm")
    V = mesh.normalize_unit_area(V, F)
    mg, Vf, Ff = smg.mg_precompute_subdiv(V, F, 3, ratio=0.25, nVCoarsest=1000, n_extra_levels=1)
    L = mesh.cotmatrix(Vf, Ff)
    Mb = mesh.massmatrix(Vf, Ff, "barycentric")
    A = (Mb - 0.01 * L).tocsr()
    A.sort_indices()
    mg.precompute(A)
    return smg, mg, A, Mb, Vf


def test_sizes_match_the_survey(c3):
    smg, mg, A, Mb, Vf = c3
    assert A.shape[0] == 1011330 and A.nnz == 7079298          # n0, nnz0
    assert mg.n_levels == 5
    assert [mg.rows(l) for l in range(4)] == [1011330, 252834, 63210, 15804]
    assert 3800 < mg.rows(4) < 4100                              # ~3 951 from the decimator
    assert mg.spmv_bytes(0, 1) == 105178180                      # 12 nnz + 4 (n+1) + 16 n
    st = mg.sell_stats(0, "A")
    assert st["padded"] / st["stored"] < 1.02


def test_spmv_properties_full_size(c3):
    smg, mg, A, Mb, Vf = c3
    rng = np.random.default_rng(3)                               # x: uniform(-1,1), seed 3
    n = A.shape[0]
    x, y = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
    Ax, Ay = mg.A(0, x)[:, 0], mg.A(0, y)[:, 0]
    ref = A @ x                                                  # scipy CSR: independent summation
    assert abs(Ax - ref).max() <= 1e-14 * abs(ref).max() * 8
    assert abs(x @ Ay - y @ Ax) <= 1e-12 * abs(x @ Ay)           # symmetry
    lin = mg.A(0, 0.5 * x - 2.0 * y)[:, 0]
    assert abs(lin - (0.5 * Ax - 2.0 * Ay)).max() <= 1e-13 * abs(Ay).max()
    assert np.array_equal(Ax, mg.A(0, x)[:, 0])                  # deterministic


def test_solve_full_size_against_oracle(c3, oracle_mod):
    smg, mg, A, Mb, Vf = c3
    n = A.shape[0]
    rhs = Mb @ Vf                                                # mean-curvature-flow RHS, 3 columns
    conv, z, rh = mg.solve(rhs, Vf, None, smg.SolveOpts(tol=5e-7, max_iter=20))
    assert conv and (np.diff(rh) < 0).all()
    true = np.linalg.norm(rhs - A @ z)
    assert true < 5e-7
    Ps = [mg.matrix(l, "P_full") for l in range(1, mg.n_levels)]
    orc = oracle_mod.OracleMG(Ps)
    orc.precompute(A)
    conv2, z2, rh2 = orc.solve(rhs, Vf, tol=5e-7, max_iter=20)
    assert conv2 and abs(len(rh) - len(rh2)) <= 2 and abs(rh[0] - rh2[0]) <= 1e-11 * rh2[0]
    assert np.linalg.norm(z - z2) <= 1e-5 * np.linalg.norm(z2)
    # Galerkin operators identical bit for bit at full size
    for l in range(mg.n_levels):
        Ao = orc.level_A(l).tocsr()
        Ao.sort_indices()
        assert np.array_equal(mg.matrix(l, "A").data, Ao.data)


def test_c5_four_million_vertices_fp64_and_mixed(smg_mod):
    """BASELINE config C5 (synthetic 4 M-vertex closed surface: torus 64 x 64, five mid-point subdivisions re-projected onto the
    torus, 6 levels): beyond the oracle's reach in test time, so size-independent properties only -- SpMV against an independent
    CSR product, symmetry, determinism, and both precisions driving the TRUE residual (recomputed on the host) below 1e-10 of
    the right-hand side with monotone histories."""
    smg, mesh = smg_mod, smg_mod.mesh
    V, F = mesh.torus(64, 64)
    mg, Vf, Ff = smg.mg_precompute_subdiv(V, F, 5, n_extra_levels=0)
    R, r = 1.0, 0.4                                              # back onto the torus (bench.py: _onto_torus)
    rho = np.maximum(np.hypot(Vf[:, 0], Vf[:, 1]), 1e-300)
    cx, cy = R * Vf[:, 0] / rho, R * Vf[:, 1] / rho
    d = Vf - np.stack([cx, cy, np.zeros_like(cx)], axis=1)
    Vf = np.stack([cx, cy, np.zeros_like(cx)], axis=1) + r * d / np.maximum(np.linalg.norm(d, axis=1), 1e-300)[:, None]
    Vf = mesh.normalize_unit_area(Vf, Ff)
    A = (mesh.massmatrix(Vf, Ff, "barycentric") - 0.01 * mesh.cotmatrix(Vf, Ff)).tocsr()
    A.sort_indices()
    n = A.shape[0]
    assert n == 4194304 and mg.n_levels == 6
    mg.precompute(A)
    rng = np.random.default_rng(5)
    x, y = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
    Ax, Ay = mg.A(0, x)[:, 0], mg.A(0, y)[:, 0]
    ref = A @ x
    assert abs(Ax - ref).max() <= 1e-13 * abs(ref).max()
    assert abs(x @ Ay - y @ Ax) <= 1e-12 * abs(x @ Ay)
    assert np.array_equal(Ax, mg.A(0, x)[:, 0])
    rhs = A @ rng.uniform(-1, 1, n)
    z0 = np.zeros(n)
    for prec in (0, 1):
        conv, z, rh = mg.solve(rhs, z0, None, smg.SolveOpts(tol=1e-10 * np.linalg.norm(rhs), max_iter=60, precision=prec))
        assert conv and (np.diff(rh) < 0).all(), (prec, rh)
        assert np.linalg.norm(rhs - A @ z[:, 0]) <= 2e-10 * np.linalg.norm(rhs)
