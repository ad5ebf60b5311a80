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


def test_memory_budget_at_full_size_default_and_lean(c3):
    """What one handle holds in HBM against the hierarchy as the reference holds it (mg_data: A, P, PT per level in CSC, 12 bytes per stored entry): the default
    layout (fixed panel pitch: no launch waits for a table) stays below 3.3 x after a solve, smg_hierarchy_set_memory_lean (compact panels) below 2.5 x --
    with bit-identical iterates: the layout of the panels is not the order of the sums."""
    import torch
    smg, mg, A, Mb, Vf = c3
    n = A.shape[0]
    dev = torch.device("cuda", 0)
    rhs = torch.from_numpy(Mb @ np.random.default_rng(100).uniform(-1.0, 1.0, n)).to(dev)      # vectors resident in HBM, as in bench.py (host blocks add 16 MB of staging)
    z0, z, z2 = torch.zeros(n, dtype=torch.float64, device=dev), torch.empty(n, dtype=torch.float64, device=dev), torch.empty(n, dtype=torch.float64, device=dev)
    o = smg.SolveOpts(tol=1e-10, max_iter=30)
    m1 = smg.Hierarchy.from_prolongs([mg.matrix(l, "P_full") for l in range(1, mg.n_levels)])      # a fresh handle: one column, nothing staged (the fixture's has served other tests)
    m1.precompute(A)
    conv, rh = m1.solve_device(rhs.data_ptr(), z0.data_ptr(), z.data_ptr(), n, 1, opts=o)
    alg = sum(12 * mg.matrix(l, "A").nnz for l in range(mg.n_levels)) + sum(24 * mg.matrix(l, "P").nnz for l in range(1, mg.n_levels))
    fat = m1.device_bytes()["total"]
    del m1
    assert conv and fat <= 3.3 * alg, (fat, alg)
    m2 = smg.Hierarchy.from_prolongs([mg.matrix(l, "P_full") for l in range(1, mg.n_levels)])
    m2.set_memory_lean(True)
    m2.precompute(A)
    conv2, rh2 = m2.solve_device(rhs.data_ptr(), z0.data_ptr(), z2.data_ptr(), n, 1, opts=o)
    lean, lean_a0 = m2.device_bytes()["total"], m2.device_bytes()["level0.A_sell"]
    assert conv2 and torch.equal(z, z2) and np.array_equal(rh, rh2)
    assert lean <= 2.5 * alg and lean < 0.8 * fat, (lean, fat, alg)
    m2.set_memory_lean(False)                                    # back: the next precompute is a full one, the pitch returns
    m2.precompute(A)
    assert m2.device_bytes()["level0.A_sell"] > 1.5 * lean_a0


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


# ----------------------------------------------------------------------------------------------- oracle at full size, all-core mode
def _oracle_in_device_numbering(oracle_mod, mg, A, known=None, threads=32):
    """The oracle on the system renumbered colour-major with the GPU path's own numbering (level by level): the reference's
    lexicographic sweep on it IS the multi-colour sweep the GPU runs, so GPU and oracle iterate identically up to the coarse solve
    (dense inverse vs LDL^T, 1e-11) and summation order of the norms -- and the colour blocks can be swept by all host cores without
    changing a bit (oracle all-core mode), which brings 1 M / 4 M-vertex solves into test time."""
    import scipy.sparse as sp
    L = mg.n_levels
    perms = [mg.perm(l) for l in range(L)]
    Ps = [sp.csr_matrix(mg.matrix(l, "P"))[perms[l - 1]][:, perms[l]].tocsc() for l in range(1, L)]
    A0 = sp.csr_matrix(mg.matrix(0, "A"))[perms[0]][:, perms[0]].tocsr()      # LHS = A(unknown, unknown), device numbering
    orc = oracle_mod.OracleMG(Ps)
    orc.precompute(A0)
    orc.set_parallel([mg.colors(l) for l in range(L - 1)], threads)
    return orc, perms[0]


def test_c3_poisson_with_346_pins_full_size_against_oracle(c3, oracle_mod):
    """C3's second system: Poisson -L with 346 pinned vertices on the 1 011 330-vertex mesh, z0 uniform(-1,1),
    tol 1e-10 -- GPU against the oracle in the device numbering, iteration for iteration."""
    smg, mg0, A_mcf, Mb, Vf = c3
    mesh = smg.mesh
    V, F = mesh.read_triangle_mesh("bunny_15K_init.smgm")
    V = mesh.normalize_unit_area(V, F)
    mg, Vf2, Ff = smg.mg_precompute_subdiv(V, F, 3, ratio=0.25, nVCoarsest=1000, n_extra_levels=1)
    n = Vf2.shape[0]
    A = (-mesh.cotmatrix(Vf2, Ff)).tocsr()
    A.sort_indices()
    known = np.sort(np.random.default_rng(0).choice(n, 346, replace=False)).astype(np.int32)
    mg.precompute(A, known)
    assert mg.rows(0) == n - 346
    B = mesh.massmatrix(Vf2, Ff, "voronoi") @ np.ones(n)
    B[known] = 0.0
    z0 = np.random.default_rng(1).uniform(-1, 1, n)
    kv = np.zeros(346)
    conv, z, rh = mg.solve(B, z0, kv, smg.SolveOpts(tol=1e-10, max_iter=60))
    assert conv and np.array_equal(z[known, 0], kv)
    unk = mg.unknown()
    true = np.linalg.norm((B - A @ z[:, 0])[unk])
    assert true < 1.5e-10
    # the same reduced system through the oracle (device numbering, all-core): identical iteration
    orc, perm0 = _oracle_in_device_numbering(oracle_mod, mg, A, known)
    rhs_u = (B - A[:, known] @ kv)[unk][perm0]
    conv2, z2, rh2 = orc.solve(rhs_u, z0[unk][perm0], tol=1e-10, max_iter=60)
    assert conv2 and len(rh2) == len(rh)
    np.testing.assert_allclose(rh, rh2, rtol=1e-6)
    assert np.linalg.norm(z[unk, 0][perm0] - z2[:, 0]) <= 1e-9 * np.linalg.norm(z2)


def test_c3_parent_under_the_references_own_hierarchy_against_oracle(smg_mod, oracle_mod):
    """The reference's own kind of hierarchy at size: mg_precompute(V, F, 0.25, 1000, midpoint) -- src/mg_precompute.cpp:15-87,
    get_prolong.cpp:45-56, the call of 03_mg_solver/main.cpp:35-39 -- on the 252 834-vertex parent of the C3 mesh: 4 levels, 3 entries per row of P,
    Galerkin operators of 18 - 27 entries per row.  Every kernel on every level bit for bit against the oracle in the order the device sweeps
    (the Galerkin levels sweep piece-wise, csrc/smg_wgs.hpp), then the solve against the oracle on the system renumbered level by level into
    those orders: the reference's lexicographic cycle on it IS the device's cycle, iteration for iteration."""
    import scipy.sparse as sp
    smg, mesh = smg_mod, smg_mod.mesh
    import bench as B
    from test_gpu_parity import gs_bit_exact, oracle_on_device_numbering
    mg, A, Mb, Vf, Ff, label, _ = B.build_workload("C3pdec", smg, mesh)
    n = A.shape[0]
    assert n == 252834 and mg.n_levels == 4
    mg.precompute(A)
    L = mg.n_levels
    rng = np.random.default_rng(7)
    for lv in range(L - 1):
        nn = np.diff(mg.matrix(lv, "A").indptr)
        P = mg.matrix(lv + 1, "P")
        assert P.nnz == 3 * mg.rows(lv)                                    # get_prolong.cpp:48-54: three stored entries per fine row
        if lv >= 1:
            assert nn.mean() > 15 and mg.wave_gs_order(lv, 1) is not None  # a Galerkin level of the reference's construction, swept piece-wise
        m, mc = mg.rows(lv), mg.rows(lv + 1)
        perm, permc = mg.perm(lv), mg.perm(lv + 1)
        oi = oracle_on_device_numbering(oracle_mod, mg, lv)
        x, b, xc = rng.uniform(-1, 1, (m, 1)), rng.uniform(-1, 1, (m, 1)), rng.uniform(-1, 1, (mc, 1))
        assert np.array_equal(mg.A(lv, x)[perm], oi.A(0, x[perm])), "SpMV, level %d" % lv
        assert np.array_equal(mg.restrict(lv, x)[permc], oi.restrict(0, x[perm])), "restriction, level %d" % lv
        assert np.array_equal(mg.prolong(lv, xc)[perm], oi.prolong(0, xc[permc])), "prolongation, level %d" % lv
        assert gs_bit_exact(oracle_mod, mg, lv, b, x, 2), "Gauss-Seidel, level %d" % lv
    rhs = Mb @ np.random.default_rng(100).uniform(-1, 1, n)
    z0 = np.zeros(n)
    tol = 1e-9 * np.linalg.norm(rhs)
    conv, z, rh = mg.solve(rhs, z0, None, smg.SolveOpts(tol=tol, max_iter=40))
    assert conv
    # the oracle on the system in the device's sweep orders (position -> caller, level by level; the coarsest level in its internal numbering)
    to = [mg.perm(l)[mg.gs_order(l, 1)] for l in range(L - 1)] + [mg.perm(L - 1)]
    Ps = [sp.csr_matrix(mg.matrix(l, "P"))[to[l - 1]][:, to[l]].tocsc() for l in range(1, L)]
    orc = oracle_mod.OracleMG(Ps)
    orc.precompute(sp.csr_matrix(A)[to[0]][:, to[0]].tocsr())
    conv2, z2, rh2 = orc.solve(rhs[to[0]], z0[to[0]], tol=tol, max_iter=40)
    assert conv2 and len(rh2) == len(rh)
    np.testing.assert_allclose(rh, rh2, rtol=1e-6)
    assert np.linalg.norm(z[to[0], 0] - z2[:, 0]) <= 1e-9 * np.linalg.norm(z2)
    # ... and against the reference's lexicographic cycle in the CALLER's numbering: another sweep order, same solution, same cycle count (+-2)
    orc_c = oracle_mod.OracleMG([mg.matrix(l, "P_full") for l in range(1, L)])
    orc_c.precompute(A)
    conv3, z3, rh3 = orc_c.solve(rhs, z0, tol=tol, max_iter=40)
    assert conv3 and abs(len(rh3) - len(rh)) <= 2 and np.linalg.norm(z[:, 0] - z3[:, 0]) <= 1e-7 * np.linalg.norm(z3)


def test_c5_solve_against_oracle_all_core(smg_mod, oracle_mod):
    """BASELINE config C5 (4 194 304 vertices, 6 levels): the fp64 solve against the oracle in the device numbering (all-core mode:
    seconds instead of minutes), iteration for iteration; the mixed-precision solve lands on the same solution."""
    smg, mesh = smg_mod, smg_mod.mesh
    import bench as B
    mg, A, Mb, Vf, Ff, label, _ = B.build_workload("C5", smg, mesh)
    n = A.shape[0]
    assert n == 4194304 and mg.n_levels == 6
    mg.precompute(A)
    rhs = Mb @ np.random.default_rng(100).uniform(-1, 1, n)
    z0 = np.zeros(n)
    tol = 1e-9 * np.linalg.norm(rhs)
    conv, z, rh = mg.solve(rhs, z0, None, smg.SolveOpts(tol=tol, max_iter=40))
    assert conv
    orc, perm0 = _oracle_in_device_numbering(oracle_mod, mg, A)
    conv2, z2, rh2 = orc.solve(rhs[perm0], z0[perm0], tol=tol, max_iter=40)
    assert conv2 and len(rh2) == len(rh)
    np.testing.assert_allclose(rh, rh2, rtol=1e-6)
    assert np.linalg.norm(z[perm0, 0] - z2[:, 0]) <= 1e-9 * np.linalg.norm(z2)
    conv3, z3, rh3 = mg.solve(rhs, z0, None, smg.SolveOpts(tol=tol, max_iter=40, precision="mixed"))
    assert conv3 and np.linalg.norm(z3 - z) <= 1e-7 * np.linalg.norm(z)
    # hybrid smoothers at this size: the oracle with the same per-level choice (Jacobi-type sweeps are numbering-independent, GS runs in
    # the device numbering) tracks them iteration for iteration.  Chebyshev-Jacobi below 300 k rows needs no more cycles than
    # Gauss-Seidel everywhere; damped Jacobi on this anisotropic mesh needs ~45 % more (which is why it is not the benchmark's default).
    thr = 300000
    for lv in range(mg.n_levels - 1):
        orc.set_smoother(lv, "chebyshev" if mg.rows(lv) <= thr else "gs", 0.1)
    conv4, z4, rh4 = mg.solve(rhs, z0, None, smg.SolveOpts(tol=tol, max_iter=40, smoother="hybrid_chebyshev", jacobi_max_rows=thr))
    conv5, z5, rh5 = orc.solve(rhs[perm0], z0[perm0], tol=tol, max_iter=40)
    assert conv4 and conv5 and len(rh4) == len(rh5) and len(rh4) <= len(rh) + 1, (len(rh), len(rh4), len(rh5))
    np.testing.assert_allclose(rh4, rh5, rtol=1e-6)
    thr = 100000
    for lv in range(mg.n_levels - 1):
        orc.set_smoother(lv, "jacobi" if mg.rows(lv) <= thr else "gs", 0.8)
    conv6, z6, rh6 = mg.solve(rhs, z0, None, smg.SolveOpts(tol=tol, max_iter=80, smoother="hybrid", jacobi_max_rows=thr))
    conv7, z7, rh7 = orc.solve(rhs[perm0], z0[perm0], tol=tol, max_iter=80)
    assert conv6 and conv7 and len(rh6) == len(rh7) and len(rh6) > len(rh)
    np.testing.assert_allclose(rh6, rh7, rtol=1e-6)


def test_03_mg_solver_on_ogre_with_its_boundary_loop(smg_mod, oracle_mod):
    """What 03_mg_solver/main.cpp:29-75 actually runs: ogre.obj, A = -cotmatrix, the (longest) boundary loop pinned to 0,
    B = M_voronoi 1, z0 = 0, defaults tol 1e-3 / maxIter 20 -- and the same at tol 1e-10 -- against the oracle (lexicographic GS)."""
    smg, mesh = smg_mod, smg_mod.mesh
    from oracle import mesh_np as M
    V, F = mesh.read_triangle_mesh("ogre.smgm")
    V = mesh.normalize_unit_area(V, F)
    n = V.shape[0]
    assert n == 19985
    mg = smg.mg_precompute(V, F, 0.25, 500, 1)                                   # main.cpp:35-39
    A = (-mesh.cotmatrix(V, F)).tocsr()
    A.sort_indices()
    b = mesh.boundary_loop(F)
    assert len(b) == 112 and np.array_equal(np.sort(b), np.sort(M.boundary_loop(F)))
    Bv = mesh.massmatrix(V, F, "voronoi") @ np.ones(n)
    Bv[b] = 0.0
    mg.precompute(A, b)
    assert mg.rows(0) == 19873
    Ps = [mg.matrix(l, "P_full") for l in range(1, mg.n_levels)]
    orc = oracle_mod.OracleMG(Ps)
    orc.precompute(A, b)
    for tol, mx in ((1e-3, 20), (1e-10, 60)):
        conv, z, rh = mg.solve(Bv, np.zeros(n), np.zeros(len(b)), smg.SolveOpts(tol=tol, max_iter=mx))
        conv2, z2, rh2 = orc.solve(Bv, np.zeros(n), np.zeros((len(b), 1)), tol=tol, max_iter=mx)
        assert conv and conv2 and abs(len(rh) - len(rh2)) <= 2 and abs(rh[0] - rh2[0]) <= 1e-12 * rh2[0]
        assert np.linalg.norm(z - z2) <= (1e-8 if tol <= 1e-9 else 1e-3) * np.linalg.norm(z2)
        assert np.array_equal(z[b, 0], np.zeros(len(b)))
