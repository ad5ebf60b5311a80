"""GPU (-m gpu): relax() on the Galerkin levels of the reference's own hierarchies (mg_precompute: SSP decimation, A_l = PT A P with 18 - 30
entries per row) -- wave Gauss-Seidel (csrc/smg_wgs.hpp): pieces of <= 64 rows, one launch per piece colour, one wavefront per piece.

It is the reference's lexicographic sweep (src/mg_VCycle.cpp:146-160) on the numbering (piece colour, piece, local colour, row): the checker is the
oracle -- the reference's loop -- on the system permuted into exactly that order, and the comparison is bitwise."""
import numpy as np
import pytest

from oracle import mesh_np as M
from test_gpu_parity import smg  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def decimated(smg, name, k, kind="mcf", nVCoarsest=200, n_pins=0, seed=0):
    """the reference's hierarchy of a mesh (03_mg_solver/main.cpp:35-39) + one of its callers' systems"""
    V, F = M.read_smgm(name)
    V = M.normalize_unit_area(V, F)
    mg = smg.mg_precompute(V, F, 0.25, nVCoarsest, 1)
    n = V.shape[0]
    L = M.cotmatrix(V, F)
    rng = np.random.default_rng(seed)
    known = None
    if kind == "mcf":
        Mb = M.massmatrix(V, F, "barycentric")
        A = (Mb - 0.01 * L).tocsr()
        RHS = Mb @ rng.uniform(-1, 1, (n, k))
    else:
        A = (-L).tocsr()
        known = M.boundary_loop(F)
        if n_pins or len(known) == 0:
            known = np.sort(rng.choice(n, max(n_pins, 8), replace=False)).astype(np.int32)
        RHS = np.repeat((M.massmatrix(V, F, "voronoi") @ np.ones(n))[:, None], k, axis=1) * rng.uniform(0.5, 1.5, (1, k))
    A.sort_indices()
    return mg, A, np.asfortranarray(RHS), known


def order_oracle(oracle_mod, mg, lv, order):
    """oracle whose level 0 is level lv in the given order (position -> internal row); Galerkin levels are not bit-symmetric: the reference's sweep
    walks COLUMN i of the CSC matrix (src/mg_VCycle.cpp:149-155), which the oracle does on what it is given"""
    A = mg.matrix(lv, "A", internal=True).tocsr()
    P = mg.matrix(lv + 1, "P", internal=True).tocsr()
    o = oracle_mod.OracleMG([P[order]])
    o.precompute(A[order][:, order].tocsr())
    return o


def check_plan(mg, lv, k):
    info = mg.wave_gs_order(lv, k)
    assert info is not None
    n = mg.rows(lv)
    rows, bp, cp = info["rows"], info["piece_ptr"], info["color_ptr"]
    assert sorted(rows.tolist()) == list(range(n)) and bp[0] == 0 and bp[-1] == n and cp[0] == 0 and cp[-1] == len(bp) - 1
    assert (np.diff(bp) > 0).all() and np.diff(bp).max() <= 64
    A = mg.matrix(lv, "A", internal=True).tocoo()
    pc_of_pos = np.repeat(np.arange(len(bp) - 1), np.diff(bp))
    pc = np.empty(n, np.int64); pc[rows] = pc_of_pos
    col_of_pc = np.repeat(np.arange(len(cp) - 1), np.diff(cp))
    cross = pc[A.row] != pc[A.col]
    assert (col_of_pc[pc[A.row[cross]]] != col_of_pc[pc[A.col[cross]]]).all(), "two coupled pieces share a colour"
    return info


@pytest.mark.parametrize("name,kind,k", [("ogre.smgm", "mcf", 1), ("bunny.smgm", "poisson", 1), ("bunny.smgm", "mcf", 3), ("bunny_15K_init.smgm", "poisson", 2),
                                         ("ogre.smgm", "mcf", 7), ("bunny.smgm", "mcf", 5), ("ogre.smgm", "mcf", 8), ("bunny.smgm", "poisson", 18)])
def test_wave_gauss_seidel_is_the_lexicographic_sweep_in_the_piece_order(smg, oracle_mod, name, kind, k):
    mg, A, RHS, known = decimated(smg, name, k, kind, n_pins=40 if name == "bunny_15K_init.smgm" else 0)
    mg.precompute(A, known)
    rng = np.random.default_rng(5)
    seen = 0
    for lv in range(mg.n_levels - 1):
        info = mg.wave_gs_order(lv, k)
        if lv >= 1 and mg.rows(lv) >= 512:
            assert info is not None, "Galerkin level %d (%d rows, %d colours) does not sweep piece-wise" % (lv, mg.rows(lv), len(mg.colors(lv)) - 1)
        if info is None:
            continue
        seen += 1
        info = check_plan(mg, lv, k)
        assert len(info["color_ptr"]) - 1 <= 8 and info["phases_max"] <= 16 and 0.0 < info["rim"] < 4.0
        n = mg.rows(lv)
        perm = mg.perm(lv)                   # internal -> caller
        order = info["rows"]                 # position -> internal
        oi = order_oracle(oracle_mod, mg, lv, order)
        x, b = rng.uniform(-1, 1, (n, k)), rng.uniform(-1, 1, (n, k))
        to_wgs = perm[order]                 # position -> caller
        for iters in (1, 2):
            got = mg.relax(lv, b, x, iters)[to_wgs]
            ref = oi.relax(0, b[to_wgs], x[to_wgs], iters)
            assert np.array_equal(got, ref), "wave Gauss-Seidel not bit-exact on level %d (%d sweeps)" % (lv, iters)
    assert seen >= 1
    # the order of a level's sweep does not depend on the number of columns (a column-sharded solve must iterate like the fused one)
    for lv in range(mg.n_levels - 1):
        a, b2 = mg.wave_gs_order(lv, k), mg.wave_gs_order(lv, 64)
        assert (a is None) == (b2 is None) and (a is None or np.array_equal(a["rows"], b2["rows"]))
    mg.set_wave_gs("never")
    assert all(mg.wave_gs_order(lv, k) is None for lv in range(mg.n_levels - 1))


@pytest.mark.parametrize("name,kind,k,tol", [("ogre.smgm", "mcf", 1, 1e-10), ("bunny.smgm", "poisson", 1, 1e-10), ("bunny_15K_init.smgm", "mcf", 3, 5e-7)])
def test_solve_with_wave_gauss_seidel_matches_the_reference_algorithm(smg, oracle_mod, name, kind, k, tol):
    """the drop-in solve on the reference's hierarchy: same solution as the oracle's lexicographic cycle (to the tolerance: the sweep order differs),
    same cycle count to +-2, same solution as the multi-colour path of the same handle; deterministic; a value-only re-precompute refreshes the plan"""
    mg, A, RHS, known = decimated(smg, name, k, kind)
    mg.precompute(A, known)
    n = A.shape[0]
    z0 = np.zeros((n, k), order="F")
    kv = None if known is None else np.zeros((len(known), k))
    o = smg.SolveOpts(tol=tol, max_iter=60)
    conv, z, rh = mg.solve(RHS, z0, kv, o)
    assert any(mg.wave_gs_order(lv, k) is not None for lv in range(mg.n_levels - 1))
    orc = oracle_mod.OracleMG([mg.matrix(l, "P_full") for l in range(1, mg.n_levels)])
    orc.precompute(A, known)
    conv2, z2, rh2 = orc.solve(RHS, z0, kv, tol=tol, max_iter=60)
    assert conv and conv2 and abs(len(rh) - len(rh2)) <= 2
    assert abs(rh[0] - rh2[0]) <= 1e-12 * rh2[0]
    scale = max(np.linalg.norm(z2), 1e-300)
    assert np.linalg.norm(z - z2) <= (1e-7 if tol <= 1e-9 else 1e-3) * scale
    conv3, z3, rh3 = mg.solve(RHS, z0, kv, o)
    assert np.array_equal(z, z3) and np.array_equal(rh, rh3)
    mg.set_wave_gs("never")
    conv_c, z_c, rh_c = mg.solve(RHS, z0, kv, o)
    assert conv_c and abs(len(rh) - len(rh_c)) <= 2 and np.linalg.norm(z - z_c) <= (1e-7 if tol <= 1e-9 else 1e-3) * scale
    mg.set_wave_gs("auto")
    import scipy.sparse as sp
    A2 = (A + 0.25 * sp.diags(A.diagonal())).tocsr(); A2.sort_indices()
    mg.precompute(A2, known)
    orc.precompute(A2, known)
    conv4, z4, rh4 = mg.solve(RHS, z0, kv, o)
    conv5, z5, rh5 = orc.solve(RHS, z0, kv, tol=tol, max_iter=60)
    assert conv4 and conv5 and np.linalg.norm(z4 - z5) <= (1e-6 if tol <= 1e-9 else 1e-3) * max(np.linalg.norm(z5), 1e-300)
    lv = next(l for l in range(mg.n_levels - 1) if mg.wave_gs_order(l, k) is not None)
    info = mg.wave_gs_order(lv, k)
    nl = mg.rows(lv)
    x, b = np.random.default_rng(1).uniform(-1, 1, (nl, k)), np.random.default_rng(2).uniform(-1, 1, (nl, k))
    to_wgs = mg.perm(lv)[info["rows"]]
    assert np.array_equal(mg.relax(lv, b, x, 1)[to_wgs], order_oracle(oracle_mod, mg, lv, info["rows"]).relax(0, b[to_wgs], x[to_wgs], 1))
