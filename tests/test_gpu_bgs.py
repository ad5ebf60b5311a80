"""GPU (-m gpu): relax() for many right-hand-side columns (k a multiple of 16) -- block Gauss-Seidel (csrc/smg_bgs.hpp).

The reference's relax() with k > 1 columns (src/mg_VCycle.cpp:161-177) is k independent lexicographic sweeps.  With k % 64 == 0 libsmg
sweeps big levels in the order (block colour, block, position in the block): the checker is the oracle -- the reference's lexicographic
loop -- on the system permuted into exactly that order, and the comparison is bitwise."""
import numpy as np
import pytest

from problems import subdiv_problem
from test_gpu_parity import smg  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def bgs_oracle(oracle_mod, mg, lv, order):
    """oracle whose level 0 is level lv in the bgs order (perm_int: position -> internal row)"""
    A = mg.matrix(lv, "A", internal=True).tocsr()
    P = mg.matrix(lv + 1, "P", internal=True).tocsr()
    o = oracle_mod.OracleMG([P[order]])
    o.precompute(A[order][:, order].tocsr())
    return o


def check_plan(mg, lv, k):
    info = mg.block_gs_order(lv, k)
    assert info is not None
    n = mg.rows(lv)
    rows, bp, cp = info["rows"], info["blk_ptr"], info["color_ptr"]
    assert sorted(rows.tolist()) == list(range(n)) and bp[0] == 0 and bp[-1] == n and cp[0] == 0 and cp[-1] == len(bp) - 1
    assert (np.diff(bp) > 0).all() and np.diff(bp).max() <= 64
    # blocks of one colour share no matrix entry
    A = mg.matrix(lv, "A", internal=True).tocoo()
    blk_of_pos = np.repeat(np.arange(len(bp) - 1), np.diff(bp))
    blk = np.empty(n, np.int64); blk[rows] = blk_of_pos
    col_of_blk = np.repeat(np.arange(len(cp) - 1), np.diff(cp))
    cross = blk[A.row] != blk[A.col]
    assert (col_of_blk[blk[A.row[cross]]] != col_of_blk[blk[A.col[cross]]]).all(), "two coupled blocks share a colour"
    return info


@pytest.mark.parametrize("kind,k", [("mcf", 64), ("poisson", 128), ("mcf", 16), ("poisson", 48)])
def test_block_gauss_seidel_is_the_lexicographic_sweep_in_the_block_order(smg, oracle_mod, kind, k):
    p = subdiv_problem(kind=kind, k=k, n_sub=2, n_pins=40 if kind == "poisson" else 0)
    mg = smg.Hierarchy.from_prolongs(p["Ps"])
    mg.set_block_gs(0)                       # every smoothed level
    mg.precompute(p["A"], p["known"])
    rng = np.random.default_rng(5)
    for lv in range(mg.n_levels - 1):
        info = check_plan(mg, lv, k)
        assert 0.0 < info["rim"] < 2.0 and 0.4 < info["fill"] <= 1.0
        n = mg.rows(lv)
        perm = mg.perm(lv)                   # internal -> caller
        order = info["rows"]                 # position -> internal
        oi = bgs_oracle(oracle_mod, mg, lv, order)
        x, b = rng.uniform(-1, 1, (n, k)), rng.uniform(-1, 1, (n, k))
        to_bgs = perm[order]                 # position -> caller
        for iters in (1, 2):
            got = mg.relax(lv, b, x, iters)[to_bgs]
            ref = oi.relax(0, b[to_bgs], x[to_bgs], iters)
            assert np.array_equal(got, ref), "block Gauss-Seidel not bit-exact on level %d (%d sweeps)" % (lv, iters)
    # fewer columns, or not a multiple of 16: the multi-colour path, untouched
    assert mg.block_gs_order(0, 8) is None and mg.block_gs_order(0, 40) is None
    mg.set_block_gs(-1)
    assert mg.block_gs_order(0, k) is None


def test_solve_with_block_gauss_seidel_matches_the_reference_algorithm(smg, oracle_mod):
    """64 columns through the block-sequential sweeps: same solution as the oracle's lexicographic cycle (to the tolerance: the sweep order
    differs), same cycle count to +-2, and the same solution as the multi-colour path of the same handle."""
    p = subdiv_problem(kind="mcf", k=64, n_sub=2)
    mg = smg.Hierarchy.from_prolongs(p["Ps"])
    mg.precompute(p["A"])
    o = smg.SolveOpts(tol=1e-9, max_iter=40)
    conv_c, z_c, rh_c = mg.solve(p["RHS"], p["z0"], None, o)          # default: the multi-colour path
    assert mg.block_gs_order(0, 64) is None
    mg.set_block_gs(0)
    conv, z, rh = mg.solve(p["RHS"], p["z0"], None, o)
    assert mg.block_gs_order(0, 64) is not None
    orc = oracle_mod.OracleMG(p["Ps"]); orc.precompute(p["A"])
    conv2, z2, rh2 = orc.solve(p["RHS"], p["z0"], tol=1e-9, max_iter=40)
    assert conv and conv2 and conv_c and abs(len(rh) - len(rh2)) <= 2
    assert abs(rh[0] - rh2[0]) <= 1e-12 * rh2[0]
    assert np.linalg.norm(z - z2) <= 1e-7 * np.linalg.norm(z2) and np.linalg.norm(z - z_c) <= 1e-7 * np.linalg.norm(z_c)
    # deterministic, and a value-only re-precompute refreshes the plan's copy of the values
    conv3, z3, rh3 = mg.solve(p["RHS"], p["z0"], None, o)
    assert np.array_equal(z, z3) and np.array_equal(rh, rh3)
    import scipy.sparse as sp
    A2 = (p["A"] + 0.25 * sp.diags(p["A"].diagonal())).tocsr(); A2.sort_indices()
    mg.precompute(A2)
    orc.precompute(A2)
    conv4, z4, rh4 = mg.solve(p["RHS"], p["z0"], None, o)
    conv5, z5, rh5 = orc.solve(p["RHS"], p["z0"], tol=1e-9, max_iter=40)
    assert conv4 and conv5 and np.linalg.norm(z4 - z5) <= 1e-6 * np.linalg.norm(z5)
    info = mg.block_gs_order(0, 64)
    x, b = np.random.default_rng(1).uniform(-1, 1, (mg.rows(0), 64)), np.random.default_rng(2).uniform(-1, 1, (mg.rows(0), 64))
    to_bgs = mg.perm(0)[info["rows"]]
    assert np.array_equal(mg.relax(0, b, x, 1)[to_bgs], bgs_oracle(oracle_mod, mg, 0, info["rows"]).relax(0, b[to_bgs], x[to_bgs], 1))
