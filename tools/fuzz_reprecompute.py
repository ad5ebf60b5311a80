#!/usr/bin/env python3
"""Soak of the value-only re-precompute path and of constraint sets on decimated hierarchies.
(1) random irregular systems: precompute(A1) -> precompute(A2 with the same pattern) [device path] must give the same bits as a
    fresh hierarchy that sees A2 first [host path]; with and without constraints.
(2) mesh hierarchies from mg_precompute with the boundary loop (or random pins) as constraints: GPU solve vs the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, scipy.sparse as sp
import surface_multigrid_code_amd as smg
from oracle.oracle import OracleMG
from oracle import mesh_np as M
src = open(os.path.join(ROOT, "tests", "test_gpu_parity.py")).read()
ns = {"np": np, "sp": sp}
exec(src[src.index("def _random_spd_hierarchy"):src.index('@pytest.mark.parametrize("seed,n,levels,k,hub"')], ns)
bad = 0
for seed in range(300, 300 + int(sys.argv[1]) if len(sys.argv) > 1 else 330):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(50, 4000)); levels = int(rng.integers(2, 5)); k = int(rng.choice([1, 2, 3, 8]))
    while n // (3 ** (levels - 1)) < 2: levels -= 1
    A, Ps = ns["_random_spd_hierarchy"](rng, n, levels, bool(rng.integers(0, 2)) and n > 200)
    known = rng.choice(n, int(rng.integers(1, max(2, n // 8))), replace=False).astype(np.int32) if rng.integers(0, 2) else None
    D = sp.diags(rng.uniform(0.8, 1.25, n))
    A2 = sp.csr_matrix(D @ A @ D); A2.sort_indices()
    assert np.array_equal(A2.indices, A.indices)
    rhs, z0 = rng.uniform(-1, 1, (n, k)), rng.uniform(-1, 1, (n, k))
    kv = rng.uniform(-1, 1, (len(known), k)) if known is not None else None
    o = smg.SolveOpts(tol=1e-10, max_iter=100)
    mg = smg.Hierarchy.from_prolongs(Ps); mg.precompute(A, known); mg.solve(rhs, z0, kv, o)
    mg.precompute(A2, known); a1 = mg.solve(rhs, z0, kv, o)      # builds the recipes
    mg.precompute(A, known); mg.precompute(A2, known); a2 = mg.solve(rhs, z0, kv, o)   # steady value-only path
    fresh = smg.Hierarchy.from_prolongs(Ps); fresh.precompute(A2, known); b = fresh.solve(rhs, z0, kv, o)
    ok = a1[0] and np.array_equal(a1[1], a2[1]) and np.array_equal(a1[2], a2[2]) and len(a1[2]) == len(b[2]) and \
        np.linalg.norm(a1[1] - b[1]) <= 1e-9 * np.linalg.norm(b[1])
    same_bits = np.array_equal(a1[1], b[1])
    print("seed %d n=%d L=%d k=%d known=%s: %s (bitwise equal to the host path: %s)" % (seed, n, len(Ps) + 1, k, None if known is None else len(known), "ok" if ok else "MISMATCH", same_bits))
    bad += not ok
for name, ratio in [("bunny.smgm", 0.25), ("bunny.smgm", 0.5), ("ogre.smgm", 0.25), ("ogre_sim.smgm", 0.35), ("bunny_15K_init.smgm", 0.25)]:
    V, F = M.read_smgm(name); V = M.normalize_unit_area(V, F)
    rng = np.random.default_rng(7)
    mg = smg.mg_precompute(V, F, ratio=ratio, nVCoarsest=200)
    Ps = [mg.matrix(l, "P_full") for l in range(1, mg.n_levels)]
    A = (-M.cotmatrix(V, F)).tocsr(); A.sort_indices()
    b = M.boundary_loop(F)
    if len(b) == 0: b = rng.choice(V.shape[0], 30, replace=False).astype(np.int32)
    B = np.repeat((M.massmatrix(V, F, "voronoi") @ np.ones(V.shape[0]))[:, None], 2, axis=1); B[b] = 0
    kv = rng.uniform(-0.1, 0.1, (len(b), 2))
    mg.precompute(A, b); orc = OracleMG(Ps); orc.precompute(A, b)
    z0 = np.zeros_like(B)
    a = mg.solve(B, z0, kv, smg.SolveOpts(tol=1e-10, max_iter=80)); c = orc.solve(B, z0, kv, tol=1e-10, max_iter=80)
    rel = np.linalg.norm(a[1] - c[1]) / np.linalg.norm(c[1])
    ok = a[0] and c[0] and rel < 1e-7 and abs(len(a[2]) - len(c[2])) <= 3
    print("%s ratio %.2f levels %d constraints %d: its %d/%d rel %.1e %s" % (name, ratio, mg.n_levels, len(b), len(a[2]), len(c[2]), rel, "ok" if ok else "MISMATCH"))
    bad += not ok
print("failures:", bad)
