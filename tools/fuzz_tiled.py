#!/usr/bin/env python3
"""Soak of the round-3 launch paths on random MESH hierarchies (tori of random resolution and subdivision depth, random systems M + c (-L),
Poisson with random pins): relax(sweeps) through the one-launch tiles, operators filled on the device (run with SMG_DEVICE_FILL_MIN=500 to
force it at these sizes), V-cycles -- bit for bit against the oracle on the level matrices in the device numbering.
usage: [SMG_DEVICE_FILL_MIN=500] [SMG_TILED_ROWS=n] tools/fuzz_tiled.py [n_cases] [first_seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, scipy.sparse as sp
import surface_multigrid_code_amd as smg
from surface_multigrid_code_amd import mesh
from oracle.oracle import OracleMG
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 7000
bad = 0
for seed in range(seed0, seed0 + ncases):
    rng = np.random.default_rng(seed)
    nu, nv = int(rng.integers(8, 40)), int(rng.integers(8, 40))
    n_sub = int(rng.integers(1, 4))
    while nu * nv * 4 ** n_sub > 400000: n_sub -= 1
    V, F = mesh.torus(nu, nv)
    mg, Vf, Ff = smg.mg_precompute_subdiv(V, F, n_sub, n_extra_levels=0)
    Vf = mesh.normalize_unit_area(Vf, Ff)
    Lc, Mb = mesh.cotmatrix(Vf, Ff), mesh.massmatrix(Vf, Ff, "barycentric")
    A = (Mb - float(rng.uniform(0.001, 0.1)) * Lc).tocsr(); A.sort_indices()
    n = A.shape[0]
    known = rng.choice(n, int(rng.integers(1, 20)), replace=False).astype(np.int32) if rng.integers(0, 2) else None
    k = int(rng.choice([1, 1, 2, 3, 4, 5, 7]))     # the tiles take up to 7 columns, in groups of 3
    try:
        mg.precompute(A, known)
        tiled = 0
        for lv in range(mg.n_levels - 1):
            m, perm = mg.rows(lv), mg.perm(lv)
            Ai = mg.matrix(lv, "A", internal=True); Pi = mg.matrix(lv + 1, "P", internal=True)
            oi = OracleMG([Pi]); oi.precompute(Ai)
            x = rng.uniform(-1, 1, (m, k)); b = rng.uniform(-1, 1, (m, k))
            assert np.array_equal(mg.A(lv, x)[perm], oi.A(0, x[perm])), "A level %d" % lv
            for sw in (1, 2, 3):
                assert np.array_equal(mg.relax(lv, b, x, sw)[perm], oi.relax(0, b[perm], x[perm], sw)), "relax(%d) level %d" % (sw, lv)
            assert np.array_equal(mg.restrict(lv, x)[mg.perm(lv + 1)], oi.restrict(0, x[perm])), "restrict level %d" % lv
            xc = rng.uniform(-1, 1, (mg.rows(lv + 1), k))
            assert np.array_equal(mg.prolong(lv, xc)[perm], oi.prolong(0, xc[mg.perm(lv + 1)])), "prolong level %d" % lv
            if 2048 <= m <= 100000: tiled += 1
        # a whole solve against the reference algorithm in the caller's numbering
        Ps = [mg.matrix(l, "P_full") for l in range(1, mg.n_levels)]
        o = OracleMG(Ps); o.precompute(A, known)
        rhs = Mb @ rng.uniform(-1, 1, (n, k)); z0 = np.zeros((n, k))
        kv = rng.uniform(-1, 1, (len(known), k)) if known is not None else None
        a = mg.solve(rhs, z0, kv, smg.SolveOpts(tol=1e-10, max_iter=200)); r = o.solve(rhs, z0, kv, tol=1e-10, max_iter=200)
        rel = np.linalg.norm(a[1] - r[1]) / np.linalg.norm(r[1])
        # (the device sweeps run in the colour-major numbering, the oracle lexicographically: iteration counts agree to a few, more on
        #  slowly converging anisotropic cases; the bar of tests/test_gpu_parity.py)
        assert a[0] == r[0] and abs(len(a[2]) - len(r[2])) <= max(3, len(r[2]) // 4) and (not a[0] or rel <= 1e-6), \
            "solve: converged %s/%s, %d/%d iterations, last residual %.2e/%.2e, rel diff %.2e" % (a[0], r[0], len(a[2]), len(r[2]), a[2][-1], r[2][-1], rel)
        print("seed %d ok: torus %dx%d x%d -> %d rows x %d columns, %d levels, %d tiled, known %s, %d cycles" % (seed, nu, nv, n_sub, n, k, mg.n_levels, tiled, None if known is None else len(known), len(a[2]) - 1))
    except AssertionError as e:
        bad += 1
        print("seed %d FAILED: %s (torus %dx%d x%d, %d rows)" % (seed, e, nu, nv, n_sub, n))
print("failures:", bad)
sys.exit(1 if bad else 0)
