#!/usr/bin/env python3
"""One-off soak: random irregular SPD systems, random prolongation hierarchies, random constraint sets and column counts;
kernels must be bit-exact against the oracle on the level matrices in the device numbering, solves must agree with the reference
algorithm.   usage: tools/fuzz_parity.py [n_cases] [first_seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, scipy.sparse as sp
import surface_multigrid_code_amd as smg
from oracle.oracle import OracleMG
src = open(os.path.join(ROOT, "tests", "test_gpu_parity.py")).read()
ns = {"np": np, "sp": sp}
exec(src[src.index("def _random_spd_hierarchy"):src.index('@pytest.mark.parametrize("seed,n,levels,k,hub"')], ns)
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
bad = 0
for seed in range(seed0, seed0 + ncases):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(5, 5000)); levels = int(rng.integers(2, 5)); k = int(rng.choice([1, 1, 2, 3, 5, 8, 9, 17, 40])); hub = bool(rng.integers(0, 2)) and n > 200
    while n // (3 ** (levels - 1)) < 2: levels -= 1
    if levels < 2: continue
    A, Ps = ns["_random_spd_hierarchy"](rng, n, levels, hub)
    known = None
    if rng.integers(0, 2):
        known = rng.choice(n, int(rng.integers(1, max(2, n // 10))), replace=False).astype(np.int32)
    prec = "mixed" if rng.integers(0, 4) == 0 else "f64"
    try:
        mg = smg.Hierarchy.from_prolongs(Ps); mg.precompute(A, known)
        o = OracleMG(Ps); o.precompute(A, known)
        for lv in range(mg.n_levels - 1):
            m = mg.rows(lv); perm = mg.perm(lv)
            Ai = mg.matrix(lv, "A", internal=True); Pi = mg.matrix(lv + 1, "P", internal=True)
            oi = OracleMG([Pi]); oi.precompute(Ai)
            x = rng.uniform(-1, 1, (m, k)); b = rng.uniform(-1, 1, (m, k))
            assert np.array_equal(mg.A(lv, x)[perm], oi.A(0, x[perm])), "A"
            # Gauss-Seidel in the order the device sweeps the level (gs_order: colour-major, or the piece / block order -- tests/test_gpu_parity.py: gs_bit_exact)
            order = mg.gs_order(lv, k); to = perm[order]
            og = OracleMG([sp.csr_matrix(Pi)[order]]); og.precompute(sp.csr_matrix(Ai)[order][:, order].tocsr())
            assert np.array_equal(mg.relax(lv, b, x, 2)[to], og.relax(0, b[to], x[to], 2)), "relax"
            assert np.array_equal(mg.restrict(lv, x)[mg.perm(lv + 1)], oi.restrict(0, x[perm])), "restrict"
            # the Jacobi-type smoothers, bit for bit as well (random damping / interval fraction, odd and even sweep counts)
            w = float(rng.uniform(0.4, 1.0)); it = int(rng.integers(1, 4))
            mg.set_smoother("jacobi", w); oi.set_smoother(0, "jacobi", w)
            assert np.array_equal(mg.relax(lv, b, x, it)[perm], oi.relax(0, b[perm], x[perm], it)), "jacobi"
            fr = float(rng.uniform(0.05, 0.5))
            mg.set_smoother("chebyshev", cheby_fraction=fr); oi.set_smoother(0, "chebyshev", fr)
            assert mg.spectral_bound(lv) == oi.spectral_bound(0), "gershgorin"
            assert np.array_equal(mg.relax(lv, b, x, it)[perm], oi.relax(0, b[perm], x[perm], it)), "chebyshev"
            mg.set_smoother("gs"); oi.set_smoother(0, "gs")
        rhs, z0 = rng.uniform(-1, 1, (n, k)), rng.uniform(-1, 1, (n, k))
        kv = rng.uniform(-1, 1, (len(known), k)) if known is not None else None
        sm = str(rng.choice(["gs", "gs", "chebyshev", "hybrid_chebyshev", "jacobi"]))
        thr = int(rng.integers(1, n + 1))
        for lv in range(mg.n_levels - 1):      # the oracle's per-level twin of the selection (random SPD systems: Gershgorin keeps Chebyshev stable;
            small = sm in ("jacobi", "chebyshev") or (sm.startswith("hybrid") and mg.rows(lv) <= thr)   # damped Jacobi may need many cycles)
            o.set_smoother(lv, ("chebyshev" if "chebyshev" in sm else "jacobi") if small else "gs", 0.1 if "chebyshev" in sm else 0.5)
        a = mg.solve(rhs, z0, kv, smg.SolveOpts(tol=1e-9, max_iter=600, precision=prec, smoother=sm, omega=0.5, jacobi_max_rows=thr))
        b = o.solve(rhs, z0, kv, tol=1e-9, max_iter=600)
        rel = np.linalg.norm(a[1] - b[1]) / max(np.linalg.norm(b[1]), 1e-300)
        ok = a[0] and b[0] and rel < 1e-6
        print("seed %d n=%d L=%d k=%d hub=%d known=%s %s %s: its %d/%d rel %.1e %s" % (seed, n, len(Ps) + 1, k, hub, None if known is None else len(known), prec, sm, len(a[2]), len(b[2]), rel, "ok" if ok else "MISMATCH"))
        bad += not ok
    except Exception as e:
        bad += 1
        print("seed %d n=%d L=%d k=%d hub=%d known=%s %s: EXCEPTION %s" % (seed, n, levels, k, hub, None if known is None else len(known), prec, str(e)[:200]))
print("failures:", bad)
