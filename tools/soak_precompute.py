#!/usr/bin/env python3
"""Soak of the threaded first precompute: the same system precomputed again and again on fresh handles (the host half on its own thread, the
pool, the hand-over to the device half: a race would show as a different numbering, a different image or a different cycle), the
numbering and one V-cycle + solve hashed each time.  usage: tools/soak_precompute.py [rounds] [n_sub]"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import surface_multigrid_code_amd as smg
from problems import subdiv_problem
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n_sub = int(sys.argv[2]) if len(sys.argv) > 2 else 3
seen = {}
for kind, k in (("mcf", 1), ("poisson", 2)):
    p = subdiv_problem(kind=kind, k=k, n_sub=n_sub)
    rng = np.random.default_rng(7)
    n = p["A"].shape[0]
    t0 = time.time()
    for r in range(rounds):
        mg = smg.Hierarchy.from_prolongs(p["Ps"])
        mg.precompute(p["A"], p["known"])
        B = rng.uniform(-1, 1, (mg.rows(0), k)) if r == 0 else B
        u = rng.uniform(-1, 1, (mg.rows(0), k)) if r == 0 else u
        v = mg.vcycle(B, u)
        conv, z, rh = mg.solve(p["RHS"], p["z0"], p["known_val"], smg.SolveOpts(tol=1e-9, max_iter=40))
        hsh = hashlib.sha256(b"".join([np.ascontiguousarray(mg.perm(l)).tobytes() for l in range(mg.n_levels - 1)] +
                                      [np.ascontiguousarray(v).tobytes(), np.ascontiguousarray(z).tobytes()])).hexdigest()
        seen.setdefault((kind, k), set()).add(hsh)
        del mg
    print("%s k=%d: %d rows, %d precomputes in %.1f s, %d distinct result(s)" % (kind, k, n, rounds, time.time() - t0, len(seen[(kind, k)])), flush=True)
bad = [key for key, s in seen.items() if len(s) != 1]
print("NOT DETERMINISTIC: %s" % bad if bad else "all runs identical")
sys.exit(1 if bad else 0)
