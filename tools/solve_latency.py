#!/usr/bin/env python3
"""Wall time of smg_solve (host buffers, the drop-in call) at one workload for several polling cadences."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench as B
import surface_multigrid_code_amd as smg
from surface_multigrid_code_amd import mesh
mg, A, Mb, Vf, Ff, label, _ = B.build_workload(sys.argv[1] if len(sys.argv) > 1 else "C3", smg, mesh)
mg.precompute(A)
n = A.shape[0]
rng = np.random.default_rng(3)
rhs = np.asfortranarray((Mb @ rng.uniform(-1, 1, n))[:, None]); z0 = np.zeros_like(rhs)
print(label)
sm = os.environ.get("SMG_TOOL_SMOOTHER", "gs")
for ce in (0, 1, 2, 4, 20):     # 0: adaptive (the default)
    o = smg.SolveOpts(tol=1e-10, max_iter=20, check_every=ce, smoother=sm, jacobi_max_rows=300000)
    mg.solve(rhs, z0, None, o)
    ts = []
    for _ in range(5):
        t = time.time(); conv, z, rh = mg.solve(rhs, z0, None, o); ts.append(time.time() - t)
    print("check_every %2d: %.2f ms per solve, %d residuals recorded (%.3f ms per iteration)" % (ce, 1e3 * min(ts), len(rh), 1e3 * min(ts) / len(rh)))
