#!/usr/bin/env python3
"""Wall time of the drop-in call smg_solve (host buffers) on a SMALL mesh -- the reference's demo sizes -- by tolerance: what the call costs beyond its
V-cycles (staging copies, graph launches, polling, the read-back).   usage: tools/small_solve_latency.py [ogre|bunny]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import surface_multigrid_code_amd as smg
from surface_multigrid_code_amd import mesh
name = sys.argv[1] if len(sys.argv) > 1 else "ogre"
V, F = mesh.read_triangle_mesh(name + ".smgm"); V = mesh.normalize_unit_area(V, F)
mg = smg.mg_precompute(V, F, 0.25, 500, 1)
Mb = mesh.massmatrix(V, F, "barycentric"); A = (Mb - 0.01 * mesh.cotmatrix(V, F)).tocsr(); A.sort_indices()
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
mg.precompute(A)
n = A.shape[0]
rhs = np.asfortranarray((Mb @ np.random.default_rng(3).uniform(-1, 1, n))[:, None]); z0 = np.zeros_like(rhs)
cyc = mg.bench_vcycle(0, 1, 2, 2, 200)
print("%s: %d rows, V(2,2) cycle %.1f us graph-replayed" % (name, n, cyc))
for tol in [float(x) for x in os.environ.get("SMG_TOOL_TOLS", "1e-1,1e-3,1e-6,1e-10").split(",")]:
    for ce in ((0,) if os.environ.get("SMG_TOOL_TOLS") else (0, 1)):
        o = smg.SolveOpts(tol=tol, max_iter=60, check_every=ce)
        mg.solve(rhs, z0, None, o)
        ts = []
        for _ in range(20):
            t = time.perf_counter(); conv, z, rh = mg.solve(rhs, z0, None, o); ts.append(time.perf_counter() - t)
        its = len(rh) - 1
        print("tol %.0e check_every %d: %7.3f ms per solve (median; min %.3f), %2d cycles -> %.3f ms of cycles, %.3f ms of everything else"
              % (tol, ce, 1e3 * np.median(ts), 1e3 * min(ts), its, 1e-3 * cyc * its, 1e3 * np.median(ts) - 1e-3 * cyc * its))
