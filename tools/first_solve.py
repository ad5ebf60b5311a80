#!/usr/bin/env python3
"""The 03_mg_solver pattern end to end: precompute once, solve once.  Wall time of smg_precompute, of the FIRST smg_solve on the fresh handle (it builds
the sweep plans of the levels and captures the graphs) and of a second one.   usage: tools/first_solve.py [bunny|ogre|C3dec|C3 ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench as B
import surface_multigrid_code_amd as smg
from surface_multigrid_code_amd import mesh
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
first = True
for wl in (sys.argv[1:] or ["bunny", "ogre"]):
    if wl == "bunny":
        V, F = mesh.read_triangle_mesh("bunny.smgm"); Vf = mesh.normalize_unit_area(V, F); Ff = F
        mg = smg.mg_precompute(Vf, Ff, 0.25, 500, 1)
        Mb = mesh.massmatrix(Vf, Ff, "barycentric"); A = (Mb - 0.01 * mesh.cotmatrix(Vf, Ff)).tocsr(); A.sort_indices()
    else:
        mg, A, Mb, Vf, Ff, label, _ = B.build_workload(wl, smg, mesh)
    n = A.shape[0]
    rhs = np.asfortranarray((Mb @ np.random.default_rng(3).uniform(-1, 1, n))[:, None]); z0 = np.zeros_like(rhs)
    if first:      # load the code object etc. on a throw-away handle of the same kind
        m0 = smg.mg_precompute(Vf, Ff, 0.25, 500, 1) if wl in ("bunny", "ogre") else None
        if m0 is not None: m0.precompute(A); m0.solve(rhs, z0, None, smg.SolveOpts(tol=1e-3, max_iter=20)); del m0
        first = False
    t = time.perf_counter(); mg.precompute(A); tp = time.perf_counter() - t
    o = smg.SolveOpts(tol=1e-3, max_iter=20, use_graph=int(os.environ.get("SMG_TOOL_USE_GRAPH", "1")))
    t = time.perf_counter(); r1 = mg.solve(rhs, z0, None, o); t1 = time.perf_counter() - t
    t = time.perf_counter(); r2 = mg.solve(rhs, z0, None, o); t2 = time.perf_counter() - t
    more = []
    for _ in range(3):
        t = time.perf_counter(); mg.solve(rhs, z0, None, o); more.append(1e3 * (time.perf_counter() - t))
    print("%-6s %8d rows: smg_precompute %8.2f ms | first smg_solve %8.2f ms | second %7.2f ms (%d cycles), then %s" % (wl, n, 1e3 * tp, 1e3 * t1, 1e3 * t2, len(r2[2]) - 1, " ".join("%.2f" % x for x in more)), flush=True)
