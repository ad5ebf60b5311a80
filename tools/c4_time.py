#!/usr/bin/env python3
"""BASELINE config C4: ogre.obj, k = 64 right-hand sides, LHS = M - 0.01 L (05_example_mean_curvature_flow): V-cycle time on the
GPU for k = 1, 3, 64 and the CPU oracle's time for the same cycles."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import surface_multigrid_code_amd as smg
from oracle.oracle import OracleMG
from oracle import mesh_np as M
V, F = M.read_smgm("ogre.smgm"); V = M.normalize_unit_area(V, F)
mg = smg.mg_precompute(V, F, 0.25, 500, 1)
A = (M.massmatrix(V, F, "barycentric") - 0.01 * M.cotmatrix(V, F)).tocsr(); A.sort_indices()
mg.precompute(A)
print("ogre.obj: levels", [mg.rows(l) for l in range(mg.n_levels)], "colours", [len(mg.colors(l)) - 1 for l in range(mg.n_levels - 1)])
sm = os.environ.get("SMG_TOOL_SMOOTHER", "gs").split(":")
mg.set_smoother(sm[0], float(sm[2]) if len(sm) > 2 else 0.8, int(sm[1]) if len(sm) > 1 else 100000)
print("smoother", sm)
for k in (1, 2, 3, 4, 8, 16, 32, 64):
    us = mg.bench_vcycle(0, k, 2, 2, 200)
    print("k = %2d: %.1f us per V(2,2) cycle (%.2f us per column)" % (k, us, us / k))
o = OracleMG([mg.matrix(l, "P_full") for l in range(1, mg.n_levels)]); o.precompute(A)
rng = np.random.default_rng(0)
B = rng.uniform(-1, 1, (V.shape[0], 64)); u = np.zeros_like(B)
t = time.time(); o.vcycle(B, u); dt = time.time() - t
print("CPU oracle, k = 64: %.1f ms per cycle" % (1e3 * dt))
