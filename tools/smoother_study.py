#!/usr/bin/env python3
"""Convergence study of per-level smoother choices (CPU, oracle in all-core mode; no GPU needed).
    python tools/smoother_study.py [C3|small|n_sub] 
Prints cycles to 1e-10 and the asymptotic factor for GS everywhere vs damped Jacobi from level L on."""
import os, sys, time
import numpy as np
import scipy.sparse as sp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import surface_multigrid_code_amd as smg
from surface_multigrid_code_amd import mesh
from oracle.oracle import OracleMG

n_sub = int(sys.argv[1]) if len(sys.argv) > 1 else 2
kind = sys.argv[2] if len(sys.argv) > 2 else "mcf"
if len(sys.argv) > 3 and sys.argv[3] == "torus":
    import bench as B
    V, F = mesh.torus(64, 64)
    mg, Vf, Ff = smg.mg_precompute_subdiv(V, F, n_sub, n_extra_levels=0)
    Vf = mesh.normalize_unit_area(B._onto_torus(Vf), Ff)
else:
    V, F = mesh.read_triangle_mesh("bunny_15K_init.smgm")
    V = mesh.normalize_unit_area(V, F)
    mg, Vf, Ff = smg.mg_precompute_subdiv(V, F, n_sub, ratio=0.25, nVCoarsest=1000, n_extra_levels=1)
L = mesh.cotmatrix(Vf, Ff)
n = Vf.shape[0]
rng = np.random.default_rng(100)
known = None
if kind == "mcf":
    Mb = mesh.massmatrix(Vf, Ff, "barycentric")
    A = (Mb - 0.01 * L).tocsr()
    rhs = Mb @ rng.uniform(-1, 1, n)
    z0 = np.zeros(n)
else:
    A = (-L).tocsr()
    Mv = mesh.massmatrix(Vf, Ff, "voronoi")
    known = np.sort(np.random.default_rng(0).choice(n, 346, replace=False)).astype(np.int32)
    rhs = Mv @ np.ones(n)
    z0 = rng.uniform(-1, 1, n)
A.sort_indices()
Ps = [mg.matrix(l, "P_full") for l in range(1, mg.n_levels)]
orc = OracleMG(Ps)
orc.precompute(A, known)
orc.L.orc_enable_parallel(orc.h, 1, 32)
nl = mg.n_levels
print("levels", [orc.rows(l) for l in range(nl)])

def run(desc, cfg):
    for lv in range(nl - 1):
        k, w = cfg.get(lv, ("gs", 1.0))
        orc.set_smoother(lv, k, w)
    t0 = time.time()
    kv = np.zeros((346, 1)) if known is not None else None
    conv, z, rh = orc.solve(rhs, z0, known_val=kv, tol=1e-10, max_iter=60)
    fac = (rh[-1] / rh[max(len(rh) - 4, 0)]) ** (1.0 / min(3, len(rh) - 1)) if len(rh) > 1 else 0
    print("%-44s cycles %2d  conv %d  last %.2e  asym factor %.3f  (%.1fs)" % (desc, len(rh) - 1, conv, rh[-1], fac, time.time() - t0), flush=True)

run("GS everywhere", {})
for w in (0.6, 0.7, 0.8, 0.9):
    run("Jacobi w=%.1f on levels >= 1" % w, {lv: ("jacobi", w) for lv in range(1, nl - 1)})
for w in (0.7, 0.8):
    run("Jacobi w=%.1f on levels >= 2" % w, {lv: ("jacobi", w) for lv in range(2, nl - 1)})
for w in (0.7,):
    run("Jacobi w=%.1f everywhere" % w, {lv: ("jacobi", w) for lv in range(0, nl - 1)})
for w in (0.8, 0.9, 1.0):
    run("Jacobi w=%.1f on the last smoothed level only" % w, {nl - 2: ("jacobi", w)})
    if nl >= 4:
        run("Jacobi w=%.1f on the last two smoothed levels" % w, {nl - 2: ("jacobi", w), nl - 3: ("jacobi", w)})
