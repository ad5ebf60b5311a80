#!/usr/bin/env python3
"""Time of one level visit (V(2,2): two relax(2), residual, both transfers) against the level's size, subdivision hierarchies of tori of growing size:
looks for cliffs at the thresholds of the launch shortcuts (overlapped tiling 2 048 .. 100 000 rows, one-XCD colour sweeps, whole-pitch look-ahead).
usage: tools/size_sweep.py [k]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import surface_multigrid_code_amd as smg
from surface_multigrid_code_amd import mesh
k = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rows = []
tori = ((16, 12), (20, 16), (24, 20), (30, 24), (36, 30), (44, 36), (52, 44), (64, 50), (76, 64), (90, 76), (110, 90), (128, 110))
if os.environ.get("SMG_TOOL_TORI"): tori = tuple(tuple(int(x) for x in t.split("x")) for t in os.environ["SMG_TOOL_TORI"].split(","))
for nu, nv in tori:
    V, F = mesh.torus(nu, nv)
    mg, Vf, Ff = smg.mg_precompute_subdiv(V, F, 3, n_extra_levels=0)
    Vf = mesh.normalize_unit_area(Vf, Ff)
    A = (mesh.massmatrix(Vf, Ff, "barycentric") - 0.01 * mesh.cotmatrix(Vf, Ff)).tocsr(); A.sort_indices()
    torch.zeros(1, device="cuda")
    mg.precompute(A)
    ts = [mg.bench_vcycle(lv, k, 2, 2, 100) for lv in range(mg.n_levels)]
    for lv in range(mg.n_levels - 1):
        rows.append((mg.rows(lv), ts[lv] - ts[lv + 1], lv, nu, nv))
    del mg
for r in sorted(rows):
    print("rows %8d: %7.1f us per visit  (level %d of torus %d x %d x3)" % r)
