#!/usr/bin/env python3
"""M copies of ogre.obj in one union handle: ms per outer iteration, level by level.  usage: tools/union_probe.py [M]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
import surface_multigrid_code_amd as smg
from surface_multigrid_code_amd import mesh
m = int(sys.argv[1]) if len(sys.argv) > 1 else 8
V, F = mesh.read_triangle_mesh("ogre.smgm"); V = mesh.normalize_unit_area(V, F)
mg0 = smg.mg_precompute(V, F, 0.25, 500, 1)
Mb = mesh.massmatrix(V, F, "barycentric"); A = (Mb - 0.01 * mesh.cotmatrix(V, F)).tocsr(); A.sort_indices()
hu = smg.Hierarchy.union([mg0] * m)
Am = sp.block_diag([A] * m, format="csr"); Am.sort_indices()
hu.precompute(Am)
ts = [hu.bench_vcycle(lv, 1, 2, 2, 200) for lv in range(hu.n_levels)]
for lv in range(hu.n_levels):
    print("level %d rows %7d: cycle from here %7.1f us, this level alone %7.1f us" % (lv, hu.rows(lv), ts[lv], ts[lv] - (ts[lv + 1] if lv + 1 < hu.n_levels else 0)))
