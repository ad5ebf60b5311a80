#!/usr/bin/env python3
"""dense coarse solve at C3 (3 952 unknowns) and C4 (ogre.obj) with k columns, us (graph-replayed): python tools/coarse_k.py [k ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench as B
import surface_multigrid_code_amd as smg
from surface_multigrid_code_amd import mesh
ks = [int(a) for a in sys.argv[1:]] or [64, 32, 16]
mg, A, Mb, Vf, Ff, label, _ = B.build_workload("C3", smg, mesh)
mg.precompute(A)
L = mg.n_levels - 1
rng = np.random.default_rng(1)
for k in ks:
    t = min(mg.bench_vcycle(L, k, 2, 2, 30) for _ in range(3))
    Bm = rng.uniform(-1, 1, (mg.rows(L), k))
    x = mg.coarse_solve(Bm, np.zeros_like(Bm))
    print("C3 coarsest level (%d unknowns), k = %d: %.1f us per coarse solve; checksum %.17g" % (mg.rows(L), k, t, float(np.abs(x).sum())), flush=True)
