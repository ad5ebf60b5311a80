#!/usr/bin/env python3
"""us per colour launch of the small levels as a function of the graph's length (sweeps per captured relax()), C3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
import surface_multigrid_code_amd as smg
from surface_multigrid_code_amd import mesh
mg, A, Mb, Vf, Ff, label, _ = B.build_workload("C3", smg, mesh)
mg.precompute(A)
for sw in (1, 2, 4, 8, 16):
    print("sweeps per graph %2d:" % sw, " ".join("L%d %.2f us/launch" % (lv, mg.bench_relax(lv, 1, sw, 100) / (4 * sw)) for lv in range(mg.n_levels - 1)))
