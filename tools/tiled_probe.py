#!/usr/bin/env python3
"""us per relax(2) call per level (graph replay), C3.  usage: tools/tiled_probe.py [columns] [sweeps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
import surface_multigrid_code_amd as smg
from surface_multigrid_code_amd import mesh
mg, A, Mb, Vf, Ff, label, _ = B.build_workload("C3", smg, mesh)
mg.precompute(A)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1
SW = int(sys.argv[2]) if len(sys.argv) > 2 else 2      # sweeps per relax()
print(" ".join("L%d %.2f" % (lv, mg.bench_relax(lv, K, SW, 200)) for lv in range(mg.n_levels - 1)), "| k", K, "sweeps", SW, "env", {k: v for k, v in os.environ.items() if k.startswith("SMG_")})
