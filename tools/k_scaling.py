#!/usr/bin/env python3
"""V(2,2) cycle time against the number of right-hand-side columns (graph replay, hipEvents).  usage: tools/k_scaling.py [workload]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
import surface_multigrid_code_amd as smg
from surface_multigrid_code_amd import mesh
wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
mg, A, Mb, Vf, Ff, label, _ = B.build_workload(wl, smg, mesh)
mg.precompute(A)
print(label)
for k in [int(x) for x in os.environ.get("SMG_TOOL_KS", "1,2,3,4,5,6,7,8,12,16,32,64").split(",")]:
    us = mg.bench_vcycle(0, k, 2, 2, 50 if k <= 8 else 20)
    print("k = %2d: %8.1f us per V(2,2) cycle, %7.1f us per column" % (k, us, us / k), flush=True)
