#!/usr/bin/env python3
"""Wall-clock of smg_precompute (first call = full host + device setup; second = value-only path): tools/precompute_time.py [workload]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
import surface_multigrid_code_amd as smg
from surface_multigrid_code_amd import mesh
mg, A, Mb, Vf, Ff, label, t_setup = B.build_workload(sys.argv[1] if len(sys.argv) > 1 else "C3", smg, mesh)
print(label, "| hierarchy construction (mg_precompute_subdiv etc.) %.2f s" % t_setup)
for i in range(3):
    t0 = time.time(); mg.precompute(A); print("precompute call %d: %.3f s" % (i, time.time() - t0))
