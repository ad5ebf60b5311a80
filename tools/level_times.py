#!/usr/bin/env python3
"""Where does a V-cycle's time go, level by level (graph replay, hipEvents).  usage: tools/level_times.py [workload] [k]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
import surface_multigrid_code_amd as smg
from surface_multigrid_code_amd import mesh
wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 1
mg, A, Mb, Vf, Ff, label, _ = B.build_workload(wl, smg, mesh)
mg.precompute(A)
sm = os.environ.get("SMG_TOOL_SMOOTHER", "gs")      # gs | jacobi | hybrid[:max_rows[:omega]]
parts = sm.split(":")
mg.set_smoother(parts[0], float(parts[2]) if len(parts) > 2 else 0.8, int(parts[1]) if len(parts) > 1 else 100000)   # gs | jacobi | hybrid | chebyshev | hybrid_chebyshev
SW = int(os.environ.get("SMG_TOOL_SWEEPS", "2"))      # pre = post sweeps of the cycle
print(label, "k =", k, "smoother", sm, "V(%d,%d)" % (SW, SW))
prev = None
ts = []
for lv in range(mg.n_levels):
    t = mg.bench_vcycle(lv, k, SW, SW, 100)
    ts.append(t)
for lv in range(mg.n_levels):
    own = ts[lv] - (ts[lv + 1] if lv + 1 < mg.n_levels else 0.0)
    ncol = len(mg.colors(lv)) - 1 if lv < mg.n_levels - 1 else 0
    print("level %d rows %8d colours %d: cycle from here %8.1f us, this level alone %7.1f us" % (lv, mg.rows(lv), ncol, ts[lv], own))
