#!/usr/bin/env python3
"""relax(sweeps) of one level of a decimated hierarchy, graph-replayed (smg_bench_relax).  usage: tools/wgs_probe.py workload [sweeps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
import surface_multigrid_code_amd as smg
from surface_multigrid_code_amd import mesh
wl = sys.argv[1] if len(sys.argv) > 1 else "C3pdec"
sw = int(sys.argv[2]) if len(sys.argv) > 2 else 2
k = int(os.environ.get("SMG_TOOL_K", "1"))
mg, A, Mb, Vf, Ff, label, _ = B.build_workload(wl, smg, mesh)
mg.precompute(A)
for lv in range(mg.n_levels - 1):
    us = mg.bench_relax(lv, k, sw, 200)
    info = mg.wave_gs_order(lv, k)
    print("level %d rows %7d: relax(%d) %7.1f us%s" % (lv, mg.rows(lv), sw, us, "" if info is None else "  (wave GS: %d piece colours -> %.2f us per launch; phases mean %.1f max %d)" % (len(info["color_ptr"]) - 1, us / sw / (len(info["color_ptr"]) - 1), info["phases_mean"], info["phases_max"])))
