#!/usr/bin/env python3
"""Anatomy of a wave Gauss-Seidel launch: relax(2) of every Galerkin level, graph-replayed, with the phase loop cut after N phases (SMG_DEBUG_WGS_PHASES: wrong
results, timing only).  usage: tools/wgs_anatomy.py [workload]   (run once per library: SMG_LIB selects an alternate build)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("SMG_ANATOMY_CHILD") != "1":
    wl = sys.argv[1] if len(sys.argv) > 1 else "C3pdec"
    for ph in ("0", "1", "2", "4", "6", "100"):
        out = subprocess.run([sys.executable, __file__, wl], env=dict(os.environ, SMG_ANATOMY_CHILD="1", SMG_DEBUG_WGS_PHASES=ph), capture_output=True, text=True)
        print("phases <= %3s: %s" % (ph, out.stdout.strip() or out.stderr[-400:]))
    sys.exit(0)
sys.path.insert(0, ROOT)
import bench as B
import surface_multigrid_code_amd as smg
from surface_multigrid_code_amd import mesh
mg, A, Mb, Vf, Ff, label, _ = B.build_workload(sys.argv[1], smg, mesh)
mg.precompute(A)
res = []
for lv in range(1, mg.n_levels - 1):
    us = min(mg.bench_relax(lv, 1, 2, 200) for _ in range(3))
    res.append("L%d %6.2f us/launch" % (lv, us / 8))
print("  ".join(res))
