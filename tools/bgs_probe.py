#!/usr/bin/env python3
"""relax(1) at k = 64 (or argv[2]) on the levels of C3: block-sequential sweep vs the multi-colour launches, us per sweep and the plan's statistics.
    python tools/bgs_probe.py [C3] [64]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
import surface_multigrid_code_amd as smg
from surface_multigrid_code_amd import mesh
wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 64
mg, A, Mb, Vf, Ff, label, _ = B.build_workload(wl, smg, mesh)
mg.precompute(A)
for lv in range(min(3, mg.n_levels - 1)):
    n = mg.rows(lv)
    mg.set_block_gs(-1)
    t_col = mg.bench_relax(lv, k, 1, 30)
    mg.set_block_gs(0)
    t_blk = mg.bench_relax(lv, k, 1, 30)
    info = mg.block_gs_order(lv, k)
    nnz = mg.matrix(lv, "A").nnz
    alg = 12 * nnz + 4 * (n + 1) + 24 * n * k
    print("level %d: %7d rows  colours %7.1f us  blocks %7.1f us  (%d blocks, %d block colours, rim %.3f, fill %.3f)  algorithmic %.0f MB -> %.2f / %.2f TB/s"
          % (lv, n, t_col, t_blk, len(info["blk_ptr"]) - 1, len(info["color_ptr"]) - 1, info["rim"], info["fill"], alg / 1e6, alg / t_col / 1e6, alg / t_blk / 1e6), flush=True)
