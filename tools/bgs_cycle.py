#!/usr/bin/env python3
"""C3 V(2,2) cycle + outer residual with k columns: multi-colour sweeps vs block sweeps on the levels of >= min_rows rows.
    python tools/bgs_cycle.py [min_rows = 500000] [k ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench as B
import surface_multigrid_code_amd as smg
from surface_multigrid_code_amd import mesh
min_rows = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
ks = [int(a) for a in sys.argv[2:]] or [64]
dev = torch.device("cuda", 0)
mg, A, Mb, Vf, Ff, label, _ = B.build_workload("C3", smg, mesh)
mg.precompute(A)
n = A.shape[0]
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st); mg.set_stream(st.cuda_stream)
for k in ks:
    rhs = torch.from_numpy(np.ascontiguousarray(np.stack([Mb @ np.random.default_rng(1000 + j).uniform(-1, 1, n) for j in range(k)], 0))).to(dev)
    z0 = torch.zeros_like(rhs); z = torch.empty_like(rhs)
    out = {}
    for name, mr in (("colours", -1), ("blocks", min_rows)):
        mg.set_block_gs(mr)
        conv, rh = mg.solve_device(rhs.data_ptr(), z0.data_ptr(), z.data_ptr(), n, k, opts=smg.SolveOpts(tol=1e-10, max_iter=60))
        mg.solve_begin(rhs.data_ptr(), n, z0.data_ptr(), n, k, opts=smg.SolveOpts(tol=0.0, max_iter=200))
        mg.outer_iterations(5)
        ts = []
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            torch.cuda.synchronize(); e0.record(st); mg.outer_iterations(10); e1.record(st); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10)
        mg.solve_end(z.data_ptr(), n, max_iter=200)
        out[name] = (float(np.median(ts)), len(rh) - 1, bool(conv), [mg.bench_relax(0, k, 1, 10), mg.bench_relax(1, k, 1, 10)])
    print("k = %3d: outer iteration %.3f ms / %d cycles to 1e-10 (colours) | %.3f ms / %d cycles (blocks on levels >= %d rows) | relax(1) level 0: %.1f vs %.1f us, level 1: %.1f vs %.1f us"
          % (k, out["colours"][0], out["colours"][1], out["blocks"][0], out["blocks"][1], min_rows, out["colours"][3][0], out["blocks"][3][0], out["colours"][3][1], out["blocks"][3][1]), flush=True)
    del rhs, z0, z
