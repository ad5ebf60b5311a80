#!/usr/bin/env python3
"""The reference's own kind of hierarchy (mg_precompute: SSP decimation, Galerkin operators of 18 - 30 entries per row) on the GPU:
level table, per-level time of a graph-replayed V(2,2) cycle, cycles to 1e-10.   usage: tools/dec_probe.py [C3pdec|C3dec|ogre ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench as B
import surface_multigrid_code_amd as smg
from surface_multigrid_code_amd import mesh
k = int(os.environ.get("SMG_TOOL_K", "1"))
for wl in (sys.argv[1:] or ["C3pdec"]):
    mg, A, Mb, Vf, Ff, label, t_host = B.build_workload(wl, smg, mesh)
    torch.zeros(1, device="cuda"); torch.cuda.synchronize()
    t0 = time.time(); mg.precompute(A); t_pre = time.time() - t0
    stream = torch.cuda.Stream(); torch.cuda.set_stream(stream); mg.set_stream(stream.cuda_stream)      # events and libsmg's launches on ONE stream
    print(label)
    print("host mesh + hierarchy %.2f s (mg_precompute %.2f s), smg_precompute %.3f s" % (t_host, getattr(B.build_workload, "mg_precompute_s", float("nan")), t_pre))
    for lv in range(mg.n_levels):
        M = mg.matrix(lv, "A"); nn = np.diff(M.indptr)
        s = "level %d rows %8d entries/row %.1f (max %d)" % (lv, M.shape[0], nn.mean(), nn.max())
        if lv < mg.n_levels - 1:
            s += " colours %d" % (len(mg.colors(lv)) - 1)
        if lv > 0:
            PT = mg.matrix(lv, "PT"); pn = np.diff(PT.indptr)
            s += " | P %.2f per fine row, PT %.1f per coarse row (max %d)" % (mg.matrix(lv, "P").nnz / mg.rows(lv - 1), pn.mean(), pn.max())
        print(s)
    n = A.shape[0]
    byt = mg.vcycle_bytes(k, 2, 2)
    ts = [mg.bench_vcycle(lv, k, 2, 2, 100) for lv in range(mg.n_levels)]
    for lv in range(mg.n_levels):
        own = ts[lv] - (ts[lv + 1] if lv + 1 < mg.n_levels else 0.0)
        print("level %d: cycle from here %8.1f us, this level alone %7.1f us" % (lv, ts[lv], own))
    rhs = torch.from_numpy(Mb @ np.random.default_rng(100).uniform(-1, 1, (n, k))).cuda() if k > 1 else torch.from_numpy(Mb @ np.random.default_rng(100).uniform(-1, 1, n)).cuda()
    z0 = torch.zeros_like(rhs); z = torch.empty_like(rhs)
    if k == 1:
        o = smg.SolveOpts(tol=1e-10, max_iter=100)
        cv, rh = mg.solve_device(rhs.data_ptr(), z0.data_ptr(), z.data_ptr(), n, 1, opts=o)
        mg.solve_begin(rhs.data_ptr(), n, z0.data_ptr(), n, 1, opts=smg.SolveOpts(tol=0.0, max_iter=1024))
        mg.outer_iterations(50)
        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); ea.record(stream); mg.outer_iterations(300); eb.record(stream); torch.cuda.synchronize()
        mg.solve_end(z.data_ptr(), n, max_iter=1024)
        ms = ea.elapsed_time(eb) / 300
        print("outer iteration %.4f ms = %.1f V-cycles/s; bytes %d -> %.3f TB/s = %.3f of peak; cycles to 1e-10: %d (converged %s)"
              % (ms, 1e3 / ms, byt, byt / ms / 1e9, byt / ms / 1e9 / 8.0, len(rh) - 1, cv))
    print("device bytes live", smg._lib.load().smg_device_bytes_live())
    del mg
