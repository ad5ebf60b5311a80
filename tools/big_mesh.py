#!/usr/bin/env python3
"""A mesh four times the size of BASELINE's largest (16.8 M vertices, 117 M stored entries, 7 levels): precompute, fine-level SpMV, the
reference's V(2,2) Gauss-Seidel cycle and a solve, with the bytes they stream.  usage: tools/big_mesh.py [workload = C6]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.zeros(1, device="cuda")
import bench as B
import surface_multigrid_code_amd as smg
from surface_multigrid_code_amd import mesh
wl = sys.argv[1] if len(sys.argv) > 1 else "C6"
t0 = time.time()
mg, A, Mb, Vf, Ff, label, t_host = B.build_workload(wl, smg, mesh)
print(label, "| host mesh + hierarchy %.1f s" % (time.time() - t0), flush=True)
t0 = time.time(); mg.precompute(A); t_pre = time.time() - t0
print("precompute %.2f s, levels %s" % (t_pre, [mg.rows(l) for l in range(mg.n_levels)]), flush=True)
n = A.shape[0]
dev = torch.device("cuda", torch.cuda.current_device())
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
mg.set_stream(stream.cuda_stream)
x = torch.from_numpy(np.random.default_rng(1).uniform(-1, 1, n)).to(dev)
y = torch.empty_like(x)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(10): mg.raw_spmv(0, 0, x.data_ptr(), None, y.data_ptr())
torch.cuda.synchronize(); e0.record(stream)
for _ in range(200): mg.raw_spmv(0, 0, x.data_ptr(), None, y.data_ptr())
e1.record(stream); torch.cuda.synchronize()
us = 1e3 * e0.elapsed_time(e1) / 200
by = mg.spmv_bytes(0, 1)
print("fine-level y = A x: %.1f us, %d B => %.2f TB/s (%.1f %% of 8 TB/s)" % (us, by, by / us / 1e6, by / us / 1e6 / 8 * 100), flush=True)
cyc = mg.bench_vcycle(0, 1, 2, 2, 50)
cb = mg.vcycle_bytes(1, 2, 2)
print("V(2,2) Gauss-Seidel cycle: %.1f us, %d B => %.2f TB/s" % (cyc, cb, cb / cyc / 1e6), flush=True)
rng = np.random.default_rng(5)
rhs = Mb @ rng.uniform(-1, 1, n)
t0 = time.time()
conv, z, rh = mg.solve(rhs[:, None], np.zeros((n, 1)), None, smg.SolveOpts(tol=1e-10, max_iter=100))
print("solve to 1e-10: converged %s in %d cycles, %.1f ms wall (incl. transfers), residuals %.2e -> %.2e" % (conv, len(rh) - 1, 1e3 * (time.time() - t0), rh[0], rh[-1]))
print("true residual |rhs - A z| = %.2e" % np.linalg.norm(rhs - A @ z[:, 0]))
