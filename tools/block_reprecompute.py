#!/usr/bin/env python3
"""Value-only re-precompute of the block (3-DOF) benchmark system (tools/block3_time.py: kron(S, C3) on C3's mesh, 3 033 990 DOFs, coarsest level
11 856 unknowns) -- what the 06 caller pays ten times per time step (implicit_euler_mg_balloon.h:48-76) -- with the dense inverse and with the
Schur-complement coarse solver, and the outer iteration with each.   usage: tools/block_reprecompute.py [workload = C3]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.zeros(1, device="cuda")
import surface_multigrid_code_amd as smg
from surface_multigrid_code_amd import mesh
import block3_time as B3
wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
A, Ps, label = B3.block_system(smg, mesh, wl)
print(label, "-> block system of", A.shape[0], "DOFs")
n = A.shape[0]
rhs = np.random.default_rng(100).uniform(-1.0, 1.0, n)
for when in ("never", "always"):
    mg = smg.Hierarchy.from_prolongs(Ps)
    mg.set_block_mode("block")
    mg.set_coarse_schur(when)
    mg.precompute(A); mg.precompute(A)
    d = torch.from_numpy(A.data).cuda()
    mg.precompute_values_device(d.data_ptr())
    ts = []
    for i in range(5):
        torch.cuda.synchronize(); t0 = time.time(); mg.precompute_values_device(d.data_ptr()); torch.cuda.synchronize(); ts.append(1e3 * (time.time() - t0))
    conv, z, rh = mg.solve(rhs, np.zeros(n), None, smg.SolveOpts(tol=1e-10, max_iter=100))
    print("%-7s coarse solver %s: value-only re-precompute median %.2f ms (%s); solve: %s in %d cycles, V-cycle %.1f us, coarse solve %.1f us" % (
        when, mg.coarse_solver(), sorted(ts)[2], " ".join("%.1f" % t for t in ts), conv, len(rh) - 1, mg.bench_vcycle(0, 1, 2, 2, 30), mg.bench_vcycle(mg.n_levels - 1, 1, 2, 2, 100)))
