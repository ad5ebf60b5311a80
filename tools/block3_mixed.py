#!/usr/bin/env python3
"""Block (3-DOF) benchmark system (tools/block3_time.py): the all-fp64 solve against the mixed-precision mode (fp32 V-cycle on the fp32 image of the
3 x 3-block panels, fp64 outer residual and update) -- ms per outer iteration, cycles and time to 1e-10."""
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
n = A.shape[0]
mg = smg.Hierarchy.from_prolongs(Ps)
mg.set_block_mode("block")
mg.precompute(A)
dev = torch.device("cuda")
rhs = torch.from_numpy(np.random.default_rng(100).uniform(-1.0, 1.0, n)).to(dev)
z0 = torch.zeros(n, dtype=torch.float64, device=dev); z = torch.empty_like(z0)
print(label, "-> block system of", n, "DOFs")
for prec in ("fp64", "mixed"):
    o = smg.SolveOpts(tol=1e-10, max_iter=100, precision=prec)
    mg.solve_device(rhs.data_ptr(), z0.data_ptr(), z.data_ptr(), n, 1, opts=o)
    torch.cuda.synchronize(); t0 = time.time()
    conv, rh = mg.solve_device(rhs.data_ptr(), z0.data_ptr(), z.data_ptr(), n, 1, opts=o)
    torch.cuda.synchronize(); dt = time.time() - t0
    res = float(torch.linalg.norm(torch.from_numpy(A @ z.cpu().numpy()) - rhs.cpu()))
    print("%-5s: converged %s in %d cycles, %.2f ms (%.3f ms per outer iteration), true residual %.2e" % (prec, conv, len(rh) - 1, 1e3 * dt, 1e3 * dt / max(1, len(rh) - 1), res))
