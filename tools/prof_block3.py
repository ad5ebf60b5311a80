#!/usr/bin/env python3
"""Run the block (3-DOF) kernels of the C3 x 3 system a fixed number of times (for rocprofv3 --pmc / --kernel-trace passes): python tools/prof_block3.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
import surface_multigrid_code_amd as smg
from surface_multigrid_code_amd import mesh
import block3_time as B3
dev = torch.device("cuda", 0)
A, Ps, label = B3.block_system(smg, mesh, sys.argv[1] if len(sys.argv) > 1 else "C3")
mg = smg.Hierarchy.from_prolongs(Ps)
mg.set_block_mode("block")
mg.precompute(A)
n = A.shape[0]
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st); mg.set_stream(st.cuda_stream)
rng = np.random.default_rng(3)
x = torch.from_numpy(rng.uniform(-1, 1, n)).to(dev); y = torch.empty_like(x)
b = torch.from_numpy(rng.uniform(-1, 1, n)).to(dev); u = torch.zeros_like(x); z = torch.empty_like(x)
for _ in range(20):
    mg.raw_spmv(0, 0, x.data_ptr(), None, y.data_ptr())
torch.cuda.synchronize()
for _ in range(20):
    mg.raw_relax(0, b.data_ptr(), u.data_ptr(), 1, 1)
torch.cuda.synchronize()
mg.solve_begin(b.data_ptr(), n, u.data_ptr(), n, 1, opts=smg.SolveOpts(tol=0.0, max_iter=4, smoother="gs"))
mg.outer_iterations(4)
mg.solve_end(z.data_ptr(), n, max_iter=4)
print("done", label, n)
