#!/usr/bin/env python3
"""rocprofv3 target: the coarse solve of C3's coarsest level alone (k = 1, 8, 64) and three value-only re-precomputes, Schur-complement solver."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.zeros(1, device="cuda")
import bench as B
import surface_multigrid_code_amd as smg
from surface_multigrid_code_amd import mesh
mg, A, Mb, Vf, Ff, label, _ = B.build_workload("C3", smg, mesh)
mg.precompute(A); mg.precompute(A)
d = torch.from_numpy(A.data).cuda()
for i in range(3):
    mg.precompute_values_device(d.data_ptr())
n = A.shape[0]
for k in (1, 8, 64):
    rhs = np.asfortranarray(Mb @ np.random.default_rng(0).uniform(-1, 1, (n, k)))
    mg.solve(rhs, np.zeros((n, k), order="F"), None, smg.SolveOpts(tol=1e-10, max_iter=2))
    print(k, mg.bench_vcycle(mg.n_levels - 1, k, 2, 2, 100))
