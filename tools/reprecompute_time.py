#!/usr/bin/env python3
"""The time-stepping callers' precompute (05_example_mean_curvature_flow/main.cpp:74: a new matrix with the same sparsity every step), C3:
first call (full), second (the value-only path builds its recipes once), then the steady state; host arrays in, and values already in HBM."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.zeros(1, device="cuda")
import bench as B
import surface_multigrid_code_amd as smg
from surface_multigrid_code_amd import mesh
mg, A, Mb, Vf, Ff, label, _ = B.build_workload("C3", smg, mesh)
print(label)
for i in range(6):
    A2 = A.copy(); A2.data = A.data * (1.0 + 0.01 * i)
    t0 = time.time(); mg.precompute(A2); torch.cuda.synchronize(); print("precompute call %d: %.2f ms" % (i + 1, 1e3 * (time.time() - t0)))
d = torch.from_numpy(A.data).cuda()
for i in range(4):
    torch.cuda.synchronize(); t0 = time.time(); mg.precompute_values_device(d.data_ptr()); torch.cuda.synchronize()
    print("values already on the device, call %d: %.2f ms" % (i + 1, 1e3 * (time.time() - t0)))
