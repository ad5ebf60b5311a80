#!/usr/bin/env python3
"""Time a full smg_precompute against the value-only (device) re-precompute on one workload."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
import bench as B
import surface_multigrid_code_amd as smg
from surface_multigrid_code_amd import mesh
wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
mg, A, Mb, Vf, Ff, label, _ = B.build_workload(wl, smg, mesh)
t = time.time(); mg.precompute(A); t_full = time.time() - t
A2 = (A + 0.1 * sp.diags(A.diagonal())).tocsr(); A2.sort_indices()
t = time.time(); mg.precompute(A2); t_first = time.time() - t     # builds the recipes
ts = []
for i in range(5):
    A3 = (A + (0.2 + 0.1 * i) * sp.diags(A.diagonal())).tocsr(); A3.sort_indices()
    ptr, col, val = A3.indptr.astype(np.int32), A3.indices.astype(np.int32), A3.data
    t = time.time(); mg.precompute(A3); ts.append(time.time() - t)
print(label)
print("full precompute %.3f s | first value-only (recipe build) %.3f s | value-only steady %.1f ms (min %.1f)" % (t_full, t_first, 1e3 * np.median(ts), 1e3 * min(ts)))
# values already in HBM: no H2D copy
import torch
dev = torch.device("cuda", 0)
vals = torch.from_numpy(A3.data).to(dev)
torch.cuda.synchronize()
ts = []
for i in range(5):
    t = time.time(); mg.precompute_values_device(vals.data_ptr()); ts.append(time.time() - t)
print("value-only from HBM (smg_precompute_values_device): %.1f ms (min %.1f)" % (1e3 * np.median(ts), 1e3 * min(ts)))
