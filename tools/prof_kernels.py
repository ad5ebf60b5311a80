#!/usr/bin/env python3
"""Run the fine-level kernels of one workload a fixed number of times (for rocprofv3 --pmc / --kernel-trace passes).
    python tools/prof_kernels.py [--workload C3] [--reps 30]
"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench as B
import surface_multigrid_code_amd as smg
from surface_multigrid_code_amd import mesh

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="C3")
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--cycles", type=int, default=10)
ap.add_argument("--smoother", default="gs")
ap.add_argument("--jacobi-max-rows", type=int, default=300000)
ap.add_argument("--k", type=int, default=0, help="right-hand-side columns of the cycle part (default: 1, C4k64: 64)")
a = ap.parse_args()
dev = torch.device("cuda", 0)
if a.workload == "C4k64":      # BASELINE config C4: ogre.obj, mean-curvature-flow system, 64 columns (k_sell_wide)
    V, F = mesh.read_triangle_mesh("ogre.smgm"); V = mesh.normalize_unit_area(V, F)
    mg = smg.mg_precompute(V, F, 0.25, 500, 1)
    Mb = mesh.massmatrix(V, F, "barycentric"); A = (Mb - 0.01 * mesh.cotmatrix(V, F)).tocsr(); A.sort_indices()
    a.k = a.k if a.k > 0 else 64
    label = "C4: ogre.obj k = %d" % a.k
else:
    mg, A, Mb, Vf, Ff, label, _ = B.build_workload(a.workload, smg, mesh)
mg.precompute(A)
n = A.shape[0]
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st); mg.set_stream(st.cuda_stream)
rng = np.random.default_rng(3)
x = torch.from_numpy(rng.uniform(-1, 1, n)).to(dev); y = torch.empty_like(x)
b = torch.from_numpy(Mb @ rng.uniform(-1, 1, n)).to(dev); u = torch.zeros_like(x)
for _ in range(a.reps):
    mg.raw_spmv(0, 0, x.data_ptr(), None, y.data_ptr())
torch.cuda.synchronize()
for _ in range(a.reps):
    mg.raw_relax(0, b.data_ptr(), u.data_ptr(), 1, 1)
torch.cuda.synchronize()
k = a.k if a.k > 0 else 1
bk = torch.from_numpy(np.ascontiguousarray((Mb @ rng.uniform(-1, 1, (n, k))).T)).to(dev); uk = torch.zeros_like(bk); z = torch.empty_like(bk)
mg.solve_begin(bk.data_ptr(), n, uk.data_ptr(), n, k, opts=smg.SolveOpts(tol=0.0, max_iter=a.cycles, smoother=a.smoother, jacobi_max_rows=a.jacobi_max_rows))
mg.outer_iterations(a.cycles)
mg.solve_end(z.data_ptr(), n, max_iter=a.cycles)
print("done", label)
