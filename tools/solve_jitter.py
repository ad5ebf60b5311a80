import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench as B
import surface_multigrid_code_amd as smg
from surface_multigrid_code_amd import mesh
mg, A, Mb, Vf, Ff, label, _ = B.build_workload(sys.argv[1], smg, mesh)
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
mg.precompute(A)
n = A.shape[0]
rhs = np.asfortranarray((Mb @ np.random.default_rng(3).uniform(-1, 1, n))[:, None]); z0 = np.zeros_like(rhs)
o = smg.SolveOpts(tol=1e-10, max_iter=40)
ts = []
for i in range(40):
    t = time.perf_counter(); mg.solve(rhs, z0, None, o); ts.append(1e3 * (time.perf_counter() - t))
print(sys.argv[1], "solves (ms):", " ".join("%.1f" % x for x in ts))
