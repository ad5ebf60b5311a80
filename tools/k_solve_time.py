#!/usr/bin/env python3
"""ms per outer iteration of smg_solve (graph-replayed) against the number of right-hand-side columns, and a check that the caller's columns do
not depend on the internal column padding (SMG_PAD_COLS, csrc/smg_cycle.cpp: internal_cols).   usage: tools/k_solve_time.py [workload] [k ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench as B
import surface_multigrid_code_amd as smg
from surface_multigrid_code_amd import mesh
wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
ks = [int(x) for x in sys.argv[2:]] or [1, 3, 4, 5, 6, 7, 8, 12, 13, 16, 24, 33, 48, 64]
mg, A, Mb, Vf, Ff, label, _ = B.build_workload(wl, smg, mesh)
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
mg.precompute(A)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream); mg.set_stream(stream.cuda_stream)
n = A.shape[0]
print(label, "SMG_PAD_COLS =", os.environ.get("SMG_PAD_COLS", "1 (default)"))
rng = np.random.default_rng(5)
G = Mb @ rng.uniform(-1, 1, (n, max(ks)))
for k in ks:
    rhs = torch.from_numpy(np.asfortranarray(G[:, :k])).cuda().t().contiguous().t() if k > 1 else torch.from_numpy(G[:, 0].copy()).cuda()
    rhs = torch.from_numpy(np.ascontiguousarray(G[:, :k].T)).cuda()          # k x n row-major == n x k column-major, ld = n
    z0 = torch.zeros_like(rhs); z = torch.empty_like(rhs)
    ms = B.steady_ms(torch, stream, mg, rhs, z0, z, n, k, dict(smoother="gs"), warm=10, iters=40, repeats=3)
    cv, rh = mg.solve_device(rhs.data_ptr(), z0.data_ptr(), z.data_ptr(), n, k, opts=smg.SolveOpts(tol=1e-10, max_iter=100))
    h = int(np.frombuffer(z.cpu().numpy().tobytes(), dtype=np.uint64).sum(dtype=np.uint64))
    print("k = %2d: %8.3f ms per outer iteration, %7.1f us per column; cycles %d, z checksum %016x" % (k, ms, 1e3 * ms / k, len(rh) - 1, h), flush=True)
