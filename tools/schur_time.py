#!/usr/bin/env python3
"""Coarse solver of C3 (3 952 unknowns): dense inverse against the Schur-complement solver (csrc/smg_schur.hpp) -- value-only re-precompute, the coarse
solve alone for k = 1 / 8 / 64 columns (HIP events around smg_level_coarse_solve-style piece calls are host-bound: the V-cycle's time is the measure), the
outer iteration, and the first solve's cycle count."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.zeros(1, device="cuda")
import bench as B
import surface_multigrid_code_amd as smg
from surface_multigrid_code_amd import mesh
wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
for when, name in ((0, "dense inverse"), (1, "schur complement")):
    os.environ["SMG_COARSE_SCHUR"] = str(when)
    os.environ["SMG_COARSE_SCHUR_MIN"] = "2048" if wl == "C3" else "1"
    mg, A, Mb, Vf, Ff, label, _ = B.build_workload(wl, smg, mesh)
    mg.precompute(A); mg.precompute(A)
    print("==", name, "|", label, "| coarse solver:", mg.coarse_solver())
    d = torch.from_numpy(A.data).cuda()
    mg.precompute_values_device(d.data_ptr())
    ts = []
    for i in range(7):
        torch.cuda.synchronize(); t0 = time.time(); mg.precompute_values_device(d.data_ptr()); torch.cuda.synchronize(); ts.append(1e3 * (time.time() - t0))
    print("   value-only re-precompute (values in HBM): median %.3f ms  (%s)" % (sorted(ts)[3], " ".join("%.2f" % t for t in ts)))
    n = A.shape[0]
    rng = np.random.default_rng(0)
    Lc = mg.n_levels - 1
    for k in (1, 3, 8, 64):
        rhs = np.asfortranarray(Mb @ rng.uniform(-1, 1, (n, k)))
        z0 = np.zeros((n, k), order="F")
        conv, z, rh = mg.solve(rhs, z0, None, smg.SolveOpts(tol=1e-10, max_iter=30))
        print("   k = %2d: converged %s in %d cycles, final residual %.3e; V-cycle %.1f us, of which the coarse solve %.1f us" % (
            k, conv, len(rh) - 1, rh[-1], mg.bench_vcycle(0, k, 2, 2, 50), mg.bench_vcycle(Lc, k, 2, 2, 200)))
