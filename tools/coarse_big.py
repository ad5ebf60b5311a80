#!/usr/bin/env python3
"""Large coarsest levels: a 1-level call on bunny_15K (15 804 unknowns, one-ring operator), on a torus of 10 000 vertices and on bunny_15K subdivided once
(63 210 unknowns: beyond the dense range) -- dense inverse resp. sparse Cholesky (policy 'never') against the Schur-complement solver: first precompute, device memory, time of a cycle (= the direct solve) for 1 and 8 columns."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.zeros(1, device="cuda")
import surface_multigrid_code_amd as smg
from surface_multigrid_code_amd import mesh
L = smg._lib.load()
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import mesh_np as M
for name in ("bunny_15K_init.smgm", "torus100", "bunny_15K x1 subdivision"):
    if name == "torus100": V, F = mesh.torus(100, 100)
    elif name.endswith("subdivision"):
        V, F = mesh.read_triangle_mesh("bunny_15K_init.smgm"); V, F, _ = M.subdivision_hierarchy(V, F, 1); F = F.astype(np.int32)
    else: V, F = mesh.read_triangle_mesh(name)
    V = mesh.normalize_unit_area(V, F)
    A = (mesh.massmatrix(V, F, "barycentric") - 0.01 * mesh.cotmatrix(V, F)).tocsr(); A.sort_indices()
    n = A.shape[0]
    rng = np.random.default_rng(1)
    for when in ("never", "always"):
        b0 = L.smg_device_bytes_live()
        mg = smg.Hierarchy(1)
        mg.set_coarse_schur(when)
        os.environ["SMG_DEBUG_SCHUR"] = "1"
        t0 = time.time(); mg.precompute(A); t1 = time.time() - t0
        t0 = time.time(); mg.precompute(A); t2 = time.time() - t0
        A2 = A.copy(); A2.data = A.data * 1.01
        t0 = time.time(); mg.precompute(A2); t3 = time.time() - t0
        mem = (L.smg_device_bytes_live() - b0) / 1e6
        out = []
        for k in (1, 8):
            rhs = rng.uniform(-1, 1, (n, k)); z0 = np.zeros((n, k))
            r = mg.solve(rhs, z0, None, smg.SolveOpts(tol=1e-9, max_iter=5))
            res = np.linalg.norm(A2 @ r[1] - rhs) / np.linalg.norm(rhs)
            out.append("k=%d: %d cycles, residual %.1e, coarse solve %.1f us" % (k, len(r[2]) - 1, res, mg.bench_vcycle(0, k, 2, 2, 20)))
        print("%-18s n=%6d %-7s %s: precompute first %.1f ms, again %.1f ms, value-only %.1f ms; device memory %.0f MB; %s" % (
            name, n, when, mg.coarse_solver()["kind"], 1e3 * t1, 1e3 * t2, 1e3 * t3, mem, "; ".join(out)))
        del mg
