#!/usr/bin/env python3
"""C3's mesh with the hierarchy stopped one level earlier (mg_precompute's nVCoarsest is the caller's: a coarsest level of 15 804 unknowns on the
Schur-complement solver costs 38 us per solve) against the benchmark's 5 levels: precompute, colours, V-cycle time, cycles to 1e-10."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.zeros(1, device="cuda")
import surface_multigrid_code_amd as smg
from surface_multigrid_code_amd import mesh
V, F = mesh.read_triangle_mesh("bunny_15K_init.smgm")
V = mesh.normalize_unit_area(V, F)
for extra in (1, 0):
    mg, Vf, Ff = smg.mg_precompute_subdiv(V, F, 3, ratio=0.25, nVCoarsest=1000, n_extra_levels=extra)
    Vf = mesh.normalize_unit_area(Vf, Ff)
    Mb = mesh.massmatrix(Vf, Ff, "barycentric")
    A = (Mb - 0.01 * mesh.cotmatrix(Vf, Ff)).tocsr(); A.sort_indices()
    t0 = time.time(); mg.precompute(A); tp = time.time() - t0
    n = A.shape[0]
    for k in (1, 3):
        rhs = np.asfortranarray(Mb @ np.random.default_rng(0).uniform(-1, 1, (n, k)))
        conv, z, rh = mg.solve(rhs, np.zeros((n, k), order="F"), None, smg.SolveOpts(tol=1e-10, max_iter=30))
        vc = mg.bench_vcycle(0, k, 2, 2, 50)
        print("%d levels (coarsest %d unknowns, %s; colours %s), k = %d: precompute %.0f ms, V-cycle %.1f us, %d cycles to 1e-10 -> %.2f ms of cycles" % (
            mg.n_levels, mg.rows(mg.n_levels - 1), mg.coarse_solver()["kind"], [len(mg.colors(l)) - 1 for l in range(mg.n_levels - 1)], k, 1e3 * tp, vc,
            len(rh) - 1, 1e-3 * vc * (len(rh) - 1)))
