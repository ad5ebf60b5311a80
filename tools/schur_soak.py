#!/usr/bin/env python3
"""Soak of the Schur-complement coarse solver: the same values factored again and again (value-only re-precompute on the device) and the coarse solve
repeated -- every result must equal the first one bit for bit (fixed summation orders; a race in the block kernel's LDS re-use or between its launches
would show as a differing bit sooner or later).  Scalar (bunny_15K two-level, 3 952 coarse unknowns) and 3-DOF (ogre_sim, blocks touching > 96 separator
rows) systems, 1 / 3 / 16 / 64 columns."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, scipy.sparse as sp
import surface_multigrid_code_amd as smg
from oracle import mesh_np as M
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
bad = 0
def soak(name, mg, A, ks):
    global bad
    rng = np.random.default_rng(3)
    mg.precompute(A); mg.precompute(A)
    nc = mg.rows(mg.n_levels - 1)
    assert mg.coarse_solver()["kind"] == "schur_complement", mg.coarse_solver()
    ref = {}
    for k in ks:
        B, u = rng.uniform(-1, 1, (nc, k)), rng.uniform(-1, 1, (nc, k))
        ref[k] = (B, u, mg.coarse_solve(B, u))
    for it in range(reps):
        mg.precompute(A)                       # value-only: the arena is factored again
        for k in ks:
            B, u, r0 = ref[k]
            r = mg.coarse_solve(B, u)
            if not np.array_equal(r, r0):
                bad += 1
                print("%s: repetition %d, k = %d: %d entries differ (max %.3e)" % (name, it, k, int((r != r0).sum()), float(abs(r - r0).max())))
    print("%s: %d re-factorisations x %s columns, coarse unknowns %d: %s" % (name, reps, ks, nc, "bit-identical" if bad == 0 else "DIFFERENCES"))
V, F = M.read_smgm("bunny_15K_init.smgm"); V = M.normalize_unit_area(V, F)
A = (M.massmatrix(V, F, "barycentric") - 0.01 * M.cotmatrix(V, F)).tocsr(); A.sort_indices()
mg = smg.mg_precompute(V, F, 0.25, 3000, 1); mg.set_coarse_schur("always", 1)
soak("scalar 15 804 -> 3 952", mg, A, (1, 3, 16, 64))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
V, F = M.read_smgm("ogre_sim.smgm"); V = M.normalize_unit_area(V, F)
S = (M.massmatrix(V, F, "barycentric") - 0.01 * M.cotmatrix(V, F)).tocsr()
C3 = np.array([[4.0, 1.0, 0.5], [1.0, 3.0, -0.75], [0.5, -0.75, 5.0]])
A3 = sp.kron(S, sp.csr_matrix(C3), format="csr"); A3.sort_indices()
mgb = smg.mg_precompute_block(V, F, 0.25, 600, 1); mgb.set_coarse_schur("always", 1)
soak("3-DOF ogre_sim", mgb, A3, (1, 3, 16))
print("failures:", bad)
