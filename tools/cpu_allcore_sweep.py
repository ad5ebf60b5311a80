#!/usr/bin/env python3
"""Thread sweep of the all-core CPU comparator (oracle all-core mode) on this host: tools/cpu_allcore_sweep.py [workload]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
import bench as B
import surface_multigrid_code_amd as smg
from surface_multigrid_code_amd import mesh
from oracle.oracle import OracleMG
mg, A, Mb, Vf, Ff, label, _ = B.build_workload(sys.argv[1] if len(sys.argv) > 1 else "C3", smg, mesh)
mg.precompute(A)
rng = np.random.default_rng(3)
rhs = np.asfortranarray((Mb @ rng.uniform(-1, 1, A.shape[0]))[:, None])
L = mg.n_levels
perms = [mg.perm(l) for l in range(L)]
Ps = [sp.csr_matrix(mg.matrix(l, "P_full"))[perms[l - 1]][:, perms[l]].tocsc() for l in range(1, L)]
Ai = sp.csr_matrix(A)[perms[0]][:, perms[0]].tocsr()
orc = OracleMG(Ps); orc.precompute(Ai)
b = np.asfortranarray(rhs[perms[0]]); z0 = np.zeros_like(b)
t0 = time.time(); orc.solve(b, z0, tol=0.0, max_iter=4); print("sequential, colour-major numbering: %.1f ms/cycle" % (1e3 * (time.time() - t0) / 4))
for th in (4, 8, 16, 32, 64, 128):
    if th > (os.cpu_count() or 1): break
    orc.set_parallel([mg.colors(l) for l in range(L - 1)], th)
    orc.solve(b, z0, tol=0.0, max_iter=2)
    t0 = time.time(); orc.solve(b, z0, tol=0.0, max_iter=10); print("%3d threads: %.1f ms/cycle" % (th, 1e3 * (time.time() - t0) / 10))
