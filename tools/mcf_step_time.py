#!/usr/bin/env python3
"""End-to-end time of one mean-curvature-flow step (05_example_mean_curvature_flow/main.cpp:57-84) with everything in HBM:
assemble M(U) on the device, LHS = M - delta L, value-only re-precompute, 3-column solve, normalize_unit_area -- against the CPU
oracle doing the same step (host assembly, full precompute, solve).   usage: tools/mcf_step_time.py [workload] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
import bench as B
import surface_multigrid_code_amd as smg
from surface_multigrid_code_amd import mesh
from oracle.oracle import OracleMG
from oracle import mesh_np as M
wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
mg, A, Mb, V, F, label, _ = B.build_workload(wl, smg, mesh)
n, delta, tol = V.shape[0], 0.01, 5e-7                         # main.cpp:58-60
dev = torch.device("cuda", 0)
L = mesh.cotmatrix(V, F)
asm = mesh.Assembler(F, n)
st = torch.cuda.Stream(device=dev)
Fi = [torch.from_numpy(F[:, c]).long().to(dev) for c in range(3)]
diag_pos = torch.from_numpy(np.flatnonzero(L.indices == np.repeat(np.arange(n), np.diff(L.indptr)))).to(dev)
times, iters = [], []
seg = {}
def lap(name, t):
    st.synchronize(); seg.setdefault(name, []).append(time.time() - t); return time.time()

with torch.cuda.stream(st):
    mg.set_stream(st.cuda_stream)
    U = torch.from_numpy(V).to(dev)
    Lval = torch.from_numpy(L.data).to(dev)
    for step in range(steps + 2):
        st.synchronize(); t0 = time.time()
        t = t0
        _, mass = asm.assemble(U, 1.0, 0.0, "barycentric", stream=st.cuda_stream)
        if step >= 2: t = lap("assemble M(U)", t)
        lhs = -delta * Lval
        lhs[diag_pos] = mass + lhs[diag_pos]
        rhs = (mass[:, None] * U).T.contiguous()
        z0 = U.T.contiguous()
        if step >= 2: t = lap("LHS / RHS (torch)", t)
        if step == 0:
            st.synchronize()
            mg.precompute(sp.csr_matrix((lhs.cpu().numpy(), L.indices, L.indptr), shape=(n, n)))
        else:
            mg.precompute_values_device(lhs.data_ptr())
        if step >= 2: t = lap("value-only precompute", t)
        z = torch.empty_like(z0)
        mg.solve_begin(rhs.data_ptr(), n, z0.data_ptr(), n, 3, opts=smg.SolveOpts(tol=tol, max_iter=20, smoother=os.environ.get("SMG_TOOL_SMOOTHER", "gs"), jacobi_max_rows=300000))
        for _ in range(10):                          # enqueue two iterations, look at the device-side flag, stop when it is up
            mg.outer_iterations(2)
            if mg.poll()[0]:
                break
        conv, rh = mg.solve_end(z.data_ptr(), n)
        if step >= 2: t = lap("solve (3 columns)", t)
        U = z.T.contiguous()
        a, b, c = U[Fi[0]], U[Fi[1]], U[Fi[2]]
        U = U / torch.sqrt(torch.linalg.norm(torch.linalg.cross(b - a, c - a), dim=1).sum() / 2)
        U = torch.stack([U[:, 0] - U[:, 0].mean(), U[:, 1] - U[:, 1].mean(), U[:, 2] - U[:, 2].min()], dim=1).contiguous()
        st.synchronize()
        if step >= 2:
            lap("normalize_unit_area (torch)", t)
            times.append(time.time() - t0); iters.append(len(rh))
    mg.set_stream(None)
print(label)
print("  segments (ms, each followed by a stream sync):", {k: round(1e3 * float(np.median(v)), 2) for k, v in seg.items()})
print("GPU step (assemble + value-only precompute + 3-column solve to %g + normalise): median %.1f ms, V-cycles per step %s" % (tol, 1e3 * np.median(times), iters))
# CPU oracle: one step
Ps = [mg.matrix(l, "P_full") for l in range(1, mg.n_levels)]
o = OracleMG(Ps)
Uh = V.copy()
t0 = time.time()
Mh = M.massmatrix(Uh, F, "barycentric")
S = (Mh - delta * L).tocsr(); S.sort_indices()
t1 = time.time(); o.precompute(S); t2 = time.time()
conv, z, rh = o.solve(np.asfortranarray(Mh @ Uh), np.asfortranarray(Uh), tol=tol, max_iter=20)
t3 = time.time()
print("CPU oracle step: assemble %.2f s + precompute %.2f s + solve %.2f s (%d cycles) = %.2f s  => GPU step is %.0fx faster" % (t1 - t0, t2 - t1, t3 - t2, len(rh), t3 - t0, (t3 - t0) / np.median(times)))
