#!/usr/bin/env python3
"""BASELINE config C5: synthetic 4M-vertex subdivided closed surface (torus 64x64, 5x mid-point subdivision -> 4 194 304
vertices, 6 levels), fp64 vs fp32(mixed): residual floors, V-cycles/s and fine-level SpMV GB/s.  Prints one JSON object."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench as B
import surface_multigrid_code_amd as smg
from surface_multigrid_code_amd import mesh

wl = sys.argv[1] if len(sys.argv) > 1 else "C5"
dev = torch.device("cuda", 0)
t0 = time.time()
mg, A, Mb, Vf, Ff, label, t_host = B.build_workload(wl, smg, mesh)
mg.precompute(A)
t_setup = time.time() - t0
n = A.shape[0]
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st); mg.set_stream(st.cuda_stream)
rng = np.random.default_rng(5)
rhs_h = Mb @ rng.uniform(-1, 1, n)
rhs = torch.from_numpy(rhs_h).to(dev); z0 = torch.zeros(n, dtype=torch.float64, device=dev); z = torch.empty_like(z0)
out = {"workload": label, "n_verts": n, "nnz": int(A.nnz), "levels": mg.n_levels, "level_rows": [mg.rows(l) for l in range(mg.n_levels)],
       "colors": [len(mg.colors(l)) - 1 for l in range(mg.n_levels - 1)], "setup_s": t_setup}
sm = dict(smoother=os.environ.get("SMG_TOOL_SMOOTHER", "gs"), jacobi_max_rows=300000)
out["smoother"] = sm["smoother"]
for prec in ("f64", "mixed"):
    # attainable residual: run 40 cycles with tol 0
    mg.solve_begin(rhs.data_ptr(), n, z0.data_ptr(), n, 1, opts=smg.SolveOpts(tol=0.0, max_iter=40, precision=prec, **sm))
    mg.outer_iterations(40)
    conv, rh = mg.solve_end(z.data_ptr(), n, max_iter=40)
    true_res = float(np.linalg.norm(rhs_h - A @ z.cpu().numpy()))
    # throughput
    K = 300
    mg.solve_begin(rhs.data_ptr(), n, z0.data_ptr(), n, 1, opts=smg.SolveOpts(tol=0.0, max_iter=K + 20, precision=prec, **sm))
    mg.outer_iterations(20); torch.cuda.synchronize()
    t = time.perf_counter(); mg.outer_iterations(K); torch.cuda.synchronize(); dt = time.perf_counter() - t
    mg.solve_end(z.data_ptr(), n, max_iter=K + 20)
    out[prec] = {"r_his_first": float(rh[0]), "residual_floor": float(rh.min()), "true_residual_after_40_cycles": true_res,
                 "cycles_to_1e-10_rel": int(np.argmax(rh < 1e-10 * rh[0])) if (rh < 1e-10 * rh[0]).any() else None,
                 "vcycles_per_s": K / dt, "ms_per_cycle": 1e3 * dt / K}
# fine-level SpMV, both precisions
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
x64 = torch.from_numpy(rng.uniform(-1, 1, n)).to(dev); y64 = torch.empty_like(x64)
x32 = x64.float(); y32 = torch.empty_like(x32)
R = 300
for _ in range(20): mg.raw_spmv(0, 0, x64.data_ptr(), None, y64.data_ptr())
torch.cuda.synchronize(); e0.record(st)
for _ in range(R): mg.raw_spmv(0, 0, x64.data_ptr(), None, y64.data_ptr())
e1.record(st); torch.cuda.synchronize(); us64 = 1e3 * e0.elapsed_time(e1) / R
for _ in range(20): mg.raw_spmv_f32(0, x32.data_ptr(), y32.data_ptr())
torch.cuda.synchronize(); e0.record(st)
for _ in range(R): mg.raw_spmv_f32(0, x32.data_ptr(), y32.data_ptr())
e1.record(st); torch.cuda.synchronize(); us32 = 1e3 * e0.elapsed_time(e1) / R
b64 = 12 * A.nnz + 4 * (n + 1) + 16 * n
b32 = 8 * A.nnz + 4 * (n + 1) + 8 * n
out["spmv_f64"] = {"us": us64, "bytes": int(b64), "GBps": b64 / us64 / 1e3, "frac_of_8TBps": b64 / us64 / 1e3 / 8000}
out["spmv_f32"] = {"us": us32, "bytes": int(b32), "GBps": b32 / us32 / 1e3, "frac_of_8TBps": b32 / us32 / 1e3 / 8000,
                   "max_rel_diff_vs_f64": float((y32.double() - y64).abs().max() / y64.abs().max())}
print(json.dumps(out))
