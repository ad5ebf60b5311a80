#!/usr/bin/env python3
"""Block (3-DOF) path against the scalar path on the same system.
usage: tools/block3_time.py [workload = C3 | small | torus1m] [--json]

System: kron(S, C3) on the workload's mesh, S = M_bary + 0.01 (-L) (the benchmark's matrix), C3 a fixed SPD 3 x 3 coupling; hierarchy
P (x) I_3 of the workload's prolongations (what mg_precompute_block builds).  Reports per outer iteration: ms, launches per sweep
(colours), algorithmic bytes (76 B per 3 x 3 block against 108 B for nine scalar entries), achieved GB/s, and the cycles to 1e-10."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp

C3 = np.array([[4.0, 1.0, 0.5], [1.0, 3.0, -0.75], [0.5, -0.75, 5.0]])      # SPD (eigenvalues 2.3 .. 5.5)


def block_system(smg, mesh, workload):
    import bench as B
    mg0, S, Mb, Vf, Ff, label, _ = B.build_workload(workload, smg, mesh)
    Ps = [sp.kron(mg0.matrix(l, "P_full"), sp.identity(3, format="csr"), format="csr") for l in range(1, mg0.n_levels)]
    for P in Ps:
        P.sort_indices()
    A = sp.kron(S, sp.csr_matrix(C3), format="csr")
    A.sort_indices()
    return A, Ps, label


def measure(smg, torch, A, Ps, mode, smoother="gs", reps=200):
    dev = torch.device("cuda", torch.cuda.current_device())
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    mg = smg.Hierarchy.from_prolongs(Ps)
    mg.set_block_mode(mode)
    mg.set_stream(stream.cuda_stream)
    t0 = time.time()
    mg.precompute(A)
    t_pre = time.time() - t0
    n = A.shape[0]
    rng = np.random.default_rng(100)
    rhs = torch.from_numpy(rng.uniform(-1.0, 1.0, n)).to(dev)
    z0 = torch.zeros(n, dtype=torch.float64, device=dev)
    z = torch.empty(n, dtype=torch.float64, device=dev)
    kw = dict(smoother=smoother, jacobi_max_rows=3 * 300000)
    o = smg.SolveOpts(tol=1e-10, max_iter=100, **kw)
    mg.solve_device(rhs.data_ptr(), z0.data_ptr(), z.data_ptr(), n, 1, opts=o)
    conv, rh = mg.solve_device(rhs.data_ptr(), z0.data_ptr(), z.data_ptr(), n, 1, opts=o)
    oo = smg.SolveOpts(tol=0.0, max_iter=1024, **kw)
    mg.solve_begin(rhs.data_ptr(), n, z0.data_ptr(), n, 1, opts=oo)
    mg.outer_iterations(20)
    ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ea.record(stream)
    mg.outer_iterations(reps)
    eb.record(stream)
    torch.cuda.synchronize()
    mg.solve_end(z.data_ptr(), n, max_iter=1024)
    ms = ea.elapsed_time(eb) / reps
    # fine-level y = A x and one Gauss-Seidel sweep
    import bench as B
    x = torch.from_numpy(rng.uniform(-1.0, 1.0, n)).to(dev)
    y = torch.empty_like(x)
    spmv_us, spmv_all = B.median_us(torch, stream, lambda: mg.raw_spmv(0, 0, x.data_ptr(), None, y.data_ptr()), 60, 5)
    u = torch.zeros_like(x)
    mg.set_smoother("gs")
    gs_us, gs_all = B.median_us(torch, stream, lambda: mg.raw_relax(0, rhs.data_ptr(), u.data_ptr(), 1, 1), 30, 5, warm=5)
    byt = mg.vcycle_bytes(1, 2, 2)
    sb = mg.spmv_bytes(0, 1)
    out = {"mode": mode, "block_size": mg.block_size(), "smoother": smoother, "dofs": n, "levels": mg.n_levels,
           "colors": [len(mg.colors(l)) - 1 for l in range(mg.n_levels - 1)], "ms_per_iteration": ms, "cycles_to_1e-10": len(rh) - 1,
           "converged": bool(conv), "time_to_tol_ms": (len(rh) - 1) * ms, "bytes_per_iteration": int(byt), "gbs": byt / (ms * 1e-3) / 1e9,
           "frac_of_hbm_peak": byt / (ms * 1e-3) / 1e9 / 8000.0, "spmv_us": spmv_us, "spmv_bytes": int(sb), "spmv_gbs": sb / (spmv_us * 1e-6) / 1e9,
           "gs_sweep_us": gs_us, "gs_sweep_gbs": (sb + 8 * n) / (gs_us * 1e-6) / 1e9, "precompute_s": t_pre, "final_residual": float(rh[-1]),
           "spmv_us_repeats": spmv_all, "gs_sweep_us_repeats": gs_all, "timing": "median of 5 HIP-event-timed loops",
           # roofline of the fine-level y = A x of this path: k_bsr3<SELL_AX> on 3 x 3 blocks (76 B per block) / k_sell on scalar entries
           "roofline": {"kernel": "k_bsr3<SELL_AX> (fine-level y = A x, 3 x 3 blocks)" if mg.block_size() == 3 else "k_sell<SELL_AX,1>", "bound": "hbm",
                        "bytes_per_launch": int(sb), "us_per_launch": spmv_us, "achieved": sb / (spmv_us * 1e-6) / 1e9, "peak": 8000.0, "unit": "GB/s",
                        "frac": sb / (spmv_us * 1e-6) / 1e9 / 8000.0}}
    if mg.block_size() == 3:
        out["block_stats"] = mg.block_stats(0)
    # the mixed-precision mode on the same handle (fp32 V-cycle on the fp32 images, fp64 outer residual and update): ms per outer iteration, cycles to 1e-10
    try:
        om = smg.SolveOpts(tol=1e-10, max_iter=100, precision="mixed", **kw)
        mg.solve_device(rhs.data_ptr(), z0.data_ptr(), z.data_ptr(), n, 1, opts=om)
        convm, rhm = mg.solve_device(rhs.data_ptr(), z0.data_ptr(), z.data_ptr(), n, 1, opts=om)
        oom = smg.SolveOpts(tol=0.0, max_iter=1024, precision="mixed", **kw)
        mg.solve_begin(rhs.data_ptr(), n, z0.data_ptr(), n, 1, opts=oom)
        mg.outer_iterations(20)
        ea.record(stream)
        mg.outer_iterations(reps)
        eb.record(stream)
        torch.cuda.synchronize()
        mg.solve_end(z.data_ptr(), n, max_iter=1024)
        out["mixed_precision"] = {"ms_per_iteration": ea.elapsed_time(eb) / reps, "cycles_to_1e-10": len(rhm) - 1, "converged": bool(convm), "final_residual": float(rhm[-1])}
    except Exception as e:
        out["mixed_precision"] = {"error": repr(e)}
    del mg
    return out


def main():
    import torch
    import surface_multigrid_code_amd as smg
    from surface_multigrid_code_amd import mesh
    wl = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "C3"
    A, Ps, label = block_system(smg, mesh, wl)
    res = {"workload": label + " -> kron(S, C3), DOF = 3 v + d, P (x) I_3", "nnz": int(A.nnz)}
    runs = (("block", "gs"),) if "--block-only" in sys.argv else (("block", "gs"), ("scalar", "gs"), ("block", "hybrid_chebyshev"), ("scalar", "hybrid_chebyshev"))
    for mode, sm in runs:
        res["%s_%s" % (mode, sm)] = measure(smg, torch, A, Ps, mode, sm)
    if "--json" in sys.argv:
        print(json.dumps(res))
        return
    print(res["workload"], "nnz", res["nnz"])
    for key, v in res.items():
        if not isinstance(v, dict):
            continue
        print("%-26s colours %-22s %.4f ms/iteration  %3d cycles to 1e-10 (%.2f ms)  %.0f GB/s of algorithmic bytes (%.0f %% of peak)  SpMV %.1f us %.0f GB/s  GS sweep %.1f us  precompute %.2f s"
              % (key, v["colors"], v["ms_per_iteration"], v["cycles_to_1e-10"], v["time_to_tol_ms"], v["gbs"], 100 * v["frac_of_hbm_peak"], v["spmv_us"], v["spmv_gbs"], v["gs_sweep_us"], v["precompute_s"]))


if __name__ == "__main__":
    main()
