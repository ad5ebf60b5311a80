#!/usr/bin/env python3
"""bench.py -- V-cycles/s + fine-level SpMV GB/s of libsmg on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 100 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one outer iteration of min_quad_with_fixed_mg_solve (reference src/min_quad_with_fixed_mg.cpp:108-125):
the residual norm ||RHS - A_0 z||_2 followed by one V(2,2) cycle, on one right-hand-side column per GPU, with the
hierarchy, RHS and z already resident in HBM.  Workload at N=1 = BASELINE config C3: bunny_15K_init.obj, unit
area, 3x mid-point subdivision -> 1 011 330 vertices, 5 levels, system M_bary + 0.01 (-L) (SPD, no constraints).
N > 1: every rank owns one independent RHS column of the same mesh (weak scaling); the only communication is the
8-byte all-reduce of the residual sum of squares per iteration (RCCL), exactly SURVEY.md section 8e.

Timing: `--steps K` iterations are timed R = --repeats (default 9) times, each repeat bracketed by barrier + synchronize on both
sides and measured with HIP events on the solve's stream (max over ranks per repeat); `ms_per_step` / `value` are the MEDIAN repeat
(min / max alongside) -- a 20-step region is 6 ms, and one scheduling hiccup must not move the headline.
The timed cycle is the REFERENCE's: V(2,2) with Gauss-Seidel on every level (`--smoother gs`, the library default); libsmg's
Chebyshev-Jacobi hybrid is reported next to it in `smoothers` as an extension, with its own byte count.

Prints ONE compact JSON line (< 8 000 characters: `compact_line`) on rank 0 as the LAST line of stdout; the full record of every leg goes to
bench_extra.json.
"""
import hashlib
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
HBM_SUSTAINABLE_GBS = 6300.0   # what the part sustains on a streaming read (same guide, HBM section: "8 TB/s peak; ~6.3 TB/s achievable")
INFINITY_CACHE_BYTES = 256 * 2 ** 20


def _onto_torus(V, R=1.0, r=0.4):
    """Closest point on the torus (axis z, radii R, r): the subdivided vertices of the C5 / torus workloads are put back
    on the surface they sample ("mid-point subdivision re-projected onto the torus")."""
    rho = np.sqrt(V[:, 0] ** 2 + V[:, 1] ** 2)
    cx, cy = R * V[:, 0] / rho, R * V[:, 1] / rho
    d = V - np.stack([cx, cy, np.zeros_like(cx)], axis=1)
    d *= (r / np.linalg.norm(d, axis=1))[:, None]
    return np.stack([cx, cy, np.zeros_like(cx)], axis=1) + d


def build_workload(name, smg, mesh):
    """Returns (mg, A (scipy csr), Vf, Ff, label)."""
    t0 = time.time()
    if name == "C3":
        V, F = mesh.read_triangle_mesh("bunny_15K_init.smgm")
        V = mesh.normalize_unit_area(V, F)
        mg, Vf, Ff = smg.mg_precompute_subdiv(V, F, 3, ratio=0.25, nVCoarsest=1000, n_extra_levels=1)
        label = "C3: bunny_15K_init x3 midpoint subdivision, 1011330 verts, 5 levels, M_bary+0.01(-L), fp64"
    elif name == "torus1m":
        V, F = mesh.torus(64, 64)
        mg, Vf, Ff = smg.mg_precompute_subdiv(V, F, 4, n_extra_levels=0)
        Vf = mesh.normalize_unit_area(_onto_torus(Vf), Ff)
        label = "torus 64x64 x4 midpoint subdivision, 1048576 verts, 5 levels, M_bary+0.01(-L), fp64"
    elif name == "C5":
        V, F = mesh.torus(64, 64)
        mg, Vf, Ff = smg.mg_precompute_subdiv(V, F, 5, n_extra_levels=0)
        Vf = mesh.normalize_unit_area(_onto_torus(Vf), Ff)
        label = "C5: torus (R=1, r=0.4) 64x64 x5 midpoint subdivision re-projected onto the torus, 4194304 verts, 6 levels, M_bary+0.01(-L), fp64"
    elif name == "C6":       # (not a BASELINE config: four times C5, far beyond every cache -- tools/level_times.py C6)
        V, F = mesh.torus(64, 64)
        mg, Vf, Ff = smg.mg_precompute_subdiv(V, F, 6, n_extra_levels=0)
        Vf = mesh.normalize_unit_area(_onto_torus(Vf), Ff)
        label = "C6: torus (R=1, r=0.4) 64x64 x6 midpoint subdivision re-projected onto the torus, 16777216 verts, 7 levels, M_bary+0.01(-L), fp64"
    elif name in ("C3dec", "C3pdec"):
        # the C3 mesh (or its 252 834-vertex parent) under the REFERENCE's own hierarchy: mg_precompute(V, F, 0.25, 1000, 1)
        # (src/mg_precompute.cpp:15-87, the call of 03_mg_solver/main.cpp:35-39) -- mid-point decimation + self-parameterisation,
        # 3 entries per row of P, Galerkin operators of 18 - 30 entries per row; the subdivision operators are NOT used
        V, F = mesh.read_triangle_mesh("bunny_15K_init.smgm")
        V = mesh.normalize_unit_area(V, F)
        nsub = 3 if name == "C3dec" else 2
        mgs, Vf, Ff = smg.mg_precompute_subdiv(V, F, nsub, ratio=0.25, nVCoarsest=1000, n_extra_levels=0)
        del mgs
        t1 = time.time()
        mg = smg.mg_precompute(Vf, Ff, 0.25, 1000, 1)
        build_workload.mg_precompute_s = time.time() - t1
        label = "%s: bunny_15K_init x%d midpoint subdivision, %d verts, hierarchy by mg_precompute(V, F, 0.25, 1000, midpoint) (SSP decimation, %d levels), M_bary+0.01(-L), fp64" % (name, nsub, Vf.shape[0], mg.n_levels)
    elif name == "ogre":
        V, F = mesh.read_triangle_mesh("ogre.smgm")
        Vf, Ff = mesh.normalize_unit_area(V, F), F
        mg = smg.mg_precompute(Vf, Ff, 0.25, 500, 1)
        label = "ogre.obj (19985 verts), hierarchy by mg_precompute defaults (%d levels), M_bary+0.01(-L), fp64" % mg.n_levels
    elif name == "small":
        V, F = mesh.read_triangle_mesh("ogre_sim.smgm")
        V = mesh.normalize_unit_area(V, F)
        mg, Vf, Ff = smg.mg_precompute_subdiv(V, F, 2, n_extra_levels=0)
        label = "small: ogre_sim x2 midpoint subdivision, 40877 verts, 3 levels"
    else:
        raise SystemExit("unknown workload " + name)
    L = mesh.cotmatrix(Vf, Ff)
    Mb = mesh.massmatrix(Vf, Ff, "barycentric")
    A = (Mb - 0.01 * L).tocsr()       # 05_example_mean_curvature_flow/main.cpp:68
    A.sort_indices()
    return mg, A, Mb, Vf, Ff, label, time.time() - t0


def precompute_known_s(smg, mesh, mg, Vf, Ff, n_pins=346):
    """First smg_precompute WITH constraints on a fresh handle over the same prolongations: the Poisson system -L with 346 pinned vertices
    (BASELINE config C2's count on the C3 mesh; reference src/min_quad_with_fixed_mg.cpp:137-257: setdiff, slices, column-drop cascade,
    Galerkin products).  Seconds, HIP already up."""
    try:
        Ps = [mg.matrix(l, "P_full") for l in range(1, mg.n_levels)]
        m2 = smg.Hierarchy.from_prolongs(Ps)
        A = (-mesh.cotmatrix(Vf, Ff)).tocsr()
        A.sort_indices()
        known = np.sort(np.random.default_rng(0).choice(A.shape[0], n_pins, replace=False)).astype(np.int32)
        t0 = time.time()
        m2.precompute(A, known)
        dt = time.time() - t0
        del m2
        return dt
    except Exception as e:   # informational: never lose the bench line over it
        return {"error": repr(e)}


def reprecompute_leg(smg, mg, A, torch):
    """What the time-stepping callers pay at every step (05_example_mean_curvature_flow/main.cpp:74, 06: implicit_euler_mg_balloon.h:75): a value-only
    smg_precompute, new values already in HBM -- Galerkin recipes, panel refresh, coarse factorisation.  With the dense inverse of the coarsest matrix and
    with the Schur-complement solver the default policy moves to at the first such call (csrc/smg_schur.hpp); the V-cycle with each beside it.
    Runs last on the handle (the headline above was measured on the dense inverse a handle factored once keeps)."""
    d = torch.from_numpy(np.ascontiguousarray(A.data)).cuda()

    def med(reps=7):
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize(); t0 = time.perf_counter(); mg.precompute_values_device(d.data_ptr()); torch.cuda.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0))
        return float(np.median(ts))

    out = {"what": "median wall ms of smg_precompute_values_device (same pattern, values in HBM), 7 calls; V-cycle us = one graph-replayed V(2,2), 1 column"}
    for when, key in (("never", "dense_inverse"), ("refactor", "schur_complement")):
        mg.set_coarse_schur(when)
        mg.precompute(A)                       # full: the policy changed
        mg.precompute(A)                       # value-only: recipes built; under 'refactor' the coarse solver moves here
        mg.precompute_values_device(d.data_ptr())
        cs = mg.coarse_solver()
        out[key] = {"coarse_solver": cs["kind"], "factor_entries": cs["factor_entries"], "reprecompute_ms": med(), "vcycle_us": mg.bench_vcycle(0, 1, 2, 2, 50),
                    "coarse_solve_us": mg.bench_vcycle(mg.n_levels - 1, 1, 2, 2, 200)}
    out["coarse_unknowns"] = int(mg.rows(mg.n_levels - 1))
    out["default_policy"] = "dense inverse until the first value-only re-precompute, Schur complement from then on (smg_hierarchy_set_coarse_schur)"
    return out


def kernel_source_hash():
    """sha256 over the kernel sources: profiles/traffic.json carries the hash of the sources its PMC passes ran on
    (tools/make_traffic.py); a committed traffic figure is only reported while it still describes the kernels that are timed."""
    hsh = hashlib.sha256()
    for f in ("smg_device.hip", "smg_device.hpp", "smg_device_inl.hpp", "smg_gj_inl.hpp"):
        with open(os.path.join(ROOT, "surface_multigrid_code_amd", "csrc", f), "rb") as fh:
            hsh.update(fh.read())
    return hsh.hexdigest()[:16]


def committed_traffic(workload):
    """(bytes per launch or None, note)"""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except Exception:
        return None, "no profiles/traffic.json"
    ent = tj.get(workload, {})
    if not ent.get("hbm_bytes_per_launch"):
        return None, "no entry for %s in profiles/traffic.json" % workload
    if tj.get("kernel_source_sha") != kernel_source_hash():
        return None, "profiles/traffic.json was measured on other kernel sources (%s, now %s): stale, not reported -- rerun tools/profile_round.sh" % (tj.get("kernel_source_sha"), kernel_source_hash())
    return ent["hbm_bytes_per_launch"], "profiles/traffic.json: committed rocprofv3 --pmc passes of this kernel on these sources (%s), NOT a measurement of this run" % ent.get("source", "")


def cpu_baseline(mg, A, rhs, budget_s=15.0):
    """The oracle (CPU restatement of the reference algorithm, 1 thread) timed on this host on a bounded sample."""
    from oracle.oracle import OracleMG
    Ps = [mg.matrix(l, "P_full") for l in range(1, mg.n_levels)]
    orc = OracleMG(Ps)
    t0 = time.time()
    orc.precompute(A)
    t_pre = time.time() - t0
    z0 = np.zeros_like(rhs)
    # one cycle to size the sample
    t0 = time.time()
    orc.solve(rhs, z0, tol=0.0, max_iter=1)
    t1 = time.time() - t0
    m = int(max(2, min(200, budget_s / max(t1, 1e-6))))
    orc.profile_reset()
    t0 = time.time()
    _, _, rh = orc.solve(rhs, z0, tol=0.0, max_iter=m)
    dt = time.time() - t0
    prof = orc.profile()
    return {"value": m / dt, "unit": "V-cycles/s", "cores": 1, "kind": "port",
            "sample": "%d outer iterations (residual + V(2,2) cycle) of the same workload, oracle/smg_oracle.c, gcc -O3, 1 thread; "
                      "host has %d cores" % (m, os.cpu_count() or 0),
            "ms_per_cycle": 1e3 * dt / m, "precompute_s": t_pre,
            "relaxation_frac": prof["MG: relaxation"][1] / max(prof["MG: total VCycle"][1], 1e-12),
            "r_his_head": [float(x) for x in rh[:4]]}


def cpu_allcore(mg, A, rhs, budget_s=6.0):
    """The "fair CPU" comparator of SURVEY.md section 8d: the same cycle on ALL host cores (OpenMP).  Not the reference's path
    (that one is single-threaded: `cpu_baseline`): the system is renumbered colour-major -- the numbering and colouring the GPU
    path uses -- so that the reference's lexicographic sweep becomes a multi-colour sweep whose colour blocks are swept in
    parallel, and the sparse products run row-parallel."""
    from oracle.oracle import OracleMG
    import scipy.sparse as sp
    L = mg.n_levels
    perms = [mg.perm(l) for l in range(L)]
    Ps = [sp.csr_matrix(mg.matrix(l, "P_full"))[perms[l - 1]][:, perms[l]].tocsc() for l in range(1, L)]
    Ai = sp.csr_matrix(A)[perms[0]][:, perms[0]].tocsr()
    orc = OracleMG(Ps)
    orc.precompute(Ai)
    b = np.asfortranarray(rhs[perms[0]])
    z0 = np.zeros_like(b)
    colors = [mg.colors(l) for l in range(L - 1)]
    ncpu = os.cpu_count() or 1
    quota = host_info().get("cpu_quota")
    best = None
    sweep = {}
    for th in (8, 16, 32, 64, 128):      # (matrices and level-0 vectors placed by first touch since round 5, orc_enable_parallel; the box's CPU quota, not NUMA, is what turns the sweep)
        if th > ncpu or (quota and th > 2 * quota):      # beyond the container's CPU quota the threads are throttled, not faster
            break
        orc.set_parallel(colors, th)
        orc.solve(b, z0, tol=0.0, max_iter=2)
        t0 = time.time()
        orc.solve(b, z0, tol=0.0, max_iter=4)
        m = int(max(8, min(200, (budget_s / 4) / max((time.time() - t0) / 4, 1e-6))))
        t0 = time.time()
        _, _, rh = orc.solve(b, z0, tol=0.0, max_iter=m)
        ms = 1e3 * (time.time() - t0) / m
        sweep[th] = round(ms, 2)
        if best is None or ms < best[1]:
            best = (th, ms, m, [float(x) for x in rh[:4]])
    if best is None:
        return {"error": "no OpenMP threads"}
    return {"value": 1e3 / best[1], "unit": "V-cycles/s", "cores": best[0],
            "kind": "port, OpenMP multi-colour variant (not the reference's path)",
            "sample": "%d outer iterations of the same workload in the colour-major numbering, oracle/smg_oracle.c all-core mode; "
                      "best of a thread sweep (ms per cycle by threads: %s); host has %d cores" % (best[2], sweep, ncpu),
            "ms_per_cycle": best[1], "r_his_head": best[3]}


def median_us(torch, stream, fn, reps, repeats=5, warm=10):
    """us per call of fn (launches on `stream`): median of `repeats` HIP-event-timed loops of `reps` calls -- one stalled loop (a clock ramp, a
    host hiccup between eager launches) must not become the reported figure (a single 200-launch loop once gave 274 us for a
    122 us kernel)."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = []
    for _ in range(max(1, repeats)):
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
        torch.cuda.synchronize()
        t.append(1e3 * e0.elapsed_time(e1) / reps)
    return float(np.median(t)), [float(v) for v in t]


def roofline_c5(smg, mesh, torch, dev, stream, reps=200):
    """The same two kernels at a working set the 256 MiB Infinity Cache cannot hold (BASELINE config C5: torus, 4 194 304 vertices,
    matrix 352 MB): the fine-level y = A x and one Gauss-Seidel sweep, HIP events on the launch stream.  This is the HBM number;
    the C3 matrix (85 MB) is served partly on-die between back-to-back launches."""
    mg, A, Mb, Vf, Ff, label, t_host = build_workload("C5", smg, mesh)
    mg.precompute(A)
    mg.set_stream(stream.cuda_stream)
    n = A.shape[0]
    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.uniform(-1.0, 1.0, n)).to(dev)
    y = torch.empty_like(x)
    b = torch.from_numpy(Mb @ rng.uniform(-1.0, 1.0, n)).to(dev)
    u = torch.zeros_like(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    spmv_us, spmv_all = median_us(torch, stream, lambda: mg.raw_spmv(0, 0, x.data_ptr(), None, y.data_ptr()), reps // 2, 5)
    gs_us = float(np.median([mg.bench_relax(0, 1, 10, max(reps // 20, 8)) / 10.0 for _ in range(5)]))      # graph replay (smg_bench_relax): no host launch cost in the figure
    cyc_us = mg.bench_vcycle(0, 1, 2, 2, 30)
    spmv_bytes = mg.spmv_bytes(0, 1)
    gs_bytes = 12 * A.nnz + 4 * (n + 1) + 24 * n
    st = mg.sell_stats(0, "A")
    ws = 12 * st["padded"] + 16 * n
    traffic, traffic_note = committed_traffic("C5")
    gbs = spmv_bytes / (spmv_us * 1e-6) / 1e9
    # ---- BASELINE config 5: fp32 vs fp64 -- the same SpMV on the fp32 image, and the mixed-precision solve (fp32 V-cycle inside the fp64
    # outer loop) against the fp64 one: ms per iteration, cycles and the residual each reaches
    f32 = None
    try:
        x32 = x.float()
        y32 = torch.empty_like(x32)
        us32, _ = median_us(torch, stream, lambda: mg.raw_spmv_f32(0, x32.data_ptr(), y32.data_ptr()), reps // 2, 5)
        b32 = 8 * A.nnz + 4 * (n + 1) + 8 * n
        mg.raw_spmv(0, 0, x.data_ptr(), None, y.data_ptr())
        torch.cuda.synchronize()
        rel = float((y32.double() - y).abs().max() / y.abs().max())
        z0 = torch.zeros_like(x)
        zz = torch.empty_like(x)
        prec = {}
        for pr in ("f64", "mixed"):
            o = smg.SolveOpts(tol=0.0, max_iter=40, precision=pr)
            mg.solve_begin(b.data_ptr(), n, z0.data_ptr(), n, 1, opts=o)
            mg.outer_iterations(40)
            _, rh = mg.solve_end(zz.data_ptr(), n, max_iter=40)
            mg.solve_begin(b.data_ptr(), n, z0.data_ptr(), n, 1, opts=smg.SolveOpts(tol=0.0, max_iter=140, precision=pr))
            mg.outer_iterations(20)
            torch.cuda.synchronize()
            e0.record(stream)
            mg.outer_iterations(100)
            e1.record(stream)
            torch.cuda.synchronize()
            mg.solve_end(zz.data_ptr(), n, max_iter=140)
            prec[pr] = {"ms_per_iteration": e0.elapsed_time(e1) / 100, "first_residual": float(rh[0]), "residual_floor_40_cycles": float(rh.min()),
                        "cycles_to_1e-10_relative": int(np.argmax(rh < 1e-10 * rh[0])) if (rh < 1e-10 * rh[0]).any() else None}
        f32 = {"spmv_f32": {"kernel": "k_sell<SELL_AX,1,float> (fp32 values and vectors, same slots)", "us_per_launch": us32, "bytes_per_launch": int(b32),
                            "achieved": b32 / (us32 * 1e-6) / 1e9, "unit": "GB/s", "frac": b32 / (us32 * 1e-6) / 1e9 / HBM_PEAK_GBS,
                            "max_rel_diff_vs_f64": rel, "speedup_vs_f64": spmv_us / us32},
               "solve_f64": prec["f64"], "solve_mixed": prec["mixed"],
               "tolerance_note": "mixed = fp32 V-cycle as the correction inside the fp64 outer loop (residual in fp64): it reaches the fp64 floor; the fp32 SpMV alone differs from fp64 by max_rel_diff_vs_f64"}
        del x32, y32, z0, zz
    except Exception as e:
        f32 = {"error": repr(e)}
    return {"workload": label, "f32": f32, "kernel": "k_sell<SELL_AX,1> (fine-level y = A x)", "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": gbs / HBM_PEAK_GBS, "frac_of_sustainable": gbs / HBM_SUSTAINABLE_GBS, "bytes_per_launch": int(spmv_bytes), "us_per_launch": spmv_us,
            "us_per_launch_repeats": spmv_all, "timing": "median of 5 HIP-event-timed loops",
            "working_set_bytes": int(ws), "infinity_cache_resident": bool(ws < INFINITY_CACHE_BYTES),
            "traffic_committed_pmc": traffic, "traffic_source": traffic_note,
            "gs_sweep": {"us_per_sweep": gs_us, "bytes_per_sweep": int(gs_bytes), "achieved": gs_bytes / (gs_us * 1e-6) / 1e9,
                         "frac": gs_bytes / (gs_us * 1e-6) / 1e9 / HBM_PEAK_GBS},
            "vcycle_us": cyc_us, "vcycle_bytes": int(mg.vcycle_bytes(1, 2, 2)), "setup_s": t_host}


def timed_repeats(torch, dist, world, dev, stream, block, K, R):
    """R repeats of block(K), each bracketed by barrier + synchronize on both sides and timed with HIP events on the solve's stream;
    per repeat the MAX over ranks.  Returns the list of R times in ms."""
    times = []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(R):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        e0.record(stream)
        block(K)
        e1.record(stream)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        if world > 1:
            t = torch.tensor([ms], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ms = float(t.item())
        times.append(ms)
    return times


def reduce_closure(world, stream_ar):
    """the reduction smg_solve_sharded calls between the two halves of an iteration"""
    if world == 1:
        return lambda p, c, s: None
    if stream_ar is not None:
        return lambda p, c, s: stream_ar(p, c)       # RCCL, enqueued on the solve's own stream
    from surface_multigrid_code_amd.dist import HostReduce
    return HostReduce()                               # any other backend: through the host


def c3_k64_sharded(smg, mg, Mb, n, torch, dist, rank, world, dev, stream, stream_ar, steps=20, warmup=3, repeats=3):
    """A column-sharded job that DOES shard usefully (ogre.obj's 64 columns are launch-latency-bound, strong scaling of that
    job is bounded at about 1.7x): the C3 mesh (1 011 330 vertices, bandwidth-bound) with k = 64 right-hand-side columns, column_range
    over the ranks, STRONG scaling (the 64 columns are the whole job at every N; N = 1 is the curve's first point).  The reference's
    Gauss-Seidel cycle; the solve runs through smg_solve_sharded (the library's own loop, RCCL closure)."""
    from surface_multigrid_code_amd.dist import column_range, sharded_solve_native
    K64, tol = 64, 1e-10
    lo, hi = column_range(K64, rank, world)
    kl = hi - lo
    cols = np.stack([Mb @ np.random.default_rng(1000 + j).uniform(-1.0, 1.0, n) for j in range(lo, hi)], axis=0) if kl else np.zeros((0, n))
    rhs = torch.from_numpy(np.ascontiguousarray(cols)).to(dev)          # (k_local, n): column-major n x k_local
    z0 = torch.zeros_like(rhs)
    red = reduce_closure(world, stream_ar)
    opts = smg.SolveOpts(tol=tol, max_iter=60, smoother="gs")
    block_gs = None
    mixed = None
    sharded_solve_native(mg, rhs if kl else None, z0 if kl else None, red, None, opts)      # warm: graph capture for this k
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    tw = time.perf_counter()
    conv, z, rh = sharded_solve_native(mg, rhs if kl else None, z0 if kl else None, red, None, opts)
    torch.cuda.synchronize()
    wall = time.perf_counter() - tw
    ms = None
    if kl:      # steady state (every rank owns columns at N <= 64)
        HIS = (warmup + steps * repeats) + 4
        sumsq = torch.zeros(1, dtype=torch.float64, device=dev)
        mg.solve_begin(rhs.data_ptr(), n, z0.data_ptr(), n, kl, opts=smg.SolveOpts(tol=0.0, max_iter=HIS, smoother="gs"))

        def block(m):
            if world == 1:
                mg.outer_iterations(m)
                return
            for _ in range(m):
                mg.iter_residual(sumsq.data_ptr())
                red(sumsq.data_ptr(), 1, stream.cuda_stream)
                mg.iter_cycle(sumsq.data_ptr())
        block(warmup)
        times = timed_repeats(torch, dist, world, dev, stream, block, steps, repeats)
        zt = torch.empty_like(z0)
        mg.solve_end(zt.data_ptr(), n, max_iter=HIS)
        ms = float(np.median(times)) / steps
        # the same job with the block Gauss-Seidel sweeps on the fine level (smg_hierarchy_set_block_gs, an option: another valid sweep order,
        # the iterate read ~1.65 instead of 3 times per sweep); only where this rank's column count is a multiple of 16
        if kl % 16 == 0 and kl >= 16 and world == 1:
            try:
                mg.set_block_gs(500000)
                cb, rhb = mg.solve_device(rhs.data_ptr(), z0.data_ptr(), zt.data_ptr(), n, kl, opts=opts)
                mg.solve_begin(rhs.data_ptr(), n, z0.data_ptr(), n, kl, opts=smg.SolveOpts(tol=0.0, max_iter=HIS, smoother="gs"))
                block(warmup)
                tb = timed_repeats(torch, dist, world, dev, stream, block, steps, repeats)
                mg.solve_end(zt.data_ptr(), n, max_iter=HIS)
                info = mg.block_gs_order(0, kl)
                block_gs = {"ms_per_step": float(np.median(tb)) / steps, "cycles_to_tol": len(rhb) - 1, "converged": bool(cb), "levels": [0],
                            "blocks": len(info["blk_ptr"]) - 1, "block_colours": len(info["color_ptr"]) - 1, "rim_rows_per_row": info["rim"],
                            "relax1_level0_us": {"block": mg.bench_relax(0, kl, 1, 10)}}
                mg.set_block_gs(-1)
                block_gs["relax1_level0_us"]["multi_colour"] = mg.bench_relax(0, kl, 1, 10)
            except Exception as e:   # noqa: BLE001
                block_gs = {"error": repr(e)}
                mg.set_block_gs(-1)
        # the same job in the mixed-precision mode (fp32 V-cycle on fp32 images, fp64 outer residual and update: BASELINE config 5's comparison on the
        # column-sharded job; an option -- the headline and this leg's ms_per_step are fp64 throughout)
        if world == 1:
            try:
                om = smg.SolveOpts(tol=tol, max_iter=60, smoother="gs", precision="mixed")
                cm, rhm = mg.solve_device(rhs.data_ptr(), z0.data_ptr(), zt.data_ptr(), n, kl, opts=om)
                mg.solve_begin(rhs.data_ptr(), n, z0.data_ptr(), n, kl, opts=smg.SolveOpts(tol=0.0, max_iter=HIS, smoother="gs", precision="mixed"))
                block(warmup)
                tm = timed_repeats(torch, dist, world, dev, stream, block, steps, repeats)
                mg.solve_end(zt.data_ptr(), n, max_iter=HIS)
                mixed = {"ms_per_step": float(np.median(tm)) / steps, "cycles_to_tol": len(rhm) - 1, "converged": bool(cm), "final_residual": float(rhm[-1])}
            except Exception as e:   # noqa: BLE001
                mixed = {"error": repr(e)}
    same = True
    if world > 1:
        hbuf = torch.zeros(64, dtype=torch.float64, device=dev)
        hbuf[: min(len(rh), 64)] = torch.from_numpy(np.asarray(rh[:64])).to(dev)
        hmin, hmax = hbuf.clone(), hbuf.clone()
        dist.all_reduce(hmin, op=dist.ReduceOp.MIN)
        dist.all_reduce(hmax, op=dist.ReduceOp.MAX)
        same = bool(torch.equal(hmin, hmax))
    del rhs, z0
    if rank != 0:
        return None
    return {"workload": "C3 mesh (1 011 330 verts, 5 levels), k = 64 RHS columns M g_j, column-sharded", "k": K64, "columns_per_gpu": kl, "n_gpus": world,
            "scaling": "strong", "smoother": "gs", "loop": "smg_solve_sharded (C++ loop, %s)" % ("no reduction at N = 1" if world == 1 else "RCCL closure on the solve stream" if stream_ar is not None else "host closure"),
            "ms_per_step": ms, "v_cycles_per_s": 1e3 / ms if ms else None, "column_cycles_per_s": K64 * 1e3 / ms if ms else None,
            # algorithmic bytes of one outer iteration with the columns THIS GPU owns (smg_vcycle_bytes: a sweep charged one read of the iterate).
            # NB the multi-colour sweep of the wide kernels cannot move that little: each of the 4 colour launches streams the other three
            # colours' iterate (512 B per row at 64 columns: beyond every cache), 5.4 n k 8 B per sweep measured against 3 n k 8 algorithmic
            # (profiles/r04_pmc_summary_C3_k64.json: 698 MB per level-0 colour launch at 5.2 TB/s, the memory's rate) -- so `frac` here is
            # bounded by ~0.45, not 1
            "bytes_per_step": int(mg.vcycle_bytes(kl, 2, 2)) if kl else None,
            "gbs": (mg.vcycle_bytes(kl, 2, 2) / (ms * 1e-3) / 1e9) if (kl and ms) else None,
            "frac": (mg.vcycle_bytes(kl, 2, 2) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (kl and ms) else None,
            "block_gs_option": block_gs,
            "mixed_precision_option": mixed,
            "solve": {"tol": tol, "converged": bool(conv), "cycles": len(rh) - 1, "wall_ms": 1e3 * wall, "same_history_on_all_ranks": same,
                      "final_residual": float(rh[-1]) if len(rh) else None}}


def c3_k3_leg(smg, mg, A, Mb, Vf, n, torch, dev, stream, with_cpu=True):
    """The reference's real multi-right-hand-side shape: THREE columns -- one mean-curvature-flow step solves (M - dt L) U' = M U for the x / y / z coordinates at once
    (05_example_mean_curvature_flow/main.cpp:74-76, tol 5e-7 at :60).  C3 mesh and hierarchy, RHS = M V, z0 = V (the step's own start), the reference's Gauss-Seidel cycle.
    ms per outer iteration graph-replayed; bytes: the cycle's model with k = 3 (per operator 12 nnz + 16 n k: smg_vcycle_bytes); the oracle on the same three columns beside it."""
    U = np.ascontiguousarray(Vf[:, :3].T)                                   # (3, n): column-major n x 3
    rhs_h = np.ascontiguousarray(np.stack([Mb @ U[c] for c in range(3)], axis=0))
    rhs = torch.from_numpy(rhs_h).to(dev)
    z0 = torch.from_numpy(U).to(dev)
    z = torch.empty_like(z0)
    ms = steady_ms(torch, stream, mg, rhs, z0, z, n, 3, dict(smoother="gs"), warm=20, iters=100)
    byt = int(mg.vcycle_bytes(3, 2, 2))
    o = smg.SolveOpts(tol=5e-7, max_iter=60, smoother="gs")
    mg.solve_device(rhs.data_ptr(), z0.data_ptr(), z.data_ptr(), n, 3, opts=o)
    torch.cuda.synchronize()
    tw = time.perf_counter()
    cv, rh = mg.solve_device(rhs.data_ptr(), z0.data_ptr(), z.data_ptr(), n, 3, opts=o)
    wall = 1e3 * (time.perf_counter() - tw)
    out = {"workload": "C3 mesh and hierarchy, k = 3: (M + 0.01(-L)) U' = M V for the three coordinate columns, z0 = V (05_example_mean_curvature_flow/main.cpp:74-76)",
           "k": 3, "smoother": "gs", "ms_per_step": ms, "v_cycles_per_s": 1e3 / ms, "column_cycles_per_s": 3e3 / ms, "bytes_per_step": byt,
           "gbs": byt / (ms * 1e-3) / 1e9, "frac": byt / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "solve": {"tol": 5e-7, "converged": bool(cv), "cycles": len(rh) - 1, "wall_ms": wall, "final_residual": float(rh[-1]) if len(rh) else None}}
    if with_cpu:
        cb = oracle_cycle_ms(mg, A, np.asfortranarray(rhs_h.T), budget_s=6.0)
        out["cpu_baseline"] = cb
        out["speedup_vs_cpu_baseline"] = cb["ms_per_cycle"] / ms
    return out


def host_info():
    """CPU model + logical cores of this box, printed once per line (the CPU comparators beside the legs ran here)"""
    model = None
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    quota = None
    try:      # the container's CPU quota (cgroup v2): "max 100000" = none; "1600000 100000" = 16 CPUs' worth of time -- what bounds every all-core CPU figure here
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        quota = None if q == "max" else float(q) / float(p)
    except Exception:
        pass
    return {"cpu_model": model, "nproc": os.cpu_count(), "cpu_quota": quota}


def oracle_cycle_ms(mg, A, rhs, budget_s=6.0, known=None, known_val=None):
    """ms per outer iteration (residual + V(2,2) cycle) of the CPU oracle -- the reference's algorithm, 1 thread -- on the hierarchy of `mg`, bounded sample"""
    from oracle.oracle import OracleMG
    orc = OracleMG([mg.matrix(l, "P_full") for l in range(1, mg.n_levels)])
    t0 = time.time()
    orc.precompute(A, known)
    t_pre = time.time() - t0
    z0 = np.zeros_like(rhs)
    t0 = time.time()
    orc.solve(rhs, z0, known_val, tol=0.0, max_iter=1)
    t1 = time.time() - t0
    m = int(max(2, min(400, budget_s / max(t1, 1e-6))))
    t0 = time.time()
    orc.solve(rhs, z0, known_val, tol=0.0, max_iter=m)
    dt = time.time() - t0
    return {"ms_per_cycle": 1e3 * dt / m, "v_cycles_per_s": m / dt, "cores": 1, "kind": "port", "precompute_s": t_pre,
            "sample": "%d outer iterations, oracle/smg_oracle.c, gcc -O3, 1 thread" % m}


def algorithmic_bytes(mg):
    """the hierarchy as the reference holds it (mg_data: A, P, PT per level in CSC: 12 bytes per stored entry)"""
    return int(sum(12 * mg.matrix(l, "A").nnz for l in range(mg.n_levels)) + sum(24 * mg.matrix(l, "P").nnz for l in range(1, mg.n_levels)))


def memory_lean_leg(smg, mg, A, rhs, z0, z, n, torch, stream):
    """smg_hierarchy_set_memory_lean: the same hierarchy with compact SELL panels -- bytes of the handle after a solve, ms per outer iteration, same bits"""
    m2 = smg.Hierarchy.from_prolongs([mg.matrix(l, "P_full") for l in range(1, mg.n_levels)])
    m2.set_memory_lean(True)
    m2.precompute(A)
    m2.set_stream(stream.cuda_stream)
    o = smg.SolveOpts(tol=1e-10, max_iter=100)
    cv, rh = m2.solve_device(rhs.data_ptr(), z0.data_ptr(), z.data_ptr(), n, 1, opts=o)
    z_lean = z.clone()
    ms = steady_ms(torch, stream, m2, rhs, z0, z, n, 1, dict(smoother="gs"), iters=100)
    live = int(m2.device_bytes()["total"])
    cv0, rh0 = mg.solve_device(rhs.data_ptr(), z0.data_ptr(), z.data_ptr(), n, 1, opts=o)
    same = bool(torch.equal(z, z_lean)) and list(rh) == list(rh0)
    alg = algorithmic_bytes(m2)
    del m2
    return {"handle_bytes": live, "ratio_to_algorithmic": live / max(alg, 1), "ms_per_step": ms, "cycles_to_1e-10": len(rh) - 1, "converged": bool(cv), "bit_identical_to_default": same}


def steady_ms(torch, stream, mg, rhs, z0, z, n, k, opts_kw, warm=30, iters=200, repeats=3, his=1024):
    """ms per outer iteration, graph-replayed, HIP events on the solve stream (tol = 0: every iteration is a full one); median of `repeats`"""
    import surface_multigrid_code_amd as smg
    ts = []
    mg.solve_begin(rhs.data_ptr(), n, z0.data_ptr(), n, k, opts=smg.SolveOpts(tol=0.0, max_iter=his, **opts_kw))
    mg.outer_iterations(warm)
    ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(repeats):
        torch.cuda.synchronize()
        ea.record(stream)
        mg.outer_iterations(iters)
        eb.record(stream)
        torch.cuda.synchronize()
        ts.append(ea.elapsed_time(eb) / iters)
    mg.solve_end(z.data_ptr(), n, max_iter=his)
    return float(np.median(ts))


def c3_decimated_leg(smg, mesh, torch, dev, stream, ms_subdiv, bytes_subdiv):
    """The C3 mesh under the REFERENCE's own kind of hierarchy: mg_precompute(V, F, 0.25, 1000, midpoint) on the 1 011 330-vertex mesh
    (src/mg_precompute.cpp:15-87, get_prolong.cpp:45-56, the call of 03_mg_solver/main.cpp:35-39) -- SSP decimation, 3 entries per row of P,
    Galerkin operators of 18 - 30 entries per row -- instead of the subdivision operators the headline's hierarchy is made of.  Same mesh,
    same system, same right-hand side, the reference's Gauss-Seidel V(2,2): hierarchy-build seconds, the level table, ms per outer iteration,
    its own algorithmic bytes and fraction of the HBM peak, cycles to 1e-10, the oracle beside it; `cost_per_byte_vs_subdivision` = (ms per
    byte of this cycle) / (ms per byte of the headline's cycle)."""
    mg, A, Mb, Vf, Ff, label, t_host = build_workload("C3dec", smg, mesh)
    n = A.shape[0]
    torch.cuda.synchronize()
    t0 = time.time()
    mg.precompute(A)
    t_pre = time.time() - t0
    mg.set_stream(stream.cuda_stream)
    rhs_h = Mb @ np.random.default_rng(100).uniform(-1.0, 1.0, n)
    rhs = torch.from_numpy(rhs_h).to(dev)
    z0 = torch.zeros(n, dtype=torch.float64, device=dev)
    z = torch.empty_like(z0)
    o = smg.SolveOpts(tol=1e-10, max_iter=100)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    mg.solve_device(rhs.data_ptr(), z0.data_ptr(), z.data_ptr(), n, 1, opts=o)
    torch.cuda.synchronize()
    t_first = time.perf_counter() - t1      # the first solve on the handle: sweep plans of the levels the precompute had no idle time for, graph capture, 10 cycles
    levels = []
    for lv in range(mg.n_levels):
        M = mg.matrix(lv, "A")
        nn = np.diff(M.indptr)
        row = {"rows": int(M.shape[0]), "entries_per_row": float(nn.mean()), "entries_per_row_max": int(nn.max())}
        if lv < mg.n_levels - 1:
            row["colors"] = len(mg.colors(lv)) - 1
            w = mg.wave_gs_order(lv, 1)
            row["relax"] = ("wave Gauss-Seidel: %d piece colours (launches per sweep), phases per piece %.1f (max %d)" % (len(w["color_ptr"]) - 1, w["phases_mean"], w["phases_max"])) if w is not None else "one launch per colour"
        if lv > 0:
            pn = np.diff(mg.matrix(lv, "PT").indptr)
            row["P_entries_per_fine_row"] = float(mg.matrix(lv, "P").nnz) / mg.rows(lv - 1)
            row["PT_entries_per_coarse_row_max"] = int(pn.max())
        levels.append(row)
    cv, rh = mg.solve_device(rhs.data_ptr(), z0.data_ptr(), z.data_ptr(), n, 1, opts=o)
    ms = steady_ms(torch, stream, mg, rhs, z0, z, n, 1, dict(smoother="gs"))
    byt = int(mg.vcycle_bytes(1, 2, 2))
    mg.set_wave_gs("never")
    ms_colour = steady_ms(torch, stream, mg, rhs, z0, z, n, 1, dict(smoother="gs"), iters=60)
    cvc, rhc = mg.solve_device(rhs.data_ptr(), z0.data_ptr(), z.data_ptr(), n, 1, opts=o)
    mg.set_wave_gs("auto")
    out = {"workload": label, "what": "outer iteration = residual + norm + break test + V(2,2), Gauss-Seidel on every level (the reference's cycle), 1 RHS column, graph-replayed; HIP events on the solve stream",
           "setup_s": {"mesh_and_hierarchy_host": t_host, "mg_precompute_decimated": getattr(build_workload, "mg_precompute_s", None), "smg_precompute": t_pre, "first_solve_s": t_first},
           "levels": levels, "ms_per_step": ms, "v_cycles_per_s": 1e3 / ms, "bytes_per_step": byt, "gbs": byt / (ms * 1e-3) / 1e9,
           "frac_of_hbm_peak": byt / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "cycles_to_1e-10": len(rh) - 1, "converged": bool(cv), "time_to_tol_ms": ms * (len(rh) - 1),
           "one_launch_per_colour": {"ms_per_step": ms_colour, "cycles_to_1e-10": len(rhc) - 1, "converged": bool(cvc), "note": "the same handle with smg_hierarchy_set_wave_gs(h, 0): multi-colour order on every level"},
           "cost_per_byte_vs_subdivision": (ms / byt) / (ms_subdiv / bytes_subdiv) if ms_subdiv and bytes_subdiv else None,
           "subdivision_cycle": {"ms_per_step": ms_subdiv, "bytes_per_step": int(bytes_subdiv) if bytes_subdiv else None},
           # THIS handle's HBM (round 5 printed the process-wide figure here, the headline's handle and the other legs' included: 2.31 GB)
           "device_bytes_live": int(mg.device_bytes()["total"]), "hierarchy_algorithmic_bytes": algorithmic_bytes(mg),
           "device_bytes_process_wide": int(smg._lib.load().smg_device_bytes_live())}
    out["device_bytes_ratio"] = out["device_bytes_live"] / max(out["hierarchy_algorithmic_bytes"], 1)
    try:
        out["cpu_baseline"] = oracle_cycle_ms(mg, A, rhs_h, budget_s=8.0)
        out["speedup_vs_cpu_baseline"] = out["cpu_baseline"]["ms_per_cycle"] / ms
    except Exception as e:     # noqa: BLE001
        out["cpu_baseline"] = {"error": repr(e)}
    # The caller's knob, not the library's: mg_precompute's nVCoarsest (src/mg_precompute.cpp:15, 03_mg_solver/main.cpp:37).  The same hierarchy stopped one
    # level earlier -- coarsest level 15 804 unknowns, solved by the Schur-complement solver -- is ANOTHER cycle (an exact solve where the 5-level cycle
    # smooths and recurses): reported beside the leg, never in place of it.
    try:
        Ps = [mg.matrix(l, "P_full") for l in range(1, mg.n_levels - 1)]
        m4 = smg.Hierarchy.from_prolongs(Ps)
        t0 = time.time()
        m4.precompute(A)
        t4 = time.time() - t0
        m4.set_stream(stream.cuda_stream)
        m4.solve_device(rhs.data_ptr(), z0.data_ptr(), z.data_ptr(), n, 1, opts=o)
        cv4, rh4 = m4.solve_device(rhs.data_ptr(), z0.data_ptr(), z.data_ptr(), n, 1, opts=o)
        ms4 = steady_ms(torch, stream, m4, rhs, z0, z, n, 1, dict(smoother="gs"), iters=60)
        out["stopped_one_level_earlier_option"] = {"levels": [m4.rows(l) for l in range(m4.n_levels)], "coarse_solver": m4.coarse_solver(), "smg_precompute_s": t4, "ms_per_step": ms4,
                                                   "cycles_to_1e-10": len(rh4) - 1, "converged": bool(cv4), "time_to_tol_ms": ms4 * (len(rh4) - 1),
                                                   "note": "mg_precompute with a larger nVCoarsest: the caller's choice; a different cycle"}
        del m4
    except Exception as e:     # noqa: BLE001
        out["stopped_one_level_earlier_option"] = {"error": repr(e)}
    return out


def c1_leg(smg, mesh, torch, dev, stream):
    """BASELINE config C1 (the reference's own CPU-runnable case): 03_mg_solver/main.cpp:29-75 -- A = -cotmatrix, the longest boundary loop pinned to 0,
    B = M_voronoi 1, z0 = 0, hierarchy by mg_precompute defaults, tol 1e-3 / maxIter 20 -- on bunny.obj (BASELINE names it) and ogre.obj (what main.cpp:29
    loads): cycles and wall time of the drop-in solve on the GPU and of the oracle (the reference's algorithm, 1 thread) on this host; steady-state ms per
    outer iteration of both."""
    out = {}
    for name in ("bunny", "ogre"):
        V, F = mesh.read_triangle_mesh(name + ".smgm")
        V = mesh.normalize_unit_area(V, F)
        n = V.shape[0]
        t0 = time.time()
        mg = smg.mg_precompute(V, F, 0.25, 500, 1)
        t_h = time.time() - t0
        A = (-mesh.cotmatrix(V, F)).tocsr()
        A.sort_indices()
        b = mesh.boundary_loop(F)
        Bv = mesh.massmatrix(V, F, "voronoi") @ np.ones(n)
        Bv[b] = 0.0
        t0 = time.time()
        mg.precompute(A, b)
        t_p = time.time() - t0
        mg.set_stream(stream.cuda_stream)
        kv = np.zeros(len(b))
        rec = {"verts": int(n), "pinned": int(len(b)), "levels": [mg.rows(l) for l in range(mg.n_levels)], "setup_s": {"mg_precompute": t_h, "smg_precompute": t_p}}
        for tol, mx in ((1e-3, 20), (1e-10, 100)):
            o = smg.SolveOpts(tol=tol, max_iter=mx)
            t0 = time.perf_counter()
            mg.solve(Bv, np.zeros(n), kv, o)                      # warm: the first solve on the handle builds the sweep plans and captures the graphs
            if "first_solve_ms" not in rec["setup_s"]:
                rec["setup_s"]["first_solve_ms"] = 1e3 * (time.perf_counter() - t0)      # (the 03 pattern: precompute once, solve once -- pays this)
            t0 = time.perf_counter()
            cv, z, rh = mg.solve(Bv, np.zeros(n), kv, o)          # host vectors in, host vectors out: what the drop-in caller sees
            rec["tol_%g" % tol] = {"converged": bool(cv), "cycles": len(rh) - 1, "solve_wall_ms_host_vectors": 1e3 * (time.perf_counter() - t0), "final_residual": float(rh[-1])}
        nu = mg.rows(0)
        rhs = torch.zeros(n, dtype=torch.float64, device=dev); rhs.copy_(torch.from_numpy(Bv))
        z0 = torch.zeros(n, dtype=torch.float64, device=dev); zz = torch.empty_like(z0)
        kvd = torch.zeros(len(b), dtype=torch.float64, device=dev)
        mg.solve_begin(rhs.data_ptr(), n, z0.data_ptr(), n, 1, known_val_ptr=kvd.data_ptr(), ld_kv=len(b), opts=smg.SolveOpts(tol=0.0, max_iter=1024))
        mg.outer_iterations(30)
        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); ea.record(stream); mg.outer_iterations(300); eb.record(stream); torch.cuda.synchronize()
        mg.solve_end(zz.data_ptr(), n, max_iter=1024)
        rec["ms_per_step"] = ea.elapsed_time(eb) / 300
        rec["unknowns"] = int(nu)
        try:
            from oracle.oracle import OracleMG
            orc = OracleMG([mg.matrix(l, "P_full") for l in range(1, mg.n_levels)])
            orc.precompute(A, b)
            cpu = {}
            for tol, mx in ((1e-3, 20), (1e-10, 100)):
                t0 = time.perf_counter()
                cv2, z2, rh2 = orc.solve(Bv, np.zeros(n), kv, tol=tol, max_iter=mx)
                cpu["tol_%g" % tol] = {"converged": bool(cv2), "cycles": len(rh2) - 1, "solve_wall_ms": 1e3 * (time.perf_counter() - t0)}
            cpu.update(oracle_cycle_ms(mg, A, Bv, budget_s=2.0, known=b, known_val=kv))
            rec["cpu_baseline"] = cpu
            rec["speedup_vs_cpu_baseline_per_cycle"] = cpu["ms_per_cycle"] / rec["ms_per_step"]
        except Exception as e:     # noqa: BLE001
            rec["cpu_baseline"] = {"error": repr(e)}
        out[name + ".obj"] = rec
        del mg
    out["what"] = "03_mg_solver on the reference's own meshes: GPU = libsmg (Gauss-Seidel V(2,2), graph-replayed); cpu_baseline = oracle/smg_oracle.c, 1 thread, this host"
    return out


def block3_leg(smg, mesh, torch, with_scalar=True):
    """the block (3-DOF) kernels against the scalar kernels on the same 3n x 3n system -- C3 mesh, kron(S, C3) with a full
    SPD 3 x 3 coupling, hierarchy P (x) I_3 -- Gauss-Seidel V(2,2): ms per iteration, launches per sweep (colours), algorithmic bytes
    (76 B per 3 x 3 block against 108 B for nine scalar entries)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import block3_time as B3
    A, Ps, label = B3.block_system(smg, mesh, "C3")
    out = {"workload": label + " -> kron(S, C3): %d DOFs, %d stored entries, DOF = 3 v + d, P (x) I_3" % (A.shape[0], A.nnz)}
    out["block"] = B3.measure(smg, torch, A, Ps, "block", "gs", reps=100)
    if with_scalar:
        out["scalar"] = B3.measure(smg, torch, A, Ps, "scalar", "gs", reps=50)
        out["speedup_block_vs_scalar"] = out["scalar"]["ms_per_iteration"] / out["block"]["ms_per_iteration"]
    return out


def multi_mesh_leg(smg, mesh, torch, dev, counts=(1, 2, 4, 8), steps=300, repeats=3):
    """BASELINE north_star: 'independent RHS columns / independent meshes shard' -- independent meshes on ONE GPU.  A mesh the size of
    ogre.obj (19 985 vertices, the reference's 03 / 05 example mesh) is pure launch latency (16 workgroups per colour launch on 256 CUs), so
    M independent solves -- M handles, each with its own stream (handles are thread-safe across handles), driven by M host threads -- should
    overlap.  Reported: aggregate V-cycles/s (outer iterations of the reference's Gauss-Seidel V(2,2) cycle, graph-replayed) for M handles
    at once, wall clock from a common start until every stream has drained, median of `repeats`; and what one handle holds in HBM."""
    import threading
    V, F = mesh.read_triangle_mesh("ogre.smgm")
    V = mesh.normalize_unit_area(V, F)
    mg0 = smg.mg_precompute(V, F, 0.25, 500, 1)
    Ps = [mg0.matrix(l, "P_full") for l in range(1, mg0.n_levels)]
    Mb = mesh.massmatrix(V, F, "barycentric")
    A = (Mb - 0.01 * mesh.cotmatrix(V, F)).tocsr()
    A.sort_indices()
    n = A.shape[0]
    L = smg._lib.load()
    live0 = int(L.smg_device_bytes_live())
    M = max(counts)
    handles, vecs = [], []
    for i in range(M):
        h = smg.Hierarchy.from_prolongs(Ps)
        h.precompute(A)
        rhs = torch.from_numpy(Mb @ np.random.default_rng(300 + i).uniform(-1.0, 1.0, n)).to(dev)
        z0 = torch.zeros(n, dtype=torch.float64, device=dev)
        handles.append(h)
        vecs.append((rhs, z0, torch.empty_like(z0)))
    torch.cuda.synchronize()
    HIS = 4096
    res = {}
    for m in counts:
        times = []
        for rep_i in range(repeats + 1):          # the first round warms (graph capture per handle)
            for i in range(m):
                handles[i].solve_begin(vecs[i][0].data_ptr(), n, vecs[i][1].data_ptr(), n, 1, opts=smg.SolveOpts(tol=0.0, max_iter=HIS, smoother="gs"))
                handles[i].outer_iterations(10)
                handles[i].synchronize()
            bar = threading.Barrier(m + 1)
            errs = []

            def work(i):
                try:
                    bar.wait()
                    handles[i].outer_iterations(steps)
                    handles[i].synchronize()
                except Exception as e:     # noqa: BLE001
                    errs.append(repr(e))
            th = [threading.Thread(target=work, args=(i,)) for i in range(m)]
            for t in th:
                t.start()
            bar.wait()
            t0 = time.perf_counter()
            for t in th:
                t.join()
            dt = time.perf_counter() - t0
            for i in range(m):
                handles[i].solve_end(vecs[i][2].data_ptr(), n, max_iter=HIS)
            if errs:
                return {"error": errs[0]}
            if rep_i > 0:
                times.append(dt)
        dt = float(np.median(times))
        res[str(m)] = {"handles": m, "v_cycles_per_s_aggregate": m * steps / dt, "ms_per_cycle_per_handle": 1e3 * dt / steps, "wall_ms": [1e3 * t for t in times]}
    live1 = int(L.smg_device_bytes_live())
    base = res[str(counts[0])]["v_cycles_per_s_aggregate"]
    # kernel nodes of one outer iteration of this hierarchy: per smoothed level 4 sweeps x (piece colours or colours, or 1 for a one-launch relax) + residual + transfers
    graph_nodes = 4
    for l in range(handles[0].n_levels - 1):
        w = handles[0].wave_gs_order(l, 1)
        graph_nodes += 4 * ((len(w["color_ptr"]) - 1) if w is not None else (len(handles[0].colors(l)) - 1)) + 4
    try:
        cpu = oracle_cycle_ms(handles[0], A, Mb @ np.random.default_rng(300).uniform(-1.0, 1.0, n), budget_s=3.0)
    except Exception as e:     # noqa: BLE001
        cpu = {"error": repr(e)}
    del handles, vecs
    # ---- the same M meshes in ONE handle through the ABI: smg_hierarchy_create_union (include/smg.h) -- block-diagonal levels, the members' OWN coarse
    # inverses, every member its own residual history and break test.  Every launch serves all M meshes: the launch latency is shared, not multiplied.
    # The hierarchy is the members' own (mg_precompute defaults, the same as by_handles): M = 1 through the union API is the baseline.
    union = {}
    try:
        import scipy.sparse as sp
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st = torch.cuda.current_stream(dev)
        for m in counts:
            hu = smg.Hierarchy.union([mg0] * m)
            Am = sp.block_diag([A] * m, format="csr") if m > 1 else A
            Am.sort_indices()
            hu.precompute(Am)
            hu.set_stream(st.cuda_stream)
            nn = Am.shape[0]
            rhs = torch.from_numpy(np.concatenate([Mb @ np.random.default_rng(300 + i).uniform(-1.0, 1.0, n) for i in range(m)])).to(dev)
            z0 = torch.zeros(nn, dtype=torch.float64, device=dev)
            zz = torch.empty_like(z0)
            # every member to its own tolerance: cycles per member (they differ: each has its own right-hand side)
            cv, rh_u = hu.solve_device(rhs.data_ptr(), z0.data_ptr(), zz.data_ptr(), nn, 1, opts=smg.SolveOpts(tol=1e-10, max_iter=100, smoother="gs"))
            member_cycles = [len(hu.union_history(i)[1]) - 1 for i in range(m)]
            hu.solve_begin(rhs.data_ptr(), nn, z0.data_ptr(), nn, 1, opts=smg.SolveOpts(tol=0.0, max_iter=HIS, smoother="gs"))
            hu.outer_iterations(20)
            ts = []
            for _ in range(repeats):
                torch.cuda.synchronize()
                e0.record(st)
                hu.outer_iterations(steps)
                e1.record(st)
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / steps)
            hu.solve_end(zz.data_ptr(), nn, max_iter=HIS)
            ms = float(np.median(ts))
            union[str(m)] = {"meshes": m, "ms_per_iteration": ms, "v_cycles_per_s_aggregate": m * 1e3 / ms, "levels": hu.n_levels,
                             "coarsest_rows_per_member": int(hu.rows(hu.n_levels - 1) // m), "converged_all": bool(cv), "cycles_to_1e-10_per_member": member_cycles,
                             "device_bytes": int(hu.device_bytes()["total"])}
            del hu, rhs, z0, zz
        union["speedup_at_%d" % max(counts)] = union[str(max(counts))]["v_cycles_per_s_aggregate"] / union[str(counts[0])]["v_cycles_per_s_aggregate"]
        union["vs_one_stand_alone_handle_at_%d" % max(counts)] = union[str(max(counts))]["v_cycles_per_s_aggregate"] / base
        union["what"] = "smg_hierarchy_create_union over M copies of the hierarchy above: block-diagonal levels, per-member coarse inverses, per-member residual history / break test (tol = 0 in the timed loop: every iteration is a full one for every member)"
    except Exception as e:     # noqa: BLE001
        union = {"error": repr(e)}
    out = {"workload": "ogre.obj (19 985 verts, %d levels from smg_mg_precompute), M_bary + 0.01(-L), one RHS column per mesh, Gauss-Seidel V(2,2)" % mg0.n_levels,
           "what": "M independent handles (own streams) driven by M host threads at once; wall clock until all streams drained; median of %d" % repeats,
           "by_handles": res, "speedup_at_%d" % max(counts): res[str(max(counts))]["v_cycles_per_s_aggregate"] / base,
           "by_handles_note": ("aggregate V-cycles/s with M handles relative to one: %s -- an outer iteration is one hipGraphLaunch of ~%d kernel nodes which the runtime enqueues under a "
                               "process-wide lock, so M host threads submit no faster than one (profiles/r04_multi_mesh.txt); independent meshes on one GPU belong in ONE handle: union_in_one_handle"
                               % (", ".join("M = %s: %.2f x" % (m, res[str(m)]["v_cycles_per_s_aggregate"] / base) for m in counts), graph_nodes)),
           "union_in_one_handle": union, "cpu_baseline": cpu,
           "device_bytes_per_handle": (live1 - live0) // M,
           "hierarchy_algorithmic_bytes": int(sum(12 * mg0.matrix(l, "P_full").nnz for l in range(1, mg0.n_levels)) * 2 + 12 * A.nnz)}
    return out


def c4_k64_sharded(smg, mesh, torch, dist, rank, world, dev, stream, stream_ar, smoother_kw, steps=200, warmup=20):
    """BASELINE config C4: mean-curvature-flow system on ogre.obj (05_example_mean_curvature_flow/main.cpp:57-76), k = 64 right-hand
    sides, column-sharded over the ranks: rank g owns columns [g k / N, (g+1) k / N), the hierarchy is
    replicated, the only communication is the 8-byte all-reduce of the residual sum of squares per outer iteration.  STRONG scaling:
    the 64 columns are the whole job at every N.  Returns the rank-0 record (None on the other ranks)."""
    from surface_multigrid_code_amd.dist import GpuEngine, column_range, sharded_solve
    K64, tol = 64, 5e-7                                              # main.cpp:60
    V, F = mesh.read_triangle_mesh("ogre.smgm")
    V = mesh.normalize_unit_area(V, F)
    t0 = time.time()
    mg = smg.mg_precompute(V, F, 0.25, 500, 1)                       # main.cpp:48-50 (defaults of mg_precompute; the reference's collapse order)
    t_hier = time.time() - t0
    n = V.shape[0]
    L = mesh.cotmatrix(V, F)
    Mb = mesh.massmatrix(V, F, "barycentric")                        # main.cpp:67
    A = (Mb - 0.01 * L).tocsr()                                      # main.cpp:68
    A.sort_indices()
    mg.precompute(A)
    mg.set_stream(stream.cuda_stream)
    cols = [V[:, 0], V[:, 1], V[:, 2]] + [np.random.default_rng(100 + j).uniform(-1.0, 1.0, n) for j in range(K64 - 3)]
    X = np.stack(cols, axis=1)                                       # n x 64
    RHS = Mb @ X                                                     # main.cpp:69
    Z0 = np.concatenate([V, np.zeros((n, K64 - 3))], axis=1)         # z0 = U for the coordinate columns (main.cpp:76)
    lo, hi = column_range(K64, rank, world)
    kl = hi - lo
    rhs = torch.from_numpy(np.ascontiguousarray(RHS[:, lo:hi].T)).to(dev)     # (k_local, n): column-major n x k_local
    z0 = torch.from_numpy(np.ascontiguousarray(Z0[:, lo:hi].T)).to(dev)
    sumsq = torch.zeros(1, dtype=torch.float64, device=dev)

    def allreduce(t):
        if world == 1:
            return
        if stream_ar is not None:
            stream_ar(t.data_ptr())
        else:
            dist.all_reduce(t)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # 1. the solve itself (tol 5e-7, the reference's setting): cycles, convergence, identical history on every rank
    opts = smg.SolveOpts(tol=tol, max_iter=100, **smoother_kw)
    eng = GpuEngine(mg, rhs, z0, None, opts)
    sync_all()
    tw = time.perf_counter()
    conv, z, rh = sharded_solve(eng, 100, allreduce, check_every=1)
    torch.cuda.synchronize()
    wall = time.perf_counter() - tw
    # 2. steady state: `steps` outer iterations of the 64-column block with tol = 0 (never converges)
    HIS = max(steps + warmup, 1)
    oo = smg.SolveOpts(tol=0.0, max_iter=HIS, **smoother_kw)
    mg.solve_begin(rhs.data_ptr(), n, z0.data_ptr(), n, kl, opts=oo)

    def block(m):
        if world == 1:
            mg.outer_iterations(m)
            return
        for _ in range(m):
            mg.iter_residual(sumsq.data_ptr())
            allreduce(sumsq)
            mg.iter_cycle(sumsq.data_ptr())
    block(warmup)
    sync_all()
    t0 = time.perf_counter()
    block(steps)
    sync_all()
    dt = time.perf_counter() - t0
    zt = torch.empty_like(z0)
    mg.solve_end(zt.data_ptr(), n, max_iter=HIS)
    # 3. the all-reduce alone, back to back on the solve stream
    ar_us = None
    if world > 1:
        for _ in range(20):
            allreduce(sumsq)
        sync_all()
        ta = time.perf_counter()
        for _ in range(200):
            allreduce(sumsq)
        torch.cuda.synchronize()
        ar_us = 1e6 * (time.perf_counter() - ta) / 200
        t = torch.tensor([dt, ar_us], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, ar_us = float(t[0].item()), float(t[1].item())
        # every rank must have seen the same residual history
        h = torch.zeros(64, dtype=torch.float64, device=dev)
        h[: min(len(rh), 64)] = torch.from_numpy(np.asarray(rh[:64])).to(dev)
        hmin, hmax = h.clone(), h.clone()
        dist.all_reduce(hmin, op=dist.ReduceOp.MIN)
        dist.all_reduce(hmax, op=dist.ReduceOp.MAX)
        same = bool(torch.equal(hmin, hmax))
    else:
        same = True
    ref_gs = None
    if world == 1 and smoother_kw["smoother"] != "gs":    # the reference's smoother on the same job, for the time-to-tolerance comparison
        g_kw = dict(smoother_kw, smoother="gs")
        cg, zg, rg = sharded_solve(GpuEngine(mg, rhs, z0, None, smg.SolveOpts(tol=tol, max_iter=100, **g_kw)), 100, allreduce)
        mg.solve_begin(rhs.data_ptr(), n, z0.data_ptr(), n, kl, opts=smg.SolveOpts(tol=0.0, max_iter=HIS, **g_kw))
        mg.outer_iterations(warmup)
        torch.cuda.synchronize()
        tg = time.perf_counter()
        mg.outer_iterations(steps)
        torch.cuda.synchronize()
        msg = 1e3 * (time.perf_counter() - tg) / steps
        mg.solve_end(zt.data_ptr(), n, max_iter=HIS)
        ref_gs = {"ms_per_step": msg, "cycles": len(rg) - 1, "converged": bool(cg), "time_to_tol_ms": msg * (len(rg) - 1)}
    if rank != 0:
        return None
    ms = 1e3 * dt / steps
    cpu = None
    if world == 1:
        try:    # the reference's algorithm on this host, 1 thread: the 64-column outer iteration (k independent lexicographic sweeps per relax())
            cpu = oracle_cycle_ms(mg, A, np.asfortranarray(RHS), budget_s=4.0)
            cpu["column_cycles_per_s"] = K64 * cpu["v_cycles_per_s"]
        except Exception as e:     # noqa: BLE001
            cpu = {"error": repr(e)}
    return {"cpu_baseline": cpu, "speedup_vs_cpu_baseline": (cpu["ms_per_cycle"] / ms) if cpu and "ms_per_cycle" in cpu else None,
            "workload": "C4: ogre.obj (19985 verts), M_bary+0.01(-L), k = 64 RHS columns (U + 61 synthetic M g_j), mg_precompute hierarchy",
            "levels": [mg.rows(l) for l in range(mg.n_levels)], "colors": [len(mg.colors(l)) - 1 for l in range(mg.n_levels - 1)],
            "k": K64, "columns_per_gpu": kl, "n_gpus": world, "scaling": "strong", "smoother": smoother_kw["smoother"],
            "ms_per_step": ms, "v_cycles_per_s": 1e3 / ms, "column_cycles_per_s": K64 * 1e3 / ms,
            "allreduce_us": ar_us, "allreduce": (("RCCL on the solve stream, %d ranks (ncclCommCount)" % stream_ar.n_ranks()) if stream_ar is not None else "torch.distributed") if world > 1 else None,
            "solve": {"tol": tol, "converged": bool(conv), "cycles": len(rh) - 1, "wall_ms": 1e3 * wall, "same_history_on_all_ranks": same,
                      "r_his_head": [float(v) for v in rh[:4]], "final_residual": float(rh[-1])},
            "time_to_tol_ms": ms * (len(rh) - 1), "reference_gs": ref_gs,
            "hierarchy_build_s": t_hier}



# ------------------------------------------------------------------------------------------------------------------------------------
# The line the driver parses.  Everything the legs measure goes to bench_extra.json (beside this file, and under gpurun_out/ when that
# exists); stdout's LAST line is the compact record below: the contract's keys, `roofline`, `cpu_baseline`, and a handful of scalars.
# Hard cap 8 000 characters (tests/test_host_logic.py::test_bench_line_is_compact): round 5's 23.7 KB line could not be parsed.
LINE_HARD_CAP = 8000


def _num(v, nd=6):
    """a JSON-safe number: finite floats rounded to `nd` significant digits, NaN / Inf -> None (strict JSON has neither)"""
    if isinstance(v, (bool, np.bool_)):
        return bool(v)
    if isinstance(v, (int, np.integer)):
        return int(v)
    if isinstance(v, (float, np.floating)):
        v = float(v)
        if v != v or v in (float("inf"), float("-inf")):
            return None
        return float("%.*g" % (nd, v))
    if isinstance(v, str) and len(v) > 240:      # prose belongs in bench_extra.json
        return v[:237] + "..."
    return v


def _dig(d, *path):
    for p in path:
        if not isinstance(d, dict) or p not in d:
            return None
        d = d[p]
    return d


def compact_line(out):
    """`out` = everything bench.py measured (the dict that goes to bench_extra.json) -> the dict printed as the last stdout line."""
    cfg = out.get("config") or {}
    line = {k: _num(out.get(k)) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                                          "dtype", "data")}
    line["config"] = {k: _num(cfg.get(k)) for k in ("workload", "n_verts", "nnz", "levels", "level_rows", "cycle", "smoother", "rhs_columns_per_gpu", "parallelism")
                      if cfg.get(k) is not None}
    rf = out.get("roofline") or {}
    line["roofline"] = {k: _num(rf.get(k)) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "bytes_per_launch", "us_per_launch", "traffic",
                                                      "traffic_kind", "traffic_source", "infinity_cache_resident")}
    cb = out.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {k: _num(cb.get(k)) for k in ("value", "unit", "cores", "kind", "sample", "ms_per_cycle")}
        line["cpu_baseline"]["host"] = _dig(out, "host", "cpu_model")
        line["speedup_vs_cpu_baseline"] = _num(out.get("speedup_vs_cpu_baseline"), 4)
    pf = out.get("multi_gpu_preflight")
    line["multi_gpu_preflight"] = None if not pf else {k: pf.get(k) for k in ("rccl_comm_ranks", "distinct_devices", "backend", "visible_devices_per_rank", "sharing_detectable")}
    if cfg.get("allreduce") is not None:
        line["config"]["allreduce"] = _num(cfg["allreduce"])
    t = out.get("timing") or {}
    extras = {
        "ms_per_step_min": _dig(t, "ms_per_step_min"), "ms_per_step_max": _dig(t, "ms_per_step_max"),
        "cycles_to_tol": out.get("cycles_to_tol"), "time_to_tol_ms": out.get("time_to_tol_ms"),
        "roofline_vcycle_frac": _dig(out, "roofline_vcycle", "frac"), "vcycle_bytes_per_step": _dig(out, "roofline_vcycle", "bytes_per_step"),
        "roofline_gs_sweep_frac": _dig(out, "roofline_gs_sweep", "frac"), "gs_sweep_us": _dig(out, "roofline_gs_sweep", "us_per_sweep"),
        "roofline_c5_frac": _dig(out, "roofline_c5", "frac"), "roofline_c5_f32_frac": _dig(out, "roofline_c5", "f32", "spmv_f32", "frac"),
        "c3_decimated_ms_per_step": _dig(out, "c3_decimated", "ms_per_step"), "c3_decimated_frac": _dig(out, "c3_decimated", "frac_of_hbm_peak"),
        "c3_decimated_cycles_to_tol": _dig(out, "c3_decimated", "cycles_to_1e-10"),
        "c3_k3_ms_per_step": _dig(out, "c3_k3", "ms_per_step"), "c3_k3_frac": _dig(out, "c3_k3", "frac"),
        "c3_k3_speedup_vs_cpu": _dig(out, "c3_k3", "speedup_vs_cpu_baseline"),
        "c3_k64_ms_per_step": _dig(out, "c3_k64_sharded", "ms_per_step"), "c4_k64_ms_per_step": _dig(out, "c4_k64_sharded", "ms_per_step"),
        "c4_k64_allreduce_us": _dig(out, "c4_k64_sharded", "allreduce_us"),
        "c1_bunny_ms_per_cycle": _dig(out, "c1_03_mg_solver", "bunny.obj", "ms_per_step"), "c1_ogre_ms_per_cycle": _dig(out, "c1_03_mg_solver", "ogre.obj", "ms_per_step"),
        "cpu_allcore_v_cycles_per_s": _dig(out, "cpu_allcore", "value"), "cpu_allcore_cores": _dig(out, "cpu_allcore", "cores"),
        "reprecompute_ms": _dig(out, "reprecompute", "schur_complement", "reprecompute_ms"),
        "block3_ms_per_step": _dig(out, "block3_c3", "block", "ms_per_step"),
        "multi_mesh_union_speedup_at_8": _dig(out, "multi_mesh", "union_in_one_handle", "speedup_at_8"),
        "device_bytes_live": _dig(out, "device_bytes", "libsmg_live"), "device_bytes_algorithmic": _dig(out, "device_bytes", "hierarchy_algorithmic"),
        "memory_lean_bytes": _dig(out, "device_bytes", "memory_lean_option", "handle_bytes"), "memory_lean_ms_per_step": _dig(out, "device_bytes", "memory_lean_option", "ms_per_step"),
        "c3_decimated_device_bytes_live": _dig(out, "c3_decimated", "device_bytes_live"),
        "c3_decimated_device_bytes_algorithmic": _dig(out, "c3_decimated", "hierarchy_algorithmic_bytes"),
        "setup_precompute_s": _dig(out, "setup_s", "precompute"),
    }
    line["env_overrides"] = list(out.get("env_overrides") or [])[:20]
    line["extra"] = {k: _num(v, 5) for k, v in extras.items() if v is not None}
    errs = [k for k, v in out.items() if isinstance(v, dict) and "error" in v]
    if errs:
        line["leg_errors"] = {k: str(out[k]["error"])[:120] for k in errs}
    line["extra_file"] = "bench_extra.json (every leg in full: same directory as bench.py, and gpurun_out/ when present)"
    s = json.dumps(line, allow_nan=False)
    if len(s) > LINE_HARD_CAP:       # never lose the line to verbosity again: drop the optional parts, keep the contract
        for k in ("extra_file", "leg_errors", "extra"):
            line.pop(k, None)
            if len(json.dumps(line, allow_nan=False)) <= LINE_HARD_CAP:
                break
    return line


def _json_safe(o):
    """the full record, strict-JSON safe (NaN / Inf -> null), numpy scalars / arrays -> Python"""
    if isinstance(o, dict):
        return {str(k): _json_safe(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_json_safe(v) for v in o]
    if isinstance(o, np.ndarray):
        return _json_safe(o.tolist())
    if isinstance(o, (np.bool_,)):
        return bool(o)
    if isinstance(o, np.integer):
        return int(o)
    if isinstance(o, (float, np.floating)):
        o = float(o)
        return o if (o == o and o not in (float("inf"), float("-inf"))) else None
    return o


def write_extra(out):
    """everything measured, for people: bench_extra.json beside bench.py and under gpurun_out/ (the directory gpurun merges back)"""
    full = json.dumps(_json_safe(out), allow_nan=False, indent=1)
    written = []
    dirs = (ROOT, os.path.join(ROOT, "gpurun_out"))
    if os.environ.get("SMG_BENCH_EXTRA_DIR"):      # tests: keep the working tree clean
        dirs = (os.environ["SMG_BENCH_EXTRA_DIR"],)
    for d in dirs:
        try:
            if d.endswith("gpurun_out"):
                os.makedirs(d, exist_ok=True)
            p = os.path.join(d, "bench_extra.json")
            with open(p, "w") as f:
                f.write(full)
            written.append(p)
        except OSError:
            pass
    return written


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="C3")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-c5", action="store_true", help="skip the out-of-cache roofline leg (C5, 4.19 M vertices)")
    ap.add_argument("--no-multi-mesh", action="store_true", help="skip the leg with M independent ogre.obj-size solves on one GPU")
    ap.add_argument("--no-c4", action="store_true", help="skip the C4 (ogre.obj, k = 64, column-sharded) leg")
    ap.add_argument("--no-reprecompute", action="store_true", help="skip the value-only re-precompute leg (dense inverse vs Schur-complement coarse solver)")
    ap.add_argument("--no-block3", action="store_true", help="skip the block (3-DOF) leg (C3 mesh, kron(S, C3) system)")
    ap.add_argument("--no-block3-scalar", action="store_true", help="block leg without the scalar-kernel comparison (its host precompute takes ~17 s)")
    ap.add_argument("--no-c3dec", action="store_true", help="skip the leg with the reference's own (mg_precompute, SSP-decimated) hierarchy on the C3 mesh")
    ap.add_argument("--no-c1", action="store_true", help="skip the C1 leg (03_mg_solver on bunny.obj / ogre.obj, GPU and oracle)")
    ap.add_argument("--no-c3k3", action="store_true", help="skip the C3 x 3 columns leg (the reference's own multi-RHS shape: 05_example_mean_curvature_flow)")
    ap.add_argument("--no-c3k64", action="store_true", help="skip the C3 x 64 columns column-sharded (strong scaling) leg")
    ap.add_argument("--repeats", type=int, default=9, help="the --steps iterations are timed this many times; the line reports the median repeat")
    ap.add_argument("--spmv-reps", type=int, default=500)
    ap.add_argument("--smoother", default="gs", choices=["gs", "jacobi", "hybrid", "chebyshev", "hybrid_chebyshev"],
                    help="gs (default): the reference's Gauss-Seidel on every level = the cycle the metric is defined on; hybrid_chebyshev: "
                         "GS on the levels with more than --jacobi-max-rows unknowns, Chebyshev-accelerated Jacobi (degree 3) "
                         "below -- as many cycles as GS everywhere, a third of the launches on the latency-bound levels (reported as an "
                         "extension in `smoothers` whatever is timed); hybrid: damped Jacobi instead; jacobi / chebyshev: everywhere")
    ap.add_argument("--omega", type=float, default=0.8)
    ap.add_argument("--jacobi-max-rows", type=int, default=300000)
    ap.add_argument("--precision", default="f64", choices=["f64", "mixed"],
                    help="f64 (default): the reference arithmetic; mixed: fp32 V-cycle inside the fp64 outer loop")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # SMG_BENCH_FORCE_SPLIT=1 exercises the multi-GPU code path (split-phase iteration + RCCL all-reduce) at world size 1
    force_split = os.environ.get("SMG_BENCH_FORCE_SPLIT") == "1"
    if world > 1 or force_split:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # "nccl" is RCCL.  SMG_BENCH_BACKEND=gloo lets the N > 1 code path be exercised on a box with fewer GPUs than ranks
        # (ranks then share devices: RCCL refuses that)
        dist.init_process_group(os.environ.get("SMG_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)
    ndev = max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank % ndev)
    dev = torch.device("cuda", local_rank % ndev)

    from surface_multigrid_code_amd import build as smg_build
    if rank == 0:
        smg_build.build(verbose=False)
    if world > 1:
        dist.barrier()
    import surface_multigrid_code_amd as smg
    from surface_multigrid_code_amd import mesh

    mg, A, Mb, Vf, Ff, label, t_host = build_workload(args.workload, smg, mesh)
    n = A.shape[0]
    # (the process's HIP runtime initialisation -- 0.1 - 0.2 s with torch's code objects registered -- is not part of the precompute: up before the clock starts)
    torch.zeros(1, device=dev)
    torch.cuda.synchronize()
    t0 = time.time()
    mg.precompute(A)
    t_pre = time.time() - t0
    stream = torch.cuda.Stream(device=dev)   # all libsmg work, the all-reduce and the timing events share it
    torch.cuda.set_stream(stream)
    mg.set_stream(stream.cuda_stream)

    # one RHS column per rank: M * g, g uniform(-1,1), seed 100 + rank (synthetic)
    rng = np.random.default_rng(100 + rank)
    g = rng.uniform(-1.0, 1.0, n)
    rhs_h = Mb @ g
    rhs = torch.from_numpy(rhs_h).to(dev)
    z0 = torch.zeros(n, dtype=torch.float64, device=dev)
    z = torch.empty(n, dtype=torch.float64, device=dev)
    sumsq = torch.zeros(1, dtype=torch.float64, device=dev)

    # multi-GPU: the 8-byte all-reduce runs on the solve's stream through RCCL directly (dist.StreamAllReduce: named enum constants,
    # version check, symmetric bootstrap and a known-answer reduction before it is trusted); torch's collective is the fallback when
    # that cannot be set up (SMG_BENCH_TORCH_ALLREDUCE=1 forces it)
    stream_ar = None
    if (world > 1 or force_split) and os.environ.get("SMG_BENCH_TORCH_ALLREDUCE", "0") != "1" and os.environ.get("SMG_BENCH_BACKEND", "nccl") == "nccl":
        from surface_multigrid_code_amd.dist import StreamAllReduce
        sar = StreamAllReduce(rank, world, stream.cuda_stream, device=dev)
        if sar.connect():       # collective: every rank gets the same verdict
            stream_ar = sar
        elif rank == 0:
            print("bench: direct RCCL all-reduce unavailable (%s), using torch.distributed" % sar.err, file=sys.stderr)
    # ---- multi-GPU pre-flight (N > 1 over RCCL): every rank on a device of its own and a communicator that really has N ranks -- or no line.
    # (A run whose ranks silently share a GPU, or whose collective fell back to something else, would print a plausible but meaningless
    # scaling point.)  SMG_BENCH_BACKEND=gloo / SMG_BENCH_ALLOW_SHARED=1: the CPU-side exercise of the N > 1 code path on fewer GPUs.
    preflight = None
    if world > 1:
        # identity of the device this rank computes on: node + device index (ranks that share a GPU share both); uuid / PCI ids ride along as
        # information only (not every runtime reports them, and a run must not be refused because they are missing or all alike)
        pr = torch.cuda.get_device_properties(dev)
        extra = "|".join(str(getattr(pr, a)) for a in ("uuid", "pci_bus_id", "pci_device_id") if getattr(pr, a, None) not in (None, ""))
        ident = "%s|device %d|%s" % (os.uname().nodename, local_rank % ndev, extra)
        idents = [None] * world
        dist.all_gather_object(idents, ident)
        comm_ranks = stream_ar.n_ranks() if stream_ar is not None else dist.get_world_size()
        infos = [None] * world
        dist.all_gather_object(infos, extra)
        preflight = {"rccl_comm_ranks": int(comm_ranks), "distinct_devices": len(set(idents)), "devices": idents, "device_info": infos, "visible_devices_per_rank": ndev, "backend": dist.get_backend(),
                     "allreduce": "RCCL communicator of libsmg's own (ncclCommCount)" if stream_ar is not None else "torch.distributed process group"}
        shared_ok = os.environ.get("SMG_BENCH_BACKEND", "nccl") != "nccl" or os.environ.get("SMG_BENCH_ALLOW_SHARED") == "1"
        # two ranks with the same key share a device for sure when they see several devices (same index) or report the same non-empty uuid / PCI
        # ids; a rank that sees ONE device and cannot name it (launcher-side masking, no uuid) proves nothing either way: RCCL itself refuses a
        # communicator with a duplicate GPU, so the communicator's rank count stays the hard criterion there
        evidence = ndev > 1 or any(infos)
        preflight["sharing_detectable"] = bool(evidence)
        if (preflight["rccl_comm_ranks"] != world or (evidence and preflight["distinct_devices"] != world)) and not shared_ok:
            if rank == 0:
                print("bench: multi-GPU pre-flight FAILED: --gpus %d needs %d RCCL ranks on %d distinct devices, found %s"
                      % (world, world, world, json.dumps(preflight)), file=sys.stderr)
            dist.barrier()
            dist.destroy_process_group()
            raise SystemExit(3)
    W, K = args.warmup, args.steps
    # tol = 0: the loop never converges, every step is a full outer iteration (residual + norm + break test + V(2,2) cycle) that
    # stores its results.  The device-side residual history holds 1024 entries (= the largest max_iter), and launches after the
    # last allowed iteration would run without storing anything -- so longer runs restart the solve (a gather launch and two
    # small copies, inside the timed region) every 1024 steps instead of silently timing such launches.
    HIS = 1024
    sm_kw = dict(smoother=args.smoother, omega=args.omega, jacobi_max_rows=args.jacobi_max_rows)
    opts = smg.SolveOpts(tol=0.0, max_iter=HIS, pre=2, post=2, precision=args.precision, **sm_kw)
    state = {"left": 0, "his": None}

    def step_block(n_it):
        if world == 1 and not force_split:
            mg.outer_iterations(n_it)
        else:
            # (a latency-hiding variant exists -- iter_cycle_speculative / iter_commit, dist.sharded_solve_overlapped --
            #  but its extra graph launch and copies cost as much as the 8-byte all-reduce it hides: measured 0.533 vs
            #  0.503 ms per iteration with the reduction forced on at world size 1)
            if os.environ.get("SMG_BENCH_SPECULATIVE", "0") == "1":
                for _ in range(n_it):             # the all-reduce hidden behind the cycle (smg.h: speculative / commit)
                    mg.iter_residual(sumsq.data_ptr())
                    work = dist.all_reduce(sumsq, async_op=True)
                    mg.iter_cycle_speculative()
                    work.wait()
                    mg.iter_commit(sumsq.data_ptr())
                return
            if stream_ar is not None:
                for _ in range(n_it):
                    mg.iter_residual(sumsq.data_ptr())
                    stream_ar(sumsq.data_ptr())   # RCCL, 8 bytes, on the solve's own stream: the Frobenius norm couples the columns
                    mg.iter_cycle(sumsq.data_ptr())
                return
            for _ in range(n_it):
                mg.iter_residual(sumsq.data_ptr())
                dist.all_reduce(sumsq)            # the same through torch.distributed (its own stream)
                mg.iter_cycle(sumsq.data_ptr())

    def run(n_it):
        while n_it > 0:
            if state["left"] == 0:
                if state["his"] is not None or mg_in_solve[0]:
                    _, state["his"] = mg.solve_end(z.data_ptr(), n, max_iter=HIS)
                mg.solve_begin(rhs.data_ptr(), n, z0.data_ptr(), n, 1, opts=opts)
                mg_in_solve[0] = True
                state["left"] = HIS
            m = min(n_it, state["left"])
            step_block(m)
            state["left"] -= m
            n_it -= m

    mg_in_solve = [False]
    run(W)
    NREP = max(1, args.repeats)
    rep_ms = timed_repeats(torch, dist, world, dev, stream, run, K, NREP)
    dt = float(np.median(rep_ms)) * 1e-3           # the median repeat of K steps
    conv, r_his = mg.solve_end(z.data_ptr(), n, max_iter=HIS)
    if state["his"] is not None and len(r_his) < 6:
        r_his = state["his"]   # the last segment was short: report the head of the previous one

    out = None
    if rank == 0:
        ms_step = 1e3 * dt / K
        vcyc_bytes = mg.vcycle_bytes(1, 2, 2)      # of the cycle just timed (the handle still carries its smoother selection)
        # ---- smoother comparison: the reference's Gauss-Seidel everywhere vs the timed configuration.  What counts is the time
        # to the tolerance: cycles needed to 1e-10 (the drop-in solve on resident vectors, device-side break test, adaptive host
        # polling) x the steady-state time of an outer iteration.
        def smoother_line(kw, ms_known=None):
            o = smg.SolveOpts(tol=1e-10, max_iter=100, pre=2, post=2, precision=args.precision, **kw)
            mg.solve_device(rhs.data_ptr(), z0.data_ptr(), z.data_ptr(), n, 1, opts=o)       # warm (graph capture)
            torch.cuda.synchronize()
            tw = time.perf_counter()
            cv, rh_ = mg.solve_device(rhs.data_ptr(), z0.data_ptr(), z.data_ptr(), n, 1, opts=o)
            wall = 1e3 * (time.perf_counter() - tw)
            ms = ms_known
            if ms is None:      # steady-state time of one outer iteration with this smoother (HIP events on the solve stream)
                oo = smg.SolveOpts(tol=0.0, max_iter=HIS, pre=2, post=2, precision=args.precision, **kw)
                mg.solve_begin(rhs.data_ptr(), n, z0.data_ptr(), n, 1, opts=oo)
                mg.outer_iterations(50)
                ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ea.record(stream)
                mg.outer_iterations(300)
                eb.record(stream)
                torch.cuda.synchronize()
                mg.solve_end(z.data_ptr(), n, max_iter=HIS)
                ms = ea.elapsed_time(eb) / 300
            cyc = len(rh_) - 1
            byt = mg.vcycle_bytes(1, 2, 2)      # bytes of THIS cycle (a Chebyshev relax(2) streams its level three times, Gauss-Seidel twice)
            return {"bytes_per_step": int(byt), "gbs": byt / (ms * 1e-3) / 1e9, "frac_of_hbm_peak": byt / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "smoother": kw["smoother"], "omega": kw.get("omega"), "jacobi_max_rows": kw.get("jacobi_max_rows"),
                    "jacobi_levels": [l for l in range(mg.n_levels - 1) if kw["smoother"] in ("jacobi", "chebyshev") or
                                      (kw["smoother"].startswith("hybrid") and mg.rows(l) <= kw["jacobi_max_rows"])],
                    "converged": bool(cv), "cycles_to_tol": cyc, "tol": 1e-10, "ms_per_step": ms, "v_cycles_per_s": 1e3 / ms,
                    "time_to_tol_ms": cyc * ms, "solve_wall_ms": wall, "final_residual": float(rh_[-1]) if len(rh_) else None}
        smoothers = None
        if world == 1 and not force_split:
            smoothers = {"timed": smoother_line(sm_kw, ms_step)}
            if args.smoother != "gs":
                smoothers["reference_gs"] = smoother_line(dict(smoother="gs", omega=args.omega, jacobi_max_rows=args.jacobi_max_rows))
            if args.smoother != "hybrid_chebyshev":   # libsmg's extension: same cycle count as GS, a third of the launches on the small levels
                smoothers["hybrid_chebyshev"] = smoother_line(dict(smoother="hybrid_chebyshev", omega=args.omega, jacobi_max_rows=300000))
            if args.smoother != "hybrid":
                smoothers["hybrid_damped_jacobi"] = smoother_line(dict(smoother="hybrid", omega=args.omega, jacobi_max_rows=100000))
            mg.set_smoother("gs")   # the kernel-level measurements below are of the reference smoother
        # ---- roofline of the fine-level SpMV kernel (k_sell<SELL_AX,1> on A_0), HIP events on the launch stream
        x = torch.from_numpy(rng.uniform(-1.0, 1.0, n)).to(dev)
        y = torch.empty_like(x)
        R = args.spmv_reps
        spmv_us, spmv_all = median_us(torch, stream, lambda: mg.raw_spmv(0, 0, x.data_ptr(), None, y.data_ptr()), max(R // 5, 50), 5, warm=20)
        spmv_bytes = mg.spmv_bytes(0, 1)
        spmv_gbs = spmv_bytes / (spmv_us * 1e-6) / 1e9
        # ---- one full Gauss-Seidel sweep on level 0 (all colours), same byte model + b read
        bvec = torch.from_numpy(rhs_h).to(dev)
        u = torch.zeros_like(bvec)
        gs_all = [mg.bench_relax(0, 1, 10, 8) / 10.0 for _ in range(5)]      # graph replay (smg_bench_relax): no host launch cost in the figure
        gs_us = float(np.median(gs_all))
        nnz0 = A.nnz
        gs_bytes = 12 * nnz0 + 4 * (n + 1) + 24 * n
        # ---- per-scope timing of the V-cycle (profc mirror; eager launches with hipEvents)
        mg.prof_enable(True)
        mg.solve_begin(rhs.data_ptr(), n, z0.data_ptr(), n, 1, opts=smg.SolveOpts(tol=0.0, max_iter=8, **sm_kw))
        mg.outer_iterations(8)
        mg.solve_end(z.data_ptr(), n, max_iter=8)
        prof = mg.prof_table()
        mg.prof_enable(False)
        st = mg.sell_stats(0, "A")
        # HBM traffic of the same kernel from the committed rocprofv3 PMC passes (bench.py cannot profile itself): reported only while
        # the committed figure was measured on the kernel sources that are being timed (hash stamped by tools/make_traffic.py)
        traffic, traffic_note = committed_traffic(args.workload)
        ref = smoothers["timed"] if (smoothers and args.smoother == "gs") else (smoothers or {}).get("reference_gs")
        out = {
            "metric": "V-cycles/sec + fine-level SpMV GB/s (% HBM peak), 1M-vert mesh fp64",
            "value": world * K / dt, "unit": "V-cycles/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": ms_step, "multi_gpu_preflight": preflight,
            "timing": {"repeats": NREP, "what": "each repeat = --steps outer iterations between barrier + synchronize, HIP events on the solve stream, max over ranks; ms_per_step / value = the median repeat",
                       "ms_per_step_min": min(rep_ms) / K, "ms_per_step_median": float(np.median(rep_ms)) / K, "ms_per_step_max": max(rep_ms) / K},
            # the reference's cycle -- V(2,2), Gauss-Seidel on every level (src/mg_VCycle.cpp) -- whatever --smoother timed
            "reference_cycle": {"v_cycles_per_s": ref["v_cycles_per_s"], "ms_per_step": ref["ms_per_step"], "cycles_to_1e-10": ref["cycles_to_tol"],
                                "bytes_per_step": ref["bytes_per_step"], "frac_of_hbm_peak": ref["frac_of_hbm_peak"]} if ref else None,
            "higher_is_better": True, "scaling": "weak",
            "scaling_note": "top-level value: one right-hand-side column per GPU, hierarchy replicated (fixed work per GPU as N grows); the c3_k64_sharded / c4_k64_sharded legs split a FIXED 64-column job over the ranks and say 'strong' themselves",
            "vs_baseline": None, "host": host_info(),
            # every SMG_* variable set in this run's environment (A/B knobs: DESIGN.md section 11): [] = the documented defaults
            "env_overrides": sorted(k_ for k_ in os.environ if k_.startswith("SMG_") and k_ not in ("SMG_BENCH_EXTRA_DIR",)),
            "dtype": "f64" if args.precision == "f64" else "f64 outer loop + f32 V-cycle (mixed)", "data": "synthetic",
            "config": {"workload": label, "n_verts": n, "nnz": int(nnz0), "levels": mg.n_levels,
                       "level_rows": [mg.rows(l) for l in range(mg.n_levels)],
                       "colors": [len(mg.colors(l)) - 1 for l in range(mg.n_levels - 1)],
                       "rhs_columns_per_gpu": 1,
                       "cycle": {"gs": "V(2,2), multi-colour Gauss-Seidel on every level (the reference's relax()), dense coarsest solve",
                                 "hybrid": "V(2,2), multi-colour Gauss-Seidel on levels > %d rows, damped Jacobi (omega %.2f) below, dense coarsest solve" % (args.jacobi_max_rows, args.omega),
                                 "jacobi": "V(2,2), damped Jacobi (omega %.2f) on every level, dense coarsest solve" % args.omega,
                                 "chebyshev": "V(2,2), Chebyshev-Jacobi (degree 3) on every level, dense coarsest solve",
                                 "hybrid_chebyshev": "V(2,2), multi-colour Gauss-Seidel on levels > %d rows, Chebyshev-Jacobi (degree 3) below, dense coarsest solve" % args.jacobi_max_rows}[args.smoother],
                       "smoother": args.smoother,
                       "relax_launches": "one launch per colour and sweep; on Gauss-Seidel levels of 2 048 - 100 000 rows the whole relax(2) is ONE launch (overlapped tiling, bit-identical; SMG_TILED=0 switches it off)",
                       "parallelism": "1 RHS column per GPU, hierarchy replicated, all-reduce of residual sumsq" if world > 1 else "single GPU",
                       "allreduce": ("RCCL on the solve stream (dist.StreamAllReduce)" if stream_ar is not None else "torch.distributed") if (world > 1 or force_split) else None},
            "roofline": {"kernel": "k_sell<SELL_AX,1> (fine-level y = A x)", "bound": "hbm",
                         "achieved": spmv_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": spmv_gbs / HBM_PEAK_GBS,
                         # the committed PMC figure of this kernel at this grid (rocprofv3 cannot run inside the benchmark): fabric-side
                         # requests, which do not tell Infinity-Cache hits from HBM reads
                         "traffic": traffic, "traffic_kind": "committed rocprofv3 PMC figure for these kernel sources (hash-checked), not a measurement of this run" if traffic is not None else None,
                         "traffic_source": traffic_note,
                         "bytes_per_launch": int(spmv_bytes), "us_per_launch": spmv_us, "us_per_launch_repeats": spmv_all,
                         "timing": "median of 5 HIP-event-timed loops on the launch stream",
                         "sell_padding": st["padded"] / max(st["stored"], 1) - 1.0,
                         # what the launch streams (SELL slots incl. padding + x + y): below 256 MiB the matrix survives in the
                         # Infinity Cache between back-to-back launches, so `achieved` may exceed what HBM alone sustains (~6.3 TB/s);
                         # the out-of-cache figure of the same kernel is `roofline_c5`
                         "working_set_bytes": int(12 * st["padded"] + 16 * n),
                         "infinity_cache_resident": bool(12 * st["padded"] + 16 * n < INFINITY_CACHE_BYTES),
                         "frac_of_sustainable": spmv_gbs / HBM_SUSTAINABLE_GBS},
            "roofline_gs_sweep": {"kernel": "k_sell<SELL_GS,1> x colours (one fine-level sweep)", "bound": "hbm",
                                  "achieved": gs_bytes / (gs_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": gs_bytes / (gs_us * 1e-6) / 1e9 / HBM_PEAK_GBS, "us_per_sweep": gs_us, "us_per_sweep_repeats": gs_all,
                                  "timing": "median of 5 graph-replayed loops of 8 x relax(10) = 80 sweeps (smg_bench_relax, HIP events on the launch stream)",
                                  "bytes_per_sweep": int(gs_bytes)},
            "roofline_vcycle": {"bound": "hbm", "cycle": args.smoother, "bytes_per_step": int(vcyc_bytes),
                                "achieved": vcyc_bytes / (ms_step * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": vcyc_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "smoothers": smoothers,
            "time_to_tol_ms": smoothers["timed"]["time_to_tol_ms"] if smoothers else None,
            "cycles_to_tol": smoothers["timed"]["cycles_to_tol"] if smoothers else None,
            "profc": {k: {"count": v[0], "ms_total": v[1]} for k, v in prof.items()},
            "residual_history_head": [float(v) for v in r_his[:6]],
            "setup_s": {"mesh_hierarchy_host": t_host, "precompute": t_pre, "precompute_known": precompute_known_s(smg, mesh, mg, Vf, Ff) if args.workload == "C3" else None,
                        "precompute_note": "first (pattern-changing) smg_precompute of the process, HIP runtime already initialised; libsmg's own code object is loaded inside it"},
            # memory budget: everything libsmg holds in HBM for this workload (operators in SELL incl. the fixed panel pitch, A^T images of the
            # Galerkin levels, dense coarse inverse, work vectors, graphs' buffers) against the algorithmic size of the hierarchy
            "device_bytes": {"libsmg_live": int(smg._lib.load().smg_device_bytes_live()), "hierarchy_algorithmic": algorithmic_bytes(mg),
                             "by_purpose": {k_: v for k_, v in sorted(mg.device_bytes().items(), key=lambda kv: -kv[1]) if v >= (1 << 20)},
                             "note": "by_purpose: what the C3 handle holds, entries >= 1 MiB (smg_debug_device_bytes): level0.A_sell carries the fixed panel pitch (12 columns of room for 7 used: "
                                     "addressing without a table, DESIGN.md section 2), the coarse inverse is kept whole for the k > 1 kernels, Galerkin levels keep an A^T image (the reference's sweep walks columns)",
                             "hbm_capacity": 288 * 10 ** 9},
        }
        out["device_bytes"]["ratio_to_algorithmic"] = out["device_bytes"]["libsmg_live"] / max(out["device_bytes"]["hierarchy_algorithmic"], 1)
        if world == 1 and not force_split and args.workload == "C3":
            try:
                out["device_bytes"]["memory_lean_option"] = memory_lean_leg(smg, mg, A, rhs, z0, z, n, torch, stream)
            except Exception as e:
                out["device_bytes"]["memory_lean_option"] = {"error": repr(e)}
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(mg, A, rhs_h)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
            try:
                out["cpu_allcore"] = cpu_allcore(mg, A, rhs_h)
            except Exception as e:  # the comparator is informational: never lose the bench line over it
                out["cpu_allcore"] = {"error": str(e)}
    # ---- the same kernels beyond the Infinity Cache (BASELINE config C5), rank 0 at N = 1 only
    if rank == 0 and world == 1 and not args.no_c5 and args.workload == "C3":
        try:
            try:
                del x, y, bvec, u
            except NameError:
                pass
            out["roofline_c5"] = roofline_c5(smg, mesh, torch, dev, stream)
        except Exception as e:
            out["roofline_c5"] = {"error": repr(e)}
    # ---- the reference's own multi-RHS shape: three coordinate columns (05_example_mean_curvature_flow), rank 0 at N = 1 only
    if rank == 0 and world == 1 and not args.no_c3k3 and args.workload == "C3":
        try:
            out["c3_k3"] = c3_k3_leg(smg, mg, A, Mb, Vf, n, torch, dev, stream, with_cpu=not args.no_cpu)
        except Exception as e:
            out["c3_k3"] = {"error": repr(e)}
    # ---- a column-sharded job that shards usefully: the C3 mesh x 64 columns (strong scaling), through smg_solve_sharded
    if not args.no_c3k64 and args.workload == "C3":
        try:
            if rank == 0 and world == 1:
                try:
                    del x, y, bvec, u
                except NameError:
                    pass
            c3k = c3_k64_sharded(smg, mg, Mb, n, torch, dist, rank, world, dev, stream, stream_ar, steps=max(1, min(args.steps, 20)), warmup=3)
        except Exception as e:
            c3k = {"error": repr(e)}
            if world > 1:
                raise
        if rank == 0:
            out["c3_k64_sharded"] = c3k
    # ---- row f-2: the value-only re-precompute of the time-stepping callers, rank 0 at N = 1 only (last use of the C3 handle)
    if rank == 0 and world == 1 and not args.no_reprecompute and args.workload == "C3":
        try:
            out["reprecompute"] = reprecompute_leg(smg, mg, A, torch)
        except Exception as e:
            out["reprecompute"] = {"error": repr(e)}
    # ---- the same mesh under the reference's own kind of hierarchy (mg_precompute: SSP decimation), rank 0 at N = 1 only
    if rank == 0 and world == 1 and not args.no_c3dec and args.workload == "C3":
        try:
            out["c3_decimated"] = c3_decimated_leg(smg, mesh, torch, dev, stream, out["reference_cycle"]["ms_per_step"] if out.get("reference_cycle") else out["ms_per_step"],
                                                   out["reference_cycle"]["bytes_per_step"] if out.get("reference_cycle") else None)
        except Exception as e:
            out["c3_decimated"] = {"error": repr(e)}
    # ---- BASELINE config C1: the reference's 03_mg_solver on its own meshes, GPU and oracle side by side, rank 0 at N = 1 only
    if rank == 0 and world == 1 and not args.no_c1:
        try:
            out["c1_03_mg_solver"] = c1_leg(smg, mesh, torch, dev, stream)
        except Exception as e:
            out["c1_03_mg_solver"] = {"error": repr(e)}
    # ---- block (3-DOF) kernels, rank 0 at N = 1 only
    if rank == 0 and world == 1 and not args.no_block3 and args.workload == "C3":
        try:
            out["block3_c3"] = block3_leg(smg, mesh, torch, with_scalar=not args.no_block3_scalar)
            torch.cuda.set_stream(stream)
        except Exception as e:
            out["block3_c3"] = {"error": repr(e)}
    # ---- independent meshes on one GPU (north_star: "independent RHS columns / independent meshes shard"), rank 0 at N = 1 only
    if rank == 0 and world == 1 and not args.no_multi_mesh:
        try:
            out["multi_mesh"] = multi_mesh_leg(smg, mesh, torch, dev)
            torch.cuda.set_stream(stream)
        except Exception as e:
            out["multi_mesh"] = {"error": repr(e)}
    # ---- BASELINE config C4: k = 64 columns sharded over the ranks (strong scaling; N = 1 is the curve's first point)
    if not args.no_c4:
        try:
            c4 = c4_k64_sharded(smg, mesh, torch, dist, rank, world, dev, stream, stream_ar, sm_kw, steps=max(args.steps, 1), warmup=max(args.warmup, 0))
        except Exception as e:   # never lose the bench line over the secondary measurement
            c4 = {"error": repr(e)}
            if world > 1:
                raise
        if rank == 0:
            out["c4_k64_sharded"] = c4
    # tear the process group down BEFORE the line is printed: RCCL may write to stdout when a communicator is created or
    # destroyed, and the JSON must be the last line
    if world > 1 or force_split:
        torch.cuda.synchronize()
        if stream_ar is not None:
            stream_ar.close()
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        try:   # C-level stdio of the libraries (RCCL prints its path there) is block-buffered on a pipe: drain it first
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        paths = write_extra(out)
        print("bench: full record (%d legs) -> %s" % (len(out), ", ".join(paths) or "nowhere (not writable)"), file=sys.stderr)
        print(json.dumps(compact_line(out), allow_nan=False), flush=True)


if __name__ == "__main__":
    main()
