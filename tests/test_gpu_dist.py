"""GPU (-m gpu): the column-sharded solve driven by TWO processes (gloo rendezvous, both ranks on GPU 0 -- RCCL refuses ranks that
share a device, the driver runs the real 8-GPU RCCL job): surface_multigrid_code_amd.dist.GpuEngine + sharded_solve and its
latency-hiding form against the fused k-column smg_solve.

Per column the sparse kernels and the dense coarse solve (k_dense_gemv_add for 2 <= k < 8) do the same arithmetic whatever the
number of columns in the block, so the sharded iterate is BIT-IDENTICAL to the fused one; only the residual norm is summed in
another order (per-rank partial sums, then the all-reduce): r_his agrees to 1e-12 relative."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, port, q, smoother):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, HERE)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        import surface_multigrid_code_amd as smg
        from surface_multigrid_code_amd.dist import EmptyEngine, GpuEngine, HostReduce, column_range, sharded_solve, sharded_solve_native, sharded_solve_overlapped
        from problems import subdiv_problem
        k, tol = 6, 5e-7
        p = subdiv_problem(kind="mcf", k=k, n_sub=2)
        mg = smg.Hierarchy.from_prolongs(p["Ps"])
        mg.precompute(p["A"])
        n = mg.rows(0)
        stream = torch.cuda.Stream(device=dev)
        kw = dict(smoother=smoother, jacobi_max_rows=mg.rows(1))
        with torch.cuda.stream(stream):
            mg.set_stream(stream.cuda_stream)
            lo, hi = column_range(k, rank, world)
            rhs = torch.from_numpy(np.ascontiguousarray(p["RHS"][:, lo:hi].T)).to(dev)
            z0 = torch.from_numpy(np.ascontiguousarray(p["z0"][:, lo:hi].T)).to(dev)

            def allreduce(t):
                dist.all_reduce(t)          # gloo on a CUDA tensor: ordered with the current (= the solve's) stream

            outs = []
            for form in ("plain", "speculative", "eager"):
                opts = smg.SolveOpts(tol=tol, max_iter=30, use_graph=0 if form == "eager" else 1, **kw)
                eng = GpuEngine(mg, rhs, z0, None, opts)
                if form == "speculative":
                    conv, z, rh = sharded_solve_overlapped(eng, 30, lambda t: dist.all_reduce(t, async_op=True), check_every=3)
                else:
                    conv, z, rh = sharded_solve(eng, 30, allreduce, check_every=2 if form == "plain" else 1)
                stream.synchronize()
                outs.append((conv, z.cpu().numpy().T.copy(), np.asarray(rh)))
            # the library's own loop (smg_solve_sharded: residual graph -> the caller's reduction -> cycle graph, all in C++), with a
            # host closure over gloo; adaptive polling (check_every = 0) and every-other-iteration polling
            red = HostReduce()
            for ce in (0, 2):
                conv, z, rh = sharded_solve_native(mg, rhs, z0, red, None, smg.SolveOpts(tol=tol, max_iter=30, check_every=ce, **kw))
                stream.synchronize()
                outs.append((conv, z.cpu().numpy().T.copy(), np.asarray(rh)))
            native_calls = red.calls
            # a rank without columns (k < world) still takes part in every reduction: k = 1 on two ranks
            lo1, hi1 = column_range(1, rank, world)
            if hi1 > lo1:
                r1 = torch.from_numpy(np.ascontiguousarray(p["RHS"][:, :1].T)).to(dev)
                s1 = torch.from_numpy(np.ascontiguousarray(p["z0"][:, :1].T)).to(dev)
                e1 = GpuEngine(mg, r1, s1, None, smg.SolveOpts(tol=tol, max_iter=30, **kw))
            else:
                e1 = EmptyEngine(tol, dev)
            c1, z1, rh1 = sharded_solve(e1, 30, allreduce)
            stream.synchronize()
            # ... and through smg_solve_sharded with k_local = 0 on the rank that owns nothing
            if hi1 > lo1:
                c1n, z1n, rh1n = sharded_solve_native(mg, r1, s1, red, None, smg.SolveOpts(tol=tol, max_iter=30, **kw))
            else:
                c1n, z1n, rh1n = sharded_solve_native(mg, None, None, red, None, smg.SolveOpts(tol=tol, max_iter=30, **kw))
            stream.synchronize()
            # the fused k-column solve (rank 0's reference for everything)
            conv_f, z_f, rh_f = mg.solve(p["RHS"], p["z0"], None, smg.SolveOpts(tol=tol, max_iter=30, **kw))
            conv_1, z_1, rh_1 = mg.solve(p["RHS"][:, :1], p["z0"][:, :1], None, smg.SolveOpts(tol=tol, max_iter=30, **kw))
        ok = True
        msg = ""
        for form, (conv, z, rh) in zip(("plain", "speculative", "eager", "native", "native-poll2"), outs):
            good = (conv == conv_f and len(rh) == len(rh_f) and np.allclose(rh, rh_f, rtol=1e-12, atol=0)
                    and np.array_equal(z, z_f[:, lo:hi]))
            if not good:
                ok = False
                msg += "%s: conv %s/%s its %d/%d zdiff %.3e; " % (form, conv, conv_f, len(rh), len(rh_f),
                                                                  abs(z - z_f[:, lo:hi]).max() if z.shape == z_f[:, lo:hi].shape else -1)
        if not (np.array_equal(outs[0][2], outs[1][2]) and np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][1], outs[2][1])):
            ok = False
            msg += "the three forms of the loop disagree; "
        if not (np.array_equal(outs[3][1], outs[0][1]) and np.array_equal(outs[3][2], outs[0][2]) and np.array_equal(outs[4][1], outs[0][1])
                and np.array_equal(outs[4][2], outs[0][2])):
            ok = False
            msg += "smg_solve_sharded disagrees with the split-phase loop; "
        if not (c1n == conv_1 and len(rh1n) == len(rh_1) and np.allclose(rh1n, rh_1, rtol=1e-12, atol=0) and np.array_equal(rh1n, rh1)
                and (z1n is None or np.array_equal(z1n.cpu().numpy().T, z_1))):
            ok = False
            msg += "smg_solve_sharded, k=1 on two ranks: %s its %d/%d; " % (c1n, len(rh1n), len(rh_1))
        if native_calls < 2 * len(outs[0][2]):
            ok = False
            msg += "the reduction was called %d times for %d loop entries; " % (native_calls, 2 * len(outs[0][2]))
        if not (c1 == conv_1 and len(rh1) == len(rh_1) and np.allclose(rh1, rh_1, rtol=1e-12, atol=0)):
            ok = False
            msg += "k=1 on two ranks: %s its %d/%d; " % (c1, len(rh1), len(rh_1))
        q.put((rank, ok, msg, len(outs[0][2]), float(outs[0][2][-1])))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:   # surface the failure instead of a queue timeout
        import traceback
        q.put((rank, False, "exception: " + traceback.format_exc()[-1500:], 0, 0.0))


@pytest.mark.parametrize("smoother", ["gs", "hybrid"])
def test_two_ranks_drive_the_gpu_engine(smg_mod, smoother):
    assert smg_mod._lib.load().smg_device_count() > 0, "GPU tests need a HIP device (no CPU fallback exists)"
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, smoother)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = [q.get(timeout=600) for _ in procs]
    for pr in procs:
        pr.join(timeout=120)
    assert all(r[1] for r in res), res
    assert res[0][3] == res[1][3] and res[0][4] == res[1][4]      # identical history on both ranks


def test_handle_lives_on_the_device_that_was_current_at_precompute(smg_mod):
    """One process, two GPUs: a handle precomputed with device 1 current keeps everything (including what the precompute's worker
    threads upload -- a std::thread starts on device 0) on device 1 and can be driven while another device is current; results are
    those of a device-0 handle, bit for bit.  Needs >= 2 visible GPUs (skipped on the single-GPU box)."""
    import torch
    if smg_mod._lib.load().smg_device_count() < 2 or torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    sys.path.insert(0, HERE)
    from problems import subdiv_problem
    smg = smg_mod
    p = subdiv_problem(kind="poisson", k=2, n_sub=2)
    o = smg.SolveOpts(tol=1e-9, max_iter=40)
    torch.cuda.set_device(0)
    mg0 = smg.Hierarchy.from_prolongs(p["Ps"]); mg0.precompute(p["A"], p["known"])
    ref = mg0.solve(p["RHS"], p["z0"], p["known_val"], o)
    torch.cuda.set_device(1)
    mg1 = smg.Hierarchy.from_prolongs(p["Ps"]); mg1.precompute(p["A"], p["known"])
    a = mg1.solve(p["RHS"], p["z0"], p["known_val"], o)
    torch.cuda.set_device(0)                                   # the caller moves on; the handle stays on device 1
    b = mg1.solve(p["RHS"], p["z0"], p["known_val"], o)
    mg1.precompute(p["A"], p["known"])                         # value-only re-precompute path, still from device 0's thread state
    c = mg1.solve(p["RHS"], p["z0"], p["known_val"], o)
    assert torch.cuda.current_device() == 0
    for r in (a, b, c):
        assert r[0] and np.array_equal(r[1], ref[1]) and np.array_equal(r[2], ref[2])
    free0 = torch.cuda.mem_get_info(0)[0]
    del mg1
    assert torch.cuda.mem_get_info(0)[0] <= free0 + (64 << 20)   # nothing of mg1 had been living on device 0


def _bench_n_ranks_on_one_gpu(tmp_path, n, port):
    import json
    import subprocess
    env = dict(os.environ, SMG_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", SMG_BENCH_EXTRA_DIR=str(tmp_path))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "6", "--warmup", "2", "--no-c5", "--no-cpu"],
                         env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                      # rank 0 prints ONE JSON line ...
    assert out.stdout.rstrip().splitlines()[-1] == lines[0] and len(lines[0]) < 8000          # ... the last one, compact (the driver parses it)
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["steps"] == 6 and d["warmup"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["allreduce"] == "torch.distributed"
    pf = d["multi_gpu_preflight"]
    assert pf["rccl_comm_ranks"] == n and pf["backend"] == "gloo" and pf["distinct_devices"] == 1      # n ranks in the communicator, ONE device: gloo only
    assert d["extra"]["c4_k64_ms_per_step"] > 0 and d["extra"]["c3_k64_ms_per_step"] > 0 and "leg_errors" not in d
    full = json.load(open(os.path.join(str(tmp_path), "bench_extra.json")))
    assert full["value"] == pytest.approx(d["value"], rel=1e-5)
    return d, full


def test_bench_runs_its_multi_rank_path_with_two_ranks(tmp_path):
    """bench.py --gpus 2 launched the way the driver launches it (torch.distributed.run, one rank per GPU), here with both ranks on GPU 0
    over gloo (SMG_BENCH_BACKEND=gloo: RCCL refuses two ranks on one device): the split-phase iteration with the all-reduce between
    its halves, the weak-scaling `value` (one C3 column per rank) and the C4 leg (64 columns split by column_range) -- every rank records
    the same residual history and the job converges in the cycles the unsharded solve needs."""
    d, full = _bench_n_ranks_on_one_gpu(tmp_path, 2, 29533)
    c4 = full["c4_k64_sharded"]
    assert c4["scaling"] == "strong" and c4["solve"]["converged"] and c4["solve"]["same_history_on_all_ranks"]
    assert c4["smoother"] == "gs" and c4["solve"]["cycles"] == 15 and c4["solve"]["final_residual"] < 5e-7     # the reference's cycle (16 with the Chebyshev hybrid)
    # the leg that shards usefully (C3 mesh x 64 columns) runs through the library's own loop, smg_solve_sharded, with a host closure here
    c3 = full["c3_k64_sharded"]
    assert c3["scaling"] == "strong" and c3["columns_per_gpu"] == 32 and c3["solve"]["converged"] and c3["solve"]["same_history_on_all_ranks"]
    assert "smg_solve_sharded" in c3["loop"] and c3["ms_per_step"] > 0


def test_bench_dry_run_of_the_eight_rank_launch(tmp_path):
    """The driver's 8-GPU launch line, dry: 8 ranks over gloo sharing GPU 0 (no 8-GPU box has ever been reachable, so the first real run must not
    fail on plumbing).  BASELINE config C4's shape: 64 columns, 8 per rank; C3 x 64 columns likewise; one C3 column per rank for `value`."""
    d, full = _bench_n_ranks_on_one_gpu(tmp_path, 8, 29541)
    c4, c3 = full["c4_k64_sharded"], full["c3_k64_sharded"]
    assert c4["columns_per_gpu"] == 8 and c4["solve"]["converged"] and c4["solve"]["same_history_on_all_ranks"] and c4["solve"]["cycles"] == 15
    assert c3["columns_per_gpu"] == 8 and c3["solve"]["converged"] and c3["solve"]["same_history_on_all_ranks"]
