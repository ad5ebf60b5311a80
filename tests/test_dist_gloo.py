"""CPU, world_size 2, gloo: the column-sharded solve loop (surface_multigrid_code_amd/dist.py).  The engine here is
an oracle-backed test double (tests may use the oracle); the loop, the column partition and the all-reduce placement
are the product code that also drives the GPU engine over RCCL."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


class OracleEngine:
    def __init__(self, orc, rhs, z0, tol):
        self.orc, self.rhs, self.z, self.tol = orc, rhs, z0.copy(order="F"), tol
        self.done, self.r_his = False, []

    def begin(self):
        pass

    def residual_sumsq(self):
        r = self.rhs - self.orc.A(0, self.z)
        return torch.tensor([float((r * r).sum())], dtype=torch.float64)

    def cycle(self, t):
        if self.done:
            return
        r = float(np.sqrt(t.item()))
        self.r_his.append(r)
        if r < self.tol:
            self.done = True
            return
        self.z = self.orc.vcycle(self.rhs, self.z)

    def cycle_speculative(self):
        self.saved = self.z.copy(order="F")
        if not self.done:
            self.z = self.orc.vcycle(self.rhs, self.z)

    def commit(self, t):
        if self.done:
            return
        r = float(np.sqrt(t.item()))
        self.r_his.append(r)
        if r < self.tol:
            self.done = True
            self.z = self.saved

    def poll(self):
        return self.done, len(self.r_his)

    def end(self):
        return (self.r_his[-1] <= self.tol), self.z, np.array(self.r_his)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.oracle import OracleMG
    from problems import subdiv_problem
    from surface_multigrid_code_amd.dist import column_range, sharded_solve
    k, tol = 6, 5e-7
    p = subdiv_problem(kind="mcf", k=k, n_sub=1)
    orc = OracleMG(p["Ps"])
    orc.precompute(p["A"])
    lo, hi = column_range(k, rank, world)
    eng = OracleEngine(orc, p["RHS"][:, lo:hi], p["z0"][:, lo:hi], tol)
    conv, z, rh = sharded_solve(eng, 20, lambda t: dist.all_reduce(t), check_every=2)
    # the latency-hiding loop (async all-reduce overlapped with the speculative cycle) must give the same answer
    from surface_multigrid_code_amd.dist import sharded_solve_overlapped
    eng2 = OracleEngine(orc, p["RHS"][:, lo:hi], p["z0"][:, lo:hi], tol)
    conv2, z2, rh2 = sharded_solve_overlapped(eng2, 20, lambda t: dist.all_reduce(t, async_op=True), check_every=3)
    assert conv2 == conv and np.array_equal(rh2, rh) and np.array_equal(z2, z)
    # check_every = 0 is the library's "adaptive" (SolveOpts' default): the model loop must treat it as 1, not spin forever on empty chunks
    eng3 = OracleEngine(orc, p["RHS"][:, lo:hi], p["z0"][:, lo:hi], tol)
    conv3, z3, rh3 = sharded_solve(eng3, 20, lambda t: dist.all_reduce(t), check_every=0)
    assert conv3 == conv and np.array_equal(rh3, rh) and np.array_equal(z3, z)
    # unsharded reference on every rank
    conv_ref, z_ref, rh_ref = orc.solve(p["RHS"], p["z0"], tol=tol, max_iter=20)
    zs = [torch.zeros(z_ref.shape[0], hi2 - lo2, dtype=torch.float64) for (lo2, hi2) in (column_range(k, r, world) for r in range(world))]
    dist.all_gather(zs, torch.from_numpy(np.ascontiguousarray(z)))
    zfull = np.concatenate([t.numpy() for t in zs], axis=1)
    ok = (conv == conv_ref and len(rh) == len(rh_ref) and np.allclose(rh, rh_ref, rtol=1e-12, atol=0)
          and np.array_equal(zfull, z_ref))
    # fewer columns than ranks: the rank without a column contributes 0 and follows the others' decision
    from surface_multigrid_code_amd.dist import EmptyEngine
    lo1, hi1 = column_range(1, rank, world)
    e1 = OracleEngine(orc, p["RHS"][:, :1], p["z0"][:, :1], tol) if hi1 > lo1 else EmptyEngine(tol)
    c1, z1, rh1 = sharded_solve(e1, 20, lambda t: dist.all_reduce(t))
    c1r, z1r, rh1r = orc.solve(p["RHS"][:, :1], p["z0"][:, :1], tol=tol, max_iter=20)
    ok = ok and c1 == c1r and len(rh1) == len(rh1r) and np.allclose(rh1, rh1r, rtol=1e-12, atol=0)
    if hi1 > lo1:
        ok = ok and np.array_equal(z1, z1r)
    q.put((rank, bool(ok), len(rh), float(rh[-1])))
    dist.barrier()
    dist.destroy_process_group()


def test_column_sharded_solve_world2():
    from oracle import oracle
    oracle.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = [q.get(timeout=300) for _ in procs]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    assert all(r[1] for r in res), res
    assert res[0][2] == res[1][2] and res[0][3] == res[1][3]      # identical history on both ranks


def test_column_range_partitions():
    from surface_multigrid_code_amd.dist import column_range
    for k in (1, 3, 8, 64, 65):
        for world in (1, 2, 4, 8):
            spans = [column_range(k, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == k
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
