"""Column-sharded multi-GPU solve: one process per GPU, the hierarchy replicated, every rank
owns a contiguous block of the right-hand-side columns.  The V-cycle over one mesh stays on one GPU; the ONLY
communication is the all-reduce of the residual sum of squares, because the reference's stopping test is one
Frobenius norm over all columns jointly (src/min_quad_with_fixed_mg.cpp:110, :332).  backend "nccl" is RCCL.

The loop below is engine-agnostic: the GPU engine is `GpuEngine` (libsmg split-phase API, decision taken on the
device, no host sync per iteration); the CPU tests drive the same loop with an oracle-backed engine over gloo.
"""
import numpy as np


def column_range(k, rank, world):
    """Contiguous block of columns owned by `rank`: [g*k/world, (g+1)*k/world)."""
    lo = (rank * k) // world
    hi = ((rank + 1) * k) // world
    return lo, hi


class GpuEngine:
    """libsmg engine: everything stays in HBM; `sumsq` is a 1-element float64 CUDA tensor."""

    def __init__(self, mg, rhs, z0, known_val=None, opts=None):
        import torch
        self.torch = torch
        self.mg = mg
        self.rhs, self.z0, self.kv = rhs, z0, known_val      # column-major device tensors: shape (k, n) contiguous
        self.k, self.n = rhs.shape
        self.opts = opts
        self.sumsq = torch.zeros(1, dtype=torch.float64, device=rhs.device)
        self.z = torch.empty_like(z0)

    def begin(self):
        kvp = self.kv.data_ptr() if self.kv is not None else None
        ldkv = self.kv.shape[1] if self.kv is not None else 0
        self.mg.solve_begin(self.rhs.data_ptr(), self.n, self.z0.data_ptr(), self.n, self.k, kvp, ldkv, self.opts)

    def residual_sumsq(self):
        self.mg.iter_residual(self.sumsq.data_ptr())
        return self.sumsq

    def cycle(self, sumsq):
        self.mg.iter_cycle(sumsq.data_ptr())

    # latency-hiding pair: the V-cycle is enqueued before the reduction result is needed
    def cycle_speculative(self):
        self.mg.iter_cycle_speculative()

    def commit(self, sumsq):
        self.mg.iter_commit(sumsq.data_ptr())

    def poll(self):
        return self.mg.poll()

    def end(self):
        conv, r_his = self.mg.solve_end(self.z.data_ptr(), self.n)
        return conv, self.z, r_his


def sharded_solve_overlapped(engine, max_iter, all_reduce_async, check_every=1):
    """Same loop with the all-reduce hidden behind the V-cycle: `all_reduce_async(t)` starts the reduction and returns a
    handle whose .wait() orders the caller's stream after it (torch.distributed.all_reduce(t, async_op=True)); the
    engine runs the cycle speculatively and `commit` restores the iterate if the reduced residual says the loop had
    already ended.  Bit-identical to sharded_solve."""
    engine.begin()
    it = 0
    while it < max_iter:
        chunk = min(check_every, max_iter - it)
        for _ in range(chunk):
            t = engine.residual_sumsq()
            work = all_reduce_async(t)
            engine.cycle_speculative()
            if work is not None:
                work.wait()
            engine.commit(t)
        it += chunk
        if it < max_iter:
            done, _ = engine.poll()
            if done:
                break
    return engine.end()


def sharded_solve(engine, max_iter, all_reduce, check_every=1):
    """for (iter < maxIter) { residual; push; if (residual < tol) break; V-cycle }  with the residual all-reduced.

    engine.residual_sumsq() -> tensor holding the LOCAL sum of squares; all_reduce(t) sums it over ranks in place;
    engine.cycle(t) appends sqrt(t) to the history, applies the break test and runs one V-cycle unless done.
    Every rank sees the same reduced value, hence takes the same decision at the same iteration."""
    engine.begin()
    it = 0
    while it < max_iter:
        chunk = min(check_every, max_iter - it)
        for _ in range(chunk):
            t = engine.residual_sumsq()
            all_reduce(t)
            engine.cycle(t)
        it += chunk
        if it < max_iter:
            done, _ = engine.poll()
            if done:
                break
    return engine.end()
