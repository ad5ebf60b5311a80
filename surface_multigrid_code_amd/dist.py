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


class StreamAllReduce:
    """Sum-all-reduce of a small fp64 device buffer ON THE CALLER'S STREAM, through RCCL directly (ctypes on the librccl.so
    that torch loaded).  torch.distributed.all_reduce always runs on the process group's own stream: the two cross-stream
    hand-overs around an 8-byte reduction cost more than the reduction (about 20 us of stream time per outer iteration of
    the column-sharded solve).  Two steps, so that a rank that cannot load the library never leaves the others waiting in a
    collective: the constructor only loads and binds (`ready`), `connect()` -- to be called by all ranks or none --
    bootstraps the communicator over the existing torch process group (the unique id travels by broadcast_object_list)
    and sets `ok`.  Callers fall back to torch's collective when either flag stays False."""

    def __init__(self, rank, world, stream_ptr, device=None):
        import ctypes as C
        import os
        self.ready = self.ok = False
        self.device = device
        self.rank, self.world = rank, world
        self.stream = C.c_void_p(stream_ptr)
        self.err = ""
        try:
            import torch
            path = None
            with open("/proc/self/maps") as f:
                for line in f:
                    if "librccl" in line:
                        path = line.split()[-1]
                        break
            if path is None:
                path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
            L = C.CDLL(path)

            class UniqueId(C.Structure):
                _fields_ = [("internal", C.c_char * 128)]

            L.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
            L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
            L.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
            L.ncclCommDestroy.argtypes = [C.c_void_p]
            self.L, self.UniqueId, self.C = L, UniqueId, C
            self.ready = True
        except Exception as e:   # pragma: no cover - depends on the installation
            self.err = repr(e)

    def connect(self, timeout_s=120.0):
        """Collective.  Runs on a helper thread with a deadline: should the bootstrap hang, the caller gets False after
        `timeout_s` (and must not use this object), instead of the whole job hanging."""
        import threading
        done = threading.Event()

        def work():
            self._connect()
            done.set()

        th = threading.Thread(target=work, daemon=True)
        th.start()
        if not done.wait(timeout_s):
            self.err = "RCCL bootstrap timed out after %.0f s" % timeout_s
            self.ok = False
            self._abandoned = True
            return False
        return self.ok

    def _connect(self):
        C = self.C
        try:
            if self.device is not None:   # the current device is per thread
                import torch
                torch.cuda.set_device(self.device)
            uid = self.UniqueId()
            if self.rank == 0 and self.L.ncclGetUniqueId(C.byref(uid)) != 0:
                raise RuntimeError("ncclGetUniqueId failed")
            if self.world > 1:
                import torch.distributed as dist
                box = [C.string_at(C.addressof(uid), 128) if self.rank == 0 else None]
                dist.broadcast_object_list(box, src=0)
                C.memmove(C.addressof(uid), box[0], 128)
            comm = C.c_void_p()
            if self.L.ncclCommInitRank(C.byref(comm), self.world, uid, self.rank) != 0:
                raise RuntimeError("ncclCommInitRank failed")
            self.comm = comm
            self.ok = not getattr(self, "_abandoned", False)
        except Exception as e:   # pragma: no cover
            self.err = repr(e)

    def __call__(self, ptr, count=1):
        """in-place sum of `count` doubles at device pointer `ptr`, enqueued on the stream given at construction"""
        rc = self.L.ncclAllReduce(ptr, ptr, count, 8, 0, self.comm, self.stream)   # ncclFloat64 = 8, ncclSum = 0
        if rc != 0:
            raise RuntimeError("ncclAllReduce failed: %d" % rc)

    def close(self):
        if self.ok:
            self.L.ncclCommDestroy(self.comm)
            self.ok = False
