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


class EmptyEngine:
    """A rank that owns no column (k < world size): it contributes 0 to the reduction and follows the others' decision, so that
    every rank still takes part in every all-reduce."""

    def __init__(self, tol, device=None, sync=None):
        """sync: callable that waits for the stream the reduction was enqueued on (StreamAllReduce runs on a raw HIP stream that
        torch's current stream is not ordered after); default: torch.cuda.synchronize() for a CUDA tensor."""
        import torch
        self.tol = tol
        self.sumsq = torch.zeros(1, dtype=torch.float64, device=device)
        self.done, self.r_his = False, []
        if sync is None and self.sumsq.is_cuda:
            sync = torch.cuda.synchronize
        self._sync = sync

    def begin(self):
        pass

    def residual_sumsq(self):
        self.sumsq.zero_()
        return self.sumsq

    def cycle(self, sumsq):
        if self.done:
            return
        if self._sync is not None:
            self._sync()
        r = float(sumsq.item()) ** 0.5
        self.r_his.append(r)
        if r < self.tol or r != r:
            self.done = True

    def cycle_speculative(self):
        pass

    commit = cycle

    def poll(self):
        return self.done, len(self.r_his)

    def end(self):
        last = self.r_his[-1] if self.r_his else float("inf")
        return not (last > self.tol), None, np.array(self.r_his)


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
        chunk = min(max(1, check_every), max_iter - it)   # check_every = 0 (the library's "adaptive") means 1 here
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
        chunk = min(max(1, check_every), max_iter - it)   # check_every = 0 (the library's "adaptive") means 1 here
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


class HostReduce:
    """A reduction closure for smg_solve_sharded that works with ANY torch.distributed backend (gloo on the test box, where two ranks
    share one GPU and RCCL refuses to run): wait for the solve's stream, fetch the doubles, all-reduce them on the host, put them back.
    The production closure is StreamAllReduce (RCCL on the solve's stream, no host round trip)."""

    def __init__(self, group=None):
        import ctypes as C
        self.C = C
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.group = group
        self.calls = 0

    def __call__(self, ptr, count, stream):
        import torch
        import torch.distributed as dist
        C = self.C
        buf = (C.c_double * count)()
        if self.hip.hipStreamSynchronize(stream) != 0 or self.hip.hipMemcpy(buf, ptr, 8 * count, 2) != 0:   # hipMemcpyDeviceToHost
            raise RuntimeError("HostReduce: device -> host copy failed")
        t = torch.tensor(list(buf), dtype=torch.float64)
        if dist.is_initialized() and dist.get_world_size(self.group) > 1:
            if dist.get_backend(self.group) == "nccl":      # RCCL reduces device tensors only: through torch's own stream, then back
                t = t.cuda()
                dist.all_reduce(t, group=self.group)
                torch.cuda.synchronize()
                t = t.cpu()
            else:
                dist.all_reduce(t, group=self.group)
        for i in range(count):
            buf[i] = float(t[i])
        if self.hip.hipMemcpy(ptr, buf, 8 * count, 1) != 0:                                                  # hipMemcpyHostToDevice
            raise RuntimeError("HostReduce: host -> device copy failed")
        self.calls += 1


def sharded_solve_native(mg, rhs, z0, reduce, known_val=None, opts=None):
    """The column-sharded solve through the library's own loop (smg_solve_sharded, include/smg.h): ONE call, the residual / reduce /
    cycle sequence, the device-side break test and the polling all happen in C++; `reduce(ptr, count, stream)` is the only thing the
    caller supplies (StreamAllReduce for RCCL: `lambda p, c, s: sar(p, c)`; HostReduce for any other backend).
    rhs, z0: (k_local, n) contiguous CUDA tensors = this rank's columns, column-major; None when the rank owns no column.
    `sharded_solve` above is the engine-agnostic model of the same loop that the CPU tests drive with the oracle."""
    import torch
    if rhs is None or rhs.shape[0] == 0:
        conv, r_his = mg.solve_sharded(None, None, None, 0, 0, reduce, opts=opts)
        return conv, None, r_his
    k, n = rhs.shape
    z = torch.empty_like(z0)
    kvp = known_val.data_ptr() if known_val is not None else None
    ldkv = known_val.shape[1] if known_val is not None else 0
    conv, r_his = mg.solve_sharded(rhs.data_ptr(), z0.data_ptr(), z.data_ptr(), n, k, reduce, kvp, ldkv, opts)
    return conv, z, r_his


# rccl.h (ROCm 7.2, NCCL API 2.27): the enum values of the two constants ncclAllReduce is called with.  They have been stable
# since NCCL 2.0; `StreamAllReduce` refuses libraries older than that and proves the whole binding (struct layout, enum values,
# stream) with a known-answer reduction before it reports `ok`.
NCCL_UNIQUE_ID_BYTES = 128     # rccl.h:40
NCCL_SUM = 0                   # ncclRedOp_t::ncclSum, rccl.h:448
NCCL_FLOAT64 = 8               # ncclDataType_t::ncclFloat64 / ncclDouble, rccl.h:467
NCCL_MIN_VERSION = 20000       # 2.0.0 in ncclGetVersion's encoding of that era (>= 2.9: major * 10000 + minor * 100 + patch)


class StreamAllReduce:
    """Sum-all-reduce of a small fp64 device buffer ON THE CALLER'S STREAM, through RCCL directly (ctypes on the librccl.so
    that torch loaded).  torch.distributed.all_reduce always runs on the process group's own stream: the two cross-stream
    hand-overs around an 8-byte reduction cost more than the reduction (about 20 us of stream time per outer iteration of
    the column-sharded solve).

    Two steps.  The constructor only loads and binds (`ready`).  `connect()` is a COLLECTIVE over the torch process group --
    every rank calls it, whatever its own `ready` says -- and every rank leaves it with the same verdict: the unique id (or an
    error sentinel) is broadcast from rank 0, each step is followed by an agreement (all-reduce MIN of a success flag), and
    before `ok` is set the communicator has to pass a known-answer reduction.  The bootstrap uses its own gloo group, so that a
    failure cannot leave a collective pending on the caller's group.  Callers fall back to torch's collective when it
    returns False."""

    def __init__(self, rank, world, stream_ptr, device=None):
        import ctypes as C
        import os
        self.ready = self.ok = False
        self.device = device
        self.rank, self.world = rank, world
        self.stream = C.c_void_p(stream_ptr)
        self.err = ""
        self.version = 0
        self.comm = None
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
                _fields_ = [("internal", C.c_char * NCCL_UNIQUE_ID_BYTES)]

            L.ncclGetVersion.argtypes = [C.POINTER(C.c_int)]
            L.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
            L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
            L.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
            L.ncclCommDestroy.argtypes = [C.c_void_p]
            L.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
            v = C.c_int(0)
            if L.ncclGetVersion(C.byref(v)) != 0 or v.value < NCCL_MIN_VERSION:
                raise RuntimeError("unsupported RCCL/NCCL version %d" % v.value)
            self.version = v.value
            self.L, self.UniqueId, self.C = L, UniqueId, C
            self.ready = True
        except Exception as e:   # pragma: no cover - depends on the installation
            self.err = repr(e)

    # -- agreement helpers (bootstrap group, CPU tensors)
    def _agree(self, group, flag):
        import torch
        import torch.distributed as dist
        t = torch.tensor([1 if flag else 0], dtype=torch.int32)
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        return int(t.item()) == 1

    def connect(self, timeout_s=120.0):
        """Collective (all ranks).  Returns the common verdict.  The RCCL calls run on a helper thread with a deadline: should the
        bootstrap hang on this rank, the rank gives up after `timeout_s`, and the final agreement turns that into False everywhere."""
        import threading
        import torch.distributed as dist
        group = final_group = None
        if self.world > 1:
            import datetime
            group = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=2 * timeout_s + 60))
            final_group = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=2 * timeout_s + 60))
        finished = threading.Event()
        verdict = [False]

        def work():
            verdict[0] = self._connect(group)
            finished.set()

        th = threading.Thread(target=work, daemon=True)
        th.start()
        timed_out = not finished.wait(timeout_s)
        if timed_out:
            self.err = "RCCL bootstrap timed out after %.0f s" % timeout_s
            self._abandoned = True     # a communicator that still arrives on the helper thread is destroyed there (_connect)
            verdict[0] = False
        # The final agreement runs on ITS OWN group (the helper thread of a timed-out rank may still sit in a collective of the
        # bootstrap group), so a rank that gave up still says so and its peers leave at once instead of waiting out their own deadline.
        try:
            self.ok = self._agree(final_group, bool(verdict[0])) and bool(verdict[0])
        except Exception as e:
            self.err = self.err or repr(e)
            self.ok = False
        if timed_out:
            return False
        if not self.ok and self.comm is not None:
            try:
                self.L.ncclCommDestroy(self.comm)
            except Exception:
                pass
            self.comm = None
        return self.ok

    def _connect(self, group):
        import torch
        import torch.distributed as dist
        C = getattr(self, "C", None)
        try:
            if not self._agree(group, self.ready):
                self.err = self.err or "another rank could not load RCCL"
                return False
            if self.device is not None:   # the current device is per thread
                torch.cuda.set_device(self.device)
            # 1. the unique id travels from rank 0; an error there travels instead of it
            uid = self.UniqueId()
            box = [None]
            if self.rank == 0:
                rc = self.L.ncclGetUniqueId(C.byref(uid))
                box[0] = C.string_at(C.addressof(uid), NCCL_UNIQUE_ID_BYTES) if rc == 0 else ("error", rc)
            if self.world > 1:
                dist.broadcast_object_list(box, src=0, group=group)
            if not isinstance(box[0], (bytes, bytearray)):
                self.err = "ncclGetUniqueId failed on rank 0: %r" % (box[0],)
                return False
            C.memmove(C.addressof(uid), box[0], NCCL_UNIQUE_ID_BYTES)
            # 2. communicator (itself a collective inside RCCL: every rank got a valid id, so every rank calls it)
            comm = C.c_void_p()
            rc = self.L.ncclCommInitRank(C.byref(comm), self.world, uid, self.rank)
            good = rc == 0
            if good:
                self.comm = comm
                cnt = C.c_int(0)
                good = self.L.ncclCommCount(comm, C.byref(cnt)) == 0 and cnt.value == self.world
            if not self._agree(group, good):
                self.err = self.err or "ncclCommInitRank failed on some rank (rc=%d here)" % rc
                return False
            # 3. known answer: sum over ranks of (rank + 1, 1) -- proves datatype / op constants, count and stream handling
            dev = self.device if self.device is not None else torch.device("cuda", torch.cuda.current_device())
            probe = torch.tensor([self.rank + 1.0, 1.0], dtype=torch.float64, device=dev)
            torch.cuda.synchronize()
            rc = self.L.ncclAllReduce(probe.data_ptr(), probe.data_ptr(), 2, NCCL_FLOAT64, NCCL_SUM, self.comm, self.stream)
            torch.cuda.synchronize()
            got = probe.tolist()
            good = rc == 0 and got == [self.world * (self.world + 1) / 2.0, float(self.world)]
            if not self._agree(group, good):
                self.err = self.err or "known-answer all-reduce gave %r (rc=%d)" % (got, rc)
                return False
            if getattr(self, "_abandoned", False):   # connect() gave up on this rank meanwhile: nobody will use (or close) this communicator
                try:
                    self.L.ncclCommDestroy(self.comm)
                except Exception:
                    pass
                self.comm = None
                return False
            return True
        except Exception as e:   # pragma: no cover
            self.err = repr(e)
            return False

    def n_ranks(self):
        """ranks of the RCCL communicator (ncclCommCount)"""
        cnt = self.C.c_int(0)
        return cnt.value if self.ok and self.L.ncclCommCount(self.comm, self.C.byref(cnt)) == 0 else 0

    def __call__(self, ptr, count=1):
        """in-place sum of `count` doubles at device pointer `ptr`, enqueued on the stream given at construction"""
        rc = self.L.ncclAllReduce(ptr, ptr, count, NCCL_FLOAT64, NCCL_SUM, self.comm, self.stream)
        if rc != 0:
            raise RuntimeError("ncclAllReduce failed: %d" % rc)

    def close(self):
        if self.ok:
            self.L.ncclCommDestroy(self.comm)
            self.ok = False
            self.comm = None
