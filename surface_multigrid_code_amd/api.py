"""Python host mirror of the reference's operator API for the solve path, over the C ABI (include/smg.h).

Reference (HTDerekLiu/surface_multigrid_code) usage pattern, README.md:47-48 / 03_mg_solver/main.cpp:38-75:

    vector<mg_data> mg;   mg_precompute(V, F, ratio, nVCoarsest, dec_type, mg);
    min_quad_with_fixed_mg_precompute(A, known, data, mg, solver);
    min_quad_with_fixed_mg_solve(data, RHS, known_val, z0, solver, tol, maxIter, mg, z, rHis);

Here `mg` is a `Hierarchy` (owning the device-resident std::vector<mg_data>, the min_quad_with_fixed_mg_data and
the coarse solver); output arguments become return values.  Nothing in this module computes on the CPU.
"""
import ctypes as C

import numpy as np
import scipy.sparse as sp

from . import _lib
from ._lib import SMG_DEVICE, SMG_HOST, SolveOptsC


class SmgError(RuntimeError):
    def __init__(self, code, where):
        msg = _lib.load().smg_last_error()
        super().__init__("%s failed (%d): %s" % (where, code, msg.decode() if msg else ""))
        self.code = code


def _chk(rc, where):
    if rc != 0:
        raise SmgError(rc, where)


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _csr(M):
    M = sp.csr_matrix(M)
    M.sort_indices()
    return (np.ascontiguousarray(M.indptr, dtype=np.int32), np.ascontiguousarray(M.indices, dtype=np.int32),
            np.ascontiguousarray(M.data, dtype=np.float64))


def _colmajor(X):
    X = np.asarray(X, dtype=np.float64)
    if X.ndim == 1:
        X = X[:, None]
    return np.asfortranarray(X)


SMOOTHERS = {"gs": 0, "jacobi": 1, "hybrid": 2, "chebyshev": 3, "hybrid_chebyshev": 4, 0: 0, 1: 1, 2: 2, 3: 3, 4: 4}


class SolveOpts:
    """tol / maxIter / pre / post with the reference's defaults (src/min_quad_with_fixed_mg.cpp:63,77,102-103)."""

    def __init__(self, tol=1e-3, max_iter=20, pre=2, post=2, verbosity=0, check_every=0, use_graph=1, precision="f64",
                 smoother="gs", omega=0.8, jacobi_max_rows=100000, cheby_fraction=0.1):
        """smoother: "gs" (the reference's relax(), default) / "jacobi" (damped Jacobi on every level) / "hybrid" (Gauss-Seidel on
        the levels with more than `jacobi_max_rows` unknowns, Jacobi below) / "chebyshev", "hybrid_chebyshev" (the same layouts with
        Chebyshev-accelerated Jacobi: one polynomial of degree iters + 1 per relax(iters), include/smg.h)."""
        prec = {"f64": 0, "fp64": 0, 0: 0, "mixed": 1, 1: 1}[precision]
        self.c = SolveOptsC(tol, max_iter, pre, post, verbosity, check_every, use_graph, prec, SMOOTHERS[smoother], omega,
                            jacobi_max_rows, cheby_fraction)


class Hierarchy:
    """std::vector<mg_data> mg (+ solver data) living in HBM.  Wraps smg_hierarchy*."""

    def __init__(self, n_levels=None, handle=None):
        self.L = _lib.load()
        if handle is None:
            handle = self.L.smg_hierarchy_create(int(n_levels))
            if not handle:
                raise SmgError(-1, "smg_hierarchy_create")
        self.h = C.c_void_p(handle)
        self.n = None
        self.known = None

    def __del__(self):
        try:
            if self.h:
                self.L.smg_hierarchy_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ---- container
    @property
    def n_levels(self):
        return self.L.smg_hierarchy_levels(self.h)

    def set_prolong(self, lv, P):
        """mg[lv].P_full = mg[lv].P = P, mg[lv].PT = P^T (src/mg_precompute.cpp:71-77)."""
        ptr, col, val = _csr(P)
        _chk(self.L.smg_level_set_prolong(self.h, lv, P.shape[0], P.shape[1], _ip(ptr), _ip(col), _dp(val)),
             "smg_level_set_prolong")

    def set_stream(self, stream_ptr):
        """Use the given HIP stream (int handle, e.g. torch.cuda.current_stream().cuda_stream); None / 0 = the default stream."""
        _chk(self.L.smg_hierarchy_set_stream(self.h, C.c_void_p(stream_ptr or 0)), "smg_hierarchy_set_stream")

    def set_smoother(self, smoother="gs", omega=0.0, jacobi_max_rows=-1, cheby_fraction=0.0):
        """Smoother of vcycle()/relax() and the raw entry points (solve() takes it from its SolveOpts)."""
        _chk(self.L.smg_hierarchy_set_smoother(self.h, SMOOTHERS[smoother], omega, jacobi_max_rows), "smg_hierarchy_set_smoother")
        _chk(self.L.smg_hierarchy_set_chebyshev(self.h, cheby_fraction), "smg_hierarchy_set_chebyshev")

    def spectral_bound(self, lv):
        """Gershgorin bound of D^-1 A on level lv (what the Chebyshev-Jacobi smoother uses)."""
        return self.L.smg_level_spectral_bound(self.h, lv)

    def save(self, path):
        _chk(self.L.smg_hierarchy_save(self.h, path.encode()), "smg_hierarchy_save")

    @classmethod
    def load(cls, path):
        out = C.c_void_p()
        _chk(_lib.load().smg_hierarchy_load(path.encode(), C.byref(out)), "smg_hierarchy_load")
        return cls(handle=out.value)

    @classmethod
    def from_prolongs(cls, Ps):
        H = cls(len(Ps) + 1)
        for l, P in enumerate(Ps, start=1):
            H.set_prolong(l, P)
        return H

    # ---- min_quad_with_fixed_mg_precompute
    def precompute(self, A, known=None):
        ptr, col, val = _csr(A)
        n = A.shape[0]
        if known is None or len(known) == 0:
            rc = self.L.smg_precompute(self.h, n, _ip(ptr), _ip(col), _dp(val), None, 0)
            self.known = None
        else:
            kn = np.ascontiguousarray(known, dtype=np.int32)
            rc = self.L.smg_precompute(self.h, n, _ip(ptr), _ip(col), _dp(val), _ip(kn), len(kn))
            self.known = kn
        _chk(rc, "smg_precompute")
        self.n = n

    def precompute_values_device(self, d_val_ptr):
        """Same sparsity as the last precompute, new values already in HBM (device pointer, caller CSR order)."""
        _chk(self.L.smg_precompute_values_device(self.h, d_val_ptr), "smg_precompute_values_device")

    # ---- min_quad_with_fixed_mg_solve (host blocks)
    def solve(self, RHS, z0, known_val=None, opts=None):
        opts = opts or SolveOpts()
        RHS, z0 = _colmajor(RHS), _colmajor(z0)
        n, k = RHS.shape
        z = np.zeros((n, k), order="F")
        r_his = np.zeros(max(opts.c.max_iter, 1))
        n_his, conv = C.c_int(0), C.c_int(0)
        kv_p, ld_kv = None, 0
        if self.known is not None:
            kv = _colmajor(known_val if known_val is not None else np.zeros((len(self.known), k)))
            kv_p, ld_kv = kv.ctypes.data, kv.shape[0]
        rc = self.L.smg_solve(self.h, RHS.ctypes.data, n, kv_p, ld_kv, z0.ctypes.data, n, k, SMG_HOST,
                              C.byref(opts.c), z.ctypes.data, n, _dp(r_his), C.byref(n_his), C.byref(conv))
        _chk(rc, "smg_solve")
        return bool(conv.value), z, r_his[: n_his.value].copy()

    def solve_device(self, rhs_ptr, z0_ptr, z_ptr, n, k=1, known_val_ptr=None, ld_kv=0, opts=None):
        """min_quad_with_fixed_mg_solve on column-major blocks already resident in HBM (device pointers, leading dimension n):
        the drop-in call, polling the device-side convergence flag every opts.check_every iterations."""
        opts = opts or SolveOpts()
        r_his = np.zeros(max(opts.c.max_iter, 1))
        n_his, conv = C.c_int(0), C.c_int(0)
        _chk(self.L.smg_solve(self.h, rhs_ptr, n, known_val_ptr, ld_kv, z0_ptr, n, k, SMG_DEVICE, C.byref(opts.c), z_ptr, n,
                              _dp(r_his), C.byref(n_his), C.byref(conv)), "smg_solve")
        return bool(conv.value), r_his[: n_his.value].copy()

    def solve_sharded(self, rhs_ptr, z0_ptr, z_ptr, n, k_local, reduce, known_val_ptr=None, ld_kv=0, opts=None):
        """smg_solve_sharded: this rank's k_local columns (device pointers, leading dimension n; k_local = 0: no columns, pointers may be
        None), the library runs the whole loop and calls reduce(d_sumsq_ptr, count, hip_stream_ptr) -- which must sum the device doubles
        over the ranks in place, ordered on that stream -- once per loop entry.  Returns (converged, r_his)."""
        from ._lib import REDUCE_FN
        opts = opts or SolveOpts()
        r_his = np.zeros(max(opts.c.max_iter, 1))
        n_his, conv = C.c_int(0), C.c_int(0)
        failure = []

        def _cb(ptr, count, stream, _ctx):
            try:
                reduce(ptr, count, stream)
                return 0
            except BaseException as e:   # an exception must not unwind through the C frames
                failure.append(e)
                return 1
        cb = REDUCE_FN(_cb)
        rc = self.L.smg_solve_sharded(self.h, rhs_ptr, n, known_val_ptr, ld_kv, z0_ptr, n, k_local, SMG_DEVICE, C.byref(opts.c), cb, None,
                                      z_ptr, n, _dp(r_his), C.byref(n_his), C.byref(conv))
        if failure:
            raise failure[0]
        _chk(rc, "smg_solve_sharded")
        return bool(conv.value), r_his[: n_his.value].copy()

    # ---- mg_VCycle.h pieces (host blocks in the level's caller numbering)
    def rows(self, lv):
        return self.L.smg_level_rows(self.h, lv)

    def _rows_checked(self, lv, name, *blocks):
        """the C ABI takes bare pointers: a block of the wrong height would be read / written out of bounds"""
        n = self.rows(lv)
        for b in blocks:
            if b.shape[0] != n:
                raise ValueError("%s: level %d has %d rows, the block has %d (rows() is 0 before smg_precompute)" % (name, lv, n, b.shape[0]))
        return n

    def _piece(self, fn, name, lv, x, nout, nin=None):
        x = _colmajor(x)
        if nin is not None and x.shape[0] != nin:
            raise ValueError("%s: expected a block of %d rows, got %d" % (name, nin, x.shape[0]))
        y = np.zeros((nout, x.shape[1]), order="F")
        _chk(fn(self.h, lv, _dp(x), x.shape[1], _dp(y)), name)
        return y

    def A(self, lv, u):
        return self._piece(self.L.smg_apply_A, "smg_apply_A", lv, u, self.rows(lv), self.rows(lv))

    def restrict(self, lv, x):
        return self._piece(self.L.smg_restrict, "smg_restrict", lv, x, self.rows(lv + 1), self.rows(lv))

    def prolong(self, lv, x):
        return self._piece(self.L.smg_prolong, "smg_prolong", lv, x, self.rows(lv), self.rows(lv + 1))

    def relax(self, lv, B, u, iters):
        B, u = _colmajor(B), _colmajor(u).copy(order="F")
        self._rows_checked(lv, "smg_relax", B, u)
        _chk(self.L.smg_relax(self.h, lv, _dp(B), B.shape[1], iters, _dp(u)), "smg_relax")
        return u

    def coarse_solve(self, B, u):
        B, u = _colmajor(B), _colmajor(u).copy(order="F")
        self._rows_checked(self.n_levels - 1, "smg_coarse_solve", B, u)
        _chk(self.L.smg_coarse_solve(self.h, _dp(B), B.shape[1], _dp(u)), "smg_coarse_solve")
        return u

    def vcycle(self, B, u, lv=0, pre=2, post=2):
        B, u = _colmajor(B), _colmajor(u).copy(order="F")
        self._rows_checked(lv, "smg_vcycle", B, u)
        _chk(self.L.smg_vcycle(self.h, _dp(B), pre, post, lv, _dp(u), B.shape[1]), "smg_vcycle")
        return u

    def residual_norm(self, lv, B, u):
        B, u = _colmajor(B), _colmajor(u)
        self._rows_checked(lv, "smg_residual_norm", B, u)
        out = C.c_double(0)
        _chk(self.L.smg_residual_norm(self.h, lv, _dp(B), _dp(u), B.shape[1], C.byref(out)), "smg_residual_norm")
        return out.value

    # ---- introspection
    def matrix(self, lv, which="A", internal=False):
        w = {"A": 0, "P": 1, "PT": 2, "P_full": 3, "Auk": 4}[which]
        nr, nc, nnz = C.c_int(), C.c_int(), C.c_int()
        _chk(self.L.smg_level_get_matrix(self.h, lv, w, int(internal), C.byref(nr), C.byref(nc), C.byref(nnz),
                                         None, None, None), "smg_level_get_matrix")
        ptr = np.zeros(nr.value + 1, np.int32)
        col = np.zeros(max(nnz.value, 1), np.int32)
        val = np.zeros(max(nnz.value, 1))
        _chk(self.L.smg_level_get_matrix(self.h, lv, w, int(internal), None, None, None, _ip(ptr), _ip(col), _dp(val)),
             "smg_level_get_matrix")
        return sp.csr_matrix((val[: nnz.value], col[: nnz.value], ptr), shape=(nr.value, nc.value))

    def perm(self, lv):
        p = np.zeros(self.rows(lv), np.int32)
        _chk(self.L.smg_level_get_perm(self.h, lv, _ip(p)), "smg_level_get_perm")
        return p

    def colors(self, lv):
        nc = C.c_int()
        _chk(self.L.smg_level_get_colors(self.h, lv, C.byref(nc), None), "smg_level_get_colors")
        cp = np.zeros(nc.value + 1, np.int32)
        _chk(self.L.smg_level_get_colors(self.h, lv, None, _ip(cp)), "smg_level_get_colors")
        return cp

    def Adiag(self, lv):
        d = np.zeros(self.rows(lv))
        _chk(self.L.smg_level_get_Adiag(self.h, lv, _dp(d)), "smg_level_get_Adiag")
        return d

    def unknown(self):
        n = C.c_int()
        _chk(self.L.smg_get_unknown(self.h, C.byref(n), None), "smg_get_unknown")
        u = np.zeros(n.value, np.int32)
        _chk(self.L.smg_get_unknown(self.h, None, _ip(u)), "smg_get_unknown")
        return u

    def sell_stats(self, lv, which="A"):
        st, pd, ns = C.c_long(), C.c_long(), C.c_int()
        _chk(self.L.smg_level_sell_stats(self.h, lv, {"A": 0, "P": 1, "PT": 2}[which], C.byref(st), C.byref(pd),
                                         C.byref(ns)), "smg_level_sell_stats")
        return {"stored": st.value, "padded": pd.value, "n_slices": ns.value}

    def first_colour_rows(self, lv):
        """rows of level lv's first colour the restriction launch of level lv - 1 can update itself (0: not available)"""
        return self.L.smg_level_first_colour_rows(self.h, lv)

    # ---- block-sequential Gauss-Seidel for solves with a multiple of 16 columns (k % 16 == 0, k >= 16)
    def set_block_gs(self, min_rows):
        """levels of at least min_rows rows sweep block-sequentially when k % 16 == 0, k >= 16 (< 0: never)"""
        _chk(self.L.smg_hierarchy_set_block_gs(self.h, int(min_rows)), "smg_hierarchy_set_block_gs")

    def block_gs_order(self, lv, k):
        """None when level lv does not sweep block-sequentially for k columns, else a dict: rows (position -> internal row), blk_ptr,
        color_ptr, rim, fill"""
        nb, nc = C.c_int(), C.c_int()
        rc = self.L.smg_level_get_block_gs_order(self.h, lv, int(k), C.byref(nb), C.byref(nc), None, None, None, None)
        if rc < 0:
            _chk(rc, "smg_level_get_block_gs_order")
        if rc == 0:
            return None
        cp, bp, rows, st = np.zeros(nc.value + 1, np.int32), np.zeros(nb.value + 1, np.int32), np.zeros(self.rows(lv), np.int32), np.zeros(2)
        _chk(min(self.L.smg_level_get_block_gs_order(self.h, lv, int(k), None, None, _ip(cp), _ip(bp), _ip(rows), _dp(st)), 0), "smg_level_get_block_gs_order")
        return {"rows": rows, "blk_ptr": bp, "color_ptr": cp, "rim": st[0], "fill": st[1]}

    # ---- piece-wise Gauss-Seidel on the Galerkin levels of decimated hierarchies (csrc/smg_wgs.hpp)
    def set_wave_gs(self, mode="auto"):
        """auto: levels the colour launches serve badly (> 5 colours or rows of > 12 entries); never; all: every Gauss-Seidel level in range"""
        _chk(self.L.smg_hierarchy_set_wave_gs(self.h, {"auto": -1, "never": 0, "all": 1}[mode]), "smg_hierarchy_set_wave_gs")

    def wave_gs_order(self, lv, k=1):
        """None when level lv does not sweep piece-wise for k columns, else a dict: rows (position -> internal row), piece_ptr, color_ptr, rim,
        phases_mean, phases_max"""
        nb, nc = C.c_int(), C.c_int()
        rc = self.L.smg_level_get_wave_gs_order(self.h, lv, int(k), C.byref(nb), C.byref(nc), None, None, None, None)
        if rc < 0:
            _chk(rc, "smg_level_get_wave_gs_order")
        if rc == 0:
            return None
        cp, bp, rows, st = np.zeros(nc.value + 1, np.int32), np.zeros(nb.value + 1, np.int32), np.zeros(self.rows(lv), np.int32), np.zeros(3)
        _chk(min(self.L.smg_level_get_wave_gs_order(self.h, lv, int(k), None, None, _ip(cp), _ip(bp), _ip(rows), _dp(st)), 0), "smg_level_get_wave_gs_order")
        return {"rows": rows, "piece_ptr": bp, "color_ptr": cp, "rim": st[0], "phases_mean": st[1], "phases_max": int(st[2])}

    def gs_order(self, lv, k=1):
        """position -> internal row of the Gauss-Seidel order relax() uses on level lv with k columns: the piece order where the level sweeps
        piece-wise, the block order where it sweeps block-wise, else the internal numbering itself (one launch per colour / overlapped tiling)"""
        w = self.wave_gs_order(lv, k)
        if w is not None:
            return w["rows"]
        b = self.block_gs_order(lv, k)
        if b is not None:
            return b["rows"]
        return np.arange(self.rows(lv), dtype=np.int32)

    # ---- independent meshes in one handle (csrc/smg_union.cpp)
    @classmethod
    def union(cls, members):
        """one block-diagonal hierarchy over the members' prolongations (smg_hierarchy_create_union); precompute() then takes the block-diagonal system"""
        L = _lib.load()
        arr = (C.c_void_p * len(members))(*[m.h for m in members])
        out = C.c_void_p()
        _chk(L.smg_hierarchy_create_union(arr, len(members), C.byref(out)), "smg_hierarchy_create_union")
        return cls(handle=out.value)

    def union_members(self):
        return self.L.smg_union_members(self.h)

    def union_member_rows(self, member):
        a, b = C.c_int(), C.c_int()
        _chk(self.L.smg_union_member_rows(self.h, member, C.byref(a), C.byref(b)), "smg_union_member_rows")
        return a.value, b.value

    def union_history(self, member, cap=4096):
        """(converged, r_his) of one member's own loop in the last solve"""
        rh, n, cv = np.zeros(cap), C.c_int(), C.c_int()
        _chk(self.L.smg_union_get_history(self.h, member, _dp(rh), cap, C.byref(n), C.byref(cv)), "smg_union_get_history")
        return bool(cv.value), rh[:min(n.value, cap)].copy()

    # ---- coarsest-level solver
    def set_coarse_dense_max(self, n_max):
        """coarsest levels of more than n_max unknowns get a sparse Cholesky factorisation instead of a dense inverse"""
        _chk(self.L.smg_hierarchy_set_coarse_dense_max(self.h, int(n_max)), "smg_hierarchy_set_coarse_dense_max")

    def set_coarse_schur(self, when="refactor", n_min=-1):
        """The Schur-complement coarse solver (block elimination + dense inverse of the separator only) for coarsest levels of n_min (default 2048;
        -1: unchanged) to 65 536 unknowns: 'never', 'always' (from the first precompute on) or 'refactor' (default, the choice by cost: below 6 144
        unknowns from the first value-only re-precompute on -- what a time-stepping caller does --, from 6 144 on and above the dense range at once)."""
        _chk(self.L.smg_hierarchy_set_coarse_schur(self.h, {"never": 0, "always": 1, "refactor": 2}[when], int(n_min)), "smg_hierarchy_set_coarse_schur")

    def set_memory_lean(self, on=True):
        """compact SELL panels instead of the fixed panel pitch: ~0.77 x the device memory, the cycle ~15 % slower, same bits; the next precompute is a full one"""
        _chk(self.L.smg_hierarchy_set_memory_lean(self.h, 1 if on else 0), "smg_hierarchy_set_memory_lean")

    def coarse_solver(self):
        ne = C.c_long()
        kind = self.L.smg_hierarchy_coarse_solver(self.h, C.byref(ne))
        return {"kind": {1: "sparse_cholesky", 2: "schur_complement"}.get(kind, "dense_inverse"), "factor_entries": ne.value}

    # ---- block (3-DOF) variant
    def set_block_mode(self, mode="auto"):
        """'auto' (decide at precompute), 'scalar' (never), 'block' (3 x 3 kernels required)."""
        _chk(self.L.smg_hierarchy_set_block_mode(self.h, {"auto": -1, "scalar": 0, "block": 3}.get(mode, mode)), "smg_hierarchy_set_block_mode")

    def block_size(self):
        return self.L.smg_hierarchy_block_size(self.h)

    def block_stats(self, lv=0):
        nb, ns, nc = C.c_long(), C.c_long(), C.c_int()
        _chk(self.L.smg_level_block_stats(self.h, lv, C.byref(nb), C.byref(ns), C.byref(nc)), "smg_level_block_stats")
        return {"blocks": nb.value, "block_slots": ns.value, "vertex_colors": nc.value}

    def block_image(self, lv=0):
        """The block SELL image of A_lv (host-built twin of what the device holds): dict of slice_row, slice_off, slice_w, col (slots x 64),
        val (slots x 9 x 64)."""
        ns, npc = C.c_int(), C.c_int()
        _chk(self.L.smg_level_get_block_image(self.h, lv, C.byref(ns), C.byref(npc), None, None, None, None, None), "smg_level_get_block_image")
        sr, so, sw = np.zeros(ns.value + 1, np.int32), np.zeros(ns.value + 1, np.int32), np.zeros(ns.value, np.int32)
        col, val = np.zeros(npc.value * 64, np.int32), np.zeros(npc.value * 9 * 64, np.float64)
        _chk(self.L.smg_level_get_block_image(self.h, lv, None, None, _ip(sr), _ip(so), _ip(sw), _ip(col), _dp(val)), "smg_level_get_block_image")
        return {"slice_row": sr, "slice_off": so, "slice_w": sw, "col": col.reshape(-1, 64), "val": val.reshape(-1, 9, 64)}

    def device_bytes(self):
        """what this handle holds in HBM, by purpose: dict name -> bytes (incl. "total")"""
        buf = C.create_string_buffer(1 << 16)
        _chk(self.L.smg_debug_device_bytes(self.h, buf, len(buf)), "smg_debug_device_bytes")
        return {ln.split()[0]: int(ln.split()[1]) for ln in buf.value.decode().splitlines() if ln.strip()}

    def spmv_bytes(self, lv=0, k=1):
        return self.L.smg_level_spmv_bytes(self.h, lv, k)

    def vcycle_bytes(self, k=1, pre=2, post=2):
        return self.L.smg_vcycle_bytes(self.h, k, pre, post)

    # ---- profc mirror
    def prof_enable(self, on=True):
        _chk(self.L.smg_prof_enable(self.h, int(on)), "smg_prof_enable")

    def prof_reset(self):
        _chk(self.L.smg_prof_reset(self.h), "smg_prof_reset")

    def prof_table(self):
        out = {}
        for i in range(self.L.smg_prof_count(self.h)):
            name = C.create_string_buffer(128)
            cnt, ms = C.c_long(), C.c_double()
            _chk(self.L.smg_prof_get(self.h, i, name, 128, C.byref(cnt), C.byref(ms)), "smg_prof_get")
            out[name.value.decode()] = (cnt.value, ms.value)
        return out

    # ---- device-resident interface (torch tensors / raw device pointers)
    def solve_begin(self, rhs_ptr, ld_rhs, z0_ptr, ld_z0, k, known_val_ptr=None, ld_kv=0, opts=None,
                    memspace=SMG_DEVICE):
        opts = opts or SolveOpts()
        self._opts = opts
        _chk(self.L.smg_solve_begin(self.h, rhs_ptr, ld_rhs, known_val_ptr, ld_kv, z0_ptr, ld_z0, k, memspace,
                                    C.byref(opts.c)), "smg_solve_begin")

    def iter_residual(self, d_sumsq_ptr):
        _chk(self.L.smg_solve_iter_residual(self.h, d_sumsq_ptr), "smg_solve_iter_residual")

    def iter_cycle(self, d_sumsq_ptr):
        _chk(self.L.smg_solve_iter_cycle(self.h, d_sumsq_ptr), "smg_solve_iter_cycle")

    def iter_cycle_speculative(self):
        _chk(self.L.smg_solve_iter_cycle_speculative(self.h), "smg_solve_iter_cycle_speculative")

    def iter_commit(self, d_sumsq_ptr):
        _chk(self.L.smg_solve_iter_commit(self.h, d_sumsq_ptr), "smg_solve_iter_commit")

    def outer_iterations(self, n):
        _chk(self.L.smg_raw_outer_iteration(self.h, n), "smg_raw_outer_iteration")

    def poll(self):
        done, nh = C.c_int(), C.c_int()
        _chk(self.L.smg_solve_poll(self.h, C.byref(done), C.byref(nh)), "smg_solve_poll")
        return bool(done.value), nh.value

    def solve_end(self, z_ptr, ld_z, memspace=SMG_DEVICE, max_iter=None):
        cap = max(max_iter or self._opts.c.max_iter, 1)
        r_his = np.zeros(max(cap, 1024))
        n_his, conv = C.c_int(0), C.c_int(0)
        _chk(self.L.smg_solve_end(self.h, z_ptr, ld_z, memspace, _dp(r_his), C.byref(n_his), C.byref(conv)),
             "smg_solve_end")
        return bool(conv.value), r_his[: n_his.value].copy()

    def raw_spmv(self, lv, mode, x_ptr, b_ptr, y_ptr, k=1):
        _chk(self.L.smg_raw_spmv(self.h, lv, mode, x_ptr, b_ptr, y_ptr, k), "smg_raw_spmv")

    def raw_spmv_f32(self, lv, x_ptr, y_ptr, k=1):
        _chk(self.L.smg_raw_spmv_f32(self.h, lv, x_ptr, y_ptr, k), "smg_raw_spmv_f32")

    def raw_relax(self, lv, b_ptr, u_ptr, k=1, iters=1):
        _chk(self.L.smg_raw_relax(self.h, lv, b_ptr, u_ptr, k, iters), "smg_raw_relax")

    def bench_vcycle(self, lv=0, k=1, pre=2, post=2, reps=50):
        out = C.c_double(0)
        _chk(self.L.smg_bench_vcycle(self.h, lv, k, pre, post, reps, C.byref(out)), "smg_bench_vcycle")
        return out.value

    def bench_relax(self, lv=0, k=1, sweeps=2, reps=50):
        out = C.c_double(0)
        _chk(self.L.smg_bench_relax(self.h, lv, k, sweeps, reps, C.byref(out)), "smg_bench_relax")
        return out.value

    def synchronize(self):
        _chk(self.L.smg_synchronize(self.h), "smg_synchronize")


# ---------------------------------------------------------------------------------------------------------------------
# the reference's free functions

def mg_precompute(V, F, ratio=0.25, nVCoarsest=500, dec_type=1, absorption_cap=0.0, keep_log=False):
    """mg_precompute(V, F, ratio, nVCoarsest, dec_type, mg)  (src/mg_precompute.cpp:15-87) -> Hierarchy.
    absorption_cap > 0: opt-in departure from the reference's collapse order (include/smg.h: smg_mg_precompute_capped).
    keep_log: keep the record of every collapse (the reference's decInfo), which query_coarse_to_fine needs."""
    L = _lib.load()
    V = np.ascontiguousarray(V, dtype=np.float64)
    F = np.ascontiguousarray(F, dtype=np.int32)
    out = C.c_void_p()
    _chk(L.smg_mg_precompute_logged(_dp(V), V.shape[0], _ip(F), F.shape[0], ratio, nVCoarsest, dec_type, absorption_cap, int(bool(keep_log)),
                                    C.byref(out)), "smg_mg_precompute")
    return Hierarchy(handle=out.value)


def query_fine_to_coarse(mg, lv, face, bary):
    """query_fine_to_coarse (src/query_fine_to_coarse.cpp): points (face of level lv - 1's mesh, barycentric coordinates) -> (face of level
    lv's mesh, barycentric coordinates); the inverse of query_coarse_to_fine (needs keep_log=True)."""
    face = np.ascontiguousarray(face, dtype=np.int32)
    bary = np.ascontiguousarray(bary, dtype=np.float64).reshape(-1, 3)
    assert bary.shape[0] == face.shape[0]
    of = np.zeros(face.shape[0], dtype=np.int32)
    ob = np.zeros_like(bary)
    _chk(mg.L.smg_query_fine_to_coarse(mg.h, int(lv), face.shape[0], _ip(face), _dp(bary), _ip(of), _dp(ob)), "smg_query_fine_to_coarse")
    return of, ob


def query_coarse_to_fine(mg, lv, face, bary):
    """query_coarse_to_fine (src/query_coarse_to_fine.cpp): points (face of level lv's mesh, barycentric coordinates) -> (face of level
    lv - 1's mesh, barycentric coordinates) through the bijection of the decimation that built level lv (needs keep_log=True)."""
    face = np.ascontiguousarray(face, dtype=np.int32)
    bary = np.ascontiguousarray(bary, dtype=np.float64).reshape(-1, 3)
    assert bary.shape[0] == face.shape[0]
    of = np.zeros(face.shape[0], dtype=np.int32)
    ob = np.zeros_like(bary)
    _chk(mg.L.smg_query_coarse_to_fine(mg.h, int(lv), face.shape[0], _ip(face), _dp(bary), _ip(of), _dp(ob)), "smg_query_coarse_to_fine")
    return of, ob


def mg_precompute_block(V, F, ratio=0.25, nVCoarsest=500, dec_type=1):
    """mg_precompute_block (src/mg_precompute_block.cpp:23-95): P (x) I_3, DOF index 3*vertex + d."""
    L = _lib.load()
    V = np.ascontiguousarray(V, dtype=np.float64)
    F = np.ascontiguousarray(F, dtype=np.int32)
    out = C.c_void_p()
    _chk(L.smg_mg_precompute_block(_dp(V), V.shape[0], _ip(F), F.shape[0], ratio, nVCoarsest, dec_type, C.byref(out)),
         "smg_mg_precompute_block")
    return Hierarchy(handle=out.value)


def mg_precompute_subdiv(V, F, n_sub, ratio=0.25, nVCoarsest=500, n_extra_levels=-1):
    """Hierarchy of a mid-point-subdivided mesh (benchmark configs C3/C5).  Returns (mg, V_fine, F_fine)."""
    L = _lib.load()
    V = np.ascontiguousarray(V, dtype=np.float64)
    F = np.ascontiguousarray(F, dtype=np.int32)
    nV, nF = V.shape[0], F.shape[0]
    # closed-form sizes: every step adds one vertex per edge and quadruples the faces
    nv, nf = nV, nF
    he = set()
    if n_sub > 0:
        E0 = np.sort(np.concatenate([F[:, [0, 1]], F[:, [1, 2]], F[:, [2, 0]]]), axis=1)
        ne = np.unique(E0, axis=0).shape[0]
        for _ in range(n_sub):
            nv, ne, nf = nv + ne, 2 * ne + 3 * nf, 4 * nf
    Vo = np.zeros((nv, 3))
    Fo = np.zeros((nf, 3), dtype=np.int32)
    out = C.c_void_p()
    _chk(L.smg_mg_precompute_subdiv(_dp(V), nV, _ip(F), nF, n_sub, ratio, nVCoarsest, n_extra_levels, C.byref(out),
                                    _dp(Vo), _ip(Fo)), "smg_mg_precompute_subdiv")
    return Hierarchy(handle=out.value), Vo, Fo


class min_quad_with_fixed_mg_data:
    """min_quad_with_fixed_mg_data (src/min_quad_with_fixed_mg.h:22-29); LHS/Auk live in the Hierarchy."""

    def __init__(self, mg):
        self.n = mg.n
        self.known = mg.known if mg.known is not None else np.zeros(0, np.int32)
        self.unknown = mg.unknown()


def min_quad_with_fixed_mg_precompute(A, known, mg):
    """min_quad_with_fixed_mg_precompute(A, [known,] data, mg, solver)  (src/min_quad_with_fixed_mg.cpp:3-51, :137-257).
    Returns `data`; the coarse `solver` is owned by `mg`."""
    mg.precompute(A, known)
    return min_quad_with_fixed_mg_data(mg)


def min_quad_with_fixed_mg_solve(data, RHS, known_val, z0, mg, tol=1e-3, maxIter=20, opts=None):
    """bool min_quad_with_fixed_mg_solve(data, RHS, [known_val,] z0, solver, [tol, [maxIter,]] mg, z, r_his)
    (src/min_quad_with_fixed_mg.cpp:80-135, :288-361).  Returns (converged, z, r_his)."""
    o = opts or SolveOpts(tol=tol, max_iter=maxIter)
    return mg.solve(RHS, z0, known_val, o)


def mg_VCycle(mg, B, preRelaxIter, postRelaxIter, lv, u):
    """mg_VCycle(solver, B, pre, post, lv, u, mg)  (src/mg_VCycle.cpp:3-59).  Returns the updated u."""
    return mg.vcycle(B, u, lv, preRelaxIter, postRelaxIter)
