"""ctypes front-end of the plain-C oracle (oracle/smg_oracle.c).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: see oracle/smg_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libsmg_oracle.so")


def build(force=False):
    """Compile the C restatement with gcc (oracle/Makefile)."""
    srcs = [os.path.join(_HERE, f) for f in ("smg_oracle.c", "smg_oracle.h", "Makefile")]
    stale = (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


class _Csc(C.Structure):
    _fields_ = [("n_rows", C.c_int), ("n_cols", C.c_int), ("colptr", C.POINTER(C.c_int)),
                ("rowidx", C.POINTER(C.c_int)), ("val", C.POINTER(C.c_double))]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_SO)
    ip, dp, vp = C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_void_p
    L.orc_mg_create.restype = vp
    L.orc_mg_create.argtypes = [C.c_int]
    L.orc_mg_destroy.argtypes = [vp]
    L.orc_mg_set_prolong.argtypes = [vp, C.c_int, C.c_int, C.c_int, ip, ip, dp]
    L.orc_precompute.argtypes = [vp, C.c_int, ip, ip, dp]
    L.orc_precompute_known.argtypes = [vp, C.c_int, ip, ip, dp, ip, C.c_int]
    L.orc_solve.argtypes = [vp, dp, C.c_int, dp, C.c_int, C.c_int, C.c_double, C.c_int, dp, C.c_int, dp, ip]
    L.orc_solve_known.argtypes = [vp, dp, C.c_int, dp, C.c_int, dp, C.c_int, C.c_int, C.c_double,
                                  C.c_int, dp, C.c_int, dp, ip]
    L.orc_vcycle.argtypes = [vp, dp, C.c_int, C.c_int, C.c_int, dp, C.c_int]
    for f in (L.orc_A, L.orc_restrict, L.orc_prolong):
        f.argtypes = [vp, C.c_int, dp, C.c_int, dp]
    L.orc_relax.argtypes = [vp, C.c_int, dp, C.c_int, C.c_int, dp]
    L.orc_coarse_solve.argtypes = [vp, C.c_int, dp, C.c_int, dp]
    L.orc_level_rows.argtypes = [vp, C.c_int]
    for f in (L.orc_level_A, L.orc_level_P, L.orc_level_PT):
        f.argtypes = [vp, C.c_int]
        f.restype = C.POINTER(_Csc)
    for f in (L.orc_data_LHS, L.orc_data_Auk):
        f.argtypes = [vp]
        f.restype = C.POINTER(_Csc)
    L.orc_level_Adiag.argtypes = [vp, C.c_int]
    L.orc_level_Adiag.restype = dp
    L.orc_data_unknown.argtypes = [vp, C.POINTER(ip)]
    L.orc_profile.argtypes = [vp, dp, C.POINTER(C.c_long), dp, C.POINTER(C.c_long)]
    L.orc_profile_reset.argtypes = [vp]
    L.orc_set_parallel.argtypes = [vp, C.c_int, C.c_int, ip]
    L.orc_enable_parallel.argtypes = [vp, C.c_int, C.c_int]
    L.orc_set_smoother.argtypes = [vp, C.c_int, C.c_int, C.c_double]
    L.orc_spectral_bound.argtypes = [vp, C.c_int]
    L.orc_spectral_bound.restype = C.c_double
    L.orc_csc_times_dense.argtypes = [C.POINTER(_Csc), dp, C.c_int, C.c_int, dp, C.c_int]
    _lib = L
    return L


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _csc_arrays(M):
    M = sp.csc_matrix(M)
    M.sort_indices()
    return (np.ascontiguousarray(M.indptr, dtype=np.int32), np.ascontiguousarray(M.indices, dtype=np.int32),
            np.ascontiguousarray(M.data, dtype=np.float64))


def _csc_to_scipy(cptr):
    m = cptr.contents
    nnz = m.colptr[m.n_cols] if m.n_cols > 0 else 0
    indptr = np.ctypeslib.as_array(m.colptr, shape=(m.n_cols + 1,)).copy()
    indices = np.ctypeslib.as_array(m.rowidx, shape=(nnz,)).copy() if nnz else np.zeros(0, np.int32)
    data = np.ctypeslib.as_array(m.val, shape=(nnz,)).copy() if nnz else np.zeros(0)
    return sp.csc_matrix((data, indices, indptr), shape=(m.n_rows, m.n_cols))


def _colmajor(X):
    X = np.asarray(X, dtype=np.float64)
    if X.ndim == 1:
        X = X[:, None]
    return np.asfortranarray(X)


class OracleMG:
    """std::vector<mg_data> + min_quad_with_fixed_mg_data + SimplicialLDLT of the reference,
    driven through the oracle's C functions."""

    def __init__(self, prolongs):
        """prolongs: list of scipy sparse P_l (#V_{l-1} x #V_l), l = 1..L-1 (what mg_precompute fills)."""
        self.L = lib()
        self.n_levels = len(prolongs) + 1
        self.h = self.L.orc_mg_create(self.n_levels)
        self.has_known = False
        for l, P in enumerate(prolongs, start=1):
            cp, ri, v = _csc_arrays(P)
            rc = self.L.orc_mg_set_prolong(self.h, l, P.shape[0], P.shape[1], _ip(cp), _ip(ri), _dp(v))
            assert rc == 0

    def __del__(self):
        try:
            if self.h:
                self.L.orc_mg_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # -- min_quad_with_fixed_mg_precompute
    def precompute(self, A, known=None):
        cp, ri, v = _csc_arrays(A)
        n = A.shape[0]
        if known is None:
            rc = self.L.orc_precompute(self.h, n, _ip(cp), _ip(ri), _dp(v))
            self.has_known = False
        else:
            kn = np.ascontiguousarray(known, dtype=np.int32)
            rc = self.L.orc_precompute_known(self.h, n, _ip(cp), _ip(ri), _dp(v), _ip(kn), len(kn))
            self.has_known = True
            self.known = kn
        if rc != 0:
            raise RuntimeError("oracle precompute failed rc=%d" % rc)
        self.n = n

    # -- all-core comparator (NOT the reference's single-threaded path; SURVEY.md section 8d "fair CPU")
    def set_parallel(self, color_ptrs, threads=0):
        """color_ptrs[lv]: row offsets of contiguous, mutually independent row blocks of level lv (a multi-colouring the
        system is numbered by), for every smoothed level.  Returns the number of OpenMP threads.  Results of sweeps and
        products are bit-identical to the sequential run; only the residual norm is summed in another order."""
        for lv, cp in enumerate(color_ptrs):
            cp = np.ascontiguousarray(cp, dtype=np.int32)
            rc = self.L.orc_set_parallel(self.h, lv, len(cp) - 1, _ip(cp))
            if rc != 0:
                raise RuntimeError("orc_set_parallel(level %d) failed rc=%d (blocks not independent?)" % (lv, rc))
        return self.L.orc_enable_parallel(self.h, 1, int(threads))

    # -- smoother per level (extension: damped Jacobi, see smg_oracle.h)
    def spectral_bound(self, lv):
        return self.L.orc_spectral_bound(self.h, lv)

    def set_smoother(self, lv, kind="gs", omega=1.0):
        """kind "gs" / "jacobi" (omega = damping) / "chebyshev" (omega = interval fraction), see smg_oracle.h"""
        rc = self.L.orc_set_smoother(self.h, lv, {"gs": 0, "jacobi": 1, "chebyshev": 2}[kind], float(omega))
        if rc != 0:
            raise RuntimeError("orc_set_smoother failed rc=%d" % rc)

    def set_sequential(self):
        self.L.orc_enable_parallel(self.h, 0, 0)

    # -- min_quad_with_fixed_mg_solve
    def solve(self, RHS, z0, known_val=None, tol=1e-3, max_iter=20):
        RHS, z0 = _colmajor(RHS), _colmajor(z0)
        n, k = RHS.shape
        z = np.zeros((n, k), order="F")
        r_his = np.zeros(max(max_iter, 1))
        n_his = C.c_int(0)
        if not self.has_known:
            conv = self.L.orc_solve(self.h, _dp(RHS), n, _dp(z0), n, k, tol, max_iter, _dp(z), n,
                                    _dp(r_his), C.byref(n_his))
        else:
            kv = _colmajor(known_val if known_val is not None else np.zeros((len(self.known), k)))
            conv = self.L.orc_solve_known(self.h, _dp(RHS), n, _dp(kv), max(kv.shape[0], 1), _dp(z0), n, k, tol,
                                          max_iter, _dp(z), n, _dp(r_his), C.byref(n_his))
        return bool(conv), z, r_his[: n_his.value].copy()

    # -- mg_VCycle.cpp pieces
    def rows(self, lv):
        return self.L.orc_level_rows(self.h, lv)

    def vcycle(self, B, u, lv=0, pre=2, post=2):
        B, u = _colmajor(B), _colmajor(u).copy(order="F")
        self.L.orc_vcycle(self.h, _dp(B), pre, post, lv, _dp(u), B.shape[1])
        return u

    def relax(self, lv, B, u, iters):
        B, u = _colmajor(B), _colmajor(u).copy(order="F")
        self.L.orc_relax(self.h, lv, _dp(B), B.shape[1], iters, _dp(u))
        return u

    def _apply(self, fn, lv, x, nout):
        x = _colmajor(x)
        y = np.zeros((nout, x.shape[1]), order="F")
        fn(self.h, lv, _dp(x), x.shape[1], _dp(y))
        return y

    def A(self, lv, u):
        return self._apply(self.L.orc_A, lv, u, self.rows(lv))

    def restrict(self, lv, x):
        return self._apply(self.L.orc_restrict, lv, x, self.rows(lv + 1))

    def prolong(self, lv, x):
        return self._apply(self.L.orc_prolong, lv, x, self.rows(lv))

    def coarse_solve(self, B, u):
        lv = self.n_levels - 1
        B, u = _colmajor(B), _colmajor(u).copy(order="F")
        self.L.orc_coarse_solve(self.h, lv, _dp(B), B.shape[1], _dp(u))
        return u

    # -- introspection
    def level_A(self, lv):
        return _csc_to_scipy(self.L.orc_level_A(self.h, lv))

    def level_P(self, lv):
        return _csc_to_scipy(self.L.orc_level_P(self.h, lv))

    def level_PT(self, lv):
        return _csc_to_scipy(self.L.orc_level_PT(self.h, lv))

    def level_Adiag(self, lv):
        return np.ctypeslib.as_array(self.L.orc_level_Adiag(self.h, lv), shape=(self.rows(lv),)).copy()

    def data_LHS(self):
        return _csc_to_scipy(self.L.orc_data_LHS(self.h))

    def data_Auk(self):
        return _csc_to_scipy(self.L.orc_data_Auk(self.h))

    def unknown(self):
        p = C.POINTER(C.c_int)()
        n = self.L.orc_data_unknown(self.h, C.byref(p))
        return np.ctypeslib.as_array(p, shape=(n,)).copy()

    def profile(self):
        tr, tv = C.c_double(), C.c_double()
        cr, cv = C.c_long(), C.c_long()
        self.L.orc_profile(self.h, C.byref(tr), C.byref(cr), C.byref(tv), C.byref(cv))
        return {"MG: relaxation": (cr.value, tr.value), "MG: total VCycle": (cv.value, tv.value)}

    def profile_reset(self):
        self.L.orc_profile_reset(self.h)
