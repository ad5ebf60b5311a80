"""Caller-side mesh numerics of the reference demos, served by libsmg's host C++ (smg_mesh.cpp):
igl::read_triangle_mesh, normalize_unit_area, igl::cotmatrix, igl::massmatrix, igl::boundary_loop and the
mid-point upsampling operator (09_random_subdiv_remesh/main.cpp:46-140)."""
import ctypes as C
import os

import numpy as np
import scipy.sparse as sp

from . import _lib
from .api import _chk, _dp, _ip

FIXTURE_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "meshes")


def _vf(V, F):
    return np.ascontiguousarray(V, dtype=np.float64), np.ascontiguousarray(F, dtype=np.int32)


def read_triangle_mesh(path):
    L = _lib.load()
    if not os.path.exists(path) and os.path.exists(os.path.join(FIXTURE_DIR, path)):
        path = os.path.join(FIXTURE_DIR, path)
    Vp, Fp = C.POINTER(C.c_double)(), C.POINTER(C.c_int)()
    nV, nF = C.c_int(), C.c_int()
    _chk(L.smg_mesh_read(path.encode(), C.byref(Vp), C.byref(nV), C.byref(Fp), C.byref(nF)), "smg_mesh_read")
    V = np.ctypeslib.as_array(Vp, shape=(nV.value, 3)).copy()
    F = np.ctypeslib.as_array(Fp, shape=(nF.value, 3)).copy()
    L.smg_free(Vp)
    L.smg_free(Fp)
    return V, F


def normalize_unit_area(V, F):
    V, F = _vf(V, F)
    V = V.copy()
    _chk(_lib.load().smg_mesh_normalize_unit_area(_dp(V), V.shape[0], _ip(F), F.shape[0]), "normalize_unit_area")
    return V


def cotmatrix(V, F):
    L = _lib.load()
    V, F = _vf(V, F)
    nnz = C.c_int()
    _chk(L.smg_mesh_cotmatrix(_dp(V), V.shape[0], _ip(F), F.shape[0], C.byref(nnz), None, None, None), "cotmatrix")
    ptr = np.zeros(V.shape[0] + 1, np.int32)
    col = np.zeros(nnz.value, np.int32)
    val = np.zeros(nnz.value)
    _chk(L.smg_mesh_cotmatrix(_dp(V), V.shape[0], _ip(F), F.shape[0], None, _ip(ptr), _ip(col), _dp(val)), "cotmatrix")
    return sp.csr_matrix((val, col, ptr), shape=(V.shape[0], V.shape[0]))


def massmatrix(V, F, kind="voronoi"):
    V, F = _vf(V, F)
    d = np.zeros(V.shape[0])
    _chk(_lib.load().smg_mesh_massmatrix(_dp(V), V.shape[0], _ip(F), F.shape[0], int(kind == "voronoi"), _dp(d)),
         "massmatrix")
    return sp.diags(d).tocsr()


def boundary_loop(F, nV=None):
    F = np.ascontiguousarray(F, dtype=np.int32)
    nV = int(F.max()) + 1 if nV is None else nV
    loop = np.zeros(nV, np.int32)
    n = C.c_int()
    _chk(_lib.load().smg_mesh_boundary_loop(_ip(F), F.shape[0], nV, _ip(loop), C.byref(n)), "boundary_loop")
    return loop[: n.value].copy()


def midpoint_upsample(nV, F):
    L = _lib.load()
    F = np.ascontiguousarray(F, dtype=np.int32)
    nF = F.shape[0]
    nE = C.c_int()
    _chk(L.smg_mesh_midpoint_upsample(nV, _ip(F), nF, C.byref(nE), None, None, None, None), "midpoint_upsample")
    ne = nE.value
    ptr = np.zeros(nV + ne + 1, np.int32)
    col = np.zeros(nV + 2 * ne, np.int32)
    val = np.zeros(nV + 2 * ne)
    NF = np.zeros((4 * nF, 3), np.int32)
    _chk(L.smg_mesh_midpoint_upsample(nV, _ip(F), nF, None, _ip(ptr), _ip(col), _dp(val), _ip(NF)), "midpoint_upsample")
    return sp.csr_matrix((val, col, ptr), shape=(nV + ne, nV)), NF


def torus(nu, nv, R=1.0, r=0.4):
    V = np.zeros((nu * nv, 3))
    F = np.zeros((2 * nu * nv, 3), np.int32)
    _chk(_lib.load().smg_mesh_torus(nu, nv, R, r, _dp(V), _ip(F)), "torus")
    return V, F


class Assembler:
    """Operator assembly on the device for a fixed connectivity (include/smg.h, row f-3): per time step
    `M(U)`, `mass_coef * M + lap_coef * L(U)` and nothing leaves HBM.  Arguments are torch CUDA tensors (float64)."""

    def __init__(self, F, nV):
        self.L = _lib.load()
        F = np.ascontiguousarray(F, dtype=np.int32)
        out = C.c_void_p()
        _chk(self.L.smg_assembler_create(_ip(F), F.shape[0], int(nV), C.byref(out)), "smg_assembler_create")
        self.a = out
        self.nV = int(nV)
        nnz = C.c_int()
        _chk(self.L.smg_assembler_pattern(self.a, C.byref(nnz), None, None), "smg_assembler_pattern")
        self.nnz = nnz.value
        self.indptr = np.zeros(self.nV + 1, np.int32)
        self.indices = np.zeros(self.nnz, np.int32)
        _chk(self.L.smg_assembler_pattern(self.a, None, _ip(self.indptr), _ip(self.indices)), "smg_assembler_pattern")

    def __del__(self):
        try:
            if self.a:
                self.L.smg_assembler_destroy(self.a)
                self.a = None
        except Exception:
            pass

    def assemble(self, V_dev, mass_coef, lap_coef, kind="barycentric", val_out=None, mass_out=None, L_out=None, stream=0):
        """val = mass_coef * M + lap_coef * L(V); returns (val, mass) device tensors."""
        import torch
        assert V_dev.is_cuda and V_dev.dtype == torch.float64 and V_dev.is_contiguous() and V_dev.shape == (self.nV, 3)
        val = val_out if val_out is not None else torch.empty(self.nnz, dtype=torch.float64, device=V_dev.device)
        mass = mass_out if mass_out is not None else torch.empty(self.nV, dtype=torch.float64, device=V_dev.device)
        _chk(self.L.smg_assemble(self.a, V_dev.data_ptr(), int(kind == "voronoi"), float(mass_coef), float(lap_coef), val.data_ptr(),
                                 mass.data_ptr(), L_out.data_ptr() if L_out is not None else None, C.c_void_p(stream or 0)),
             "smg_assemble")
        return val, mass

    def pattern_matrix(self, val_host):
        return sp.csr_matrix((val_host, self.indices, self.indptr), shape=(self.nV, self.nV))
