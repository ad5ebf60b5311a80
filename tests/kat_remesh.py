"""Shared body of the one REFERENCE-PINNED known-answer test: the reference's checked-in 08_subdiv_remesh outputs (tests/golden/bunny_remesh_500.npz,
made from /root/reference by tests/golden/make_remesh_golden.py: data, not source).  Used by tests/test_host_logic.py (CPU lane) and by
tests/test_gpu_reference_kat.py (driver's -m gpu lane, which then solves on the very hierarchy this check has pinned)."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _level_mesh(smg, mg, lv):
    import ctypes as C
    L = smg._lib.load()
    nV, nF = C.c_int(), C.c_int()
    L.smg_level_get_mesh(mg.h, lv, C.byref(nV), C.byref(nF), None, None)
    Vc = np.zeros((nV.value, 3)); Fc = np.zeros((nF.value, 3), np.int32)
    L.smg_level_get_mesh(mg.h, lv, None, None, Vc.ctypes.data_as(C.POINTER(C.c_double)), Fc.ctypes.data_as(C.POINTER(C.c_int)))
    return Vc, Fc


def check_subdiv_remesh_kat(smg_mod):
    """REFERENCE-DERIVED golden vectors (tests/golden/make_remesh_golden.py): the reference checks in what its 08_subdiv_remesh example
    writes -- bunny.obj decimated to 500 faces by mid-point collapse, the coarse mesh mid-point-upsampled 0 / 1 / 2 times, every vertex
    carried back onto the input surface by query_coarse_to_fine (08_subdiv_remesh/main.cpp:131-166; output_s0/_s1/_s2.obj, 15 digits).
    libsmg's decimator + smg_query_coarse_to_fine must land on the same 261 / 1020 / 4035 points: that needs the same collapse sequence,
    the same joint flattenings (all three cases: bunny.obj has a boundary) and the same walk back through them.  Compared as point sets
    (the upsampled meshes' vertex numbering is libigl's): every point of ours has a reference point within 1e-9 of the bounding-box
    diagonal and vice versa (measured: 8e-14 absolute), and the coarse triangulation is the reference's."""
    from scipy.spatial import cKDTree
    smg, mesh = smg_mod, smg_mod.mesh
    G = np.load(os.path.join(ROOT, "tests", "golden", "bunny_remesh_500.npz"))
    V, F = mesh.read_triangle_mesh("bunny.smgm")
    diag = float(np.linalg.norm(V.max(0) - V.min(0)))
    ratio = float(np.float32(500 / 18555))
    mg = smg.mg_precompute(V, F, ratio, 200, 1, keep_log=True)
    Vc, Fc = _level_mesh(smg, mg, 1)
    assert Vc.shape == (261, 3) and Fc.shape == (499, 3)
    assert np.array_equal(Fc, G["s0_F"])      # the reference's coarse mesh face for face, corner for corner (and vertex for vertex, below)
    for k, den in ((0, 1), (1, 2), (2, 4)):
        faces, bar = [], []
        for f in range(Fc.shape[0]):       # all points with barycentric coordinates (i, j, den - i - j) / den of every coarse face
            for i in range(den + 1):
                for j in range(den + 1 - i):
                    faces.append(f); bar.append((i / den, j / den, (den - i - j) / den))
        of, ob = smg.query_coarse_to_fine(mg, 1, np.array(faces, np.int32), np.array(bar))
        assert ob.min() >= 0.0 and np.abs(ob.sum(1) - 1.0).max() < 1e-14
        P = (ob[:, :, None] * V[F[of]]).sum(1)
        ref = G["s%d_V" % k]
        d_mine, nearest = cKDTree(ref).query(P)
        d_ref, _ = cKDTree(P).query(ref)
        assert d_mine.max() <= 1e-9 * diag and d_ref.max() <= 1e-9 * diag, (k, d_mine.max(), d_ref.max())
        if k == 0:   # the coarse triangulation itself: our faces, named by the reference's vertices, are the reference's faces
            corner = {}
            for q, (f, b) in enumerate(zip(faces, bar)):
                corner[(f, int(np.argmax(b)))] = nearest[q]
            ours = {tuple(sorted(int(corner[(f, c)]) for c in range(3))) for f in range(Fc.shape[0])}
            theirs = {tuple(sorted(int(x) for x in t)) for t in G["s0_F"]}
            assert ours == theirs
    # the forward map is the inverse of the walk back: every row of P (a fine vertex as a point of a coarse face) returns to its vertex
    Pm = mg.matrix(1, "P_full").tocsr()
    fmap = {tuple(sorted(int(x) for x in t)): f for f, t in enumerate(Fc)}
    faces, bar = [], []
    for v in range(V.shape[0]):
        cols, vals = Pm.indices[Pm.indptr[v]:Pm.indptr[v + 1]], Pm.data[Pm.indptr[v]:Pm.indptr[v + 1]]
        f = fmap[tuple(sorted(int(c) for c in cols))]
        b = np.zeros(3)
        for c, val in zip(cols, vals):
            b[list(Fc[f]).index(c)] = val
        faces.append(f); bar.append(b)
    of, ob = smg.query_coarse_to_fine(mg, 1, np.array(faces, np.int32), np.array(bar))
    back = (ob[:, :, None] * V[F[of]]).sum(1)
    assert np.linalg.norm(back - V, axis=1).max() <= 1e-12 * diag
    return mg, V, F
