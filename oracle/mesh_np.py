"""numpy/scipy restatement of the caller-side numerics of the reference demos.  TEST INFRASTRUCTURE ONLY.

These are the libigl operations the three callers of the hot path use to *define the workload*
(03_mg_solver/main.cpp:29-61, 04_mg_solver_nobd/main.cpp:73-94, 05_example_mean_curvature_flow/main.cpp:57-69):
read mesh, normalize_unit_area, cotmatrix, massmatrix (VORONOI / BARYCENTRIC), boundary_loop, and the
mid-point upsampling operator of 09_random_subdiv_remesh/main.cpp:46-140.

libigl itself is not under /root/reference (empty submodule), so these follow libigl's documented
semantics -- PARITY UNPINNED at this third-party boundary.
"""
import os
import struct

import numpy as np
import scipy.sparse as sp

MESH_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "meshes")


def read_smgm(path):
    """Binary mesh fixture written by tests/golden/make_meshes.py."""
    if not os.path.isabs(path) and not os.path.exists(path):
        path = os.path.join(MESH_DIR, path)
    with open(path, "rb") as f:
        assert f.read(4) == b"SMGM"
        ver, nv, nf = struct.unpack("<Iii", f.read(12))
        assert ver == 1
        V = np.frombuffer(f.read(nv * 24), dtype="<f8").reshape(nv, 3).copy()
        F = np.frombuffer(f.read(nf * 12), dtype="<i4").reshape(nf, 3).copy()
    return V, F


def doublearea(V, F):
    """igl::doublearea: twice the triangle areas."""
    a, b, c = V[F[:, 0]], V[F[:, 1]], V[F[:, 2]]
    return np.linalg.norm(np.cross(b - a, c - a), axis=1)


def normalize_unit_area(V, F):
    """src/normalize_unit_area.cpp:10-24: scale to unit total area, centre x,y on the mean, put min z at 0."""
    FA = doublearea(V, F)
    scale = np.sqrt(FA.sum() / 2)
    V = V / scale
    V = V.copy()
    V[:, 0] -= V[:, 0].mean()
    V[:, 1] -= V[:, 1].mean()
    V[:, 2] -= V[:, 2].min()
    return V


def _edge_lengths(V, F):
    l0 = np.linalg.norm(V[F[:, 1]] - V[F[:, 2]], axis=1)
    l1 = np.linalg.norm(V[F[:, 2]] - V[F[:, 0]], axis=1)
    l2 = np.linalg.norm(V[F[:, 0]] - V[F[:, 1]], axis=1)
    return np.stack([l0, l1, l2], axis=1)


def cotmatrix(V, F):
    """igl::cotmatrix: L(i,j) = 1/2 (cot a_ij + cot b_ij), L(i,i) = -sum_j L(i,j)  (negative semi-definite).
    Entries via igl::cotmatrix_entries: C(f,e) = (l_j^2 + l_k^2 - l_e^2) / dblA / 4 for the edge e opposite
    corner e; assembly pattern of src/cotmatrix_dense.cpp:11-38."""
    n = V.shape[0]
    l = _edge_lengths(V, F)
    l2 = l * l
    dblA = doublearea(V, F)
    Cc = np.stack([(l2[:, 1] + l2[:, 2] - l2[:, 0]) / dblA / 4.0,
                   (l2[:, 2] + l2[:, 0] - l2[:, 1]) / dblA / 4.0,
                   (l2[:, 0] + l2[:, 1] - l2[:, 2]) / dblA / 4.0], axis=1)
    I, J, Vv = [], [], []
    edges = [(1, 2), (2, 0), (0, 1)]
    for e, (s, d) in enumerate(edges):
        src, dst = F[:, s], F[:, d]
        I += [src, dst, src, dst]
        J += [dst, src, src, dst]
        Vv += [Cc[:, e], Cc[:, e], -Cc[:, e], -Cc[:, e]]
    L = sp.coo_matrix((np.concatenate(Vv), (np.concatenate(I), np.concatenate(J))), shape=(n, n)).tocsc()
    L.sum_duplicates()
    L.sort_indices()
    return L


def massmatrix(V, F, kind="voronoi"):
    """igl::massmatrix, lumped diagonal.  'barycentric': M_ii = sum_{f ni i} area_f/3.
    'voronoi': mixed Voronoi cells with the 1/2,1/4,1/4 rule for obtuse triangles (libigl massmatrix.cpp)."""
    n = V.shape[0]
    dblA = doublearea(V, F)
    if kind == "barycentric":
        MV = np.repeat((dblA / 6.0)[:, None], 3, axis=1)
    else:
        l = _edge_lengths(V, F)
        cosines = np.stack([
            (l[:, 2] ** 2 + l[:, 1] ** 2 - l[:, 0] ** 2) / (l[:, 1] * l[:, 2] * 2.0),
            (l[:, 0] ** 2 + l[:, 2] ** 2 - l[:, 1] ** 2) / (l[:, 2] * l[:, 0] * 2.0),
            (l[:, 1] ** 2 + l[:, 0] ** 2 - l[:, 2] ** 2) / (l[:, 0] * l[:, 1] * 2.0)], axis=1)
        bary = cosines * l
        bary = bary / bary.sum(axis=1, keepdims=True)
        partial = bary * (dblA * 0.5)[:, None]
        quads = np.stack([(partial[:, 1] + partial[:, 2]) * 0.5,
                          (partial[:, 2] + partial[:, 0]) * 0.5,
                          (partial[:, 0] + partial[:, 1]) * 0.5], axis=1)
        for c in range(3):
            ob = cosines[:, c] < 0
            for cc in range(3):
                quads[ob, cc] = (0.25 if cc == c else 0.125) * dblA[ob]
        MV = quads
    m = np.zeros(n)
    for c in range(3):
        np.add.at(m, F[:, c], MV[:, c])
    return sp.diags(m).tocsc()


def boundary_loop(F):
    """igl::boundary_loop(F, VectorXi): the LONGEST boundary loop, as an ordered vertex list."""
    he = {}
    for c in range(3):
        for a, b in zip(F[:, c], F[:, (c + 1) % 3]):
            he[(int(a), int(b))] = True
    nxt = {}
    for (a, b) in he:
        if (b, a) not in he:
            nxt[a] = b
    loops, seen = [], set()
    for s in sorted(nxt):
        if s in seen:
            continue
        loop, v = [], s
        while v not in seen:
            seen.add(v)
            loop.append(v)
            v = nxt[v]
        loops.append(loop)
    if not loops:
        return np.zeros(0, dtype=np.int32)
    return np.asarray(max(loops, key=len), dtype=np.int32)


def midpoint_upsample(nV, F):
    """Mid-point upsampling with Loop connectivity (09_random_subdiv_remesh/main.cpp:46-140):
    old vertices keep their index, one new vertex per unique edge, numbered nV + rank of the edge in the
    lexicographically sorted unique (min,max) list; each face splits into 4.  Returns (S, NF) with NV = S @ V;
    S has a 1.0 on old rows and two 0.5 on new rows."""
    nF = F.shape[0]
    hE = np.concatenate([np.stack([F[:, i], F[:, (i + 1) % 3]], axis=1) for i in range(3)], axis=0)
    hE = np.sort(hE, axis=1)
    E, inv = np.unique(hE, axis=0, return_inverse=True)
    inv = inv.reshape(-1)
    m01, m12, m20 = nV + inv[:nF], nV + inv[nF:2 * nF], nV + inv[2 * nF:]
    NF = np.concatenate([
        np.stack([F[:, 0], m01, m20], axis=1),
        np.stack([F[:, 1], m12, m01], axis=1),
        np.stack([F[:, 2], m20, m12], axis=1),
        np.stack([m12, m20, m01], axis=1)], axis=0).astype(np.int32)
    nE = E.shape[0]
    rows = np.concatenate([np.arange(nV), nV + np.arange(nE), nV + np.arange(nE)])
    cols = np.concatenate([np.arange(nV), E[:, 0], E[:, 1]])
    vals = np.concatenate([np.ones(nV), 0.5 * np.ones(nE), 0.5 * np.ones(nE)])
    S = sp.coo_matrix((vals, (rows, cols)), shape=(nV + nE, nV)).tocsc()
    S.sort_indices()
    return S, NF


def subdivision_hierarchy(V, F, n_sub):
    """Returns (V_fine, F_fine, [P_1 .. P_nsub]) with P_l mapping level l (coarser) -> level l-1 (finer);
    level 0 is the finest mesh.  Positions are plain mid-points (S @ V)."""
    Ps = []
    for _ in range(n_sub):
        S, NF = midpoint_upsample(V.shape[0], F)
        V = S @ V
        F = NF
        Ps.append(S)
    return V, F, Ps[::-1]


def torus(nu, nv, R=1.0, r=0.4):
    """Closed torus grid, 2 triangles per quad (BASELINE config C5 base mesh)."""
    u = np.arange(nu) * (2 * np.pi / nu)
    v = np.arange(nv) * (2 * np.pi / nv)
    U, W = np.meshgrid(u, v, indexing="ij")
    V = np.stack([(R + r * np.cos(W)) * np.cos(U), (R + r * np.cos(W)) * np.sin(U), r * np.sin(W)], axis=-1).reshape(-1, 3)
    idx = lambda i, j: (i % nu) * nv + (j % nv)
    F = []
    for i in range(nu):
        for j in range(nv):
            a, b, c, d = idx(i, j), idx(i + 1, j), idx(i + 1, j + 1), idx(i, j + 1)
            F.append((a, b, c))
            F.append((a, c, d))
    return V, np.asarray(F, dtype=np.int32)
