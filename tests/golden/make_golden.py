#!/usr/bin/env python3
"""Generate restatement-derived golden vectors for the solve path (tests/golden/*.npz).

The reference has no tests / golden vectors of its own and cannot be built here (Eigen + libigl absent), so these
vectors come from an INDEPENDENT scipy restatement of the published algorithm (V(2,2) cycle, forward lexicographic
Gauss-Seidel = sparse triangular solve with tril(A), Galerkin PT*A*P, +1e-12 on the coarsest diagonal, exact coarse
solve with SuperLU, residual measured before every cycle) -- NOT from the C oracle and NOT from the reference binary.
They pin the oracle (tests/test_oracle_golden.py); labelled "restatement-derived, not reference-derived".

Run in the build container:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))
from problems import subdiv_problem  # noqa: E402


def gs_sweeps(A, b, u, iters):
    Lw = sp.tril(A, 0).tocsr()
    U = sp.triu(A, 1).tocsr()
    for _ in range(iters):
        u = spla.spsolve_triangular(Lw, b - U @ u, lower=True)
    return u


def vcycle(As, Ps, lus, b, u, lv):
    if lv == len(As) - 1:
        return u + lus.solve(b)
    u = gs_sweeps(As[lv], b, u, 2)
    r = b - As[lv] @ u
    rc = Ps[lv].T @ r
    uc = vcycle(As, Ps, lus, rc, np.zeros_like(rc), lv + 1)
    u = u + Ps[lv] @ uc
    return gs_sweeps(As[lv], b, u, 2)


def eliminate(A, Ps, known):
    """min_quad_with_fixed_mg_precompute with `known` (reference src/min_quad_with_fixed_mg.cpp:137-257)."""
    n = A.shape[0]
    unknown = np.setdiff1d(np.arange(n), known)
    A = A.tocsr()
    LHS = A[unknown][:, unknown]
    Auk = A[unknown][:, known]
    Ps = [P.tocsc() for P in Ps]
    Ps[0] = Ps[0].tocsr()[unknown].tocsc()
    for l in range(len(Ps)):
        P = Ps[l].tocsc()
        keep = np.array([c for c in range(P.shape[1]) if (P.data[P.indptr[c]:P.indptr[c + 1]] > 1e-15).any()], dtype=int)
        if len(keep) < P.shape[1]:
            Ps[l] = P[:, keep]
            if l + 1 < len(Ps):
                Ps[l + 1] = Ps[l + 1].tocsr()[keep].tocsc()
        else:
            break
    return unknown, LHS, Auk, Ps


def solve(A, Ps, RHS, z0, known, known_val, tol, max_iter):
    if known is not None:
        unknown, LHS, Auk, Ps = eliminate(A, Ps, known)
        rhs = RHS[unknown] - Auk @ known_val
        z = z0[unknown].copy()
    else:
        LHS, rhs, z = A, RHS.copy(), z0.copy()
    As = [LHS.tocsr()]
    for P in Ps:
        As.append((P.T @ As[-1] @ P).tocsr())
    As[-1] = (As[-1] + 1e-12 * sp.eye(As[-1].shape[0])).tocsr()
    lus = spla.splu(As[-1].tocsc())
    r_his = []
    for it in range(max_iter):
        r = np.linalg.norm(rhs - LHS @ z)
        r_his.append(r)
        if r < tol:
            break
        if z.ndim == 1:
            z = vcycle(As, Ps, lus, rhs, z, 0)
        else:
            z = np.stack([vcycle(As, Ps, lus, rhs[:, c], z[:, c], 0) for c in range(z.shape[1])], axis=1)
    out = np.zeros_like(RHS)
    if known is not None:
        out[unknown] = z
        out[known] = known_val
    else:
        out = z
    return np.array(r_his), out, As


def sample_idx(n, m=48, seed=7):
    return np.sort(np.random.default_rng(seed).choice(n, size=min(m, n), replace=False))


if __name__ == "__main__":
    # G1: mean-curvature-flow system, 3 columns, tol 5e-7 (05_example_mean_curvature_flow/main.cpp:60,76)
    p = subdiv_problem(kind="mcf", k=3, n_sub=2)
    rh, z, As = solve(p["A"], p["Ps"], p["RHS"], p["z0"], None, None, 5e-7, 20)
    idx = sample_idx(z.shape[0])
    np.savez(os.path.join(HERE, "g1_mcf_k3.npz"), r_his=rh, z_norm=np.linalg.norm(z), idx=idx, z_samples=z[idx],
             level_nnz=np.array([a.nnz for a in As]), level_diag_sum=np.array([a.diagonal().sum() for a in As]))
    print("G1", rh)
    # G2: Poisson with the boundary loop pinned, tol 1e-10 (03_mg_solver/main.cpp:44-75 with 04's tolerance)
    p = subdiv_problem(kind="poisson", k=1, n_sub=2)
    rh, z, As = solve(p["A"], p["Ps"], p["RHS"][:, 0], p["z0"][:, 0], p["known"], p["known_val"][:, 0], 1e-10, 30)
    idx = sample_idx(z.shape[0])
    np.savez(os.path.join(HERE, "g2_poisson_bd.npz"), r_his=rh, z_norm=np.linalg.norm(z), idx=idx, z_samples=z[idx],
             level_rows=np.array([a.shape[0] for a in As]), level_nnz=np.array([a.nnz for a in As]))
    print("G2", rh)
    # G3: per-kernel vectors on a 2-level hierarchy (ogre_sim x1): SpMV, one GS sweep, PT r, P u, coarse solve
    p = subdiv_problem(kind="mcf", k=1, n_sub=1)
    A, P = p["A"].tocsr(), p["Ps"][0]
    n, nc = A.shape[0], P.shape[1]
    rng = np.random.default_rng(11)
    x, b, xc = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n), rng.uniform(-1, 1, nc)
    Ac = (P.T @ A @ P + 1e-12 * sp.eye(nc)).tocsc()
    np.savez(os.path.join(HERE, "g3_kernels.npz"), x=x, b=b, xc=xc, Ax=A @ x, gs1=gs_sweeps(A, b, x, 1),
             gs2=gs_sweeps(A, b, x, 2), PTx=P.T @ x, Pxc=P @ xc, coarse=xc + spla.spsolve(Ac, xc), Ac_nnz=Ac.nnz)
    print("G3 done")
