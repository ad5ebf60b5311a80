"""Shared workload builders for the tests: the reference demos' systems on small meshes.
Caller-side numerics come from the numpy restatement (oracle/mesh_np.py) so that the oracle and the HIP path
see bit-identical inputs."""
import numpy as np
import scipy.sparse as sp

from oracle import mesh_np as M


def subdiv_problem(mesh="ogre_sim.smgm", n_sub=2, kind="mcf", k=1, seed=0, n_pins=0):
    """Returns dict(A, Ps, RHS, z0, known, known_val, V, F).
    kind 'mcf'    : LHS = M_bary - 0.01 L, RHS = M * X (05_example_mean_curvature_flow/main.cpp:66-69), no constraints
    kind 'poisson': A = -L, B = M_voronoi * 1 (03_mg_solver/main.cpp:44-61), constraints = boundary loop
                    (or n_pins random vertices on a closed mesh, 04_mg_solver_nobd/main.cpp:73-94)."""
    rng = np.random.default_rng(seed)
    if mesh == "torus":
        V, F = M.torus(24, 16)
    else:
        V, F = M.read_smgm(mesh)
    V = M.normalize_unit_area(V, F)
    Vf, Ff, Ps = M.subdivision_hierarchy(V, F, n_sub)
    n = Vf.shape[0]
    L = M.cotmatrix(Vf, Ff)
    out = dict(V=Vf, F=Ff, Ps=Ps)
    if kind == "mcf":
        Mb = M.massmatrix(Vf, Ff, "barycentric")
        A = (Mb - 0.01 * L).tocsr()
        X = Vf if k == 3 else np.concatenate([Vf, rng.uniform(-1, 1, (n, max(k - 3, 0)))], axis=1)[:, :k]
        out.update(A=A, RHS=np.asfortranarray(Mb @ X), z0=np.asfortranarray(X.copy()), known=None, known_val=None)
    else:
        A = (-L).tocsr()
        Mv = M.massmatrix(Vf, Ff, "voronoi")
        b = M.boundary_loop(Ff)
        if n_pins > 0 or len(b) == 0:
            b = rng.choice(n, size=max(n_pins, 8), replace=False).astype(np.int32)
        B = np.repeat((Mv @ np.ones(n))[:, None], k, axis=1)
        if k > 1:
            B = B * rng.uniform(0.5, 1.5, (1, k))
        bval = np.zeros((len(b), k))
        B[b, :] = bval
        z0 = rng.uniform(-1, 1, (n, k))
        out.update(A=A, RHS=np.asfortranarray(B), z0=np.asfortranarray(z0), known=b, known_val=bval)
    A.sort_indices()
    return out
