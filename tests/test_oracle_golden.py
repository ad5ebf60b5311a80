"""CPU: pin the C oracle against the restatement-derived golden vectors (tests/golden/make_golden.py).
The reference itself ships no tests/golden vectors and cannot be built here => parity unpinned beyond this."""
import os

import numpy as np
import pytest

from problems import subdiv_problem

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_g1_mcf_three_columns(oracle_mod):
    g = np.load(os.path.join(G, "g1_mcf_k3.npz"))
    p = subdiv_problem(kind="mcf", k=3, n_sub=2)
    orc = oracle_mod.OracleMG(p["Ps"])
    orc.precompute(p["A"])
    conv, z, rh = orc.solve(p["RHS"], p["z0"], tol=5e-7, max_iter=20)
    assert conv and len(rh) == len(g["r_his"])
    np.testing.assert_allclose(rh, g["r_his"], rtol=1e-9)
    np.testing.assert_allclose(z[g["idx"]], g["z_samples"], rtol=0, atol=1e-12)
    assert abs(np.linalg.norm(z) - g["z_norm"]) < 1e-11
    nnz = [orc.level_A(l).nnz for l in range(orc.n_levels)]
    assert nnz == list(g["level_nnz"])
    dsum = [orc.level_Adiag(l).sum() for l in range(orc.n_levels)]
    np.testing.assert_allclose(dsum, g["level_diag_sum"], rtol=1e-12)


def test_g2_poisson_boundary(oracle_mod):
    g = np.load(os.path.join(G, "g2_poisson_bd.npz"))
    p = subdiv_problem(kind="poisson", k=1, n_sub=2)
    orc = oracle_mod.OracleMG(p["Ps"])
    orc.precompute(p["A"], p["known"])
    conv, z, rh = orc.solve(p["RHS"], p["z0"], p["known_val"], tol=1e-10, max_iter=30)
    assert conv and len(rh) == len(g["r_his"])
    # residuals near 1e-11 sit at the rounding floor of ||RHS|| ~ 5e2: absolute slack 1e-14 * r_his[0]
    np.testing.assert_allclose(rh, g["r_his"], rtol=1e-7, atol=1e-14 * g["r_his"][0])
    np.testing.assert_allclose(z[g["idx"], 0], g["z_samples"], rtol=0, atol=1e-9 * g["z_norm"])
    assert [orc.rows(l) for l in range(orc.n_levels)] == list(g["level_rows"])
    assert [orc.level_A(l).nnz for l in range(orc.n_levels)] == list(g["level_nnz"])
    # constrained rows come back exactly as known_val (min_quad_with_fixed_mg.cpp:355)
    assert np.array_equal(z[p["known"], 0], p["known_val"][:, 0])


def test_g3_kernels(oracle_mod):
    g = np.load(os.path.join(G, "g3_kernels.npz"))
    p = subdiv_problem(kind="mcf", k=1, n_sub=1)
    orc = oracle_mod.OracleMG(p["Ps"])
    orc.precompute(p["A"])
    x, b, xc = g["x"], g["b"], g["xc"]
    np.testing.assert_allclose(orc.A(0, x)[:, 0], g["Ax"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(orc.relax(0, b, x, 1)[:, 0], g["gs1"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(orc.relax(0, b, x, 2)[:, 0], g["gs2"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(orc.restrict(0, x)[:, 0], g["PTx"], rtol=0, atol=1e-14)
    np.testing.assert_allclose(orc.prolong(0, xc)[:, 0], g["Pxc"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(orc.coarse_solve(xc, xc)[:, 0], g["coarse"], rtol=1e-9)
    assert orc.level_A(1).nnz == int(g["Ac_nnz"])


def test_outer_loop_bookkeeping(oracle_mod):
    """residual measured before each cycle; r_his has one entry per loop entry incl. the
    one that breaks; after max_iter cycles the final iterate's residual is never measured."""
    p = subdiv_problem(kind="mcf", k=1, n_sub=1)
    orc = oracle_mod.OracleMG(p["Ps"])
    orc.precompute(p["A"])
    conv, z, rh = orc.solve(p["RHS"], p["z0"], tol=1e-30, max_iter=3)
    assert not conv and len(rh) == 3
    conv, z, rh = orc.solve(p["RHS"], p["z0"], tol=1e30, max_iter=3)
    assert conv and len(rh) == 1 and np.array_equal(z, p["z0"])
    # k columns: Frobenius norm couples them; each column is an independent Gauss-Seidel
    p3 = subdiv_problem(kind="mcf", k=3, n_sub=1)
    o3 = oracle_mod.OracleMG(p3["Ps"])
    o3.precompute(p3["A"])
    u = o3.relax(0, p3["RHS"], p3["z0"], 2)
    for c in range(3):
        assert np.array_equal(u[:, c], o3.relax(0, p3["RHS"][:, c], p3["z0"][:, c], 2)[:, 0])
    prof = o3.profile()
    assert prof["MG: relaxation"][0] == 4


def test_all_core_mode_of_the_oracle_is_the_same_arithmetic(oracle_mod):
    """bench.py's `cpu_allcore` leg runs the oracle with OpenMP over the colour blocks of a colour-major numbering.  Rows of a
    block are independent, products run row-wise in the same ascending order: the iterate must be bit-identical to the
    sequential (reference-order) run on the same renumbered system; only the residual norm is reduced in another order."""
    import scipy.sparse as sp
    from problems import subdiv_problem
    p = subdiv_problem(kind="mcf", k=2, n_sub=2)
    o = oracle_mod.OracleMG(p["Ps"])
    o.precompute(p["A"])
    L = o.n_levels

    def greedy(A):
        A = A.tocsr()
        col = -np.ones(A.shape[0], int)
        for i in range(A.shape[0]):
            used = set(col[A.indices[A.indptr[i]:A.indptr[i + 1]]])
            c = 0
            while c in used:
                c += 1
            col[i] = c
        return col

    perms, cps = [], []
    for lv in range(L - 1):
        c = greedy(o.level_A(lv))
        perms.append(np.argsort(c, kind="stable"))
        cps.append(np.concatenate([[0], np.cumsum(np.bincount(c))]))
    perms.append(np.arange(o.rows(L - 1)))
    A0 = sp.csr_matrix(p["A"])[perms[0]][:, perms[0]]
    Ps = [sp.csr_matrix(p["Ps"][l])[perms[l]][:, perms[l + 1]] for l in range(L - 1)]
    o2 = oracle_mod.OracleMG(Ps)
    o2.precompute(A0)
    rhs, z0 = p["RHS"][perms[0]], p["z0"][perms[0]]
    seq = o2.solve(rhs, z0, tol=1e-10, max_iter=30)
    assert o2.set_parallel(cps, 4) >= 1
    par = o2.solve(rhs, z0, tol=1e-10, max_iter=30)
    assert seq[0] and par[0] and len(seq[2]) == len(par[2])
    assert np.array_equal(seq[1], par[1])
    assert np.allclose(seq[2], par[2], rtol=1e-10, atol=0)
    # blocks that are not independent sets are refused
    with pytest.raises(RuntimeError):
        o2.set_parallel([np.array([0, o2.rows(0)])] + cps[1:], 2)
    # and the renumbered problem is the same problem
    back = np.empty_like(seq[1]); back[perms[0]] = seq[1]
    ref = o.solve(p["RHS"], p["z0"], tol=1e-10, max_iter=30)
    assert np.linalg.norm(back - ref[1]) <= 1e-7 * np.linalg.norm(ref[1])


def test_oracle_jacobi_matches_a_numpy_restatement(oracle_mod):
    """orc_set_smoother(JACOBI): u <- u + omega ((b - (A - D) u) / d - u), all rows from the old iterate -- against dense-free numpy
    (summation order differs: 1e-14), and a hybrid V-cycle still contracts."""
    from problems import subdiv_problem
    import scipy.sparse as sp
    p = subdiv_problem(kind="mcf", k=2, n_sub=2)
    orc = oracle_mod.OracleMG(p["Ps"])
    orc.precompute(p["A"])
    rng = np.random.default_rng(0)
    for lv in range(orc.n_levels - 1):
        A = orc.level_A(lv).tocsr()
        d = A.diagonal()
        R = A - sp.diags(d)
        n = A.shape[0]
        b, u = rng.uniform(-1, 1, (n, 2)), rng.uniform(-1, 1, (n, 2))
        for omega in (0.8, 1.0):
            orc.set_smoother(lv, "jacobi", omega)
            got = orc.relax(lv, b, u, 3)
            ref = u.copy()
            for _ in range(3):
                t = (b - R @ ref) / d[:, None]
                ref = ref + omega * (t - ref)
            assert abs(got - ref).max() <= 1e-13 * abs(ref).max()
            orc.set_smoother(lv, "gs")
    assert np.array_equal(orc.relax(0, b0 := rng.uniform(-1, 1, (orc.rows(0), 1)), np.zeros((orc.rows(0), 1)), 1),
                          orc.relax(0, b0, np.zeros((orc.rows(0), 1)), 1))
    # hybrid: GS on level 0, Jacobi below
    for lv in range(1, orc.n_levels - 1):
        orc.set_smoother(lv, "jacobi", 0.8)
    conv, z, rh = orc.solve(p["RHS"], p["z0"], tol=1e-9, max_iter=40)
    assert conv and (np.diff(rh) < 0).all()
    # all-core mode runs the same Jacobi arithmetic
    colors = [np.array([0, orc.rows(lv)], dtype=np.int32) for lv in range(orc.n_levels - 1)]
    orc.L.orc_enable_parallel(orc.h, 1, 4)
    conv2, z2, rh2 = orc.solve(p["RHS"], p["z0"], tol=1e-9, max_iter=40)
    orc.set_sequential()
    assert len(rh2) == len(rh) and np.linalg.norm(z - z2) <= 1e-12 * np.linalg.norm(z)


def test_oracle_chebyshev_matches_a_numpy_restatement(oracle_mod):
    """orc_set_smoother(CHEBY): relax(iters) = one Chebyshev-Jacobi polynomial of degree iters + 1 on [fraction lam, lam], lam the
    Gershgorin bound -- against the textbook three-term recurrence in numpy; and its optimality property: on the target interval the
    polynomial is bounded by 1 / T_deg(sigma)."""
    from problems import subdiv_problem
    import scipy.sparse as sp
    p = subdiv_problem(kind="poisson", k=2, n_sub=2)
    orc = oracle_mod.OracleMG(p["Ps"])
    orc.precompute(p["A"], p["known"])
    rng = np.random.default_rng(1)
    for lv in range(orc.n_levels - 1):
        A = orc.level_A(lv).tocsr()
        d = A.diagonal()
        n = A.shape[0]
        lam = float((abs(A).sum(axis=0).A1 / d).max())
        assert abs(orc.spectral_bound(lv) - lam) <= 1e-14 * lam
        b, u = rng.uniform(-1, 1, (n, 2)), rng.uniform(-1, 1, (n, 2))
        for frac, iters in ((0.1, 2), (0.3, 1), (0.1, 4)):
            orc.set_smoother(lv, "chebyshev", frac)
            got = orc.relax(lv, b, u, iters)
            lmin = lam * frac
            theta, delta = (lam + lmin) / 2, (lam - lmin) / 2
            sigma = theta / delta
            rho = 1 / sigma
            x = u.copy()
            r = (b - A @ x) / d[:, None]
            dd = r / theta
            x = x + dd
            for _ in range(iters):
                rho_new = 1 / (2 * sigma - rho)
                r = (b - A @ x) / d[:, None]
                dd = rho_new * rho * dd + (2 * rho_new / delta) * r
                x = x + dd
                rho = rho_new
            assert abs(got - x).max() <= 1e-12 * abs(x).max()
            orc.set_smoother(lv, "gs")
    # error propagation of the degree-3 polynomial on eigenvectors of D^-1 A inside the interval: |p(lambda)| <= 1 / T_3(sigma)
    p = subdiv_problem(mesh="torus", kind="mcf", k=1, n_sub=1)
    orc = oracle_mod.OracleMG(p["Ps"])
    orc.precompute(p["A"])
    lv = 0
    A = orc.level_A(lv).toarray()
    d = np.diag(A)
    w, Vv = np.linalg.eig(A / d[:, None])
    w = w.real
    lam, frac = orc.spectral_bound(lv), 0.1
    sigma = (1 + frac) / (1 - frac)
    bound = 1.0 / (4 * sigma ** 3 - 3 * sigma)
    orc.set_smoother(lv, "chebyshev", frac)
    for idx in np.argsort(w)[[-1, -len(w) // 4, -len(w) // 2]]:
        if w[idx] < frac * lam:
            continue
        e = Vv[:, idx].real[:, None].copy()
        out = orc.relax(lv, np.zeros_like(e), e, 2)         # b = 0: the iterate IS the error
        assert np.linalg.norm(out) <= (bound + 1e-9) * np.linalg.norm(e) * 1.0001
    orc.set_smoother(lv, "gs")


def test_mesh_operators_against_closed_form_answers(smg_mod):
    """Caller-side numerics (igl::cotmatrix, igl::massmatrix, igl::boundary_loop, normalize_unit_area) are third-party semantics that
    neither the product's host C++ (csrc/smg_mesh.cpp) nor the checker (oracle/mesh_np.py) can be compared with libigl here; both are held
    against configurations whose answers are known in closed form -- independent of either implementation:
      * unit squares split along one diagonal: edges along the axes get 1/2 (cot 45 + cot 45) = 1, diagonals (opposite two right angles)
        0; an interior vertex has Voronoi = barycentric area h^2; cotmatrix is negative semi-definite with zero row sums;
      * equilateral triangles: every interior edge 1/2 (cot 60 + cot 60) = 1/sqrt(3); interior Voronoi area = sqrt(3)/2 h^2;
      * one obtuse triangle: mixed-Voronoi areas A/2 at the obtuse corner, A/4 at the others; barycentric A/3 each;
      * the boundary loop of the square grid is its perimeter, and normalize_unit_area leaves total area 1, centroid x = y = 0, min z = 0."""
    from oracle import mesh_np as M
    mesh = smg_mod.mesh
    n, h = 7, 0.25
    xs, ys = np.meshgrid(np.arange(n) * h, np.arange(n) * h, indexing="ij")
    V = np.stack([xs.ravel(), ys.ravel(), 0 * xs.ravel()], axis=1)
    idx = lambda i, j: i * n + j
    F = np.array([(idx(i, j), idx(i + 1, j), idx(i + 1, j + 1)) for i in range(n - 1) for j in range(n - 1)] +
                 [(idx(i, j), idx(i + 1, j + 1), idx(i, j + 1)) for i in range(n - 1) for j in range(n - 1)], dtype=np.int32)
    for name, cot, mass, bl in (("product", mesh.cotmatrix, mesh.massmatrix, mesh.boundary_loop), ("checker", M.cotmatrix, M.massmatrix, M.boundary_loop)):
        L = cot(V, F).tocsr()
        c = idx(3, 3)
        assert abs(L[c, idx(4, 3)] - 1.0) < 1e-14 and abs(L[c, idx(3, 4)] - 1.0) < 1e-14, name      # axis edges
        assert abs(L[c, idx(4, 4)]) < 1e-14 and L[c, idx(4, 2)] == 0.0, name                           # the diagonal / no edge
        assert abs(L[c, c] + 4.0) < 1e-14 and abs(L.sum(axis=1)).max() < 1e-13, name                   # -sum of the off-diagonals
        assert abs(L - L.T).max() < 1e-15, name
        for kind in ("voronoi", "barycentric"):
            d = mass(V, F, kind).diagonal()
            assert abs(d[c] - h * h) < 1e-15 and abs(d.sum() - ((n - 1) * h) ** 2) < 1e-13, (name, kind)
        b = np.asarray(bl(F))
        on = {idx(i, j) for i in range(n) for j in range(n) if i in (0, n - 1) or j in (0, n - 1)}
        assert set(b.tolist()) == on and len(b) == 4 * (n - 1), name
    # equilateral grid
    m = 6
    P = np.array([[i + 0.5 * j, j * np.sqrt(3) / 2, 0.0] for i in range(m) for j in range(m)]) * h
    idq = lambda i, j: i * m + j
    Fq = np.array([(idq(i, j), idq(i + 1, j), idq(i, j + 1)) for i in range(m - 1) for j in range(m - 1)] +
                  [(idq(i + 1, j), idq(i + 1, j + 1), idq(i, j + 1)) for i in range(m - 1) for j in range(m - 1)], dtype=np.int32)
    for name, cot, mass in (("product", mesh.cotmatrix, mesh.massmatrix), ("checker", M.cotmatrix, M.massmatrix)):
        L = cot(P, Fq).tocsr()
        c = idq(2, 2)
        nb = [idq(3, 2), idq(2, 3), idq(1, 3), idq(1, 2), idq(2, 1), idq(3, 1)]
        assert all(abs(L[c, j] - 1 / np.sqrt(3)) < 1e-14 for j in nb) and abs(L[c, c] + 6 / np.sqrt(3)) < 1e-13, name
        assert abs(mass(P, Fq, "voronoi").diagonal()[c] - np.sqrt(3) / 2 * h * h) < 1e-15, name
        assert abs(mass(P, Fq, "barycentric").diagonal()[c] - np.sqrt(3) / 2 * h * h) < 1e-15, name
    # one obtuse triangle (angle at vertex 0 > 90 degrees), plus a far-away second triangle so that the mesh has 4+ entries per call
    T = np.array([[0.0, 0.0, 0.0], [2.0, 0.3, 0.0], [-1.5, 0.4, 0.0]])
    Ft = np.array([[0, 1, 2]], dtype=np.int32)
    area = 0.5 * abs(np.cross(T[1] - T[0], T[2] - T[0])[2])
    for name, mass in (("product", mesh.massmatrix), ("checker", M.massmatrix)):
        dv = mass(T, Ft, "voronoi").diagonal()
        assert np.allclose(dv, [area / 2, area / 4, area / 4], rtol=1e-14), (name, dv)
        assert np.allclose(mass(T, Ft, "barycentric").diagonal(), area / 3, rtol=1e-14), name
    # normalize_unit_area (src/normalize_unit_area.cpp:13-24): total area 1, x / y centred, lowest point on z = 0
    rng = np.random.default_rng(0)
    W = V + 0.05 * rng.uniform(-1, 1, V.shape) + np.array([3.0, -2.0, 5.0])
    for name, norm, mass in (("product", mesh.normalize_unit_area, mesh.massmatrix), ("checker", M.normalize_unit_area, M.massmatrix)):
        U = norm(W, F)
        assert abs(mass(U, F, "barycentric").diagonal().sum() - 1.0) < 1e-13, name
        assert abs(U[:, 0].mean()) < 1e-13 and abs(U[:, 1].mean()) < 1e-13 and abs(U[:, 2].min()) < 1e-13, name
