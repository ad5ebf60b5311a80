"""GPU (-m gpu): the damped-Jacobi smoother and the per-level GS/Jacobi hybrid (include/smg.h: SMG_SMOOTH_*).

The reference's relax() is Gauss-Seidel only (src/mg_VCycle.cpp:113-178); Jacobi fills the same slot (BASELINE.json north_star:
"Gauss-Seidel/Jacobi smoothing").  Its arithmetic is defined once, in oracle/smg_oracle.h (orc_set_smoother), and the HIP kernel
(k_sell<SELL_JACOBI>) follows it bit for bit:
  K/S-level  one, two, three sweeps: BIT-EXACT against the oracle on the level's matrix in the device numbering, any k;
  cycle      a Jacobi sweep does not depend on the numbering (only the per-row summation order does): an all-Jacobi V-cycle / solve
             agrees with the oracle IN THE CALLER'S NUMBERING to rounding (1e-10), iteration for iteration;
  hybrid     Gauss-Seidel levels differ from the lexicographic oracle by the sweep order: both converge, counts within +-2.
"""
import hashlib
import os

import numpy as np
import pytest

from problems import subdiv_problem
from test_gpu_parity import build, oracle_on_device_numbering, smg  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind,k", [("mcf", 1), ("poisson", 2), ("mcf", 3), ("mcf", 5), ("mcf", 8), ("poisson", 27), ("mcf", 64)])
def test_jacobi_sweeps_bit_exact_in_device_numbering(smg, oracle_mod, kind, k):
    p, mg, orc = build(smg, oracle_mod, kind=kind, k=k, n_sub=2)
    rng = np.random.default_rng(11)
    for omega in (0.8, 1.0, 0.6180339887):
        mg.set_smoother("jacobi", omega)
        for lv in range(mg.n_levels - 1):
            n = mg.rows(lv)
            perm = mg.perm(lv)
            oi = oracle_on_device_numbering(oracle_mod, mg, lv)
            oi.set_smoother(0, "jacobi", omega)
            x, b = rng.uniform(-1, 1, (n, k)), rng.uniform(-1, 1, (n, k))
            for iters in (1, 2, 3):     # odd counts end in the second buffer and are copied back
                got = mg.relax(lv, b, x, iters)[perm]
                ref = oi.relax(0, b[perm], x[perm], iters)
                assert np.array_equal(got, ref), "Jacobi sweep not bit-exact: level %d, %d sweeps, omega %g" % (lv, iters, omega)
            # and against the oracle in the caller's numbering: the sweep itself is numbering-independent
            ref = orc_relax_jacobi(orc, lv, b, x, 2, omega)
            got = mg.relax(lv, b, x, 2)
            assert abs(got - ref).max() <= 1e-13 * abs(ref).max()
    mg.set_smoother("gs")
    # back on Gauss-Seidel the level relaxes exactly as before
    lv = 0
    oi = oracle_on_device_numbering(oracle_mod, mg, lv)
    perm = mg.perm(lv)
    x, b = rng.uniform(-1, 1, (mg.rows(lv), k)), rng.uniform(-1, 1, (mg.rows(lv), k))
    assert np.array_equal(mg.relax(lv, b, x, 2)[perm], oi.relax(0, b[perm], x[perm], 2))


@pytest.mark.parametrize("kind,k", [("mcf", 1), ("poisson", 2), ("mcf", 3), ("mcf", 8), ("poisson", 27), ("mcf", 64)])
def test_chebyshev_polynomials_bit_exact_in_device_numbering(smg, oracle_mod, kind, k):
    """relax(iters) = one Chebyshev-Jacobi polynomial of degree iters + 1 (include/smg.h): the Gershgorin bound and every step of the
    recurrence agree bit for bit with the oracle on the level's matrix in the device numbering."""
    p, mg, orc = build(smg, oracle_mod, kind=kind, k=k, n_sub=2)
    rng = np.random.default_rng(12)
    for frac in (0.1, 0.3):
        mg.set_smoother("chebyshev", cheby_fraction=frac)
        for lv in range(mg.n_levels - 1):
            n, perm = mg.rows(lv), mg.perm(lv)
            oi = oracle_on_device_numbering(oracle_mod, mg, lv)
            oi.set_smoother(0, "chebyshev", frac)
            assert mg.spectral_bound(lv) == oi.spectral_bound(0) and 1.0 < mg.spectral_bound(lv) < 4.0
            assert abs(mg.spectral_bound(lv) - orc.spectral_bound(lv)) <= 1e-14 * orc.spectral_bound(lv)
            x, b = rng.uniform(-1, 1, (n, k)), rng.uniform(-1, 1, (n, k))
            for iters in (1, 2, 3):     # degree 2, 3, 4: both parities of the ping-pong
                got = mg.relax(lv, b, x, iters)[perm]
                ref = oi.relax(0, b[perm], x[perm], iters)
                assert np.array_equal(got, ref), "Chebyshev polynomial not bit-exact: level %d, degree %d, fraction %g" % (lv, iters + 1, frac)
    mg.set_smoother("gs")


def orc_relax_jacobi(orc, lv, b, x, iters, omega):
    orc.set_smoother(lv, "jacobi", omega)
    out = orc.relax(lv, b, x, iters)
    orc.set_smoother(lv, "gs")
    return out


def _set_all(orc, n_levels, kind, omega, jacobi_rows=None):
    """the oracle's per-level smoothers for a libsmg smoother selection (omega: damping, or the Chebyshev interval fraction)"""
    base = "chebyshev" if "chebyshev" in kind else "jacobi"
    for lv in range(n_levels - 1):
        small = kind in ("jacobi", "chebyshev") or (kind.startswith("hybrid") and orc.rows(lv) <= jacobi_rows)
        orc.set_smoother(lv, base if small else "gs", omega)


@pytest.mark.parametrize("kind,k,pre,post", [("mcf", 1, 2, 2), ("poisson", 1, 2, 2), ("mcf", 3, 1, 2), ("mcf", 2, 3, 1), ("mcf", 8, 2, 1), ("poisson", 1, 0, 3)])
def test_all_jacobi_vcycle_matches_oracle_in_caller_numbering(smg, oracle_mod, kind, k, pre, post):
    """Every parity of pre/post sweep counts (the iterate ping-pongs between two buffers and must end in u), with and without
    the restriction launch producing the coarse level's first sweep."""
    p, mg, orc = build(smg, oracle_mod, kind=kind, k=k, n_sub=3)
    assert mg.n_levels >= 4
    omega = 0.7
    mg.set_smoother("jacobi", omega)
    _set_all(orc, mg.n_levels, "jacobi", omega)
    rng = np.random.default_rng(5)
    n = mg.rows(0)
    B, u = rng.uniform(-1, 1, (n, k)), rng.uniform(-1, 1, (n, k))
    got = mg.vcycle(B, u, pre=pre, post=post)
    ref = orc.vcycle(B, u, pre=pre, post=post)
    assert abs(got - ref).max() <= 1e-10 * abs(ref).max()
    # a cycle started on a coarser level
    n1 = mg.rows(1)
    B1, u1 = rng.uniform(-1, 1, (n1, k)), rng.uniform(-1, 1, (n1, k))
    got, ref = mg.vcycle(B1, u1, lv=1, pre=pre, post=post), orc.vcycle(B1, u1, lv=1, pre=pre, post=post)
    assert abs(got - ref).max() <= 1e-10 * abs(ref).max()


@pytest.mark.parametrize("kind,k,pre,post", [("mcf", 1, 2, 2), ("poisson", 2, 2, 2), ("mcf", 3, 1, 2), ("mcf", 2, 2, 1), ("mcf", 8, 3, 3), ("poisson", 1, 0, 2)])
def test_all_chebyshev_vcycle_matches_oracle_in_caller_numbering(smg, oracle_mod, kind, k, pre, post):
    p, mg, orc = build(smg, oracle_mod, kind=kind, k=k, n_sub=3)
    mg.set_smoother("chebyshev", cheby_fraction=0.1)
    _set_all(orc, mg.n_levels, "chebyshev", 0.1)
    rng = np.random.default_rng(6)
    n = mg.rows(0)
    B, u = rng.uniform(-1, 1, (n, k)), rng.uniform(-1, 1, (n, k))
    got, ref = mg.vcycle(B, u, pre=pre, post=post), orc.vcycle(B, u, pre=pre, post=post)
    assert abs(got - ref).max() <= 1e-10 * abs(ref).max()
    n1 = mg.rows(1)
    B1, u1 = rng.uniform(-1, 1, (n1, k)), rng.uniform(-1, 1, (n1, k))
    got, ref = mg.vcycle(B1, u1, lv=1, pre=pre, post=post), orc.vcycle(B1, u1, lv=1, pre=pre, post=post)
    assert abs(got - ref).max() <= 1e-10 * abs(ref).max()


@pytest.mark.parametrize("kind,k,tol", [("mcf", 1, 1e-10), ("poisson", 2, 1e-9), ("mcf", 3, 5e-7)])
def test_hybrid_chebyshev_solve(smg, oracle_mod, kind, k, tol):
    """Gauss-Seidel on the finest level, Chebyshev-Jacobi below: converges like the oracle with the same selection (counts within +-2:
    the GS level differs by the sweep order) and needs no more cycles than Gauss-Seidel everywhere + 1."""
    p, mg, orc = build(smg, oracle_mod, kind=kind, k=k, n_sub=3)
    thr = mg.rows(1)
    _set_all(orc, mg.n_levels, "hybrid_chebyshev", 0.1, thr)
    opts = smg.SolveOpts(tol=tol, max_iter=60, smoother="hybrid_chebyshev", jacobi_max_rows=thr)
    conv, z, rh = mg.solve(p["RHS"], p["z0"], p["known_val"], opts)
    conv2, z2, rh2 = orc.solve(p["RHS"], p["z0"], p["known_val"], tol=tol, max_iter=60)
    assert conv and conv2 and abs(len(rh) - len(rh2)) <= 2
    assert np.linalg.norm(z - z2) <= (1e-8 if tol <= 1e-9 else 1e-3) * np.linalg.norm(z2)
    conv3, z3, rh3 = mg.solve(p["RHS"], p["z0"], p["known_val"], smg.SolveOpts(tol=tol, max_iter=60))
    assert conv3 and len(rh) <= len(rh3) + 1
    # all-Chebyshev: numbering-independent, tracks the oracle iteration by iteration
    _set_all(orc, mg.n_levels, "chebyshev", 0.1)
    conv4, z4, rh4 = mg.solve(p["RHS"], p["z0"], p["known_val"], smg.SolveOpts(tol=tol, max_iter=60, smoother="chebyshev"))
    conv5, z5, rh5 = orc.solve(p["RHS"], p["z0"], p["known_val"], tol=tol, max_iter=60)
    assert conv4 and conv5 and len(rh4) == len(rh5)
    np.testing.assert_allclose(rh4, rh5, rtol=1e-6)
    # mixed precision and eager launches
    c6, z6, r6 = mg.solve(p["RHS"], p["z0"], p["known_val"], smg.SolveOpts(tol=tol, max_iter=60, smoother="hybrid_chebyshev", jacobi_max_rows=thr, precision="mixed"))
    assert c6 and abs(len(r6) - len(rh)) <= 2 and np.linalg.norm(z6 - z) <= 1e-6 * np.linalg.norm(z)
    c7, z7, r7 = mg.solve(p["RHS"], p["z0"], p["known_val"], smg.SolveOpts(tol=tol, max_iter=60, smoother="hybrid_chebyshev", jacobi_max_rows=thr, use_graph=0))
    assert np.array_equal(z7, z) and np.array_equal(r7, rh)


@pytest.mark.parametrize("kind,k,tol", [("mcf", 1, 1e-10), ("poisson", 1, 1e-9), ("mcf", 3, 5e-7)])
def test_all_jacobi_solve_tracks_the_oracle_iteration_by_iteration(smg, oracle_mod, kind, k, tol):
    p, mg, orc = build(smg, oracle_mod, kind=kind, k=k, n_sub=2)
    omega = 0.7
    _set_all(orc, mg.n_levels, "jacobi", omega)
    opts = smg.SolveOpts(tol=tol, max_iter=100, smoother="jacobi", omega=omega)
    conv, z, rh = mg.solve(p["RHS"], p["z0"], p["known_val"], opts)
    conv2, z2, rh2 = orc.solve(p["RHS"], p["z0"], p["known_val"], tol=tol, max_iter=100)
    assert conv and conv2 and len(rh) == len(rh2)
    np.testing.assert_allclose(rh, rh2, rtol=1e-6)
    assert np.linalg.norm(z - z2) <= 1e-9 * np.linalg.norm(z2)
    if p["known"] is not None:
        assert np.array_equal(z[p["known"]], p["known_val"])


@pytest.mark.parametrize("kind,k,tol", [("mcf", 1, 1e-10), ("poisson", 2, 1e-9), ("mcf", 3, 5e-7)])
def test_hybrid_solve_converges_like_the_oracle_hybrid(smg, oracle_mod, kind, k, tol):
    p, mg, orc = build(smg, oracle_mod, kind=kind, k=k, n_sub=3)
    rows = [mg.rows(l) for l in range(mg.n_levels)]
    thr = rows[1]                                   # Jacobi from level 1 on, Gauss-Seidel on the finest
    _set_all(orc, mg.n_levels, "hybrid", 0.8, thr)
    opts = smg.SolveOpts(tol=tol, max_iter=60, smoother="hybrid", omega=0.8, jacobi_max_rows=thr)
    conv, z, rh = mg.solve(p["RHS"], p["z0"], p["known_val"], opts)
    conv2, z2, rh2 = orc.solve(p["RHS"], p["z0"], p["known_val"], tol=tol, max_iter=60)
    assert conv and conv2 and abs(len(rh) - len(rh2)) <= 2
    assert abs(rh[0] - rh2[0]) <= 1e-12 * rh2[0]
    assert np.linalg.norm(z - z2) <= (1e-8 if tol <= 1e-9 else 1e-3) * np.linalg.norm(z2)
    # ... and needs at most a few more cycles than Gauss-Seidel everywhere (that is the point of keeping GS on the big level)
    conv3, z3, rh3 = mg.solve(p["RHS"], p["z0"], p["known_val"], smg.SolveOpts(tol=tol, max_iter=60))
    assert conv3 and len(rh) <= len(rh3) + 4
    # the default is still the reference's smoother: same bits as before the hybrid solve
    conv4, z4, rh4 = mg.solve(p["RHS"], p["z0"], p["known_val"], smg.SolveOpts(tol=tol, max_iter=60, smoother="gs"))
    assert np.array_equal(z3, z4) and np.array_equal(rh3, rh4)


def test_two_level_cycle_with_jacobi_in_device_numbering(smg, oracle_mod):
    """Level L-2 -> coarsest, Jacobi smoothed: everything but the dense coarse solve is bit-exact (as for Gauss-Seidel)."""
    p, mg, orc = build(smg, oracle_mod, kind="poisson", k=2, n_sub=2)
    lv = mg.n_levels - 2
    oi = oracle_on_device_numbering(oracle_mod, mg, lv)
    oi.set_smoother(0, "jacobi", 0.8)
    mg.set_smoother("jacobi", 0.8)
    rng = np.random.default_rng(9)
    n, perm = mg.rows(lv), mg.perm(lv)
    B, u = rng.uniform(-1, 1, (n, 2)), rng.uniform(-1, 1, (n, 2))
    got = mg.vcycle(B, u, lv=lv)[perm]
    ref = oi.vcycle(B[perm], u[perm], lv=0)
    assert abs(got - ref).max() <= 1e-11 * abs(ref).max()


def test_mixed_precision_and_split_phase_with_hybrid(smg, oracle_mod):
    p, mg, orc = build(smg, oracle_mod, kind="mcf", k=3, n_sub=3)
    thr = mg.rows(1)
    o64 = smg.SolveOpts(tol=1e-10, max_iter=60, smoother="hybrid", jacobi_max_rows=thr)
    omx = smg.SolveOpts(tol=1e-10, max_iter=60, smoother="hybrid", jacobi_max_rows=thr, precision="mixed")
    c1, z1, r1 = mg.solve(p["RHS"], p["z0"], None, o64)
    c2, z2, r2 = mg.solve(p["RHS"], p["z0"], None, omx)
    assert c1 and c2 and abs(len(r1) - len(r2)) <= 2
    assert np.linalg.norm(z1 - z2) <= 1e-8 * np.linalg.norm(z1)
    # eager launches == graph replay, bit for bit (the ping-pong buffers are baked into the graph)
    c3, z3, r3 = mg.solve(p["RHS"], p["z0"], None, smg.SolveOpts(tol=1e-10, max_iter=60, smoother="hybrid", jacobi_max_rows=thr, use_graph=0))
    assert np.array_equal(z1, z3) and np.array_equal(r1, r3)
    # changing omega re-captures the graph (the factor is a kernel argument)
    c4, z4, r4 = mg.solve(p["RHS"], p["z0"], None, smg.SolveOpts(tol=1e-10, max_iter=60, smoother="hybrid", jacobi_max_rows=thr, omega=0.6))
    c5, z5, r5 = mg.solve(p["RHS"], p["z0"], None, smg.SolveOpts(tol=1e-10, max_iter=60, smoother="hybrid", jacobi_max_rows=thr, omega=0.6, use_graph=0))
    assert c4 and not np.array_equal(z1, z4) and np.array_equal(z4, z5) and np.array_equal(r4, r5)


_CHILD = r"""
import hashlib, os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np
import surface_multigrid_code_amd as smg
from problems import subdiv_problem
for kind, k, sm in (("mcf", 1, "jacobi"), ("mcf", 3, "hybrid"), ("poisson", 2, "hybrid"), ("poisson", 9, "jacobi"), ("mcf", 1, "chebyshev"),
                    ("poisson", 2, "hybrid_chebyshev"), ("mcf", 9, "hybrid_chebyshev")):
    p = subdiv_problem(kind=kind, k=k, n_sub=3)
    mg = smg.Hierarchy.from_prolongs(p["Ps"])
    mg.precompute(p["A"], p["known"])
    thr = mg.rows(1)
    mg.set_smoother(sm, 0.8, thr)
    rng = np.random.default_rng(3)
    n = mg.rows(0)
    B, u = rng.uniform(-1, 1, (n, k)), rng.uniform(-1, 1, (n, k))
    v = mg.vcycle(B, u)
    conv, z, rh = mg.solve(p["RHS"], p["z0"], p["known_val"], smg.SolveOpts(tol=1e-9, max_iter=60, smoother=sm, jacobi_max_rows=thr))
    print(kind, k, mg.n_levels, hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest(),
          hashlib.sha256(np.ascontiguousarray(z).tobytes()).hexdigest(), len(rh))
"""


def test_fused_first_jacobi_sweep_does_not_change_a_bit(smg):
    """The restriction launch writing the coarse level's first Jacobi sweep (0 + omega (rc_i / a_ii - 0), all rows) is the sweep's own
    arithmetic: cycles and solves give identical bits with SMG_FUSE_FIRST=0."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for off in (False, True):
        env = dict(os.environ)
        if off:
            env.update(SMG_FUSE_FIRST="0")
        r = subprocess.run([sys.executable, "-c", _CHILD, root], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith(("mcf", "poisson"))]
        assert len(lines) == 7, r.stdout
        outs.append(lines)
    assert outs[0] == outs[1]


_CHILD_HEAD = r"""
import hashlib, os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np
import surface_multigrid_code_amd as smg
from problems import subdiv_problem
for kind, k, sm, pre in (("mcf", 1, "gs", 2), ("poisson", 3, "gs", 2), ("mcf", 9, "gs", 3), ("mcf", 64, "gs", 2), ("poisson", 1, "jacobi", 1),
                         ("mcf", 12, "jacobi", 2), ("mcf", 1, "chebyshev", 2), ("poisson", 17, "chebyshev", 1), ("mcf", 2, "hybrid_chebyshev", 2),
                         ("poisson", 1, "hybrid", 2), ("mcf", 1, "gs", 1)):
    p = subdiv_problem(kind=kind, k=k, n_sub=3)
    mg = smg.Hierarchy.from_prolongs(p["Ps"])
    mg.precompute(p["A"], p["known"])
    thr = mg.rows(1)
    conv, z, rh = mg.solve(p["RHS"], p["z0"], p["known_val"], smg.SolveOpts(tol=1e-9, max_iter=80, pre=pre, smoother=sm, jacobi_max_rows=thr))
    conv2, z2, rh2 = mg.solve(p["RHS"], p["z0"], p["known_val"], smg.SolveOpts(tol=1e-9, max_iter=80, pre=pre, smoother=sm, jacobi_max_rows=thr, use_graph=0))
    assert np.array_equal(z, z2) and np.array_equal(rh, rh2)
    print("case", kind, k, sm, int(conv), hashlib.sha256(np.ascontiguousarray(z).tobytes()).hexdigest(), " ".join(repr(float(x)) for x in rh))
"""


def test_outer_residual_folded_into_the_first_sweep_changes_nothing_but_rounding_of_the_norm(smg):
    """SELL_*_HEAD (csrc/smg_device.hpp): the first pre-smoothing sweep of level 0 also forms the outer residual of the iterate it starts
    from and leaves that iterate intact.  Against SMG_FUSE_HEAD=0 (separate residual launch, in-place sweeps): the solutions are
    bit-identical (same sweeps, same break iteration), the residual history agrees to the rounding of the sum of squares (the per-row
    residuals are the same bits; their squares are summed colour by colour instead of in launch order)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for off in (False, True):
        env = dict(os.environ)
        if off:
            env.update(SMG_FUSE_HEAD="0")
        r = subprocess.run([sys.executable, "-c", _CHILD_HEAD, root], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [ln.split() for ln in r.stdout.splitlines() if ln.startswith("case")]
        assert len(lines) == 11, r.stdout
        outs.append(lines)
    for a, b in zip(*outs):
        assert a[:6] == b[:6], (a[:6], b[:6])                      # same case, converged flag, solution bits
        ra, rb = np.array([float(x) for x in a[6:]]), np.array([float(x) for x in b[6:]])
        assert len(ra) == len(rb) and a[4] == "1"
        np.testing.assert_allclose(ra, rb, rtol=1e-13)


# ----------------------------------------------------------------------------------------------- boundary clean-ups
def test_single_level_hierarchy_goes_straight_to_the_coarse_solve(smg, oracle_mod):
    """mg.size() == 1: mg_VCycle is coarseSolve only, u += LDLT.solve(B) (src/mg_VCycle.cpp:28-33) -- exact after one cycle from
    z0 = 0, and (reference behaviour) NOT a fixed point iteration otherwise."""
    p = subdiv_problem(kind="mcf", k=2, n_sub=0)
    A = p["A"]
    mg = smg.Hierarchy(1)
    mg.precompute(A, None)
    orc = oracle_mod.OracleMG([])
    orc.precompute(A, None)
    z0 = np.zeros_like(p["z0"])
    conv, z, rh = mg.solve(p["RHS"], z0, None, smg.SolveOpts(tol=1e-9, max_iter=5))
    conv2, z2, rh2 = orc.solve(p["RHS"], z0, None, tol=1e-9, max_iter=5)
    assert conv and conv2 and len(rh) == len(rh2) == 2
    assert np.linalg.norm(z - z2) <= 1e-9 * np.linalg.norm(z2)
    # z0 != 0: u + A^-1 b is what the reference computes, too
    conv, z, rh = mg.solve(p["RHS"], p["z0"], None, smg.SolveOpts(tol=1e-9, max_iter=3))
    conv2, z2, rh2 = orc.solve(p["RHS"], p["z0"], None, tol=1e-9, max_iter=3)
    assert conv == conv2 and len(rh) == len(rh2)
    np.testing.assert_allclose(rh, rh2, rtol=1e-7)
    # with constraints (the reference's `known` overload indexes mg[1] unconditionally, .cpp:185: a single level is out of its
    # range; libsmg just eliminates the rows): one cycle from 0 is the direct solve of the reduced system
    import scipy.sparse.linalg as spla
    pp = subdiv_problem(kind="poisson", k=1, n_sub=0)
    mg2 = smg.Hierarchy(1)
    mg2.precompute(pp["A"], pp["known"])
    conv, z, rh = mg2.solve(pp["RHS"], np.zeros_like(pp["z0"]), pp["known_val"], smg.SolveOpts(tol=1e-9, max_iter=4))
    unk = mg2.unknown()
    Auu = pp["A"].tocsr()[unk][:, unk]
    ref = spla.spsolve(Auu.tocsc(), pp["RHS"][unk, 0])
    assert conv and len(rh) == 2 and np.linalg.norm(z[unk, 0] - ref) <= 1e-8 * np.linalg.norm(ref)


def test_more_than_1024_iterations(smg, oracle_mod):
    """maxIter is unbounded in the reference (src/min_quad_with_fixed_mg.cpp:77): the device-side history is sized from it."""
    p, mg, orc = build(smg, oracle_mod, kind="mcf", k=1, n_sub=1)
    conv, z, rh = mg.solve(p["RHS"], p["z0"], None, smg.SolveOpts(tol=0.0, max_iter=1500, check_every=500))
    assert not conv and len(rh) == 1500 and np.isfinite(rh).all() and rh[-1] <= rh[0]
    conv, z, rh = mg.solve(p["RHS"], p["z0"], None, smg.SolveOpts(tol=1e-9, max_iter=3000))
    assert conv and len(rh) < 40
    conv, z, rh = mg.solve(p["RHS"], p["z0"], None, smg.SolveOpts(tol=1e-9, max_iter=0))
    assert len(rh) == 0 and np.array_equal(z, p["z0"])


def test_solve_on_the_default_stream(smg, oracle_mod):
    """smg_hierarchy_set_stream(h, NULL): the legacy default stream cannot be captured into a graph -- launches go eager there
    and give the same bits."""
    p, mg, orc = build(smg, oracle_mod, kind="mcf", k=2, n_sub=2)
    o = smg.SolveOpts(tol=1e-9, max_iter=30)
    a = mg.solve(p["RHS"], p["z0"], None, o)
    mg.set_stream(None)
    b = mg.solve(p["RHS"], p["z0"], None, o)
    assert a[0] and b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    v = mg.vcycle(p["RHS"], p["z0"])
    assert np.isfinite(v).all()
