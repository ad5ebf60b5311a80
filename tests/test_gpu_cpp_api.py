"""GPU (-m gpu): the C++ mirror of the reference API (surface_multigrid_code_amd/csrc/mg_api.hpp) through the example
program examples/03_mg_solver.cpp (the reference's canonical caller, 03_mg_solver/main.cpp)."""
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stale(exe, srcs):
    """an example binary must be rebuilt when its sources, the C ABI header, the C++ mirror or the library changed (the options
    struct of include/smg.h is passed by pointer: a binary built against an older layout would corrupt its stack)"""
    deps = list(srcs) + [os.path.join(ROOT, "include", "smg.h"), os.path.join(ROOT, "surface_multigrid_code_amd", "csrc", "mg_api.hpp"),
                         os.path.join(ROOT, "surface_multigrid_code_amd", "lib", "libsmg.so")]
    return (not os.path.exists(exe)) or os.path.getmtime(exe) < max(os.path.getmtime(d) for d in deps if os.path.exists(d))


def test_cpp_example_matches_python_path(smg_mod, oracle_mod):
    smg, mesh = smg_mod, smg_mod.mesh
    exe = os.path.join(ROOT, "examples", "03_mg_solver")
    src = os.path.join(ROOT, "examples", "03_mg_solver.cpp")
    if _stale(exe, [src]):
        subprocess.check_call(["hipcc", "-std=c++17", "-O2", src, "-L" + os.path.join(ROOT, "surface_multigrid_code_amd", "lib"),
                               "-lsmg", "-Wl,-rpath," + os.path.join(ROOT, "surface_multigrid_code_amd", "lib"), "-o", exe])
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "surface_multigrid_code_amd", "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.check_output([exe, os.path.join(ROOT, "tests", "golden", "meshes", "bunny.smgm"), "1e-8"], env=env, text=True)
    res = [float(x) for x in re.findall(r"MG iteration: \d+, residual: ([0-9.eE+-]+)", out)]
    m = re.search(r"converged: (\d)  iterations: (\d+)  \|z\|\^2: ([0-9.eE+-]+)  unknowns: (\d+)", out)
    assert m and m.group(1) == "1" and int(m.group(4)) == 9353 - 149
    # same problem through the python mirror
    V, F = mesh.read_triangle_mesh("bunny.smgm")
    V = mesh.normalize_unit_area(V, F)
    mg = smg.mg_precompute(V, F, 0.25, 500, 1)
    A = (-mesh.cotmatrix(V, F)).tocsr()
    b = mesh.boundary_loop(F)
    B = mesh.massmatrix(V, F, "voronoi") @ np.ones(V.shape[0])
    B[b] = 0.0
    mg.precompute(A, b)
    conv, z, rh = mg.solve(B, np.zeros(V.shape[0]), np.zeros(len(b)), smg.SolveOpts(tol=1e-8, max_iter=20))
    assert conv and len(rh) == len(res) == int(m.group(2))
    np.testing.assert_allclose(res, rh, rtol=1e-5)          # printed with %g
    assert abs(float(m.group(3)) - float((z * z).sum())) <= 1e-9 * float((z * z).sum())


def test_cpp_mean_curvature_flow_example(smg_mod, oracle_mod):
    """examples/05_mean_curvature_flow.cpp: no-constraint overloads with a 3-column RHS, re-precompute every step
    (value-only path after the first), mg_VCycle mirror -- against the host/oracle pipeline."""
    from oracle import mesh_np as M
    smg, mesh = smg_mod, smg_mod.mesh
    exe = os.path.join(ROOT, "examples", "05_mean_curvature_flow")
    src = os.path.join(ROOT, "examples", "05_mean_curvature_flow.cpp")
    if _stale(exe, [src]):
        subprocess.check_call(["hipcc", "-std=c++17", "-O2", src, "-L" + os.path.join(ROOT, "surface_multigrid_code_amd", "lib"),
                               "-lsmg", "-Wl,-rpath," + os.path.join(ROOT, "surface_multigrid_code_amd", "lib"), "-o", exe])
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "surface_multigrid_code_amd", "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.check_output([exe, os.path.join(ROOT, "tests", "golden", "meshes", "ogre_sim.smgm"), "3"], env=env, text=True)
    got = [float(x) for x in re.findall(r"step \d+: converged 1 in \d+ iterations, \|U\|\^2 = ([0-9.eE+-]+)", out)]
    assert len(got) == 3 and "mg_VCycle: |u|^2" in out
    V, F = mesh.read_triangle_mesh("ogre_sim.smgm")
    V = mesh.normalize_unit_area(V, F)
    mg = smg.mg_precompute(V, F, 0.25, 100, 1)
    Ps = [mg.matrix(l, "P_full") for l in range(1, mg.n_levels)]
    L = M.cotmatrix(V, F)
    U = V.copy()
    for s in range(3):
        Mb = M.massmatrix(U, F, "barycentric")
        orc = oracle_mod.OracleMG(Ps)
        orc.precompute((Mb - 0.01 * L).tocsr())
        conv, z, rh = orc.solve(Mb @ U, U, tol=5e-7, max_iter=20)
        assert conv
        U = M.normalize_unit_area(z, F)
        assert abs(got[s] - float((U * U).sum())) <= 1e-5 * float((U * U).sum())


def test_cpp_closed_mesh_with_pins_example(smg_mod, oracle_mod):
    """examples/04_mg_solver_nobd.cpp (the reference's 04_mg_solver_nobd/main.cpp: closed surface, pinned vertices, random initial
    guess, tol 1e-10) against the same problem through the python mirror."""
    smg, mesh = smg_mod, smg_mod.mesh
    exe = os.path.join(ROOT, "examples", "04_mg_solver_nobd")
    src = os.path.join(ROOT, "examples", "04_mg_solver_nobd.cpp")
    if _stale(exe, [src]):
        subprocess.check_call(["hipcc", "-std=c++17", "-O2", src, "-L" + os.path.join(ROOT, "surface_multigrid_code_amd", "lib"),
                               "-lsmg", "-Wl,-rpath," + os.path.join(ROOT, "surface_multigrid_code_amd", "lib"), "-o", exe])
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "surface_multigrid_code_amd", "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.check_output([exe, os.path.join(ROOT, "tests", "golden", "meshes", "bunny_15K_init.smgm"), "346"], env=env, text=True)
    m = re.search(r"converged: (\d)  iterations: (\d+)  \|z\|\^2: ([0-9.eE+-]+)  unknowns: (\d+)", out)
    assert m and m.group(1) == "1" and int(m.group(4)) == 15804 - 346
    V, F = mesh.read_triangle_mesh("bunny_15K_init.smgm")
    V = mesh.normalize_unit_area(V, F)
    n = V.shape[0]
    mg = smg.mg_precompute(V, F, 0.25, 500, 1)
    A = (-mesh.cotmatrix(V, F)).tocsr()
    b = (np.arange(346) * (n // 346)).astype(np.int32)
    B = mesh.massmatrix(V, F, "voronoi") @ np.ones(n)
    B[b] = 0.0
    z0 = np.empty(n)
    x = 12345
    for i in range(n):
        x = (1103515245 * x + 12345) % 2147483648
        z0[i] = x / 1073741824.0 - 1.0
    mg.precompute(A, b)
    conv, z, rh = mg.solve(B, z0, np.zeros(len(b)), smg.SolveOpts(tol=1e-10, max_iter=20))
    assert conv and len(rh) == int(m.group(2))
    assert abs(float(m.group(3)) - float((z * z).sum())) <= 1e-9 * float((z * z).sum())
    unk = np.setdiff1d(np.arange(n), b)
    assert np.linalg.norm((B - A @ z[:, 0])[unk]) < 1e-10 and abs(z[b, 0]).max() == 0.0
    # the C++ mirror's opt-in (smgCoarseSolver::opts): the hybrid Gauss-Seidel / Chebyshev-Jacobi cycle lands on the same solution
    out2 = subprocess.check_output([exe, os.path.join(ROOT, "tests", "golden", "meshes", "bunny_15K_init.smgm"), "346", "1"], env=env, text=True)
    m2 = re.search(r"converged: (\d)  iterations: (\d+)  \|z\|\^2: ([0-9.eE+-]+)  unknowns: (\d+)", out2)
    conv2, z2, rh2 = mg.solve(B, z0, np.zeros(len(b)), smg.SolveOpts(tol=1e-10, max_iter=20, smoother="hybrid_chebyshev", jacobi_max_rows=300000))
    assert m2 and m2.group(1) == "1" and conv2 and len(rh2) == int(m2.group(2))
    assert abs(float(m2.group(3)) - float((z2 * z2).sum())) <= 1e-9 * float((z2 * z2).sum())
    assert abs(float(m2.group(3)) - float(m.group(3))) <= 1e-8 * float(m.group(3))


def test_eigen_adapter_runs_through_the_reference_signatures(smg_mod):
    """examples/smg_eigen_adapter.cpp (the reference's Eigen signatures on libsmg) built against tests/mock_eigen -- a
    stand-in with the Eigen members the adapter touches, NOT Eigen -- and driven like 03_mg_solver/main.cpp:38-75 by
    examples/adapter_check.cpp: precompute fills data.n/known/unknown/LHS/Auk and writes mg[l].A/A_diag/P/PT back, solve converges
    to the true residual, mg_VCycle / A() work on a coarser level, the no-constraint overloads re-precompute on the same mg."""
    exe = os.path.join(ROOT, "examples", "adapter_check")
    srcs = [os.path.join(ROOT, "examples", f) for f in ("adapter_check.cpp", "smg_eigen_adapter.cpp")]
    lib = os.path.join(ROOT, "surface_multigrid_code_amd", "lib")
    if _stale(exe, srcs):
        subprocess.check_call(["hipcc", "-std=c++17", "-O2", "-DSMG_ADAPTER_MOCK", "-I" + os.path.join(ROOT, "tests", "mock_eigen"),
                               "-I" + os.path.join(ROOT, "include")] + srcs + ["-L" + lib, "-lsmg", "-Wl,-rpath," + lib, "-o", exe])
    env = dict(os.environ, LD_LIBRARY_PATH=lib + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "meshes", "bunny.smgm")], env=env, text=True, capture_output=True)
    assert out.returncode == 0 and "ADAPTER_CHECK OK" in out.stdout, (out.stdout[-2000:], out.stderr[-2000:])
    assert "MG iteration: 0, residual:" in out.stdout and "residual norm:" in out.stdout      # the reference's prints (.cpp:111,127)


def test_cpp_sharded_mean_curvature_flow_example(smg_mod):
    """examples/05_mean_curvature_flow_sharded.cpp: the C++ caller shards its right-hand-side columns through the C++ mirror
    (smgCoarseSolver::reduce -> smg_solve_sharded) with an RCCL closure on a communicator it created.  On this box: world size 1 (the
    driver's 8-GPU job runs the same binary per rank); the first step's 3 + 5 columns must converge exactly like the python path's
    fused 8-column solve of the same system, and the reduction must have been issued once per loop entry."""
    smg, mesh = smg_mod, smg_mod.mesh
    if not os.path.exists("/opt/rocm/include/rccl/rccl.h"):
        pytest.skip("no RCCL headers")
    exe = os.path.join(ROOT, "examples", "05_mean_curvature_flow_sharded")
    src = os.path.join(ROOT, "examples", "05_mean_curvature_flow_sharded.cpp")
    libdir = os.path.join(ROOT, "surface_multigrid_code_amd", "lib")
    if _stale(exe, [src]):
        subprocess.check_call(["hipcc", "-std=c++17", "-O2", src, "-I/opt/rocm/include", "-L" + libdir, "-lsmg", "-L/opt/rocm/lib", "-lrccl",
                               "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    env.pop("RANK", None); env.pop("WORLD_SIZE", None)
    out = subprocess.check_output([exe, os.path.join(ROOT, "tests", "golden", "meshes", "ogre_sim.smgm"), "2", "8"], env=env, text=True, timeout=600)
    steps = re.findall(r"step (\d+): 8 columns on 1 rank\(s\) \(this rank: 8\), converged (\d) in (\d+) iterations, last residual ([0-9.eE+-]+), \|U\|\^2 = ([0-9.eE+-]+)", out)
    assert len(steps) == 2 and all(st[1] == "1" for st in steps), out[-2000:]
    calls = int(re.search(r"reductions issued by rank 0: (\d+) \(RCCL closure\)", out).group(1))
    assert calls >= sum(int(st[2]) for st in steps)
    # step 0 through the python mirror: same hierarchy, same matrix, the same 8 columns in one fused solve
    V, F = mesh.read_triangle_mesh("ogre_sim.smgm")
    V = mesh.normalize_unit_area(V, F)
    mg = smg.mg_precompute(V, F, 0.25, 100, 1)
    Mb = mesh.massmatrix(V, F, "barycentric")
    A = (Mb - 0.01 * mesh.cotmatrix(V, F)).tocsr()
    A.sort_indices()
    Z0 = np.zeros((V.shape[0], 8))
    Z0[:, :3] = V
    for c in range(3, 8):
        q = c - 3
        Z0[:, c] = np.sin(0.7 * (q + 1) * V[:, q % 3]) + 0.25 * V[:, (q + 1) % 3]
    mg.precompute(A)
    conv, z, rh = mg.solve(Mb @ Z0, Z0, None, smg.SolveOpts(tol=5e-7, max_iter=20))
    assert conv and len(rh) == int(steps[0][2])
    assert abs(float(steps[0][3]) - rh[-1]) <= 1e-5 * rh[-1]      # printed with %.6e


def test_cpp_sharded_example_with_two_ranks_on_one_gpu(smg_mod, tmp_path):
    """smg_solve_sharded's loop driven from C++ with WORLD SIZE 2: two processes of examples/05_mean_curvature_flow_sharded on
    this one GPU, the reduction closure going through the host (SMG_HOST_COMM_FILE: RCCL refuses ranks that share a device).  Both ranks
    must stop after the same iterations as the single-rank run (the break test sees the same Frobenius norm over ALL columns, reference
    src/min_quad_with_fixed_mg.cpp:110 -- up to the order of the partial sums), and the mesh after two flow steps must agree to 1e-12."""
    if not os.path.exists("/opt/rocm/include/rccl/rccl.h"):
        pytest.skip("no RCCL headers")
    exe = os.path.join(ROOT, "examples", "05_mean_curvature_flow_sharded")
    src = os.path.join(ROOT, "examples", "05_mean_curvature_flow_sharded.cpp")
    libdir = os.path.join(ROOT, "surface_multigrid_code_amd", "lib")
    if _stale(exe, [src]):
        subprocess.check_call(["hipcc", "-std=c++17", "-O2", src, "-I/opt/rocm/include", "-L" + libdir, "-lsmg", "-L/opt/rocm/lib", "-lrccl",
                               "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    base = dict(os.environ, LD_LIBRARY_PATH=libdir + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SMG_NCCL_ID_FILE", "SMG_HOST_COMM_FILE"):
        base.pop(k, None)
    mesh_file = os.path.join(ROOT, "tests", "golden", "meshes", "ogre_sim.smgm")
    pat = r"step (\d+): 8 columns on (\d) rank\(s\) \(this rank: (\d)\), converged (\d) in (\d+) iterations, last residual ([0-9.eE+-]+), \|U\|\^2 = ([0-9.eE+-]+)"
    one = subprocess.check_output([exe, mesh_file, "2", "8"], env=dict(base, SMG_HOST_COMM_FILE=str(tmp_path / "comm1")), text=True, timeout=600)
    s1 = re.findall(pat, one)
    assert len(s1) == 2 and all(st[1] == "1" and st[3] == "1" for st in s1) and "host closure" in one, one[-2000:]
    comm = str(tmp_path / "comm2")
    procs = [subprocess.Popen([exe, mesh_file, "2", "8"], env=dict(base, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", SMG_HOST_COMM_FILE=comm),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, o, e))
    assert all(rc == 0 for rc, _, _ in outs), [(rc, o[-500:], e[-500:]) for rc, o, e in outs]
    s2 = re.findall(pat, outs[0][1])
    assert len(s2) == 2 and all(st[1] == "2" and st[2] == "4" and st[3] == "1" for st in s2), outs[0][1][-2000:]
    for a, b in zip(s1, s2):
        assert a[4] == b[4], (a, b)                                                     # same iteration count per step
        assert abs(float(a[5]) - float(b[5])) <= 1e-5 * float(a[5])                     # printed with %.6e
        assert abs(float(a[6]) - float(b[6])) <= 1e-12 * float(a[6])                    # the mesh after the step
    calls = int(re.search(r"reductions issued by rank 0: (\d+) \(host closure\)", outs[0][1]).group(1))
    assert calls >= sum(int(st[4]) for st in s2)


def test_eigen_adapter_against_real_eigen():
    """If __graft_entry__.build() found an Eigen header tree (and the reference's headers) it compiled examples/smg_eigen_adapter.cpp
    against them: run that binary.  Skipped in this image (no Eigen anywhere: the adapter has only met tests/mock_eigen)."""
    exe = os.path.join(ROOT, "oracle", "_ref", "adapter_check_eigen")
    if not os.path.exists(exe):
        pytest.skip("no Eigen in the image: oracle/_ref/adapter_check_eigen was not built")
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "surface_multigrid_code_amd", "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.check_output([exe, os.path.join(ROOT, "tests", "golden", "meshes", "bunny.smgm")], env=env, text=True, timeout=600)
    assert "converged: 1" in out or "converged 1" in out, out[-1500:]
