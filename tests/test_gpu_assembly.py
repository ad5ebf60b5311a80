"""GPU (-m gpu): SURVEY section 8 row f-3 -- operator assembly on the device and a whole mean-curvature-flow time step
(05_example_mean_curvature_flow/main.cpp:57-79) that never leaves HBM, against the host/oracle pipeline."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


def test_device_assembly_against_the_independent_restatement(smg_mod):
    """The reference side of this comparison is oracle/mesh_np.py (numpy restatement of igl::cotmatrix / igl::massmatrix, written
    independently of libsmg's C++): same pattern, values to rounding (the two sum per-face terms in different orders).  Below, the
    device kernels are additionally held bit-identical to libsmg's own host assembly (a regression check, not a parity claim)."""
    import torch
    from oracle import mesh_np as M
    smg, mesh = smg_mod, smg_mod.mesh
    dev = torch.device("cuda", 0)
    for name in ("ogre_sim.smgm", "bunny.smgm", "ogre.smgm"):
        V, F = mesh.read_triangle_mesh(name)
        V = mesh.normalize_unit_area(V, F)
        Vn = M.normalize_unit_area(*M.read_smgm(name))
        assert abs(V - Vn).max() <= 1e-14 * abs(Vn).max()
        asm = mesh.Assembler(F, V.shape[0])
        Vd = torch.from_numpy(V).to(dev)
        Lr = M.cotmatrix(V, F).tocsr()
        Lr.sort_indices()
        assert np.array_equal(asm.indptr, Lr.indptr) and np.array_equal(asm.indices, Lr.indices)
        for kind in ("barycentric", "voronoi"):
            Lval = torch.empty(asm.nnz, dtype=torch.float64, device=dev)
            val, mass = asm.assemble(Vd, 1.0, -0.01, kind, L_out=Lval)
            torch.cuda.synchronize()
            Mr = M.massmatrix(V, F, kind).diagonal()
            assert abs(Lval.cpu().numpy() - Lr.data).max() <= 1e-12 * abs(Lr.data).max()
            assert abs(mass.cpu().numpy() - Mr).max() <= 1e-13 * abs(Mr).max()
            assert abs(mass.cpu().numpy().sum() - 1.0) <= 1e-12                      # unit area
            ref = (sp.diags(Mr) - 0.01 * Lr).tocsr()                                 # 05_example_mean_curvature_flow/main.cpp:68
            ref.sort_indices()
            assert abs(val.cpu().numpy() - ref.data).max() <= 1e-12 * abs(ref.data).max()


def test_device_assembly_is_bit_identical_to_host(smg_mod):
    import torch
    smg, mesh = smg_mod, smg_mod.mesh
    dev = torch.device("cuda", 0)
    for name in ("ogre_sim.smgm", "bunny.smgm"):
        V, F = mesh.read_triangle_mesh(name)
        V = mesh.normalize_unit_area(V, F)
        asm = mesh.Assembler(F, V.shape[0])
        Lh = mesh.cotmatrix(V, F)
        assert np.array_equal(asm.indptr, Lh.indptr) and np.array_equal(asm.indices, Lh.indices)
        Vd = torch.from_numpy(V).to(dev)
        for kind in ("barycentric", "voronoi"):
            Lval = torch.empty(asm.nnz, dtype=torch.float64, device=dev)
            val, mass = asm.assemble(Vd, 1.0, -0.01, kind, L_out=Lval)
            torch.cuda.synchronize()
            Mh = mesh.massmatrix(V, F, kind).diagonal()
            assert np.array_equal(Lval.cpu().numpy(), Lh.data), "cotangent values differ from the host assembly"
            assert np.array_equal(mass.cpu().numpy(), Mh), "mass differs from the host assembly"
            ref = (sp.diags(Mh) - 0.01 * Lh).tocsr()            # 05_example_mean_curvature_flow/main.cpp:68
            ref.sort_indices()
            assert np.array_equal(val.cpu().numpy(), ref.data)


def test_mean_curvature_flow_steps_stay_on_the_gpu(smg_mod, oracle_mod):
    import torch
    from oracle import mesh_np as M
    smg, mesh = smg_mod, smg_mod.mesh
    dev = torch.device("cuda", 0)
    V, F = mesh.read_triangle_mesh("ogre_sim.smgm")
    V = mesh.normalize_unit_area(V, F)
    n, delta, tol = V.shape[0], 0.01, 1e-11      # tight: both pipelines are compared at solver precision
    mg = smg.mg_precompute(V, F, 0.25, 100, 1)
    Ps = [mg.matrix(l, "P_full") for l in range(1, mg.n_levels)]
    L = mesh.cotmatrix(V, F)                                   # fixed: built from the ORIGINAL mesh (main.cpp:44)
    asm = mesh.Assembler(F, n)
    # ---- GPU pipeline: U lives in HBM for all steps
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        mg.set_stream(st.cuda_stream)
        U = torch.from_numpy(V).to(dev)
        Lval = torch.from_numpy(L.data).to(dev)
        for step in range(3):
            _, mass = asm.assemble(U, 1.0, 0.0, "barycentric", stream=st.cuda_stream)   # M(U)
            lhs = -delta * Lval
            diag_pos = torch.from_numpy(np.flatnonzero(L.indices == np.repeat(np.arange(n), np.diff(L.indptr)))).to(dev)
            lhs[diag_pos] = mass + lhs[diag_pos]                # LHS = M - delta L
            rhs = (mass[:, None] * U).T.contiguous()           # RHS = M U, column-major n x 3
            z0 = U.T.contiguous()
            if step == 0:
                st.synchronize()
                A0 = sp.csr_matrix((lhs.cpu().numpy(), L.indices, L.indptr), shape=(n, n))
                mg.precompute(A0)                                # full precompute once
            else:
                mg.precompute_values_device(lhs.data_ptr())      # every later step: device only
            z = torch.empty_like(z0)
            mg.solve_begin(rhs.data_ptr(), n, z0.data_ptr(), n, 3, opts=smg.SolveOpts(tol=tol, max_iter=40))
            mg.outer_iterations(40)
            conv, rh = mg.solve_end(z.data_ptr(), n)
            assert conv
            U = z.T.contiguous()
            # normalize_unit_area on the device (src/normalize_unit_area.cpp:10-24)
            a, b, c = U[torch.from_numpy(F[:, 0]).long().to(dev)], U[torch.from_numpy(F[:, 1]).long().to(dev)], U[torch.from_numpy(F[:, 2]).long().to(dev)]
            area2 = torch.linalg.norm(torch.linalg.cross(b - a, c - a), dim=1).sum()
            U = U / torch.sqrt(area2 / 2)
            U = torch.stack([U[:, 0] - U[:, 0].mean(), U[:, 1] - U[:, 1].mean(), U[:, 2] - U[:, 2].min()], dim=1).contiguous()
        st.synchronize()
        mg.set_stream(None)
    # ---- host/oracle pipeline
    Uh = V.copy()
    for step in range(3):
        Mb = M.massmatrix(Uh, F, "barycentric")
        LHS = (Mb - delta * L).tocsr()
        orc = oracle_mod.OracleMG(Ps)
        orc.precompute(LHS)
        conv, z, rh = orc.solve(Mb @ Uh, Uh, tol=1e-12, max_iter=60)
        assert conv
        Uh = M.normalize_unit_area(z, F)
    err = np.abs(U.cpu().numpy() - Uh).max()
    assert err < 1e-7, err
