// ThreadSanitizer driver (tests/test_sanitized_host.py, surface_multigrid_code_amd/build.py: build_tsan): the host half of a first
// smg_precompute -- its own thread, the locality-order and colouring threads, the persistent pool, the hand-over to the device half -- on a
// subdivided torus through the C ABI, under its subdivision hierarchy and under smg_mg_precompute's (decimated) one.  No device needed: without one the call ends with SMG_ERR_NO_DEVICE after the host half has run.
#include <cstdio>
#include <cmath>
#include <vector>
#include "smg.h"
int main()
{
    const int nu = 40, nv = 30;
    std::vector<double> V(3 * nu * nv); std::vector<int> F(3 * 2 * nu * nv);
    smg_mesh_torus(nu, nv, 1.0, 0.4, V.data(), F.data());
    for (int round = 0; round < 3; round++) {
        smg_hierarchy* h = nullptr;
        const int n_sub = 3;
        int nF0 = 2 * nu * nv, nFf = nF0 * 64;
        // fine vertex count of a closed genus-1 mesh: V - E + F = 0, E = 3F/2  =>  V = F/2
        const int nVf = nFf / 2;
        std::vector<double> Vf(3 * (size_t)nVf); std::vector<int> Ff(3 * (size_t)nFf);
        int rc = smg_mg_precompute_subdiv(V.data(), nu * nv, F.data(), nF0, n_sub, 0.25f, 100, 0, &h, Vf.data(), Ff.data());
        if (rc) { printf("subdiv: %d %s\n", rc, smg_last_error()); return 1; }
        int nnz = 0;
        smg_mesh_cotmatrix(Vf.data(), nVf, Ff.data(), nFf, &nnz, nullptr, nullptr, nullptr);
        std::vector<int> ptr(nVf + 1), col(nnz); std::vector<double> val(nnz), m(nVf);
        smg_mesh_cotmatrix(Vf.data(), nVf, Ff.data(), nFf, &nnz, ptr.data(), col.data(), val.data());
        smg_mesh_massmatrix(Vf.data(), nVf, Ff.data(), nFf, 0, m.data());
        for (int i = 0; i < nVf; i++) for (int p = ptr[i]; p < ptr[i + 1]; p++) val[p] = (col[p] == i ? m[i] : 0.0) - 0.01 * val[p];
        rc = smg_precompute(h, nVf, ptr.data(), col.data(), val.data(), nullptr, 0);
        printf("round %d: %d rows, %d levels, precompute rc = %d (%s)\n", round, nVf, smg_hierarchy_levels(h), rc, smg_last_error());
        smg_hierarchy_destroy(h);
        // the same fine mesh under the reference's own kind of hierarchy (smg_mg_precompute: edges enumerated on the host threads, the queue's
        // sorts) -- no level can inherit colours there, so level 0 is coloured on its own thread beside the host half and the other levels inside
        // the locality-order tasks
        if (round < 2) {
            smg_hierarchy* hd = nullptr;
            rc = smg_mg_precompute(Vf.data(), nVf, Ff.data(), nFf, 0.25f, 500, 1, &hd);
            if (rc) { printf("mg_precompute: %d %s\n", rc, smg_last_error()); return 1; }
            rc = smg_precompute(hd, nVf, ptr.data(), col.data(), val.data(), nullptr, 0);
            printf("round %d (decimated): %d levels, decimated precompute rc = %d (%s)\n", round, smg_hierarchy_levels(hd), rc, smg_last_error());
            smg_hierarchy_destroy(hd);
        }
    }
    return 0;
}
