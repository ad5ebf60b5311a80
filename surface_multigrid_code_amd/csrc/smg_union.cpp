// smg_union.cpp -- independent meshes in ONE handle (include/smg.h: smg_hierarchy_create_union).
//
// BASELINE north_star: "independent RHS columns / independent meshes shard".  Across GPUs: one handle per device.  On ONE GPU many small meshes do not
// overlap as separate handles (a hipGraphLaunch of ~50 kernel nodes is enqueued under a process-wide lock: bench.py multi_mesh.by_handles), so they go
// into one block-diagonal handle whose every launch serves all of them -- while everything the reference does PER MESH stays per mesh: each member is its
// own min_quad_with_fixed_mg_solve loop (src/min_quad_with_fixed_mg.cpp:105-134) with its own residual norm, history and break test (a member whose
// test has passed keeps the iterate it had then), and the coarse solve uses the members' own inverses (the inverse of a block-diagonal matrix is
// block-diagonal: sum n_i^2 entries, not (sum n_i)^2).
#include <algorithm>
#include <numeric>

#include "smg_internal.hpp"

using namespace smg;

// P_full of the union's level lv = diag(P_full of the members' level lv)
extern "C" int smg_hierarchy_create_union(const smg_hierarchy* const* members, int m, smg_hierarchy** out)
{
    return guarded("smg_hierarchy_create_union", [&]() -> int {
        if (!members || m < 1 || !out) return fail(SMG_ERR_INVALID, "smg_hierarchy_create_union: bad arguments");
        *out = nullptr;
        for (int i = 0; i < m; i++) if (!members[i]) return fail(SMG_ERR_INVALID, "smg_hierarchy_create_union: member %d is null", i);
        const int L = members[0]->n_levels;
        if (L < 2) return fail(SMG_ERR_INVALID, "smg_hierarchy_create_union: the members need at least two levels");
        for (int i = 0; i < m; i++) {
            if (members[i]->n_levels != L) return fail(SMG_ERR_INVALID, "smg_hierarchy_create_union: member %d has %d levels, member 0 has %d", i, members[i]->n_levels, L);
            if (members[i]->union_m) return fail(SMG_ERR_INVALID, "smg_hierarchy_create_union: member %d is itself a union", i);
            for (int lv = 1; lv < L; lv++) {
                const Csr& P = members[i]->lv[lv].P_full;
                if (P.nr == 0 || P.nc == 0) return fail(SMG_ERR_INVALID, "smg_hierarchy_create_union: member %d has no prolongation on level %d", i, lv);
                if (lv > 1 && P.nr != members[i]->lv[lv - 1].P_full.nc) return fail(SMG_ERR_INVALID, "smg_hierarchy_create_union: member %d: P_%d does not fit P_%d", i, lv, lv - 1);
            }
        }
        HierarchyOwner own(smg_hierarchy_create(L));
        if (!own.h) return SMG_ERR_ALLOC;
        smg_hierarchy* h = own.h;
        for (int lv = 1; lv < L; lv++) {
            long nr = 0, nc = 0, nz = 0;
            for (int i = 0; i < m; i++) { const Csr& P = members[i]->lv[lv].P_full; nr += P.nr; nc += P.nc; nz += P.nnz(); }
            if (nr > 0x7fffffffl || nc > 0x7fffffffl || nz > 0x7fffffffl) return fail(SMG_ERR_INVALID, "smg_hierarchy_create_union: level %d of the union exceeds 32-bit indices", lv);
            Csr U;
            U.nr = (int)nr; U.nc = (int)nc;
            U.ptr.resize((size_t)nr + 1); U.col.resize((size_t)nz); U.val.resize((size_t)nz);
            U.ptr[0] = 0;
            int r0 = 0, c0 = 0, z0 = 0;
            for (int i = 0; i < m; i++) {
                const Csr& P = members[i]->lv[lv].P_full;
                for (int r = 0; r < P.nr; r++) U.ptr[(size_t)r0 + r + 1] = z0 + P.ptr[(size_t)r + 1];
                for (long p = 0; p < P.nnz(); p++) { U.col[(size_t)z0 + p] = P.col[(size_t)p] + c0; U.val[(size_t)z0 + p] = P.val[(size_t)p]; }
                r0 += P.nr; c0 += P.nc; z0 += (int)P.nnz();
            }
            int rc = set_prolong(h, lv, std::move(U));
            if (rc) return rc;
        }
        h->union_m = m;
        h->union_off0.assign((size_t)m + 1, 0);
        for (int i = 0; i < m; i++) h->union_off0[(size_t)i + 1] = h->union_off0[(size_t)i] + members[i]->lv[1].P_full.nr;
        *out = own.release();
        return SMG_OK;
    });
}

extern "C" int smg_union_members(const smg_hierarchy* h) { return h ? h->union_m : SMG_ERR_INVALID; }

extern "C" int smg_union_member_rows(const smg_hierarchy* h, int member, int* first, int* count)
{
    if (!h || h->union_m < 1 || member < 0 || member >= h->union_m) return fail(SMG_ERR_INVALID, "smg_union_member_rows: not a union / bad member");
    if (first) *first = h->union_off0[(size_t)member];
    if (count) *count = h->union_off0[(size_t)member + 1] - h->union_off0[(size_t)member];
    return SMG_OK;
}

// member of every row of the unknown-only system, level by level (block-diagonal P: a coarse column belongs to the member of the fine rows that touch it)
static int member_maps(const smg_hierarchy* h, std::vector<std::vector<int>>& mem)
{
    const int L = h->n_levels, m = h->union_m;
    mem.assign((size_t)L, std::vector<int>());
    const int n0 = h->lv[0].A.nr;
    mem[0].resize((size_t)n0);
    for (int u = 0; u < n0; u++) {
        const int full = h->has_known ? h->unknown[(size_t)u] : u;
        mem[0][(size_t)u] = (int)(std::upper_bound(h->union_off0.begin(), h->union_off0.end(), full) - h->union_off0.begin()) - 1;
        if (mem[0][(size_t)u] < 0 || mem[0][(size_t)u] >= m) return fail(SMG_ERR_INVALID, "union: row %d of the system lies outside every member (the system must have %d rows)", full, h->union_off0.back());
    }
    for (int lv = 1; lv < L; lv++) {
        const Csr& P = h->lv[lv].P;
        mem[(size_t)lv].assign((size_t)P.nc, -1);
        for (int r = 0; r < P.nr; r++)
            for (int p = P.ptr[(size_t)r]; p < P.ptr[(size_t)r + 1]; p++) {
                int& mc = mem[(size_t)lv][(size_t)P.col[(size_t)p]];
                const int mr = mem[(size_t)lv - 1][(size_t)r];
                if (mc < 0) mc = mr;
                else if (mc != mr) return fail(SMG_ERR_INVALID, "union: P_%d couples members %d and %d (the prolongations are not block-diagonal)", lv, mc, mr);
            }
        int last = 0;
        for (int& mc : mem[(size_t)lv]) { if (mc < 0) mc = last; if (mc < last) return fail(SMG_ERR_INVALID, "union: the members' columns of P_%d are not contiguous", lv); last = mc; }
    }
    return SMG_OK;
}

// The coarsest level of a union: one dense inverse PER MEMBER (blocked Gauss-Jordan on the device, like the single-mesh dense path), laid side by side in
// d_Ainv.  vals: the coarsest matrix's values on the device (CSR order of Lc.A); first: build the bookkeeping (a full precompute) -- else only re-invert.
int smg::union_coarse_factor(smg_hierarchy* h, const double* d_vals, bool first)
{
    const int L = h->n_levels, m = h->union_m;
    Level& Lc = h->lv[L - 1];
    if (first) {
        std::vector<std::vector<int>> mem;
        int rc = member_maps(h, mem);
        if (rc) return rc;
        const std::vector<int>& mc = mem[(size_t)L - 1];
        h->union_offc.assign((size_t)m + 1, Lc.n);
        h->union_offc[0] = 0;
        for (int i = 1; i <= m; i++) h->union_offc[(size_t)i] = (int)(std::lower_bound(mc.begin(), mc.end(), i) - mc.begin());
        h->union_moff.assign((size_t)m, 0);
        h->union_mlda.assign((size_t)m, 0);
        long long tot = 0;
        int max_pad = 64;
        for (int i = 0; i < m; i++) {
            const int ni = h->union_offc[(size_t)i + 1] - h->union_offc[(size_t)i];
            if (ni <= 0) return fail(SMG_ERR_INVALID, "union: member %d has no unknown on the coarsest level", i);
            if (ni > h->coarse_dense_max) return fail(SMG_ERR_INVALID, "union: member %d has %d unknowns on the coarsest level, beyond the dense range (%d)", i, ni, h->coarse_dense_max);
            const int pad = (ni + 63) / 64 * 64;
            h->union_moff[(size_t)i] = tot; h->union_mlda[(size_t)i] = pad;
            tot += (long long)pad * pad;
            max_pad = std::max(max_pad, pad);
        }
        std::vector<long long> pos((size_t)Lc.A.nnz());
        for (int r = 0; r < Lc.n; r++) {
            const int i = mc[(size_t)r], r0 = h->union_offc[(size_t)i], pad = h->union_mlda[(size_t)i];
            for (int p = Lc.A.ptr[(size_t)r]; p < Lc.A.ptr[(size_t)r + 1]; p++) {
                const int c = Lc.A.col[(size_t)p];
                if (mc[(size_t)c] != i) return fail(SMG_ERR_INVALID, "union: the coarsest matrix couples members %d and %d", i, mc[(size_t)c]);
                pos[(size_t)p] = h->union_moff[(size_t)i] + (long long)(r - r0) * pad + (c - r0);
            }
        }
        smg_hierarchy::UnionBuf& B = h->un;
        HIPCHK(h->d_dense_pos.upload(pos));
        HIPCHK(h->d_Ainv.ensure((size_t)tot));
        HIPCHK(B.crow_member.upload(mc)); HIPCHK(B.moff.upload(h->union_moff)); HIPCHK(B.mlda.upload(h->union_mlda));
        HIPCHK(B.mrow0.upload(h->union_offc));      // m + 1 entries: member i owns rows [mrow0[i], mrow0[i + 1]) of the coarsest level
        HIPCHK(B.ss.alloc((size_t)m)); HIPCHK(B.nhis.alloc((size_t)m)); HIPCHK(B.done.alloc((size_t)m));
        UnionDev& V = B.view;
        V = UnionDev();      // (rows / rptr: union_begin_solve -- level 0's numbering does not exist yet when the coarsest level's images are built)
        V.m = m;
        V.crow_member = B.crow_member.p; V.moff = B.moff.p; V.mlda = B.mlda.p; V.mrow0 = B.mrow0.p;
        V.ss = B.ss.p; V.nhis = B.nhis.p; V.done = B.done.p;
        h->nc = Lc.n;
        h->nc_pad = (Lc.n + 63) / 64 * 64 + 64;      // the last member's padded columns read (zeros times) rows up to 63 beyond the level
        h->coarse_sparse = false; h->coarse_schur = false;
        h->d_sympart.release(); h->d_Ainv32.release();
    }
    int max_pad = 64;
    for (int i = 0; i < m; i++) max_pad = std::max(max_pad, h->union_mlda[(size_t)i]);
    DevBuf<double> work;
    HIPCHK(work.alloc((size_t)2 * max_pad * 64 + 2 * 64 * 64));
    for (int i = 0; i < m; i++) {
        const int ni = h->union_offc[(size_t)i + 1] - h->union_offc[(size_t)i];
        HIPCHK(launch_dense_identity(h->d_Ainv.p + h->union_moff[(size_t)i], h->union_mlda[(size_t)i], ni, h->stream));
    }
    HIPCHK(launch_scatter_dense(h->d_Ainv.p, d_vals, h->d_dense_pos.p, (int)Lc.A.nnz(), h->stream));
    for (int i = 0; i < m; i++) HIPCHK(launch_spd_inverse(h->d_Ainv.p + h->union_moff[(size_t)i], h->union_mlda[(size_t)i], work.p, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return SMG_OK;
}

// level 0: the members' rows in the internal numbering (ascending inside a member)
static int union_level0_rows(smg_hierarchy* h)
{
    const int m = h->union_m;
    const Level& L0 = h->lv[0];
    std::vector<int> mem0((size_t)L0.n), rptr((size_t)m + 1, 0), rows((size_t)L0.n);
    for (int u = 0; u < L0.n; u++) {
        const int full = h->has_known ? h->unknown[(size_t)u] : u;
        mem0[(size_t)u] = (int)(std::upper_bound(h->union_off0.begin(), h->union_off0.end(), full) - h->union_off0.begin()) - 1;
        rptr[(size_t)mem0[(size_t)u] + 1]++;
    }
    for (int i = 0; i < m; i++) rptr[(size_t)i + 1] += rptr[(size_t)i];
    std::vector<int> fill(rptr.begin(), rptr.end() - 1);
    for (int t = 0; t < L0.n; t++) { const int u = L0.ord.perm[(size_t)t]; rows[(size_t)fill[(size_t)mem0[(size_t)u]]++] = t; }
    int max_rows = 0;
    for (int i = 0; i < m; i++) max_rows = std::max(max_rows, rptr[(size_t)i + 1] - rptr[(size_t)i]);
    smg_hierarchy::UnionBuf& B = h->un;
    HIPCHK(B.rows.upload(rows)); HIPCHK(B.rptr.upload(rptr));
    B.view.rows = B.rows.p; B.view.rptr = B.rptr.p; B.view.max_rows = max_rows;
    return SMG_OK;
}

// per-member solve state, sized and zeroed at smg_solve_begin (enqueued on the solve's stream)
int smg::union_begin_solve(smg_hierarchy* h, int k)
{
    smg_hierarchy::UnionBuf& B = h->un;
    if (!B.view.rows) { drop_graphs(h); int rc = union_level0_rows(h); if (rc) return rc; }
    const int m = h->union_m, cap = std::max(h->max_iter, 1);
    if (B.his.n < (size_t)m * cap) { drop_graphs(h); HIPCHK(B.his.alloc((size_t)m * cap)); }
    if (B.zsave.n < (size_t)h->lv[0].n * k) { drop_graphs(h); HIPCHK(B.zsave.alloc((size_t)h->lv[0].n * k)); }
    if (B.view.his_cap != cap) drop_graphs(h);
    B.view.his = B.his.p; B.view.zsave = B.zsave.p; B.view.his_cap = cap;
    HIPCHK(hipMemsetAsync(B.nhis.p, 0, (size_t)m * sizeof(int), h->stream));
    HIPCHK(hipMemsetAsync(B.done.p, 0, (size_t)m * sizeof(int), h->stream));
    HIPCHK(hipMemsetAsync(B.ss.p, 0, (size_t)m * sizeof(double), h->stream));
    return SMG_OK;
}

extern "C" int smg_union_get_history(smg_hierarchy* h, int member, double* r_his, int cap, int* n_his, int* converged)
{
    if (!h || h->union_m < 1 || member < 0 || member >= h->union_m) return fail(SMG_ERR_INVALID, "smg_union_get_history: not a union / bad member");
    if (h->in_solve) return fail(SMG_ERR_INVALID, "smg_union_get_history: a solve is in progress (smg_solve_end first)");
    if (!h->un.his.p || h->un.view.his_cap < 1) return fail(SMG_ERR_INVALID, "smg_union_get_history: no solve has run on this handle");
    DeviceScope dsc(h->device);
    int n = 0, dn = 0;
    HIPCHK(hipMemcpy(&n, h->un.nhis.p + member, sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(&dn, h->un.done.p + member, sizeof(int), hipMemcpyDeviceToHost));
    n = std::min(n, h->un.view.his_cap);
    if (n_his) *n_his = n;
    std::vector<double> tmp((size_t)std::max(n, 1));
    if (n > 0) HIPCHK(hipMemcpy(tmp.data(), h->un.his.p + (size_t)member * h->un.view.his_cap, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    if (r_his) for (int i = 0; i < std::min(n, cap); i++) r_his[i] = tmp[(size_t)i];
    // the reference's return value per member (src/min_quad_with_fixed_mg.cpp:131-134): the last recorded residual against the tolerance
    // (a member whose residual stopped being finite ended its own loop as failed: done == 2)
    if (converged) *converged = (n > 0 && dn != 2 && !(tmp[(size_t)n - 1] > h->tol)) ? 1 : 0;
    return SMG_OK;
}
