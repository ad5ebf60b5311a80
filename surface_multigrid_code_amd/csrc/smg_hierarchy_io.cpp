// smg_hierarchy_io.cpp -- the hierarchy builders mg_precompute / mg_precompute_block (reference src/mg_precompute.cpp:15-87,
// src/mg_precompute_block.cpp:23-95, src/get_prolong.cpp:59-115), the point queries through the collapse record, and the .smgh files.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "smg_internal.hpp"

using namespace smg;

// ------------------------------------------------------------------------------------------------ mg_precompute
namespace smg {
// smg_decimate.cpp: one coarsening step (reference get_prolong(), src/get_prolong.cpp:3-57)
int decimate_level(const Mesh& fine, int tarF, int dec_type, int absorption_cap_tenths, Mesh& coarse, Csr& P, std::string& err, DecimationLog* log);
}

// number of levels by the reference's float rule (src/mg_precompute.cpp:27-38)
static int level_count(int nV, float ratio, int nVCoarsest)
{
    int nLvs = 1;
    float nv = (float)nV;
    while (true) {
        nv *= ratio;
        if (nv > (float)nVCoarsest) nLvs += 1;
        else break;
    }
    return nLvs;
}

static int build_decimated_levels(smg_hierarchy* h, int first_lv, const Mesh& base, int n_new, float ratio, int dec_type, int cap_tenths = 0, bool keep_log = false)
{
    Mesh cur = base;
    for (int s = 0; s < n_new; s++) {
        const int lv = first_lv + s;
        const int tarF = (int)std::round((float)cur.nF() * ratio);  // src/mg_precompute.cpp:59
        Mesh coarse;
        Csr P;
        std::string err;
        std::shared_ptr<DecimationLog> log = keep_log ? std::make_shared<DecimationLog>() : nullptr;
        if (decimate_level(cur, tarF, dec_type, cap_tenths, coarse, P, err, log.get()) != 0) return fail(SMG_ERR_INVALID, "mg_precompute: %s", err.c_str());
        h->lv[lv].dec_log = log;
        h->lv[lv].V = coarse.V;
        h->lv[lv].F = coarse.F;
        set_prolong(h, lv, std::move(P));
        cur = std::move(coarse);
    }
    return SMG_OK;
}

extern "C" int smg_mg_precompute(const double* V, int nV, const int* F, int nF, float ratio, int nVCoarsest, int dec_type,
                                 smg_hierarchy** out)
{
    return smg_mg_precompute_capped(V, nV, F, nF, ratio, nVCoarsest, dec_type, 0.0f, out);
}

static int smg_mg_precompute_capped_impl(const double* V, int nV, const int* F, int nF, float ratio, int nVCoarsest, int dec_type,
                                        float absorption_cap, smg_hierarchy** out, bool keep_log = false)
{
    if (!V || !F || !out || nV <= 0 || nF <= 0 || !(ratio > 0.f && ratio < 1.f) || !(absorption_cap >= 0.f))
        return fail(SMG_ERR_INVALID, "smg_mg_precompute: bad arguments");
    const int nLvs = level_count(nV, ratio, nVCoarsest);
    HierarchyOwner own(smg_hierarchy_create(nLvs));   // destroyed again if anything below fails or throws
    smg_hierarchy* h = own.h;
    if (!h) return SMG_ERR_ALLOC;
    Mesh m = wrap_mesh(V, nV, F, nF);
    h->lv[0].V = m.V; h->lv[0].F = m.F;   // src/mg_precompute.cpp:46-47
    int rc = build_decimated_levels(h, 1, m, nLvs - 1, ratio, dec_type, (int)std::lround(10.0 * absorption_cap), keep_log);
    if (rc) return rc;
    *out = own.release();
    return SMG_OK;
}

extern "C" int smg_mg_precompute_capped(const double* V, int nV, const int* F, int nF, float ratio, int nVCoarsest, int dec_type,
                                        float absorption_cap, smg_hierarchy** out)
{
    return guarded("smg_mg_precompute_capped", [&]() { return smg_mg_precompute_capped_impl(V, nV, F, nF, ratio, nVCoarsest, dec_type, absorption_cap, out); });
}

extern "C" int smg_mg_precompute_logged(const double* V, int nV, const int* F, int nF, float ratio, int nVCoarsest, int dec_type,
                                        float absorption_cap, int keep_log, smg_hierarchy** out)
{
    return guarded("smg_mg_precompute_logged", [&]() { return smg_mg_precompute_capped_impl(V, nV, F, nF, ratio, nVCoarsest, dec_type, absorption_cap, out, keep_log != 0); });
}

extern "C" int smg_query_coarse_to_fine(const smg_hierarchy* h, int lv, int n, const int* face, const double* bary, int* out_face,
                                        double* out_bary)
{
    return guarded("smg_query_coarse_to_fine", [&]() {
        if (!h || lv < 1 || lv >= h->n_levels || n < 0 || (n > 0 && (!face || !bary || !out_face || !out_bary)))
            return fail(SMG_ERR_INVALID, "smg_query_coarse_to_fine: bad arguments");
        const Level& Lv = h->lv[lv];
        if (!Lv.dec_log) return fail(SMG_ERR_INVALID, "smg_query_coarse_to_fine: level %d keeps no decimation log (smg_mg_precompute_logged)", lv);
        const int nFc = (int)Lv.dec_log->coarse_face.size();
        for (int i = 0; i < n; i++) {
            if (face[i] < 0 || face[i] >= nFc) return fail(SMG_ERR_INVALID, "smg_query_coarse_to_fine: face %d out of range", face[i]);
            for (int c = 0; c < 3; c++) if (!(bary[3 * i + c] == bary[3 * i + c])) return fail(SMG_ERR_INVALID, "smg_query_coarse_to_fine: NaN coordinate");
            if (!(bary[3 * i] + bary[3 * i + 1] + bary[3 * i + 2] > 0.0)) return fail(SMG_ERR_INVALID, "smg_query_coarse_to_fine: barycentric coordinates of point %d do not sum to a positive number", i);
        }
        query_coarse_to_fine(*Lv.dec_log, n, face, bary, out_face, out_bary);
        return (int)SMG_OK;
    });
}

extern "C" int smg_query_fine_to_coarse(const smg_hierarchy* h, int lv, int n, const int* face, const double* bary, int* out_face,
                                        double* out_bary)
{
    return guarded("smg_query_fine_to_coarse", [&]() {
        if (!h || lv < 1 || lv >= h->n_levels || n < 0 || (n > 0 && (!face || !bary || !out_face || !out_bary)))
            return fail(SMG_ERR_INVALID, "smg_query_fine_to_coarse: bad arguments");
        const Level& Lv = h->lv[lv];
        if (!Lv.dec_log) return fail(SMG_ERR_INVALID, "smg_query_fine_to_coarse: level %d keeps no decimation log (smg_mg_precompute_logged)", lv);
        const int nFf = (int)Lv.dec_log->face_recs.size();
        for (int i = 0; i < n; i++) {
            if (face[i] < 0 || face[i] >= nFf) return fail(SMG_ERR_INVALID, "smg_query_fine_to_coarse: face %d out of range", face[i]);
            for (int c = 0; c < 3; c++) if (!(bary[3 * i + c] == bary[3 * i + c])) return fail(SMG_ERR_INVALID, "smg_query_fine_to_coarse: NaN coordinate");
            if (!(bary[3 * i] + bary[3 * i + 1] + bary[3 * i + 2] > 0.0)) return fail(SMG_ERR_INVALID, "smg_query_fine_to_coarse: barycentric coordinates of point %d do not sum to a positive number", i);
        }
        query_fine_to_coarse(*Lv.dec_log, n, face, bary, out_face, out_bary);
        for (int i = 0; i < n; i++)
            if (out_face[i] < 0) return fail(SMG_ERR_INVALID, "smg_query_fine_to_coarse: point %d did not arrive on a face of the coarse mesh (inconsistent collapse record)", i);
        return (int)SMG_OK;
    });
}

static int smg_mg_precompute_block_impl(const double* V, int nV, const int* F, int nF, float ratio, int nVCoarsest, int dec_type,
                                       smg_hierarchy** out)
{
    if (!out) return fail(SMG_ERR_INVALID, "smg_mg_precompute_block: bad arguments");
    smg_hierarchy* raw = nullptr;
    int rc = smg_mg_precompute(V, nV, F, nF, ratio, nVCoarsest, dec_type, &raw);
    if (rc) return rc;
    HierarchyOwner own(raw);
    smg_hierarchy* h = own.h;
    for (int lv = 1; lv < h->n_levels; lv++) {
        Csr B = kron3(h->lv[lv].P_full);   // row 3r+d holds P(r,c) at column 3c+d  (src/get_prolong.cpp:108-110)
        set_prolong(h, lv, std::move(B));
    }
    *out = own.release();
    return SMG_OK;
}

extern "C" int smg_mg_precompute_block(const double* V, int nV, const int* F, int nF, float ratio, int nVCoarsest, int dec_type,
                                       smg_hierarchy** out)
{
    return guarded("smg_mg_precompute_block", [&]() { return smg_mg_precompute_block_impl(V, nV, F, nF, ratio, nVCoarsest, dec_type, out); });
}

extern "C" int smg_hierarchy_save(const smg_hierarchy* h, const char* path)
{
    if (!h || !path) return fail(SMG_ERR_INVALID, "smg_hierarchy_save: bad arguments");
    FILE* f = std::fopen(path, "wb");
    if (!f) return fail(SMG_ERR_IO, "cannot open '%s' for writing", path);
    const uint32_t ver = 1;
    const int32_t L = h->n_levels;
    bool ok = std::fwrite("SMGH", 1, 4, f) == 4 && std::fwrite(&ver, 4, 1, f) == 1 && std::fwrite(&L, 4, 1, f) == 1;
    for (int lv = 0; lv < L && ok; lv++) {
        const Level& Lv = h->lv[lv];
        const int32_t nV = (int32_t)(Lv.V.size() / 3), nF = (int32_t)(Lv.F.size() / 3);
        ok = std::fwrite(&nV, 4, 1, f) == 1 && std::fwrite(&nF, 4, 1, f) == 1 &&
             std::fwrite(Lv.V.data(), 8, Lv.V.size(), f) == Lv.V.size() && std::fwrite(Lv.F.data(), 4, Lv.F.size(), f) == Lv.F.size();
        if (lv >= 1 && ok) {
            const Csr& P = Lv.P_full;
            const int32_t hdr[3] = {P.nr, P.nc, (int32_t)P.nnz()};
            ok = std::fwrite(hdr, 4, 3, f) == 3 && std::fwrite(P.ptr.data(), 4, P.ptr.size(), f) == P.ptr.size() &&
                 std::fwrite(P.col.data(), 4, P.col.size(), f) == P.col.size() && std::fwrite(P.val.data(), 8, P.val.size(), f) == P.val.size();
        }
    }
    ok = (std::fclose(f) == 0) && ok;
    return ok ? SMG_OK : fail(SMG_ERR_IO, "short write to '%s'", path);
}

static int smg_hierarchy_load_impl(const char* path, smg_hierarchy** out)
{
    if (!path || !out) return fail(SMG_ERR_INVALID, "smg_hierarchy_load: bad arguments");
    struct File { FILE* f; ~File() { if (f) std::fclose(f); } } file{std::fopen(path, "rb")};   // closed on every way out, exceptions included
    FILE* f = file.f;
    if (!f) return fail(SMG_ERR_IO, "cannot open '%s'", path);
    // every count read from the file is checked against what the file can still hold before anything is allocated from it
    long fsize = 0;
    if (std::fseek(f, 0, SEEK_END) == 0) { fsize = std::ftell(f); std::rewind(f); }
    auto room = [&](double bytes) { const long at = std::ftell(f); return at >= 0 && bytes >= 0 && (double)at + bytes <= (double)fsize; };
    char magic[4];
    uint32_t ver = 0;
    int32_t L = 0;
    bool ok = std::fread(magic, 1, 4, f) == 4 && std::memcmp(magic, "SMGH", 4) == 0 && std::fread(&ver, 4, 1, f) == 1 && ver == 1 &&
              std::fread(&L, 4, 1, f) == 1 && L >= 1 && L < 64;
    HierarchyOwner own(ok ? smg_hierarchy_create(L) : nullptr);
    smg_hierarchy* h = own.h;
    const char* why = "not a hierarchy file";
    int prev_cols = -1;
    for (int lv = 0; lv < L && ok && h; lv++) {
        int32_t nV = 0, nF = 0;
        ok = std::fread(&nV, 4, 1, f) == 1 && std::fread(&nF, 4, 1, f) == 1 && nV >= 0 && nF >= 0 && room(24.0 * nV + 12.0 * nF);
        if (!ok) { why = "truncated or corrupt mesh block"; break; }
        h->lv[lv].V.resize((size_t)nV * 3); h->lv[lv].F.resize((size_t)nF * 3);
        ok = std::fread(h->lv[lv].V.data(), 8, h->lv[lv].V.size(), f) == h->lv[lv].V.size() &&
             std::fread(h->lv[lv].F.data(), 4, h->lv[lv].F.size(), f) == h->lv[lv].F.size();
        for (size_t i = 0; ok && i < h->lv[lv].F.size(); i++) if (h->lv[lv].F[i] < 0 || h->lv[lv].F[i] >= nV) { ok = false; why = "face index out of range"; }
        if (lv >= 1 && ok) {
            int32_t hdr[3];
            ok = std::fread(hdr, 4, 3, f) == 3 && hdr[0] >= 0 && hdr[1] >= 0 && hdr[2] >= 0 && room(4.0 * (hdr[0] + 1.0) + 12.0 * hdr[2]);
            if (!ok) { why = "truncated or corrupt prolongation block"; break; }
            Csr P;
            P.nr = hdr[0]; P.nc = hdr[1];
            P.ptr.resize((size_t)P.nr + 1); P.col.resize(hdr[2]); P.val.resize(hdr[2]);
            ok = std::fread(P.ptr.data(), 4, P.ptr.size(), f) == P.ptr.size() && std::fread(P.col.data(), 4, P.col.size(), f) == P.col.size() &&
                 std::fread(P.val.data(), 8, P.val.size(), f) == P.val.size() && P.ptr.back() == hdr[2];
            if (ok) if (const char* e = check_compressed(P.nr, P.nc, P.ptr.data(), P.col.data())) { ok = false; why = e; }
            if (ok && prev_cols >= 0 && P.nr != prev_cols) { ok = false; why = "prolongation sizes of consecutive levels do not chain"; }
            if (ok) { prev_cols = P.nc; set_prolong(h, lv, std::move(P)); }
        }
    }
    if (!ok || !h) return fail(SMG_ERR_IO, "'%s' is not a valid hierarchy file (%s)", path, why);
    *out = own.release();
    return SMG_OK;
}

extern "C" int smg_hierarchy_load(const char* path, smg_hierarchy** out)
{
    return guarded("smg_hierarchy_load", [&]() { return smg_hierarchy_load_impl(path, out); });
}

static int smg_mg_precompute_subdiv_impl(const double* V, int nV, const int* F, int nF, int n_sub, float ratio, int nVCoarsest,
                                        int n_extra_levels, smg_hierarchy** out, double* V_out, int* F_out)
{
    if (!V || !F || !out || nV <= 0 || nF <= 0 || n_sub < 0) return fail(SMG_ERR_INVALID, "smg_mg_precompute_subdiv: bad arguments");
    int extra = n_extra_levels >= 0 ? n_extra_levels : level_count(nV, ratio, nVCoarsest) - 1;
    Mesh base = wrap_mesh(V, nV, F, nF);
    Mesh fine = base;
    std::vector<Csr> Ps;
    subdivide(fine, n_sub, Ps);
    HierarchyOwner own(smg_hierarchy_create(1 + n_sub + extra));
    smg_hierarchy* h = own.h;
    if (!h) return SMG_ERR_ALLOC;
    h->lv[0].V = fine.V; h->lv[0].F = fine.F;
    for (int l = 1; l <= n_sub; l++) set_prolong(h, l, std::move(Ps[l - 1]));
    h->lv[n_sub].V = base.V; h->lv[n_sub].F = base.F;
    int rc = build_decimated_levels(h, n_sub + 1, base, extra, ratio, SMG_DEC_MIDPOINT);
    if (rc) return rc;
    if (V_out) std::copy(fine.V.begin(), fine.V.end(), V_out);
    if (F_out) std::copy(fine.F.begin(), fine.F.end(), F_out);
    *out = own.release();
    return SMG_OK;
}

extern "C" int smg_mg_precompute_subdiv(const double* V, int nV, const int* F, int nF, int n_sub, float ratio, int nVCoarsest,
                                        int n_extra_levels, smg_hierarchy** out, double* V_out, int* F_out)
{
    return guarded("smg_mg_precompute_subdiv", [&]() { return smg_mg_precompute_subdiv_impl(V, nV, F, nF, n_sub, ratio, nVCoarsest, n_extra_levels, out, V_out, F_out); });
}
