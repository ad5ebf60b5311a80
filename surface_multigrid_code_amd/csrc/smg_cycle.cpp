// smg_cycle.cpp -- mg_VCycle (reference src/mg_VCycle.cpp:3-201) and min_quad_with_fixed_mg_solve (reference
// src/min_quad_with_fixed_mg.cpp:80-135, :288-361) behind smg_solve*: the launch sequence of a cycle, the hipGraph cache, the outer loop
// with its device-side break test, the V-cycle pieces on host blocks and the raw device interface.
// The V-cycle never leaves the GPU: every kernel is enqueued on the handle's stream, the outer loop's break test runs on the device
// (Ctrl, smg_device.hpp) and one outer iteration is replayed as a hipGraph.
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

// ------------------------------------------------------------------------------------------------ V-cycle

// ---- overlapped tiling of the Gauss-Seidel sweeps of the latency-bound levels (smg_tiled.hpp): relax(sweeps) as ONE launch ----------
// Which levels: scalar fp64 hierarchies, up to 7 columns (groups of 3 per launch; 8 and more take the wide colour kernels), Gauss-Seidel, SMG_TILED_MIN_ROWS <= rows <= SMG_TILED_MAX_ROWS (default 512 ..
// 122880 = one round of 240 parts of 512 rows: above, the redundant halo work of the tiles costs more than the launches it saves -- measured at C3 level 1, 253 k rows, and again on a 160 k-row
// union level; below 2 048 rows it pays as well: a 768-row level 37.5 -> 18.0 us per visit, tools/size_sweep.py), at most 5 colours and 12 entries per row.
// SMG_TILED=0 switches it off (A/B knob; the results are bit-identical either way).
// The plans of the one-launch / piece-wise / block-wise sweeps hold copies of the level's values; the maps that refresh them (value slot -> index into
// Level::d_Aval) are only needed by a value-only re-precompute: they stay on the host until the first one (a quarter of a plan's bytes).
static hipError_t ensure_map(DevBuf<int>& d, const std::vector<int>& host)
{
    if (d.n == host.size() && (d.p || host.empty())) return hipSuccess;
    return d.upload(host);
}
static bool wgs_forced(const smg_hierarchy* h, int lv);
static bool tiled_wanted(const smg_hierarchy* h, int lv, int k, int sweeps)
{
    static const int on = env_int("SMG_TILED", 1);
    if (wgs_forced(h, lv)) return false;      // smg_hierarchy_set_wave_gs(h, 1): the level sweeps piece-wise for EVERY k (the one-launch relax exists for k <= 7 only,
                                              // and the order of a level's sweep must not depend on the number of columns: column-sharded == fused)
    static const int max_rows = env_int("SMG_TILED_MAX_ROWS", 122880), min_rows = env_int("SMG_TILED_MIN_ROWS", 512);
    if (!on || h->bs != 1 || k < 1 || k > 7 || lv < 0 || lv >= h->n_levels - 1 || sweeps < 1 || sweeps > 3) return false;
    if (level_kind(h, lv) != LV_GS) return false;
    const int n = h->lv[lv].n;
    return n >= min_rows && n <= max_rows;
}
// the plan of relax(sweeps) on level lv, or nullptr (not wanted / the level does not qualify / not built yet)
static const TiledDev* tiled_plan(const smg_hierarchy* h, int lv, int k, int sweeps)
{
    if (!tiled_wanted(h, lv, k, sweeps)) return nullptr;
    const TiledBuf& B = h->lv[lv].tiled[sweeps];
    // k columns go through the tiles in groups of up to 3, whose iterates share the workgroup's 64 KB of LDS
    return B.view.n_tiles > 0 && (size_t)B.view.max_ext * std::min(k, 3) * sizeof(double) + TILED_LDS_STATIC <= 64 * 1024 ? &B.view : nullptr;
}
static int ensure_tiled(smg_hierarchy* h, int lv, int sweeps)
{
    Level& Lv = h->lv[lv];
    TiledBuf& B = Lv.tiled[sweeps];
    if (B.tried) return SMG_OK;
    B.tried = true;
    // Tile size (measured at C3, tools/tiled_sweep.sh): parts of 128 .. 256 rows, 512 threads (one row of every colour per thread).
    // Smaller tiles put more CUs to work but the halo of P rings then dominates (6x redundant row updates at 64 rows: slower);
    // larger ones run too few workgroups.
    // Beyond 65 536 rows parts of 256 rows are more workgroups than the part has compute units (a second round of them: tools/size_sweep.py, a
    // 69 120-row level 45.8 us per visit against 28 us at 56 320 rows): the parts grow to 512 rows so that the level stays one round up to 122 880
    // rows (69 120 rows: 34.2 us, 77 824: 43.4 -> 32.2, 101 376: 47.5 (colour launches) -> 36.7; at 30 720 rows parts of 512 rows lose: 25.1 -> 28.0).
    static const int rows_env = env_int("SMG_TILED_ROWS", 0), nt_env = env_int("SMG_TILED_NT", 0);
    const int tile_rows0 = rows_env > 0 ? rows_env : std::min(512, std::max(256, (Lv.n + 239) / 240));
    constexpr int max_ext = (64 * 1024 - TILED_LDS_STATIC) / 8;    // 64 KB of LDS, the kernel's static header included
    // the matrix the smoother streams, in the internal numbering; entry -> index into the level's values in the caller's CSR order
    std::vector<int> tsrc;
    Csr AT;
    { int rc = ensure_A_int(h, lv); if (rc) return rc; }
    if (Lv.gs_on_transpose) AT = transpose(Lv.A_int, &tsrc);
    const Csr& G = Lv.gs_on_transpose ? AT : Lv.A_int;
    // a tile whose halo makes a colour's panel longer than the workgroup gets smaller tiles
    TiledGs P;
    int threads = 512;
    for (int tile_rows = tile_rows0, tries = 0; tries < 3 && P.empty(); tile_rows = tile_rows * 2 / 3, tries++) {
        threads = nt_env > 0 ? nt_env : 512;
        P = build_tiled_gs(G, Lv.ord.color_ptr, sweeps, tile_rows, max_ext, threads);
    }
    if (P.empty()) return SMG_OK;
    auto to_level_value = [&](const std::vector<int>& entries) {
        std::vector<int> m(entries.size());
        for (size_t i = 0; i < m.size(); i++) {
            const int e = entries[i];
            m[i] = e < 0 ? -1 : Lv.A_int_src[(size_t)(Lv.gs_on_transpose ? tsrc[(size_t)e] : e)];
        }
        return m;
    };
    std::vector<int> map = to_level_value(P.pentry), mapd = to_level_value(P.pdentry);
    HIPCHK(B.hdr.upload(P.hdr)); HIPCHK(B.ext_rows.upload(P.ext_rows)); HIPCHK(B.pcol.upload(P.pcol)); HIPCHK(B.pval.upload(P.pval));
    HIPCHK(B.prow.upload(P.prow)); HIPCHK(B.pdiag.upload(P.pdiag));
    B.host_map = std::move(map); B.host_mapd = std::move(mapd);      // (uploaded when a value-only re-precompute first needs them: ensure_map)
    HIPCHK(tiled_gs_prepare(P.max_ext));
    int wmax = 0;
    for (int t = 0; t < P.n_tiles; t++) wmax = std::max(wmax, P.hdr[(size_t)t * TILED_HDR + 2]);
    B.view.threads = threads;
    B.view.n_tiles = P.n_tiles; B.view.nc = P.nc; B.view.P = P.P; B.view.sweeps = sweeps; B.view.max_ext = P.max_ext; B.view.w_max = wmax;
    B.view.hdr = B.hdr.p; B.view.ext_rows = B.ext_rows.p; B.view.pcol = B.pcol.p; B.view.pval = B.pval.p; B.view.prow = B.prow.p; B.view.pdiag = B.pdiag.p;
    B.updates = P.updates;
    // after a value-only re-precompute the host copy of the values is stale: take them from the device copy
    if (h->host_stale && Lv.d_Aval.p) {
        HIPCHK(ensure_map(B.map, B.host_map)); HIPCHK(ensure_map(B.mapd, B.host_mapd));
        HIPCHK(launch_gather_vals(B.pval.p, Lv.d_Aval.p, B.map.p, B.pval.n, h->stream));
        HIPCHK(launch_gather_vals(B.pdiag.p, Lv.d_Aval.p, B.mapd.p, B.pdiag.n, h->stream));
    }
    if (env_int("SMG_DEBUG_TILED", 0))
        std::fprintf(stderr, "tiled relax(%d) level %d: %d rows, %d tiles x %d threads, %d phases, extended tile <= %d rows, %.2fx row updates, entries per row <= %d\n", sweeps, lv, Lv.n,
                     P.n_tiles, threads, P.P, P.max_ext, (double)P.updates / ((double)sweeps * Lv.n), wmax);
    return SMG_OK;
}
int smg::refresh_tiled_values(smg_hierarchy* h)
{
    for (int lv = 0; lv < h->n_levels - 1; lv++) {
        for (int s = 1; s <= 3; s++) {
            TiledBuf& B = h->lv[lv].tiled[s];
            if (B.view.n_tiles > 0) {
                HIPCHK(ensure_map(B.map, B.host_map)); HIPCHK(ensure_map(B.mapd, B.host_mapd));
                HIPCHK(launch_gather_vals(B.pval.p, h->lv[lv].d_Aval.p, B.map.p, B.pval.n, h->stream));
                HIPCHK(launch_gather_vals(B.pdiag.p, h->lv[lv].d_Aval.p, B.mapd.p, B.pdiag.n, h->stream));
            }
        }
        WgsBuf& W = h->lv[lv].wgs;
        if (W.view.n_pieces > 0) {
            HIPCHK(ensure_map(W.map, W.host_map)); HIPCHK(ensure_map(W.mapd, W.host_mapd));
            HIPCHK(launch_gather_vals(W.eval.p, h->lv[lv].d_Aval.p, W.map.p, W.eval.n, h->stream));
            HIPCHK(launch_gather_vals(W.diag.p, h->lv[lv].d_Aval.p, W.mapd.p, W.diag.n, h->stream));
        }
        BgsBuf& Q = h->lv[lv].bgs;
        if (Q.view.n_blocks > 0) {
            HIPCHK(ensure_map(Q.map, Q.host_map)); HIPCHK(ensure_map(Q.mapd, Q.host_mapd));
            HIPCHK(launch_gather_vals(Q.eval.p, h->lv[lv].d_Aval.p, Q.map.p, Q.eval.n, h->stream));
            HIPCHK(launch_gather_vals(Q.udiag.p, h->lv[lv].d_Aval.p, Q.mapd.p, Q.udiag.n, h->stream));
        }
    }
    return SMG_OK;
}
void smg::drop_tiled(smg_hierarchy* h)
{
    for (auto& Lv : h->lv) { for (auto& B : Lv.tiled) B = TiledBuf(); Lv.bgs = BgsBuf(); Lv.wgs = WgsBuf(); }
}

// ---- block Gauss-Seidel for solves with a multiple of 16 columns (smg_bgs.hpp): one launch per BLOCK colour -----------------
// Which levels: scalar fp64 hierarchies, Gauss-Seidel, k % 16 == 0, at least bgs_min_rows rows (smg_hierarchy_set_block_gs; default: never;
// SMG_BGS_MIN_ROWS; SMG_BGS=0 switches it off).  Measured at C3 (tools/bgs_cycle.py): worth it from ~500 000 rows on.
static bool bgs_wanted(const smg_hierarchy* h, int lv, int k)
{
    static const int on = env_int("SMG_BGS", 1);
    if (!on || h->bs != 1 || h->precision != 0 || k < BGS_COLS || k % BGS_COLS != 0 || lv < 0 || lv >= h->n_levels - 1 || h->bgs_min_rows < 0) return false;
    if (level_kind(h, lv) != LV_GS) return false;
    return h->lv[lv].n >= h->bgs_min_rows;
}
static const BgsBuf* bgs_plan(const smg_hierarchy* h, int lv, int k)
{
    if (!bgs_wanted(h, lv, k)) return nullptr;
    const BgsBuf& B = h->lv[lv].bgs;
    return B.view.n_blocks > 0 ? &B : nullptr;
}
static int ensure_bgs(smg_hierarchy* h, int lv)
{
    Level& Lv = h->lv[lv];
    BgsBuf& B = Lv.bgs;
    if (B.tried) return SMG_OK;
    B.tried = true;
    static const int rows_env = env_int("SMG_BGS_ROWS", 64);
    std::vector<int> tsrc;
    Csr AT;
    { int rc = ensure_A_int(h, lv); if (rc) return rc; }
    if (Lv.gs_on_transpose) AT = transpose(Lv.A_int, &tsrc);
    const Csr& G = Lv.gs_on_transpose ? AT : Lv.A_int;
    const auto t_plan0 = std::chrono::steady_clock::now();
    BgsPlan P = build_bgs(G, Lv.ord.color_ptr, std::min(std::max(rows_env, 8), (int)BGS_ROWS));
    const double plan_ms = 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t_plan0).count();
    if (P.empty()) return SMG_OK;
    auto to_level_value = [&](const std::vector<int>& entries) {
        std::vector<int> m(entries.size());
        for (size_t i = 0; i < m.size(); i++) {
            const int e = entries[i];
            m[i] = e < 0 ? -1 : Lv.A_int_src[(size_t)(Lv.gs_on_transpose ? tsrc[(size_t)e] : e)];
        }
        return m;
    };
    const std::vector<int> map = to_level_value(P.eentry), mapd = to_level_value(P.dentry);
    HIPCHK(B.hdr.upload(P.hdr)); HIPCHK(B.xrow.upload(P.xrow)); HIPCHK(B.ugrow.upload(P.ugrow)); HIPCHK(B.ulrow.upload(P.ulrow)); HIPCHK(B.udiag.upload(P.udiag));
    HIPCHK(B.eidx.upload(P.eidx)); HIPCHK(B.eval.upload(P.eval)); B.host_map = map; B.host_mapd = mapd;
    B.view.n_blocks = P.n_blocks; B.view.n_colors = P.n_colors; B.view.xrows = P.xrows;
    B.view.hdr = B.hdr.p; B.view.xrow = B.xrow.p; B.view.ugrow = B.ugrow.p; B.view.ulrow = B.ulrow.p; B.view.udiag = B.udiag.p; B.view.eidx = B.eidx.p; B.view.eval = B.eval.p;
    B.color_ptr = P.color_ptr; B.host_rows = P.rows; B.host_blk_ptr = P.blk_ptr; B.rim = P.rim; B.fill = P.fill;
    // after a value-only re-precompute the host copy of the values is stale: take them from the device copy
    if (h->host_stale && Lv.d_Aval.p) {
        HIPCHK(ensure_map(B.map, B.host_map)); HIPCHK(ensure_map(B.mapd, B.host_mapd));
        HIPCHK(launch_gather_vals(B.eval.p, Lv.d_Aval.p, B.map.p, B.eval.n, h->stream));
        HIPCHK(launch_gather_vals(B.udiag.p, Lv.d_Aval.p, B.mapd.p, B.udiag.n, h->stream));
    }
    if (env_int("SMG_DEBUG_BGS", 0))
        std::fprintf(stderr, "block Gauss-Seidel level %d: %d rows, %d blocks in %d colours, %.0f %% of the units' row slots hold a row of their own, rim %.3f rows read per row beyond the iterate, LDS image of %d rows; plan built in %.0f ms\n",
                     lv, Lv.n, P.n_blocks, P.n_colors, 100.0 * P.fill, P.rim, P.xrows, plan_ms);
    return SMG_OK;
}

// ---- wave Gauss-Seidel on the Galerkin levels of decimated hierarchies (smg_wgs.hpp): one launch per PIECE colour ------------------
// Which levels: scalar fp64 hierarchies, Gauss-Seidel, any number of columns, SMG_WGS_MIN_ROWS <= rows <= SMG_WGS_MAX_ROWS, no one-launch relax()
// (overlapped tiling) available; automatic mode: only levels the colour launches serve badly -- more than TILED_NCMAX colours or rows of more
// than TILED_WMAX entries, i.e. the Galerkin levels of the reference's own hierarchies (mg_precompute).  smg_hierarchy_set_wave_gs / SMG_WGS=0|1|2.
static int wgs_mode_now(const smg_hierarchy* h)
{
    static const int env = env_int("SMG_WGS", -1);
    return env >= 0 ? (env == 0 ? 0 : env == 1 ? -1 : 1) : h->wgs_mode;      // SMG_WGS: 0 off, 1 automatic, 2 every level in range
}
// mode 1 (every Gauss-Seidel level in range): what the level needs to sweep piece-wise, whatever k -- such a level takes no one-launch relax (tiled_wanted)
static bool wgs_forced(const smg_hierarchy* h, int lv)
{
    static const int max_rows = env_int("SMG_WGS_MAX_ROWS", 600000), min_rows = env_int("SMG_WGS_MIN_ROWS", 512);
    if (wgs_mode_now(h) != 1 || h->bs != 1 || h->precision != 0 || lv < 0 || lv >= h->n_levels - 1 || level_kind(h, lv) != LV_GS) return false;
    return h->lv[lv].n >= min_rows && h->lv[lv].n <= max_rows;
}
static bool wgs_wanted(const smg_hierarchy* h, int lv, int k)
{
    static const int max_rows = env_int("SMG_WGS_MAX_ROWS", 600000), min_rows = env_int("SMG_WGS_MIN_ROWS", 512);
    const int mode = wgs_mode_now(h);
    if (mode == 0 || h->bs != 1 || h->precision != 0 || k < 1 || lv < 0 || lv >= h->n_levels - 1) return false;      // (every k: the order of a level's sweep must not depend on how the columns are sharded)
    if (level_kind(h, lv) != LV_GS) return false;
    const Level& Lv = h->lv[lv];
    if (Lv.n < min_rows || Lv.n > max_rows) return false;
    if (mode == 1) return true;
    const SellBuf& Gs = Lv.gs_on_transpose ? Lv.dAT : Lv.dA;
    return Lv.ord.n_colors() > TILED_NCMAX || Gs.view.w_max > TILED_WMAX;
}
static const WgsBuf* wgs_plan(const smg_hierarchy* h, int lv, int k)
{
    if (!wgs_wanted(h, lv, k)) return nullptr;
    const WgsBuf& B = h->lv[lv].wgs;
    return B.view.n_pieces > 0 ? &B : nullptr;
}
static int ensure_wgs(smg_hierarchy* h, int lv)
{
    Level& Lv = h->lv[lv];
    WgsBuf& B = Lv.wgs;
    if (B.tried) return SMG_OK;
    B.tried = true;
    static const int rows_env = env_int("SMG_WGS_ROWS", WGS_ROWS), mode_env = env_int("SMG_WGS_PIECES", 1);
    std::vector<int> tsrc;
    Csr AT;
    { int rc = ensure_A_int(h, lv); if (rc) return rc; }
    if (Lv.gs_on_transpose) AT = transpose(Lv.A_int, &tsrc);
    const Csr& G = Lv.gs_on_transpose ? AT : Lv.A_int;
    const auto t_plan0 = std::chrono::steady_clock::now();
    WgsPlan P = build_wgs(G, std::min(std::max(rows_env, 8), (int)WGS_ROWS), mode_env);
    const double plan_ms = 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t_plan0).count();
    if (P.empty()) return SMG_OK;
    auto to_level_value = [&](const std::vector<int>& entries) {
        std::vector<int> m(entries.size());
        for (size_t i = 0; i < m.size(); i++) {
            const int e = entries[i];
            m[i] = e < 0 ? -1 : Lv.A_int_src[(size_t)(Lv.gs_on_transpose ? tsrc[(size_t)e] : e)];
        }
        return m;
    };
    const std::vector<int> map = to_level_value(P.eentry), mapd = to_level_value(P.dentry);
    HIPCHK(B.hdr.upload(P.hdr)); HIPCHK(B.grow.upload(P.grow)); HIPCHK(B.meta.upload(P.meta)); HIPCHK(B.diag.upload(P.diag)); HIPCHK(B.rim.upload(P.rim));
    HIPCHK(B.eoff.upload(P.eoff)); HIPCHK(B.eval.upload(P.eval)); B.host_map = map; B.host_mapd = mapd;
    B.view.n_pieces = P.n_pieces; B.view.n_colors = P.n_colors; B.view.rim_pitch = P.rim_pitch; B.view.nb_max = P.nb_max;
    B.view.hdr = B.hdr.p; B.view.grow = B.grow.p; B.view.meta = B.meta.p; B.view.diag = B.diag.p; B.view.rim = B.rim.p; B.view.eoff = B.eoff.p; B.view.eval = B.eval.p;
    B.color_ptr = P.color_ptr; B.host_rows = P.rows; B.host_piece_ptr = P.piece_ptr; B.rim_ratio = P.rim_ratio; B.phases_mean = P.phases_mean; B.phases_max = P.phases_max;
    // after a value-only re-precompute the host copy of the values is stale: take them from the device copy (padding slots keep their +0.0, lanes without a row their 1.0)
    if (h->host_stale && Lv.d_Aval.p) {
        HIPCHK(ensure_map(B.map, B.host_map)); HIPCHK(ensure_map(B.mapd, B.host_mapd));
        HIPCHK(launch_gather_vals(B.eval.p, Lv.d_Aval.p, B.map.p, B.eval.n, h->stream));
        HIPCHK(launch_gather_vals(B.diag.p, Lv.d_Aval.p, B.mapd.p, B.diag.n, h->stream));
    }
    if (env_int("SMG_DEBUG_WGS", 0))
        std::fprintf(stderr, "wave Gauss-Seidel level %d: %d rows, %d pieces in %d colours, phases per piece %.1f (max %d), rim %.2f rows read per row beyond the iterate, rim pitch %d; plan built in %.0f ms\n",
                     lv, Lv.n, P.n_pieces, P.n_colors, P.phases_mean, P.phases_max, P.rim_ratio, P.rim_pitch, plan_ms);
    return SMG_OK;
}

// Called by the first precompute for a level whose images exist, while its device half would otherwise wait for the host half (smg_precompute.cpp):
// the plans prepare_tiled() below would build at the first solve with the handle's present selection (smoother, pre / post sweeps), one column.
// Small levels only (a plan of tens of milliseconds at most: the 63 210-row Galerkin level of decimated C3 64 ms, its 252 834-row level 220 ms -- building
// that one here kept level 0's images waiting and cost the precompute more than it saved the first solve).  First smg_solve on a fresh handle:
// bunny.obj 8.4 -> 4.2 ms, ogre.obj 17 -> 11 ms (tools/first_solve.py).
int smg::prepare_level_plans(smg_hierarchy* h, int lv)
{
    static const int on = env_int("SMG_EARLY_PLANS", 1), max_rows = env_int("SMG_EARLY_PLANS_MAX_ROWS", 70000);      // A/B knobs
    if (!on || lv <= 0 || lv >= h->n_levels - 1 || h->precision != 0 || h->lv[lv].n > max_rows) return SMG_OK;
    const int sa = h->pre, sb = h->post;
    for (int sw : {sa, sb}) if (sw > 0 && tiled_wanted(h, lv, 1, sw)) { int rc = ensure_tiled(h, lv, sw); if (rc) return rc; }
    if (wgs_wanted(h, lv, 1) && !h->lv[lv].wgs.tried && !tiled_plan(h, lv, 1, sa) && !tiled_plan(h, lv, 1, sb)) { int rc = ensure_wgs(h, lv); if (rc) return rc; }
    return SMG_OK;
}

// plans + second iterate for relax(sa) / relax(sb) wherever they are wanted (host work and uploads: never inside a graph capture)
static int prepare_tiled(smg_hierarchy* h, int k, int sa, int sb)
{
    for (int lv = 0; lv < h->n_levels - 1; lv++) {
        Level& Lv = h->lv[lv];
        for (int sw : {sa, sb}) if (tiled_wanted(h, lv, k, sw)) { int rc = ensure_tiled(h, lv, sw); if (rc) return rc; }
        if (bgs_wanted(h, lv, k) && !Lv.bgs.tried) { drop_graphs(h); int rc = ensure_bgs(h, lv); if (rc) return rc; }
        if (wgs_wanted(h, lv, k) && !Lv.wgs.tried && !tiled_plan(h, lv, k, sa) && !tiled_plan(h, lv, k, sb)) { drop_graphs(h); int rc = ensure_wgs(h, lv); if (rc) return rc; }
        if ((tiled_plan(h, lv, k, sa) || tiled_plan(h, lv, k, sb)) && Lv.t.n < (size_t)Lv.n * std::max(h->kcap, 1)) {
            drop_graphs(h);
            HIPCHK(Lv.t.alloc((size_t)Lv.n * std::max(h->kcap, 1)));
            HIPCHK(hipMemsetAsync(Lv.t.p, 0, Lv.t.n * sizeof(double), h->stream));
        }
    }
    return SMG_OK;
}

static int ensure_work(smg_hierarchy* h, int k)
{
    const int L = h->n_levels;
    if (k > h->kcap) {
        drop_graphs(h);
        size_t maxblocks = 0;
        for (int lv = 0; lv < L; lv++) {
            Level& Lv = h->lv[lv];
            size_t rows = (lv == L - 1) ? (size_t)h->nc_pad : (size_t)Lv.n;
            HIPCHK(Lv.b.alloc(rows * k));
            HIPCHK(Lv.u.alloc(rows * k));
            HIPCHK(hipMemsetAsync(Lv.b.p, 0, rows * k * sizeof(double), h->stream));
            HIPCHK(hipMemsetAsync(Lv.u.p, 0, rows * k * sizeof(double), h->stream));
            Lv.t.release(); Lv.d.release();
            if (lv < L - 1 || L == 1) HIPCHK(Lv.r.alloc(rows * k));
            if (lv < L - 1 || L == 1) {
                if (h->bs == 3) maxblocks = std::max(maxblocks, (size_t)bsr3_blocks(Lv.bA.view.n_slices) * (size_t)k);
                else maxblocks = std::max(maxblocks, (size_t)sell_blocks(Lv.dA.view.n_slices) * ((k + 3) / 4) + (size_t)sell_wide_blocks(Lv.dA.view.n_slices, k));
            }
        }
        // colour by colour (the level-0 head of an outer iteration, enqueue_residual_ss) every launch rounds its block count up on its own
        maxblocks += (h->lv[0].dA.color_slice_ptr.size() + 1) * (size_t)((k + 3) / 4 + 8);
        HIPCHK(h->d_partials.alloc(std::max<size_t>(maxblocks, 1) + (size_t)ss_partials_room()));
        h->kcap = k;
    }
    // Jacobi-smoothed levels ping-pong between u and a second iterate
    for (int lv = 0; lv < L - 1; lv++) {
        Level& Lv = h->lv[lv];
        // (level 0 always: the first sweep of an outer iteration is written out of place, see enqueue_residual_ss)
        if ((level_is_jacobi(h, lv) || lv == 0) && Lv.t.n < (size_t)Lv.n * h->kcap) {
            drop_graphs(h);
            HIPCHK(Lv.t.alloc((size_t)Lv.n * h->kcap));
            HIPCHK(hipMemsetAsync(Lv.t.p, 0, (size_t)Lv.n * h->kcap * sizeof(double), h->stream));
        }
        if (level_kind(h, lv) == LV_CHEBY && Lv.d.n < (size_t)Lv.n * h->kcap) {
            drop_graphs(h);
            HIPCHK(Lv.d.alloc((size_t)Lv.n * h->kcap));
            HIPCHK(hipMemsetAsync(Lv.d.p, 0, (size_t)Lv.n * h->kcap * sizeof(double), h->stream));
        }
    }
    if (h->coarse_schur) {       // the separator's right-hand side and solution (the solver may have arrived with a value-only re-precompute, after the vectors)
        const size_t need = (size_t)h->sch.view.ns_pad * std::max(h->kcap, 1);
        if (h->sch.g.n < need || h->sch.xs.n < need) {
            drop_graphs(h);
            HIPCHK(h->sch.g.alloc(need));
            HIPCHK(h->sch.xs.alloc(need));
            HIPCHK(hipMemsetAsync(h->sch.g.p, 0, need * sizeof(double), h->stream));
            HIPCHK(hipMemsetAsync(h->sch.xs.p, 0, need * sizeof(double), h->stream));
        }
        h->sch.view.g = h->sch.g.p; h->sch.view.xs = h->sch.xs.p;
    }
    if (h->coarse_sparse) {      // the triangular solves take up to 64 columns per pass: 2 n doubles of scratch per column of a pass
        const size_t need = (size_t)2 * h->chol.n * sparse_coarse_work_cols(std::max(h->kcap, 1));
        if (h->c_work.n < need) {
            drop_graphs(h);          // the captured launches hold the scratch pointer
            HIPCHK(h->c_work.alloc(need));
            h->c_view.work = h->c_work.p;
        }
    }
    int rc = prepare_tiled(h, k, h->pre, h->post);
    if (rc) return rc;
    return ensure_spectral_bounds(h);
}

// ---- mixed precision: fp32 images of the operators and an fp32 V-cycle ------------------------------------------------
static int ensure_fp32(smg_hierarchy* h, int k)
{
    const int L = h->n_levels;
    if (h->union_m > 0) return fail(SMG_ERR_INVALID, "a union handle solves in fp64 (its coarse inverses are per-member blocks: no fp32 image)");
    if (h->coarse_sparse) return fail(SMG_ERR_INVALID, "the mixed-precision cycle is not available with a sparse coarse factorisation (coarsest level of %d unknowns)", h->nc);
    if (!h->f32_valid) {
        drop_graphs(h);
        auto mk = [&](SellBuf& src, DevBuf<float>& dst, SellDev& view) -> int {
            if (src.view.long_n > 0) {     // the long rows' values, too
                HIPCHK(src.long_valf.ensure(src.long_val.n));
                HIPCHK(launch_cvt_f64_f32(src.long_valf.p, src.long_val.p, src.long_val.n, h->stream));
                src.view.long_valf = src.long_valf.p;
            }
            view = src.view;
            if (src.padded == 0) { view.valf = nullptr; return SMG_OK; }
            HIPCHK(dst.ensure((size_t)src.padded));
            HIPCHK(launch_cvt_f64_f32(dst.p, src.view.val, (size_t)src.padded, h->stream));
            view.valf = dst.p;
            return SMG_OK;
        };
        for (int lv = 0; lv < L; lv++) {
            Level& Lv = h->lv[lv];
            int rc;
            if (lv < L - 1 && h->bs == 3) {     // block hierarchies: the nine value planes of every panel column
                auto mkb = [&](Bsr3Buf& B) -> int {
                    HIPCHK(B.valf.ensure(B.val.n));
                    HIPCHK(launch_cvt_f64_f32(B.valf.p, B.val.p, B.val.n, h->stream));
                    B.view.valf = B.valf.p;
                    return SMG_OK;
                };
                if ((rc = mkb(Lv.bA))) return rc;
                if (Lv.gs_on_transpose) { if ((rc = mkb(Lv.bAT))) return rc; }
            } else if (lv < L - 1) {
                if ((rc = mk(Lv.dA, Lv.a32, Lv.dA32))) return rc;
                if (Lv.gs_on_transpose) { if ((rc = mk(Lv.dAT, Lv.at32, Lv.dAT32))) return rc; }
            }
            if (lv >= 1) {
                if ((rc = mk(Lv.dP, Lv.p32, Lv.dP32))) return rc;
                if ((rc = mk(Lv.dPT, Lv.pt32, Lv.dPT32))) return rc;
            }
        }
        if (h->coarse_schur) {
            HIPCHK(h->sch.arena32.ensure((size_t)h->schur.off_C));          // blocks, panels and the separator's inverse (the products behind them are scratch)
            HIPCHK(launch_cvt_f64_f32(h->sch.arena32.p, h->sch.arena.p, (size_t)h->schur.off_C, h->stream));
            h->sch.view.arena32 = h->sch.arena32.p;
        } else {
            HIPCHK(h->d_Ainv32.ensure((size_t)h->nc_pad * h->nc_pad));
            HIPCHK(launch_cvt_f64_f32(h->d_Ainv32.p, h->d_Ainv.p, (size_t)h->nc_pad * h->nc_pad, h->stream));
        }
        h->f32_valid = true;
    }
    if (k > h->kcap32) {
        drop_graphs(h);
        for (int lv = 0; lv < L; lv++) {
            Level& Lv = h->lv[lv];
            const size_t rows = (lv == L - 1) ? (size_t)h->nc_pad : (size_t)Lv.n;
            HIPCHK(Lv.b32.alloc(rows * k));
            HIPCHK(Lv.u32.alloc(rows * k));
            HIPCHK(hipMemsetAsync(Lv.b32.p, 0, rows * k * sizeof(float), h->stream));
            HIPCHK(hipMemsetAsync(Lv.u32.p, 0, rows * k * sizeof(float), h->stream));
            if (lv < L - 1) HIPCHK(Lv.r32.alloc(rows * k));
            Lv.t32.release(); Lv.d32.release();
        }
        h->kcap32 = k;
    }
    for (int lv = 0; lv < L - 1; lv++) {
        Level& Lv = h->lv[lv];
        if (level_is_jacobi(h, lv) && Lv.t32.n < (size_t)Lv.n * h->kcap32) {
            drop_graphs(h);
            HIPCHK(Lv.t32.alloc((size_t)Lv.n * h->kcap32));
            HIPCHK(hipMemsetAsync(Lv.t32.p, 0, (size_t)Lv.n * h->kcap32 * sizeof(float), h->stream));
        }
        if (level_kind(h, lv) == LV_CHEBY && Lv.d32.n < (size_t)Lv.n * h->kcap32) {
            drop_graphs(h);
            HIPCHK(Lv.d32.alloc((size_t)Lv.n * h->kcap32));
            HIPCHK(hipMemsetAsync(Lv.d32.p, 0, (size_t)Lv.n * h->kcap32 * sizeof(float), h->stream));
        }
    }
    if (h->coarse_schur) {
        const size_t need = (size_t)h->sch.view.ns_pad * std::max(h->kcap32, 1);
        if (h->sch.g32.n < need || h->sch.xs32.n < need) {
            drop_graphs(h);
            HIPCHK(h->sch.g32.alloc(need));
            HIPCHK(h->sch.xs32.alloc(need));
            HIPCHK(hipMemsetAsync(h->sch.g32.p, 0, need * sizeof(float), h->stream));
            HIPCHK(hipMemsetAsync(h->sch.xs32.p, 0, need * sizeof(float), h->stream));
        }
        h->sch.view.g32 = h->sch.g32.p; h->sch.view.xs32 = h->sch.xs32.p;
    }
    return SMG_OK;
}

// ---- the smoother of a level -------------------------------------------------------------------------------------------
// SMG_SMOOTH_GS (default): the reference's relax().  SMG_SMOOTH_JACOBI / _HYBRID: damped Jacobi on all / on the small levels
// (BASELINE.json north_star: "Gauss-Seidel/Jacobi smoothing"; one whole-matrix launch per sweep instead of one per colour).
int smg::level_kind(const smg_hierarchy* h, int lv)
{
    if (lv < 0 || lv >= h->n_levels - 1) return LV_GS;
    switch (h->smoother) {
        case SMG_SMOOTH_JACOBI: return LV_JACOBI;
        case SMG_SMOOTH_HYBRID: return h->lv[lv].n <= h->jacobi_max_rows ? LV_JACOBI : LV_GS;
        case SMG_SMOOTH_CHEBYSHEV: return LV_CHEBY;
        case SMG_SMOOTH_HYBRID_CHEBYSHEV: return h->lv[lv].n <= h->jacobi_max_rows ? LV_CHEBY : LV_GS;
    }
    return LV_GS;
}

// Coefficients of the Chebyshev-Jacobi recurrence (include/smg.h, SMG_SMOOTH_CHEBYSHEV): step s computes d = c1 d + c2 r, u += d.
// The same statements, in the same order, as the CPU restatement used by the tests -- both are compiled without FMA contraction.
struct ChebyCoef { double c1, c2; };
static void cheby_coefs(double lam, double frac, int degree, std::vector<ChebyCoef>& out)
{
    out.resize((size_t)std::max(degree, 0));
    const double lmax = lam, lmin = lam * frac;
    const double theta = (lmax + lmin) / 2.0, delta = (lmax - lmin) / 2.0;
    const double sigma = theta / delta;
    double rho = 1.0 / sigma;
    for (int s = 0; s < degree; s++) {
        if (s == 0) { out[s].c1 = 0.0; out[s].c2 = 1.0 / theta; }
        else {
            const double rho_new = 1.0 / (2.0 * sigma - rho);
            out[s].c1 = rho_new * rho;
            out[s].c2 = 2.0 * rho_new / delta;
            rho = rho_new;
        }
    }
}

// one accessor set per arithmetic: fp64 (the reference's) and the fp32 images of the mixed-precision V-cycle
template <typename T> struct Prec;
// The dense coarse product of a padded solve (internal_cols below) is formed for the caller's columns only: which of its kernels serves a column
// (1 / 2 - 7 / 8 and more columns) then follows the CALLER's column count, as it would without padding -- a column-sharded solve stays bit-identical
// to the fused one (tests/test_gpu_dist.py), and the padding columns' coarse iterate stays the zero the restriction wrote.
static inline int coarse_cols(const smg_hierarchy* h, int k) { return (h->coarse_cols > 0 && h->coarse_cols < k) ? h->coarse_cols : k; }

template <> struct Prec<double> {
    static double* b(Level& L) { return L.b.p; }
    static double* u(Level& L) { return L.u.p; }
    static double* r(Level& L) { return L.r.p; }
    static double* t(Level& L) { return L.t.p; }
    static double* d(Level& L) { return L.d.p; }
    static void set_d(FirstColour& fc, Level& L) { fc.d = L.d.p; }
    static const SellDev& A(Level& L) { return L.dA.view; }
    static const SellDev& G(Level& L) { return L.gs_on_transpose ? L.dAT.view : L.dA.view; }   // what the smoother streams
    static const SellDev& P(Level& L) { return L.dP.view; }
    static const SellDev& PT(Level& L) { return L.dPT.view; }
    static bool has_vals(const SellDev& V) { return V.val != nullptr; }
    static hipError_t sell(SellMode m, const SellDev& V, int s0, int s1, const double* x, const double* bb, double* y, int k, const Ctrl* ctrl,
                           hipStream_t st, double* zero_rows = nullptr, const FirstColour* first = nullptr, double omega = 1.0)
    { return launch_sell(m, V, s0, s1, x, bb, y, k, ctrl, nullptr, nullptr, st, zero_rows, first, omega); }
    static hipError_t coarse(smg_hierarchy* h, Level& L, int k, const Ctrl* ctrl)
    {
        if (h->union_m > 0) return launch_blockdiag_gemv_add(h->un.view, h->d_Ainv.p, h->nc, L.b.p, L.u.p, k, ctrl, h->stream);   // the members' own inverses (smg_union.cpp)
        if (h->coarse_sparse) return launch_sparse_coarse_solve(h->c_view, L.b.p, L.u.p, k, ctrl, h->stream);
        if (h->coarse_schur) return launch_schur_solve(h->sch.view, L.b.p, L.u.p, k, ctrl, h->stream);
        return launch_dense_gemv_add(h->d_Ainv.p, h->nc, h->nc_pad, L.b.p, L.u.p, coarse_cols(h, k), k, ctrl, h->stream, h->d_sympart.p);
    }
    // an operation with the level's matrix in whatever format it lives in: SELL panels, or 3 x 3 blocks on block hierarchies.
    // smoother_image: the matrix the smoother streams (A^T where A is not bit-symmetric), else A.  s1 < 0: all slices.
    static hipError_t opA(smg_hierarchy* h, Level& L, bool smoother_image, SellMode m, int s0, int s1, const double* x, const double* bb, double* y, int k,
                          const Ctrl* ctrl, const FirstColour* fc = nullptr, double omega = 1.0)
    {
        if (h->bs == 3) {
            const Bsr3Dev& V = (smoother_image && L.gs_on_transpose) ? L.bAT.view : L.bA.view;
            return launch_bsr3(m, V, s0, s1 < 0 ? V.n_slices : s1, x, bb, y, k, ctrl, nullptr, nullptr, h->stream, omega, fc ? fc->c1 : 0.0, fc ? fc->d : nullptr);
        }
        const SellDev& V = smoother_image ? G(L) : A(L);
        return sell(m, V, s0, s1 < 0 ? V.n_slices : s1, x, bb, y, k, ctrl, h->stream, nullptr, fc, omega);
    }
};
template <> struct Prec<float> {
    static float* b(Level& L) { return L.b32.p; }
    static float* u(Level& L) { return L.u32.p; }
    static float* r(Level& L) { return L.r32.p; }
    static float* t(Level& L) { return L.t32.p; }
    static float* d(Level& L) { return L.d32.p; }
    static void set_d(FirstColour& fc, Level& L) { fc.df = L.d32.p; }
    static const SellDev& A(Level& L) { return L.dA32; }
    static const SellDev& G(Level& L) { return L.gs_on_transpose ? L.dAT32 : L.dA32; }
    static const SellDev& P(Level& L) { return L.dP32; }
    static const SellDev& PT(Level& L) { return L.dPT32; }
    static bool has_vals(const SellDev& V) { return V.valf != nullptr; }
    static hipError_t sell(SellMode m, const SellDev& V, int s0, int s1, const float* x, const float* bb, float* y, int k, const Ctrl* ctrl,
                           hipStream_t st, float* zero_rows = nullptr, const FirstColour* first = nullptr, double omega = 1.0)
    { return launch_sell_f32(m, V, s0, s1, x, bb, y, k, ctrl, st, zero_rows, first, omega); }
    static hipError_t coarse(smg_hierarchy* h, Level& L, int k, const Ctrl* ctrl)
    {
        if (h->coarse_schur) return launch_schur_solve_f32(h->sch.view, L.b32.p, L.u32.p, k, ctrl, h->stream);
        return launch_dense_gemv_add_f32(h->d_Ainv32.p, h->nc, h->nc_pad, L.b32.p, L.u32.p, coarse_cols(h, k), k, ctrl, h->stream, (float*)h->d_sympart.p);
    }
    static hipError_t opA(smg_hierarchy* h, Level& L, bool smoother_image, SellMode m, int s0, int s1, const float* x, const float* bb, float* y, int k,
                          const Ctrl* ctrl, const FirstColour* fc = nullptr, double omega = 1.0)
    {
        if (h->bs == 3) {
            const Bsr3Dev& V = (smoother_image && L.gs_on_transpose) ? L.bAT.view : L.bA.view;
            return launch_bsr3_f32(m, V, s0, s1 < 0 ? V.n_slices : s1, x, bb, y, k, ctrl, h->stream, omega, fc ? fc->c1 : 0.0, fc ? fc->df : nullptr);
        }
        const SellDev& V = smoother_image ? G(L) : A(L);
        return sell(m, V, s0, s1 < 0 ? V.n_slices : s1, x, bb, y, k, ctrl, h->stream, nullptr, fc, omega);
    }
};

// slice ranges of the colours of the matrix the smoother streams
static const std::vector<int>& colour_slices(const smg_hierarchy* h, const Level& Lv)
{
    if (h->bs == 3) return (Lv.gs_on_transpose ? Lv.bAT : Lv.bA).color_slice_ptr;
    return (Lv.gs_on_transpose ? Lv.dAT : Lv.dA).color_slice_ptr;
}

// what of a level's first pre-smoothing sweep exists when its V-cycle starts
enum { FIRST_NONE = 0,
       FIRST_LAUNCH = 1,   // its first launch, produced by the restriction launch of the finer level (FirstColour): the first colour
                           // (Gauss-Seidel, in Lv.u) or the whole first sweep / step (Jacobi / Chebyshev, in Lv.t)
       FIRST_SWEEP = 2 };  // level 0 inside an outer iteration: the whole first sweep / step, produced out of place into Lv.t by the
                           // launches that also formed the outer residual (enqueue_head)

// `iters` forward Gauss-Seidel sweeps in place: one launch per colour (reference relax(), src/mg_VCycle.cpp:113-178)
// first = FIRST_LAUNCH: the first colour of the first sweep is already in u.  FIRST_SWEEP: the whole first sweep is in `t`: the second
// sweep goes from t back into u (out-of-place colour launches: same values), the rest run in place on u; needs iters >= 2.
template <typename T>
static int enqueue_gs(smg_hierarchy* h, int lv, const T* b, T* u, int k, int iters, const Ctrl* ctrl, int first = FIRST_NONE, T* t = nullptr)
{
    Level& Lv = h->lv[lv];
    ProfGuard pg(h, "MG: relaxation");  // PROFC_NODE at src/mg_VCycle.cpp:121
    if (std::is_same<T, double>::value && first == FIRST_NONE) {
        if (const WgsBuf* W = wgs_plan(h, lv, k)) {      // Galerkin levels of decimated hierarchies: one launch per piece colour (smg_wgs.hpp)
            for (int it = 0; it < iters; it++)
                for (size_t c = 0; c + 1 < W->color_ptr.size(); c++)
                    HIPCHK(launch_wgs(W->view, W->color_ptr[c], W->color_ptr[c + 1], (const double*)b, (double*)u, k, ctrl, h->stream));
            return SMG_OK;
        }
        if (const BgsBuf* Q = bgs_plan(h, lv, k)) {      // many columns: one launch per block colour (smg_bgs.hpp)
            for (int it = 0; it < iters; it++)
                for (size_t c = 0; c + 1 < Q->color_ptr.size(); c++)
                    HIPCHK(launch_bgs(Q->view, Q->color_ptr[c], Q->color_ptr[c + 1], (const double*)b, (double*)u, k, ctrl, h->stream));
            return SMG_OK;
        }
    }
    const std::vector<int>& cs = colour_slices(h, Lv);
    for (int it = first == FIRST_SWEEP ? 1 : 0; it < iters; it++)
        for (size_t c = (it == 0 && first == FIRST_LAUNCH) ? 1 : 0; c + 1 < cs.size(); c++) {
            if (it == 1 && first == FIRST_SWEEP) HIPCHK(Prec<T>::sell(SELL_GS_OOP, Prec<T>::G(Lv), cs[c], cs[c + 1], t, b, u, k, ctrl, h->stream));   // (scalar levels only)
            else HIPCHK(Prec<T>::opA(h, Lv, true, SELL_GS, cs[c], cs[c + 1], u, b, u, k, ctrl));
        }
    return SMG_OK;
}

// relax(iters) as one launch (overlapped tiling): from buf[*cur] into the other buffer, flips *cur.  fp64 only.
static int enqueue_gs_tiled(smg_hierarchy* h, int lv, const TiledDev& plan, const double* b, double* const buf[2], int* cur, int k, const Ctrl* ctrl)
{
    ProfGuard pg(h, "MG: relaxation");
    HIPCHK(launch_tiled_gs(plan, buf[*cur], b, buf[1 - *cur], k, ctrl, h->stream));
    *cur ^= 1;
    return SMG_OK;
}
template <typename T> static const TiledDev* tiled_for(const smg_hierarchy*, Level&, int, int, int) { return nullptr; }
template <> const TiledDev* tiled_for<double>(const smg_hierarchy* h, Level& Lv, int lv, int k, int sweeps)
{
    const TiledDev* p = tiled_plan(h, lv, k, sweeps);
    return (p && Lv.t.p) ? p : nullptr;
}
template <typename T> static int enqueue_gs_tiled_t(smg_hierarchy*, int, const TiledDev&, const T*, T* const*, int*, int, const Ctrl*) { return SMG_ERR_INVALID; }
template <> int enqueue_gs_tiled_t<double>(smg_hierarchy* h, int lv, const TiledDev& plan, const double* b, double* const* buf, int* cur, int k, const Ctrl* ctrl)
{
    double* const two[2] = {buf[0], buf[1]};
    return enqueue_gs_tiled(h, lv, plan, b, two, cur, k, ctrl);
}

// `iters` damped-Jacobi sweeps, ping-pong between buf[0] and buf[1]: sweep s reads buf[*cur], writes the other, flips *cur.
template <typename T>
static int enqueue_jacobi(smg_hierarchy* h, int lv, const T* b, T* const buf[2], int* cur, int k, int iters, const Ctrl* ctrl)
{
    Level& Lv = h->lv[lv];
    ProfGuard pg(h, "MG: relaxation");
    for (int it = 0; it < iters; it++) {
        HIPCHK(Prec<T>::opA(h, Lv, true, SELL_JACOBI, 0, -1, buf[*cur], b, buf[1 - *cur], k, ctrl, nullptr, h->omega));
        *cur ^= 1;
    }
    return SMG_OK;
}

// relax(iters) on a Chebyshev-Jacobi level: ONE polynomial of degree iters + 1, i.e. iters + 1 whole-matrix launches ping-ponging like
// the Jacobi sweeps; first_done: step 0 was produced by the restriction launch.
template <typename T>
static int enqueue_cheby(smg_hierarchy* h, int lv, const T* b, T* const buf[2], int* cur, int k, int iters, const Ctrl* ctrl, bool first_done = false)
{
    if (iters <= 0) return SMG_OK;
    Level& Lv = h->lv[lv];
    ProfGuard pg(h, "MG: relaxation");
    std::vector<ChebyCoef> cf;
    cheby_coefs(Lv.lam, h->cheby_fraction, iters + 1, cf);
    for (int s = first_done ? 1 : 0; s <= iters; s++) {
        FirstColour fc;
        Prec<T>::set_d(fc, Lv);
        fc.c1 = cf[s].c1;
        HIPCHK(Prec<T>::opA(h, Lv, true, SELL_CHEBY, 0, -1, buf[*cur], b, buf[1 - *cur], k, ctrl, &fc, cf[s].c2));
        *cur ^= 1;
    }
    return SMG_OK;
}

// reference mg_VCycle(), src/mg_VCycle.cpp:3-59.  B and u of level lv are Lv.b / Lv.u (level 0: RHS_u / z_u).
static bool fuse_first_colour() { static const int on = env_int("SMG_FUSE_FIRST", 1); return on != 0; }

// first: what of this level's first pre-smoothing sweep already exists (FIRST_*).
template <typename T>
static int enqueue_vcycle_t(smg_hierarchy* h, int lv, int k, int pre, int post, const Ctrl* ctrl, int first = FIRST_NONE)
{
    const bool first_done = first != FIRST_NONE;
    const int L = h->n_levels;
    Level& Lv = h->lv[lv];
    if (lv == L - 1) {  // coarseSolve: u = u + solver.solve(B)  (:28-33, :199-200)
        ProfGuard pg(h, "MG: coarse solve");
        HIPCHK(Prec<T>::coarse(h, Lv, k, ctrl));
        return SMG_OK;
    }
    Level& Lc = h->lv[lv + 1];
    const int kind = level_kind(h, lv);
    const bool jac = kind != LV_GS;
    T* const buf[2] = {Prec<T>::u(Lv), Prec<T>::t(Lv)};   // Jacobi-type levels ping-pong; the level's result always ends in buf[0] = u
    int cur = 0;
    // Gauss-Seidel levels whose relax() runs as one out-of-place launch (overlapped tiling): they ping-pong like the Jacobi-type ones.
    // (Not when the first sweep already exists: level 0 inside an outer iteration, FIRST_SWEEP.)
    const TiledDev* tl_pre = (kind == LV_GS && pre > 0 && first != FIRST_SWEEP) ? tiled_for<T>(h, Lv, lv, k, pre) : nullptr;
    const TiledDev* tl_post = (kind == LV_GS && post > 0) ? tiled_for<T>(h, Lv, lv, k, post) : nullptr;
    int rc;
    if (kind == LV_JACOBI) {
        if (first_done) cur = 1;
        rc = enqueue_jacobi<T>(h, lv, Prec<T>::b(Lv), buf, &cur, k, pre - (first_done ? 1 : 0), ctrl);            // :36
    } else if (kind == LV_CHEBY) {
        if (first_done) cur = 1;
        rc = enqueue_cheby<T>(h, lv, Prec<T>::b(Lv), buf, &cur, k, pre, ctrl, first_done);                         // :36
    } else if (tl_pre) rc = enqueue_gs_tiled_t<T>(h, lv, *tl_pre, Prec<T>::b(Lv), buf, &cur, k, ctrl);               // :36, one launch
    else rc = enqueue_gs<T>(h, lv, Prec<T>::b(Lv), buf[0], k, pre, ctrl, first, buf[1]);                          // :36
    if (rc) return rc;
    {   // r = B - A u  (:40-42)
        ProfGuard pg(h, "MG: residual");
        HIPCHK(Prec<T>::opA(h, Lv, false, SELL_RESID, 0, -1, buf[cur], Prec<T>::b(Lv), Prec<T>::r(Lv), k, ctrl));
    }
    // With uc = 0 the first launch of the coarse level's first pre-smoothing sweep computes (rc_i - 0) / a_ii for the rows it covers
    // (the first colour / with Jacobi all rows, damped): the restriction launch writes that itself, bit for bit the same value, and
    // the sweep starts one launch later.
    const SellBuf& Gc = Lc.gs_on_transpose ? Lc.dAT : Lc.dA;
    const int kind_c = level_kind(h, lv + 1);
    const bool jac_c = kind_c != LV_GS;
    // (block hierarchies: the first launch of a coarse sweep is not a plain division -- row 3v+1 of the first colour already reads 3v)
    const bool tiled_c = level_kind(h, lv + 1) == LV_GS && pre > 0 && tiled_for<T>(h, Lc, lv + 1, k, pre) != nullptr;   // the coarse level runs all its phases itself
    const bool fuse = h->bs == 1 && !tiled_c && !(std::is_same<T, double>::value && (bgs_plan(h, lv + 1, k) || wgs_plan(h, lv + 1, k))) && fuse_first_colour() && lv + 1 < L - 1 && pre > 0 && Prec<T>::has_vals(Prec<T>::G(Lc)) && (jac_c ? Gc.n_all > 0 : Gc.n_first > 0);
    const int kt = k * h->bs;   // block hierarchies: dP / dPT hold the vertex-level factor of P (x) I_3, applied to 3 k columns
    {   // rc = PT r  (:43-44, :80) and uc = 0 (:46-47) in one launch: both are indexed by the coarse row
        ProfGuard pg(h, "MG: restrict");
        FirstColour fc;
        if (fuse) {
            fc.diag_slot = Gc.diag_slot.p; fc.n_first = jac_c ? Gc.n_all : Gc.n_first;
            fc.val = Prec<T>::G(Lc).val; fc.valf = Prec<T>::G(Lc).valf;
            fc.jacobi = kind_c == LV_CHEBY ? 2 : (jac_c ? 1 : 0); fc.omega = h->omega;
            if (kind_c == LV_CHEBY) {   // step 0 of the coarse level's polynomial: d = (rc_i / a_ii - 0) / theta, uc = 0 + d
                std::vector<ChebyCoef> cf;
                cheby_coefs(Lc.lam, h->cheby_fraction, 1, cf);
                fc.omega = cf[0].c2;
                Prec<T>::set_d(fc, Lc);
            }
        }
        // Jacobi + fuse: the first sweep's output buffer (t) receives the sweep, u = 0 is never read
        T* init = (fuse && jac_c) ? Prec<T>::t(Lc) : Prec<T>::u(Lc);
        HIPCHK(Prec<T>::sell(SELL_AX, Prec<T>::PT(Lc), 0, Prec<T>::PT(Lc).n_slices, Prec<T>::r(Lv), nullptr, Prec<T>::b(Lc), kt, ctrl, h->stream, init,
                             fuse ? &fc : nullptr));
    }
    rc = enqueue_vcycle_t<T>(h, lv + 1, k, pre, post, ctrl, fuse ? FIRST_LAUNCH : FIRST_NONE);  // :48
    if (rc) return rc;
    {   // u = u + P uc  (:51-53, :91).  A Jacobi level with an odd number of post-smoothing sweeps to go adds out of place, so that
        // the last sweep lands in u.
        ProfGuard pg(h, "MG: prolong");
        int dst = cur;
        const int flips = tl_post ? 1 : kind == LV_CHEBY ? (post > 0 ? post + 1 : 0) : post;   // buffer switches of the post-smoothing
        if ((jac || tl_post) && ((cur + flips) & 1)) dst = 1 - cur;
        if (!jac && !tl_post && cur == 1) dst = 0;      // in-place Gauss-Seidel sweeps follow: they work on u
        HIPCHK(Prec<T>::sell(SELL_ADD, Prec<T>::P(Lc), 0, Prec<T>::P(Lc).n_slices, Prec<T>::u(Lc), buf[cur], buf[dst], kt, ctrl, h->stream));
        cur = dst;
    }
    if (kind == LV_CHEBY) return enqueue_cheby<T>(h, lv, Prec<T>::b(Lv), buf, &cur, k, post, ctrl);   // :57  (ends with cur == 0)
    if (jac) return enqueue_jacobi<T>(h, lv, Prec<T>::b(Lv), buf, &cur, k, post, ctrl);   // :57  (ends with cur == 0)
    if (tl_post) return enqueue_gs_tiled_t<T>(h, lv, *tl_post, Prec<T>::b(Lv), buf, &cur, k, ctrl);   // :57  (ends with cur == 0)
    return enqueue_gs<T>(h, lv, Prec<T>::b(Lv), buf[0], k, post, ctrl);                    // :57
}

static int enqueue_vcycle(smg_hierarchy* h, int lv, int k, int pre, int post, const Ctrl* ctrl, int first = FIRST_NONE)
{
    return enqueue_vcycle_t<double>(h, lv, k, pre, post, ctrl, first);
}
static int enqueue_vcycle32(smg_hierarchy* h, int lv, int k, int pre, int post, const Ctrl* ctrl)
{
    return enqueue_vcycle_t<float>(h, lv, k, pre, post, ctrl);
}

// relax() on caller-provided device vectors (pieces, raw interface): the result always ends in u
static int enqueue_relax(smg_hierarchy* h, int lv, const double* b, double* u, int k, int iters, const Ctrl* ctrl)
{
    if (!level_is_jacobi(h, lv)) {
        Level& Lg = h->lv[lv];
        if (const TiledDev* tl = tiled_for<double>(h, Lg, lv, k, iters)) {
            double* const two[2] = {u, Lg.t.p};
            int c2 = 0;
            int rc = enqueue_gs_tiled(h, lv, *tl, b, two, &c2, k, ctrl);
            if (rc) return rc;
            HIPCHK(hipMemcpyAsync(u, Lg.t.p, (size_t)Lg.n * k * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
            return SMG_OK;
        }
        return enqueue_gs<double>(h, lv, b, u, k, iters, ctrl);
    }
    Level& Lv = h->lv[lv];
    double* const buf[2] = {u, Lv.t.p};
    int cur = 0;
    int rc = level_kind(h, lv) == LV_CHEBY ? enqueue_cheby<double>(h, lv, b, buf, &cur, k, iters, ctrl)
                                           : enqueue_jacobi<double>(h, lv, b, buf, &cur, k, iters, ctrl);
    if (rc) return rc;
    if (cur == 1) HIPCHK(hipMemcpyAsync(u, Lv.t.p, (size_t)Lv.n * k * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    return SMG_OK;
}

// The outer residual of iterate z (min_quad_with_fixed_mg.cpp:110) and the first pre-smoothing sweep of the V-cycle that follows
// (mg_VCycle.cpp:36) stream the same matrix against the same z: when this returns true the sweep's launches form both -- the sweep's
// result out of place in L0.t (z itself stays intact for the case that the break test stops the loop), the squared residual through a
// second accumulator that repeats SELL_RESID_SS's additions (SELL_*_HEAD in smg_device.hpp) -- and the cycle starts with FIRST_SWEEP.
// fp64 cycles only (the mixed mode's residual IS the right-hand side of its fp32 cycle); Gauss-Seidel needs a second sweep to come back
// into u; a level 0 that smooths on A^T (non-symmetric storage) forms other sums than the residual.
static bool head_fusable(smg_hierarchy* h, int k)
{
    static const int on = env_int("SMG_FUSE_HEAD", 1);
    if (!on || h->precision != 0 || h->n_levels < 2 || h->prof_on || h->bs != 1 || h->union_m > 0) return false;      // (a union needs the residual VECTOR: per-member norms)
    Level& L0 = h->lv[0];
    if (L0.gs_on_transpose) return false;
    if (bgs_plan(h, 0, k) || wgs_plan(h, 0, k)) return false;   // block- / piece-sequential sweeps run in place; their head is the residual launch
    // a level 0 whose relax(pre) is ONE launch (overlapped tiling): residual launch + one launch beat a head of (colours x 2) launches
    if (level_kind(h, 0) == LV_GS && tiled_plan(h, 0, k, h->pre) && L0.t.p) return false;
    const int kind = level_kind(h, 0);
    return kind == LV_GS ? h->pre >= 2 : h->pre >= 1;
}

// sum of squares of RHS_u - A_0 z_u into ctrl->sumsq  (min_quad_with_fixed_mg.cpp:110 / :332)
static int enqueue_residual_ss(smg_hierarchy* h, int k, bool fuse_decide = false, double* sumsq_out = nullptr)
{
    Level& L0 = h->lv[0];
    int nb = 0;
    if (h->head_fuse) {
        ProfGuard pg(h, "MG: relaxation");
        const int kind = level_kind(h, 0);
        const SellDev& G = L0.dA.view;
        if (kind == LV_GS) {
            const std::vector<int>& cs = L0.dA.color_slice_ptr;
            for (size_t c = 0; c + 1 < cs.size(); c++) {
                int nbc = 0;
                HIPCHK(launch_sell(SELL_GS_HEAD, G, cs[c], cs[c + 1], L0.u.p, L0.b.p, L0.t.p, k, h->d_ctrl.p, h->d_partials.p + nb, &nbc, h->stream));
                nb += nbc;
            }
        } else if (kind == LV_JACOBI) {
            HIPCHK(launch_sell(SELL_JACOBI_HEAD, G, 0, G.n_slices, L0.u.p, L0.b.p, L0.t.p, k, h->d_ctrl.p, h->d_partials.p, &nb, h->stream, nullptr, nullptr, h->omega));
        } else {
            std::vector<ChebyCoef> cf;
            cheby_coefs(L0.lam, h->cheby_fraction, h->pre + 1, cf);
            FirstColour fc;
            fc.d = L0.d.p;
            fc.c1 = cf[0].c1;
            HIPCHK(launch_sell(SELL_CHEBY_HEAD, G, 0, G.n_slices, L0.u.p, L0.b.p, L0.t.p, k, h->d_ctrl.p, h->d_partials.p, &nb, h->stream, nullptr, &fc, cf[0].c2));
        }
        if (fuse_decide) HIPCHK(launch_ss_finalize_decide(h->d_partials.p, nb, h->d_ctrl.p, h->stream));
        else HIPCHK(launch_ss_finalize(h->d_partials.p, nb, h->d_ctrl.p, h->stream, sumsq_out));
        return SMG_OK;
    }
    ProfGuard pg(h, "MG: outer residual");
    if (h->union_m > 0) {
        // independent meshes in one handle: r = RHS - A z as a vector, then every member's own norm, history and break test (smg_union_device.hip)
        if (!fuse_decide) return fail(SMG_ERR_INVALID, "a union handle runs through smg_solve / smg_solve_begin + smg_raw_outer_iteration (no split-phase iteration: its members stop one by one)");
        HIPCHK(launch_sell(SELL_RESID, L0.dA.view, 0, L0.dA.view.n_slices, L0.u.p, L0.b.p, L0.r.p, k, h->d_ctrl.p, nullptr, nullptr, h->stream));
        HIPCHK(launch_union_sumsq_decide(h->un.view, L0.r.p, L0.u.p, k, h->d_ctrl.p, h->stream));
        return SMG_OK;
    }
    if (h->precision == 1 && h->bs == 3)
        HIPCHK(launch_bsr3(SELL_RESID_BOTH, L0.bA.view, 0, L0.bA.view.n_slices, L0.u.p, L0.b.p, L0.r.p, k, h->d_ctrl.p, h->d_partials.p, &nb, h->stream));
    else if (h->precision == 1)   // mixed: the residual itself is the right-hand side of the fp32 correction cycle
        HIPCHK(launch_sell(SELL_RESID_BOTH, L0.dA.view, 0, L0.dA.view.n_slices, L0.u.p, L0.b.p, L0.r.p, k, h->d_ctrl.p, h->d_partials.p, &nb, h->stream));
    else if (h->bs == 3)
        HIPCHK(launch_bsr3(SELL_RESID_SS, L0.bA.view, 0, L0.bA.view.n_slices, L0.u.p, L0.b.p, nullptr, k, h->d_ctrl.p, h->d_partials.p, &nb, h->stream));
    else
        HIPCHK(launch_sell(SELL_RESID_SS, L0.dA.view, 0, L0.dA.view.n_slices, L0.u.p, L0.b.p, nullptr, k, h->d_ctrl.p, h->d_partials.p, &nb, h->stream));
    if (fuse_decide) HIPCHK(launch_ss_finalize_decide(h->d_partials.p, nb, h->d_ctrl.p, h->stream));
    else HIPCHK(launch_ss_finalize(h->d_partials.p, nb, h->d_ctrl.p, h->stream, sumsq_out));
    return SMG_OK;
}

// d_sumsq == nullptr: the break test already ran inside the residual launch (single-GPU path)
static int enqueue_cycle_part(smg_hierarchy* h, int k, const double* d_sumsq)
{
    if (d_sumsq) HIPCHK(launch_decide(h->d_ctrl.p, d_sumsq, h->stream));
    {
        ProfGuard pg(h, "MG: total VCycle");  // PROFC_NODE at src/min_quad_with_fixed_mg.cpp:123
        if (h->precision == 1) {
            // z += V32(r): the V-cycle is affine in (B, u), so V(B, z) = z + V(B - A z, 0) in exact arithmetic
            Level& L0 = h->lv[0];
            const size_t cnt = (size_t)L0.n * k;
            HIPCHK(launch_residual_to_f32(L0.b32.p, L0.u32.p, L0.r.p, cnt, h->d_ctrl.p, h->stream));
            int rc = enqueue_vcycle32(h, 0, k, h->pre, h->post, h->d_ctrl.p);
            if (rc) return rc;
            HIPCHK(launch_add_correction(L0.u.p, L0.u32.p, cnt, h->d_ctrl.p, h->stream));
        } else {
            int rc = enqueue_vcycle(h, 0, k, h->pre, h->post, h->d_ctrl.p, h->head_fuse ? FIRST_SWEEP : FIRST_NONE);
            if (rc) return rc;
            if (h->union_m > 0) HIPCHK(launch_union_restore(h->un.view, h->lv[0].u.p, k, h->d_ctrl.p, h->stream));   // members whose loop has ended keep their iterate
        }
    }
    return SMG_OK;
}

template <typename Fn>
static int capture_graph(smg_hierarchy* h, hipGraphExec_t* out, Fn&& body)
{
    hipGraph_t g = nullptr;
    HIPCHK(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    int rc = body();
    hipError_t e = hipStreamEndCapture(h->stream, &g);
    if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
    if (e != hipSuccess) return fail(SMG_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
    e = hipGraphInstantiate(out, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) return fail(SMG_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
    return SMG_OK;
}

// The two halves of a split-phase iteration work on ONE buffer that the caller all-reduces in between: the residual graph leaves
// the local sum of squares there, the cycle graph's break test reads the reduced value from there (no staging copies: an 8-byte
// device-to-device copy costs several microseconds of stream time).  Re-captured when the caller hands in another buffer.
static int capture_split_graphs(smg_hierarchy* h, double* buf)
{
    if (h->g_resid) { (void)hipGraphExecDestroy(h->g_resid); h->g_resid = nullptr; }
    if (h->g_cycle) { (void)hipGraphExecDestroy(h->g_cycle); h->g_cycle = nullptr; }
    const int k = h->k;
    int rc = capture_graph(h, &h->g_resid, [&]() { return enqueue_residual_ss(h, k, false, buf); });
    if (rc) return rc;
    rc = capture_graph(h, &h->g_cycle, [&]() { return enqueue_cycle_part(h, k, buf); });
    if (rc) return rc;
    h->g_sumsq_ptr = buf;
    return SMG_OK;
}

static GraphKey current_graph_key(const smg_hierarchy* h)
{
    GraphKey key;
    key.k = h->k; key.k_user = h->k_user; key.pre = h->pre; key.post = h->post; key.precision = h->precision; key.smoother = h->smoother;
    key.jacobi_max_rows = h->jacobi_max_rows; key.omega = h->omega; key.cheby_fraction = h->cheby_fraction; key.head_fuse = h->head_fuse;
    return key;
}

static int graph_iters() { static const int v = std::max(1, std::min(16, env_int("SMG_GRAPH_ITERS", 4))); return v; }
static int ensure_graphs(smg_hierarchy* h)
{
    const GraphKey key = current_graph_key(h);
    if (h->g_iter && h->g_key == key) return SMG_OK;
    drop_graphs(h);
    const int k = h->k;
    int rc = capture_graph(h, &h->g_iter, [&]() {
        int r = enqueue_residual_ss(h, k, true);
        if (r) return r;
        return enqueue_cycle_part(h, k, nullptr);
    });
    if (rc) return rc;
    // Between two graph launches the stream idles for the runtime's hand-over (8.7 us in the rocprof timeline of a 316 us iteration); several iterations
    // in one graph pay it once.  Semantics unchanged: every launch of an iteration after the one whose break test fired writes nothing (Ctrl::done), as
    // for iterations enqueued ahead of the host's polling.  SMG_GRAPH_ITERS (default 4; 1 = off): C3 headline 3 168 (1) / 3 174 (2) / 3 194 (4) V-cycles/s, same box, alternating.
    if (graph_iters() > 1) {
        rc = capture_graph(h, &h->g_iter_n, [&]() {
            for (int i = 0; i < graph_iters(); i++) {
                int r = enqueue_residual_ss(h, k, true);
                if (r) return r;
                if ((r = enqueue_cycle_part(h, k, nullptr))) return r;
            }
            return (int)SMG_OK;
        });
        if (rc) return rc;
    }
    if (!h->union_m) {      // (a union has no split-phase iteration: its members stop one by one)
        rc = capture_split_graphs(h, h->g_sumsq_ptr ? h->g_sumsq_ptr : &h->d_ctrl.p->sumsq);
        if (rc) return rc;
        // the two halves of an iteration the host looks into (enqueue_checked_iteration)
        rc = capture_graph(h, &h->g_rd, [&]() { return enqueue_residual_ss(h, k, true); });
        if (rc) return rc;
        rc = capture_graph(h, &h->g_cyc, [&]() { return enqueue_cycle_part(h, k, nullptr); });
        if (rc) return rc;
    }
    h->g_key = key;
    return SMG_OK;
}

// hipStreamBeginCapture is not allowed on the legacy default stream (smg_hierarchy_set_stream(h, NULL)): eager launches there
static bool graphs_usable(const smg_hierarchy* h) { return h->use_graph && !h->prof_on && h->stream != nullptr; }

// n full outer iterations, single-GPU form
static int enqueue_outer_iteration(smg_hierarchy* h);
static int enqueue_outer_iterations(smg_hierarchy* h, int n)
{
    if (graphs_usable(h) && graph_iters() > 1 && n >= graph_iters()) {
        int rc = ensure_graphs(h);
        if (rc) return rc;
        for (; n >= graph_iters(); n -= graph_iters()) { HIPCHK(hipGraphLaunch(h->g_iter_n, h->stream)); h->iters_enqueued += graph_iters(); }
    }
    for (; n > 0; n--) { int rc = enqueue_outer_iteration(h); if (rc) return rc; }
    return SMG_OK;
}
// one full outer iteration, single-GPU form
static int enqueue_outer_iteration(smg_hierarchy* h)
{
    if (graphs_usable(h)) {
        int rc = ensure_graphs(h);
        if (rc) return rc;
        HIPCHK(hipGraphLaunch(h->g_iter, h->stream));
    } else {
        int rc = enqueue_residual_ss(h, h->k, true);
        if (rc) return rc;
        rc = enqueue_cycle_part(h, h->k, nullptr);
        if (rc) return rc;
    }
    h->iters_enqueued++;
    return SMG_OK;
}

// the control block as the stream has it now (one synchronisation); through page-locked memory
static int read_ctrl(smg_hierarchy* h, Ctrl* out)
{
    HIPCHK(h->pin_ctrl.ensure(1));
    HIPCHK(hipMemcpyAsync(h->pin_ctrl.p, h->d_ctrl.p, sizeof(Ctrl), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    *out = *h->pin_ctrl.p;
    return SMG_OK;
}

// One outer iteration the host looks INTO: residual + break test, a look at the flag, and the V-cycle only if the loop goes on.  An iteration
// enqueued whole runs its cycle even when its own break test has just fired (every launch after the break stores nothing, but does its work):
// the last iteration of every solve -- 0.31 ms at C3, of a 3.6 ms solve; a whole cycle more than the one a tol = 1e-3 solve of a small mesh
// needs.  The host looks at the flag after every chunk of iterations anyway; where the chunk is a single iteration (the end of every solve
// under the adaptive schedule), the look moves in front of the cycle.  Same launches in the same order as the whole iteration.
static int enqueue_checked_iteration(smg_hierarchy* h, Ctrl* seen)
{
    const bool graphs = graphs_usable(h);
    if (graphs) {
        int rc = ensure_graphs(h);
        if (rc) return rc;
        HIPCHK(hipGraphLaunch(h->g_rd, h->stream));
    } else {
        int rc = enqueue_residual_ss(h, h->k, true);
        if (rc) return rc;
    }
    h->iters_enqueued++;      // (its residual is recorded whether or not the cycle follows)
    { int rc = read_ctrl(h, seen); if (rc) return rc; }      // (the flag and what the adaptive schedule reads)
    if (seen->done) return SMG_OK;
    if (graphs) HIPCHK(hipGraphLaunch(h->g_cyc, h->stream));
    else { int rc = enqueue_cycle_part(h, h->k, nullptr); if (rc) return rc; }
    return SMG_OK;
}

// ------------------------------------------------------------------------------------------------ solve
// The sparse triangular solves raise c_err when a wait gave up (smg_coarse_device.hip): the values they then wrote are NaN.  Every entry point
// that has just synchronised with work that may contain such a solve reads the flag, clears it (it is sticky on the device: later waits give
// up at once while it is set) and fails with SMG_ERR_HIP.  The stream is idle when this runs.
static int coarse_stall_check(smg_hierarchy* h)
{
    if (!h->coarse_sparse || !h->c_err.p) return SMG_OK;
    int cerr = 0;
    HIPCHK(hipMemcpy(&cerr, h->c_err.p, sizeof(int), hipMemcpyDeviceToHost));
    if (!cerr) return SMG_OK;
    (void)hipMemset(h->c_err.p, 0, sizeof(int));
    return fail(SMG_ERR_HIP, "the triangular solves of the sparse coarse factorisation stalled (results are NaN)");
}

int smg::check_ready(const smg_hierarchy* h, const char* who)
{
    if (!h) return fail(SMG_ERR_INVALID, "%s: null handle", who);
    if (!h->precomputed) return fail(SMG_ERR_INVALID, "%s: call smg_precompute first", who);
    if (h->device < 0) return fail(SMG_ERR_NO_DEVICE, "%s: no HIP device", who);
    return SMG_OK;
}

// Columns of the solve's internal (row-major n x kin) blocks.  The kernels for 8 and more columns read a row's columns as 64- to 512-byte
// segments; a row length that is no multiple of such a segment puts every row across cache-line boundaries and splits the columns over a
// wide and one or two narrow launches per operation.  C3, ms per outer iteration (tools/k_solve_time.py, same box, SMG_PAD_COLS=0 / 1):
//   k = 5: 1.068 / 1.044   6: 1.180 / 1.045   7: 1.393 / 1.066   (8: 0.97)   12: 1.981 / 1.602   13: 2.857 / 1.602   (16: 1.60)
//   24: 2.91 / 2.87   48: 4.59 / 4.34   (64: 4.33);  33 columns as 64: 4.10 -> 4.37 -- not padded.
// So 5 - 32 columns run as the next power of two and 41 - 63 as 64, with zero columns as padding: a zero right-hand side and iterate stay
// exactly zero through every kernel of the cycle and add exact zeros to the residual's sum of squares.  The sparse kernels compute every
// column independently of how many others there are, and the dense coarse product is formed for the caller's columns only (coarse_cols
// above), so the caller's columns come out bit for bit as without padding (the tool prints a checksum of z; SMG_PAD_COLS=0: A/B knob).
// (Schur / sparse coarse solvers take the padded block as it is.)  Union handles keep their own per-member bookkeeping and are not padded.
static int internal_cols(const smg_hierarchy* h, int k)
{
    static const int on = env_int("SMG_PAD_COLS", 1);
    if (!on || k <= 4 || k > 64 || h->union_m > 0) return k;
    if (k > 32) return k > 40 ? 64 : k;
    int p = 8;
    while (p < k) p *= 2;
    return p;
}

// host blocks of up to 1 MiB travel through page-locked staging (pin_vec): packed by the host, one DMA each way
static bool small_host_block(int n, int k) { return (size_t)n * k * 8 <= ((size_t)1 << 20); }

static int smg_solve_begin_impl(smg_hierarchy* h, const double* RHS, int ld_rhs, const double* known_val, int ld_kv,
                               const double* z0, int ld_z0, int k, int memspace, const smg_solve_opts* opts)
{
    int rc = check_ready(h, "smg_solve_begin");
    if (rc) return rc;
    smg_solve_opts o;
    smg_solve_opts_default(&o);
    if (opts) o = *opts;
    const int n = h->n_full;
    if (!RHS || !z0 || k < 1 || ld_rhs < n || ld_z0 < n) return fail(SMG_ERR_INVALID, "smg_solve: bad RHS/z0/k/ld");
    if (o.max_iter < 0) return fail(SMG_ERR_INVALID, "max_iter must be >= 0");
    if (h->has_known && (!known_val || ld_kv < (int)h->known.size())) return fail(SMG_ERR_INVALID, "known_val missing or ld_kv too small");
    // everything is validated before anything of the handle changes: a refused call leaves the handle as it was
    if (o.precision != 0 && o.precision != 1) return fail(SMG_ERR_INVALID, "precision must be 0 (fp64) or 1 (mixed)");
    if (o.pre < 0 || o.post < 0) return fail(SMG_ERR_INVALID, "pre / post must be >= 0");
    if (o.smoother < SMG_SMOOTH_GS || o.smoother > SMG_SMOOTH_HYBRID_CHEBYSHEV) return fail(SMG_ERR_INVALID, "smoother must be one of SMG_SMOOTH_*");
    if (o.omega > 2.0 || o.omega != o.omega) return fail(SMG_ERR_INVALID, "omega must be in (0, 2]");
    if (o.cheby_fraction >= 1.0 || o.cheby_fraction != o.cheby_fraction) return fail(SMG_ERR_INVALID, "cheby_fraction must be in (0, 1)");
    if (h->in_solve) return fail(SMG_ERR_INVALID, "smg_solve_begin: a split-phase solve is already in progress (smg_solve_end)");
    if (h->union_m > 0 && o.precision != 0) return fail(SMG_ERR_INVALID, "a union handle solves in fp64 (no mixed-precision cycle)");
    h->tol = o.tol; h->max_iter = o.max_iter; h->pre = o.pre; h->post = o.post; h->verbosity = o.verbosity;
    h->check_every = std::max(0, o.check_every); h->use_graph = o.use_graph;
    h->precision = o.precision;
    if ((rc = smg_hierarchy_set_smoother(h, o.smoother, o.omega, o.jacobi_max_rows))) return rc;
    if ((rc = smg_hierarchy_set_chebyshev(h, o.cheby_fraction))) return rc;
    DeviceScope dsc(h->device);
    const int kin = internal_cols(h, k);
    rc = ensure_work(h, kin);
    if (rc) return rc;
    if (h->precision == 1 && (rc = ensure_fp32(h, kin))) return rc;
    h->k = kin; h->k_user = k;
    h->coarse_cols = kin > k ? k : 0;
    const int nk = (int)h->known.size();
    // stage host inputs
    const double *dR = RHS, *dZ = z0, *dK = known_val;
    int ldR = ld_rhs, ldZ = ld_z0, ldK = ld_kv;
    if (memspace == SMG_HOST) {
        HIPCHK(h->d_stage_rhs.ensure((size_t)n * k));
        HIPCHK(h->d_stage_z.ensure((size_t)n * k));
        if (small_host_block(n, k)) {
            // small blocks: packed into page-locked memory by the host, then ONE copy each
            HIPCHK(h->pin_vec.ensure((size_t)2 * n * k));
            for (int c = 0; c < k; c++) {
                std::memcpy(h->pin_vec.p + (size_t)c * n, RHS + (size_t)c * ld_rhs, (size_t)n * 8);
                std::memcpy(h->pin_vec.p + (size_t)(k + c) * n, z0 + (size_t)c * ld_z0, (size_t)n * 8);
            }
            HIPCHK(hipMemcpyAsync(h->d_stage_rhs.p, h->pin_vec.p, (size_t)n * k * 8, hipMemcpyHostToDevice, h->stream));
            HIPCHK(hipMemcpyAsync(h->d_stage_z.p, h->pin_vec.p + (size_t)n * k, (size_t)n * k * 8, hipMemcpyHostToDevice, h->stream));
        } else {
            HIPCHK(hipMemcpy2DAsync(h->d_stage_rhs.p, (size_t)n * 8, RHS, (size_t)ld_rhs * 8, (size_t)n * 8, k, hipMemcpyHostToDevice, h->stream));
            HIPCHK(hipMemcpy2DAsync(h->d_stage_z.p, (size_t)n * 8, z0, (size_t)ld_z0 * 8, (size_t)n * 8, k, hipMemcpyHostToDevice, h->stream));
        }
        dR = h->d_stage_rhs.p; dZ = h->d_stage_z.p; ldR = n; ldZ = n;
        if (h->has_known) {
            HIPCHK(h->d_stage_kv.ensure((size_t)nk * k));
            HIPCHK(hipMemcpy2DAsync(h->d_stage_kv.p, (size_t)nk * 8, known_val, (size_t)ld_kv * 8, (size_t)nk * 8, k, hipMemcpyHostToDevice, h->stream));
            dK = h->d_stage_kv.p; ldK = nk;
        }
    } else if (h->has_known) {
        // keep a private copy: the caller may reuse its buffer before smg_solve_end scatters z(known)
        HIPCHK(h->d_stage_kv.ensure((size_t)nk * k));
        HIPCHK(hipMemcpy2DAsync(h->d_stage_kv.p, (size_t)nk * 8, known_val, (size_t)ld_kv * 8, (size_t)nk * 8, k, hipMemcpyDeviceToDevice, h->stream));
        dK = h->d_stage_kv.p; ldK = nk;
    }
    h->cur_kv = dK; h->cur_ld_kv = ldK;
    Level& L0 = h->lv[0];
    // z_u = z0(unknown)  (:310-311)  /  z = z0 (:97)
    HIPCHK(launch_gather_in(L0.u.p, dZ, h->d_map0.p, L0.n, k, kin, ldZ, h->stream));
    if (h->has_known) {
        // RHS_u = RHS(unknown) - Auk * known_val  (:316-318)
        const int nu = L0.n;
        HIPCHK(h->d_tmp_cm.ensure((size_t)nu * k));
        HIPCHK(launch_gather_cm(h->d_tmp_cm.p, dR, h->d_unknown.p, nu, k, ldR, nu, h->stream));
        HIPCHK(launch_csr_sub(nu, h->d_auk_ptr.p, h->d_auk_col.p, h->d_auk_val.p, dK, ldK, h->d_tmp_cm.p, nu, k, h->stream));
        HIPCHK(launch_gather_in(L0.b.p, h->d_tmp_cm.p, h->d_perm0.p, nu, k, kin, nu, h->stream));
    } else {
        HIPCHK(launch_gather_in(L0.b.p, dR, h->d_map0.p, L0.n, k, kin, ldR, h->stream));
    }
    // the residual history lives in HBM, sized from max_iter (the reference's r_his grows with the loop, .cpp:112)
    HIPCHK(h->d_rhis.ensure((size_t)std::max(h->max_iter, 1)));
    Ctrl& zero = h->host_ctrl;   // lives in the handle: the asynchronous copy may read it after this call returns
    std::memset(&zero, 0, sizeof(zero));
    zero.tol = h->tol;
    zero.r_his = h->d_rhis.p;
    zero.his_cap = (int)std::min<size_t>(h->d_rhis.n, (size_t)std::max(h->max_iter, 1));
    HIPCHK(hipMemcpyAsync(h->d_ctrl.p, &zero, sizeof(Ctrl), hipMemcpyHostToDevice, h->stream));
    if (memspace == SMG_HOST) HIPCHK(hipStreamSynchronize(h->stream));  // the caller's host blocks may change after this call
    if (h->union_m > 0) {
        if ((rc = union_begin_solve(h, k))) return rc;
    }
    h->head_fuse = head_fusable(h, kin);   // latched: both halves of every iteration of this solve follow it
    h->iters_enqueued = 0;
    h->in_solve = true;
    return SMG_OK;
}

extern "C" int smg_solve_begin(smg_hierarchy* h, const double* RHS, int ld_rhs, const double* known_val, int ld_kv,
                               const double* z0, int ld_z0, int k, int memspace, const smg_solve_opts* opts)
{
    return guarded("smg_solve_begin", [&]() { return smg_solve_begin_impl(h, RHS, ld_rhs, known_val, ld_kv, z0, ld_z0, k, memspace, opts); });
}

extern "C" int smg_solve_iter_residual(smg_hierarchy* h, double* d_sumsq)
{
    if (!h || !h->in_solve) return fail(SMG_ERR_INVALID, "smg_solve_iter_residual: no solve in progress");
    DeviceScope dsc(h->device);
    double* buf = d_sumsq ? d_sumsq : &h->d_ctrl.p->sumsq;
    if (graphs_usable(h)) {
        int rc = ensure_graphs(h);
        if (rc) return rc;
        if (h->g_sumsq_ptr != buf) { rc = capture_split_graphs(h, buf); if (rc) return rc; }
        HIPCHK(hipGraphLaunch(h->g_resid, h->stream));
    } else {
        int rc = enqueue_residual_ss(h, h->k, false, buf);
        if (rc) return rc;
    }
    return SMG_OK;
}

extern "C" int smg_solve_iter_cycle(smg_hierarchy* h, const double* d_sumsq)
{
    if (!h || !h->in_solve) return fail(SMG_ERR_INVALID, "smg_solve_iter_cycle: no solve in progress");
    DeviceScope dsc(h->device);
    double* buf = d_sumsq ? const_cast<double*>(d_sumsq) : &h->d_ctrl.p->sumsq;
    if (graphs_usable(h)) {
        int rc = ensure_graphs(h);
        if (rc) return rc;
        if (h->g_sumsq_ptr != buf) { rc = capture_split_graphs(h, buf); if (rc) return rc; }
        HIPCHK(hipGraphLaunch(h->g_cycle, h->stream));
    } else {
        int rc = enqueue_cycle_part(h, h->k, buf);
        if (rc) return rc;
    }
    h->iters_enqueued++;
    return SMG_OK;
}

// save z, V-cycle in place -- nothing here reads the reduced residual
static int enqueue_cycle_speculative(smg_hierarchy* h)
{
    Level& L0 = h->lv[0];
    const size_t cnt = (size_t)L0.n * h->k;
    HIPCHK(launch_copy_unless_done(h->d_zsave.p, L0.u.p, cnt, h->d_ctrl.p, h->stream));
    return enqueue_cycle_part(h, h->k, nullptr);   // nullptr: no decide in front of the cycle
}

extern "C" int smg_solve_iter_cycle_speculative(smg_hierarchy* h)
{
    if (!h || !h->in_solve) return fail(SMG_ERR_INVALID, "smg_solve_iter_cycle_speculative: no solve in progress");
    DeviceScope dsc(h->device);
    HIPCHK(h->d_zsave.ensure((size_t)h->lv[0].n * h->k));
    if (graphs_usable(h)) {
        int rc = ensure_graphs(h);
        if (rc) return rc;
        if (!h->g_spec) { rc = capture_graph(h, &h->g_spec, [&]() { return enqueue_cycle_speculative(h); }); if (rc) return rc; }
        HIPCHK(hipGraphLaunch(h->g_spec, h->stream));
    } else {
        int rc = enqueue_cycle_speculative(h);
        if (rc) return rc;
    }
    h->iters_enqueued++;
    return SMG_OK;
}

extern "C" int smg_solve_iter_commit(smg_hierarchy* h, const double* d_sumsq)
{
    if (!h || !h->in_solve) return fail(SMG_ERR_INVALID, "smg_solve_iter_commit: no solve in progress");
    DeviceScope dsc(h->device);
    Level& L0 = h->lv[0];
    HIPCHK(launch_decide_spec(h->d_ctrl.p, d_sumsq ? d_sumsq : &h->d_ctrl.p->sumsq, h->stream));
    HIPCHK(launch_restore_if_just_done(L0.u.p, h->d_zsave.p, (size_t)L0.n * h->k, h->d_ctrl.p, h->stream));
    return SMG_OK;
}

extern "C" int smg_solve_poll(smg_hierarchy* h, int* done, int* n_his)
{
    if (!h || !h->in_solve) return fail(SMG_ERR_INVALID, "smg_solve_poll: no solve in progress");
    DeviceScope dsc(h->device);
    int hdr[4];
    HIPCHK(hipMemcpyAsync(hdr, h->d_ctrl.p, sizeof(hdr), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (done) *done = hdr[0];
    if (n_his) *n_his = hdr[1];
    return SMG_OK;
}

extern "C" int smg_solve_end(smg_hierarchy* h, double* z, int ld_z, int memspace, double* r_his, int* n_his, int* converged)
{
    if (!h || !h->in_solve) return fail(SMG_ERR_INVALID, "smg_solve_end: no solve in progress");
    DeviceScope dsc(h->device);
    const int n = h->n_full, k = h->k_user;
    if (!z || ld_z < n) return fail(SMG_ERR_INVALID, "smg_solve_end: bad z / ld_z");
    Level& L0 = h->lv[0];
    double* dz = z;
    int ldz = ld_z;
    if (memspace == SMG_HOST) {
        HIPCHK(h->d_stage_z.ensure((size_t)n * k));
        dz = h->d_stage_z.p; ldz = n;
    }
    // z(unknown) = z_u ; z(known) = known_val  (:353-355)
    HIPCHK(launch_scatter_out(dz, L0.u.p, h->d_map0.p, L0.n, k, h->k, ldz, h->stream));
    if (h->has_known)
        HIPCHK(launch_scatter_cm(dz, h->cur_kv, h->d_known.p, (int)h->known.size(), k, h->cur_ld_kv, ldz, h->stream));
    const bool z_pinned = memspace == SMG_HOST && small_host_block(n, k);
    if (z_pinned) {
        HIPCHK(h->pin_vec.ensure((size_t)2 * n * k));
        HIPCHK(hipMemcpyAsync(h->pin_vec.p, dz, (size_t)n * k * 8, hipMemcpyDeviceToHost, h->stream));
    } else if (memspace == SMG_HOST)
        HIPCHK(hipMemcpy2DAsync(z, (size_t)ld_z * 8, dz, (size_t)n * 8, (size_t)n * 8, k, hipMemcpyDeviceToHost, h->stream));
    Ctrl hc;
    // the history can hold at most one entry per enqueued iteration: fetched together with the control block, one synchronisation
    const int cap = (int)std::min<size_t>(h->d_rhis.n, (size_t)std::max(std::min(h->iters_enqueued, std::max(h->max_iter, 1)), 1));
    HIPCHK(h->pin_his.ensure((size_t)cap));
    HIPCHK(h->pin_ctrl.ensure(1));
    const double* his = h->pin_his.p;
    HIPCHK(hipMemcpyAsync(h->pin_ctrl.p, h->d_ctrl.p, sizeof(Ctrl), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(h->pin_his.p, h->d_rhis.p, (size_t)cap * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    hc = *h->pin_ctrl.p;
    if (z_pinned) for (int c = 0; c < k; c++) std::memcpy(z + (size_t)c * ld_z, h->pin_vec.p + (size_t)c * n, (size_t)n * 8);
    const int cnt = std::max(0, std::min(std::min(hc.n_his, hc.his_cap), cap));
    h->in_solve = false; h->coarse_cols = 0;
    prof_collect(h);
    if (r_his) for (int i = 0; i < cnt; i++) r_his[i] = his[i];
    if (n_his) *n_his = cnt;
    const double last = cnt > 0 ? his[cnt - 1] : HUGE_VAL;
    if (converged) *converged = (last > h->tol) ? 0 : 1;  // :131-134 / :357-360
    if (h->union_m > 0 && converged) {      // every member's own loop ended below the tolerance (the handle's history holds the norm over all members)
        std::vector<int> md((size_t)h->union_m, 0);
        HIPCHK(hipMemcpy(md.data(), h->un.done.p, md.size() * sizeof(int), hipMemcpyDeviceToHost));
        *converged = (hc.status == 0 && std::all_of(md.begin(), md.end(), [](int d) { return d == 1; })) ? 1 : 0;      // 2 = that member's residual went non-finite
    }
    if (h->verbosity > 0) {
        for (int i = 0; i < cnt; i++) std::printf("MG iteration: %d, residual: %g\n", i, his[i]);  // :111
        if (cnt) std::printf("residual norm: %g\n", his[cnt - 1]);                                    // :127
    }
    { int rc = coarse_stall_check(h); if (rc) return rc; }
    if (hc.status != 0) return fail(SMG_ERR_NONFINITE, "non-finite residual at iteration %d", cnt - 1);
    return SMG_OK;
}

// for (iter < maxIter) { residual; push; if (residual < tol) break; V-cycle }   (:108-125 / :330-347)
// The break happens on the device; the host only decides how many iterations to enqueue before it looks at the flag again.
// check_every >= 1: that many.  check_every == 0 (default): adaptive -- from the two most recent residuals the host extrapolates
// how many more cycles the tolerance needs and enqueues all but the last of them before the next look (the results do not depend
// on this: an iteration enqueued after the break stores nothing).  The schedule is a function of the residual history alone, so the
// ranks of a column-sharded solve -- who all see the same reduced residuals -- enqueue (and reduce) the same number of times.
template <typename Iter>
static int run_outer_loop(smg_hierarchy* h, Iter&& iterations, bool look_into_single_iterations = false)
{
    int it = 0;
    int chunk_next = 1;
    static const int look_env = env_int("SMG_LOOK_INTO", 1);      // A/B knob
    const bool look = look_into_single_iterations && look_env != 0;
    while (it < h->max_iter) {
        const int want = h->check_every > 0 ? h->check_every : chunk_next;
        const int chunk = std::min(want, h->max_iter - it);
        Ctrl hc;
        if (look && chunk == 1) {
            // the look happens between the iteration's break test and its cycle; the cycle is enqueued behind it and the loop goes straight on
            int rc = enqueue_checked_iteration(h, &hc);
            if (rc) return rc;
            if (hc.done) break;
            it += 1;
        } else {
            { int rc = iterations(chunk); if (rc) return rc; }
            it += chunk;
            if (it >= h->max_iter) break;
            { int rc = read_ctrl(h, &hc); if (rc) return rc; }
            if (hc.done) break;
        }
        chunk_next = 1;
        if (h->check_every == 0 && hc.n_his >= 2 && hc.r_last > 0.0 && hc.r_last < hc.r_prev && h->tol > 0.0 && hc.r_last > h->tol) {
            const double need = std::ceil(std::log(h->tol / hc.r_last) / std::log(hc.r_last / hc.r_prev));   // more residuals until < tol
            if (need > 2.0) chunk_next = (int)std::min(need - 1.0, 64.0);
        }
    }
    return SMG_OK;
}

extern "C" int smg_solve(smg_hierarchy* h, const double* RHS, int ld_rhs, const double* known_val, int ld_kv,
                         const double* z0, int ld_z0, int k, int memspace, const smg_solve_opts* opts, double* z, int ld_z,
                         double* r_his, int* n_his, int* converged)
{
    int rc = smg_solve_begin(h, RHS, ld_rhs, known_val, ld_kv, z0, ld_z0, k, memspace, opts);
    if (rc) return rc;
    rc = run_outer_loop(h, [&](int n) { return enqueue_outer_iterations(h, n); }, h->union_m == 0);
    if (rc) { h->in_solve = false; h->coarse_cols = 0; return rc; }
    return smg_solve_end(h, z, ld_z, memspace, r_his, n_his, converged);
}

// ---- column-sharded solve (include/smg.h: smg_solve_sharded) ------------------------------------------------------------------------
// A rank without columns: no vectors, no cycle -- it adds 0 to every reduction and lets the device take the same decision from the
// reduced value as everybody else (same control block, same launch_decide, same polling schedule).
static int solve_sharded_empty(smg_hierarchy* h, const smg_solve_opts* opts, smg_reduce_fn reduce, void* ctx, double* r_his, int* n_his, int* converged)
{
    int rc = check_ready(h, "smg_solve_sharded");
    if (rc) return rc;
    if (h->in_solve) return fail(SMG_ERR_INVALID, "smg_solve_sharded: a split-phase solve is in progress");
    smg_solve_opts o;
    smg_solve_opts_default(&o);
    if (opts) o = *opts;
    if (o.max_iter < 0) return fail(SMG_ERR_INVALID, "max_iter must be >= 0");
    DeviceScope dsc(h->device);
    h->tol = o.tol; h->max_iter = o.max_iter; h->check_every = std::max(0, o.check_every); h->verbosity = o.verbosity;
    HIPCHK(h->d_rhis.ensure((size_t)std::max(h->max_iter, 1)));
    Ctrl& zero = h->host_ctrl;
    std::memset(&zero, 0, sizeof(zero));
    zero.tol = h->tol;
    zero.r_his = h->d_rhis.p;
    zero.his_cap = (int)std::min<size_t>(h->d_rhis.n, (size_t)std::max(h->max_iter, 1));
    HIPCHK(hipMemcpyAsync(h->d_ctrl.p, &zero, sizeof(Ctrl), hipMemcpyHostToDevice, h->stream));
    double* buf = &h->d_ctrl.p->sumsq;
    int n_it = 0;
    rc = run_outer_loop(h, [&](int n) -> int {
        for (int i = 0; i < n; i++) {
            HIPCHK(hipMemsetAsync(buf, 0, sizeof(double), h->stream));
            if (reduce(buf, 1, (void*)h->stream, ctx) != 0) return fail(SMG_ERR_REDUCE, "smg_solve_sharded: the caller's reduction failed");
            HIPCHK(launch_decide(h->d_ctrl.p, buf, h->stream));
            n_it++;
        }
        return SMG_OK;
    });
    if (rc) return rc;
    Ctrl hc;
    std::vector<double> his((size_t)std::max(std::min(n_it, std::max(h->max_iter, 1)), 1));
    HIPCHK(hipMemcpyAsync(&hc, h->d_ctrl.p, sizeof(Ctrl), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(his.data(), h->d_rhis.p, his.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    const int cnt = std::max(0, std::min(std::min(hc.n_his, hc.his_cap), (int)his.size()));
    if (r_his) for (int i = 0; i < cnt; i++) r_his[i] = his[(size_t)i];
    if (n_his) *n_his = cnt;
    const double last = cnt > 0 ? his[(size_t)cnt - 1] : HUGE_VAL;
    if (converged) *converged = (last > h->tol) ? 0 : 1;
    if (hc.status != 0) return fail(SMG_ERR_NONFINITE, "non-finite residual at iteration %d", cnt - 1);
    return SMG_OK;
}

extern "C" int smg_solve_sharded(smg_hierarchy* h, const double* RHS, int ld_rhs, const double* known_val, int ld_kv, const double* z0,
                                 int ld_z0, int k_local, int memspace, const smg_solve_opts* opts, smg_reduce_fn reduce, void* ctx,
                                 double* z, int ld_z, double* r_his, int* n_his, int* converged)
{
    return guarded("smg_solve_sharded", [&]() -> int {
        if (!reduce) return fail(SMG_ERR_INVALID, "smg_solve_sharded: no reduction given");
        if (k_local < 0) return fail(SMG_ERR_INVALID, "smg_solve_sharded: k_local must be >= 0");
        if (k_local == 0) return solve_sharded_empty(h, opts, reduce, ctx, r_his, n_his, converged);
        int rc = smg_solve_begin(h, RHS, ld_rhs, known_val, ld_kv, z0, ld_z0, k_local, memspace, opts);
        if (rc) return rc;
        DeviceScope dsc(h->device);
        double* buf = &h->d_ctrl.p->sumsq;   // the word both halves of an iteration work on in place; the reduction too
        rc = run_outer_loop(h, [&](int n) -> int {
            for (int i = 0; i < n; i++) {
                int r = smg_solve_iter_residual(h, buf);
                if (r) return r;
                if (reduce(buf, 1, (void*)h->stream, ctx) != 0) return fail(SMG_ERR_REDUCE, "smg_solve_sharded: the caller's reduction failed");
                if ((r = smg_solve_iter_cycle(h, buf))) return r;
            }
            return SMG_OK;
        });
        if (rc) { h->in_solve = false; h->coarse_cols = 0; return rc; }
        return smg_solve_end(h, z, ld_z, memspace, r_his, n_his, converged);
    });
}

extern "C" int smg_raw_outer_iteration(smg_hierarchy* h, int n_iter)
{
    if (!h || !h->in_solve) return fail(SMG_ERR_INVALID, "smg_raw_outer_iteration: call smg_solve_begin first");
    DeviceScope dsc(h->device);
    return enqueue_outer_iterations(h, n_iter);
}

static int piece_prolog(smg_hierarchy* h, int lv, int k, const char* who, bool need_coarser);

extern "C" int smg_bench_vcycle(smg_hierarchy* h, int lv, int k, int pre, int post, int reps, double* us_per_cycle)
{
    int rc = piece_prolog(h, lv, k, "smg_bench_vcycle", false);
    if (rc) return rc;
    DeviceScope dsc(h->device);
    if (reps < 1 || !us_per_cycle) return fail(SMG_ERR_INVALID, "smg_bench_vcycle: bad arguments");
    if ((rc = prepare_tiled(h, k, pre, post))) return rc;
    hipGraphExec_t g = nullptr;
    rc = capture_graph(h, &g, [&]() { return enqueue_vcycle(h, lv, k, pre, post, nullptr); });
    if (rc) return rc;
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) HIPCHK(hipGraphLaunch(g, h->stream));
    HIPCHK(hipEventRecord(e0, h->stream));
    for (int i = 0; i < reps; i++) HIPCHK(hipGraphLaunch(g, h->stream));
    HIPCHK(hipEventRecord(e1, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    *us_per_cycle = 1e3 * ms / reps;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipGraphExecDestroy(g);
    return coarse_stall_check(h);
}

extern "C" int smg_hierarchy_set_block_gs(smg_hierarchy* h, int min_rows)
{
    if (!h) return fail(SMG_ERR_INVALID, "smg_hierarchy_set_block_gs: null handle");
    if (h->in_solve) return fail(SMG_ERR_INVALID, "smg_hierarchy_set_block_gs called during a split-phase solve");
    if (min_rows != h->bgs_min_rows) { h->bgs_min_rows = min_rows; if (h->stream) drop_graphs(h); }
    return SMG_OK;
}
extern "C" int smg_level_get_block_gs_order(smg_hierarchy* h, int lv, int k, int* n_blocks, int* n_colors, int* color_ptr, int* blk_ptr, int* rows, double* stats)
{
    int rc = check_ready(h, "smg_level_get_block_gs_order");
    if (rc) return rc;
    if (lv < 0 || lv >= h->n_levels || k < 1) return fail(SMG_ERR_INVALID, "smg_level_get_block_gs_order: bad level / k");
    if (!bgs_wanted(h, lv, k)) return 0;
    if (!h->lv[lv].bgs.tried) { drop_graphs(h); if ((rc = ensure_bgs(h, lv))) return rc; }
    const BgsBuf* Q = bgs_plan(h, lv, k);
    if (!Q) return 0;
    if (n_blocks) *n_blocks = Q->view.n_blocks;
    if (n_colors) *n_colors = Q->view.n_colors;
    if (color_ptr) std::copy(Q->color_ptr.begin(), Q->color_ptr.end(), color_ptr);
    if (blk_ptr) std::copy(Q->host_blk_ptr.begin(), Q->host_blk_ptr.end(), blk_ptr);
    if (rows) std::copy(Q->host_rows.begin(), Q->host_rows.end(), rows);
    if (stats) { stats[0] = Q->rim; stats[1] = Q->fill; }
    return 1;
}

extern "C" int smg_hierarchy_set_wave_gs(smg_hierarchy* h, int mode)
{
    if (!h) return fail(SMG_ERR_INVALID, "smg_hierarchy_set_wave_gs: null handle");
    if (h->in_solve) return fail(SMG_ERR_INVALID, "smg_hierarchy_set_wave_gs called during a split-phase solve");
    if (mode < -1 || mode > 1) return fail(SMG_ERR_INVALID, "smg_hierarchy_set_wave_gs: mode must be -1 (automatic), 0 (never) or 1 (every Gauss-Seidel level in range)");
    if (mode != h->wgs_mode) { h->wgs_mode = mode; if (h->stream) drop_graphs(h); }
    return SMG_OK;
}
extern "C" int smg_level_get_wave_gs_order(smg_hierarchy* h, int lv, int k, int* n_pieces, int* n_colors, int* color_ptr, int* piece_ptr, int* rows, double* stats)
{
    int rc = check_ready(h, "smg_level_get_wave_gs_order");
    if (rc) return rc;
    if (lv < 0 || lv >= h->n_levels || k < 1) return fail(SMG_ERR_INVALID, "smg_level_get_wave_gs_order: bad level / k");
    if (!wgs_wanted(h, lv, k)) return 0;
    DeviceScope dsc(h->device);
    if ((rc = prepare_tiled(h, k, h->pre, h->post))) return rc;      // the one-launch relax() has precedence where it exists: decided there
    const WgsBuf* Q = wgs_plan(h, lv, k);
    if (!Q || tiled_plan(h, lv, k, h->pre) || tiled_plan(h, lv, k, h->post)) return 0;
    if (n_pieces) *n_pieces = Q->view.n_pieces;
    if (n_colors) *n_colors = Q->view.n_colors;
    if (color_ptr) std::copy(Q->color_ptr.begin(), Q->color_ptr.end(), color_ptr);
    if (piece_ptr) std::copy(Q->host_piece_ptr.begin(), Q->host_piece_ptr.end(), piece_ptr);
    if (rows) std::copy(Q->host_rows.begin(), Q->host_rows.end(), rows);
    if (stats) { stats[0] = Q->rim_ratio; stats[1] = Q->phases_mean; stats[2] = (double)Q->phases_max; }
    return 1;
}

extern "C" int smg_bench_relax(smg_hierarchy* h, int lv, int k, int sweeps, int reps, double* us_per_call)
{
    int rc = piece_prolog(h, lv, k, "smg_bench_relax", true);
    if (rc) return rc;
    DeviceScope dsc(h->device);
    if (reps < 1 || sweeps < 1 || !us_per_call) return fail(SMG_ERR_INVALID, "smg_bench_relax: bad arguments");
    if ((rc = prepare_tiled(h, k, sweeps, sweeps))) return rc;
    Level& Lv = h->lv[lv];
    hipGraphExec_t g = nullptr;
    rc = capture_graph(h, &g, [&]() { return enqueue_relax(h, lv, Lv.b.p, Lv.u.p, k, sweeps, nullptr); });
    if (rc) return rc;
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) HIPCHK(hipGraphLaunch(g, h->stream));
    HIPCHK(hipEventRecord(e0, h->stream));
    for (int i = 0; i < reps; i++) HIPCHK(hipGraphLaunch(g, h->stream));
    HIPCHK(hipEventRecord(e1, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    *us_per_call = 1e3 * ms / reps;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipGraphExecDestroy(g);
    return SMG_OK;
}

extern "C" int smg_synchronize(smg_hierarchy* h)
{
    if (!h || h->device < 0) return fail(SMG_ERR_INVALID, "smg_synchronize: no device");
    DeviceScope dsc(h->device);
    HIPCHK(hipStreamSynchronize(h->stream));
    return coarse_stall_check(h);
}

// ------------------------------------------------------------------------------------------------ V-cycle pieces (host blocks)
extern "C" int smg_level_rows(const smg_hierarchy* h, int lv)
{
    if (!h || lv < 0 || lv >= h->n_levels) return SMG_ERR_INVALID;
    return h->lv[lv].n;
}

// host column-major (caller numbering of level lv) -> device internal layout
static int put_block(smg_hierarchy* h, int lv, const double* src, int k, double* dst)
{
    const Level& Lv = h->lv[lv];
    std::vector<double> tmp((size_t)Lv.n * k);
    for (int i = 0; i < Lv.n; i++)
        for (int c = 0; c < k; c++) tmp[(size_t)i * k + c] = src[(size_t)Lv.ord.perm[i] + (size_t)c * Lv.n];
    HIPCHK(hipMemcpyAsync(dst, tmp.data(), tmp.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return SMG_OK;
}
static int get_block(smg_hierarchy* h, int lv, const double* src, int k, double* dst)
{
    const Level& Lv = h->lv[lv];
    std::vector<double> tmp((size_t)Lv.n * k);
    HIPCHK(hipMemcpyAsync(tmp.data(), src, tmp.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int i = 0; i < Lv.n; i++)
        for (int c = 0; c < k; c++) dst[(size_t)Lv.ord.perm[i] + (size_t)c * Lv.n] = tmp[(size_t)i * k + c];
    return coarse_stall_check(h);      // (the pieces' results leave through here: a stalled coarse solve must not pass for a result)
}

static int piece_prolog(smg_hierarchy* h, int lv, int k, const char* who, bool need_coarser)
{
    int rc = check_ready(h, who);
    if (rc) return rc;
    if (h->in_solve) return fail(SMG_ERR_INVALID, "%s: a split-phase solve is in progress", who);
    if (lv < 0 || lv >= h->n_levels || (need_coarser && lv >= h->n_levels - 1) || k < 1)
        return fail(SMG_ERR_INVALID, "%s: bad level %d or k %d", who, lv, k);
    DeviceScope dsc(h->device);
    return ensure_work(h, k);
}

extern "C" int smg_apply_A(smg_hierarchy* h, int lv, const double* u, int k, double* Au)
{
    int rc = piece_prolog(h, lv, k, "smg_apply_A", true);
    if (rc) return rc;
    DeviceScope dsc(h->device);
    Level& Lv = h->lv[lv];
    if ((rc = put_block(h, lv, u, k, Lv.u.p))) return rc;
    HIPCHK(Prec<double>::opA(h, Lv, false, SELL_AX, 0, -1, Lv.u.p, nullptr, Lv.r.p, k, nullptr));
    return get_block(h, lv, Lv.r.p, k, Au);
}

extern "C" int smg_restrict(smg_hierarchy* h, int lv, const double* x, int k, double* Rx)
{
    int rc = piece_prolog(h, lv, k, "smg_restrict", true);
    if (rc) return rc;
    DeviceScope dsc(h->device);
    Level &Lv = h->lv[lv], &Lc = h->lv[lv + 1];
    if ((rc = put_block(h, lv, x, k, Lv.r.p))) return rc;
    HIPCHK(launch_sell(SELL_AX, Lc.dPT.view, 0, Lc.dPT.view.n_slices, Lv.r.p, nullptr, Lc.b.p, k * h->bs, nullptr, nullptr, nullptr, h->stream));
    return get_block(h, lv + 1, Lc.b.p, k, Rx);
}

extern "C" int smg_prolong(smg_hierarchy* h, int lv, const double* x, int k, double* Px)
{
    int rc = piece_prolog(h, lv, k, "smg_prolong", true);
    if (rc) return rc;
    DeviceScope dsc(h->device);
    Level &Lv = h->lv[lv], &Lc = h->lv[lv + 1];
    if ((rc = put_block(h, lv + 1, x, k, Lc.u.p))) return rc;
    HIPCHK(launch_sell(SELL_AX, Lc.dP.view, 0, Lc.dP.view.n_slices, Lc.u.p, nullptr, Lv.r.p, k * h->bs, nullptr, nullptr, nullptr, h->stream));
    return get_block(h, lv, Lv.r.p, k, Px);
}

extern "C" int smg_relax(smg_hierarchy* h, int lv, const double* B, int k, int iters, double* u)
{
    int rc = piece_prolog(h, lv, k, "smg_relax", true);
    if (rc) return rc;
    DeviceScope dsc(h->device);
    Level& Lv = h->lv[lv];
    if ((rc = prepare_tiled(h, k, iters, iters))) return rc;
    if ((rc = put_block(h, lv, B, k, Lv.b.p))) return rc;
    if ((rc = put_block(h, lv, u, k, Lv.u.p))) return rc;
    if ((rc = enqueue_relax(h, lv, Lv.b.p, Lv.u.p, k, iters, nullptr))) return rc;
    return get_block(h, lv, Lv.u.p, k, u);
}

extern "C" int smg_coarse_solve(smg_hierarchy* h, const double* B, int k, double* u)
{
    const int lv = h ? h->n_levels - 1 : 0;
    int rc = piece_prolog(h, lv, k, "smg_coarse_solve", false);
    if (rc) return rc;
    DeviceScope dsc(h->device);
    Level& Lv = h->lv[lv];
    if ((rc = put_block(h, lv, B, k, Lv.b.p))) return rc;
    if ((rc = put_block(h, lv, u, k, Lv.u.p))) return rc;
    HIPCHK(Prec<double>::coarse(h, Lv, k, nullptr));
    return get_block(h, lv, Lv.u.p, k, u);
}

extern "C" int smg_vcycle(smg_hierarchy* h, const double* B, int pre, int post, int lv, double* u, int k)
{
    int rc = piece_prolog(h, lv, k, "smg_vcycle", false);
    if (rc) return rc;
    DeviceScope dsc(h->device);
    Level& Lv = h->lv[lv];
    if ((rc = prepare_tiled(h, k, pre, post))) return rc;
    if ((rc = put_block(h, lv, B, k, Lv.b.p))) return rc;
    if ((rc = put_block(h, lv, u, k, Lv.u.p))) return rc;
    if ((rc = enqueue_vcycle(h, lv, k, pre, post, nullptr))) return rc;
    return get_block(h, lv, Lv.u.p, k, u);
}

extern "C" int smg_residual_norm(smg_hierarchy* h, int lv, const double* B, const double* u, int k, double* norm)
{
    int rc = piece_prolog(h, lv, k, "smg_residual_norm", true);
    if (rc) return rc;
    DeviceScope dsc(h->device);
    Level& Lv = h->lv[lv];
    if ((rc = put_block(h, lv, B, k, Lv.b.p))) return rc;
    if ((rc = put_block(h, lv, u, k, Lv.u.p))) return rc;
    int nb = 0;
    Ctrl zero;
    std::memset(&zero, 0, sizeof(zero));
    zero.r_his = h->d_rhis.p; zero.his_cap = (int)h->d_rhis.n;
    HIPCHK(hipMemcpyAsync(h->d_ctrl.p, &zero, sizeof(Ctrl), hipMemcpyHostToDevice, h->stream));
    if (h->bs == 3) HIPCHK(launch_bsr3(SELL_RESID_SS, Lv.bA.view, 0, Lv.bA.view.n_slices, Lv.u.p, Lv.b.p, nullptr, k, nullptr, h->d_partials.p, &nb, h->stream));
    else HIPCHK(launch_sell(SELL_RESID_SS, Lv.dA.view, 0, Lv.dA.view.n_slices, Lv.u.p, Lv.b.p, nullptr, k, nullptr, h->d_partials.p, &nb, h->stream));
    HIPCHK(launch_ss_finalize(h->d_partials.p, nb, h->d_ctrl.p, h->stream));
    double ss = 0.0;
    HIPCHK(hipMemcpyAsync(&ss, &h->d_ctrl.p->sumsq, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    *norm = std::sqrt(ss);
    return SMG_OK;
}

// ------------------------------------------------------------------------------------------------ raw device interface
extern "C" int smg_raw_spmv(smg_hierarchy* h, int lv, int mode, const double* x, const double* b, double* y, int k)
{
    int rc = check_ready(h, "smg_raw_spmv");
    if (rc) return rc;
    DeviceScope dsc(h->device);
    if (lv < 0 || lv >= h->n_levels - 1 || k < 1 || (mode != SELL_AX && mode != SELL_RESID && mode != SELL_ADD))
        return fail(SMG_ERR_INVALID, "smg_raw_spmv: bad level/mode");
    Level& Lv = h->lv[lv];
    if (h->bs == 3) {
        if (mode == SELL_ADD) return fail(SMG_ERR_INVALID, "smg_raw_spmv: y += A x is not available on block (3-DOF) hierarchies");
        HIPCHK(launch_bsr3((SellMode)mode, Lv.bA.view, 0, Lv.bA.view.n_slices, x, b, y, k, nullptr, nullptr, nullptr, h->stream));
        return SMG_OK;
    }
    HIPCHK(launch_sell((SellMode)mode, Lv.dA.view, 0, Lv.dA.view.n_slices, x, b, y, k, nullptr, nullptr, nullptr, h->stream));
    return SMG_OK;
}

extern "C" int smg_raw_spmv_f32(smg_hierarchy* h, int lv, const float* x, float* y, int k)
{
    int rc = check_ready(h, "smg_raw_spmv_f32");
    if (rc) return rc;
    DeviceScope dsc(h->device);
    if (lv < 0 || lv >= h->n_levels - 1 || k < 1) return fail(SMG_ERR_INVALID, "smg_raw_spmv_f32: bad level");
    if ((rc = ensure_work(h, k))) return rc;
    if ((rc = ensure_fp32(h, k))) return rc;
    Level& Lv = h->lv[lv];
    HIPCHK(launch_sell_f32(SELL_AX, Lv.dA32, 0, Lv.dA32.n_slices, x, nullptr, y, k, nullptr, h->stream));
    return SMG_OK;
}

extern "C" int smg_raw_relax(smg_hierarchy* h, int lv, const double* b, double* u, int k, int iters)
{
    int rc = check_ready(h, "smg_raw_relax");
    if (rc) return rc;
    DeviceScope dsc(h->device);
    if (lv < 0 || lv >= h->n_levels - 1 || k < 1) return fail(SMG_ERR_INVALID, "smg_raw_relax: bad level");
    if ((rc = ensure_work(h, k))) return rc;   // second iterate / update vector / spectral bound of a Jacobi-type level
    if ((rc = prepare_tiled(h, k, iters, iters))) return rc;
    return enqueue_relax(h, lv, b, u, k, iters, nullptr);
}

// Algorithmic bytes of one outer iteration OF THE CYCLE THE HANDLE IS SET TO RUN (smoother selection of the
// last solve / smg_hierarchy_set_smoother): per smoothed level
//   relax(iters), Gauss-Seidel:      iters sweeps of  matA + 24 n k  [b, u read, u write]   (the reference also reads A_diag: +8n; the
//                                    HIP kernel takes the diagonal from the row, so it is not counted)
//   relax(iters), damped Jacobi:     iters sweeps of  matA + 24 n k
//   relax(iters), Chebyshev-Jacobi:  ONE polynomial of degree iters + 1 = iters + 1 passes of  matA + 40 n k  [b, u, d read; u, d write],
//                                    the first without the read of d
//   residual:             matA + 24 n k
//   restrict:             12 nnzPT + 4(nc+1) + 8 n k + 8 nc k
//   prolong-add:          12 nnzP + 4(n+1) + 8 nc k + 16 n k
//   + coarsest dense solve 8 nc^2 + 24 nc k, + outer residual matA_0 + 16 n0 k.
// matA = 12 nnz + 4(n+1); on a block hierarchy 76 bytes per 3 x 3 block + 4(n/3+1), and P (x) I_3 streams its vertex-level factor once.
extern "C" long smg_vcycle_bytes(const smg_hierarchy* h, int k, int pre, int post)
{
    if (!h || !h->precomputed) return -1;
    long tot = 0;
    const int L = h->n_levels;
    for (int lv = 0; lv < L - 1; lv++) {
        const Level &Lv = h->lv[lv], &Lc = h->lv[lv + 1];
        const long n = Lv.n, nc = Lc.n, nnz = Lv.A.nnz();
        const long nnzP = h->bs == 3 ? Lc.Pv.nnz() : Lc.P.nnz();
        const long matA = h->bs == 3 ? 76 * Lv.bA.blocks + 4 * (n / 3 + 1) : 12 * nnz + 4 * (n + 1);
        auto relax_bytes = [&](int iters) -> long {
            if (iters <= 0) return 0;
            if (level_kind(h, lv) == LV_CHEBY) return (long)(iters + 1) * (matA + 40 * n * k) - 8 * n * k;
            return (long)iters * (matA + 24 * n * k);
        };
        tot += relax_bytes(pre) + relax_bytes(post);
        tot += matA + 24 * n * k;
        tot += 12 * nnzP + 4 * (nc + 1) + 8 * n * k + 8 * nc * k;
        tot += 12 * nnzP + 4 * (n + 1) + 8 * nc * k + 16 * n * k;
    }
    const long nc = h->lv[L - 1].n;
    if (h->coarse_sparse) tot += (2 * 12 * h->chol.nnzL() + 3 * 8 * nc + 24 * nc) * k;    // both triangles of L, once per column
    else if (h->coarse_schur) tot += 8 * (h->schur.off_P + 2 * (h->schur.off_S - h->schur.off_W) + (long)h->schur.ns_pad * h->schur.ns_pad) + 24 * nc * k;   // blocks, the panels twice, S^-1
    else tot += 8 * nc * nc + 24 * nc * k;
    if (h->bs == 3) tot += 76 * h->lv[0].bA.blocks + 4L * (h->lv[0].n / 3 + 1) + 16L * h->lv[0].n * k;
    else tot += 12 * h->lv[0].A.nnz() + 4L * (h->lv[0].n + 1) + 16L * h->lv[0].n * k;
    return tot;
}
