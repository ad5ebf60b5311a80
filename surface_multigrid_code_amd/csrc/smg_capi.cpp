// smg_capi.cpp -- the C ABI declared in include/smg.h: errors, the hierarchy handle (container, setters, introspection), the profc
// mirror and the mesh numerics shims.  The three heavy parts live in their own translation units (smg_internal.hpp):
//   min_quad_with_fixed_mg_precompute  (reference src/min_quad_with_fixed_mg.cpp:3-51, :137-257)  -> smg_precompute.cpp
//   min_quad_with_fixed_mg_solve, mg_VCycle (reference src/min_quad_with_fixed_mg.cpp:80-135, :288-361, src/mg_VCycle.cpp:3-201) -> smg_cycle.cpp
//   mg_precompute / mg_precompute_block / .smgh files  (reference src/mg_precompute.cpp, src/mg_precompute_block.cpp) -> smg_hierarchy_io.cpp
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
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

// ------------------------------------------------------------------------------------------------ errors
static thread_local std::string g_err;
int smg::fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

extern "C" const char* smg_last_error(void) { return g_err.c_str(); }
extern "C" int smg_version(void) { return SMG_VERSION; }
extern "C" long long smg_device_bytes_live(void) { return (long long)smg::devbuf_live_bytes().load(); }
// What one handle holds in HBM, by purpose (memory budget reporting; bench.py's device_bytes, tools/mem_probe.py): text into buf, one "name bytes" line each.
extern "C" int smg_debug_device_bytes(const smg_hierarchy* h, char* buf, int cap)
{
    if (!h || !buf || cap < 1) return SMG_ERR_INVALID;
    std::string out;
    auto B = [](const auto& d) { return (long long)(d.n * sizeof(*d.p)); };
    auto sell = [&](const smg::SellBuf& S) { return B(S.slice_row) + B(S.slice_off) + B(S.slice_w) + B(S.col) + B(S.order) + B(S.val) + B(S.diag_slot) + B(S.long_row) + B(S.long_ptr) + B(S.long_col) + B(S.long_val) + B(S.long_valf); };
    long long tot = 0;
    auto line = [&](const std::string& nm, long long v) { if (v) { out += nm + " " + std::to_string(v) + "\n"; tot += v; } };
    for (int lv = 0; lv < h->n_levels; lv++) {
        const smg::Level& L = h->lv[lv];
        const std::string p = "level" + std::to_string(lv) + ".";
        line(p + "A_sell", sell(L.dA)); line(p + "AT_sell", sell(L.dAT)); line(p + "P_sell", sell(L.dP)); line(p + "PT_sell", sell(L.dPT));
        line(p + "A_block3", B(L.bA.slice_row) + B(L.bA.slice_off) + B(L.bA.slice_w) + B(L.bA.col) + B(L.bA.order) + B(L.bA.val) + B(L.bA.valf) + B(L.bAT.slice_row) + B(L.bAT.slice_off) + B(L.bAT.slice_w) + B(L.bAT.col) + B(L.bAT.order) + B(L.bAT.val) + B(L.bAT.valf) + B(L.mapB) + B(L.mapBT));
        line(p + "values_caller_order", B(L.d_Aval) + B(L.d_Tval));
        line(p + "refresh_maps", B(L.mapA) + B(L.mapAT));
        line(p + "galerkin_recipes", B(L.r1_ptr) + B(L.r1_idx) + B(L.r2_ptr) + B(L.r2_idx) + B(L.r1_coef) + B(L.r2_coef));
        long long t = 0;
        for (const auto& T : L.tiled) t += B(T.hdr) + B(T.ext_rows) + B(T.pcol) + B(T.prow) + B(T.map) + B(T.mapd) + B(T.pval) + B(T.pdiag);
        line(p + "tiled_plans", t);
        line(p + "bgs_plan", B(L.bgs.hdr) + B(L.bgs.xrow) + B(L.bgs.ugrow) + B(L.bgs.ulrow) + B(L.bgs.eidx) + B(L.bgs.map) + B(L.bgs.mapd) + B(L.bgs.eval) + B(L.bgs.udiag));
        line(p + "wgs_plan", B(L.wgs.hdr) + B(L.wgs.grow) + B(L.wgs.meta) + B(L.wgs.rim) + B(L.wgs.map) + B(L.wgs.mapd) + B(L.wgs.eoff) + B(L.wgs.eval) + B(L.wgs.diag));
        line(p + "fp32_images", B(L.a32) + B(L.at32) + B(L.p32) + B(L.pt32) + B(L.b32) + B(L.u32) + B(L.r32) + B(L.t32) + B(L.d32));
        line(p + "vectors", B(L.b) + B(L.u) + B(L.r) + B(L.t) + B(L.d));
    }
    line("coarse.dense_inverse", B(h->d_Ainv) + B(h->d_Ainv32));
    line("coarse.sym_partials", B(h->d_sympart));
    line("coarse.sparse_factor", B(h->c_perm) + B(h->c_rptr) + B(h->c_rcol) + B(h->c_cptr) + B(h->c_crow) + B(h->c_err) + B(h->c_rval) + B(h->c_cval) + B(h->c_diag) + B(h->c_work));
    line("coarse.schur", B(h->sch.irow) + B(h->sch.bsize) + B(h->sch.srow) + B(h->sch.sptr) + B(h->sch.sidx) + B(h->sch.aptr) + B(h->sch.ablk) + B(h->sch.acol) + B(h->sch.rptr) + B(h->sch.coff) + B(h->sch.pos) +
                         B(h->sch.pos2) + B(h->sch.ones) + B(h->sch.rdst) + B(h->sch.rdst2) + B(h->sch.rsrc) + B(h->sch.arena) + B(h->sch.g) + B(h->sch.xs) + B(h->sch.sym) + B(h->sch.gj) + B(h->sch.arena32) + B(h->sch.g32) + B(h->sch.xs32));
    line("maps_level0", B(h->d_map0) + B(h->d_perm0) + B(h->d_unknown) + B(h->d_known) + B(h->d_auk_ptr) + B(h->d_auk_col) + B(h->d_auk_val));
    line("reprecompute_bookkeeping", B(h->d_lhs_src) + B(h->d_auk_src) + B(h->d_diag_idx) + B(h->d_dense_pos) + B(h->d_Afull));
    line("early_upload", B(h->early0.ptr) + B(h->early0.col) + B(h->early0.val));
    line("solve_state", B(h->d_ctrl) + B(h->d_rhis) + B(h->d_partials) + B(h->d_lam) + B(h->d_stage_rhs) + B(h->d_stage_z) + B(h->d_stage_kv) + B(h->d_tmp_cm) + B(h->d_zsave));
    out += "total " + std::to_string(tot) + "\n";
    std::snprintf(buf, (size_t)cap, "%s", out.c_str());
    return SMG_OK;
}
extern "C" int smg_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
extern "C" void smg_solve_opts_default(smg_solve_opts* o)
{
    if (!o) return;
    o->tol = 1e-3;       // reference src/min_quad_with_fixed_mg.cpp:63, :270
    o->max_iter = 20;    // :77, :285
    o->pre = 2;          // :102, :324
    o->post = 2;         // :103, :325
    o->verbosity = 0;
    o->check_every = 0;   // adaptive polling (see smg_solve)
    o->use_graph = 1;
    o->precision = 0;
    o->smoother = SMG_SMOOTH_GS;   // the reference's relax()
    o->omega = 0.8;
    o->jacobi_max_rows = 100000;
    o->cheby_fraction = 0.1;
}

// ------------------------------------------------------------------------------------------------ device plumbing
int smg::ensure_device(smg_hierarchy* h)
{
    if (h->device >= 0) return SMG_OK;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(SMG_ERR_NO_DEVICE, "no HIP device: libsmg has no CPU fallback (the CPU oracle lives in oracle/)");
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    h->device = dev;
    if (!h->stream && !h->user_stream) {
        HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        h->own_stream = true;
    }
    HIPCHK(h->d_ctrl.alloc(1));
    HIPCHK(hipMemset(h->d_ctrl.p, 0, sizeof(Ctrl)));
    HIPCHK(h->d_rhis.alloc(1));
    return SMG_OK;
}

void smg::drop_graphs(smg_hierarchy* h)
{
    if (h->g_iter) (void)hipGraphExecDestroy(h->g_iter);
    if (h->g_iter_n) (void)hipGraphExecDestroy(h->g_iter_n);
    if (h->g_resid) (void)hipGraphExecDestroy(h->g_resid);
    if (h->g_cycle) (void)hipGraphExecDestroy(h->g_cycle);
    if (h->g_spec) (void)hipGraphExecDestroy(h->g_spec);
    if (h->g_rd) (void)hipGraphExecDestroy(h->g_rd);
    if (h->g_cyc) (void)hipGraphExecDestroy(h->g_cyc);
    h->g_iter = h->g_iter_n = h->g_resid = h->g_cycle = h->g_spec = h->g_rd = h->g_cyc = nullptr;
    h->g_key = smg::GraphKey();
}

hipError_t SellBuf::upload(const Sell& S)
{
    hipError_t e;
    if ((e = slice_row.upload(S.slice_row)) != hipSuccess) return e;
    if ((e = slice_off.upload(S.slice_off)) != hipSuccess) return e;
    if ((e = slice_w.upload(S.slice_w)) != hipSuccess) return e;
    const bool layout_only = S.col.empty() && S.padded() > 0;     // sell_layout(): the panels are filled on the device
    if (layout_only) {
        if ((e = col.alloc((size_t)S.padded())) != hipSuccess) return e;
        if ((e = val.alloc((size_t)S.padded())) != hipSuccess) return e;
    } else {
        if ((e = col.upload(S.col)) != hipSuccess) return e;
        if ((e = val.upload(S.val)) != hipSuccess) return e;
    }
    if ((e = order.upload(S.region_order)) != hipSuccess) return e;
    view.n_rows = S.n_rows; view.n_cols = S.n_cols; view.n_slices = S.n_slices; view.C = S.C;
    view.order = S.region_order.empty() ? nullptr : order.p;
    view.slice_row = slice_row.p; view.slice_off = slice_off.p; view.slice_w = slice_w.p; view.col = col.p; view.val = val.p;
    view.stride = S.stride; view.w_lo = S.w_lo;
    view.w_max = 0;
    for (int w : S.slice_w) view.w_max = std::max(view.w_max, w);
    if (env_int("SMG_DEBUG_SELL", 0)) {
        int hist[33] = {0};
        for (int w : S.slice_w) hist[std::min(w, 32)]++;
        std::fprintf(stderr, "sell %d x %d: %d slices, stride %d, w_lo %d, widths:", S.n_rows, S.n_cols, S.n_slices, S.stride, S.w_lo);
        for (int w = 0; w <= 32; w++) if (hist[w]) std::fprintf(stderr, " %d:%d", w, hist[w]);
        std::fprintf(stderr, "\n");
    }
    color_slice_ptr = S.color_slice_ptr;
    stored = S.nnz; padded = S.padded(); used = S.used();
    // where the diagonal of each row sits in the value array (restriction launches that produce the first launch of the coarse
    // level's first sweep themselves: the first colour of a Gauss-Seidel sweep / the whole first Jacobi sweep)
    n_first = 0; n_all = 0;
    if (S.n_rows == S.n_cols && S.color_slice_ptr.size() >= 2 && !layout_only) {
        const int s1 = S.color_slice_ptr.size() >= 3 ? S.color_slice_ptr[1] : S.n_slices;
        const int nf = S.slice_row[s1];
        std::vector<int> slot((size_t)S.n_rows, -1);
        int first_missing = S.n_rows;
        for (int sl = 0; sl < S.n_slices; sl++) {
            const int r0 = S.slice_row[sl], r1 = S.slice_row[sl + 1];
            for (int r = r0; r < r1; r++) {
                for (int j = 0; j < S.slice_w[sl]; j++) {
                    const size_t at = ((size_t)S.slice_off[sl] + j) * S.C + (r - r0);
                    if (S.col[at] == r) { slot[r] = (int)at; break; }
                }
                if (slot[r] < 0 && r < first_missing) first_missing = r;
            }
        }
        if (S.n_rows > 0 && first_missing >= nf) {
            if ((e = diag_slot.upload(slot)) != hipSuccess) return e;
            if (S.color_slice_ptr.size() >= 3) n_first = nf;
            if (first_missing == S.n_rows) n_all = S.n_rows;
        }
    }
    return hipSuccess;
}

hipError_t SellBuf::upload_long(const std::vector<int>& rows, const std::vector<int>& ptr, const std::vector<int>& col, const std::vector<double>& val)
{
    hipError_t e;
    view.long_n = 0; view.long_row = view.long_ptr = view.long_col = nullptr; view.long_val = nullptr; view.long_valf = nullptr;
    long_valf.release();
    if (rows.empty()) { long_row.release(); long_ptr.release(); long_col.release(); long_val.release(); return hipSuccess; }
    if ((e = long_row.upload(rows)) != hipSuccess) return e;
    if ((e = long_ptr.upload(ptr)) != hipSuccess) return e;
    if ((e = long_col.upload(col)) != hipSuccess) return e;
    if ((e = long_val.upload(val)) != hipSuccess) return e;
    view.long_n = (int)rows.size(); view.long_row = long_row.p; view.long_ptr = long_ptr.p; view.long_col = long_col.p; view.long_val = long_val.p;
    return hipSuccess;
}

hipError_t Bsr3Buf::upload(const Bsr3Sell& S)
{
    hipError_t e;
    if ((e = slice_row.upload(S.slice_row)) != hipSuccess) return e;
    if ((e = slice_off.upload(S.slice_off)) != hipSuccess) return e;
    if ((e = slice_w.upload(S.slice_w)) != hipSuccess) return e;
    const size_t slots = (size_t)64 * (size_t)(S.slice_off.empty() ? 0 : S.slice_off.back());
    const bool layout_only = S.col.empty() && slots > 0;       // bsr3_layout(): the panels are filled on the device (launch_bsr3_fill)
    if (layout_only) {
        if ((e = col.alloc(slots)) != hipSuccess) return e;
        if ((e = val.alloc(slots * 9)) != hipSuccess) return e;
    } else {
        if ((e = col.upload(S.col)) != hipSuccess) return e;
        if ((e = val.upload(S.val)) != hipSuccess) return e;
    }
    if ((e = order.upload(S.region_order)) != hipSuccess) return e;
    view.n_vert = S.n_vert; view.n_slices = S.n_slices; view.w_max = S.w_max;
    view.slice_row = slice_row.p; view.slice_off = slice_off.p; view.slice_w = slice_w.p; view.col = col.p; view.val = val.p;
    view.order = S.region_order.empty() ? nullptr : order.p;
    color_slice_ptr = S.color_slice_ptr;
    stored = S.nnz_scalar; blocks = S.n_blocks; padded = (long)(slots * 9);
    return hipSuccess;
}

// ------------------------------------------------------------------------------------------------ profc mirror
int smg::prof_scope_id(smg_hierarchy* h, const char* name)
{
    for (size_t i = 0; i < h->scopes.size(); i++) if (h->scopes[i].name == name) return (int)i;
    ProfScope s; s.name = name;
    h->scopes.push_back(s);
    return (int)h->scopes.size() - 1;
}
hipEvent_t smg::prof_event(smg_hierarchy* h)
{
    if (!h->ev_pool.empty()) { hipEvent_t e = h->ev_pool.back(); h->ev_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
void smg::prof_collect(smg_hierarchy* h)
{
    if (h->recs.empty()) return;
    (void)hipStreamSynchronize(h->stream);
    for (auto& r : h->recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) { h->scopes[r.scope].ms += ms; h->scopes[r.scope].count++; }
        h->ev_pool.push_back(r.e0); h->ev_pool.push_back(r.e1);
    }
    h->recs.clear();
}

// ------------------------------------------------------------------------------------------------ container
extern "C" smg_hierarchy* smg_hierarchy_create(int n_levels)
{
    if (n_levels < 1) { fail(SMG_ERR_INVALID, "n_levels must be >= 1"); return nullptr; }
    smg_hierarchy* h = new (std::nothrow) smg_hierarchy();
    if (!h) { fail(SMG_ERR_ALLOC, "out of memory"); return nullptr; }
    h->n_levels = n_levels;
    h->lv.resize(n_levels);
    h->coarse_dense_max = env_int("SMG_COARSE_DENSE_MAX", 16384);
    h->coarse_dense_max_user = std::getenv("SMG_COARSE_DENSE_MAX") != nullptr;
    h->bgs_min_rows = env_int("SMG_BGS_MIN_ROWS", -1);
    h->coarse_schur_when = std::min(2, std::max(0, env_int("SMG_COARSE_SCHUR", 2)));
    h->coarse_schur_min = env_int("SMG_COARSE_SCHUR_MIN", 2048);
    h->coarse_schur_big = env_int("SMG_COARSE_SCHUR_BIG", 6144);
    h->coarse_schur_max = env_int("SMG_COARSE_SCHUR_MAX", 65536);
    return h;
}

extern "C" void smg_hierarchy_destroy(smg_hierarchy* h)
{
    if (!h) return;
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    drop_graphs(h);
    for (auto& r : h->recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    for (auto e : h->ev_pool) (void)hipEventDestroy(e);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    for (hipStream_t a : h->aux) if (a) (void)hipStreamDestroy(a);
    delete h;
}

extern "C" int smg_hierarchy_levels(const smg_hierarchy* h) { return h ? h->n_levels : SMG_ERR_INVALID; }

extern "C" int smg_hierarchy_set_stream(smg_hierarchy* h, void* hip_stream)
{
    if (!h) return fail(SMG_ERR_INVALID, "null handle");
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    drop_graphs(h);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    h->stream = (hipStream_t)hip_stream;  // NULL = the legacy default stream
    h->own_stream = false;
    h->user_stream = true;
    return SMG_OK;
}

extern "C" int smg_hierarchy_set_smoother(smg_hierarchy* h, int smoother, double omega, int jacobi_max_rows)
{
    if (!h) return fail(SMG_ERR_INVALID, "null handle");
    if (smoother < SMG_SMOOTH_GS || smoother > SMG_SMOOTH_HYBRID_CHEBYSHEV)
        return fail(SMG_ERR_INVALID, "smoother must be one of SMG_SMOOTH_GS, _JACOBI, _HYBRID, _CHEBYSHEV, _HYBRID_CHEBYSHEV");
    if (h->in_solve) return fail(SMG_ERR_INVALID, "smg_hierarchy_set_smoother called during a split-phase solve");
    if (omega > 2.0 || omega != omega) return fail(SMG_ERR_INVALID, "omega must be in (0, 2]");
    h->smoother = smoother;
    if (omega > 0.0) h->omega = omega;
    if (jacobi_max_rows >= 0) h->jacobi_max_rows = jacobi_max_rows;
    return SMG_OK;
}

extern "C" int smg_hierarchy_set_block_mode(smg_hierarchy* h, int mode)
{
    if (!h) return fail(SMG_ERR_INVALID, "null handle");
    if (mode != -1 && mode != 0 && mode != 3) return fail(SMG_ERR_INVALID, "smg_hierarchy_set_block_mode: mode must be -1 (automatic), 0 (scalar) or 3");
    if (h->in_solve) return fail(SMG_ERR_INVALID, "smg_hierarchy_set_block_mode called during a split-phase solve");
    if (mode != h->block_mode) { h->block_mode = mode; h->precomputed = false; }   // the next smg_precompute is a full one
    return SMG_OK;
}
extern "C" int smg_hierarchy_block_size(const smg_hierarchy* h) { return h ? h->bs : SMG_ERR_INVALID; }
extern "C" int smg_level_block_stats(const smg_hierarchy* h, int lv, long* n_blocks, long* n_block_slots, int* n_vertex_colors)
{
    if (!h || lv < 0 || lv >= h->n_levels - 1) return fail(SMG_ERR_INVALID, "smg_level_block_stats: bad level");
    if (h->bs != 3 || !h->precomputed) return fail(SMG_ERR_INVALID, "smg_level_block_stats: not a precomputed block hierarchy");
    const Bsr3Buf& B = h->lv[lv].bA;
    if (n_blocks) *n_blocks = B.blocks;
    if (n_block_slots) *n_block_slots = B.padded / 9;
    if (n_vertex_colors) *n_vertex_colors = h->lv[lv].vord.n_colors();
    return SMG_OK;
}

extern "C" int smg_level_get_block_image(const smg_hierarchy* h, int lv, int* n_slices, int* n_panel_cols, int* slice_row, int* slice_off, int* slice_w,
                                         int* col, double* val)
{
    return guarded("smg_level_get_block_image", [&]() {
        if (!h || lv < 0 || lv >= h->n_levels - 1) return fail(SMG_ERR_INVALID, "smg_level_get_block_image: bad level");
        const Level& Lv = h->lv[lv];
        if (h->bs != 3 || Lv.A_int.nr != Lv.n || Lv.n == 0 || (int)Lv.vord.perm.size() * 3 != Lv.n)
            return fail(SMG_ERR_INVALID, "smg_level_get_block_image: the host half of smg_precompute has not run on a block hierarchy");
        const Bsr3Sell S = build_bsr3(Lv.A_int, &Lv.vord.color_ptr, false);
        if (n_slices) *n_slices = S.n_slices;
        if (n_panel_cols) *n_panel_cols = S.slice_off.back();
        if (slice_row) std::copy(S.slice_row.begin(), S.slice_row.end(), slice_row);
        if (slice_off) std::copy(S.slice_off.begin(), S.slice_off.end(), slice_off);
        if (slice_w) std::copy(S.slice_w.begin(), S.slice_w.end(), slice_w);
        if (col) std::copy(S.col.begin(), S.col.end(), col);
        if (val) std::copy(S.val.begin(), S.val.end(), val);
        return (int)SMG_OK;
    });
}

extern "C" int smg_hierarchy_set_coarse_dense_max(smg_hierarchy* h, int n_max)
{
    if (!h || n_max < 0) return fail(SMG_ERR_INVALID, "smg_hierarchy_set_coarse_dense_max: bad arguments");
    if (h->in_solve) return fail(SMG_ERR_INVALID, "smg_hierarchy_set_coarse_dense_max called during a split-phase solve");
    if (n_max != h->coarse_dense_max || !h->coarse_dense_max_user) { h->coarse_dense_max = n_max; h->coarse_dense_max_user = true; h->precomputed = false; }   // the next smg_precompute is a full one
    return SMG_OK;
}
extern "C" int smg_hierarchy_set_memory_lean(smg_hierarchy* h, int on)
{
    if (!h || (on != 0 && on != 1)) return fail(SMG_ERR_INVALID, "smg_hierarchy_set_memory_lean: bad arguments (on: 0 fixed panel pitch, 1 compact panels)");
    if (h->in_solve) return fail(SMG_ERR_INVALID, "smg_hierarchy_set_memory_lean called during a split-phase solve");
    if ((on != 0) != h->mem_lean) { h->mem_lean = on != 0; h->precomputed = false; }   // the next smg_precompute is a full one
    return SMG_OK;
}
extern "C" int smg_hierarchy_coarse_solver(const smg_hierarchy* h, long* factor_entries)
{
    if (!h) return SMG_ERR_INVALID;
    if (factor_entries && h->union_m > 0) {      // a union keeps one dense inverse per member: sum of pad_i^2 entries
        long tot = 0;
        for (int l : h->union_mlda) tot += (long)l * l;
        *factor_entries = tot;
    } else if (factor_entries) *factor_entries = h->coarse_sparse ? h->chol.nnzL() : h->coarse_schur ? (long)h->schur.off_C : (long)h->nc_pad * h->nc_pad;
    return h->coarse_sparse ? 1 : h->coarse_schur ? 2 : 0;
}
extern "C" int smg_hierarchy_set_coarse_schur(smg_hierarchy* h, int when, int n_min)
{
    if (!h || when < 0 || when > 2) return fail(SMG_ERR_INVALID, "smg_hierarchy_set_coarse_schur: bad arguments (when: 0 never, 1 always, 2 from the first value-only re-precompute on)");
    if (h->in_solve) return fail(SMG_ERR_INVALID, "smg_hierarchy_set_coarse_schur called during a split-phase solve");
    if (n_min < 0) n_min = h->coarse_schur_min;
    if (when != h->coarse_schur_when || n_min != h->coarse_schur_min) { h->coarse_schur_when = when; h->coarse_schur_min = n_min; h->precomputed = false; }   // the next smg_precompute is a full one
    return SMG_OK;
}

extern "C" int smg_hierarchy_set_chebyshev(smg_hierarchy* h, double cheby_fraction)
{
    if (!h) return fail(SMG_ERR_INVALID, "null handle");
    if (h->in_solve) return fail(SMG_ERR_INVALID, "smg_hierarchy_set_chebyshev called during a split-phase solve");
    if (cheby_fraction >= 1.0 || cheby_fraction != cheby_fraction) return fail(SMG_ERR_INVALID, "cheby_fraction must be in (0, 1)");
    if (cheby_fraction > 0.0) h->cheby_fraction = cheby_fraction;
    return SMG_OK;
}
extern "C" double smg_level_spectral_bound(const smg_hierarchy* h, int lv)
{
    if (!h || lv < 0 || lv >= h->n_levels) return 0.0;
    if (!h->lam_valid && h->precomputed && h->device >= 0) {
        DeviceScope dsc(h->device);
        if (spectral_bounds(const_cast<smg_hierarchy*>(h)) != SMG_OK) return 0.0;
    }
    return h->lv[lv].lam;
}

// CSR/CSC array sanity: monotone pointers, indices in range.  Returns an error string or nullptr.
const char* smg::check_compressed(int n_major, int n_minor, const int* ptr, const int* idx)
{
    if (ptr[0] != 0) return "pointer array must start at 0";
    // (row-parallel: a time step's re-precompute checks the 8 M pattern entries of a 1 M-vertex mesh before anything else)
    std::atomic<int> bad{0};
    parallel_for(n_major, 1 << 16, [&](long a, long b) { for (long i = a; i < b; i++) if (ptr[i + 1] < ptr[i]) { bad.store(1, std::memory_order_relaxed); return; } });
    if (bad.load()) return "pointer array is not monotone";
    const long nnz = ptr[n_major];
    parallel_for(nnz, 1 << 18, [&](long a, long b) { for (long p = a; p < b; p++) if (idx[p] < 0 || idx[p] >= n_minor) { bad.store(1, std::memory_order_relaxed); return; } });
    return bad.load() ? "index out of range" : nullptr;
}

int smg::set_prolong(smg_hierarchy* h, int lv, Csr&& P)
{
    Level& L = h->lv[lv];
    L.P_full = std::move(P);          // reference src/mg_precompute.cpp:76
    L.P = L.P_full;                   // :74
    L.PT = transpose(L.P);            // :75
    h->precomputed = false;
    h->p_version++;
    return SMG_OK;
}

static int smg_level_set_prolong_impl(smg_hierarchy* h, int lv, int n_fine, int n_coarse, const int* rowptr,
                                     const int* col, const double* val)
{
    if (!h || lv < 1 || lv >= h->n_levels || !rowptr || n_fine < 0 || n_coarse < 0)
        return fail(SMG_ERR_INVALID, "smg_level_set_prolong: bad arguments (lv=%d)", lv);
    if (const char* e = check_compressed(n_fine, n_coarse, rowptr, col)) return fail(SMG_ERR_INVALID, "smg_level_set_prolong: %s", e);
    return set_prolong(h, lv, csr_from_arrays(n_fine, n_coarse, rowptr, col, val));
}

extern "C" int smg_level_set_prolong(smg_hierarchy* h, int lv, int n_fine, int n_coarse, const int* rowptr,
                                     const int* col, const double* val)
{
    return guarded("smg_level_set_prolong", [&]() { return smg_level_set_prolong_impl(h, lv, n_fine, n_coarse, rowptr, col, val); });
}

static int smg_level_set_prolong_csc_impl(smg_hierarchy* h, int lv, int n_fine, int n_coarse, const int* colptr,
                                         const int* rowidx, const double* val)
{
    if (!h || lv < 1 || lv >= h->n_levels || !colptr || n_fine < 0 || n_coarse < 0)
        return fail(SMG_ERR_INVALID, "smg_level_set_prolong_csc: bad arguments (lv=%d)", lv);
    if (const char* e = check_compressed(n_coarse, n_fine, colptr, rowidx)) return fail(SMG_ERR_INVALID, "smg_level_set_prolong_csc: %s", e);
    return set_prolong(h, lv, csr_from_csc_arrays(n_fine, n_coarse, colptr, rowidx, val));
}

extern "C" int smg_level_set_prolong_csc(smg_hierarchy* h, int lv, int n_fine, int n_coarse, const int* colptr,
                                         const int* rowidx, const double* val)
{
    return guarded("smg_level_set_prolong_csc", [&]() { return smg_level_set_prolong_csc_impl(h, lv, n_fine, n_coarse, colptr, rowidx, val); });
}

static int smg_level_set_mesh_impl(smg_hierarchy* h, int lv, const double* V, int nV, const int* F, int nF)
{
    if (!h || lv < 0 || lv >= h->n_levels) return fail(SMG_ERR_INVALID, "smg_level_set_mesh: bad level");
    h->lv[lv].V.assign(V, V + (size_t)nV * 3);
    h->lv[lv].F.assign(F, F + (size_t)nF * 3);
    return SMG_OK;
}

extern "C" int smg_level_set_mesh(smg_hierarchy* h, int lv, const double* V, int nV, const int* F, int nF)
{
    return guarded("smg_level_set_mesh", [&]() { return smg_level_set_mesh_impl(h, lv, V, nV, F, nF); });
}

extern "C" int smg_level_get_mesh(const smg_hierarchy* h, int lv, int* nV, int* nF, double* V, int* F)
{
    if (!h || lv < 0 || lv >= h->n_levels) return fail(SMG_ERR_INVALID, "smg_level_get_mesh: bad level");
    const Level& L = h->lv[lv];
    if (nV) *nV = (int)(L.V.size() / 3);
    if (nF) *nF = (int)(L.F.size() / 3);
    if (V) std::copy(L.V.begin(), L.V.end(), V);
    if (F) std::copy(L.F.begin(), L.F.end(), F);
    return SMG_OK;
}

// ------------------------------------------------------------------------------------------------ introspection
static const Csr* pick_matrix(const smg_hierarchy* h, int lv, int which, int internal)
{
    const Level& Lv = h->lv[lv];
    switch (which) {
        case 0: return internal ? &Lv.A_int : &Lv.A;
        case 1: return lv >= 1 ? (internal ? &Lv.P_int : &Lv.P) : nullptr;
        case 2: return lv >= 1 ? (internal ? &Lv.PT_int : &Lv.PT) : nullptr;
        case 3: return (lv >= 1 && !internal) ? &Lv.P_full : nullptr;
        case 4: return (lv == 0 && !internal) ? &h->Auk : nullptr;
    }
    return nullptr;
}

extern "C" int smg_level_get_matrix(const smg_hierarchy* h, int lv, int which, int internal, int* n_rows, int* n_cols,
                                    int* nnz, int* rowptr, int* col, double* val)
{
    if (!h || lv < 0 || lv >= h->n_levels) return fail(SMG_ERR_INVALID, "smg_level_get_matrix: bad level");
    if (h->host_stale) { int rc = refresh_host_values(const_cast<smg_hierarchy*>(h)); if (rc) return rc; }
    if (which == 0 && internal) { int rc = ensure_A_int(const_cast<smg_hierarchy*>(h), lv); if (rc) return rc; }
    if ((which == 1 || which == 2) && internal) { int rc = ensure_P_int(const_cast<smg_hierarchy*>(h), lv); if (rc) return rc; }
    const Csr* M = pick_matrix(h, lv, which, internal);
    if (!M) return fail(SMG_ERR_INVALID, "smg_level_get_matrix: no such matrix");
    if (n_rows) *n_rows = M->nr;
    if (n_cols) *n_cols = M->nc;
    if (nnz) *nnz = (int)M->nnz();
    if (rowptr) { if (M->ptr.empty()) rowptr[0] = 0; else std::copy(M->ptr.begin(), M->ptr.end(), rowptr); }
    if (col) std::copy(M->col.begin(), M->col.end(), col);
    if (val) std::copy(M->val.begin(), M->val.end(), val);
    return SMG_OK;
}

extern "C" int smg_level_get_perm(const smg_hierarchy* h, int lv, int* perm)
{
    if (!h || lv < 0 || lv >= h->n_levels || !perm) return fail(SMG_ERR_INVALID, "smg_level_get_perm: bad arguments");
    std::copy(h->lv[lv].ord.perm.begin(), h->lv[lv].ord.perm.end(), perm);
    return SMG_OK;
}

extern "C" int smg_level_get_colors(const smg_hierarchy* h, int lv, int* n_colors, int* color_ptr)
{
    if (!h || lv < 0 || lv >= h->n_levels) return fail(SMG_ERR_INVALID, "smg_level_get_colors: bad level");
    const Ordering& o = h->lv[lv].ord;
    if (n_colors) *n_colors = o.n_colors();
    if (color_ptr) std::copy(o.color_ptr.begin(), o.color_ptr.end(), color_ptr);
    return SMG_OK;
}

extern "C" int smg_level_get_Adiag(const smg_hierarchy* h, int lv, double* diag)
{
    if (!h || lv < 0 || lv >= h->n_levels || !diag) return fail(SMG_ERR_INVALID, "smg_level_get_Adiag: bad arguments");
    if (h->host_stale) { int rc = refresh_host_values(const_cast<smg_hierarchy*>(h)); if (rc) return rc; }
    std::copy(h->lv[lv].A_diag.begin(), h->lv[lv].A_diag.end(), diag);
    return SMG_OK;
}

extern "C" int smg_get_unknown(const smg_hierarchy* h, int* n_unknown, int* unknown)
{
    if (!h) return fail(SMG_ERR_INVALID, "null handle");
    if (n_unknown) *n_unknown = h->has_known ? (int)h->unknown.size() : h->n_full;
    if (unknown) {
        if (h->has_known) std::copy(h->unknown.begin(), h->unknown.end(), unknown);
        else for (int i = 0; i < h->n_full; i++) unknown[i] = i;
    }
    return SMG_OK;
}

extern "C" int smg_level_sell_stats(const smg_hierarchy* h, int lv, int which, long* stored, long* padded, int* n_slices)
{
    if (!h || lv < 0 || lv >= h->n_levels) return fail(SMG_ERR_INVALID, "smg_level_sell_stats: bad level");
    const SellBuf* S = which == 0 ? &h->lv[lv].dA : which == 1 ? &h->lv[lv].dP : which == 2 ? &h->lv[lv].dPT : nullptr;
    if (!S) return fail(SMG_ERR_INVALID, "smg_level_sell_stats: which must be 0,1,2");
    if (which == 0 && h->bs == 3) {   // scalar entries stored / scalar slots of the 3 x 3 block image
        const Bsr3Buf& B = h->lv[lv].bA;
        if (stored) *stored = B.stored;
        if (padded) *padded = B.padded;
        if (n_slices) *n_slices = B.view.n_slices;
        return SMG_OK;
    }
    if (stored) *stored = S->stored;
    if (padded) *padded = S->used;   // slots read per pass (the allocation may be larger: fixed-stride panels)
    if (n_slices) *n_slices = S->view.n_slices;
    return SMG_OK;
}

extern "C" int smg_level_first_colour_rows(const smg_hierarchy* h, int lv)
{
    if (!h || lv < 0 || lv >= h->n_levels) return fail(SMG_ERR_INVALID, "smg_level_first_colour_rows: bad level");
    const Level& Lv = h->lv[lv];
    return (Lv.gs_on_transpose ? Lv.dAT : Lv.dA).n_first;
}

extern "C" long smg_level_spmv_bytes(const smg_hierarchy* h, int lv, int k)
{
    if (!h || lv < 0 || lv >= h->n_levels) return -1;
    const Csr& A = h->lv[lv].A;
    if (h->bs == 3 && lv < h->n_levels - 1) return 76L * h->lv[lv].bA.blocks + 4L * (A.nr / 3 + 1) + 16L * A.nr * k;
    return 12L * A.nnz() + 4L * (A.nr + 1) + 16L * A.nr * k;
}

// ------------------------------------------------------------------------------------------------ host-side self-checks
extern "C" int smg_debug_check_tiling_plan(smg_hierarchy* h, int lv, int sweeps, int tile_rows, int* n_tiles, int* max_ext_rows, double* redundancy,
                                           double* max_abs_diff)
{
    return guarded("smg_debug_check_tiling_plan", [&]() -> int {
        if (!h || lv < 0 || lv >= h->n_levels - 1 || sweeps < 1 || tile_rows < 8) return fail(SMG_ERR_INVALID, "smg_debug_check_tiling_plan: bad arguments");
        int rc = ensure_A_int(h, lv);
        if (rc) return rc;
        Level& Lv = h->lv[lv];
        if (Lv.A_int.nr != Lv.n || Lv.n == 0 || h->bs != 1) return fail(SMG_ERR_INVALID, "smg_debug_check_tiling_plan: the host half of smg_precompute has not run (scalar hierarchies only)");
        const Csr& G = Lv.A_int;
        const int n = G.nr;
        const TiledGs P = build_tiled_gs(G, Lv.ord.color_ptr, sweeps, tile_rows, 1 << 20, 1 << 20);
        if (n_tiles) *n_tiles = P.n_tiles;
        if (max_ext_rows) *max_ext_rows = P.max_ext;
        if (redundancy) *redundancy = P.n_tiles ? (double)P.updates / ((double)sweeps * n) : 0.0;
        if (max_abs_diff) *max_abs_diff = 0.0;
        if (P.empty()) return SMG_OK;
        std::vector<double> x((size_t)n), b((size_t)n), ref, y((size_t)n, 0.0), xs;
        for (int i = 0; i < n; i++) { x[(size_t)i] = std::sin(0.37 * i) + 0.25 * std::cos(1.3 * i); b[(size_t)i] = std::cos(0.11 * i) - 0.5 * std::sin(2.1 * i); }
        // reference: the colour-by-colour sweeps in place (what one launch per colour computes)
        ref = x;
        const std::vector<int>& cp = Lv.ord.color_ptr;
        for (int s = 0; s < sweeps; s++)
            for (size_t c = 0; c + 1 < cp.size(); c++)
                for (int i = cp[c]; i < cp[c + 1]; i++) {
                    double acc = 0.0, diag = 1.0;
                    for (int p = G.ptr[(size_t)i]; p < G.ptr[(size_t)i + 1]; p++) {
                        if (G.col[(size_t)p] == i) diag = G.val[(size_t)p];
                        else acc += G.val[(size_t)p] * ref[(size_t)G.col[(size_t)p]];
                    }
                    ref[(size_t)i] = (b[(size_t)i] - acc) / diag;
                }
        // the plan, executed like k_tiled_gs: x -> y
        const int nc = P.nc, PP = P.P;
        for (int t = 0; t < P.n_tiles; t++) {
            const int* H = P.hdr.data() + (size_t)t * TILED_HDR;
            const int ext_off = H[0], n_ext = H[1], w = H[2];
            xs.assign((size_t)n_ext, 0.0);
            for (int i = 0; i < n_ext; i++) xs[(size_t)i] = x[(size_t)P.ext_rows[(size_t)ext_off + i]];
            for (int p = 1; p <= PP; p++) {
                const int* C = H + 4 + ((p - 1) % nc) * TILED_CSTRIDE;
                const int pan = C[0], m = C[1], ro = C[2], lbase = C[3], cnt = C[4 + (PP - p)];
                for (int i = 0; i < cnt; i++) {
                    double acc = 0.0;      // exactly the kernel's loop: every slot, the diagonal's and the padding's hold +0.0 at the row's own index
                    for (int j = 0; j < w; j++) {
                        const int cl = P.pcol[(size_t)pan + (size_t)j * m + i];
                        if (cl < 0 || cl >= n_ext) return fail(SMG_ERR_INVALID, "tiling plan: tile %d holds a column outside its image", t);
                        acc += P.pval[(size_t)pan + (size_t)j * m + i] * xs[(size_t)cl];
                    }
                    xs[(size_t)lbase + i] = (b[(size_t)P.prow[(size_t)ro + i]] - acc) / P.pdiag[(size_t)ro + i];
                }
            }
            for (int c = 0; c < nc; c++) {
                const int* C = H + 4 + c * TILED_CSTRIDE;
                for (int i = 0; i < C[4]; i++) y[(size_t)P.prow[(size_t)C[2] + i]] = xs[(size_t)C[3] + i];
            }
        }
        double d = 0.0;
        for (int i = 0; i < n; i++) d = std::max(d, std::fabs(y[(size_t)i] - ref[(size_t)i]));
        if (max_abs_diff) *max_abs_diff = d;
        return SMG_OK;
    });
}

// The wave Gauss-Seidel plan of level lv (smg_wgs.hpp) built on the host and EXECUTED on the host the way k_wgs executes it (wgs_sweep_host: per piece an
// image of its rows and its rim, phases in place, packed byte offsets) against the plain lexicographic sweep in the wgs order (the reference's relax(),
// src/mg_VCycle.cpp:146-160, on that numbering): *max_abs_diff must be 0.  Checks the plan's invariants on the way.  Needs no GPU once the host half of
// smg_precompute has run.  *n_pieces = 0: the level does not qualify.
extern "C" int smg_debug_check_wave_gs_plan(smg_hierarchy* h, int lv, int piece_rows, int pieces_mode, int* n_pieces, int* n_colors, double* stats, double* max_abs_diff)
{
    return guarded("smg_debug_check_wave_gs_plan", [&]() -> int {
        if (!h || lv < 0 || lv >= h->n_levels - 1 || piece_rows < 8) return fail(SMG_ERR_INVALID, "smg_debug_check_wave_gs_plan: bad arguments");
        int rc = ensure_A_int(h, lv);
        if (rc) return rc;
        Level& Lv = h->lv[lv];
        if (Lv.A_int.nr != Lv.n || Lv.n == 0 || h->bs != 1) return fail(SMG_ERR_INVALID, "smg_debug_check_wave_gs_plan: the host half of smg_precompute has not run (scalar hierarchies only)");
        const Csr& G = Lv.A_int;
        const int n = G.nr;
        const WgsPlan P = build_wgs(G, std::min(piece_rows, (int)WGS_ROWS), pieces_mode);
        if (n_pieces) *n_pieces = P.n_pieces;
        if (n_colors) *n_colors = P.n_colors;
        if (stats) { stats[0] = P.rim_ratio; stats[1] = P.phases_mean; stats[2] = (double)P.phases_max; }
        if (max_abs_diff) *max_abs_diff = 0.0;
        if (P.empty()) return SMG_OK;
        // invariants: every row in exactly one piece of <= 64 rows; pieces of one colour share no entry
        std::vector<int> pc_of((size_t)n, -1), col_of_pc((size_t)P.n_pieces, -1);
        for (int c = 0; c < P.n_colors; c++) for (int q = P.color_ptr[(size_t)c]; q < P.color_ptr[(size_t)c + 1]; q++) col_of_pc[(size_t)q] = c;
        for (int q = 0; q < P.n_pieces; q++) {
            if (P.piece_ptr[(size_t)q + 1] - P.piece_ptr[(size_t)q] > WGS_ROWS) return fail(SMG_ERR_INVALID, "wave plan: piece %d has more than 64 rows", q);
            for (int t = P.piece_ptr[(size_t)q]; t < P.piece_ptr[(size_t)q + 1]; t++) {
                const int i = P.rows[(size_t)t];
                if (i < 0 || i >= n || pc_of[(size_t)i] >= 0) return fail(SMG_ERR_INVALID, "wave plan: row %d is not in exactly one piece", i);
                pc_of[(size_t)i] = q;
            }
        }
        for (int i = 0; i < n; i++) {
            if (pc_of[(size_t)i] < 0) return fail(SMG_ERR_INVALID, "wave plan: row %d is in no piece", i);
            for (int p = G.ptr[(size_t)i]; p < G.ptr[(size_t)i + 1]; p++) {
                const int j = G.col[(size_t)p];
                if (pc_of[(size_t)j] != pc_of[(size_t)i] && col_of_pc[(size_t)pc_of[(size_t)j]] == col_of_pc[(size_t)pc_of[(size_t)i]])
                    return fail(SMG_ERR_INVALID, "wave plan: pieces %d and %d share an entry and a colour", pc_of[(size_t)i], pc_of[(size_t)j]);
            }
        }
        std::vector<double> x((size_t)n), b((size_t)n), ref, y;
        for (int i = 0; i < n; i++) { x[(size_t)i] = std::sin(0.37 * i) + 0.25 * std::cos(1.3 * i); b[(size_t)i] = std::cos(0.11 * i) - 0.5 * std::sin(2.1 * i); }
        // reference: rows one after the other in the wgs order, products in ascending column OF THAT ORDER
        std::vector<int> pos((size_t)n);
        for (int t = 0; t < n; t++) pos[(size_t)P.rows[(size_t)t]] = t;
        ref = x;
        std::vector<std::pair<int, int>> ent;
        for (int t = 0; t < n; t++) {
            const int i = P.rows[(size_t)t];
            ent.clear();
            double diag = 1.0;
            for (int p = G.ptr[(size_t)i]; p < G.ptr[(size_t)i + 1]; p++) {
                if (G.col[(size_t)p] == i) diag = G.val[(size_t)p]; else ent.emplace_back(pos[(size_t)G.col[(size_t)p]], p);
            }
            std::sort(ent.begin(), ent.end());
            double acc = 0.0;
            for (const auto& e : ent) acc += G.val[(size_t)e.second] * ref[(size_t)G.col[(size_t)e.second]];
            ref[(size_t)i] = (b[(size_t)i] - acc) / diag;
        }
        y = x;
        wgs_sweep_host(P, b.data(), y.data());
        double d = 0.0;
        for (int i = 0; i < n; i++) d = std::max(d, std::fabs(y[(size_t)i] - ref[(size_t)i]));
        if (max_abs_diff) *max_abs_diff = d;
        return SMG_OK;
    });
}

// The block Gauss-Seidel plan of level lv (smg_bgs.hpp) built on the host and EXECUTED on the host the way k_bgs executes it -- per block an image
// of its rows and its rim, units of <= 16 rows updated in place from local indices -- against the plain lexicographic sweep in the bgs order
// (the reference's relax(), src/mg_VCycle.cpp:146-160, on that numbering): *max_abs_diff must be 0.  Also checks the plan's invariants (every row in
// exactly one block, blocks of one colour share no entry, local indices inside the image).  Works without a GPU once the host half of
// smg_precompute has run.  Returns SMG_OK with *n_blocks = 0 when the level does not qualify.
extern "C" int smg_debug_check_block_gs_plan(smg_hierarchy* h, int lv, int block_rows, int* n_blocks, int* n_colors, double* rim, double* fill, double* max_abs_diff)
{
    return guarded("smg_debug_check_block_gs_plan", [&]() -> int {
        if (!h || lv < 0 || lv >= h->n_levels - 1 || block_rows < 8) return fail(SMG_ERR_INVALID, "smg_debug_check_block_gs_plan: bad arguments");
        int rc = ensure_A_int(h, lv);
        if (rc) return rc;
        Level& Lv = h->lv[lv];
        if (Lv.A_int.nr != Lv.n || Lv.n == 0 || h->bs != 1) return fail(SMG_ERR_INVALID, "smg_debug_check_block_gs_plan: the host half of smg_precompute has not run (scalar hierarchies only)");
        const Csr& G = Lv.A_int;
        const int n = G.nr;
        const BgsPlan P = build_bgs(G, Lv.ord.color_ptr, std::min(block_rows, (int)BGS_ROWS));
        if (n_blocks) *n_blocks = P.n_blocks;
        if (n_colors) *n_colors = P.n_colors;
        if (rim) *rim = P.rim;
        if (fill) *fill = P.fill;
        if (max_abs_diff) *max_abs_diff = 0.0;
        if (P.empty()) return SMG_OK;
        // invariants
        std::vector<int> blk_of((size_t)n, -1), col_of_blk((size_t)P.n_blocks, -1);
        for (int c = 0; c < P.n_colors; c++) for (int q = P.color_ptr[(size_t)c]; q < P.color_ptr[(size_t)c + 1]; q++) col_of_blk[(size_t)q] = c;
        for (int q = 0; q < P.n_blocks; q++)
            for (int t = P.blk_ptr[(size_t)q]; t < P.blk_ptr[(size_t)q + 1]; t++) {
                const int i = P.rows[(size_t)t];
                if (i < 0 || i >= n || blk_of[(size_t)i] >= 0) return fail(SMG_ERR_INVALID, "block plan: row %d is not in exactly one block", i);
                blk_of[(size_t)i] = q;
            }
        for (int i = 0; i < n; i++) {
            if (blk_of[(size_t)i] < 0) return fail(SMG_ERR_INVALID, "block plan: row %d is in no block", i);
            for (int p = G.ptr[(size_t)i]; p < G.ptr[(size_t)i + 1]; p++) {
                const int j = G.col[(size_t)p];
                if (blk_of[(size_t)j] != blk_of[(size_t)i] && col_of_blk[(size_t)blk_of[(size_t)j]] == col_of_blk[(size_t)blk_of[(size_t)i]])
                    return fail(SMG_ERR_INVALID, "block plan: blocks %d and %d share an entry and a colour", blk_of[(size_t)i], blk_of[(size_t)j]);
            }
        }
        std::vector<double> x((size_t)n), b((size_t)n), ref, y;
        for (int i = 0; i < n; i++) { x[(size_t)i] = std::sin(0.37 * i) + 0.25 * std::cos(1.3 * i); b[(size_t)i] = std::cos(0.11 * i) - 0.5 * std::sin(2.1 * i); }
        // reference: rows one after the other in the bgs order, products in ascending column OF THAT ORDER
        std::vector<int> pos((size_t)n);
        for (int t = 0; t < n; t++) pos[(size_t)P.rows[(size_t)t]] = t;
        ref = x;
        std::vector<std::pair<int, int>> ent;
        for (int t = 0; t < n; t++) {
            const int i = P.rows[(size_t)t];
            ent.clear();
            double diag = 1.0;
            for (int p = G.ptr[(size_t)i]; p < G.ptr[(size_t)i + 1]; p++) {
                if (G.col[(size_t)p] == i) diag = G.val[(size_t)p]; else ent.emplace_back(pos[(size_t)G.col[(size_t)p]], p);
            }
            std::sort(ent.begin(), ent.end());
            double acc = 0.0;
            for (const auto& e : ent) acc += G.val[(size_t)e.second] * ref[(size_t)G.col[(size_t)e.second]];
            ref[(size_t)i] = (b[(size_t)i] - acc) / diag;
        }
        // the plan, executed like k_bgs (one column): block colour by block colour, an image per block, units in place
        y = x;
        std::vector<double> xs((size_t)P.xrows);
        for (int q = 0; q < P.n_blocks; q++) {
            const int* H = P.hdr.data() + (size_t)q * BGS_HDR;
            const int unit0 = H[0], nu = H[1], S = H[2] * BGS_BATCH, ent0 = H[3];
            for (int l = 0; l < P.xrows; l++) xs[(size_t)l] = y[(size_t)P.xrow[(size_t)q * P.xrows + l]];
            for (int un = 0; un < nu; un++) {
                double out[BGS_UROWS];
                for (int r = 0; r < BGS_UROWS; r++) {
                    const size_t w = ((size_t)unit0 + un) * BGS_UROWS + r, e = (size_t)ent0 + ((size_t)un * BGS_UROWS + r) * S;
                    double acc = 0.0;
                    for (int t = 0; t < S; t++) {
                        const int l = P.eidx[e + t];
                        if (l < 0 || l >= P.xrows) return fail(SMG_ERR_INVALID, "block plan: local index %d outside the image of %d rows", l, P.xrows);
                        acc += P.eval[e + t] * xs[(size_t)l];
                    }
                    out[r] = (b[(size_t)P.ugrow[w]] - acc) / P.udiag[w];
                }
                for (int r = 0; r < BGS_UROWS; r++) {
                    const size_t w = ((size_t)unit0 + un) * BGS_UROWS + r;
                    xs[(size_t)P.ulrow[w]] = out[r];
                    y[(size_t)P.ugrow[w]] = out[r];
                }
            }
        }
        double d = 0.0;
        for (int i = 0; i < n; i++) d = std::max(d, std::fabs(y[(size_t)i] - ref[(size_t)i]));
        if (max_abs_diff) *max_abs_diff = d;
        return SMG_OK;
    });
}

extern "C" int smg_debug_raise_coarse_stall(smg_hierarchy* h)
{
    if (!h || !h->coarse_sparse || !h->c_err.p) return fail(SMG_ERR_INVALID, "smg_debug_raise_coarse_stall: no sparse coarse factorisation on this handle");
    const int one = 1;
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(h->c_err.p, &one, sizeof(int), hipMemcpyHostToDevice));
    return SMG_OK;
}

extern "C" int smg_debug_check_sparse_cholesky(int n, const int* rowptr, const int* col, const double* val, long* factor_entries, int* dependency_depth,
                                               double* rel_residual)
{
    return guarded("smg_debug_check_sparse_cholesky", [&]() -> int {
        if (n <= 0 || !rowptr || !col || !val) return fail(SMG_ERR_INVALID, "smg_debug_check_sparse_cholesky: bad arguments");
        if (const char* e = check_compressed(n, n, rowptr, col)) return fail(SMG_ERR_INVALID, "smg_debug_check_sparse_cholesky: %s", e);
        const Csr A = csr_from_arrays(n, n, rowptr, col, val);
        SparseChol F;
        if (!sparse_cholesky(A, F)) return fail(SMG_ERR_INVALID, "smg_debug_check_sparse_cholesky: not positive definite");
        std::vector<double> b((size_t)n), z((size_t)n), x((size_t)n);
        for (int i = 0; i < n; i++) b[(size_t)i] = std::sin(0.01 * i) + 1.0;
        for (int i = 0; i < n; i++) { double s = b[(size_t)F.perm[(size_t)i]]; for (int p = F.rptr[(size_t)i]; p < F.rptr[(size_t)i + 1]; p++) s -= F.rval[(size_t)p] * z[(size_t)F.rcol[(size_t)p]]; z[(size_t)i] = s / F.diag[(size_t)i]; }
        for (int i = n - 1; i >= 0; i--) { double s = z[(size_t)i]; for (int p = F.cptr[(size_t)i]; p < F.cptr[(size_t)i + 1]; p++) s -= F.cval[(size_t)p] * z[(size_t)F.crow[(size_t)p]]; z[(size_t)i] = s / F.diag[(size_t)i]; }
        for (int i = 0; i < n; i++) x[(size_t)F.perm[(size_t)i]] = z[(size_t)i];
        double rn = 0.0, bn = 0.0;
        for (int i = 0; i < n; i++) { double s = b[(size_t)i]; for (int p = A.ptr[(size_t)i]; p < A.ptr[(size_t)i + 1]; p++) s -= A.val[(size_t)p] * x[(size_t)A.col[(size_t)p]]; rn += s * s; bn += b[(size_t)i] * b[(size_t)i]; }
        std::vector<int> depth((size_t)n, 0);
        int dmax = 0;
        for (int i = 0; i < n; i++) { int d = 0; for (int p = F.rptr[(size_t)i]; p < F.rptr[(size_t)i + 1]; p++) d = std::max(d, depth[(size_t)F.rcol[(size_t)p]] + 1); depth[(size_t)i] = d; dmax = std::max(dmax, d); }
        if (factor_entries) *factor_entries = F.nnzL();
        if (dependency_depth) *dependency_depth = dmax;
        if (rel_residual) *rel_residual = std::sqrt(rn / bn);
        return SMG_OK;
    });
}

// ------------------------------------------------------------------------------------------------ profc mirror (API)
extern "C" int smg_prof_enable(smg_hierarchy* h, int on)
{
    if (!h) return fail(SMG_ERR_INVALID, "null handle");
    h->prof_on = on != 0;
    return SMG_OK;
}
extern "C" int smg_prof_reset(smg_hierarchy* h)
{
    if (!h) return fail(SMG_ERR_INVALID, "null handle");
    prof_collect(h);
    for (auto& s : h->scopes) { s.count = 0; s.ms = 0.0; }
    return SMG_OK;
}
extern "C" int smg_prof_count(smg_hierarchy* h)
{
    if (!h) return SMG_ERR_INVALID;
    prof_collect(h);
    return (int)h->scopes.size();
}
extern "C" int smg_prof_get(smg_hierarchy* h, int idx, char* name, int name_cap, long* count, double* total_ms)
{
    if (!h || idx < 0 || idx >= (int)h->scopes.size()) return fail(SMG_ERR_INVALID, "smg_prof_get: bad index");
    prof_collect(h);
    const ProfScope& s = h->scopes[idx];
    if (name && name_cap > 0) { std::strncpy(name, s.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
    if (count) *count = s.count;
    if (total_ms) *total_ms = s.ms;
    return SMG_OK;
}

// ------------------------------------------------------------------------------------------------ mesh numerics (host)
Mesh smg::wrap_mesh(const double* V, int nV, const int* F, int nF)
{
    Mesh m;
    if (V) m.V.assign(V, V + (size_t)nV * 3);
    if (F) m.F.assign(F, F + (size_t)nF * 3);
    return m;
}

static int smg_mesh_read_impl(const char* path, double** V, int* nV, int** F, int* nF)
{
    if (!path || !V || !nV || !F || !nF) return fail(SMG_ERR_INVALID, "smg_mesh_read: bad arguments");
    Mesh m;
    if (!read_mesh(path, m)) return fail(SMG_ERR_IO, "cannot read mesh '%s'", path);
    *nV = m.nV(); *nF = m.nF();
    *V = (double*)std::malloc(m.V.size() * sizeof(double) + 8);
    *F = (int*)std::malloc(m.F.size() * sizeof(int) + 8);
    if (!*V || !*F) return fail(SMG_ERR_ALLOC, "out of memory");
    std::memcpy(*V, m.V.data(), m.V.size() * sizeof(double));
    std::memcpy(*F, m.F.data(), m.F.size() * sizeof(int));
    return SMG_OK;
}

extern "C" int smg_mesh_read(const char* path, double** V, int* nV, int** F, int* nF)
{
    return guarded("smg_mesh_read", [&]() { return smg_mesh_read_impl(path, V, nV, F, nF); });
}
extern "C" void smg_free(void* p) { std::free(p); }

extern "C" int smg_mesh_normalize_unit_area(double* V, int nV, const int* F, int nF)
{
    if (!V || !F) return fail(SMG_ERR_INVALID, "smg_mesh_normalize_unit_area: bad arguments");
    Mesh m = wrap_mesh(V, nV, F, nF);
    normalize_unit_area(m);
    std::copy(m.V.begin(), m.V.end(), V);
    return SMG_OK;
}

static int smg_mesh_cotmatrix_impl(const double* V, int nV, const int* F, int nF, int* nnz, int* rowptr, int* col, double* val)
{
    if (!V || !F) return fail(SMG_ERR_INVALID, "smg_mesh_cotmatrix: bad arguments");
    static thread_local Csr cache;
    static thread_local const double* cacheV = nullptr;
    // two-call protocol (size query, then fill): keep the result of the query for the fill
    if (!(rowptr && cacheV == V && cache.nr == nV)) { cache = cotmatrix(wrap_mesh(V, nV, F, nF)); cacheV = V; }
    if (nnz) *nnz = (int)cache.nnz();
    if (rowptr) {
        std::copy(cache.ptr.begin(), cache.ptr.end(), rowptr);
        if (col) std::copy(cache.col.begin(), cache.col.end(), col);
        if (val) std::copy(cache.val.begin(), cache.val.end(), val);
        cache = Csr(); cacheV = nullptr;
    }
    return SMG_OK;
}

extern "C" int smg_mesh_cotmatrix(const double* V, int nV, const int* F, int nF, int* nnz, int* rowptr, int* col, double* val)
{
    return guarded("smg_mesh_cotmatrix", [&]() { return smg_mesh_cotmatrix_impl(V, nV, F, nF, nnz, rowptr, col, val); });
}

extern "C" int smg_mesh_massmatrix(const double* V, int nV, const int* F, int nF, int voronoi, double* diag)
{
    if (!V || !F || !diag) return fail(SMG_ERR_INVALID, "smg_mesh_massmatrix: bad arguments");
    std::vector<double> M = massmatrix_diag(wrap_mesh(V, nV, F, nF), voronoi ? MASS_VORONOI : MASS_BARYCENTRIC);
    std::copy(M.begin(), M.end(), diag);
    return SMG_OK;
}

extern "C" int smg_mesh_boundary_loop(const int* F, int nF, int nV, int* loop, int* n_loop)
{
    if (!F || !loop || !n_loop) return fail(SMG_ERR_INVALID, "smg_mesh_boundary_loop: bad arguments");
    (void)nV;
    std::vector<int> b = boundary_loop(wrap_mesh(nullptr, 0, F, nF));
    *n_loop = (int)b.size();
    std::copy(b.begin(), b.end(), loop);
    return SMG_OK;
}

static int smg_mesh_midpoint_upsample_impl(int nV, const int* F, int nF, int* nE, int* S_rowptr, int* S_col, double* S_val, int* NF)
{
    if (!F) return fail(SMG_ERR_INVALID, "smg_mesh_midpoint_upsample: bad arguments");
    std::vector<int> Fv(F, F + (size_t)nF * 3), NFv;
    Csr S;
    midpoint_upsample(nV, Fv, S, NFv);
    if (nE) *nE = S.nr - nV;
    if (S_rowptr) std::copy(S.ptr.begin(), S.ptr.end(), S_rowptr);
    if (S_col) std::copy(S.col.begin(), S.col.end(), S_col);
    if (S_val) std::copy(S.val.begin(), S.val.end(), S_val);
    if (NF) std::copy(NFv.begin(), NFv.end(), NF);
    return SMG_OK;
}

extern "C" int smg_mesh_midpoint_upsample(int nV, const int* F, int nF, int* nE, int* S_rowptr, int* S_col, double* S_val, int* NF)
{
    return guarded("smg_mesh_midpoint_upsample", [&]() { return smg_mesh_midpoint_upsample_impl(nV, F, nF, nE, S_rowptr, S_col, S_val, NF); });
}

extern "C" int smg_mesh_torus(int nu, int nv, double R, double r, double* V, int* F)
{
    if (nu < 3 || nv < 3 || !V || !F) return fail(SMG_ERR_INVALID, "smg_mesh_torus: bad arguments");
    Mesh m = make_torus(nu, nv, R, r);
    std::copy(m.V.begin(), m.V.end(), V);
    std::copy(m.F.begin(), m.F.end(), F);
    return SMG_OK;
}

// Test hook (tests/test_host_logic.py): the plan of the Schur-complement coarse solver (smg_schur.hpp) built for the matrix (ptr, col, val) and EXECUTED ON THE
// HOST exactly as the kernels of smg_schur_device.hip use it -- scatter lists, unit padding, per-block inverse and panels, the sum lists of the Schur
// complement, the three solve steps -- so every index the device reads is checked without a GPU.  x = A^-1 b (k = 1); *n_blocks = 0: no plan for this matrix.
extern "C" int smg_debug_schur_solve_host(int n, const int* ptr, const int* col, const double* val, const double* b, double* x, int* n_blocks, int* n_sep)
{
    return guarded("smg_debug_schur_solve_host", [&]() -> int {
        if (n <= 0 || !ptr || !col || !val || !b || !x) return fail(SMG_ERR_INVALID, "smg_debug_schur_solve_host: bad arguments");
        const Csr A = csr_from_arrays(n, n, ptr, col, val);
        const SchurPlan P = build_schur(A);
        if (n_blocks) *n_blocks = P.nb;
        if (n_sep) *n_sep = P.ns;
        if (P.empty()) return SMG_OK;
        // in-place inverse of an SPD matrix by Gauss-Jordan without pivoting (what the device does, unblocked)
        auto invert = [](double* M, int m, int ld) {
            for (int p = 0; p < m; p++) {
                const double d = 1.0 / M[(size_t)p * ld + p];
                for (int j = 0; j < m; j++) M[(size_t)p * ld + j] *= d;
                M[(size_t)p * ld + p] = d;
                for (int i = 0; i < m; i++) {
                    if (i == p) continue;
                    const double f = M[(size_t)i * ld + p];
                    if (f == 0.0) continue;
                    for (int j = 0; j < m; j++) if (j != p) M[(size_t)i * ld + j] -= f * M[(size_t)p * ld + j];
                    M[(size_t)i * ld + p] = -f * d;
                }
            }
        };
        std::vector<double> arena((size_t)P.total, 0.0);
        for (size_t e = 0; e < P.pos.size(); e++) {
            if (P.pos[e] >= 0) arena[(size_t)P.pos[e]] = A.val[e];
            if (P.pos2[e] >= 0) arena[(size_t)P.pos2[e]] = A.val[e];
        }
        for (long long o : P.ones) arena[(size_t)o] = 1.0;
        double* D = arena.data() + P.off_D; double* PT = arena.data() + P.off_P; double* WT = arena.data() + P.off_W; double* S = arena.data() + P.off_S; double* C = arena.data() + P.off_C;
        for (int i = 0; i < P.nb; i++) {
            double* Di = D + (size_t)i * 4096;
            invert(Di, 64, 64);
            for (int r = 0; r < 64; r++) for (int c = r + 1; c < 64; c++) Di[r * 64 + c] = Di[c * 64 + r];
            const int s0 = P.sptr[(size_t)i], m = P.sptr[(size_t)i + 1] - s0;
            for (int c = 0; c < m; c++)
                for (int r = 0; r < 64; r++) {
                    double acc = 0.0;
                    for (int rp = 0; rp < 64; rp++) acc += Di[r * 64 + rp] * PT[(size_t)64 * (s0 + c) + rp];
                    WT[(size_t)64 * (s0 + c) + r] = acc;
                }
            for (int c1 = 0; c1 < m; c1++)
                for (int c2 = 0; c2 <= c1; c2++) {
                    double acc = 0.0;
                    for (int r = 0; r < 64; r++) acc += PT[(size_t)64 * (s0 + c1) + r] * WT[(size_t)64 * (s0 + c2) + r];
                    C[(size_t)P.coff[(size_t)i] + (size_t)c1 * m + c2] = acc;
                }
        }
        for (size_t d = 0; d < P.rdst.size(); d++) {
            double s = arena[(size_t)P.rdst[d]];
            for (int q = P.rptr[d]; q < P.rptr[d + 1]; q++) s -= C[(size_t)P.rsrc[(size_t)q]];
            arena[(size_t)P.rdst[d]] = s;
            if (P.rdst2[d] >= 0) arena[(size_t)P.rdst2[d]] = s;
        }
        invert(S, P.ns_pad, P.ns_pad);
        std::vector<double> g((size_t)P.ns_pad, 0.0), xs((size_t)P.ns_pad, 0.0);
        for (int j = 0; j < P.ns; j++) {
            double acc = 0.0;
            for (int q = P.aptr[(size_t)j]; q < P.aptr[(size_t)j + 1]; q++) {
                const int i = P.ablk[(size_t)q];
                for (int r = 0; r < 64; r++) {
                    const int row = P.irow[(size_t)i * 64 + r];
                    if (row >= 0) acc += WT[(size_t)64 * P.apan[(size_t)q] + r] * b[row];
                }
            }
            g[(size_t)j] = b[P.srow[(size_t)j]] - acc;
        }
        for (int i = 0; i < P.ns_pad; i++) { double acc = 0.0; for (int j = 0; j < P.ns_pad; j++) acc += S[(size_t)i * P.ns_pad + j] * g[(size_t)j]; xs[(size_t)i] = acc; }
        for (int j = 0; j < P.ns; j++) x[P.srow[(size_t)j]] = xs[(size_t)j];
        for (int i = 0; i < P.nb; i++) {
            const int s0 = P.sptr[(size_t)i], m = P.sptr[(size_t)i + 1] - s0;
            if (P.bsize[(size_t)i] < 1 || P.bsize[(size_t)i] > 64) return fail(SMG_ERR_INVALID, "schur plan: block %d has %d rows", i, P.bsize[(size_t)i]);
            for (int r = 0; r < 64; r++) {
                const int row = P.irow[(size_t)i * 64 + r];
                if ((row >= 0) != (r < P.bsize[(size_t)i])) return fail(SMG_ERR_INVALID, "schur plan: block %d: slots and size disagree", i);
                if (row < 0) continue;
                double acc = 0.0;
                for (int rp = 0; rp < 64; rp++) { const int rw = P.irow[(size_t)i * 64 + rp]; if (rw >= 0) acc += D[(size_t)i * 4096 + rp * 64 + r] * b[rw]; }
                for (int s = 0; s < m; s++) acc -= WT[(size_t)64 * (s0 + s) + r] * xs[(size_t)P.sidx[(size_t)s0 + s]];
                x[row] = acc;
            }
        }
        return SMG_OK;
    });
}
