// smg_capi.cpp -- implementation of the C ABI declared in include/smg.h.
//
// Host orchestration of the reference's solve path on one MI355X:
//   min_quad_with_fixed_mg_precompute  (reference src/min_quad_with_fixed_mg.cpp:3-51, :137-257)  -> smg_precompute
//   min_quad_with_fixed_mg_solve       (reference src/min_quad_with_fixed_mg.cpp:80-135, :288-361) -> smg_solve*
//   mg_VCycle and its pieces           (reference src/mg_VCycle.cpp:3-201)                          -> enqueue_vcycle
// The V-cycle never leaves the GPU: every kernel is enqueued on the handle's stream, the outer loop's
// break test runs on the device (Ctrl, smg_device.hpp) and one outer iteration is replayed as a hipGraph.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "../../include/smg.h"
#include "smg_hier.hpp"
#include "smg_mesh.hpp"

using namespace smg;

// ------------------------------------------------------------------------------------------------ errors
static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e__ = (expr);                                                                  \
        if (e__ != hipSuccess) return fail(SMG_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e__)); \
    } while (0)

// Nothing may throw across the C ABI: the entry points that allocate host memory run their bodies through this guard.
template <typename Fn>
static int guarded(const char* who, Fn&& body)
{
    try { return body(); }
    catch (const std::bad_alloc&) { return fail(SMG_ERR_ALLOC, "%s: out of host memory", who); }
    catch (const std::exception& e) { return fail(SMG_ERR_INVALID, "%s: %s", who, e.what()); }
    catch (...) { return fail(SMG_ERR_INVALID, "%s: unknown exception", who); }
}

extern "C" const char* smg_last_error(void) { return g_err.c_str(); }
extern "C" int smg_version(void) { return SMG_VERSION; }
extern "C" long long smg_device_bytes_live(void) { return (long long)smg::devbuf_live_bytes().load(); }
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
static int ensure_device(smg_hierarchy* h)
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

// The current HIP device is a per-thread setting: every entry point that touches the device -- and every worker thread of the
// precompute -- runs on the handle's device, whatever the calling thread had selected (one process may drive several GPUs, and a
// std::thread starts on device 0).  Restores the caller's selection on scope exit.
struct DeviceScope {
    int prev = -1, dev = -1;
    explicit DeviceScope(int d) : dev(d)
    {
        if (d < 0) return;
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != d) (void)hipSetDevice(d);
    }
    ~DeviceScope() { if (dev >= 0 && prev >= 0 && prev != dev) (void)hipSetDevice(prev); }
};

static int env_int(const char* name, int dflt)
{
    const char* v = std::getenv(name);
    return v && *v ? std::atoi(v) : dflt;
}

// SMG_TIMING=1: wall-clock of the precompute stages on stderr
struct StageTimer {
    bool on;
    std::chrono::steady_clock::time_point t0;
    StageTimer() : on(env_int("SMG_TIMING", 0) != 0), t0(std::chrono::steady_clock::now()) {}
    void lap(const char* what)
    {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[smg timing] %-38s %8.1f ms\n", what, 1e3 * std::chrono::duration<double>(t1 - t0).count());
        t0 = t1;
    }
};

static void drop_graphs(smg_hierarchy* h)
{
    if (h->g_iter) (void)hipGraphExecDestroy(h->g_iter);
    if (h->g_resid) (void)hipGraphExecDestroy(h->g_resid);
    if (h->g_cycle) (void)hipGraphExecDestroy(h->g_cycle);
    if (h->g_spec) (void)hipGraphExecDestroy(h->g_spec);
    h->g_iter = h->g_resid = h->g_cycle = h->g_spec = nullptr;
    h->g_k = 0;
}

hipError_t SellBuf::upload(const Sell& S)
{
    hipError_t e;
    if ((e = slice_row.upload(S.slice_row)) != hipSuccess) return e;
    if ((e = slice_off.upload(S.slice_off)) != hipSuccess) return e;
    if ((e = slice_w.upload(S.slice_w)) != hipSuccess) return e;
    if ((e = col.upload(S.col)) != hipSuccess) return e;
    if ((e = val.upload(S.val)) != hipSuccess) return e;
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
    if (S.n_rows == S.n_cols && S.color_slice_ptr.size() >= 2) {
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

// ------------------------------------------------------------------------------------------------ profc mirror
static int prof_scope_id(smg_hierarchy* h, const char* name)
{
    for (size_t i = 0; i < h->scopes.size(); i++) if (h->scopes[i].name == name) return (int)i;
    ProfScope s; s.name = name;
    h->scopes.push_back(s);
    return (int)h->scopes.size() - 1;
}
static hipEvent_t prof_event(smg_hierarchy* h)
{
    if (!h->ev_pool.empty()) { hipEvent_t e = h->ev_pool.back(); h->ev_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
struct ProfGuard {  // PROFC_NODE(name) (reference src/profc.h:9-13), timed on the GPU timeline
    smg_hierarchy* h; int idx = -1;
    ProfGuard(smg_hierarchy* hh, const char* name) : h(hh)
    {
        if (!h->prof_on) return;
        ProfRec r; r.scope = prof_scope_id(h, name); r.e0 = prof_event(h); r.e1 = prof_event(h);
        (void)hipEventRecord(r.e0, h->stream);
        h->recs.push_back(r);
        idx = (int)h->recs.size() - 1;
    }
    ~ProfGuard() { if (idx >= 0) (void)hipEventRecord(h->recs[idx].e1, h->stream); }
};
static void prof_collect(smg_hierarchy* h)
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

extern "C" int smg_hierarchy_set_chebyshev(smg_hierarchy* h, double cheby_fraction)
{
    if (!h) return fail(SMG_ERR_INVALID, "null handle");
    if (h->in_solve) return fail(SMG_ERR_INVALID, "smg_hierarchy_set_chebyshev called during a split-phase solve");
    if (cheby_fraction >= 1.0 || cheby_fraction != cheby_fraction) return fail(SMG_ERR_INVALID, "cheby_fraction must be in (0, 1)");
    if (cheby_fraction > 0.0) h->cheby_fraction = cheby_fraction;
    return SMG_OK;
}
static int spectral_bounds(smg_hierarchy* h);
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
static const char* check_compressed(int n_major, int n_minor, const int* ptr, const int* idx)
{
    if (ptr[0] != 0) return "pointer array must start at 0";
    for (int i = 0; i < n_major; i++) if (ptr[i + 1] < ptr[i]) return "pointer array is not monotone";
    const long nnz = ptr[n_major];
    for (long p = 0; p < nnz; p++) if (idx[p] < 0 || idx[p] >= n_minor) return "index out of range";
    return nullptr;
}

static int set_prolong(smg_hierarchy* h, int lv, Csr&& P)
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

// ------------------------------------------------------------------------------------------------ precompute
// Host half: the reference's sparse algebra, in the caller's numbering, bit-compatible accumulation order.
static int precompute_host(smg_hierarchy* h, Csr&& A, const int* known, int n_known)
{
    const int n = A.nr;
    const int L = h->n_levels;
    h->n_full = n;
    h->has_known = (known != nullptr && n_known > 0);
    h->known.clear(); h->unknown.clear();
    for (int lv = 1; lv < L; lv++) {
        if (h->lv[lv].P_full.empty()) return fail(SMG_ERR_INVALID, "level %d has no prolongation (smg_level_set_prolong)", lv);
        h->lv[lv].P = h->lv[lv].P_full;  // always restart from P_full (see smg.h)
    }
    if (L > 1 && h->lv[1].P_full.nr != n)
        return fail(SMG_ERR_INVALID, "A is %d x %d but P_1 has %d rows", n, n, h->lv[1].P_full.nr);
    h->nnz_input = (int)A.nnz();
    StageTimer tm;
    if (!h->has_known) {
        // reference src/min_quad_with_fixed_mg.cpp:17-22
        h->lhs_src.resize(A.nnz());
        std::iota(h->lhs_src.begin(), h->lhs_src.end(), 0);
        h->auk_src.clear();
        h->lv[0].A = std::move(A);
        h->Auk = Csr();
        {
            std::vector<std::function<void()>> tasks;
            for (int lv = 1; lv < L; lv++) tasks.push_back([h, lv] { h->lv[lv].PT = transpose(h->lv[lv].P); });
            parallel_tasks(tasks);
        }
    } else {
        // unknown = setdiff(0..n-1, known), ascending (:155-158); known keeps the caller's order (:178)
        std::vector<char> isk(n, 0);
        for (int i = 0; i < n_known; i++) {
            if (known[i] < 0 || known[i] >= n) return fail(SMG_ERR_INVALID, "known[%d] = %d out of range", i, known[i]);
            if (isk[known[i]]) return fail(SMG_ERR_INVALID, "known[%d] = %d appears twice", i, known[i]);
            isk[known[i]] = 1;
        }
        h->known.assign(known, known + n_known);
        for (int i = 0; i < n; i++) if (!isk[i]) h->unknown.push_back(i);
        h->lv[0].A = slice(A, &h->unknown, &h->unknown, &h->lhs_src);  // LHS = A(unknown, unknown)   (:166-167, :175)
        h->Auk = slice(A, &h->unknown, &h->known, &h->auk_src);        // Auk = A(unknown, known)     (:169-170, :176)
        if (L > 1) {
            h->lv[1].P = slice(h->lv[1].P_full, &h->unknown, nullptr);  // :185
            for (int lv = 1; lv < L; lv++) {
                Csr& P = h->lv[lv].P;
                // keep the columns holding at least one entry > 1e-15 (:190-203)
                std::vector<char> keepflag(P.nc, 0);
                for (long p = 0; p < P.nnz(); p++) if (P.val[p] > 1e-15) keepflag[P.col[p]] = 1;
                std::vector<int> keep;
                for (int c = 0; c < P.nc; c++) if (keepflag[c]) keep.push_back(c);
                if ((int)keep.size() < P.nc) {                                   // :206
                    P = slice(P, nullptr, &keep);                                // :210-211
                    if (lv < L - 1) h->lv[lv + 1].P = slice(h->lv[lv + 1].P_full, &keep, nullptr);  // :213-214
                } else break;                                                    // :216-219
            }
        }
        {
            std::vector<std::function<void()>> tasks;
            for (int lv = 1; lv < L; lv++) tasks.push_back([h, lv] { h->lv[lv].PT = transpose(h->lv[lv].P); });  // :226
            parallel_tasks(tasks);
        }
    }
    tm.lap("host: slices / transposes of P");
    // The locality order of the finest level is the longest sequential piece of the whole precompute (a Cuthill-McKee search over
    // all rows) and needs nothing but A_0's pattern: it starts now, on its own thread, beside the Galerkin products.
    auto pattern_key = [&](int lv) {
        const Csr& M = h->lv[lv].A;
        uint64_t key = 1469598103934665603ull;  // FNV-1a over (n, ptr, col)
        auto mix = [&](const int* p, size_t cnt) { for (size_t i = 0; i < cnt; i++) { key ^= (uint32_t)p[i]; key *= 1099511628211ull; } };
        const int hdr[2] = {M.nr, lv < L - 1 ? 1 : 0};
        mix(hdr, 2); mix(M.ptr.data(), M.ptr.size()); mix(M.col.data(), M.col.size());
        return key;
    };
    static const bool use_rcm = [] { const char* v = std::getenv("SMG_ORDER"); return !(v && std::string(v) == "induced"); }();
    std::vector<int> rcm0;
    std::thread rcm0_thread;
    struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } rcm0_joiner{rcm0_thread};
    uint64_t key0 = 0;
    if (L >= 3 && use_rcm && host_threads() > 1) {
        key0 = pattern_key(0);
        const Level& L0 = h->lv[0];
        if (!(key0 == L0.ord_key && (int)L0.ord.perm.size() == L0.A.nr)) rcm0_thread = std::thread([&] {
            const auto t0 = std::chrono::steady_clock::now();
            rcm0 = rcm_order(h->lv[0].A);
            if (tm.on) std::fprintf(stderr, "[smg timing] host:   (level 0 locality order, own thread: %.1f ms)\n", 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        });
    }
    // Galerkin  A_l = (PT_l * A_{l-1}) * P_l  (:25, :227)
    for (int lv = 1; lv < L; lv++) {
        Level& Lv = h->lv[lv];
        if (Lv.P.nc == 0) return fail(SMG_ERR_INVALID, "level %d has no unknowns left after constraint elimination", lv);
        if (Lv.P.nr != h->lv[lv - 1].A.nr)
            return fail(SMG_ERR_INVALID, "P_%d has %d rows but level %d has %d unknowns", lv, Lv.P.nr, lv - 1, h->lv[lv - 1].A.nr);
        Csr tmp = spgemm(Lv.PT, h->lv[lv - 1].A);
        Lv.A = spgemm(tmp, Lv.P);
    }
    tm.lap("host: Galerkin products");
    // small diagonal shift on the coarsest level only (:32-36, :236-241)
    {
        Csr& Ac = h->lv[L - 1].A;
        for (int i = 0; i < Ac.nr; i++) {
            bool found = false;
            for (int p = Ac.ptr[i]; p < Ac.ptr[i + 1]; p++) if (Ac.col[p] == i) { Ac.val[p] += 1e-12; found = true; break; }
            if (!found) return fail(SMG_ERR_INVALID, "coarsest matrix has no stored diagonal at row %d", i);
        }
    }
    for (int lv = 0; lv < L; lv++) {                       // A_diag (:39-41, :244-246)
        h->lv[lv].A_diag = diagonal(h->lv[lv].A);
        h->lv[lv].n = h->lv[lv].A.nr;
        // relax() divides by A_diag (src/mg_VCycle.cpp:157): a missing or zero diagonal would give Inf/NaN there
        if (lv < L - 1)
            for (int i = 0; i < h->lv[lv].n; i++)
                if (h->lv[lv].A_diag[i] == 0.0) return fail(SMG_ERR_INVALID, "level %d: zero or missing diagonal at row %d", lv, i);
    }
    tm.lap("host: shift, diagonals");
    // ---- device numbering (still host work): colour-major ordering of every smoothed level and the operators
    // expressed in it.  The coarsest level is only ever hit by the dense solve and keeps the caller's numbering.
    // coarse to fine, so that a subdivision level can inherit a 4-colouring from its parent; the RCM orders (the expensive,
    // sequential part of an ordering) of all levels that need one are computed concurrently first
    std::vector<uint64_t> keys(L);
    std::vector<char> need(L, 0);
    {
        std::vector<std::function<void()>> tasks;
        for (int lv = 0; lv < L; lv++) tasks.push_back([&, lv] {
            Level& Lv = h->lv[lv];
            const uint64_t key = (lv == 0 && key0) ? key0 : pattern_key(lv);
            keys[lv] = key;
            need[lv] = !(key == Lv.ord_key && (int)Lv.ord.perm.size() == Lv.n);   // else: same pattern as last time
        });
        parallel_tasks(tasks);
    }
    // Locality order of every smoothed level (new -> old).  Default: reverse Cuthill-McKee of each level's matrix (the per-level
    // searches run concurrently).  SMG_ORDER=induced: RCM on the coarsest smoothed level only, every finer level takes the
    // order induced by its parent level through P -- O(nnz) instead of a sequential search over a million rows; measured at C3:
    // 0.1 s less setup, sweeps 1-3 % slower.
    tm.lap("host:   pattern hashes");
    std::vector<std::vector<int>> rcm(L);
    const bool any_need = std::any_of(need.begin(), need.end(), [](char c) { return c != 0; });
    if (any_need) {
        if (use_rcm) {
            // the coarsest smoothed level is coloured from scratch (a search that can take longer than all the RCMs together):
            // it goes first in the task list and runs beside the finer levels' searches
            std::vector<std::function<void()>> tasks;
            if (L >= 2 && need[L - 2]) tasks.push_back([&] {
                Level& Lv = h->lv[L - 2];
                const auto t0 = std::chrono::steady_clock::now();
                rcm[L - 2] = rcm_order(Lv.A);
                Lv.ord = make_ordering(Lv.A, 512, nullptr, &rcm[L - 2]);
                if (tm.on) std::fprintf(stderr, "[smg timing] host:   (coarsest smoothed level: order + colouring from scratch %.1f ms)\n", 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
                Lv.ord_key = keys[L - 2];
                need[L - 2] = 0;
            });
            const bool early0 = rcm0_thread.joinable();
            for (int lv = 0; lv < L - 2; lv++) if (need[lv] && !(lv == 0 && early0)) tasks.push_back([&, lv] { rcm[lv] = rcm_order(h->lv[lv].A); });
            parallel_tasks(tasks);
            if (early0) { rcm0_thread.join(); rcm[0] = std::move(rcm0); }
        } else {
            std::vector<int> rank;
            for (int lv = L - 2; lv >= 0; lv--) {
                rcm[lv] = (lv == L - 2) ? rcm_order(h->lv[lv].A) : induced_order(h->lv[lv + 1].P, rank);
                rank.assign(h->lv[lv].n, 0);
                for (int t = 0; t < h->lv[lv].n; t++) rank[rcm[lv][t]] = t;
            }
        }
    }
    tm.lap("host:   locality orders (RCM) + coarsest colouring");
    for (int lv = L - 1; lv >= 0; lv--) {
        Level& Lv = h->lv[lv];
        if (!need[lv]) continue;
        if (lv == L - 1) Lv.ord = identity_ordering(Lv.n);
        else {
            std::vector<int> inherited;
            const Level& Lc = h->lv[lv + 1];
            const bool ok = (lv + 1 < L - 1) && Lc.ord.n_colors() <= 4 && (int)Lc.ord.color_of.size() == Lc.n &&
                            subdivision_colors(Lc.P, Lc.ord.color_of, Lv.A, inherited);
            if (tm.on) { char nm[64]; std::snprintf(nm, sizeof nm, "host:   level %d colours inherited=%d", lv, (int)ok); tm.lap(nm); }
            Lv.ord = make_ordering(Lv.A, 512, ok ? &inherited : nullptr, &rcm[lv]);
            if (tm.on) { char nm[64]; std::snprintf(nm, sizeof nm, "host:   level %d make_ordering", lv); tm.lap(nm); }
        }
        Lv.ord_key = keys[lv];
    }
    tm.lap("host: orderings + colourings");
    {
        std::vector<std::function<void()>> tasks;
        for (int lv = 0; lv < L; lv++) {
            tasks.push_back([h, lv, L] {
                Level& Lv = h->lv[lv];
                if (lv < L - 1) Lv.A_int = permute(Lv.A, Lv.ord.perm, Lv.ord.perm, &Lv.A_int_src);
                else { Lv.A_int = Lv.A; Lv.A_int_src.resize(Lv.A.nnz()); std::iota(Lv.A_int_src.begin(), Lv.A_int_src.end(), 0); }
            });
            if (lv >= 1) {
                tasks.push_back([h, lv] { Level& Lv = h->lv[lv]; Lv.P_int = permute(Lv.P, h->lv[lv - 1].ord.perm, Lv.ord.perm); });
                tasks.push_back([h, lv] { Level& Lv = h->lv[lv]; Lv.PT_int = permute(Lv.PT, Lv.ord.perm, h->lv[lv - 1].ord.perm); });
            }
        }
        parallel_tasks(tasks);
    }
    tm.lap("host: permuted operators");
    return SMG_OK;
}

// Gershgorin bound of D^-1 A per smoothed level (what the Chebyshev-Jacobi smoother is built on), from the SELL image the smoother
// streams, i.e. in the device numbering's summation order -- the same value the oracle computes on the level matrix in that numbering.
static int spectral_bounds(smg_hierarchy* h)
{
    const int L = h->n_levels;
    if (L < 2) return SMG_OK;
    HIPCHK(h->d_lam.ensure((size_t)L));
    for (int lv = 0; lv < L - 1; lv++) {
        Level& Lv = h->lv[lv];
        HIPCHK(launch_gershgorin(Lv.gs_on_transpose ? Lv.dAT.view : Lv.dA.view, h->d_lam.p + lv, h->stream));
    }
    std::vector<double> lam((size_t)L, 0.0);
    HIPCHK(hipMemcpyAsync(lam.data(), h->d_lam.p, (size_t)(L - 1) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int lv = 0; lv < L - 1; lv++) {
        if (!(lam[lv] > 0.0) || !std::isfinite(lam[lv])) return fail(SMG_ERR_INVALID, "level %d: no positive diagonal to scale by", lv);
        if (lam[lv] != h->lv[lv].lam) drop_graphs(h);   // the coefficients are kernel arguments of the captured launches
        h->lv[lv].lam = lam[lv];
    }
    h->lam_valid = true;
    return SMG_OK;
}
static int level_kind(const smg_hierarchy* h, int lv);
// lazily: only handles that smooth with Chebyshev-Jacobi pay the four small launches and the read-back
static int ensure_spectral_bounds(smg_hierarchy* h)
{
    if (h->lam_valid) return SMG_OK;
    bool need = false;
    for (int lv = 0; lv < h->n_levels - 1; lv++) if (level_kind(h, lv) == 2 /* LV_CHEBY */) need = true;
    return need ? spectral_bounds(h) : SMG_OK;
}

// Device half: renumber every level colour-major, build the SELL images, invert the coarsest matrix.
static int precompute_device(smg_hierarchy* h)
{
    const int L = h->n_levels;
    const int sellC = SELL_C;
    const bool region = env_int("SMG_REGION_ORDER", 1) != 0;   // A/B knob: region-major launch order
    HIPCHK(hipStreamSynchronize(h->stream));
    drop_graphs(h);
    for (int lv = 0; lv < L; lv++) {
        Level& Lv = h->lv[lv];
        Lv.b.release(); Lv.u.release(); Lv.r.release(); Lv.t.release(); Lv.d.release();
        Lv.b32.release(); Lv.u32.release(); Lv.r32.release(); Lv.t32.release(); Lv.d32.release();
    }
    h->kcap = 0; h->kcap32 = 0; h->f32_valid = false;
    StageTimer tm;
    // all SELL images concurrently on host threads, each uploaded by the task that built it (pageable-memory copies are bound by
    // the host-side staging copy, so they overlap with the other tasks' work and with each other)
    std::vector<int> bad(L, 0);
    {
        std::vector<std::function<void()>> tasks;
        std::vector<hipError_t> errs;
        errs.reserve((size_t)4 * L);
        if (L == 1) {
            // a single level goes straight to coarseSolve (src/mg_VCycle.cpp:28-33); the outer loop still needs A_0 for its residual
            errs.push_back(hipSuccess);
            hipError_t* eA = &errs.back();
            tasks.push_back([&, eA] {
                DeviceScope ds(h->device);
                Sell S = build_sell(h->lv[0].A_int, nullptr, sellC, false);
                *eA = h->lv[0].dA.upload(S);
            });
        }
        for (int lv = 0; lv < L; lv++) {
            if (lv < L - 1) {
                errs.push_back(hipSuccess);
                hipError_t* eA = &errs.back();
                tasks.push_back([&, lv, eA] {
                    DeviceScope ds(h->device);   // worker threads start on device 0
                    Sell S = build_sell(h->lv[lv].A_int, &h->lv[lv].ord.color_ptr, sellC, region);
                    *eA = h->lv[lv].dA.upload(S);
                });
                // relax() iterates InnerIterator(A, colIdx): the entries A(j, i) of COLUMN i (src/mg_VCycle.cpp:149-155,
                // "legal" because A is symmetric).  Galerkin products are symmetric only up to rounding, so the sweep
                // streams A^T wherever the two differ in any bit; the SpMV / residual keep the true rows.
                errs.push_back(hipSuccess);
                hipError_t* eT = &errs.back();
                tasks.push_back([&, lv, eT] {
                    DeviceScope ds(h->device);
                    Level& Lw = h->lv[lv];
                    Csr AT = transpose(Lw.A_int);
                    Lw.gs_on_transpose = !(AT.ptr == Lw.A_int.ptr && AT.col == Lw.A_int.col && AT.val == Lw.A_int.val);
                    Lw.dAT = SellBuf();
                    if (Lw.gs_on_transpose) {
                        if (!(AT.ptr == Lw.A_int.ptr && AT.col == Lw.A_int.col)) { bad[lv] = 1; return; }
                        Sell S = build_sell(AT, &Lw.ord.color_ptr, sellC, false);
                        *eT = Lw.dAT.upload(S);
                    }
                });
            }
            if (lv >= 1) {
                errs.push_back(hipSuccess);
                hipError_t* eP = &errs.back();
                // P and PT are launched whole: with their rows cut at the colour boundaries of the level they belong to, the slices get
                // the same region-major launch order as A, and the workgroups an XCD receives (a contiguous piece of that order) read
                // their gathers from one region of the mesh instead of from all over it (restriction at C3: 54 MB of HBM traffic per
                // launch for 33 MB of algorithmic bytes before)
                static const bool tr_region = env_int("SMG_TRANSFER_REGION_ORDER", 1) != 0;
                tasks.push_back([&, lv, eP] {
                    DeviceScope ds(h->device);
                    const bool cut = tr_region && region && h->lv[lv - 1].ord.color_ptr.size() > 2;
                    Sell S = build_sell(h->lv[lv].P_int, cut ? &h->lv[lv - 1].ord.color_ptr : nullptr, sellC, cut);
                    *eP = h->lv[lv].dP.upload(S);
                });
                errs.push_back(hipSuccess);
                hipError_t* eQ = &errs.back();
                tasks.push_back([&, lv, eQ] {
                    DeviceScope ds(h->device);
                    const bool cut = tr_region && region && lv < L - 1 && h->lv[lv].ord.color_ptr.size() > 2;
                    // rows with many entries (a coarse vertex of a decimated level that absorbed dozens of fine ones) leave the panels:
                    // a panel row is one chain of dependent batches and the longest one sets the duration of the restriction launch
                    // (ogre.obj level 0 -> 1: a row of 177 entries, 32 us of a 260 us cycle); see SellDev::long_* / k_long_ax
                    static const int long_min = env_int("SMG_LONG_ROW_MIN", 17);
                    const Csr& M = h->lv[lv].PT_int;
                    std::vector<int> lrow, lptr{0}, lcol;
                    std::vector<double> lval;
                    if (long_min > 0)
                        for (int r = 0; r < M.nr; r++)
                            if (M.ptr[r + 1] - M.ptr[r] >= long_min) {
                                lrow.push_back(r);
                                lcol.insert(lcol.end(), M.col.begin() + M.ptr[r], M.col.begin() + M.ptr[r + 1]);
                                lval.insert(lval.end(), M.val.begin() + M.ptr[r], M.val.begin() + M.ptr[r + 1]);
                                lptr.push_back((int)lcol.size());
                            }
                    if (lrow.empty()) {
                        Sell S = build_sell(M, cut ? &h->lv[lv].ord.color_ptr : nullptr, sellC, cut);
                        *eQ = h->lv[lv].dPT.upload(S);
                        if (*eQ == hipSuccess) *eQ = h->lv[lv].dPT.upload_long(lrow, lptr, lcol, lval);
                        return;
                    }
                    Csr Ms;     // M with the long rows emptied
                    Ms.nr = M.nr; Ms.nc = M.nc; Ms.ptr.assign((size_t)M.nr + 1, 0);
                    {
                        size_t li = 0;
                        for (int r = 0; r < M.nr; r++) {
                            const bool is_long = li < lrow.size() && lrow[li] == r;
                            if (is_long) li++;
                            else { Ms.col.insert(Ms.col.end(), M.col.begin() + M.ptr[r], M.col.begin() + M.ptr[r + 1]); Ms.val.insert(Ms.val.end(), M.val.begin() + M.ptr[r], M.val.begin() + M.ptr[r + 1]); }
                            Ms.ptr[(size_t)r + 1] = (int)Ms.col.size();
                        }
                    }
                    Sell S = build_sell(Ms, cut ? &h->lv[lv].ord.color_ptr : nullptr, sellC, cut);
                    *eQ = h->lv[lv].dPT.upload(S);
                    if (*eQ == hipSuccess) *eQ = h->lv[lv].dPT.upload_long(lrow, lptr, lcol, lval);
                });
            }
        }
        parallel_tasks(tasks);
        for (int lv = 0; lv < L; lv++)
            if (bad[lv]) return fail(SMG_ERR_INVALID, "level %d matrix is not structurally symmetric", lv);
        for (hipError_t e : errs) HIPCHK(e);
    }
    tm.lap("device: SELL images built and uploaded");
    // level-0 index maps
    {
        const Level& L0 = h->lv[0];
        std::vector<int> map0(L0.n);
        for (int i = 0; i < L0.n; i++) map0[i] = h->has_known ? h->unknown[L0.ord.perm[i]] : L0.ord.perm[i];
        HIPCHK(h->d_map0.upload(map0));
        HIPCHK(h->d_perm0.upload(L0.ord.perm));
        if (h->has_known) {
            HIPCHK(h->d_unknown.upload(h->unknown));
            HIPCHK(h->d_known.upload(h->known));
            HIPCHK(h->d_auk_ptr.upload(h->Auk.ptr));
            HIPCHK(h->d_auk_col.upload(h->Auk.col));
            HIPCHK(h->d_auk_val.upload(h->Auk.val));
        }
    }
    tm.lap("device: index maps");
    // coarsest level: dense inverse on the device (stands in for solver.compute(Ac), :47-48 / :253-254)
    {
        const Level& Lc = h->lv[L - 1];
        const int nc = Lc.n;
        const int np = ((nc + 63) / 64) * 64;
        h->nc = nc; h->nc_pad = np;
        if ((double)np * np * 8.0 > 96e9)
            return fail(SMG_ERR_ALLOC, "coarsest level has %d unknowns: its dense inverse (%.0f GB) is out of range -- add levels", nc, (double)np * np * 8e-9);
        // dense image on the device: the few entries travel, not n^2 zeros
        std::vector<long long> pos(Lc.A.nnz());
        for (int i = 0; i < nc; i++)
            for (int p = Lc.A.ptr[i]; p < Lc.A.ptr[i + 1]; p++) pos[p] = (long long)i * np + Lc.A.col[p];
        DevBuf<long long> d_pos;
        DevBuf<double> d_val;
        HIPCHK(d_pos.upload(pos));
        HIPCHK(d_val.upload(Lc.A.val));
        HIPCHK(h->d_Ainv.ensure((size_t)np * np));
        HIPCHK(launch_dense_from_csr(h->d_Ainv.p, np, nc, d_val.p, d_pos.p, (int)Lc.A.nnz(), h->stream));
        if (env_int("SMG_SYM_COARSE", 1)) HIPCHK(h->d_sympart.ensure((size_t)(np / 64) * (np / 64) * 64)); else h->d_sympart.release();
        DevBuf<double> work;
        HIPCHK(work.alloc((size_t)2 * np * 64 + 2 * 64 * 64));
        HIPCHK(launch_spd_inverse(h->d_Ainv.p, np, work.p, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    tm.lap("device: coarse dense inverse");
    h->lam_valid = false;   // the Gershgorin bounds are computed when a Chebyshev smoother first asks for them (ensure_spectral_bounds)
    return SMG_OK;
}

// ---- value-only re-precompute --------------------------------------------------------
// Time-stepping callers hand in a new matrix with the SAME sparsity every step (05_example_mean_curvature_flow/
// main.cpp:74, 06_example_balloon_sim/implicit_euler_mg_balloon.h:75).  Then everything structural (unknown set,
// sliced P, Galerkin patterns, colouring, SELL layout, graphs) is unchanged and the numeric work moves to the GPU:
// slice gathers, two fixed-recipe SpGEMM stages per level (bit-identical to the host spgemm), SELL value refresh and
// the dense coarse inverse.

static uint64_t fnv_mix(uint64_t key, const int* p, size_t cnt)
{
    // hashed in fixed blocks of 64 Ki entries, the blocks concurrently on the host threads, the block hashes chained in order:
    // the value does not depend on the number of threads (a time step's re-precompute hashes the 8 M pattern entries of a
    // 1 M-vertex mesh before anything else: 6.5 ms as one sequential chain)
    constexpr size_t B = 65536;
    const size_t nblk = (cnt + B - 1) / B;
    if (nblk <= 1) {
        for (size_t i = 0; i < cnt; i++) { key ^= (uint32_t)p[i]; key *= 1099511628211ull; }
        return key;
    }
    std::vector<uint64_t> part(nblk);
    parallel_for((long)nblk, 4, [&](long b0, long b1) {
        for (long b = b0; b < b1; b++) {
            uint64_t k = 1469598103934665603ull;
            const size_t e = std::min(cnt, (size_t)(b + 1) * B);
            for (size_t i = (size_t)b * B; i < e; i++) { k ^= (uint32_t)p[i]; k *= 1099511628211ull; }
            part[b] = k;
        }
    });
    for (size_t b = 0; b < nblk; b++) { key ^= part[b]; key *= 1099511628211ull; }
    return key;
}

static uint64_t precompute_key(const smg_hierarchy* h, int n, const int* rowptr, const int* col, const int* known, int n_known)
{
    uint64_t key = 1469598103934665603ull;
    const int hdr[4] = {n, n_known, h->p_version, h->n_levels};
    key = fnv_mix(key, hdr, 4);
    key = fnv_mix(key, rowptr, (size_t)n + 1);
    key = fnv_mix(key, col, (size_t)rowptr[n]);
    if (known) key = fnv_mix(key, known, (size_t)n_known);
    return key ? key : 1;
}

static int build_recipes(smg_hierarchy* h)
{
    const int L = h->n_levels;
    const int sellC = SELL_C;
    HIPCHK(hipStreamSynchronize(h->stream));
    drop_graphs(h);  // the GS launches move to the A^T images on every level
    // all levels concurrently (maps of the SELL slots; the two numeric Galerkin stages as recipes); every task uploads what it built
    std::vector<int> bad(L, 0);
    std::vector<hipError_t> errs((size_t)2 * L, hipSuccess);
    StageTimer tm;
    {
        std::vector<std::function<void()>> tasks;
        auto up = [](hipError_t& acc, hipError_t e) { if (acc == hipSuccess) acc = e; };
        for (int lv = 0; lv < L; lv++) {
            if (lv < L - 1) tasks.push_back([&, lv] {
                DeviceScope ds(h->device);
                // SELL slot -> caller CSR entry, for A and for A^T (the sweep always reads A^T in this mode: whether new
                // values are bit-symmetric cannot be known in advance)
                Level& Lv = h->lv[lv];
                hipError_t& er = errs[2 * lv];
                std::vector<int> m;
                {
                    Sell S = build_sell(Lv.A_int, &Lv.ord.color_ptr, sellC, false);
                    m.resize(S.entry.size());
                    for (size_t i = 0; i < S.entry.size(); i++) m[i] = S.entry[i] >= 0 ? Lv.A_int_src[S.entry[i]] : -1;
                }
                up(er, Lv.mapA.upload(m));
                std::vector<int> tsrc;
                Csr AT = transpose(Lv.A_int, &tsrc);
                if (!(AT.ptr == Lv.A_int.ptr && AT.col == Lv.A_int.col)) { bad[lv] = 1; return; }
                Sell ST = build_sell(AT, &Lv.ord.color_ptr, sellC, false);
                m.resize(ST.entry.size());
                for (size_t i = 0; i < ST.entry.size(); i++) m[i] = ST.entry[i] >= 0 ? Lv.A_int_src[tsrc[ST.entry[i]]] : -1;
                up(er, Lv.mapAT.upload(m));
                if (!Lv.gs_on_transpose) { up(er, Lv.dAT.upload(ST)); Lv.gs_on_transpose = true; }
            });
            tasks.push_back([&, lv] {
                DeviceScope ds(h->device);
                Level& Lv = h->lv[lv];
                hipError_t& er = errs[2 * lv + 1];
                up(er, Lv.d_Aval.upload(Lv.A.val));
                if (lv == 0) return;
                const Csr& Af = h->lv[lv - 1].A;
                Csr T = spgemm(Lv.PT, Af);
                Recipe r;
                spgemm_recipe(Lv.PT, Af, true, T, r);      // T = PT * A_{lv-1}:  PT constant
                up(er, Lv.r1_ptr.upload(r.ptr)); up(er, Lv.r1_idx.upload(r.idx)); up(er, Lv.r1_coef.upload(r.coef));
                spgemm_recipe(T, Lv.P, false, Lv.A, r);    // A_lv = T * P:       P constant
                up(er, Lv.r2_ptr.upload(r.ptr)); up(er, Lv.r2_idx.upload(r.idx)); up(er, Lv.r2_coef.upload(r.coef));
                Lv.nnzT = (int)T.nnz();
                up(er, Lv.d_Tval.alloc(T.nnz()));
            });
        }
        parallel_tasks(tasks);
    }
    for (int lv = 0; lv < L; lv++)
        if (bad[lv]) return fail(SMG_ERR_INVALID, "level %d matrix is not structurally symmetric", lv);
    for (hipError_t e : errs) HIPCHK(e);
    tm.lap("recipes: host work + uploads");
    {
        const Level& Lc = h->lv[L - 1];
        std::vector<long long> pos(Lc.A.nnz());
        std::vector<int> dg;
        for (int i = 0; i < Lc.n; i++)
            for (int p = Lc.A.ptr[i]; p < Lc.A.ptr[i + 1]; p++) {
                pos[p] = (long long)i * h->nc_pad + Lc.A.col[p];
                if (Lc.A.col[p] == i) dg.push_back(p);
            }
        HIPCHK(h->d_dense_pos.upload(pos));
        HIPCHK(h->d_diag_idx.upload(dg));
    }
    HIPCHK(h->d_lhs_src.upload(h->lhs_src));
    if (h->has_known) HIPCHK(h->d_auk_src.upload(h->auk_src));
    HIPCHK(h->d_Afull.alloc((size_t)std::max(h->nnz_input, 1)));
    h->recipes_built = true;
    return SMG_OK;
}

// d_val: the caller's new values (device, caller CSR order)
static int precompute_values_device(smg_hierarchy* h, const double* d_val)
{
    const int L = h->n_levels;
    hipStream_t st = h->stream;
    Level& L0 = h->lv[0];
    HIPCHK(launch_gather_vals(L0.d_Aval.p, d_val, h->d_lhs_src.p, (size_t)L0.A.nnz(), st));          // LHS = A(unknown, unknown)
    if (h->has_known) HIPCHK(launch_gather_vals(h->d_auk_val.p, d_val, h->d_auk_src.p, (size_t)h->Auk.nnz(), st));  // Auk
    for (int lv = 0; lv < L; lv++) {
        Level& Lv = h->lv[lv];
        if (lv >= 1) {
            Level& Lf = h->lv[lv - 1];
            HIPCHK(launch_recipe(Lv.nnzT, Lv.r1_ptr.p, Lv.r1_idx.p, Lv.r1_coef.p, Lf.d_Aval.p, Lv.d_Tval.p, st));
            HIPCHK(launch_recipe((int)Lv.A.nnz(), Lv.r2_ptr.p, Lv.r2_idx.p, Lv.r2_coef.p, Lv.d_Tval.p, Lv.d_Aval.p, st));
        }
        if (lv == L - 1) {
            HIPCHK(launch_add_at(Lv.d_Aval.p, h->d_diag_idx.p, (int)h->d_diag_idx.n, 1e-12, st));          // :32-36 / :236-241
        } else {
            HIPCHK(launch_gather_vals(const_cast<double*>(Lv.dA.view.val), Lv.d_Aval.p, Lv.mapA.p, (size_t)Lv.dA.padded, st));
            HIPCHK(launch_gather_vals(const_cast<double*>(Lv.dAT.view.val), Lv.d_Aval.p, Lv.mapAT.p, (size_t)Lv.dAT.padded, st));
        }
    }
    // coarsest: dense image + inverse (solver.compute(Ac), :47-48 / :253-254)
    {
        const Level& Lc = h->lv[L - 1];
        HIPCHK(launch_dense_from_csr(h->d_Ainv.p, h->nc_pad, h->nc, Lc.d_Aval.p, h->d_dense_pos.p, (int)Lc.A.nnz(), st));
        DevBuf<double> work;
        HIPCHK(work.alloc((size_t)2 * h->nc_pad * 64 + 2 * 64 * 64));
        HIPCHK(launch_spd_inverse(h->d_Ainv.p, h->nc_pad, work.p, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    h->host_stale = true;
    h->f32_valid = false;   // the fp32 copies are re-made from the new values when a mixed solve asks for them
    h->lam_valid = false;
    return SMG_OK;
}

// bring the host copies (mg[l].A, A_diag, Auk, A_int) up to date after a device-side re-precompute
static int refresh_host_values(smg_hierarchy* h)
{
    if (!h->host_stale) return SMG_OK;
    for (int lv = 0; lv < h->n_levels; lv++) {
        Level& Lv = h->lv[lv];
        HIPCHK(hipMemcpy(Lv.A.val.data(), Lv.d_Aval.p, Lv.A.val.size() * sizeof(double), hipMemcpyDeviceToHost));
        Lv.A_diag = diagonal(Lv.A);
        for (size_t e = 0; e < Lv.A_int.val.size(); e++) Lv.A_int.val[e] = Lv.A.val[Lv.A_int_src[e]];
    }
    if (h->has_known && h->Auk.nnz() > 0)
        HIPCHK(hipMemcpy(h->Auk.val.data(), h->d_auk_val.p, h->Auk.val.size() * sizeof(double), hipMemcpyDeviceToHost));
    h->host_stale = false;
    return SMG_OK;
}

extern "C" int smg_precompute_values_device(smg_hierarchy* h, const double* d_val)
{
    if (!h || !d_val) return fail(SMG_ERR_INVALID, "smg_precompute_values_device: bad arguments");
    if (!h->precomputed || h->device < 0) return fail(SMG_ERR_INVALID, "smg_precompute_values_device: run a full smg_precompute with this sparsity first");
    if (h->in_solve) return fail(SMG_ERR_INVALID, "smg_precompute_values_device called during a split-phase solve");
    if (!h->input_canonical) return fail(SMG_ERR_INVALID, "the matrix given to smg_precompute had unsorted or duplicate entries: entry indices are not stable");
    if (h->n_levels < 2) return fail(SMG_ERR_INVALID, "smg_precompute_values_device: single-level hierarchies take the full smg_precompute");
    DeviceScope dsc(h->device);
    if (!h->recipes_built) { int rc = build_recipes(h); if (rc) return rc; }
    int rc = precompute_values_device(h, d_val);
    if (rc != SMG_OK) h->precomputed = false;
    return rc;
}

struct smg_assembler {
    smg::AssemblyPlan plan;
    smg::DevBuf<int> F, l_ptr, l_idx, m_ptr, m_idx, diag_of;
    smg::DevBuf<signed char> l_sgn;
    smg::DevBuf<double> Qc, Qm, Md;
};

static int smg_assembler_create_impl(const int* F, int nF, int nV, smg_assembler** out)
{
    if (!F || nF <= 0 || nV <= 0 || !out) return fail(SMG_ERR_INVALID, "smg_assembler_create: bad arguments");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(SMG_ERR_NO_DEVICE, "no HIP device: libsmg has no CPU fallback");
    smg_assembler* a = new (std::nothrow) smg_assembler();
    if (!a) return fail(SMG_ERR_ALLOC, "out of memory");
    std::vector<int> Fv(F, F + (size_t)nF * 3);
    for (int v : Fv) if (v < 0 || v >= nV) { delete a; return fail(SMG_ERR_INVALID, "face index out of range"); }
    a->plan = make_assembly_plan(Fv, nV);
    hipError_t e = hipSuccess;
    if (e == hipSuccess) e = a->F.upload(Fv);
    if (e == hipSuccess) e = a->l_ptr.upload(a->plan.l_ptr);
    if (e == hipSuccess) e = a->l_idx.upload(a->plan.l_idx);
    if (e == hipSuccess) e = a->l_sgn.upload(a->plan.l_sgn);
    if (e == hipSuccess) e = a->m_ptr.upload(a->plan.m_ptr);
    if (e == hipSuccess) e = a->m_idx.upload(a->plan.m_idx);
    if (e == hipSuccess) e = a->diag_of.upload(a->plan.diag_of);
    if (e == hipSuccess) e = a->Qc.alloc((size_t)nF * 3);
    if (e == hipSuccess) e = a->Qm.alloc((size_t)nF * 3);
    if (e == hipSuccess) e = a->Md.alloc((size_t)nV);
    if (e != hipSuccess) { delete a; return fail(SMG_ERR_HIP, "smg_assembler_create: %s", hipGetErrorString(e)); }
    *out = a;
    return SMG_OK;
}

extern "C" int smg_assembler_create(const int* F, int nF, int nV, smg_assembler** out)
{
    return guarded("smg_assembler_create", [&]() { return smg_assembler_create_impl(F, nF, nV, out); });
}
extern "C" void smg_assembler_destroy(smg_assembler* a) { delete a; }
extern "C" int smg_assembler_pattern(const smg_assembler* a, int* nnz, int* rowptr, int* col)
{
    if (!a) return fail(SMG_ERR_INVALID, "null assembler");
    if (nnz) *nnz = (int)a->plan.pattern.nnz();
    if (rowptr) std::copy(a->plan.pattern.ptr.begin(), a->plan.pattern.ptr.end(), rowptr);
    if (col) std::copy(a->plan.pattern.col.begin(), a->plan.pattern.col.end(), col);
    return SMG_OK;
}
extern "C" int smg_assemble(smg_assembler* a, const double* d_V, int voronoi, double mass_coef, double lap_coef, double* d_val,
                            double* d_mass, double* d_Lval, void* hip_stream)
{
    if (!a || !d_V || !d_val) return fail(SMG_ERR_INVALID, "smg_assemble: bad arguments");
    hipStream_t st = (hipStream_t)hip_stream;
    HIPCHK(launch_assemble(a->plan.nV, a->plan.nF, (int)a->plan.pattern.nnz(), d_V, a->F.p, voronoi, a->l_ptr.p, a->l_idx.p, a->l_sgn.p,
                           a->m_ptr.p, a->m_idx.p, a->diag_of.p, a->Qc.p, a->Qm.p, a->Md.p, mass_coef, lap_coef, d_val, d_Lval, st));
    if (d_mass) HIPCHK(hipMemcpyAsync(d_mass, a->Md.p, (size_t)a->plan.nV * sizeof(double), hipMemcpyDeviceToDevice, st));
    return SMG_OK;
}

static int smg_precompute_impl(smg_hierarchy* h, int n, const int* rowptr, const int* col, const double* val,
                              const int* known, int n_known)
{
    if (!h || n <= 0 || !rowptr || !col || !val) return fail(SMG_ERR_INVALID, "smg_precompute: bad arguments");
    if (h->in_solve) return fail(SMG_ERR_INVALID, "smg_precompute called between smg_solve_begin and smg_solve_end");
    if (known == nullptr) n_known = 0;
    if (n_known < 0 || n_known >= n) return fail(SMG_ERR_INVALID, "smg_precompute: n_known = %d must be in [0, n)", n_known);
    StageTimer tmv;
    if (const char* e = check_compressed(n, n, rowptr, col)) return fail(SMG_ERR_INVALID, "smg_precompute: %s", e);
    tmv.lap("precompute: input check");
    const uint64_t key = precompute_key(h, n, rowptr, col, known, n_known);
    tmv.lap("precompute: pattern key");
    if (h->precomputed && h->device >= 0 && key == h->pre_key && h->input_canonical && h->n_levels > 1 && env_int("SMG_NO_FAST_PRECOMPUTE", 0) == 0) {
        DeviceScope dsc(h->device);
        // same sparsity, same constraints, same prolongations: only the values changed
        int rc = SMG_OK;
        if (!h->recipes_built) rc = build_recipes(h);
        if (rc == SMG_OK) {
            hipError_t e = hipMemcpyAsync(h->d_Afull.p, val, (size_t)rowptr[n] * sizeof(double), hipMemcpyHostToDevice, h->stream);
            if (e != hipSuccess) rc = fail(SMG_ERR_HIP, "hipMemcpyAsync: %s", hipGetErrorString(e));
        }
        if (rc == SMG_OK && tmv.on) { (void)hipStreamSynchronize(h->stream); tmv.lap("precompute: values to the device"); }
        if (rc == SMG_OK) rc = precompute_values_device(h, h->d_Afull.p);
        tmv.lap("precompute: value-only device work");
        if (rc != SMG_OK) h->precomputed = false;
        return rc;
    }
    h->precomputed = false;
    h->recipes_built = false;
    h->host_stale = false;
    Csr A = csr_from_arrays(n, n, rowptr, col, val);
    h->input_canonical = (A.nnz() == (long)rowptr[n]) && std::equal(A.col.begin(), A.col.end(), col);
    int rc = precompute_host(h, std::move(A), known, n_known);
    if (rc != SMG_OK) return rc;
    rc = ensure_device(h);
    if (rc != SMG_OK) return rc;
    DeviceScope dsc(h->device);
    rc = precompute_device(h);
    if (rc != SMG_OK) return rc;
    h->pre_key = key;
    h->precomputed = true;
    return SMG_OK;
}

extern "C" int smg_precompute(smg_hierarchy* h, int n, const int* rowptr, const int* col, const double* val,
                              const int* known, int n_known)
{
    return guarded("smg_precompute", [&]() { return smg_precompute_impl(h, n, rowptr, col, val, known, n_known); });
}

// ------------------------------------------------------------------------------------------------ V-cycle
enum { LV_GS = 0, LV_JACOBI = 1, LV_CHEBY = 2 };
static int level_kind(const smg_hierarchy* h, int lv);
static bool level_is_jacobi(const smg_hierarchy* h, int lv);

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
            if (lv < L - 1 || L == 1) maxblocks = std::max(maxblocks, (size_t)sell_blocks(Lv.dA.view.n_slices) * ((k + 3) / 4) + (size_t)sell_wide_blocks(Lv.dA.view.n_slices, k));
        }
        // colour by colour (the level-0 head of an outer iteration, enqueue_residual_ss) every launch rounds its block count up on its own
        maxblocks += (h->lv[0].dA.color_slice_ptr.size() + 1) * (size_t)((k + 3) / 4 + 8);
        HIPCHK(h->d_partials.alloc(std::max<size_t>(maxblocks, 1)));
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
    return ensure_spectral_bounds(h);
}

// ---- mixed precision: fp32 images of the operators and an fp32 V-cycle ------------------------------------------------
static int ensure_fp32(smg_hierarchy* h, int k)
{
    const int L = h->n_levels;
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
            if (lv < L - 1) {
                if ((rc = mk(Lv.dA, Lv.a32, Lv.dA32))) return rc;
                if (Lv.gs_on_transpose) { if ((rc = mk(Lv.dAT, Lv.at32, Lv.dAT32))) return rc; }
            }
            if (lv >= 1) {
                if ((rc = mk(Lv.dP, Lv.p32, Lv.dP32))) return rc;
                if ((rc = mk(Lv.dPT, Lv.pt32, Lv.dPT32))) return rc;
            }
        }
        HIPCHK(h->d_Ainv32.ensure((size_t)h->nc_pad * h->nc_pad));
        HIPCHK(launch_cvt_f64_f32(h->d_Ainv32.p, h->d_Ainv.p, (size_t)h->nc_pad * h->nc_pad, h->stream));
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
    return SMG_OK;
}

// ---- the smoother of a level -------------------------------------------------------------------------------------------
// SMG_SMOOTH_GS (default): the reference's relax().  SMG_SMOOTH_JACOBI / _HYBRID: damped Jacobi on all / on the small levels
// (BASELINE.json north_star: "Gauss-Seidel/Jacobi smoothing"; one whole-matrix launch per sweep instead of one per colour).
static int level_kind(const smg_hierarchy* h, int lv)
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
static bool level_is_jacobi(const smg_hierarchy* h, int lv) { return level_kind(h, lv) != LV_GS; }   // needs the second iterate buffer

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
    { return launch_dense_gemv_add(h->d_Ainv.p, h->nc, h->nc_pad, L.b.p, L.u.p, k, ctrl, h->stream, h->d_sympart.p); }
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
    { return launch_dense_gemv_add_f32(h->d_Ainv32.p, h->nc, h->nc_pad, L.b32.p, L.u32.p, k, ctrl, h->stream, (float*)h->d_sympart.p); }
};

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
    const SellBuf& G = Lv.gs_on_transpose ? Lv.dAT : Lv.dA;
    const std::vector<int>& cs = G.color_slice_ptr;
    for (int it = first == FIRST_SWEEP ? 1 : 0; it < iters; it++)
        for (size_t c = (it == 0 && first == FIRST_LAUNCH) ? 1 : 0; c + 1 < cs.size(); c++) {
            if (it == 1 && first == FIRST_SWEEP) HIPCHK(Prec<T>::sell(SELL_GS_OOP, Prec<T>::G(Lv), cs[c], cs[c + 1], t, b, u, k, ctrl, h->stream));
            else HIPCHK(Prec<T>::sell(SELL_GS, Prec<T>::G(Lv), cs[c], cs[c + 1], u, b, u, k, ctrl, h->stream));
        }
    return SMG_OK;
}

// `iters` damped-Jacobi sweeps, ping-pong between buf[0] and buf[1]: sweep s reads buf[*cur], writes the other, flips *cur.
template <typename T>
static int enqueue_jacobi(smg_hierarchy* h, int lv, const T* b, T* const buf[2], int* cur, int k, int iters, const Ctrl* ctrl)
{
    Level& Lv = h->lv[lv];
    ProfGuard pg(h, "MG: relaxation");
    const SellDev& G = Prec<T>::G(Lv);
    for (int it = 0; it < iters; it++) {
        HIPCHK(Prec<T>::sell(SELL_JACOBI, G, 0, G.n_slices, buf[*cur], b, buf[1 - *cur], k, ctrl, h->stream, nullptr, nullptr, h->omega));
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
    const SellDev& G = Prec<T>::G(Lv);
    std::vector<ChebyCoef> cf;
    cheby_coefs(Lv.lam, h->cheby_fraction, iters + 1, cf);
    for (int s = first_done ? 1 : 0; s <= iters; s++) {
        FirstColour fc;
        Prec<T>::set_d(fc, Lv);
        fc.c1 = cf[s].c1;
        HIPCHK(Prec<T>::sell(SELL_CHEBY, G, 0, G.n_slices, buf[*cur], b, buf[1 - *cur], k, ctrl, h->stream, nullptr, &fc, cf[s].c2));
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
    int rc;
    if (kind == LV_JACOBI) {
        if (first_done) cur = 1;
        rc = enqueue_jacobi<T>(h, lv, Prec<T>::b(Lv), buf, &cur, k, pre - (first_done ? 1 : 0), ctrl);            // :36
    } else if (kind == LV_CHEBY) {
        if (first_done) cur = 1;
        rc = enqueue_cheby<T>(h, lv, Prec<T>::b(Lv), buf, &cur, k, pre, ctrl, first_done);                         // :36
    } else rc = enqueue_gs<T>(h, lv, Prec<T>::b(Lv), buf[0], k, pre, ctrl, first, buf[1]);                         // :36
    if (rc) return rc;
    {   // r = B - A u  (:40-42)
        ProfGuard pg(h, "MG: residual");
        HIPCHK(Prec<T>::sell(SELL_RESID, Prec<T>::A(Lv), 0, Prec<T>::A(Lv).n_slices, buf[cur], Prec<T>::b(Lv), Prec<T>::r(Lv), k, ctrl, h->stream));
    }
    // With uc = 0 the first launch of the coarse level's first pre-smoothing sweep computes (rc_i - 0) / a_ii for the rows it covers
    // (the first colour / with Jacobi all rows, damped): the restriction launch writes that itself, bit for bit the same value, and
    // the sweep starts one launch later.
    const SellBuf& Gc = Lc.gs_on_transpose ? Lc.dAT : Lc.dA;
    const int kind_c = level_kind(h, lv + 1);
    const bool jac_c = kind_c != LV_GS;
    const bool fuse = fuse_first_colour() && lv + 1 < L - 1 && pre > 0 && Prec<T>::has_vals(Prec<T>::G(Lc)) && (jac_c ? Gc.n_all > 0 : Gc.n_first > 0);
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
        HIPCHK(Prec<T>::sell(SELL_AX, Prec<T>::PT(Lc), 0, Prec<T>::PT(Lc).n_slices, Prec<T>::r(Lv), nullptr, Prec<T>::b(Lc), k, ctrl, h->stream, init,
                             fuse ? &fc : nullptr));
    }
    rc = enqueue_vcycle_t<T>(h, lv + 1, k, pre, post, ctrl, fuse ? FIRST_LAUNCH : FIRST_NONE);  // :48
    if (rc) return rc;
    {   // u = u + P uc  (:51-53, :91).  A Jacobi level with an odd number of post-smoothing sweeps to go adds out of place, so that
        // the last sweep lands in u.
        ProfGuard pg(h, "MG: prolong");
        int dst = cur;
        const int flips = kind == LV_CHEBY ? (post > 0 ? post + 1 : 0) : post;   // buffer switches of the post-smoothing
        if (jac && ((cur + flips) & 1)) dst = 1 - cur;
        HIPCHK(Prec<T>::sell(SELL_ADD, Prec<T>::P(Lc), 0, Prec<T>::P(Lc).n_slices, Prec<T>::u(Lc), buf[cur], buf[dst], k, ctrl, h->stream));
        cur = dst;
    }
    if (kind == LV_CHEBY) return enqueue_cheby<T>(h, lv, Prec<T>::b(Lv), buf, &cur, k, post, ctrl);   // :57  (ends with cur == 0)
    if (jac) return enqueue_jacobi<T>(h, lv, Prec<T>::b(Lv), buf, &cur, k, post, ctrl);   // :57  (ends with cur == 0)
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
    if (!level_is_jacobi(h, lv)) return enqueue_gs<double>(h, lv, b, u, k, iters, ctrl);
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
static bool head_fusable(smg_hierarchy* h)
{
    static const int on = env_int("SMG_FUSE_HEAD", 1);
    if (!on || h->precision != 0 || h->n_levels < 2 || h->prof_on) return false;
    Level& L0 = h->lv[0];
    if (L0.gs_on_transpose) return false;
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
    if (h->precision == 1)   // mixed: the residual itself is the right-hand side of the fp32 correction cycle
        HIPCHK(launch_sell(SELL_RESID_BOTH, L0.dA.view, 0, L0.dA.view.n_slices, L0.u.p, L0.b.p, L0.r.p, k, h->d_ctrl.p, h->d_partials.p, &nb, h->stream));
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

static int ensure_graphs(smg_hierarchy* h)
{
    if (h->g_iter && h->g_k == h->k && h->g_pre == h->pre && h->g_post == h->post && h->g_prec == h->precision &&
        h->g_smoother == h->smoother && h->g_omega == h->omega && h->g_jmax == h->jacobi_max_rows && h->g_frac == h->cheby_fraction && h->g_head == h->head_fuse) return SMG_OK;
    drop_graphs(h);
    const int k = h->k;
    int rc = capture_graph(h, &h->g_iter, [&]() {
        int r = enqueue_residual_ss(h, k, true);
        if (r) return r;
        return enqueue_cycle_part(h, k, nullptr);
    });
    if (rc) return rc;
    rc = capture_split_graphs(h, h->g_sumsq_ptr ? h->g_sumsq_ptr : &h->d_ctrl.p->sumsq);
    if (rc) return rc;
    h->g_k = k; h->g_pre = h->pre; h->g_post = h->post; h->g_prec = h->precision;
    h->g_smoother = h->smoother; h->g_omega = h->omega; h->g_jmax = h->jacobi_max_rows; h->g_frac = h->cheby_fraction; h->g_head = h->head_fuse;
    return SMG_OK;
}

// hipStreamBeginCapture is not allowed on the legacy default stream (smg_hierarchy_set_stream(h, NULL)): eager launches there
static bool graphs_usable(const smg_hierarchy* h) { return h->use_graph && !h->prof_on && h->stream != nullptr; }

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

// ------------------------------------------------------------------------------------------------ solve
static int check_ready(const smg_hierarchy* h, const char* who)
{
    if (!h) return fail(SMG_ERR_INVALID, "%s: null handle", who);
    if (!h->precomputed) return fail(SMG_ERR_INVALID, "%s: call smg_precompute first", who);
    if (h->device < 0) return fail(SMG_ERR_NO_DEVICE, "%s: no HIP device", who);
    return SMG_OK;
}

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
    h->tol = o.tol; h->max_iter = o.max_iter; h->pre = o.pre; h->post = o.post; h->verbosity = o.verbosity;
    h->check_every = std::max(0, o.check_every); h->use_graph = o.use_graph;
    if (o.precision != 0 && o.precision != 1) return fail(SMG_ERR_INVALID, "precision must be 0 (fp64) or 1 (mixed)");
    h->precision = o.precision;
    if ((rc = smg_hierarchy_set_smoother(h, o.smoother, o.omega, o.jacobi_max_rows))) return rc;
    if ((rc = smg_hierarchy_set_chebyshev(h, o.cheby_fraction))) return rc;
    DeviceScope dsc(h->device);
    rc = ensure_work(h, k);
    if (rc) return rc;
    if (h->precision == 1 && (rc = ensure_fp32(h, k))) return rc;
    h->k = k;
    const int nk = (int)h->known.size();
    // stage host inputs
    const double *dR = RHS, *dZ = z0, *dK = known_val;
    int ldR = ld_rhs, ldZ = ld_z0, ldK = ld_kv;
    if (memspace == SMG_HOST) {
        HIPCHK(h->d_stage_rhs.ensure((size_t)n * k));
        HIPCHK(h->d_stage_z.ensure((size_t)n * k));
        HIPCHK(hipMemcpy2DAsync(h->d_stage_rhs.p, (size_t)n * 8, RHS, (size_t)ld_rhs * 8, (size_t)n * 8, k, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpy2DAsync(h->d_stage_z.p, (size_t)n * 8, z0, (size_t)ld_z0 * 8, (size_t)n * 8, k, hipMemcpyHostToDevice, h->stream));
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
    HIPCHK(launch_gather_in(L0.u.p, dZ, h->d_map0.p, L0.n, k, ldZ, h->stream));
    if (h->has_known) {
        // RHS_u = RHS(unknown) - Auk * known_val  (:316-318)
        const int nu = L0.n;
        HIPCHK(h->d_tmp_cm.ensure((size_t)nu * k));
        HIPCHK(launch_gather_cm(h->d_tmp_cm.p, dR, h->d_unknown.p, nu, k, ldR, nu, h->stream));
        HIPCHK(launch_csr_sub(nu, h->d_auk_ptr.p, h->d_auk_col.p, h->d_auk_val.p, dK, ldK, h->d_tmp_cm.p, nu, k, h->stream));
        HIPCHK(launch_gather_in(L0.b.p, h->d_tmp_cm.p, h->d_perm0.p, nu, k, nu, h->stream));
    } else {
        HIPCHK(launch_gather_in(L0.b.p, dR, h->d_map0.p, L0.n, k, ldR, h->stream));
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
    h->head_fuse = head_fusable(h);   // latched: both halves of every iteration of this solve follow it
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
    const int n = h->n_full, k = h->k;
    if (!z || ld_z < n) return fail(SMG_ERR_INVALID, "smg_solve_end: bad z / ld_z");
    Level& L0 = h->lv[0];
    double* dz = z;
    int ldz = ld_z;
    if (memspace == SMG_HOST) {
        HIPCHK(h->d_stage_z.ensure((size_t)n * k));
        dz = h->d_stage_z.p; ldz = n;
    }
    // z(unknown) = z_u ; z(known) = known_val  (:353-355)
    HIPCHK(launch_scatter_out(dz, L0.u.p, h->d_map0.p, L0.n, k, ldz, h->stream));
    if (h->has_known)
        HIPCHK(launch_scatter_cm(dz, h->cur_kv, h->d_known.p, (int)h->known.size(), k, h->cur_ld_kv, ldz, h->stream));
    if (memspace == SMG_HOST)
        HIPCHK(hipMemcpy2DAsync(z, (size_t)ld_z * 8, dz, (size_t)n * 8, (size_t)n * 8, k, hipMemcpyDeviceToHost, h->stream));
    static thread_local Ctrl hc;
    static thread_local std::vector<double> his;
    // the history can hold at most one entry per enqueued iteration: fetched together with the control block, one synchronisation
    const int cap = (int)std::min<size_t>(h->d_rhis.n, (size_t)std::max(std::min(h->iters_enqueued, std::max(h->max_iter, 1)), 1));
    his.resize((size_t)cap);
    HIPCHK(hipMemcpyAsync(&hc, h->d_ctrl.p, sizeof(Ctrl), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(his.data(), h->d_rhis.p, (size_t)cap * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    const int cnt = std::max(0, std::min(std::min(hc.n_his, hc.his_cap), cap));
    h->in_solve = false;
    prof_collect(h);
    if (r_his) for (int i = 0; i < cnt; i++) r_his[i] = his[i];
    if (n_his) *n_his = cnt;
    const double last = cnt > 0 ? his[cnt - 1] : HUGE_VAL;
    if (converged) *converged = (last > h->tol) ? 0 : 1;  // :131-134 / :357-360
    if (h->verbosity > 0) {
        for (int i = 0; i < cnt; i++) std::printf("MG iteration: %d, residual: %g\n", i, his[i]);  // :111
        if (cnt) std::printf("residual norm: %g\n", his[cnt - 1]);                                    // :127
    }
    if (hc.status != 0) return fail(SMG_ERR_NONFINITE, "non-finite residual at iteration %d", cnt - 1);
    return SMG_OK;
}

extern "C" int smg_solve(smg_hierarchy* h, const double* RHS, int ld_rhs, const double* known_val, int ld_kv,
                         const double* z0, int ld_z0, int k, int memspace, const smg_solve_opts* opts, double* z, int ld_z,
                         double* r_his, int* n_his, int* converged)
{
    int rc = smg_solve_begin(h, RHS, ld_rhs, known_val, ld_kv, z0, ld_z0, k, memspace, opts);
    if (rc) return rc;
    // for (iter < maxIter) { residual; push; if (residual < tol) break; V-cycle }   (:108-125 / :330-347)
    // The break happens on the device; the host only decides how many iterations to enqueue before it looks at the flag again.
    // check_every >= 1: that many.  check_every == 0 (default): adaptive -- from the two most recent residuals the host extrapolates
    // how many more cycles the tolerance needs and enqueues all but the last of them before the next look (the results do not depend
    // on this: an iteration enqueued after the break stores nothing).
    int it = 0;
    int chunk_next = 1;
    while (it < h->max_iter) {
        const int want = h->check_every > 0 ? h->check_every : chunk_next;
        const int chunk = std::min(want, h->max_iter - it);
        for (int c = 0; c < chunk; c++) {
            rc = enqueue_outer_iteration(h);
            if (rc) { h->in_solve = false; return rc; }
        }
        it += chunk;
        if (it < h->max_iter) {
            Ctrl hc;
            hipError_t e = hipMemcpyAsync(&hc, h->d_ctrl.p, sizeof(Ctrl), hipMemcpyDeviceToHost, h->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
            if (e != hipSuccess) { h->in_solve = false; return fail(SMG_ERR_HIP, "smg_solve: %s", hipGetErrorString(e)); }
            if (hc.done) break;
            chunk_next = 1;
            if (h->check_every == 0 && hc.n_his >= 2 && hc.r_last > 0.0 && hc.r_last < hc.r_prev && h->tol > 0.0 && hc.r_last > h->tol) {
                const double need = std::ceil(std::log(h->tol / hc.r_last) / std::log(hc.r_last / hc.r_prev));   // more residuals until < tol
                if (need > 2.0) chunk_next = (int)std::min(need - 1.0, 64.0);
            }
        }
    }
    return smg_solve_end(h, z, ld_z, memspace, r_his, n_his, converged);
}

extern "C" int smg_raw_outer_iteration(smg_hierarchy* h, int n_iter)
{
    if (!h || !h->in_solve) return fail(SMG_ERR_INVALID, "smg_raw_outer_iteration: call smg_solve_begin first");
    DeviceScope dsc(h->device);
    for (int i = 0; i < n_iter; i++) {
        int rc = enqueue_outer_iteration(h);
        if (rc) return rc;
    }
    return SMG_OK;
}

static int piece_prolog(smg_hierarchy* h, int lv, int k, const char* who, bool need_coarser);

extern "C" int smg_bench_vcycle(smg_hierarchy* h, int lv, int k, int pre, int post, int reps, double* us_per_cycle)
{
    int rc = piece_prolog(h, lv, k, "smg_bench_vcycle", false);
    if (rc) return rc;
    DeviceScope dsc(h->device);
    if (reps < 1 || !us_per_cycle) return fail(SMG_ERR_INVALID, "smg_bench_vcycle: bad arguments");
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
    return SMG_OK;
}

extern "C" int smg_bench_relax(smg_hierarchy* h, int lv, int k, int sweeps, int reps, double* us_per_call)
{
    int rc = piece_prolog(h, lv, k, "smg_bench_relax", true);
    if (rc) return rc;
    DeviceScope dsc(h->device);
    if (reps < 1 || sweeps < 1 || !us_per_call) return fail(SMG_ERR_INVALID, "smg_bench_relax: bad arguments");
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
    return SMG_OK;
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
    return SMG_OK;
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
    HIPCHK(launch_sell(SELL_AX, Lv.dA.view, 0, Lv.dA.view.n_slices, Lv.u.p, nullptr, Lv.r.p, k, nullptr, nullptr, nullptr, h->stream));
    return get_block(h, lv, Lv.r.p, k, Au);
}

extern "C" int smg_restrict(smg_hierarchy* h, int lv, const double* x, int k, double* Rx)
{
    int rc = piece_prolog(h, lv, k, "smg_restrict", true);
    if (rc) return rc;
    DeviceScope dsc(h->device);
    Level &Lv = h->lv[lv], &Lc = h->lv[lv + 1];
    if ((rc = put_block(h, lv, x, k, Lv.r.p))) return rc;
    HIPCHK(launch_sell(SELL_AX, Lc.dPT.view, 0, Lc.dPT.view.n_slices, Lv.r.p, nullptr, Lc.b.p, k, nullptr, nullptr, nullptr, h->stream));
    return get_block(h, lv + 1, Lc.b.p, k, Rx);
}

extern "C" int smg_prolong(smg_hierarchy* h, int lv, const double* x, int k, double* Px)
{
    int rc = piece_prolog(h, lv, k, "smg_prolong", true);
    if (rc) return rc;
    DeviceScope dsc(h->device);
    Level &Lv = h->lv[lv], &Lc = h->lv[lv + 1];
    if ((rc = put_block(h, lv + 1, x, k, Lc.u.p))) return rc;
    HIPCHK(launch_sell(SELL_AX, Lc.dP.view, 0, Lc.dP.view.n_slices, Lc.u.p, nullptr, Lv.r.p, k, nullptr, nullptr, nullptr, h->stream));
    return get_block(h, lv, Lv.r.p, k, Px);
}

extern "C" int smg_relax(smg_hierarchy* h, int lv, const double* B, int k, int iters, double* u)
{
    int rc = piece_prolog(h, lv, k, "smg_relax", true);
    if (rc) return rc;
    DeviceScope dsc(h->device);
    Level& Lv = h->lv[lv];
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
    HIPCHK(launch_dense_gemv_add(h->d_Ainv.p, h->nc, h->nc_pad, Lv.b.p, Lv.u.p, k, nullptr, h->stream, h->d_sympart.p));
    return get_block(h, lv, Lv.u.p, k, u);
}

extern "C" int smg_vcycle(smg_hierarchy* h, const double* B, int pre, int post, int lv, double* u, int k)
{
    int rc = piece_prolog(h, lv, k, "smg_vcycle", false);
    if (rc) return rc;
    DeviceScope dsc(h->device);
    Level& Lv = h->lv[lv];
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
    HIPCHK(launch_sell(SELL_RESID_SS, Lv.dA.view, 0, Lv.dA.view.n_slices, Lv.u.p, Lv.b.p, nullptr, k, nullptr, h->d_partials.p, &nb, h->stream));
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
    return enqueue_relax(h, lv, b, u, k, iters, nullptr);
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
    if (stored) *stored = S->stored;
    if (padded) *padded = S->used;   // slots read per pass (the allocation may be larger: fixed-stride panels)
    if (n_slices) *n_slices = S->view.n_slices;
    return SMG_OK;
}

extern "C" long smg_level_spmv_bytes(const smg_hierarchy* h, int lv, int k)
{
    if (!h || lv < 0 || lv >= h->n_levels) return -1;
    const Csr& A = h->lv[lv].A;
    return 12L * A.nnz() + 4L * (A.nr + 1) + 16L * A.nr * k;
}

// Algorithmic bytes of one outer iteration: per smoothed level
//   (pre+post) GS sweeps: 12 nnz + 4(n+1) + 24 n k [b, u read, u write]   (the reference also reads A_diag: +8n; the
//                          HIP kernel takes the diagonal from the row, so it is not counted)
//   residual:             12 nnz + 4(n+1) + 24 n k
//   restrict:             12 nnzPT + 4(nc+1) + 8 n k + 8 nc k
//   prolong-add:          12 nnzP + 4(n+1) + 8 nc k + 16 n k
//   + coarsest dense solve 8 nc^2 + 24 nc k, + outer residual 12 nnz0 + 4(n0+1) + 16 n0 k.
extern "C" long smg_vcycle_bytes(const smg_hierarchy* h, int k, int pre, int post)
{
    if (!h || !h->precomputed) return -1;
    long tot = 0;
    const int L = h->n_levels;
    for (int lv = 0; lv < L - 1; lv++) {
        const Level &Lv = h->lv[lv], &Lc = h->lv[lv + 1];
        const long n = Lv.n, nc = Lc.n, nnz = Lv.A.nnz(), nnzP = Lc.P.nnz();
        const long sweep = 12 * nnz + 4 * (n + 1) + 24 * n * k;
        tot += (long)(pre + post) * sweep;
        tot += 12 * nnz + 4 * (n + 1) + 24 * n * k;
        tot += 12 * nnzP + 4 * (nc + 1) + 8 * n * k + 8 * nc * k;
        tot += 12 * nnzP + 4 * (n + 1) + 8 * nc * k + 16 * n * k;
    }
    const long nc = h->lv[L - 1].n;
    tot += 8 * nc * nc + 24 * nc * k;
    tot += 12 * h->lv[0].A.nnz() + 4L * (h->lv[0].n + 1) + 16L * h->lv[0].n * k;
    return tot;
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
static Mesh wrap_mesh(const double* V, int nV, const int* F, int nF)
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
    smg_hierarchy* h = smg_hierarchy_create(nLvs);
    if (!h) return SMG_ERR_ALLOC;
    Mesh m = wrap_mesh(V, nV, F, nF);
    h->lv[0].V = m.V; h->lv[0].F = m.F;   // src/mg_precompute.cpp:46-47
    int rc = build_decimated_levels(h, 1, m, nLvs - 1, ratio, dec_type, (int)std::lround(10.0 * absorption_cap), keep_log);
    if (rc) { smg_hierarchy_destroy(h); return rc; }
    *out = h;
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
        }
        query_fine_to_coarse(*Lv.dec_log, n, face, bary, out_face, out_bary);
        return (int)SMG_OK;
    });
}

static int smg_mg_precompute_block_impl(const double* V, int nV, const int* F, int nF, float ratio, int nVCoarsest, int dec_type,
                                       smg_hierarchy** out)
{
    int rc = smg_mg_precompute(V, nV, F, nF, ratio, nVCoarsest, dec_type, out);
    if (rc) return rc;
    smg_hierarchy* h = *out;
    for (int lv = 1; lv < h->n_levels; lv++) {
        const Csr& P = h->lv[lv].P_full;
        Csr B;
        B.nr = 3 * P.nr; B.nc = 3 * P.nc;
        B.ptr.resize((size_t)B.nr + 1);
        B.col.resize((size_t)3 * P.nnz()); B.val.resize((size_t)3 * P.nnz());
        int q = 0;
        for (int r = 0; r < P.nr; r++)
            for (int d = 0; d < 3; d++) {   // row 3r+d holds P(r,c) at column 3c+d  (src/get_prolong.cpp:108-110)
                B.ptr[3 * r + d] = q;
                for (int p = P.ptr[r]; p < P.ptr[r + 1]; p++) { B.col[q] = 3 * P.col[p] + d; B.val[q] = P.val[p]; q++; }
            }
        B.ptr[B.nr] = q;
        set_prolong(h, lv, std::move(B));
    }
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
    FILE* f = std::fopen(path, "rb");
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
    smg_hierarchy* h = ok ? smg_hierarchy_create(L) : nullptr;
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
    std::fclose(f);
    if (!ok || !h) { if (h) smg_hierarchy_destroy(h); return fail(SMG_ERR_IO, "'%s' is not a valid hierarchy file (%s)", path, why); }
    *out = h;
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
    smg_hierarchy* h = smg_hierarchy_create(1 + n_sub + extra);
    if (!h) return SMG_ERR_ALLOC;
    h->lv[0].V = fine.V; h->lv[0].F = fine.F;
    for (int l = 1; l <= n_sub; l++) set_prolong(h, l, std::move(Ps[l - 1]));
    h->lv[n_sub].V = base.V; h->lv[n_sub].F = base.F;
    int rc = build_decimated_levels(h, n_sub + 1, base, extra, ratio, SMG_DEC_MIDPOINT);
    if (rc) { smg_hierarchy_destroy(h); return rc; }
    if (V_out) std::copy(fine.V.begin(), fine.V.end(), V_out);
    if (F_out) std::copy(fine.F.begin(), fine.F.end(), F_out);
    *out = h;
    return SMG_OK;
}

extern "C" int smg_mg_precompute_subdiv(const double* V, int nV, const int* F, int nF, int n_sub, float ratio, int nVCoarsest,
                                        int n_extra_levels, smg_hierarchy** out, double* V_out, int* F_out)
{
    return guarded("smg_mg_precompute_subdiv", [&]() { return smg_mg_precompute_subdiv_impl(V, nV, F, nF, n_sub, ratio, nVCoarsest, n_extra_levels, out, V_out, F_out); });
}
