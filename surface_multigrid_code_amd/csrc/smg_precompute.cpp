// smg_precompute.cpp -- min_quad_with_fixed_mg_precompute (reference src/min_quad_with_fixed_mg.cpp:3-51, :137-257) behind smg_precompute:
// the reference's sparse algebra on the host (caller numbering, bit-compatible accumulation order), the device images (colour-major
// numbering, SELL panels, coarse inverse), the value-only re-precompute on the device and the operator assembly.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <functional>
#include <memory>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "smg_internal.hpp"

using namespace smg;

// ------------------------------------------------------------------------------------------------ precompute
// A == A^T, bit for bit?  (row-parallel: every entry looks its mirror image up by bisection; no transpose is materialised.  The levels whose
// panels the device fills are tested there, launch_bit_symmetric.)
static bool bit_symmetric(const Csr& A)
{
    if (A.nr != A.nc) return false;
    std::atomic<int> any{0};
    parallel_for(A.nr, 4096, [&](long r0, long r1) {
        for (long i = r0; i < r1 && !any.load(std::memory_order_relaxed); i++)
            for (int p = A.ptr[(size_t)i]; p < A.ptr[(size_t)i + 1]; p++) {
                const int j = A.col[(size_t)p];
                const int* b = A.col.data() + A.ptr[(size_t)j];
                const int* e = A.col.data() + A.ptr[(size_t)j + 1];
                const int* q = std::lower_bound(b, e, (int)i);
                if (q == e || *q != (int)i || std::memcmp(&A.val[(size_t)(q - A.col.data())], &A.val[(size_t)p], sizeof(double)) != 0) { any.store(1); break; }
            }
    });
    return any.load() == 0;
}
// FNV-1a over an int array, hashed in fixed blocks of 64 Ki entries -- the blocks concurrently on the host threads, the block hashes chained in
// order: the value does not depend on the number of threads (a time step's re-precompute hashes the 8 M pattern entries of a 1 M-vertex mesh
// before anything else: 6.5 ms as one sequential chain; the 63 M of the block benchmark: 54 ms)
static uint64_t fnv_mix(uint64_t key, const int* p, size_t cnt)
{
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

// key of a level's numbering: (rows, smoothed?, block size, ptr, col)
static uint64_t pattern_key_arrays(int nr, bool smoothed, int bs, const int* ptr, const int* col)
{
    uint64_t key = 1469598103934665603ull;
    const int hdr[3] = {nr, smoothed ? 1 : 0, bs};
    key = fnv_mix(key, hdr, 3);
    key = fnv_mix(key, ptr, (size_t)nr + 1);
    key = fnv_mix(key, col, (size_t)ptr[nr]);
    return key;
}
// Work on level 0 that needs nothing but the caller's arrays, started by smg_precompute before anything else when they ARE level 0's
// matrix (canonical rows, no constraints, scalar numbering): the locality order -- the longest sequential piece of the whole precompute
// (a Cuthill-McKee search over all rows).
static bool use_rcm_order()
{
    static const bool v = [] { const char* e = std::getenv("SMG_ORDER"); return !(e && std::string(e) == "induced"); }();
    return v;
}
struct Early0 {
    std::thread rcm_t;
    bool rcm_started = false;
    std::vector<int> rcm;      // empty after the join: the level's pattern is the one its present numbering was built on
    double rcm_ms = 0.0;
    // Level 0's colours, when it has none to inherit (its prolongation is no subdivision operator: known from the start): the from-scratch
    // colouring of a million rows is the longest sequential piece of such a precompute (0.8 s) and needs nothing but level 0's pattern and its
    // locality order -- colour_t joins rcm_t, then colours, beside everything else the host half does.  wait() is the ONLY way to join either.
    std::thread colour_t;
    std::vector<int> colours;
    double colour_ms = 0.0;
    void wait() { if (colour_t.joinable()) colour_t.join(); if (rcm_t.joinable()) rcm_t.join(); }
    ~Early0() { wait(); }
};
// The panels of a big level's A can be filled on the device straight from the caller's arrays and the permutation (launch_sell_fill):
// the host then skips the permuted copy and the transposition test of 7 M entries, and ships 85 MB instead of 145 MB of padded
// panels (C3 level 0: ~ 100 ms of the first precompute).  Conditions: scalar path, a level that is smoothed, at least
// SMG_DEVICE_FILL_MIN rows (default 200 000; smaller levels are done before it would pay), A bit-symmetric (no A^T image needed).
static bool device_fill_candidate(const smg_hierarchy* h, int lv)
{
    static const int on = env_int("SMG_DEVICE_FILL", 1), min_rows = env_int("SMG_DEVICE_FILL_MIN", 200000);
    return on && lv < h->n_levels - 1 && h->lv[lv].A.nr >= min_rows;       // (block hierarchies too: launch_bsr3_fill)
}
static bool device_fill_rows(const smg_hierarchy* h, int n_rows)
{
    static const int on = env_int("SMG_DEVICE_FILL", 1), min_rows = env_int("SMG_DEVICE_FILL_MIN", 200000);
    return on && h->bs == 1 && n_rows >= min_rows;
}
// A matrix's CSR arrays on the device (caller numbering), for the fills below; `sent`: they are there already (smg_hierarchy::EarlyUpload).
struct DeviceCsr {
    DevBuf<int> d_ptr, d_col;
    DevBuf<double> d_val;
    const int* ptr = nullptr;
    const int* col = nullptr;
    const double* val = nullptr;
    hipError_t put(const Csr& M, const smg_hierarchy::EarlyUpload* sent)
    {
        if (sent && sent->valid && sent->ptr.n == M.ptr.size() && sent->col.n == M.col.size()) { ptr = sent->ptr.p; col = sent->col.p; val = sent->val.p; return hipSuccess; }
        hipError_t e = d_ptr.upload(M.ptr);
        if (e == hipSuccess) e = d_col.upload(M.col);
        if (e == hipSuccess) e = d_val.upload(M.val);
        ptr = d_ptr.p; col = d_col.p; val = d_val.p;
        return e;
    }
};
// SELL image of B(i, j) = M(rperm[i], cperm[j]) (ciperm = the inverse of cperm) -- or of B^T when `transposed` (M square, structurally
// symmetric, rperm == cperm) -- built on the device from M's arrays: layout from the row lengths on the host, panels by launch_sell_fill.
// Called from the precompute's worker threads (own stream).
static hipError_t device_fill_sell(SellBuf& dst, const Csr& M, const DeviceCsr& D, const std::vector<int>& rperm, const std::vector<int>& ciperm,
                                   const std::vector<int>* breaks, bool region, hipStream_t st2, bool transposed = false, int pitch_policy = -1)
{
    static const bool tm_on = env_int("SMG_TIMING", 0) >= 2;
    auto t_last = std::chrono::steady_clock::now();
    std::string log;
    auto lap = [&](const char* what) {
        if (!tm_on) return;
        const auto t = std::chrono::steady_clock::now();
        char b[64]; std::snprintf(b, sizeof b, " %s %.1f", what, 1e3 * std::chrono::duration<double>(t - t_last).count());
        log += b; t_last = t;
    };
    std::vector<int> row_len((size_t)M.nr);
    parallel_for(M.nr, 1 << 16, [&](long r0, long r1) {
        for (long r = r0; r < r1; r++) { const int o = rperm[(size_t)r]; row_len[(size_t)r] = M.ptr[(size_t)o + 1] - M.ptr[(size_t)o]; }
    });
    lap("row_len");
    Sell S = sell_layout(row_len, M.nc, M.nnz(), breaks, SELL_C, region, pitch_policy);
    lap("layout");
    hipError_t e = dst.upload(S);
    lap("panels");
    DevBuf<int> d_perm, d_iperm;
    if (e == hipSuccess) e = d_perm.upload(rperm);
    if (e == hipSuccess) e = d_iperm.upload(ciperm);
    lap("perms");
    // (st2: one of the handle's auxiliary streams -- the other tasks' uploads and fills go on beside this one)
    if (e == hipSuccess) e = launch_sell_fill(D.ptr, D.col, D.val, d_perm.p, d_iperm.p, dst.view, (size_t)dst.padded, st2, transposed);
    // coloured square images: where each row's diagonal sits (SellBuf::diag_slot / n_first / n_all: what lets the restriction launch of the
    // finer level produce this level's first colour, as SellBuf::upload works out for the images built on the host)
    dst.n_first = 0; dst.n_all = 0;
    const bool want_slots = M.nr == M.nc && S.color_slice_ptr.size() >= 2 && M.nr > 0;
    DevBuf<int> d_missing;
    int first_missing = M.nr;
    if (e == hipSuccess && want_slots) {
        e = dst.diag_slot.alloc((size_t)M.nr);
        if (e == hipSuccess) e = d_missing.alloc(1);
        if (e == hipSuccess) e = hipMemcpyAsync(d_missing.p, &first_missing, sizeof(int), hipMemcpyHostToDevice, st2);
        if (e == hipSuccess) e = launch_sell_diag_slots(dst.view, dst.diag_slot.p, d_missing.p, st2);
        if (e == hipSuccess) e = hipMemcpyAsync(&first_missing, d_missing.p, sizeof(int), hipMemcpyDeviceToHost, st2);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st2);
    if (e == hipSuccess && want_slots) {
        const int s1 = S.color_slice_ptr.size() >= 3 ? S.color_slice_ptr[1] : S.n_slices;
        const int nf = S.slice_row[(size_t)s1];
        if (first_missing >= nf) {
            if (S.color_slice_ptr.size() >= 3) dst.n_first = nf;
            if (first_missing == M.nr) dst.n_all = M.nr;
        } else dst.diag_slot.release();
    }
    lap("fill");
    if (tm_on) std::fprintf(stderr, "[smg timing] device:     fill of %d x %d (ms):%s\n", M.nr, M.nc, log.c_str());
    return e;
}
static hipError_t device_fill_sell(SellBuf& dst, const Csr& M, const std::vector<int>& rperm, const std::vector<int>& ciperm, const std::vector<int>* breaks, bool region,
                                   hipStream_t st2, int pitch_policy)
{
    DeviceCsr D;
    hipError_t e = D.put(M, nullptr);
    return e == hipSuccess ? device_fill_sell(dst, M, D, rperm, ciperm, breaks, region, st2, false, pitch_policy) : e;
}
int smg::ensure_P_int(smg_hierarchy* h, int lv)
{
    if (!h || lv < 1 || lv >= h->n_levels) return SMG_OK;
    Level& Lv = h->lv[lv];
    const Level& Lf = h->lv[lv - 1];
    if ((int)Lf.ord.perm.size() != Lv.P.nr || (int)Lv.ord.perm.size() != Lv.P.nc) return SMG_OK;
    if (Lv.P_int.nr != Lv.P.nr || Lv.P_int.nnz() != Lv.P.nnz()) Lv.P_int = permute(Lv.P, Lf.ord.perm, Lv.ord.perm);
    if (Lv.PT_int.nr != Lv.PT.nr || Lv.PT_int.nnz() != Lv.PT.nnz()) Lv.PT_int = permute(Lv.PT, Lv.ord.perm, Lf.ord.perm);
    return SMG_OK;
}
int smg::ensure_A_int(smg_hierarchy* h, int lv)
{
    if (!h || lv < 0 || lv >= h->n_levels) return fail(SMG_ERR_INVALID, "bad level");
    Level& Lv = h->lv[lv];
    if (Lv.A_int.nr == Lv.A.nr && (long)Lv.A_int.nnz() == Lv.A.nnz()) return SMG_OK;
    if ((int)Lv.ord.perm.size() != Lv.A.nr) return SMG_OK;        // no numbering yet: nothing to express
    if (h->host_stale) { int rc = refresh_host_values(h); if (rc) return rc; }
    Lv.A_int = permute(Lv.A, Lv.ord.perm, Lv.ord.perm, &Lv.A_int_src);
    return SMG_OK;
}

// Where the two halves of a first precompute meet.  The host half posts what has become final -- the coarsest matrix, then the
// numbering of every level, coarse to fine -- and the device half (the calling thread) builds a level's images as soon as its numbering and
// that of the next coarser level exist, while the host half is still working on the finer ones.
struct Handoff {
    std::mutex m;
    std::condition_variable cv;
    bool coarse_ready = false, host_over = false;
    std::vector<char> level_ready;
    explicit Handoff(int L) : level_ready((size_t)L, 0) {}
    void post_coarse() { { std::lock_guard<std::mutex> g(m); coarse_ready = true; } cv.notify_all(); }
    void post_level(int lv) { { std::lock_guard<std::mutex> g(m); level_ready[(size_t)lv] = 1; } cv.notify_all(); }
    void post_over() { { std::lock_guard<std::mutex> g(m); host_over = true; } cv.notify_all(); }
    // false: the host half ended without getting there
    bool wait_coarse() { std::unique_lock<std::mutex> g(m); cv.wait(g, [&] { return coarse_ready || host_over; }); return coarse_ready; }
    bool wait_level(int lv) { std::unique_lock<std::mutex> g(m); cv.wait(g, [&] { return level_ready[(size_t)lv] || host_over; }); return level_ready[(size_t)lv] != 0; }
    bool level_posted(int lv) { std::lock_guard<std::mutex> g(m); return level_ready[(size_t)lv] != 0 || host_over; }      // (would wait_level return at once?)
};

// Host half: the reference's sparse algebra, in the caller's numbering, bit-compatible accumulation order.
static int precompute_host(smg_hierarchy* h, Csr&& A, const int* known, int n_known, Early0& e0, Handoff& hand)
{
    const int n = A.nr;
    const int L = h->n_levels;
    h->n_full = n;
    h->has_known = (known != nullptr && n_known > 0);
    h->known.clear(); h->unknown.clear();
    for (int lv = 1; lv < L; lv++) {
        if (h->lv[lv].P_full.empty()) return fail(SMG_ERR_INVALID, "level %d has no prolongation (smg_level_set_prolong)", lv);
        h->lv[lv].P = copy_of(h->lv[lv].P_full);  // always restart from P_full (see smg.h)
    }
    if (L > 1 && h->lv[1].P_full.nr != n)
        return fail(SMG_ERR_INVALID, "A is %d x %d but P_1 has %d rows", n, n, h->lv[1].P_full.nr);
    h->nnz_input = (int)A.nnz();
    StageTimer tm;
    if (!h->has_known) {
        // reference src/min_quad_with_fixed_mg.cpp:17-22
        h->lhs_src.clear();      // (identity: written out when the value-only path first needs it, build_recipes)
        h->auk_src.clear();
        h->lv[0].A = std::move(A);
        h->Auk = Csr();
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
    }
    tm.lap("host: constraint slices");
    // ---- block (3-DOF) variant?  (smg_bsr3.hpp)  Every prolongation must be Pv (x) I_3 (mg_precompute_block builds them so) and --
    // unless the caller insists -- the 3 x 3 blocks of A must be at least half full (a system like kron(S, I_3) is three scalar
    // problems: the scalar kernels with k = 3 columns serve it better).  Constraints keep the structure when they are VERTEX-wise --
    // all three degrees of freedom of a pinned vertex known, which is what pinning a vertex means: `unknown` then consists of whole
    // triples, A(unknown, unknown) is a block matrix again, P_full(unknown, :) = Pv(unknown vertices, :) (x) I_3, and the column-drop
    // cascade (:190-219) removes the three columns of a coarse vertex together (they hold the same entries) -- so the reference's
    // slicing, run on the scalar matrices above as always, hands the block path operators it can factor like any others.  Constraints
    // on single degrees of freedom break the structure: scalar path (the factoring below fails).
    h->bs = 1;
    for (int lv = 0; lv < L; lv++) h->lv[lv].vpat = Csr();     // bs == 3: the n_v x n_v pattern of the blocks of A_l (Level::vpat), the graph the numbering is built on
    bool triples = h->lv[0].A.nr % 3 == 0 && n % 3 == 0;
    if (triples && h->has_known)
        for (size_t u = 0; u < h->unknown.size() && triples; u += 3)
            triples = h->unknown[u] % 3 == 0 && h->unknown[u + 1] == h->unknown[u] + 1 && h->unknown[u + 2] == h->unknown[u] + 2;
    if (h->block_mode != 0 && triples && L >= 2) {
        bool ok = true;
        for (int lv = 1; lv < L && ok; lv++) ok = kron3_factor(h->lv[lv].P, h->lv[lv].Pv);
        if (ok) {
            h->lv[0].vpat = block_pattern3(h->lv[0].A);
            const double fill = (double)h->lv[0].A.nnz() / (9.0 * (double)std::max<long>(h->lv[0].vpat.nnz(), 1));
            if (h->block_mode == 3 || fill >= 0.5) h->bs = 3;
        }
        tm.lap("host: block structure (P = Pv (x) I_3, block pattern of A_0)");
    }
    if (h->block_mode == 3 && h->bs != 3)
        return fail(SMG_ERR_INVALID, "smg_precompute: block mode was required (smg_hierarchy_set_block_mode) but %s",
                    !triples ? (h->has_known ? "the constraints do not cover whole vertices (all three degrees of freedom of each pinned vertex)" : "the system is not a 3-DOF system")
                             : L < 2 ? "the hierarchy has one level" : "a prolongation is not of the form Pv (x) I_3");
    const bool blk = h->bs == 3;
    if (!blk) for (int lv = 0; lv < L; lv++) { h->lv[lv].Pv = Csr(); h->lv[lv].PTv = Csr(); h->lv[lv].vord = Ordering(); }
    if (!blk) h->lv[0].vpat = Csr();
    auto graph = [&](int lv) -> const Csr& { return blk ? h->lv[lv].vpat : h->lv[lv].A; };                 // what a level's numbering is built on
    auto order_of = [&](int lv) -> Ordering& { return blk ? h->lv[lv].vord : h->lv[lv].ord; };      // ... and where it goes
    auto pattern_key = [&](int lv) {
        const Csr& M = h->lv[lv].A;
        return pattern_key_arrays(M.nr, lv < L - 1, h->bs, M.ptr.data(), M.col.data());
    };
    const bool use_rcm = use_rcm_order();
    struct E0Join { Early0& e; ~E0Join() { e.wait(); } } e0_join{e0};   // (they may read this frame)
    // Level 0's locality order (Early0) runs on a thread of its own, beside the Galerkin products: started by smg_precompute on the
    // caller's arrays where it could, else here.
    if (e0.rcm_started && (blk || h->has_known)) {       // (cannot happen: the caller starts them only for scalar, unconstrained systems)
        e0.wait();
        e0.rcm_started = false; e0.rcm.clear();
    }
    uint64_t key0 = 0;
    if (L >= 3 && use_rcm && host_threads() > 1) {
        key0 = pattern_key(0);
        tm.lap("host:   pattern hash of level 0");
        const Level& L0 = h->lv[0];
        if (!e0.rcm_started && !(key0 == L0.ord_key && (int)L0.ord.perm.size() == L0.A.nr)) {
            e0.rcm_started = true;
            e0.rcm_t = std::thread([&] {
                const auto t0 = std::chrono::steady_clock::now();
                e0.rcm = rcm_order(graph(0));
                e0.rcm_ms = 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            });
        }
    }
    // A level can inherit colours only through a prolongation whose rows have one or two entries (subdivision_colors, preset_from_coarsest): where
    // the next level's P has a wider row -- the reference's own mg_precompute: three per row -- the level is coloured from scratch whatever the
    // coarser levels' colours turn out to be, so its colouring need not wait for them.
    auto may_inherit = [&](int lv) {
        if (lv + 1 >= L) return false;
        const Csr& Pn = blk ? h->lv[lv + 1].Pv : h->lv[lv + 1].P;
        for (int i = 0; i < Pn.nr; i++) if (Pn.ptr[i + 1] - Pn.ptr[i] > 2) return false;
        return true;
    };
    if (e0.rcm_started && use_rcm && L >= 3 && host_threads() > 1 && !may_inherit(0)) {
        e0.colour_t = std::thread([&] {
            if (e0.rcm_t.joinable()) e0.rcm_t.join();
            if ((int)e0.rcm.size() != graph(0).nr) return;      // (pattern unchanged since the last precompute: nothing to number)
            const auto t0 = std::chrono::steady_clock::now();
            e0.colours = colours_for_ordering(graph(0), e0.rcm);
            e0.colour_ms = 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        });
    }
    {
        std::vector<std::function<void()>> tasks;
        for (int lv = 1; lv < L; lv++) {
            // (block hierarchies: P = Pv (x) I_3, so PT = PTv (x) I_3 -- the same entries in the same order as the transposition of three times as many)
            if (blk) tasks.push_back([h, lv, &tm] {
                const auto t0 = std::chrono::steady_clock::now();
                h->lv[lv].PTv = transpose(h->lv[lv].Pv);
                const auto t1 = std::chrono::steady_clock::now();
                h->lv[lv].PT = kron3(h->lv[lv].PTv);
                if (tm.on) std::fprintf(stderr, "[smg timing] host:   (level %d: PTv %.1f ms, PT = PTv (x) I_3 %.1f ms)\n", lv, 1e3 * std::chrono::duration<double>(t1 - t0).count(),
                                        1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count());
            });
            else tasks.push_back([h, lv] { h->lv[lv].PT = transpose(h->lv[lv].P); });      // :226
        }
        parallel_tasks(tasks);
    }
    tm.lap("host: transposes of P");
    std::thread cl_thread;
    std::atomic<int> coarsest_A{0};      // 1: the coarsest level's matrix exists (the colouring thread may look at it), 2: it never will (error path)
    struct ClJoin { std::thread& t; std::atomic<int>& st; ~ClJoin() { int z = 0; st.compare_exchange_strong(z, 2); if (t.joinable()) t.join(); } } cl_join{cl_thread, coarsest_A};     // (it reads this frame)
    bool cl_started = false, cl_fresh = false;
    uint64_t cl_key = 0;
    double cl_ms = 0.0;
    // The coarsest smoothed level inherits a 4-colouring from the coarsest level when it is a mid-point subdivision of it (every row of P has one or two
    // entries): the coarsest level's small graph gets the all-out search, its colours are handed down exactly like between the finer levels.  The search
    // on the level itself may end with five colours, which every finer level then pays for (no inheritance: a from-scratch colouring of a million rows
    // took 1.3 s and the cycle 5 launches per sweep instead of 4 on the C3 mesh stopped at 15 804 coarse unknowns).  Used by BOTH ways the level gets its
    // numbering -- the side thread below and the task of the from-scratch path -- so that the numbering depends on the matrix alone, not on how many host
    // threads there are.
    auto looks_subdivided = [&](int lv) {
        if (blk || lv != L - 2 || L < 2) return false;
        const Level& Lw = h->lv[lv];
        const Level& Lc = h->lv[lv + 1];
        bool subdiv = Lc.P.nr == Lw.A.nr && Lc.P.nc <= 65536;
        for (int i = 0; i < Lc.P.nr && subdiv; i++) if (Lc.P.ptr[i + 1] - Lc.P.ptr[i] > 2) subdiv = false;
        return subdiv;
    };
    auto preset_from_coarsest = [&](int lv, std::vector<int>& inherited) {      // needs the coarsest level's matrix
        const Level& Lw = h->lv[lv];
        const Level& Lc = h->lv[lv + 1];
        const Ordering oc = make_ordering(Lc.A, 512, nullptr, nullptr);
        return oc.n_colors() <= 4 && (int)oc.color_of.size() == Lc.A.nr && subdivision_colors(Lc.P, oc.color_of, Lw.A, inherited);
    };
    // Galerkin  A_l = (PT_l * A_{l-1}) * P_l  (:25, :227)
    for (int lv = 1; lv < L; lv++) {
        Level& Lv = h->lv[lv];
        if (Lv.P.nc == 0) return fail(SMG_ERR_INVALID, "level %d has no unknowns left after constraint elimination", lv);
        if (Lv.P.nr != h->lv[lv - 1].A.nr)
            return fail(SMG_ERR_INVALID, "P_%d has %d rows but level %d has %d unknowns", lv, Lv.P.nr, lv - 1, h->lv[lv - 1].A.nr);
        Csr tmp = spgemm(Lv.PT, h->lv[lv - 1].A);
        Lv.A = spgemm(tmp, Lv.P);
        if (lv == L - 1) coarsest_A.store(1, std::memory_order_release);
        // The coarsest smoothed level is coloured from scratch, every finer one waits for its colours (they are inherited, coarse to
        // fine), and the search takes longer than all that is left to do here: it starts the moment that level's matrix exists.
        if (lv == L - 2 && L >= 3 && !blk && use_rcm && host_threads() > 1) {
            cl_started = true;
            cl_thread = std::thread([&, lv] {
                Level& Lw = h->lv[lv];
                const auto t0 = std::chrono::steady_clock::now();
                cl_key = pattern_key(lv);
                if (!(cl_key == Lw.ord_key && (int)Lw.ord.perm.size() == Lw.A.nr)) {
                    const std::vector<int> r = rcm_order(Lw.A);
                    std::vector<int> inherited;
                    const std::vector<int>* preset = nullptr;
                    if (looks_subdivided(lv)) {
                        while (coarsest_A.load(std::memory_order_acquire) == 0) std::this_thread::yield();
                        if (coarsest_A.load(std::memory_order_acquire) == 1 && preset_from_coarsest(lv, inherited)) preset = &inherited;
                    }
                    Lw.ord = make_ordering(Lw.A, 512, preset, &r);
                    Lw.ord_key = cl_key;
                    cl_fresh = true;
                }
                cl_ms = 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            });
        }
    }
    tm.lap("host: Galerkin products");
    if (blk) {
        std::vector<std::function<void()>> tasks;
        for (int lv = 1; lv < L - 1; lv++) tasks.push_back([&, lv] { h->lv[lv].vpat = block_pattern3(h->lv[lv].A); });
        parallel_tasks(tasks);
        tm.lap("host: block patterns of the Galerkin levels");
    }
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
        if (lv < L - 1) {
            std::atomic<int> first_bad{h->lv[lv].n};
            const std::vector<double>& dg = h->lv[lv].A_diag;
            parallel_for(h->lv[lv].n, 1 << 16, [&](long a, long b) {
                for (long i = a; i < b; i++)
                    if (dg[(size_t)i] == 0.0) { int cur = first_bad.load(); while ((int)i < cur && !first_bad.compare_exchange_weak(cur, (int)i)) {} break; }
            });
            if (first_bad.load() < h->lv[lv].n) return fail(SMG_ERR_INVALID, "level %d: zero or missing diagonal at row %d", lv, first_bad.load());
        }
    }
    tm.lap("host: shift, diagonals");
    hand.post_coarse();
    // ---- device numbering (still host work): colour-major ordering of every smoothed level and the operators
    // expressed in it.  The coarsest level is only ever hit by the dense solve and keeps the caller's numbering.
    // coarse to fine, so that a subdivision level can inherit a 4-colouring from its parent; the RCM orders (the expensive,
    // sequential part of an ordering) of all levels that need one are computed concurrently first
    std::vector<uint64_t> keys(L);
    std::vector<char> need(L, 0), fresh_early(L, 0);     // fresh_early: numbered anew ahead of the level loop below
    {
        std::vector<std::function<void()>> tasks;
        for (int lv = 0; lv < L; lv++) tasks.push_back([&, lv] {
            if (cl_started && lv == L - 2) return;      // (being numbered right now)
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
    if (cl_started) { keys[(size_t)L - 2] = 0; need[(size_t)L - 2] = 0; }      // (numbered on its own thread: joined below, after the other levels' searches)
    std::vector<std::vector<int>> rcm(L), pre_colours(L);      // pre_colours: from-scratch colours of levels with nothing to inherit, found beside the searches
    const bool any_need = std::any_of(need.begin(), need.end(), [](char c) { return c != 0; });
    if (any_need) {
        if (use_rcm) {
            // the coarsest smoothed level is coloured from scratch (a search that can take longer than all the RCMs together):
            // it goes first in the task list and runs beside the finer levels' searches
            std::vector<std::function<void()>> tasks;
            if (L >= 2 && need[L - 2]) tasks.push_back([&] {
                Level& Lv = h->lv[L - 2];
                const auto t0 = std::chrono::steady_clock::now();
                rcm[L - 2] = rcm_order(graph(L - 2));
                std::vector<int> inherited;
                const bool have = looks_subdivided(L - 2) && preset_from_coarsest(L - 2, inherited);      // (all Galerkin products exist here)
                order_of(L - 2) = make_ordering(graph(L - 2), 512, have ? &inherited : nullptr, &rcm[L - 2]);
                (void)Lv;
                if (tm.on) std::fprintf(stderr, "[smg timing] host:   (coarsest smoothed level: order + colouring from scratch %.1f ms)\n", 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
                Lv.ord_key = keys[L - 2];
                need[L - 2] = 0; fresh_early[(size_t)L - 2] = 1;
            });
            const bool early0 = e0.rcm_started;
            for (int lv = 0; lv < L - 2; lv++) if (need[lv] && !(lv == 0 && early0)) tasks.push_back([&, lv] {
                rcm[lv] = rcm_order(graph(lv));
                if (!may_inherit(lv)) pre_colours[lv] = colours_for_ordering(graph(lv), rcm[lv]);
            });
            parallel_tasks(tasks);      // (level 0's search, on its own thread since the start, is waited for when level 0's turn comes)
        } else {
            std::vector<int> rank;
            for (int lv = L - 2; lv >= 0; lv--) {
                rcm[lv] = (lv == L - 2) ? rcm_order(graph(lv)) : induced_order(blk ? h->lv[lv + 1].Pv : h->lv[lv + 1].P, rank);
                const int ng = graph(lv).nr;
                rank.assign(ng, 0);
                for (int t = 0; t < ng; t++) rank[rcm[lv][t]] = t;
            }
        }
    }
    tm.lap("host:   locality orders (RCM) + coarsest colouring");
    if (cl_started) {
        cl_thread.join();
        keys[(size_t)L - 2] = cl_key; fresh_early[(size_t)L - 2] = cl_fresh ? 1 : 0;
        if (tm.on) { std::fprintf(stderr, "[smg timing] host:   (coarsest smoothed level: order + colouring, own thread %.1f ms)\n", cl_ms); tm.lap("host:   waiting for the coarsest smoothed level's colours"); }
    }
    for (int lv = L - 1; lv >= 0; lv--) {
        Level& Lv = h->lv[lv];
        bool fresh = fresh_early[(size_t)lv] != 0;
        if (need[lv]) {
            fresh = true;
            if (lv == L - 1) { Lv.ord = identity_ordering(Lv.n); if (blk) Lv.vord = identity_ordering(Lv.n / 3); }
            else {
                if (lv == 0 && use_rcm && e0.rcm_started) {
                    e0.wait();
                    if (tm.on) { std::fprintf(stderr, "[smg timing] host:   (level 0 locality order, own thread: %.1f ms; colours from scratch beside it: %.1f ms)\n", e0.rcm_ms, e0.colour_ms); tm.lap("host:   waiting for level 0's locality order / colours"); }
                    if ((int)e0.rcm.size() != graph(0).nr) { e0.rcm = rcm_order(graph(0)); e0.colours.clear(); }     // (the early thread found the pattern unchanged, the key says otherwise: cannot happen)
                    rcm[0] = std::move(e0.rcm);
                    if ((int)e0.colours.size() == graph(0).nr) pre_colours[0] = std::move(e0.colours);
                }
                std::vector<int> inherited;
                const Level& Lc = h->lv[lv + 1];
                const Ordering& Oc = order_of(lv + 1);
                const bool have_pre = (int)pre_colours[lv].size() == graph(lv).nr && graph(lv).nr > 0;      // (only ever set where nothing can be inherited)
                const bool ok = !have_pre &&
                                (((lv + 1 < L - 1) && Oc.n_colors() <= 4 && (int)Oc.color_of.size() == graph(lv + 1).nr &&
                                  subdivision_colors(blk ? Lc.Pv : Lc.P, Oc.color_of, graph(lv), inherited)) ||
                                 (looks_subdivided(lv) && preset_from_coarsest(lv, inherited)));      // (SMG_ORDER=induced: the coarsest smoothed level is numbered here)
                if (tm.on) { char nm[64]; std::snprintf(nm, sizeof nm, "host:   level %d colours inherited=%d", lv, (int)ok); tm.lap(nm); }
                order_of(lv) = have_pre ? make_ordering(graph(lv), 512, &pre_colours[lv], &rcm[lv], true) : make_ordering(graph(lv), 512, ok ? &inherited : nullptr, &rcm[lv]);
                if (tm.on) { char nm[64]; std::snprintf(nm, sizeof nm, "host:   level %d make_ordering", lv); tm.lap(nm); }
            }
            Lv.ord_key = keys[lv];
        }
        if (blk && lv < L - 1 && (fresh || (int)Lv.ord.perm.size() != Lv.n)) {
            // the DOF numbering a vertex numbering induces: DOF 3v+d of vertex v, colours = vertex colours
            const Ordering& O = Lv.vord;
            const int nv = (int)O.perm.size();
            Lv.ord = Ordering();
            Lv.ord.perm.resize((size_t)3 * nv); Lv.ord.iperm.resize((size_t)3 * nv); Lv.ord.color_of.resize((size_t)3 * nv);
            for (int i = 0; i < nv; i++)
                for (int d = 0; d < 3; d++) {
                    Lv.ord.perm[(size_t)3 * i + d] = 3 * O.perm[i] + d;
                    Lv.ord.iperm[(size_t)3 * O.perm[i] + d] = 3 * i + d;
                    Lv.ord.color_of[(size_t)3 * i + d] = O.color_of.empty() ? 0 : O.color_of[i];   // (caller numbering, like O.color_of)
                }
            for (int c : O.color_ptr) Lv.ord.color_ptr.push_back(3 * c);
        }
        hand.post_level(lv);     // the device half takes it from here (level_images)
    }
    tm.lap("host: orderings + colourings");
    return SMG_OK;
}

// Gershgorin bound of D^-1 A per smoothed level (what the Chebyshev-Jacobi smoother is built on), from the SELL image the smoother
// streams, i.e. in the device numbering's summation order -- the same value the oracle computes on the level matrix in that numbering.
int smg::spectral_bounds(smg_hierarchy* h)
{
    const int L = h->n_levels;
    if (L < 2) return SMG_OK;
    HIPCHK(h->d_lam.ensure((size_t)L));
    for (int lv = 0; lv < L - 1; lv++) {
        Level& Lv = h->lv[lv];
        if (h->bs == 3) HIPCHK(launch_bsr3_gershgorin(Lv.gs_on_transpose ? Lv.bAT.view : Lv.bA.view, h->d_lam.p + lv, h->stream));
        else HIPCHK(launch_gershgorin(Lv.gs_on_transpose ? Lv.dAT.view : Lv.dA.view, h->d_lam.p + lv, h->stream));
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
// lazily: only handles that smooth with Chebyshev-Jacobi pay the four small launches and the read-back
int smg::ensure_spectral_bounds(smg_hierarchy* h)
{
    if (h->lam_valid) return SMG_OK;
    bool need = false;
    for (int lv = 0; lv < h->n_levels - 1; lv++) if (level_kind(h, lv) == 2 /* LV_CHEBY */) need = true;
    return need ? spectral_bounds(h) : SMG_OK;
}

// ---- sparse coarse solver (smg_coarse.hpp): coarsest levels beyond the dense range -----------------------------------------------
// solver.compute(Ac) (reference src/min_quad_with_fixed_mg.cpp:47-48, :253-254) on the host, like the reference; the factor goes to HBM.
// reuse: same pattern as the last factorisation (value-only re-precompute): the ordering is kept.
static int coarse_factor_sparse(smg_hierarchy* h, const Csr& Ac, bool reuse)
{
    StageTimer tm;
    if (!sparse_cholesky(Ac, h->chol, reuse && h->coarse_sparse))
        return fail(SMG_ERR_INVALID, "coarsest matrix (%d unknowns) is not positive definite: sparse Cholesky met a non-positive pivot", Ac.nr);
    tm.lap("host: sparse Cholesky of the coarsest matrix");
    const SparseChol& F = h->chol;
    // The captured graphs hold SparseCholDev BY VALUE (kernel arguments): a value-only refactorisation writes into the buffers they point at;
    // whatever has to be reallocated (first factorisation, another pattern) invalidates them.  The solve stream may still be reading the old factor.
    HIPCHK(hipStreamSynchronize(h->stream));
    bool moved = !h->coarse_sparse;
    const double *work0 = h->c_work.p; const int* err0 = h->c_err.p;
    HIPCHK(h->c_perm.upload_in_place(F.perm, &moved)); HIPCHK(h->c_rptr.upload_in_place(F.rptr, &moved)); HIPCHK(h->c_rcol.upload_in_place(F.rcol, &moved));
    HIPCHK(h->c_cptr.upload_in_place(F.cptr, &moved)); HIPCHK(h->c_crow.upload_in_place(F.crow, &moved)); HIPCHK(h->c_rval.upload_in_place(F.rval, &moved));
    HIPCHK(h->c_cval.upload_in_place(F.cval, &moved)); HIPCHK(h->c_diag.upload_in_place(F.diag, &moved));
    HIPCHK(h->c_work.ensure((size_t)2 * F.n));
    HIPCHK(h->c_err.ensure(4));          // [0] the stall flag, [1] / [2] the ticket counters of the forward / backward launch
    HIPCHK(hipMemset(h->c_err.p, 0, 4 * sizeof(int)));
    if (moved || h->c_work.p != work0 || h->c_err.p != err0) drop_graphs(h);
    SparseCholDev& V = h->c_view;
    V.n = F.n; V.perm = h->c_perm.p; V.rptr = h->c_rptr.p; V.rcol = h->c_rcol.p; V.cptr = h->c_cptr.p; V.crow = h->c_crow.p;
    V.rval = h->c_rval.p; V.cval = h->c_cval.p; V.diag = h->c_diag.p; V.work = h->c_work.p; V.err = h->c_err.p;
    h->coarse_sparse = true;
    h->coarse_schur = false; h->sch.release(); h->schur = SchurPlan();
    h->d_Ainv.release(); h->d_Ainv32.release(); h->d_sympart.release();
    tm.lap("device: sparse coarse factor uploaded");
    return SMG_OK;
}

// ---- Schur-complement coarse solver (smg_schur.hpp): the plan of a coarsest matrix in the upper part of the dense range, on the device --
// phase 1: at a full precompute; phase 2: at a value-only re-precompute of a handle that still holds the dense inverse
static bool schur_wanted(const smg_hierarchy* h, int n, int phase)
{
    if (h->coarse_schur_when == 0 || h->coarse_schur_min < 0 || n < h->coarse_schur_min || n > h->coarse_schur_max || h->schur_declined) return false;
    if (phase == 2) return h->coarse_schur_when == 2;
    // from the start: on request, where it is the cheaper solver to apply as well, and where a dense inverse of the whole matrix is not allowed
    // (a caller who SET coarse_dense_max asked for the sparse factorisation above it -- to bound memory, say: the Schur solver stands in for the sparse one
    //  beyond the dense range only while that range is the default)
    if (n > h->coarse_dense_max) return h->coarse_schur_when == 1 || !h->coarse_dense_max_user;
    return h->coarse_schur_when == 1 || n >= h->coarse_schur_big;
}
// after launch_schur_factor on h->stream: synchronises, and fails like the sparse path when the matrix cannot have been positive definite
static int schur_check_spd(smg_hierarchy* h, int n)
{
    HIPCHK(h->c_err.ensure(4));
    HIPCHK(launch_schur_check(h->sch.view, h->c_err.p + 3, h->stream));
    int flag = 0;
    HIPCHK(hipMemcpyAsync(&flag, h->c_err.p + 3, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (flag) return fail(SMG_ERR_INVALID, "coarsest matrix (%d unknowns) is not positive definite: the block elimination met a non-positive or non-finite pivot", n);
    return SMG_OK;
}
// *planned: h->sch holds the plan of Ac (the arena still to be factored); not: no plan for this matrix (the dense inverse serves)
static int coarse_plan_schur(smg_hierarchy* h, const Csr& Ac, bool* planned)
{
    *planned = false;
    static_assert(SCHUR_M_MAX == SCHUR_M_MAX_DEV && SCHUR_B == 64, "smg_schur.hpp and smg_device.hpp disagree");
    StageTimer tm;
    h->schur = build_schur(Ac);
    const SchurPlan& P = h->schur;
    if (P.empty()) return SMG_OK;
    smg_hierarchy::SchurBuf& B = h->sch;
    // Any allocation or upload that fails -- the arena of the separator's dense inverse above all, on the value-only path with the dense inverse of the
    // whole matrix still resident -- declines the plan (nothing half-built stays behind) and the other solvers serve: a device that is nearly full keeps
    // working.  A byte budget bounds the arena besides the row cap (SMG_SCHUR_ARENA_MAX_MB, default 6144: 4.8 GB at the 24 576-row cap of the separator).
    static const long long arena_max = (long long)env_int("SMG_SCHUR_ARENA_MAX_MB", 6144) * (1ll << 20);
    hipError_t e = (long long)P.total * (long long)sizeof(double) > arena_max ? hipErrorOutOfMemory : hipSuccess;
    auto up = [&](auto& buf, const auto& v) { if (e == hipSuccess) e = buf.upload(v); };
    up(B.irow, P.irow); up(B.bsize, P.bsize); up(B.srow, P.srow); up(B.sptr, P.sptr); up(B.sidx, P.sidx);
    up(B.aptr, P.aptr); up(B.ablk, P.ablk); up(B.acol, P.apan); up(B.rptr, P.rptr);
    up(B.coff, P.coff); up(B.pos, P.pos); up(B.pos2, P.pos2); up(B.ones, P.ones);
    up(B.rdst, P.rdst); up(B.rdst2, P.rdst2); up(B.rsrc, P.rsrc);
    if (e == hipSuccess) e = B.arena.alloc((size_t)P.total);
    if (e == hipSuccess) e = B.gj.alloc((size_t)2 * P.ns_pad * 64 + 2 * 64 * 64);
    if (e == hipSuccess) { if (env_int("SMG_SYM_COARSE", 1)) e = B.sym.alloc((size_t)(P.ns_pad / 64) * (P.ns_pad / 64) * 64); else B.sym.release(); }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        B.release(); h->schur = SchurPlan();
        return SMG_OK;
    }
    B.arena32.release(); B.g.release(); B.xs.release(); B.g32.release(); B.xs32.release();
    SchurDev& V = B.view;
    V = SchurDev();
    V.n = P.n; V.nb = P.nb; V.ns = P.ns; V.ns_pad = P.ns_pad;
    V.irow = B.irow.p; V.bsize = B.bsize.p; V.srow = B.srow.p; V.sptr = B.sptr.p; V.sidx = B.sidx.p; V.aptr = B.aptr.p; V.ablk = B.ablk.p; V.apan = B.acol.p;
    V.arena = B.arena.p;
    V.off_D = P.off_D; V.off_P = P.off_P; V.off_W = P.off_W; V.off_S = P.off_S; V.off_C = P.off_C;
    V.coff = B.coff.p; V.pos = B.pos.p; V.pos2 = B.pos2.p; V.ones = B.ones.p; V.rdst = B.rdst.p; V.rdst2 = B.rdst2.p; V.rsrc = B.rsrc.p; V.rptr = B.rptr.p;
    V.nnz = (int)P.pos.size(); V.n_ones = (int)P.ones.size(); V.n_red = (int)P.rdst.size();
    V.sym_work = B.sym.p; V.gj_work = B.gj.p;
    if (env_int("SMG_DEBUG_SCHUR", 0))
        std::fprintf(stderr, "[smg schur] %d unknowns: %d interior blocks, %d separator rows (padded %d), %.1f separator rows per block, arena %.1f MB\n", P.n, P.nb, P.ns, P.ns_pad,
                     (double)P.sidx.size() / P.nb, 8e-6 * (double)P.total);
    tm.lap("host: plan of the Schur-complement coarse solver, uploaded");
    *planned = true;
    return SMG_OK;
}

// ---- device half: the images of every level in the colour-major numbering, the coarse factorisation ----------------------------------
// Runs on the calling thread beside the host half (Handoff): device_begin, coarse_images once the coarsest matrix is final, level_images
// for every smoothed level as its numbering arrives (coarse to fine), finish_images.
static int device_begin(smg_hierarchy* h)
{
    HIPCHK(hipStreamSynchronize(h->stream));
    drop_graphs(h);
    drop_tiled(h);
    for (Level& Lv : h->lv) {
        Lv.b.release(); Lv.u.release(); Lv.r.release(); Lv.t.release(); Lv.d.release();
        Lv.b32.release(); Lv.u32.release(); Lv.r32.release(); Lv.t32.release(); Lv.d32.release();
    }
    h->kcap = 0; h->kcap32 = 0; h->f32_valid = false;
    return SMG_OK;
}

// run the tasks; with SMG_TIMING their durations are listed
static void run_image_tasks(const char* what, int lv, std::vector<std::function<void()>>& tasks)
{
    static const bool on = env_int("SMG_TIMING", 0) != 0;
    if (!on) { parallel_tasks(tasks); return; }
    std::vector<double> ms(tasks.size(), 0.0);
    std::vector<std::function<void()>> timed;
    for (size_t i = 0; i < tasks.size(); i++)
        timed.push_back([&, i] {
            const auto t0 = std::chrono::steady_clock::now();
            tasks[i]();
            ms[i] = 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        });
    const auto t0 = std::chrono::steady_clock::now();
    parallel_tasks(timed);
    std::fprintf(stderr, "[smg timing] device:   level %d %s: %.1f ms, tasks", lv, what, 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    for (double v : ms) std::fprintf(stderr, " %.1f", v);
    std::fprintf(stderr, "\n");
}

// The coarsest level: only ever hit by coarseSolve, keeps the caller's numbering.  Dense inverse on the device (stands in for
// solver.compute(Ac), :47-48 / :253-254) up to smg_hierarchy_set_coarse_dense_max unknowns; beyond, the reference's own method: sparse
// Cholesky (smg_coarse.hpp).
static int coarse_images(smg_hierarchy* h)
{
    const int L = h->n_levels;
    Level& Lc = h->lv[L - 1];
    Lc.device_filled = false;
    Lc.A_int = Lc.A;
    Lc.A_int_src.resize((size_t)Lc.A.nnz());
    std::iota(Lc.A_int_src.begin(), Lc.A_int_src.end(), 0);
    if (L == 1) {
        // a single level goes straight to coarseSolve (src/mg_VCycle.cpp:28-33); the outer loop still needs A_0 for its residual
        Sell S = build_sell(Lc.A_int, nullptr, SELL_C, false, h->mem_lean ? 0 : -1);
        HIPCHK(Lc.dA.upload(S));
    }
    const int nc = Lc.n;
    const int np = ((nc + 63) / 64) * 64;
    h->coarse_schur = false;
    h->schur_declined = false;
    if (h->union_m > 0) {      // independent meshes in one handle: the members' own dense inverses (smg_union.cpp)
        if (h->bs != 1) return fail(SMG_ERR_INVALID, "a union handle needs scalar hierarchies (no 3-DOF block structure)");
        h->sch.release(); h->schur = SchurPlan();
        DevBuf<double> d_val;
        HIPCHK(d_val.upload(Lc.A.val));
        return union_coarse_factor(h, d_val.p, true);
    }
    if (schur_wanted(h, nc, 1)) {
        bool planned = false;
        const int rc = coarse_plan_schur(h, Lc.A, &planned);
        if (rc) return rc;
        if (planned) {
            DevBuf<double> d_val;
            HIPCHK(d_val.upload(Lc.A.val));
            HIPCHK(launch_schur_factor(h->sch.view, d_val.p, h->stream));
            { int rc2 = schur_check_spd(h, nc); if (rc2) return rc2; }
            h->nc = nc; h->nc_pad = np;
            h->coarse_schur = true;
            if (h->coarse_sparse) {   // the handle held a sparse factorisation (another matrix, another policy)
                h->coarse_sparse = false;
                h->c_perm.release(); h->c_rptr.release(); h->c_rcol.release(); h->c_cptr.release(); h->c_crow.release(); h->c_rval.release(); h->c_cval.release(); h->c_diag.release(); h->c_work.release();
            }
            h->d_Ainv.release(); h->d_Ainv32.release(); h->d_sympart.release();
            return SMG_OK;
        }
    }
    h->sch.release(); h->schur = SchurPlan();
    if (Lc.n > h->coarse_dense_max) {
        h->nc = Lc.n; h->nc_pad = Lc.n;
        return coarse_factor_sparse(h, Lc.A, false);
    }
    h->coarse_sparse = false;
    h->nc = nc; h->nc_pad = np;
    // dense image on the device: the few entries travel, not n^2 zeros
    std::vector<long long> pos((size_t)Lc.A.nnz());
    for (int i = 0; i < nc; i++)
        for (int p = Lc.A.ptr[i]; p < Lc.A.ptr[i + 1]; p++) pos[(size_t)p] = (long long)i * np + Lc.A.col[p];
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
    return SMG_OK;
}

// Level lv < L - 1, once its numbering and that of level lv + 1 are final: A_lv (and A_lv^T where the two differ in any bit) and the
// transfer operators between the two levels, P_{lv+1} and PT_{lv+1}, expressed in the device numbering and stored as SELL images.
// sym0: level 0 only -- 1 / 0: A_0 == A_0^T was already decided (on the device, from the arrays sent ahead), -1: not.
static int level_images(smg_hierarchy* h, int lv, int sym0)
{
    const int L = h->n_levels;
    const int sellC = SELL_C;
    const bool region = env_int("SMG_REGION_ORDER", 1) != 0;   // A/B knob: region-major launch order
    const bool blk = h->bs == 3;
    const int lp = lv + 1;                                    // the level whose P / PT connect the two
    Level& Lw = h->lv[lv];
    Level& Lp = h->lv[lp];
    // ---- host: the operators in the device numbering (or the decision that the device fills the panels from the caller-order arrays)
    {
        std::vector<std::function<void()>> tasks;
        tasks.push_back([h, lv, &Lw] {
            Lw.device_filled = false;
            if (device_fill_candidate(h, lv)) {
                Lw.A_int = Csr(); Lw.A_int_src.clear();      // built on demand (ensure_A_int): the device fills the panels from A itself
                Lw.device_filled = true;
            } else Lw.A_int = permute(Lw.A, Lw.ord.perm, Lw.ord.perm, &Lw.A_int_src);
        });
        // (the transfer operators of a big level are filled on the device as well: their permuted host copies are built on demand)
        tasks.push_back([h, &Lw, &Lp] {
            Lp.P_device_filled = device_fill_rows(h, Lp.P.nr);
            if (Lp.P_device_filled) Lp.P_int = Csr(); else Lp.P_int = permute(Lp.P, Lw.ord.perm, Lp.ord.perm);
        });
        tasks.push_back([h, &Lw, &Lp] {
            static const int long_min = env_int("SMG_LONG_ROW_MIN", 65);
            bool ok = device_fill_rows(h, Lp.PT.nr);
            if (ok && long_min > 0) for (int r = 0; r < Lp.PT.nr && ok; r++) if (Lp.PT.ptr[(size_t)r + 1] - Lp.PT.ptr[(size_t)r] >= long_min) ok = false;   // long rows leave the panels: host path
            Lp.PT_device_filled = ok;
            if (ok) Lp.PT_int = Csr(); else Lp.PT_int = permute(Lp.PT, Lp.ord.perm, Lw.ord.perm);
        });
        run_image_tasks("operators in the device numbering", lv, tasks);
    }
    // ---- device: the SELL images, concurrently on host threads, each uploaded by the task that built it (pageable-memory copies are
    // bound by the host-side staging copy, so they overlap with the other tasks' work and with each other)
    int bad = 0;
    hipError_t eA = hipSuccess, eT = hipSuccess, eP = hipSuccess, eQ = hipSuccess;
    std::vector<std::function<void()>> tasks;
    tasks.push_back([&] {
        DeviceScope ds(h->device);   // worker threads start on device 0
        if (blk && Lw.device_filled) {
            // the block image from the scalar arrays in the caller's numbering, the pattern of the blocks and the vertex numbering: layout on the
            // host (block-row lengths), panels on the device -- and, as for the scalar images, A == A^T decided there and the A^T image too
            Lw.dA = SellBuf();
            DeviceCsr D;
            eA = D.put(Lw.A, lv == 0 && !h->has_known ? &h->early0 : nullptr);
            int differs = sym0 == 1 ? 0 : -1;
            DevBuf<int> d_differs, d_gptr, d_gcol, d_perm, d_iperm;
            if (eA == hipSuccess && differs < 0) {
                eA = d_differs.alloc(1);
                if (eA == hipSuccess) eA = launch_bit_symmetric(Lw.A.nr, D.ptr, D.col, D.val, d_differs.p, h->aux[0]);
                if (eA == hipSuccess) eA = hipMemcpyAsync(&differs, d_differs.p, sizeof(int), hipMemcpyDeviceToHost, h->aux[0]);
                if (eA == hipSuccess) eA = hipStreamSynchronize(h->aux[0]);
            }
            if (eA != hipSuccess) return;
            if (Lw.A.nr != Lw.A.nc || (differs & 2)) { bad = 1; return; }
            Lw.A_bit_symmetric = differs == 0;
            Lw.gs_on_transpose = differs != 0;
            Lw.bAT = Bsr3Buf(); Lw.dAT = SellBuf();
            const Csr& G = Lw.vpat;
            const std::vector<int>& vp = Lw.vord.perm;
            std::vector<int> row_len(vp.size());
            parallel_for((long)vp.size(), 1 << 16, [&](long r0, long r1) { for (long r = r0; r < r1; r++) row_len[(size_t)r] = G.ptr[(size_t)vp[(size_t)r] + 1] - G.ptr[(size_t)vp[(size_t)r]]; });
            if (eA == hipSuccess) eA = d_gptr.upload(G.ptr);
            if (eA == hipSuccess) eA = d_gcol.upload(G.col);
            if (eA == hipSuccess) eA = d_perm.upload(vp);
            if (eA == hipSuccess) eA = d_iperm.upload(Lw.vord.iperm);
            const Bsr3Sell S = bsr3_layout(row_len, &Lw.vord.color_ptr, region, Lw.A.nnz(), G.nnz());
            if (eA == hipSuccess) eA = Lw.bA.upload(S);
            if (eA == hipSuccess) eA = launch_bsr3_fill(D.ptr, D.col, D.val, d_gptr.p, d_gcol.p, d_perm.p, d_iperm.p, Lw.bA.view, (size_t)S.slice_off.back(), false, h->aux[0]);
            if (eA == hipSuccess && Lw.gs_on_transpose) {
                const Bsr3Sell ST = bsr3_layout(row_len, &Lw.vord.color_ptr, false, Lw.A.nnz(), G.nnz());
                eT = Lw.bAT.upload(ST);
                if (eT == hipSuccess) eT = launch_bsr3_fill(D.ptr, D.col, D.val, d_gptr.p, d_gcol.p, d_perm.p, d_iperm.p, Lw.bAT.view, (size_t)ST.slice_off.back(), true, h->aux[0]);
            }
            if (eA == hipSuccess) eA = hipStreamSynchronize(h->aux[0]);
            return;
        }
        if (blk) {
            Lw.dA = SellBuf();
            Bsr3Sell S = build_bsr3(Lw.A_int, &Lw.vord.color_ptr, region, false);
            eA = Lw.bA.upload(S);
            return;
        }
        Lw.bA = Bsr3Buf();
        if (Lw.device_filled) {
            // layout from the row lengths; the caller's arrays and the permutation travel, the panels are written on the device.  The sweep
            // streams A^T where the two differ in any bit (see below): decided on the device as well, and that image, too, is filled there.
            DeviceCsr D;
            eA = D.put(Lw.A, lv == 0 && !h->has_known ? &h->early0 : nullptr);
            int differs = sym0 == 1 ? 0 : -1;      // (known to be symmetric / to be found out: which of the two ways it is not matters)
            if (eA == hipSuccess && differs < 0) {
                DevBuf<int> d_differs;
                eA = d_differs.alloc(1);
                if (eA == hipSuccess) eA = launch_bit_symmetric(Lw.A.nr, D.ptr, D.col, D.val, d_differs.p, h->aux[0]);
                if (eA == hipSuccess) eA = hipMemcpyAsync(&differs, d_differs.p, sizeof(int), hipMemcpyDeviceToHost, h->aux[0]);
                if (eA == hipSuccess) eA = hipStreamSynchronize(h->aux[0]);
            }
            if (eA != hipSuccess) return;
            if (Lw.A.nr != Lw.A.nc || (differs & 2)) { bad = 1; return; }
            Lw.A_bit_symmetric = differs == 0;
            Lw.gs_on_transpose = differs != 0;
            Lw.dAT = SellBuf(); Lw.bAT = Bsr3Buf();
            eA = device_fill_sell(Lw.dA, Lw.A, D, Lw.ord.perm, Lw.ord.iperm, &Lw.ord.color_ptr, region, h->aux[0], false, h->mem_lean ? 0 : -1);
            if (eA == hipSuccess && Lw.gs_on_transpose) eT = device_fill_sell(Lw.dAT, Lw.A, D, Lw.ord.perm, Lw.ord.iperm, &Lw.ord.color_ptr, false, h->aux[0], true, h->mem_lean ? 0 : -1);
            return;
        }
        Sell S = build_sell(Lw.A_int, &Lw.ord.color_ptr, sellC, region, h->mem_lean ? 0 : -1);
        eA = Lw.dA.upload(S);
    });
    // relax() iterates InnerIterator(A, colIdx): the entries A(j, i) of COLUMN i (src/mg_VCycle.cpp:149-155,
    // "legal" because A is symmetric).  Galerkin products are symmetric only up to rounding, so the sweep
    // streams A^T wherever the two differ in any bit; the SpMV / residual keep the true rows.
    tasks.push_back([&] {
        DeviceScope ds(h->device);
        if (Lw.device_filled) return;      // (the task above decides, and fills that image as well)
        Lw.dAT = SellBuf();
        Lw.bAT = Bsr3Buf();
        // big matrices: the symmetric ones (a caller's level 0, as a rule) are recognised without transposing tens of millions of entries
        if (Lw.A_int.nnz() >= (1L << 22) && bit_symmetric(Lw.A_int)) { Lw.gs_on_transpose = false; return; }
        Csr AT = transpose(Lw.A_int);
        Lw.gs_on_transpose = !(AT.ptr == Lw.A_int.ptr && AT.col == Lw.A_int.col && AT.val == Lw.A_int.val);
        if (Lw.gs_on_transpose) {
            if (!(AT.ptr == Lw.A_int.ptr && AT.col == Lw.A_int.col)) { bad = 1; return; }
            if (blk) { Bsr3Sell S = build_bsr3(AT, &Lw.vord.color_ptr, false, false); eT = Lw.bAT.upload(S); }
            else { Sell S = build_sell(AT, &Lw.ord.color_ptr, sellC, false, h->mem_lean ? 0 : -1); eT = Lw.dAT.upload(S); }
        }
    });
    // P and PT are launched whole: with their rows cut at the colour boundaries of the level they belong to, the slices get
    // the same region-major launch order as A, and the workgroups an XCD receives (a contiguous piece of that order) read
    // their gathers from one region of the mesh instead of from all over it (restriction at C3: 54 MB of HBM traffic per
    // launch for 33 MB of algorithmic bytes before)
    static const bool tr_region = env_int("SMG_TRANSFER_REGION_ORDER", 1) != 0;
    tasks.push_back([&] {
        DeviceScope ds(h->device);
        // block hierarchies: the device applies the VERTEX-level factor of P (x) I_3 to 3 k columns (smg_bsr3.hpp)
        const Ordering& Of = blk ? Lw.vord : Lw.ord;
        const bool cut = tr_region && region && Of.color_ptr.size() > 2;
        if (!blk && Lp.P_device_filled) {
            eP = device_fill_sell(Lp.dP, Lp.P, Of.perm, Lp.ord.iperm, cut ? &Of.color_ptr : nullptr, cut, h->aux[1], h->mem_lean ? 0 : -1);
            return;
        }
        Csr Pvi;
        if (blk) Pvi = permute(Lp.Pv, Of.perm, Lp.vord.perm);
        Sell S = build_sell(blk ? Pvi : Lp.P_int, cut ? &Of.color_ptr : nullptr, sellC, cut, h->mem_lean ? 0 : -1);
        eP = Lp.dP.upload(S);
    });
    tasks.push_back([&] {
        DeviceScope ds(h->device);
        const Ordering& Oc = blk ? Lp.vord : Lp.ord;
        const bool cut = tr_region && region && lp < L - 1 && Oc.color_ptr.size() > 2;
        // rows with many entries (a coarse vertex of a decimated level that absorbed dozens of fine ones) leave the panels:
        // a panel row is one chain of dependent batches and the longest one sets the duration of the restriction launch
        // (ogre.obj level 0 -> 1: a row of 177 entries, 32 us of a 260 us cycle); see SellDev::long_* / k_long_ax
        static const int long_min = env_int("SMG_LONG_ROW_MIN", 65);
        if (!blk && Lp.PT_device_filled) {
            eQ = device_fill_sell(Lp.dPT, Lp.PT, Oc.perm, Lw.ord.iperm, cut ? &Oc.color_ptr : nullptr, cut, h->aux[2], h->mem_lean ? 0 : -1);
            if (eQ == hipSuccess) eQ = Lp.dPT.upload_long({}, {0}, {}, {});
            return;
        }
        Csr PTvi;
        if (blk) PTvi = permute(Lp.PTv, Oc.perm, Lw.vord.perm);
        const Csr& M = blk ? PTvi : Lp.PT_int;
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
            Sell S = build_sell(M, cut ? &Oc.color_ptr : nullptr, sellC, cut, h->mem_lean ? 0 : -1);
            eQ = Lp.dPT.upload(S);
            if (eQ == hipSuccess) eQ = Lp.dPT.upload_long(lrow, lptr, lcol, lval);
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
        Sell S = build_sell(Ms, cut ? &Oc.color_ptr : nullptr, sellC, cut, h->mem_lean ? 0 : -1);
        eQ = Lp.dPT.upload(S);
        if (eQ == hipSuccess) eQ = Lp.dPT.upload_long(lrow, lptr, lcol, lval);
    });
    run_image_tasks("images", lv, tasks);
    if (bad) return fail(SMG_ERR_INVALID, "level %d matrix is not structurally symmetric", lv);
    HIPCHK(eA); HIPCHK(eT); HIPCHK(eP); HIPCHK(eQ);
    return SMG_OK;
}

// level-0 index maps; the spectral bounds are computed when a Chebyshev smoother first asks for them (ensure_spectral_bounds)
static int finish_images(smg_hierarchy* h)
{
    const Level& L0 = h->lv[0];
    std::vector<int> map0((size_t)L0.n);
    for (int i = 0; i < L0.n; i++) map0[(size_t)i] = h->has_known ? h->unknown[(size_t)L0.ord.perm[(size_t)i]] : L0.ord.perm[(size_t)i];
    HIPCHK(h->d_map0.upload(map0));
    HIPCHK(h->d_perm0.upload(L0.ord.perm));
    if (h->has_known) {
        HIPCHK(h->d_unknown.upload(h->unknown));
        HIPCHK(h->d_known.upload(h->known));
        HIPCHK(h->d_auk_ptr.upload(h->Auk.ptr));
        HIPCHK(h->d_auk_col.upload(h->Auk.col));
        HIPCHK(h->d_auk_val.upload(h->Auk.val));
    }
    h->lam_valid = false;
    return SMG_OK;
}

// ---- value-only re-precompute --------------------------------------------------------
// Time-stepping callers hand in a new matrix with the SAME sparsity every step (05_example_mean_curvature_flow/
// main.cpp:74, 06_example_balloon_sim/implicit_euler_mg_balloon.h:75).  Then everything structural (unknown set,
// sliced P, Galerkin patterns, colouring, SELL layout, graphs) is unchanged and the numeric work moves to the GPU:
// slice gathers, two fixed-recipe SpGEMM stages per level (bit-identical to the host spgemm), SELL value refresh and
// the dense coarse inverse.

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
    if (h->bs == 3) for (int lv = 0; lv < L; lv++) { int rc = ensure_A_int(h, lv); if (rc) return rc; }     // (the block maps are built on the host)
    for (hipStream_t& a : h->aux) if (!a) HIPCHK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    drop_graphs(h);  // the GS launches move to the A^T images on every level
    drop_tiled(h);   // ... and so do the overlapped-tiling plans (rebuilt on demand)
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
                if (h->bs != 3) {
                    // Scalar images: the maps are written on the device, by the kernel that can also fill the image (k_sell_fill<MAP>: the slot
                    // an entry lands in is its rank among the row's new column numbers, whoever built the image) -- no permuted copy, no
                    // transposition, no second SELL build on the host.  (The pattern was found symmetric when the images were made.)
                    up(er, Lv.d_Aval.upload(Lv.A.val));
                    DevBuf<int> d_ptr, d_col, d_perm, d_iperm;
                    up(er, d_ptr.upload(Lv.A.ptr)); up(er, d_col.upload(Lv.A.col)); up(er, d_perm.upload(Lv.ord.perm)); up(er, d_iperm.upload(Lv.ord.iperm));
                    hipStream_t st2 = h->aux[lv % 3];
                    up(er, Lv.mapA.alloc((size_t)Lv.dA.padded));
                    if (er == hipSuccess) up(er, launch_sell_fill_map(d_ptr.p, d_col.p, d_perm.p, d_iperm.p, Lv.dA.view, (size_t)Lv.dA.padded, false, Lv.mapA.p, st2));
                    if (!Lv.gs_on_transpose && er == hipSuccess) {      // (values: whatever A holds now -- refreshed through the map right after)
                        DeviceCsr D;
                        D.ptr = d_ptr.p; D.col = d_col.p; D.val = Lv.d_Aval.p;
                        up(er, device_fill_sell(Lv.dAT, Lv.A, D, Lv.ord.perm, Lv.ord.iperm, &Lv.ord.color_ptr, false, st2, true, h->mem_lean ? 0 : -1));
                        Lv.gs_on_transpose = true;
                    }
                    up(er, Lv.mapAT.alloc((size_t)Lv.dAT.padded));
                    if (er == hipSuccess) up(er, launch_sell_fill_map(d_ptr.p, d_col.p, d_perm.p, d_iperm.p, Lv.dAT.view, (size_t)Lv.dAT.padded, true, Lv.mapAT.p, st2));
                    if (er == hipSuccess) up(er, hipStreamSynchronize(st2));
                    return;
                }
                std::vector<int> m;
                std::vector<int> tsrc;
                Csr AT = transpose(Lv.A_int, &tsrc);
                if (!(AT.ptr == Lv.A_int.ptr && AT.col == Lv.A_int.col)) { bad[lv] = 1; return; }
                if (h->bs == 3) {   // the same maps for the value planes of the block images
                    {
                        Bsr3Sell S = build_bsr3(Lv.A_int, &Lv.vord.color_ptr, false);
                        m.resize(S.entry.size());
                        for (size_t i = 0; i < S.entry.size(); i++) m[i] = S.entry[i] >= 0 ? Lv.A_int_src[S.entry[i]] : -1;
                    }
                    up(er, Lv.mapB.upload(m));
                    Bsr3Sell ST = build_bsr3(AT, &Lv.vord.color_ptr, false);
                    m.resize(ST.entry.size());
                    for (size_t i = 0; i < ST.entry.size(); i++) m[i] = ST.entry[i] >= 0 ? Lv.A_int_src[tsrc[ST.entry[i]]] : -1;
                    up(er, Lv.mapBT.upload(m));
                    if (!Lv.gs_on_transpose) { up(er, Lv.bAT.upload(ST)); Lv.gs_on_transpose = true; }
                    return;
                }
                {
                    Sell S = build_sell(Lv.A_int, &Lv.ord.color_ptr, sellC, false, h->mem_lean ? 0 : -1);
                    m.resize(S.entry.size());
                    for (size_t i = 0; i < S.entry.size(); i++) m[i] = S.entry[i] >= 0 ? Lv.A_int_src[S.entry[i]] : -1;
                }
                up(er, Lv.mapA.upload(m));
                Sell ST = build_sell(AT, &Lv.ord.color_ptr, sellC, false, h->mem_lean ? 0 : -1);
                m.resize(ST.entry.size());
                for (size_t i = 0; i < ST.entry.size(); i++) m[i] = ST.entry[i] >= 0 ? Lv.A_int_src[tsrc[ST.entry[i]]] : -1;
                up(er, Lv.mapAT.upload(m));
                if (!Lv.gs_on_transpose) { up(er, Lv.dAT.upload(ST)); Lv.gs_on_transpose = true; }
            });
            tasks.push_back([&, lv] {
                DeviceScope ds(h->device);
                Level& Lv = h->lv[lv];
                hipError_t& er = errs[2 * lv + 1];
                if (h->bs == 3 || lv == L - 1) up(er, Lv.d_Aval.upload(Lv.A.val));      // (scalar smoothed levels: the task above)
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
        if (!h->coarse_sparse && !h->coarse_schur && !h->union_m) HIPCHK(h->d_dense_pos.upload(pos));      // (a union keeps its members' block positions)
        HIPCHK(h->d_diag_idx.upload(dg));
    }
    if (!h->has_known && h->lhs_src.size() != (size_t)h->lv[0].A.nnz()) {
        h->lhs_src.resize((size_t)h->lv[0].A.nnz());
        std::iota(h->lhs_src.begin(), h->lhs_src.end(), 0);
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
            if (h->bs == 3) {
                HIPCHK(launch_gather_vals(Lv.bA.val.p, Lv.d_Aval.p, Lv.mapB.p, (size_t)Lv.bA.padded, st));
                HIPCHK(launch_gather_vals(Lv.bAT.val.p, Lv.d_Aval.p, Lv.mapBT.p, (size_t)Lv.bAT.padded, st));
            } else {
                HIPCHK(launch_gather_vals(const_cast<double*>(Lv.dA.view.val), Lv.d_Aval.p, Lv.mapA.p, (size_t)Lv.dA.padded, st));
                HIPCHK(launch_gather_vals(const_cast<double*>(Lv.dAT.view.val), Lv.d_Aval.p, Lv.mapAT.p, (size_t)Lv.dAT.padded, st));
            }
        }
    }
    // coarsest: solver.compute(Ac) (:47-48 / :253-254) -- the sparse factorisation on the host from the new values, or on the device the
    // Schur-complement factorisation resp. the dense inverse
    if (!h->coarse_sparse && !h->coarse_schur && !h->union_m && schur_wanted(h, h->nc, 2)) {
        // New values for an old pattern: this caller is a time stepper (05: a new matrix every flow step, 06: ten per step), and from here on the
        // coarse factorisation is what each of its steps pays -- the Schur-complement solver factors in a third of the dense inverse's time and
        // solves within a few us of it (csrc/smg_schur.hpp).  The plan is built once, now.
        bool planned = false;
        const int rc = coarse_plan_schur(h, h->lv[L - 1].A, &planned);
        if (rc) return rc;
        if (planned) {
            HIPCHK(hipStreamSynchronize(st));
            drop_graphs(h);                      // the captured coarse solves point at the dense inverse
            h->coarse_schur = true;
            h->d_Ainv.release(); h->d_Ainv32.release(); h->d_sympart.release(); h->d_dense_pos.release();
        } else {
            h->schur_declined = true;
            h->sch.release(); h->schur = SchurPlan();
        }
    }
    if (h->union_m > 0) {
        int rc = union_coarse_factor(h, h->lv[L - 1].d_Aval.p, false);
        if (rc) return rc;
    } else if (h->coarse_sparse) {
        Level& Lc = h->lv[L - 1];
        HIPCHK(hipMemcpyAsync(Lc.A.val.data(), Lc.d_Aval.p, Lc.A.val.size() * sizeof(double), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        int rc = coarse_factor_sparse(h, Lc.A, true);
        if (rc) return rc;
    } else if (h->coarse_schur) {
        HIPCHK(launch_schur_factor(h->sch.view, h->lv[L - 1].d_Aval.p, st));
        { int rc2 = schur_check_spd(h, h->nc); if (rc2) return rc2; }
    } else {
        const Level& Lc = h->lv[L - 1];
        HIPCHK(launch_dense_from_csr(h->d_Ainv.p, h->nc_pad, h->nc, Lc.d_Aval.p, h->d_dense_pos.p, (int)Lc.A.nnz(), st));
        DevBuf<double> work;
        HIPCHK(work.alloc((size_t)2 * h->nc_pad * 64 + 2 * 64 * 64));
        HIPCHK(launch_spd_inverse(h->d_Ainv.p, h->nc_pad, work.p, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    {
        int rc = refresh_tiled_values(h);   // the tiling plans hold copies of the level values
        if (rc) return rc;
    }
    h->host_stale = true;
    h->f32_valid = false;   // the fp32 copies are re-made from the new values when a mixed solve asks for them
    h->lam_valid = false;
    return SMG_OK;
}

// bring the host copies (mg[l].A, A_diag, Auk, A_int) up to date after a device-side re-precompute
int smg::refresh_host_values(smg_hierarchy* h)
{
    if (!h->host_stale) return SMG_OK;
    for (int lv = 0; lv < h->n_levels; lv++) {
        Level& Lv = h->lv[lv];
        HIPCHK(hipMemcpy(Lv.A.val.data(), Lv.d_Aval.p, Lv.A.val.size() * sizeof(double), hipMemcpyDeviceToHost));
        Lv.A_diag = diagonal(Lv.A);
        if (Lv.A_int_src.size() == Lv.A_int.val.size())
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
    // Two halves side by side.  The host half (the reference's sparse algebra, numberings; no device call in it) runs on a thread of its own;
    // this thread meanwhile brings the device up -- in a process that has not used HIP yet that alone is ~ 0.1 s of runtime
    // initialisation, then the stream, the library's code object -- and, when the caller's arrays are what level 0's panels will be
    // filled from (canonical rows, no constraints, a level big enough for launch_sell_fill), sends them ahead.
    const bool canonical = rowptr[0] == 0 && rows_strictly_ascending(n, rowptr, col);
    h->input_canonical = canonical;
    h->early0 = smg_hierarchy::EarlyUpload();
    Early0 e0;
    {
        const int L = h->n_levels;
        static const int early_host = env_int("SMG_EARLY_HOST", 1);
        bool maybe_block = h->block_mode != 0 && n_known == 0 && L >= 2 && n % 3 == 0;
        if (maybe_block) { Csr tmp; maybe_block = kron3_factor(h->lv[1].P_full, tmp); }      // (a scalar prolongation fails this within a row or two)
        if (early_host && canonical && n_known == 0 && !maybe_block && host_threads() > 1) {
            if (L >= 3 && use_rcm_order()) {
                e0.rcm_started = true;
                e0.rcm_t = std::thread([&e0, h, n, rowptr, col] {
                    const auto t0 = std::chrono::steady_clock::now();
                    const uint64_t k0 = pattern_key_arrays(n, true, 1, rowptr, col);
                    if (!(k0 == h->lv[0].ord_key && (int)h->lv[0].ord.perm.size() == n)) e0.rcm = rcm_order_arrays(n, rowptr, col);
                    e0.rcm_ms = 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                });
            }
        }
    }
    int host_rc = SMG_OK;
    std::string host_err;
    std::exception_ptr host_exc;
    const int L = h->n_levels;
    Handoff hand(L);
    std::thread host_half([&] {
        struct Over { Handoff& hd; ~Over() { hd.post_over(); } } over{hand};     // whatever happens, the other half stops waiting
        try {
            StageTimer tmh;
            Csr A = csr_from_arrays(n, n, rowptr, col, val);
            tmh.lap("precompute: input copy (sorted, duplicates summed)");
            host_rc = precompute_host(h, std::move(A), known, n_known, e0, hand);
            if (host_rc != SMG_OK) host_err = smg_last_error();
            tmh.lap("precompute: host half");
        } catch (...) { host_exc = std::current_exception(); }      // rethrown on the calling thread (smg_precompute's guard reports it)
    });
    struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } host_joiner{host_half};
    struct DropEarly { smg_hierarchy* h; ~DropEarly() { h->early0 = smg_hierarchy::EarlyUpload(); } } drop_early{h};
    int rc = ensure_device(h);
    int sym0 = -1;
    if (rc == SMG_OK) {
        DeviceScope dsc(h->device);
        hipError_t e = warm_device_code(h->stream);
        for (hipStream_t& a : h->aux) if (!a && e == hipSuccess) e = hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
        static const int fill_on = env_int("SMG_DEVICE_FILL", 1), fill_min = env_int("SMG_DEVICE_FILL_MIN", 200000), early_on = env_int("SMG_EARLY_UPLOAD", 1);
        if (e == hipSuccess && early_on && fill_on && canonical && n_known == 0 && L > 1 && n >= fill_min && h->block_mode != 3) {
            if (e == hipSuccess) e = h->early0.ptr.alloc((size_t)n + 1);
            if (e == hipSuccess) e = h->early0.col.alloc((size_t)rowptr[n]);
            if (e == hipSuccess) e = h->early0.val.alloc((size_t)rowptr[n]);
            if (e == hipSuccess) e = hipMemcpy(h->early0.ptr.p, rowptr, ((size_t)n + 1) * sizeof(int), hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(h->early0.col.p, col, (size_t)rowptr[n] * sizeof(int), hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(h->early0.val.p, val, (size_t)rowptr[n] * sizeof(double), hipMemcpyHostToDevice);
            h->early0.valid = e == hipSuccess;
            // ... and with the arrays there, A == A^T bit for bit (what lets the device fill level 0's panels by itself) is one short launch
            DevBuf<int> d_differs;
            int differs = 0;
            if (e == hipSuccess) e = d_differs.alloc(1);
            if (e == hipSuccess) e = launch_bit_symmetric(n, h->early0.ptr.p, h->early0.col.p, h->early0.val.p, d_differs.p, h->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(&differs, d_differs.p, sizeof(int), hipMemcpyDeviceToHost, h->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
            if (e == hipSuccess) sym0 = differs == 0 ? 1 : 0;
        }
        if (e != hipSuccess) rc = fail(SMG_ERR_HIP, "device bring-up: %s", hipGetErrorString(e));
        if (rc == SMG_OK) rc = device_begin(h);
        tmv.lap("precompute: device / stream / code object / early upload (beside the host half)");
        // the images, as the host half hands the levels over
        if (rc == SMG_OK && hand.wait_coarse()) {
            rc = coarse_images(h);
            tmv.lap("precompute: coarse factorisation (beside the host half)");
            // While this thread would only wait for the next finer level's numbering, it builds the sweep plans (overlapped tiles, wave Gauss-Seidel
            // pieces: host work + uploads, csrc/smg_cycle.cpp: prepare_level_plans) of the small levels whose images exist -- otherwise the first solve
            // pays for them.  A level that is ready is never kept waiting for more than the plan in hand.
            std::vector<int> plans_due;
            for (int lv = L - 2; lv >= 0 && rc == SMG_OK; lv--) {
                if (lv == L - 2 && !hand.wait_level(L - 1)) break;
                while (rc == SMG_OK && !plans_due.empty() && !hand.level_posted(lv)) { rc = prepare_level_plans(h, plans_due.front()); plans_due.erase(plans_due.begin()); }
                if (rc != SMG_OK || !hand.wait_level(lv)) break;
                rc = level_images(h, lv, lv == 0 ? sym0 : -1);
                if (lv > 0) plans_due.push_back(lv);
            }
        }
    }
    host_half.join();
    if (host_exc) std::rethrow_exception(host_exc);
    if (host_rc != SMG_OK) return fail(host_rc, "%s", host_err.c_str());
    if (rc != SMG_OK) return rc;
    tmv.lap("precompute: images of all levels (the finest after the host half)");
    {
        DeviceScope dsc(h->device);
        rc = finish_images(h);
        if (rc != SMG_OK) return rc;
    }
    tmv.lap("precompute: index maps");
    h->pre_key = key;
    h->precomputed = true;
    return SMG_OK;
}

extern "C" int smg_precompute(smg_hierarchy* h, int n, const int* rowptr, const int* col, const double* val,
                              const int* known, int n_known)
{
    return guarded("smg_precompute", [&]() { return smg_precompute_impl(h, n, rowptr, col, val, known, n_known); });
}
