// smg_internal.hpp -- what the translation units behind the C ABI (include/smg.h) share with each other:
//   smg_capi.cpp        errors, the handle (container, setters, introspection), profc mirror, mesh numerics shims
//   smg_precompute.cpp  min_quad_with_fixed_mg_precompute: host sparse algebra, device images, value-only re-precompute, assembly
//   smg_cycle.cpp       mg_VCycle / min_quad_with_fixed_mg_solve: launch sequence of a cycle, graph cache, outer loop, pieces
//   smg_hierarchy_io.cpp mg_precompute / mg_precompute_block builders, point queries, .smgh files
// Nothing here is part of the ABI.
#pragma once
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <new>
#include <string>

#include "../../include/smg.h"
#include "smg_hier.hpp"
#include "smg_mesh.hpp"

namespace smg {

// ---- errors: the message of the calling thread's last failure (smg_last_error) -----------------------------------------------
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

#define HIPCHK(expr)                                                                                          \
    do {                                                                                                      \
        hipError_t e__ = (expr);                                                                              \
        if (e__ != hipSuccess) return ::smg::fail(SMG_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e__)); \
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

// a handle under construction: destroyed unless release()d (an entry point that builds a hierarchy must not leak it when a later
// step fails or throws)
struct HierarchyOwner {
    smg_hierarchy* h = nullptr;
    explicit HierarchyOwner(smg_hierarchy* p) : h(p) {}
    HierarchyOwner(const HierarchyOwner&) = delete;
    HierarchyOwner& operator=(const HierarchyOwner&) = delete;
    ~HierarchyOwner() { if (h) smg_hierarchy_destroy(h); }
    smg_hierarchy* release() { smg_hierarchy* p = h; h = nullptr; return p; }
};

// ---- device plumbing -----------------------------------------------------------------------------------------------------------
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

inline int env_int(const char* name, int dflt)
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

int ensure_device(smg_hierarchy* h);            // first use of the device by a handle: stream, control block
void drop_graphs(smg_hierarchy* h);             // the cached hipGraphs no longer describe the handle
// the sweep plans a small level would want for a default solve (one column, the handle's pre / post sweeps), built ahead of the first solve (smg_cycle.cpp)
int prepare_level_plans(smg_hierarchy* h, int lv);
int check_ready(const smg_hierarchy* h, const char* who);

// ---- profc mirror (PROFC_NODE, reference src/profc.h:9-13), timed on the GPU timeline -----------------------------------------
int prof_scope_id(smg_hierarchy* h, const char* name);
hipEvent_t prof_event(smg_hierarchy* h);
void prof_collect(smg_hierarchy* h);
struct ProfGuard {
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

// ---- container helpers -----------------------------------------------------------------------------------------------------------
// CSR/CSC array sanity: monotone pointers, indices in range.  Returns an error string or nullptr.
const char* check_compressed(int n_major, int n_minor, const int* ptr, const int* idx);
int set_prolong(smg_hierarchy* h, int lv, Csr&& P);   // mg[lv].P = PT^T = P_full = P  (reference src/mg_precompute.cpp:74-76)
Mesh wrap_mesh(const double* V, int nV, const int* F, int nF);

// ---- precompute (smg_precompute.cpp) ------------------------------------------------------------------------------------------------
int spectral_bounds(smg_hierarchy* h);          // Gershgorin bounds of D^-1 A on every smoothed level (Chebyshev-Jacobi)
int ensure_spectral_bounds(smg_hierarchy* h);   // ... only when a level is smoothed that way and the values changed
int refresh_host_values(smg_hierarchy* h);      // host copies of mg[l].A after a device-side value-only re-precompute
int ensure_P_int(smg_hierarchy* h, int lv);     // ... and P / PT of the level
int ensure_A_int(smg_hierarchy* h, int lv);     // the level matrix in the internal numbering on the host (built on demand where the device filled the panels)

// ---- independent meshes in one handle (smg_union.cpp) ---------------------------------------------------------------------------------
int union_coarse_factor(smg_hierarchy* h, const double* d_vals, bool first);   // the members' dense inverses, side by side (first: + the bookkeeping)
int union_begin_solve(smg_hierarchy* h, int k);                                // per-member solve state

// ---- cycle (smg_cycle.cpp) -----------------------------------------------------------------------------------------------------------
enum { LV_GS = 0, LV_JACOBI = 1, LV_CHEBY = 2 };
int level_kind(const smg_hierarchy* h, int lv);   // the smoother of a level under the handle's selection
inline bool level_is_jacobi(const smg_hierarchy* h, int lv) { return level_kind(h, lv) != LV_GS; }   // needs the second iterate buffer

int refresh_tiled_values(smg_hierarchy* h);     // the overlapped-tiling plans hold copies of the level values (value-only re-precompute)
void drop_tiled(smg_hierarchy* h);              // ... and describe one matrix image: dropped when the images are rebuilt

}  // namespace smg
