// smg_coarse_device.hip -- the two triangular solves of the sparse coarse solver (smg_coarse.hpp) on gfx950, ONE launch each.
// One wavefront per row, rows in dependency order (forward: ascending, backward: descending); a row's lanes wait for the rows they read
// by polling the values themselves in HBM (agent-scope atomics; a word of all ones means "not there yet"), form their products, and the
// wave reduces them in a fixed order: the result does not depend on timing.
// FORWARD PROGRESS: a wave draws a ticket when it starts (one atomic per wave) and takes the row of that number in the dependency order, so
// a wave only ever waits for rows whose waves have STARTED before it -- whatever order the dispatcher starts workgroups in (rounds 3's
// version took the launch index and with it the assumption that workgroups start in index order).  What is still assumed: a resident wave
// keeps being scheduled (no preemption that parks it for good behind spinning ones).  Hence the bounded spins all the same: a wait that
// gives up raises *err, poisons its result with NaN, every later wait gives up at once, and the host sees SMG_ERR_HIP at the next
// synchronising call (smg_cycle.cpp: coarse_stall_check) -- a lost solve, never a hung device and never a silently wrong correction.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "smg_device.hpp"
#include "smg_device_inl.hpp"

namespace smg {

// "not there yet": all bits set (a NaN no computation produces).  The value IS the flag: a consumer waits until the word differs -- one
// round trip per dependency instead of flag + value, and nothing to order between two stores.
__device__ __forceinline__ bool not_ready(double v) { return __double_as_longlong(v) == -1ll; }

// KC columns at once (KC in {1, 2, 4, 8, 16}): a lane still owns ENTRIES of the row (p0 + lane, p0 + lane + 64, ...: the long rows of the top
// separators, hundreds of entries, set the length of the dependency chain and must stay spread over all 64 lanes) and carries KC running
// sums, one per column; the solution vectors are n x KC row-major, so the KC values a lane polls for an entry are one 8 KC-byte segment.
// Per (row, column): products summed per lane in entry order, then over the lanes in a fixed tree -- deterministic, and the SAME partition
// for every KC (a column's value does not depend on how many columns travel with it).  KC = 1 is the one-column kernel of round 3.
// The reference solves all columns in one solver.solve(B) (src/mg_VCycle.cpp:199-200); one PAIR of launches per <= 16 columns here
// (round 3: one pair and a memset per column).  Measured at 15 804 unknowns: 2.9 ms for one column, 7.3 ms for 8 (round 3: 23), 53 ms for 64
// as four passes of 16 (round 3: 186): the solves are bound by the NUMBER of agent-scope requests -- which bypass the L2s -- not by the depth
// of the dependency chain alone, so columns are not free; a 16-byte request carries two of them.  Also measured, all slower at 8 columns:
// lanes across columns, one entry slot per 64 / KC lanes (the long rows of the top separators take 64 / E times the rounds: 9.7 ms); KC
// eight-byte atomic loads per entry (the compiler waits after each: 29 ms); lane 0 storing the row's KC results one after the other (KC
// store round trips on every row's chain: 29 ms -> 11.6 with one store instruction); polling all KC values while spinning (11.6 -> 7.3 with
// the first value as the flag); one wave per (row, column) in one launch (11.3 ms).
// The KC values of one row of the solution block as ONE round trip: KC / 2 sixteen-byte loads at agent scope (sc1: past the per-XCD L2s, like
// the relaxed agent-scope atomic load of the one-column kernel), issued back to back, one wait.  (As KC separate __hip_atomic_load the
// compiler waits after each: 8 serial round trips at 8 columns, 10 x the one-column time, measured.)  Every double sits in its own aligned
// 8 bytes and carries its own "not there yet" pattern, so it does not matter whether the 16 bytes of a load arrive as one.
typedef double v2f64 __attribute__((ext_vector_type(2)));
template <int KC>
__device__ __forceinline__ void load_row_agent(const double* src, double (&x)[KC])
{
    if constexpr (KC == 1) {
        x[0] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        constexpr int NP = KC / 2;
        v2f64 r[NP];
#pragma unroll
        for (int q = 0; q < NP; q++) asm volatile("global_load_dwordx4 %0, %1, off offset:%2 sc1" : "=v"(r[q]) : "v"(src), "i"(16 * q) : "memory");
        if constexpr (NP == 1) asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0]) : : "memory");
        else if constexpr (NP == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0]), "+v"(r[1]) : : "memory");
        else if constexpr (NP == 4) asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : : "memory");
#pragma unroll
        for (int q = 0; q < NP; q++) { x[2 * q] = r[q][0]; x[2 * q + 1] = r[q][1]; }
    }
}

template <bool BACK, int KC>
__global__ __launch_bounds__(256) void k_sptrsv(SparseCholDev F, const double* __restrict__ b, double* u, int ld, const int* done)
{
    if (load_flag(done)) return;      // the loop has ended: uniform over the launch
    const int lane = threadIdx.x & 63;
    // The wave's place in the dependency order is a TICKET drawn when it starts, not its launch index: whatever order the dispatcher starts
    // workgroups in, every row this wave can wait for belongs to a wave that drew its ticket earlier, i.e. is running or done.
    // (one ticket of four rows per workgroup: 4 k atomics on one word instead of 16 k -- as one per wave they cost 0.5 of 2.9 ms)
    __shared__ int ticket;
    if (threadIdx.x == 0) ticket = __hip_atomic_fetch_add(F.err + 1 + (BACK ? 1 : 0), 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int r = __builtin_amdgcn_readfirstlane(ticket + (int)(threadIdx.x >> 6));
    if (r >= F.n) return;
    const int i = BACK ? F.n - 1 - r : r;
    const int* ptr = BACK ? F.cptr : F.rptr;
    const int* idx = BACK ? F.crow : F.rcol;
    const double* val = BACK ? F.cval : F.rval;
    double* sol = F.work + (BACK ? (size_t)F.n * KC : 0);          // forward: z in work[0, n KC); backward: x in work[n KC, 2 n KC)
    const int p0 = ptr[i], p1 = ptr[i + 1];
    const double dg = F.diag[i];
    const int gi = F.perm[i];
    double acc[KC];
#pragma unroll
    for (int c = 0; c < KC; c++) acc[c] = 0.0;
    for (int p = p0 + lane; p < p1; p += 64) {
        const int j = idx[p];
        const double v = val[p];
        // Relaxed agent-scope atomics (served at the device's coherence point, past the per-XCD L2s); acquire / release at agent scope
        // would write back and invalidate the whole L2 at every row: 15 us per row measured, 1000 x what this costs.
        const double* src = sol + (size_t)j * KC;
        // Wait on the row's FIRST value only (one 8-byte request per spin, as with one column: thousands of resident waves spin at any
        // time, and their requests compete with the ones the chain is waiting for -- polling all KC values made 8 columns 4 x slower
        // than one), then fetch the row; its other values were stored by the same instruction and are there, or a moment later.
        double xj[KC];
        bool lost = false;
        int spins = 0;
        auto give_up = [&]() {
            // A wait that gives up (the forward-progress assumption above failed, or another wait already did): the solve is lost. Raise
            // the flag -- every synchronising entry point checks it and fails with SMG_ERR_HIP -- and poison the result (NaN, not 0: a
            // caller that ignores the error code must not be handed a plausible but wrong correction).
            if ((++spins & 255) != 0 || !(spins > (1 << 22) || __hip_atomic_load(F.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) return false;
            __hip_atomic_store(F.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return true;
        };
        double x0 = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (not_ready(x0) && !lost) {
            if (give_up()) { lost = true; break; }
            __builtin_amdgcn_s_sleep(1);
            x0 = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if constexpr (KC == 1) xj[0] = x0;
        else {
            for (;;) {
                if (lost) break;
                load_row_agent<KC>(src, xj);
                bool all = true;
#pragma unroll
                for (int c = 0; c < KC; c++) all = all && !not_ready(xj[c]);
                if (all) break;
                if (give_up()) lost = true;
            }
        }
        if (lost) {
#pragma unroll
            for (int c = 0; c < KC; c++) xj[c] = __longlong_as_double(0x7ff8000000000000ll);
        }
#pragma unroll
        for (int c = 0; c < KC; c++) acc[c] += v * xj[c];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int c = 0; c < KC; c++) acc[c] += __shfl_down(acc[c], o, 64);
    // lane c finishes column c: one store instruction for the row's KC values (lane 0 storing them one after the other put KC store
    // round trips on the dependency chain of every row: 10 x the one-column time at 8 columns, measured)
    double mine = 0.0;
#pragma unroll
    for (int c = 0; c < KC; c++) { const double t = __shfl(acc[c], 0, 64); if (lane == c) mine = t; }
    if (lane < KC) {
        // forward: right-hand side b (caller order); backward: the forward solve's z_i, complete since that launch has ended
        const double rhs = BACK ? F.work[(size_t)i * KC + lane] : b[(size_t)gi * ld + lane];
        const double xi = (rhs - mine) / dg;
        // a result with the sentinel's bits (a NaN: the solve has failed anyway) must not stall its readers
        __hip_atomic_store(sol + (size_t)i * KC + lane, not_ready(xi) ? __longlong_as_double(0x7ff8000000000000ll) : xi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (BACK) u[(size_t)gi * ld + lane] = u[(size_t)gi * ld + lane] + xi;      // u += solver.solve(B)   (reference src/mg_VCycle.cpp:199-200)
    }
}

// columns per pass: the largest power of two <= min(k, 16); F.work holds 2 n sparse_coarse_work_cols(k) doubles
int sparse_coarse_work_cols(int k)
{
    int kc = 1;
    while (kc * 2 <= k && kc < 16) kc *= 2;
    return kc;
}

template <int KC>
static hipError_t sptrsv_pass(const SparseCholDev& F, const double* b, double* u, int ld, const int* done, hipStream_t st)
{
    hipError_t e = hipMemsetAsync(F.work, 0xFF, (size_t)2 * F.n * KC * sizeof(double), st);      // every value "not there yet"
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(F.err + 1, 0, 2 * sizeof(int), st);                                       // the two launches' ticket counters
    if (e != hipSuccess) return e;
    const int nb = (F.n + 3) / 4;
    hipLaunchKernelGGL((k_sptrsv<false, KC>), dim3(nb), dim3(256), 0, st, F, b, u, ld, done);
    hipLaunchKernelGGL((k_sptrsv<true, KC>), dim3(nb), dim3(256), 0, st, F, b, u, ld, done);
    return hipGetLastError();
}

hipError_t launch_sparse_coarse_solve(const SparseCholDev& F, const double* b, double* u, int k, const Ctrl* ctrl, hipStream_t st)
{
    if (F.n <= 0) return hipSuccess;
    const int* done = ctrl ? &ctrl->done : never_done();
    // blocks of 16 / 8 / 4 / 2 / 1 columns, largest first (each pass: one memset + one pair of launches)
    for (int c = 0; c < k;) {
        const int kc = sparse_coarse_work_cols(k - c);
        hipError_t e;
        switch (kc) {
            case 16: e = sptrsv_pass<16>(F, b + c, u + c, k, done, st); break;
            case 8: e = sptrsv_pass<8>(F, b + c, u + c, k, done, st); break;
            case 4: e = sptrsv_pass<4>(F, b + c, u + c, k, done, st); break;
            case 2: e = sptrsv_pass<2>(F, b + c, u + c, k, done, st); break;
            default: e = sptrsv_pass<1>(F, b + c, u + c, k, done, st); break;
        }
        if (e != hipSuccess) return e;
        c += kc;
    }
    return hipSuccess;
}

}  // namespace smg
