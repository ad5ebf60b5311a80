// smg_coarse_device.hip -- the two triangular solves of the sparse coarse solver (smg_coarse.hpp) on gfx950, ONE launch each.
// One wavefront per row, rows in dependency order (forward: ascending, backward: descending); a row's lanes wait for the rows they read
// by polling the values themselves in HBM (agent-scope atomics; a word of all ones means "not there yet"), form their products, and the
// wave reduces them in a fixed order: the result does not depend on timing.  A wave only ever waits for rows whose wave has a lower
// launch index, i.e. was dispatched before it: no deadlock; the spins are bounded all the same (a stalled solve raises *err and every
// later wait gives up at once, instead of hanging the device).
#include <hip/hip_runtime.h>

#include "smg_device.hpp"
#include "smg_device_inl.hpp"

namespace smg {

// "not there yet": all bits set (a NaN no computation produces).  The value IS the flag: a consumer waits until the word differs -- one
// round trip per dependency instead of flag + value, and nothing to order between two stores.
__device__ __forceinline__ bool not_ready(double v) { return __double_as_longlong(v) == -1ll; }

template <bool BACK>
__global__ __launch_bounds__(256) void k_sptrsv(SparseCholDev F, const double* __restrict__ b, double* u, int ld, const int* done)
{
    if (load_flag(done)) return;      // the loop has ended: uniform over the launch
    const int lane = threadIdx.x & 63;
    const int r = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (r >= F.n) return;
    const int i = BACK ? F.n - 1 - r : r;
    const int* ptr = BACK ? F.cptr : F.rptr;
    const int* idx = BACK ? F.crow : F.rcol;
    const double* val = BACK ? F.cval : F.rval;
    double* sol = F.work + (BACK ? F.n : 0);          // forward: z in work[0, n); backward: x in work[n, 2n)
    const int p0 = ptr[i], p1 = ptr[i + 1];
    const double dg = F.diag[i];
    const int gi = F.perm[i];
    // forward: right-hand side b (caller order); backward: the forward solve's z_i, complete since that launch has ended
    const double rhs = BACK ? F.work[i] : b[(size_t)gi * ld];
    double acc = 0.0;
    for (int p = p0 + lane; p < p1; p += 64) {
        const int j = idx[p];
        const double v = val[p];
        // Relaxed agent-scope atomics (served at the device's coherence point, past the per-XCD L2s); acquire / release at agent scope
        // would write back and invalidate the whole L2 at every row: 15 us per row measured, 1000 x what this costs.
        double xj = __hip_atomic_load(sol + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (not_ready(xj)) {
            if ((++spins & 255) == 0 && (spins > (1 << 22) || __hip_atomic_load(F.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                __hip_atomic_store(F.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                xj = 0.0;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
            xj = __hip_atomic_load(sol + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        acc += v * xj;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if (lane == 0) {
        const double xi = (rhs - acc) / dg;
        // a result with the sentinel's bits (a NaN: the solve has failed anyway) must not stall its readers
        __hip_atomic_store(sol + i, not_ready(xi) ? __longlong_as_double(0x7ff8000000000000ll) : xi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (BACK) u[(size_t)gi * ld] = u[(size_t)gi * ld] + xi;      // u += solver.solve(B)   (reference src/mg_VCycle.cpp:199-200)
    }
}

hipError_t launch_sparse_coarse_solve(const SparseCholDev& F, const double* b, double* u, int k, const Ctrl* ctrl, hipStream_t st)
{
    if (F.n <= 0) return hipSuccess;
    const int* done = ctrl ? &ctrl->done : never_done();
    const int nb = (F.n + 3) / 4;
    for (int c = 0; c < k; c++) {
        hipError_t e = hipMemsetAsync(F.work, 0xFF, (size_t)2 * F.n * sizeof(double), st);      // every value "not there yet"
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_sptrsv<false>, dim3(nb), dim3(256), 0, st, F, b + c, u + c, k, done);
        hipLaunchKernelGGL(k_sptrsv<true>, dim3(nb), dim3(256), 0, st, F, b + c, u + c, k, done);
    }
    return hipGetLastError();
}

}  // namespace smg
