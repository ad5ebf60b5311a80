// smg_device_inl.hpp -- device-side helpers shared by the kernel files (smg_device.hip, smg_bsr3_device.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace smg {

__device__ __forceinline__ int xcd_remap(int bid, int nb)
{
    // block b runs on XCD b % 8 (observed dispatch order; used for locality only, never for correctness):
    // give every XCD one contiguous range of logical block ids.  Bijective for any nb.
    // XCD x owns q blocks, the first r XCDs one more: its range starts at x q + min(x, r)  (branch-free: this sits in front of
    // the first load of every launch)
    const int q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
    return xcd * q + (xcd < r ? xcd : r) + idx;
}

// `done` is never null (launchers substitute a zero word).  A per-lane (vector) load: the compiler waits for it only
// where the value is used -- a scalar load of the same word was waited for before anything else was issued.
__device__ __forceinline__ int load_flag(const int* done)
{
    return __builtin_nontemporal_load(done + (__builtin_amdgcn_mbcnt_lo(~0u, 0u) >> 6));   // + 0, opaque to the optimiser
}

// The KB columns of a neighbour are contiguous: one (KB = 2, 4) or two (KB = 3) wide loads instead of KB narrow ones -- the gathers
// are what a small-level launch waits for (element-aligned only: gfx950 global loads do not need more).  !use: zeros, no request.
template <int KB, typename T>
__device__ __forceinline__ void gather_kb(const T* px, bool use, T (&out)[KB])
{
    if constexpr (KB == 1) {
        out[0] = use ? px[0] : (T)0;
    } else if constexpr (KB == 3) {   // a 3-vector would be padded to 4 elements: the load would run past the row
        typedef T V2 __attribute__((ext_vector_type(2), aligned(sizeof(T))));
        V2 g = {(T)0, (T)0};
        T g2 = (T)0;
        if (use) { g = *reinterpret_cast<const V2*>(px); g2 = px[2]; }
        out[0] = g[0]; out[1] = g[1]; out[2] = g2;
    } else {
        typedef T VK __attribute__((ext_vector_type(KB), aligned(sizeof(T))));
        VK g;
#pragma unroll
        for (int q = 0; q < KB; q++) g[q] = (T)0;
        if (use) g = *reinterpret_cast<const VK*>(px);
#pragma unroll
        for (int q = 0; q < KB; q++) out[q] = g[q];
    }
}

// the zero word launchers substitute for a null convergence flag (defined in smg_device.hip)
const int* never_done();

}  // namespace smg
