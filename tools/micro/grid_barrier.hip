// grid_barrier.hip -- what does one grid-wide phase boundary cost inside a persistent kernel on MI355X?
//   mode 0: workgroups on all XCDs, agent-scope arrive + sc1 poll, sc1 data hand-off
//   mode 1: workgroups of ONE XCD only (others exit), L2-local arrive/poll (workgroup-scope RMW), plain stores,
//           `buffer_inv sc0` (L1 invalidate) + plain loads for the hand-off
// Every phase: each workgroup writes phase-stamped values, barrier, reads another workgroup's values and checks them.
// All spins are bounded.  build: hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ int xcc_id()
{
    int v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15;
}

template <int MODE, int BS>
__global__ __launch_bounds__(BS) void k_phases(int n_phases, int target_xcd, unsigned* ctr, double* data, int* bad, int* xcd_hist,
                                                int payload)
{
    __shared__ int s_rank, s_n;
    const int xcd = xcc_id();
    if (threadIdx.x == 0) {
        atomicAdd(&xcd_hist[xcd], 1);
        s_rank = -1; s_n = 0;
        if (!(MODE == 1 || (MODE >= 4 && MODE <= 9))) { s_rank = blockIdx.x; s_n = gridDim.x; }
        else {
            // registration: everybody announces itself (agent scope); members of the target XCD take a rank and wait until the
            // whole grid has been seen, so the number of participants is known
            int r = -1;
            if (xcd == target_xcd) r = (int)__hip_atomic_fetch_add(ctr + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(ctr + 1, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            if (r >= 0) {
                unsigned spins = 0;
                while (__hip_atomic_load(ctr + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > (1u << 18)) { atomicAdd(bad + 1, 1); r = -1; break; }
                }
                s_n = (int)__hip_atomic_load(ctr + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            s_rank = r;
        }
    }
    __syncthreads();
    const int rank = s_rank, n = s_n;
    if (rank < 0) return;
    int nbad = 0;
    for (int p = 0; p < n_phases; p++) {
        // write: payload doubles per thread
        for (int q = 0; q < payload; q++) {
            double* dst = data + ((size_t)rank * BS + threadIdx.x) * payload + q;
            const double v = (double)(p * 1000003 + rank * 257 + (int)threadIdx.x + q);
            if (!(MODE == 1 || (MODE >= 4 && MODE <= 9))) __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else *dst = v;
        }
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned target = (unsigned)n * (unsigned)(p + 1);
            unsigned spins = 0;
            if (MODE == 0) {
                __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1u << 18)) { atomicAdd(bad + 1, 1); break; }
                }
            } else if (MODE == 2 || MODE == 3) {
                // arrivals and the release flag live in different cache lines; the last arriver publishes the phase number
                unsigned* flag = ctr + 64;
                bool last;
                if (MODE == 2) last = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == target - 1;
                else {
                    // two levels: one counter per XCD (32 words apart), then one per grid
                    unsigned* mine = ctr + 128 + 32 * xcd;
                    const unsigned per = (unsigned)n / 8u;   // the dispatcher deals workgroups round-robin over the XCDs
                    last = false;
                    if (__hip_atomic_fetch_add(mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == per * (unsigned)(p + 1) - 1)
                        last = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 8u * (unsigned)(p + 1) - 1;
                }
                if (last) __hip_atomic_store(flag, (unsigned)(p + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(p + 1)) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1u << 18)) { atomicAdd(bad + 1, 1); break; }
                }
            } else if (MODE == 5) {
                __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                for (;;) {
                    unsigned v;
                    asm volatile("buffer_inv sc1\n\tglobal_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(ctr) : "memory");
                    if (v >= target) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1u << 18)) { atomicAdd(bad + 1, 1); break; }
                }
            } else {
                __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                while (atomicCAS(ctr, 0xffffffffu, 0u) < target) {   // an RMW that cannot be folded into an (L1-cacheable) load
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1u << 18)) { atomicAdd(bad + 1, 1); break; }
                }
            }
        }
        __syncthreads();
        if (MODE == 1) asm volatile("buffer_inv sc0" ::: "memory");
        if (MODE == 4 || MODE == 5) asm volatile("buffer_inv sc1" ::: "memory");
        // read the neighbour's values
        const int other = (rank + 1) % n;   // a fixed neighbour: its lines are still in this CU's L1 from two phases ago
        for (int q = 0; q < payload; q++) {
            const double* src = data + ((size_t)other * BS + threadIdx.x) * payload + q;
            double v;
            if (MODE == 6) asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(src) : "memory");
            else if (MODE == 7) asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(src) : "memory");
            else if (MODE == 8) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(src) : "memory");
            else if (MODE == 9) asm volatile("global_load_dwordx2 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(src) : "memory");
            else if (!(MODE == 1 || MODE == 4 || MODE == 5)) v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else v = *src;
            if (v != (double)(p * 1000003 + other * 257 + (int)threadIdx.x + q)) nbad++;
        }
        // second barrier half is implied by the next phase's barrier only if nobody overwrites what a slow reader still needs:
        // double-buffer by phase parity instead of a second barrier
        data += (p & 1) ? -(ptrdiff_t)((size_t)gridDim.x * BS * payload) : (ptrdiff_t)((size_t)gridDim.x * BS * payload);
    }
    if (nbad) atomicAdd(bad, nbad);
}

int main(int argc, char** argv)
{
    const int n_phases = 200;
    unsigned* ctr; double* data; int* bad; int* hist;
    CHK(hipMalloc(&ctr, 4096)); CHK(hipMalloc(&bad, 8)); CHK(hipMalloc(&hist, 64));
    const int maxgrid = 2048 * 4, payload_max = 4;
    CHK(hipMalloc(&data, (size_t)2 * maxgrid * 256 * payload_max * 8));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    auto run = [&](int mode, int grid, int payload) {
        float best = 1e30f; int hb[2] = {0, 0}; int hh[16];
        for (int rep = 0; rep < 4; rep++) {
            CHK(hipMemset(ctr, 0, 4096)); CHK(hipMemset(bad, 0, 8)); CHK(hipMemset(hist, 0, 64));
            CHK(hipEventRecord(e0, 0));
            if (mode == 0) hipLaunchKernelGGL((k_phases<0, 256>), dim3(grid), dim3(256), 0, 0, n_phases, 0, ctr, data, bad, hist, payload);
            else if (mode == 4) hipLaunchKernelGGL((k_phases<4, 256>), dim3(grid), dim3(256), 0, 0, n_phases, 0, ctr, data, bad, hist, payload);
            else if (mode == 5) hipLaunchKernelGGL((k_phases<5, 256>), dim3(grid), dim3(256), 0, 0, n_phases, 0, ctr, data, bad, hist, payload);
            else if (mode == 6) hipLaunchKernelGGL((k_phases<6, 256>), dim3(grid), dim3(256), 0, 0, n_phases, 0, ctr, data, bad, hist, payload);
            else if (mode == 7) hipLaunchKernelGGL((k_phases<7, 256>), dim3(grid), dim3(256), 0, 0, n_phases, 0, ctr, data, bad, hist, payload);
            else if (mode == 8) hipLaunchKernelGGL((k_phases<8, 256>), dim3(grid), dim3(256), 0, 0, n_phases, 0, ctr, data, bad, hist, payload);
            else if (mode == 9) hipLaunchKernelGGL((k_phases<9, 256>), dim3(grid), dim3(256), 0, 0, n_phases, 0, ctr, data, bad, hist, payload);
            else if (mode == 1) hipLaunchKernelGGL((k_phases<1, 256>), dim3(grid), dim3(256), 0, 0, n_phases, 0, ctr, data, bad, hist, payload);
            else if (mode == 2) hipLaunchKernelGGL((k_phases<2, 256>), dim3(grid), dim3(256), 0, 0, n_phases, 0, ctr, data, bad, hist, payload);
            else if (mode == 3) hipLaunchKernelGGL((k_phases<3, 256>), dim3(grid), dim3(256), 0, 0, n_phases, 0, ctr, data, bad, hist, payload);
            else if (mode == 12) hipLaunchKernelGGL((k_phases<2, 1024>), dim3(grid), dim3(1024), 0, 0, n_phases, 0, ctr, data, bad, hist, payload);
            else if (mode == 13) hipLaunchKernelGGL((k_phases<3, 1024>), dim3(grid), dim3(1024), 0, 0, n_phases, 0, ctr, data, bad, hist, payload);
            CHK(hipEventRecord(e1, 0));
            CHK(hipDeviceSynchronize());
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
            CHK(hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost)); CHK(hipMemcpy(hh, hist, 64, hipMemcpyDeviceToHost));
        }
        unsigned c[4]; CHK(hipMemcpy(c, ctr, 16, hipMemcpyDeviceToHost));
        printf("mode %d grid %4d payload %d: %.2f us/phase  mismatches %d timeouts %d  participants %u  xcd hist", mode, grid, payload,
               1e3 * best / n_phases, hb[0], hb[1], (mode == 1 || (mode >= 4 && mode <= 9)) ? c[2] : (unsigned)grid);
        for (int i = 0; i < 8; i++) printf(" %d", hh[i]);
        printf("\n");
    };
    for (int payload : {0, 1, 4}) {
        run(0, 32, payload);
        for (int m : {1, 6, 7, 8, 9}) run(m, 256, payload);
    }
    return 0;
}
