// smg_union_device.hip -- independent meshes in ONE handle (smg_hierarchy_create_union, csrc/smg_union.cpp): the kernels only a union needs.
//
// north_star: "independent RHS columns / independent meshes shard".  On one GPU M handles do not overlap (a hipGraphLaunch of ~50 kernel nodes is
// enqueued under a process-wide lock), so many small meshes go into one block-diagonal handle: every launch serves all of them.  What must NOT be
// shared is what the reference does per mesh -- one min_quad_with_fixed_mg_solve loop each (src/min_quad_with_fixed_mg.cpp:105-134): its own residual
// norm, its own history, its own break test -- and the coarse solve (the inverse of a block-diagonal matrix is block-diagonal: m small inverses, not one
// (sum n_i)^2 matrix).  Here: the per-member sum of squares and decision, the iterate of a converged member put back after every further cycle, the
// block-diagonal coarse product.
#include <hip/hip_runtime.h>

#include "smg_device.hpp"
#include "smg_device_inl.hpp"

namespace smg {

// u[row, 0..KB) += Ainv_i[row - row0_i, :] * b[row0_i .., 0..KB): one wavefront per row, the arithmetic of k_dense_gemv_add (16 B per lane per load,
// shuffle-tree reduction) on the member's own inverse.  The padding columns of an inverse are zero, and what they would multiply -- the next member's rows -- is
// not read (0 x NaN of a diverging neighbour would be NaN): members are numerically isolated.  mrow0 has m + 1 entries.
template <int KB>
__global__ __launch_bounds__(256) void k_blockdiag_gemv_add(const double* __restrict__ Ainv, const int* __restrict__ row_member, const long long* __restrict__ moff,
                                                            const int* __restrict__ mlda, const int* __restrict__ mrow0, int n, const double* __restrict__ b, double* u, int ld,
                                                            const int* done)
{
    const int stop = load_flag(done);
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const int i = row_member[row];
    const int lda = mlda[i], r0 = mrow0[i], ni = mrow0[i + 1] - r0;
    typedef double V2 __attribute__((ext_vector_type(2)));
    const V2* a2 = reinterpret_cast<const V2*>(Ainv + moff[i] + (size_t)(row - r0) * lda);
    const double* bb = b + (size_t)r0 * ld;
    double acc[KB];
#pragma unroll
    for (int q = 0; q < KB; q++) acc[q] = 0.0;
    const int n2 = lda >> 1;
#pragma unroll 4
    for (int jj = lane; jj < n2; jj += 64) {
        const V2 a = a2[jj];
#pragma unroll
        for (int q = 0; q < KB; q++) {
            acc[q] += a.x * (2 * jj < ni ? bb[(size_t)(2 * jj) * ld + q] : 0.0);
            acc[q] += a.y * (2 * jj + 1 < ni ? bb[(size_t)(2 * jj + 1) * ld + q] : 0.0);
        }
    }
#pragma unroll
    for (int q = 0; q < KB; q++) {
        double s = acc[q];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
        if (lane == 0 && !stop) u[(size_t)row * ld + q] = u[(size_t)row * ld + q] + s;
    }
}

hipError_t launch_blockdiag_gemv_add(const UnionDev& U, const double* Ainv, int n, const double* b, double* u, int k, const Ctrl* ctrl, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    const int* done = ctrl ? &ctrl->done : never_done();
    const int nb = (n + 3) / 4;
    for (int c0 = 0; c0 < k; c0 += 4) {
        const int kb = (k - c0) < 4 ? (k - c0) : 4;
        switch (kb) {
            case 1: hipLaunchKernelGGL((k_blockdiag_gemv_add<1>), dim3(nb), dim3(256), 0, st, Ainv, U.crow_member, U.moff, U.mlda, U.mrow0, n, b + c0, u + c0, k, done); break;
            case 2: hipLaunchKernelGGL((k_blockdiag_gemv_add<2>), dim3(nb), dim3(256), 0, st, Ainv, U.crow_member, U.moff, U.mlda, U.mrow0, n, b + c0, u + c0, k, done); break;
            case 3: hipLaunchKernelGGL((k_blockdiag_gemv_add<3>), dim3(nb), dim3(256), 0, st, Ainv, U.crow_member, U.moff, U.mlda, U.mrow0, n, b + c0, u + c0, k, done); break;
            default: hipLaunchKernelGGL((k_blockdiag_gemv_add<4>), dim3(nb), dim3(256), 0, st, Ainv, U.crow_member, U.moff, U.mlda, U.mrow0, n, b + c0, u + c0, k, done); break;
        }
    }
    return hipGetLastError();
}

// ss[i] = sum over member i's rows (all k columns) of r^2: one workgroup per member, every thread a fixed share of the member's rows in a fixed order,
// then a fixed tree over the threads -- deterministic (the summation order is not Eigen's linear one: <= 1e-13 relative, as for the single-mesh norm).
__global__ __launch_bounds__(1024) void k_union_sumsq(const double* __restrict__ r, const double* __restrict__ u, double* __restrict__ zsave, const int* __restrict__ rows,
                                                       const int* __restrict__ rptr, int k, double* ss, const int* done)
{
    __shared__ double red[1024];
    const int i = blockIdx.x, t = threadIdx.x;
    if (load_flag(done)) return;
    double s = 0.0;
    for (int p = rptr[i] + t; p < rptr[i + 1]; p += 1024) {
        const size_t o = (size_t)rows[p] * k;
        for (int c = 0; c < k; c++) { s += r[o + c] * r[o + c]; zsave[o + c] = u[o + c]; }      // ... and the iterate before the cycle that follows (k_union_restore)
    }
    red[t] = s;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (t < o) red[t] += red[t + o];
        __syncthreads();
    }
    if (t == 0) ss[i] = red[0];
}

// The break test of every member's own loop (src/min_quad_with_fixed_mg.cpp:108-116): residual recorded, `res < tol` ends THAT member's loop; the handle's
// loop ends when every member's has.  The handle's own history keeps the Frobenius norm over all members (what a caller of the plain API reads).
// A member whose residual is not finite ends ITS loop (mdone = 2: failed; smg_union_get_history reports it, the handle's `converged` is 0); the others
// go on, and the failed member's sum no longer enters the handle's norm.
__global__ void k_union_decide(Ctrl* ctrl, const double* ss, int m, double* his, int* nhis, int* mdone, int cap)
{
    if (threadIdx.x != 0 || ctrl->done) return;
    const double tol = ctrl->tol;
    double tot = 0.0;
    int all = 1;
    for (int i = 0; i < m; i++) {
        if (!mdone[i]) {
            const double r = sqrt(ss[i]);
            const int j = nhis[i];
            if (j < cap) his[(size_t)i * cap + j] = r;
            nhis[i] = j + 1;
            if (!(r == r) || r > 1.7e308) mdone[i] = 2;
            else if (r < tol) mdone[i] = 1;
        }
        if (mdone[i] != 2) tot += ss[i];
        if (!mdone[i]) all = 0;
    }
    const double r = sqrt(tot);
    const int j = ctrl->n_his;
    if (j < ctrl->his_cap) ctrl->r_his[j] = r;
    ctrl->n_his = j + 1;
    ctrl->r_prev = ctrl->r_last; ctrl->r_last = r;
    ctrl->sumsq = tot;
    if (all || ctrl->status != 0) ctrl->done = 1;
}

// after a V-cycle: the rows of every member whose loop has ended get back the iterate they had when it ended (zsave = the iterate before this cycle;
// a member that ended earlier was restored after every cycle since, so zsave holds its final iterate as well)
__global__ __launch_bounds__(256) void k_union_restore(double* u, const double* __restrict__ zsave, const int* __restrict__ rows, const int* __restrict__ rptr,
                                                       const int* __restrict__ mdone, int k, const int* done)
{
    const int i = blockIdx.y;
    if (!mdone[i] || load_flag(done)) return;
    const int p = rptr[i] + blockIdx.x * 256 + threadIdx.x;
    if (p >= rptr[i + 1]) return;
    const size_t o = (size_t)rows[p] * k;
    for (int c = 0; c < k; c++) u[o + c] = zsave[o + c];
}

hipError_t launch_union_sumsq_decide(const UnionDev& U, const double* r, const double* u, int k, Ctrl* ctrl, hipStream_t st)
{
    hipLaunchKernelGGL(k_union_sumsq, dim3(U.m), dim3(1024), 0, st, r, u, U.zsave, U.rows, U.rptr, k, U.ss, &ctrl->done);
    hipLaunchKernelGGL(k_union_decide, dim3(1), dim3(64), 0, st, ctrl, U.ss, U.m, U.his, U.nhis, U.done, U.his_cap);
    return hipGetLastError();
}
hipError_t launch_union_restore(const UnionDev& U, double* u, int k, const Ctrl* ctrl, hipStream_t st)
{
    hipLaunchKernelGGL(k_union_restore, dim3((unsigned)((U.max_rows + 255) / 256), (unsigned)U.m), dim3(256), 0, st, u, U.zsave, U.rows, U.rptr, U.done, k, &ctrl->done);
    return hipGetLastError();
}
}  // namespace smg
