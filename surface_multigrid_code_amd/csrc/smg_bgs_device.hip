// smg_bgs_device.hip -- block Gauss-Seidel sweep for many right-hand-side columns (plan: smg_bgs.hpp / smg_bgs.cpp).
//
// One wavefront = (one block of <= 64 rows of the level, 16 columns).  The block's rows and its rim -- <= 128 rows x 16 columns -- are read
// once into LDS (128 B per row: a cache line); then the lanes are 16 rows x 4 columns: a lane keeps its row's entries (local indices into
// the LDS image + values, the diagonal, four right-hand sides) in registers and runs four passes, one per column it owns; the block's rows
// are updated unit by unit (<= 16 rows that read none of each other), in place in LDS, results stored straight to memory.
// No barrier (one wave, in-order LDS), no wave-uniform bookkeeping.  Bound: HBM -- per row and 16 columns 128 B of the iterate once, 128 B
// of b, 128 B stored, ~0.65 x 128 B of rim, and the matrix entries once per 16 columns.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "smg_bgs.hpp"
#include "smg_device.hpp"
#include "smg_device_inl.hpp"

namespace smg {


typedef double v2f64b __attribute__((ext_vector_type(2)));
typedef int v4i32b __attribute__((ext_vector_type(4)));

// the units of a block; NB batches of 8 entry slots per row
// PA: passes whose operands are read before the arithmetic starts; BGS_BR: requests per round of the image.  Measured at C3, k = 64 (level-0 sweep): (2, 32)
// 483 us, (1, 32) 485, (1 / 2 / 4, 16) 514 - 543, (1 / 2, 8) 542 - 575; metadata two units ahead instead of one: 505.
constexpr int BGS_PA = 2, BGS_BR = 32;
template <int NB>
__device__ __forceinline__ void bgs_units(const int* __restrict__ ugrow, const int* __restrict__ ulrow, const double* __restrict__ udiag, const int* __restrict__ eidx,
                                          const double* __restrict__ eval, const int unit0, const int nu, const int ent0, const double* __restrict__ b, double* u,
                                          const int ld, const int colbase, double* xs, const int xp, const int lane, const int xrows, const int* __restrict__ xrow_blk)
{
    constexpr int S = NB * BGS_BATCH;
    const int r16 = lane >> 2, c4 = lane & 3;
    struct Meta { int gr, lr; double dg; int idx[S]; double val[S]; double bv[4]; };
    auto load = [&](const int un) {
        Meta M;
        const size_t w = ((size_t)unit0 + un) * BGS_UROWS + r16;
        M.gr = ugrow[w]; M.lr = ulrow[w]; M.dg = udiag[w];
        const size_t e = (size_t)ent0 + ((size_t)un * BGS_UROWS + r16) * S;
#pragma unroll
        for (int t = 0; t < S; t += 4) { const v4i32b q = *reinterpret_cast<const v4i32b*>(eidx + e + t); M.idx[t] = q[0]; M.idx[t + 1] = q[1]; M.idx[t + 2] = q[2]; M.idx[t + 3] = q[3]; }
#pragma unroll
        for (int t = 0; t < S; t += 2) { const v2f64b q = *reinterpret_cast<const v2f64b*>(eval + e + t); M.val[t] = q[0]; M.val[t + 1] = q[1]; }
        const double* bp = b + (size_t)M.gr * ld + colbase + 4 * c4;
        const v2f64b b0 = *reinterpret_cast<const v2f64b*>(bp), b1 = *reinterpret_cast<const v2f64b*>(bp + 2);
        M.bv[0] = b0[0]; M.bv[1] = b0[1]; M.bv[2] = b1[0]; M.bv[3] = b1[1];
        return M;
    };
    Meta cur = load(0);           // requested before the image: its round trips overlap the image's
    {   // the block's rows and rim, once: 4 rows x 16 columns per request, ALL row numbers of a round of 32 requests (128 rows: one round for
        // the usual image) first, then all their values: two round trips for the whole image, not two per handful of rows
        const int lr4 = lane >> 4, lc = lane & 15;
        const int* xr = xrow_blk + lr4;
        for (int i0 = 0; i0 < xrows / 4; i0 += BGS_BR) {      // xrows is a multiple of 128 (unused local rows repeat the block's first row)
            int g[BGS_BR];
            double v[BGS_BR];
#pragma unroll
            for (int i = 0; i < BGS_BR; i++) g[i] = xr[4 * (i0 + i)];
#pragma unroll
            for (int i = 0; i < BGS_BR; i++) v[i] = u[(size_t)g[i] * ld + colbase + lc];
#pragma unroll
            for (int i = 0; i < BGS_BR; i++) xs[lc * xp + 4 * (i0 + i) + lr4] = v[i];
        }
    }
    for (int un = 0; un < nu; un++) {
        // the next unit's entries and right-hand sides travel while this one computes (blocks with 16 slots per row: fetched after the unit
        // instead -- they are few, and their second register set would cost every block of the level a third of its occupancy)
        Meta nxt;
        if constexpr (NB == 1) nxt = load(un + 1 < nu ? un + 1 : un);
        // all operands of the unit's four passes first, then the arithmetic, then the stores: the rows of a unit share no entry, so no pass
        // reads what another writes -- and the compiler, which cannot know that, would otherwise finish pass q before it starts q + 1
        double out[4];
#pragma unroll
        for (int q0 = 0; q0 < 4; q0 += BGS_PA) {
            double x[BGS_PA][S];
#pragma unroll
            for (int q = 0; q < BGS_PA; q++) {
                const double* xc = xs + (4 * c4 + q0 + q) * xp;
#pragma unroll
                for (int t = 0; t < S; t++) x[q][t] = xc[cur.idx[t]];
            }
#pragma unroll
            for (int q = 0; q < BGS_PA; q++) {
                double acc = 0.0;
#pragma unroll
                for (int t = 0; t < S; t++) acc += cur.val[t] * x[q][t];      // ascending column of the bgs order; padding: +0.0 times a finite value
                out[q0 + q] = (cur.bv[q0 + q] - acc) / cur.dg;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) xs[(4 * c4 + q) * xp + cur.lr] = out[q];
        double* up = u + (size_t)cur.gr * ld + colbase + 4 * c4;
        *reinterpret_cast<v2f64b*>(up) = (v2f64b){out[0], out[1]};
        *reinterpret_cast<v2f64b*>(up + 2) = (v2f64b){out[2], out[3]};
        if constexpr (NB != 1) nxt = load(un + 1 < nu ? un + 1 : un);
        cur = nxt;
    }
}

__global__ __launch_bounds__(64) void k_bgs(const int* __restrict__ hdr, const int* __restrict__ xrow, const int* __restrict__ ugrow, const int* __restrict__ ulrow,
                                            const double* __restrict__ udiag, const int* __restrict__ eidx, const double* __restrict__ eval, int b_begin, int n_wg, int kg, int xrows,
                                            const double* __restrict__ b, double* u, int ld, const int* done)
{
    extern __shared__ double xs[];     // 16 columns x (xrows + 1) doubles: the pitch of a column is odd, the 4 columns of a pass fall on different banks
    const int xp = xrows + 1;
    const int lane = threadIdx.x;
    if (load_flag(done)) return;       // after convergence the stream's launches write nothing (uniform over the launch)
    // consecutive logical ids -- the column groups of one block, then the next block of the launch -- stay on one XCD: a block's entries
    // and rim meet in one L2
    const int L = xcd_remap(blockIdx.x, n_wg);
    const int blk = b_begin + L / kg, colbase = (L % kg) * BGS_COLS;
    const int* H = hdr + (size_t)blk * BGS_HDR;
    const int unit0 = H[0], nu = H[1], nb = H[2], ent0 = H[3];
    if (nb == 1) bgs_units<1>(ugrow, ulrow, udiag, eidx, eval, unit0, nu, ent0, b, u, ld, colbase, xs, xp, lane, xrows, xrow + (size_t)blk * xrows);
    else bgs_units<2>(ugrow, ulrow, udiag, eidx, eval, unit0, nu, ent0, b, u, ld, colbase, xs, xp, lane, xrows, xrow + (size_t)blk * xrows);
}

hipError_t launch_bgs(const BgsDev& P, int b_begin, int b_end, const double* b, double* u, int k, const Ctrl* ctrl, hipStream_t st)
{
    if (b_end <= b_begin) return hipSuccess;
    if (k % BGS_COLS != 0) return hipErrorInvalidValue;
    const int* done = ctrl ? &ctrl->done : never_done();
    const int kg = k / BGS_COLS;
    const long n_wg = (long)(b_end - b_begin) * kg;
    if (n_wg > 0x7fffffffl) return hipErrorInvalidValue;
    const size_t lds = (size_t)BGS_COLS * (P.xrows + 1) * sizeof(double);
    hipLaunchKernelGGL(k_bgs, dim3((unsigned)n_wg), dim3(64), lds, st, P.hdr, P.xrow, P.ugrow, P.ulrow, P.udiag, P.eidx, P.eval, b_begin, (int)n_wg, kg, P.xrows, b, u, k, done);
    return hipGetLastError();
}

}  // namespace smg
