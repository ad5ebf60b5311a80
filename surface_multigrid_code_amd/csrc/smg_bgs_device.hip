// smg_bgs_device.hip -- block Gauss-Seidel sweep for blocks of 64 right-hand-side columns (plan: smg_bgs.hpp / smg_bgs.cpp).
//
// One workgroup (4 waves) = one block of <= 64 rows of the level; one lane = one of 64 columns; the block's 64 x 64 iterate lives in LDS.
// Everything about a row is wave-uniform: its entry codes and values are picked out of the lanes of a coalesced metadata load
// (v_readlane, constant lane) into scalar registers; a gather is one 512-byte segment of the row-major n x k block.
// Bound: HBM.  Per row and 64-column block: 512 B of the iterate read once (the block's own rows, in bulk at the start), 512 B of b,
// 512 B written, and the block's rim gathered (~0.65 rows per row).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "smg_bgs.hpp"
#include "smg_device.hpp"
#include "smg_device_inl.hpp"

namespace smg {

// The phases of one block as one of its 4 waves sees them.  LP: row slots per phase and wave; NB: batches per row, S = 8 NB entry slots.
//  * Unit (block, phase, wave): 64 NB codes + values and 16 row words, NB + NB + 1 coalesced loads requested two phases ahead.
//  * The memory requests of a row slot -- its S gathers and its right-hand side -- go out one PHASE ahead: slot r of phase p + 1 is
//    requested when slot r of phase p has been stored (what is gathered from memory belongs to other blocks, which this launch does not
//    touch: the barrier between the phases does not concern it).  Every vector-memory instruction of the loop is unconditional -- a slot
//    that is not a gather requests row 0 of u (one line, resident in the CU's L1 after the first touch) and its value is never used: the
//    compiler then counts the requests in flight exactly and waits for this row's only (requests behind wave-uniform branches make it
//    wait for ALL of them at every use); the scheduling barriers keep the rows in program order.
//  * Neighbours inside the block: xs[local][lane], old or new as the order requires -- LDS is updated in place, the rows of one phase
//    (one vertex colour) share no entry, phases are separated by workgroup barriers.
template <int LP, int NB>
__device__ __forceinline__ void bgs_block(const int* urow, const int* ecol, const double* eval, const int unit0, const int nph, const int ent0,
                                          const double* b, double* u, const int ld, const size_t coff, double (*xs)[64], const int lane, const int wave)
{
    constexpr int S = NB * BGS_BATCH;
    struct Meta { int c[NB]; double v[NB]; int r; };
    auto load_meta = [&](const int p) {
        Meta M;
        const size_t uu = (size_t)p * BGS_WAVES + wave;
        const size_t e = (size_t)ent0 + uu * 64 * NB + lane;
#pragma unroll
        for (int h = 0; h < NB; h++) { M.c[h] = ecol[e + 64 * h]; M.v[h] = eval[e + 64 * h]; }
        M.r = urow[((size_t)unit0 + uu) * 16 + (lane & 15)];
        return M;
    };
    auto code_of = [](const Meta& M, const int e) { return __builtin_amdgcn_readlane(M.c[e / 64], e % 64); };
    auto val_of = [](const Meta& M, const int e) {
        const long long bits = __double_as_longlong(M.v[e / 64]);
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(bits & 0xffffffffll), e % 64), hi = (unsigned)__builtin_amdgcn_readlane((int)(bits >> 32), e % 64);
        return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    };
    double xg[LP][S], bb[LP];
    int rid[LP];
    auto fetch = [&](const int r, const Meta& M) {
        const int row = __builtin_amdgcn_readlane(M.r, r);
        rid[r] = row;
        bb[r] = b[(size_t)row * ld + coff];
#pragma unroll
        for (int t = 0; t < S; t++) {
            const int c = code_of(M, r * S + t);
            xg[r][t] = u[(size_t)(c > 0 ? c : 0) * ld + coff];
        }
    };
    Meta cur = load_meta(0), nxt = load_meta(nph > 1 ? 1 : 0);
#pragma unroll
    for (int r = 0; r < LP; r++) fetch(r, cur);
    for (int p = 0; p < nph; p++) {
        const Meta nn = load_meta(p + 2 < nph ? p + 2 : nph - 1);
#pragma unroll
        for (int r = 0; r < LP; r++) {
            __builtin_amdgcn_sched_barrier(0);
            int cc[S];
            double vv[S], rv[S];
#pragma unroll
            for (int t = 0; t < S; t++) { cc[t] = code_of(cur, r * S + t); vv[t] = val_of(cur, r * S + t); }
            // operands from the block's iterate (all S slots: a slot that is not a local entry reads local row 0 and drops it)
#pragma unroll
            for (int t = 0; t < S; t++) {
                int l = BGS_LOCAL0 - cc[t];
                l = (l < 0 || l > BGS_ROWS - 1) ? 0 : l;
                rv[t] = xs[l][lane];
            }
            // the row: products in ascending slot = ascending column of the bgs order, separate multiply and add
            double acc = 0.0, diag = 1.0;
#pragma unroll
            for (int t = 0; t < S; t++) {
                const double x = cc[t] >= 0 ? xg[r][t] : rv[t];
                const double nacc = acc + vv[t] * x;
                acc = (cc[t] >= 0 || cc[t] <= BGS_LOCAL0) ? nacc : acc;
                diag = cc[t] == BGS_DIAG ? vv[t] : diag;
            }
            const double nv = (bb[r] - acc) / diag;
            xs[__builtin_amdgcn_readlane(cur.r, 8 + r)][lane] = nv;
            u[(size_t)rid[r] * ld + coff] = nv;
            __builtin_amdgcn_sched_barrier(0);
            fetch(r, nxt);       // the same slot of the next phase (after the last phase: that phase's rows once more, never consumed)
        }
        __syncthreads();
        cur = nxt;
        nxt = nn;
    }
}

template <int LP>
__global__ __launch_bounds__(256) void k_bgs(const int* hdr, const int* brow, const int* urow, const int* ecol, const double* eval, int b_begin, int n_wg,
                                             const double* b, double* u, int ld, const int* done)
{
    __shared__ double xs[BGS_ROWS][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (load_flag(done)) return;       // after convergence the stream's launches write nothing (uniform over the launch)
    const int blk = b_begin + xcd_remap(blockIdx.x, n_wg);
    const size_t coff = (size_t)blockIdx.y * 64 + lane;
    const int* H = hdr + (size_t)blk * BGS_HDR;
    const int unit0 = H[0], nph = H[1], nb = H[3], ent0 = H[4];
    // the block's own rows, once: local row l of wave l % 4
    {
        double own[BGS_ROWS / BGS_WAVES];
#pragma unroll
        for (int q = 0; q < BGS_ROWS / BGS_WAVES; q++) own[q] = u[(size_t)brow[(size_t)blk * BGS_ROWS + wave + BGS_WAVES * q] * ld + coff];
#pragma unroll
        for (int q = 0; q < BGS_ROWS / BGS_WAVES; q++) xs[wave + BGS_WAVES * q][lane] = own[q];
    }
    __syncthreads();
    if (nb == 1) bgs_block<LP, 1>(urow, ecol, eval, unit0, nph, ent0, b, u, ld, coff, xs, lane, wave);
    else bgs_block<LP, 2>(urow, ecol, eval, unit0, nph, ent0, b, u, ld, coff, xs, lane, wave);
}

hipError_t launch_bgs(const BgsDev& P, int b_begin, int b_end, const double* b, double* u, int k, const Ctrl* ctrl, hipStream_t st)
{
    if (b_end <= b_begin) return hipSuccess;
    if (k % 64 != 0) return hipErrorInvalidValue;
    const int* done = ctrl ? &ctrl->done : never_done();
    const int n_wg = b_end - b_begin;
    const dim3 grid((unsigned)n_wg, (unsigned)(k / 64));
#define SMG_BGS_LAUNCH(LL) hipLaunchKernelGGL((k_bgs<LL>), grid, dim3(256), 0, st, P.hdr, P.brow, P.urow, P.ecol, P.eval, b_begin, n_wg, b, u, k, done)
    switch (P.lp) {
        case 4: SMG_BGS_LAUNCH(4); break;
        case 5: SMG_BGS_LAUNCH(5); break;
        case 6: SMG_BGS_LAUNCH(6); break;
        case 7: case 8: SMG_BGS_LAUNCH(8); break;
        default: return hipErrorInvalidValue;
    }
#undef SMG_BGS_LAUNCH
    return hipGetLastError();
}

}  // namespace smg
