// smg_bgs_device.hip -- block-sequential Gauss-Seidel sweep for blocks of 64 right-hand-side columns (plan: smg_bgs.hpp / smg_bgs.cpp).
//
// One wavefront = one block of <= 64 rows of the level, walked row by row; one lane = one of 64 columns.  Everything about a row is
// wave-uniform: its entries arrive through the scalar cache (a row's batch of 8 codes + 8 values is 96 contiguous bytes), the choice
// "ring / gather / diagonal" is a scalar branch, a gather is one 512-byte segment of the row-major n x k block.
// Memory order: the wave's own stores to u and its later gathers from u are ordered by the program (same lane, same address); rows of
// other blocks read by this launch belong to other block colours and are not written by it.
// Bound: HBM.  Per row and 64-column block 512 B of b, 512 B of u written, and the gathers: the row's own old value (read by the earlier
// rows of its block: once from HBM, again from L2) plus the block's rim.  Arithmetic per row is ~8 dependent multiply-adds and a division,
// ~0.15 us; with 16-20 waves per CU that is an order of magnitude above what the memory delivers.
#include <hip/hip_runtime.h>

#include "smg_bgs.hpp"
#include "smg_device.hpp"
#include "smg_device_inl.hpp"

namespace smg {

// The walk of one block.  D: rows whose gathers are in flight (the requests of row r + D go out when row r has been stored; D <= BGS_RING,
// see smg_bgs.hpp); NB: batches per row of this block, S = 8 NB slots per row, C = 8 / NB rows per chunk of 64 slots.
//  * The metadata of a chunk -- 64 codes, 64 values, C rows -- is three coalesced loads (one slot per lane), requested two chunks ahead;
//    a row's codes and values are picked out of the lanes (v_readlane, constant lane): wave-uniform, in scalar registers, no dependent
//    scalar-load round trips in the walk (a first version that fetched them per row through the scalar cache spent 0.9 us per row).
//  * Every vector-memory instruction of the loop is unconditional -- a slot that is not a gather requests row 0 of u (one line, resident
//    in the CU's L1 after the first touch) and its value is never used: the compiler can then count the requests in flight exactly and
//    wait for row r's only (with requests behind wave-uniform branches it waits for ALL of them at every use: no prefetching at all).
//  * Blocks are padded to whole chunks with copies of their last row (smg_bgs.hpp): no tail code.  Requests beyond the last chunk
//    repeat rows of the last chunk and are never consumed.
template <int D, int NB>
__device__ __forceinline__ void bgs_walk(const int* prow, const int* ecol, const double* eval, const int m, const int chunk0, const double* b,
                                         double* u, const int ld, const size_t coff, double* ringw, const int lane)
{
    constexpr int S = NB * BGS_BATCH, C = 8 / NB;
    static_assert(C % D == 0, "the slot of a row must be a constant of the unrolled chunk");
    const int nch = (m + C - 1) / C;
    struct Meta { int c; double v; int r; };
    auto load_meta = [&](const int ch) {
        Meta M;
        const size_t e = ((size_t)chunk0 + ch) * 64 + lane;
        M.c = ecol[e];
        M.v = eval[e];
        M.r = prow[(size_t)ch * C + (lane & (C - 1))];
        return M;
    };
    auto code_of = [](const Meta& M, const int e) { return __builtin_amdgcn_readlane(M.c, e); };
    auto val_of = [](const Meta& M, const int e) {
        const long long bits = __double_as_longlong(M.v);
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(bits & 0xffffffffll), e), hi = (unsigned)__builtin_amdgcn_readlane((int)(bits >> 32), e);
        return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    };
    double xg[D][S], bb[D];
    int rid[D];
    // requests of row rr of a chunk into slot s: its S gathers and its right-hand side
    auto fetch = [&](const int s, const Meta& M, const int rr) {
        const int row = __builtin_amdgcn_readlane(M.r, rr);
        rid[s] = row;
        bb[s] = b[(size_t)row * ld + coff];
#pragma unroll
        for (int t = 0; t < S; t++) {
            const int c = code_of(M, rr * S + t);
            xg[s][t] = u[(size_t)(c > 0 ? c : 0) * ld + coff];
        }
    };
    Meta cur = load_meta(0), nxt = load_meta(nch > 1 ? 1 : 0);
#pragma unroll
    for (int s = 0; s < D; s++) fetch(s, cur, s);
    for (int ch = 0; ch < nch; ch++) {
        const Meta nn = load_meta(ch + 2 < nch ? ch + 2 : nch - 1);
#pragma unroll
        for (int r = 0; r < C; r++) {
            __builtin_amdgcn_sched_barrier(0);      // rows stay in program order: the requests of row r + D go out BEHIND row r's arithmetic
            const int s = r % D;
            int cc[S];
            double vv[S], rv[S];
#pragma unroll
            for (int t = 0; t < S; t++) { cc[t] = code_of(cur, r * S + t); vv[t] = val_of(cur, r * S + t); }
            // ring operands (all S slots: a slot that is not a ring entry reads ring slot 0 and drops it)
#pragma unroll
            for (int t = 0; t < S; t++) {
                int rs = BGS_RING0 - cc[t];
                rs = (rs < 0 || rs > BGS_RING - 1) ? 0 : rs;
                rv[t] = ringw[rs * 64];
            }
            // the row: products in ascending slot = ascending column of the bgs order, separate multiply and add
            double acc = 0.0, diag = 1.0;
#pragma unroll
            for (int t = 0; t < S; t++) {
                const double x = cc[t] >= 0 ? xg[s][t] : rv[t];
                const double nacc = acc + vv[t] * x;
                acc = (cc[t] >= 0 || cc[t] <= BGS_RING0) ? nacc : acc;
                diag = cc[t] == BGS_DIAG ? vv[t] : diag;
            }
            const double nv = (bb[s] - acc) / diag;
            const int pos = ch * C + r;
            ringw[((pos < m ? pos : m - 1) % BGS_RING) * 64] = nv;       // (copies of the last row rewrite its slot)
            u[(size_t)rid[s] * ld + coff] = nv;
            // requests of row r + D
            __builtin_amdgcn_sched_barrier(0);
            if (r + D < C) fetch(s, cur, r + D);
            else fetch(s, nxt, r + D - C);
        }
        cur = nxt;
        nxt = nn;
    }
}

template <int D>
__global__ __launch_bounds__(256) void k_bgs(const int* hdr, const int* prow, const int* ecol, const double* eval, int b_begin, int b_end, int n_wg,
                                             const double* b, double* u, int ld, const int* done)
{
    __shared__ double ring[4][BGS_RING][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (load_flag(done)) return;       // after convergence the stream's launches write nothing (60 us launches: the wait costs nothing here)
    const int bid = xcd_remap(blockIdx.x, n_wg);
    const int blk = __builtin_amdgcn_readfirstlane(b_begin + bid * 4 + wave);
    if (blk >= b_end) return;
    const size_t coff = (size_t)blockIdx.y * 64 + lane;
    const int roff = hdr[blk * 4 + 0], m = hdr[blk * 4 + 1], chunk0 = hdr[blk * 4 + 2], nb = hdr[blk * 4 + 3];
    double* ringw = &ring[wave][0][lane];
    if (nb == 1) bgs_walk<D, 1>(prow + roff, ecol, eval, m, chunk0, b, u, ld, coff, ringw, lane);
    else bgs_walk<(D >= 2 ? D / 2 : 1), 2>(prow + roff, ecol, eval, m, chunk0, b, u, ld, coff, ringw, lane);
}

static int bgs_depth()
{
    static const int v = getenv("SMG_BGS_DEPTH") ? atoi(getenv("SMG_BGS_DEPTH")) : 4;
    return v;
}

hipError_t launch_bgs(const BgsDev& P, int b_begin, int b_end, const double* b, double* u, int k, const Ctrl* ctrl, hipStream_t st)
{
    if (b_end <= b_begin) return hipSuccess;
    if (k % 64 != 0) return hipErrorInvalidValue;
    const int* done = ctrl ? &ctrl->done : never_done();
    const int n_wg = (b_end - b_begin + 3) / 4;
    const dim3 grid((unsigned)n_wg, (unsigned)(k / 64));
#define SMG_BGS_LAUNCH(DD) hipLaunchKernelGGL((k_bgs<DD>), grid, dim3(256), 0, st, P.hdr, P.prow, P.ecol, P.eval, b_begin, b_end, n_wg, b, u, k, done)
    switch (bgs_depth()) {
        case 1: SMG_BGS_LAUNCH(1); break;
        case 2: SMG_BGS_LAUNCH(2); break;
        case 8: SMG_BGS_LAUNCH(8); break;
        default: SMG_BGS_LAUNCH(4); break;
    }
#undef SMG_BGS_LAUNCH
    return hipGetLastError();
}

}  // namespace smg
