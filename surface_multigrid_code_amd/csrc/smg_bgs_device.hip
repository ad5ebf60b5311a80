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

// D: rows whose gathers are in flight (row q + D is requested when row q has been stored; D - 1 <= BGS_RING).
template <int D>
__global__ __launch_bounds__(256) void k_bgs(const int* blk_ptr, const int* rows, const int* row_bat, const int* ecol, const double* eval,
                                             int b_begin, int b_end, int n_wg, const double* b, double* u, int ld, const int* done)
{
    __shared__ double ring[4][BGS_RING][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int stop = load_flag(done);
    const int bid = xcd_remap(blockIdx.x, n_wg);
    const int blk = __builtin_amdgcn_readfirstlane(b_begin + bid * 4 + wave);
    if (blk >= b_end) return;
    const size_t coff = (size_t)blockIdx.y * 64 + lane;
    const int q0 = blk_ptr[blk], q1 = blk_ptr[blk + 1];
    int c[D][BGS_BATCH], row[D], bat[D], nbat[D];
    double v[D][BGS_BATCH], xg[D][BGS_BATCH], bb[D];

    // request everything row q needs from memory into slot s (s is a constant after unrolling)
    auto fetch = [&](const int s, const int q) {
        row[s] = rows[q];
        bat[s] = row_bat[q];
        nbat[s] = row_bat[q + 1] - bat[s];
        const int* cp = ecol + (size_t)bat[s] * BGS_BATCH;
        const double* vp = eval + (size_t)bat[s] * BGS_BATCH;
#pragma unroll
        for (int t = 0; t < BGS_BATCH; t++) { c[s][t] = cp[t]; v[s][t] = vp[t]; }
        bb[s] = b[(size_t)row[s] * ld + coff];
#pragma unroll
        for (int t = 0; t < BGS_BATCH; t++) xg[s][t] = c[s][t] >= 0 ? u[(size_t)c[s][t] * ld + coff] : 0.0;
    };
    // one batch into the running sum, ascending slot = ascending column of the bgs order
    auto consume = [&](const int (&cc)[BGS_BATCH], const double (&vv)[BGS_BATCH], const double (&xx)[BGS_BATCH], double& acc, double& diag) {
#pragma unroll
        for (int t = 0; t < BGS_BATCH; t++) {
            if (cc[t] >= 0) acc += vv[t] * xx[t];
            else if (cc[t] == BGS_DIAG) diag = vv[t];
            else if (cc[t] != BGS_PAD) acc += vv[t] * ring[wave][BGS_RING0 - cc[t]][lane];
        }
    };
    auto compute = [&](const int s, const int q) {
        double acc = 0.0, diag = 1.0;
        consume(c[s], v[s], xg[s], acc, diag);
        for (int j = 1; j < nbat[s]; j++) {        // rows of more than 8 entries (few on mesh levels): their further batches, on the spot
            int c2[BGS_BATCH];
            double v2[BGS_BATCH], x2[BGS_BATCH];
            const int* cp = ecol + (size_t)(bat[s] + j) * BGS_BATCH;
            const double* vp = eval + (size_t)(bat[s] + j) * BGS_BATCH;
#pragma unroll
            for (int t = 0; t < BGS_BATCH; t++) { c2[t] = cp[t]; v2[t] = vp[t]; }
#pragma unroll
            for (int t = 0; t < BGS_BATCH; t++) x2[t] = c2[t] >= 0 ? u[(size_t)c2[t] * ld + coff] : 0.0;
            consume(c2, v2, x2, acc, diag);
        }
        const double nv = (bb[s] - acc) / diag;
        ring[wave][(q - q0) % BGS_RING][lane] = nv;
        if (!stop) u[(size_t)row[s] * ld + coff] = nv;
    };
#pragma unroll
    for (int s = 0; s < D; s++)
        if (q0 + s < q1) fetch(s, q0 + s);
    for (int q = q0; q < q1; q += D) {
#pragma unroll
        for (int s = 0; s < D; s++) {
            if (q + s < q1) {
                compute(s, q + s);
                if (q + s + D < q1) fetch(s, q + s + D);
            }
        }
    }
}

static int bgs_depth()
{
    static const int v = getenv("SMG_BGS_DEPTH") ? atoi(getenv("SMG_BGS_DEPTH")) : 2;
    return v;
}

hipError_t launch_bgs(const BgsDev& P, int b_begin, int b_end, const double* b, double* u, int k, const Ctrl* ctrl, hipStream_t st)
{
    if (b_end <= b_begin) return hipSuccess;
    if (k % 64 != 0) return hipErrorInvalidValue;
    const int* done = ctrl ? &ctrl->done : never_done();
    const int n_wg = (b_end - b_begin + 3) / 4;
    const dim3 grid((unsigned)n_wg, (unsigned)(k / 64));
#define SMG_BGS_LAUNCH(DD) hipLaunchKernelGGL((k_bgs<DD>), grid, dim3(256), 0, st, P.blk_ptr, P.rows, P.row_bat, P.ecol, P.eval, b_begin, b_end, n_wg, b, u, k, done)
    switch (bgs_depth()) {
        case 1: SMG_BGS_LAUNCH(1); break;
        case 3: SMG_BGS_LAUNCH(3); break;
        case 4: SMG_BGS_LAUNCH(4); break;
        default: SMG_BGS_LAUNCH(2); break;
    }
#undef SMG_BGS_LAUNCH
    return hipGetLastError();
}

}  // namespace smg
