// smg_wgs_device.hip -- wave Gauss-Seidel sweep on the Galerkin levels of decimated hierarchies (plan: smg_wgs.hpp / smg_wgs.cpp).
//
// One wavefront = one piece of <= 64 rows of the level, lane = row, KB <= 4 right-hand-side columns per lane.  The lane's row -- values and the
// byte offsets of its columns in the piece's LDS image -- is requested at once and stays in registers; the image (the piece's rows, then its
// rim) is gathered once; then the rows are updated phase by phase in place in LDS (a phase = rows that read none of each other, all of whose
// earlier neighbours sit in earlier phases), results stored straight to memory.  One wave per workgroup: LDS executes a wave's instructions in
// order, so a phase's stores are seen by the next phase's loads without a barrier.  Per row the products are added in ascending column of the
// wgs order with separate multiply and add: the oracle's lexicographic sweep on that numbering, bit for bit.
// Bound: latency -- launch, two dependent round trips (row numbers + header -> row + iterate), then `phases` x (longest active row) LDS reads.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "smg_device.hpp"
#include "smg_device_inl.hpp"
#include "smg_wgs.hpp"

namespace smg {

// One lane's row: `nbl` batches of 8 entry slots (values + packed byte offsets into the one-column image).  Only the batches the row needs are
// requested (per-lane guard: on a big level the padding of a piece's shorter rows would otherwise be a third of the matrix stream); the offsets of
// the others stay 0 -- a valid address, their LDS reads are issued but never consumed.
// Everything in front of the phases is straight-line code: a branch between two requests makes the compiler wait for the first one (the rim's gathers
// behind `if (slot < rim_pitch)` were one round trip EACH: 8.7 -> 5.x us per launch).  RIMI = rim slots per lane: the plan pads every piece's rim list to
// 64 RIMI row numbers (unused ones repeat the piece's first row: harmless gathers of a line that is needed anyway).
template <int NBMAX, int KB, int RIMI>
__global__ __launch_bounds__(64) void k_wgs(const int* __restrict__ hdr, const int* __restrict__ grow, const int* __restrict__ meta, const double* __restrict__ diag,
                                            const int* __restrict__ rim, const unsigned* __restrict__ eoff, const double* __restrict__ eval, int q_begin, int n_wg, int rim_pitch,
                                            const double* __restrict__ b_in, double* u_in, int ld, const int* done, int dbg_phases)
{
    __shared__ double xsd[(WGS_ROWS + RIMI * WGS_ROWS) * KB];       // the piece's rows, then its rim
    const char* xs = reinterpret_cast<const char*>(xsd);
    constexpr int S = NBMAX * WGS_BATCH;
    const int lane = threadIdx.x;
    const int stop = load_flag(done);     // after convergence the stream's launches write nothing; waited for at the first store only
    const int q = q_begin + xcd_remap(blockIdx.x, n_wg);      // neighbouring pieces on one XCD: shared rims meet in one L2
    const double* __restrict__ b = b_in + (size_t)blockIdx.y * KB;          // many columns: groups of KB ride in the grid's second dimension (the piece's row re-read from L2 per group, the vectors once)
    double* u = u_in + (size_t)blockIdx.y * KB;
    const int* H = hdr + (size_t)q * WGS_HDR;
    const size_t w = (size_t)q * WGS_ROWS + lane;
    // ---- round trip 1: the lane's row number, phase and length; the rim's row numbers; the piece header (scalar)
    const int gr = grow[w], mt = meta[w];
    const double dg = diag[w];
    int rg[RIMI];
    {
        const int* rq = rim + (size_t)q * rim_pitch + lane;
#pragma unroll
        for (int i = 0; i < RIMI; i++) rg[i] = rq[i * WGS_ROWS];
    }
    const int e0 = H[0], nph = H[3] < dbg_phases ? H[3] : dbg_phases;
    const int ph = mt & 0xffff, nbl = mt >> 16;
    // ---- round trip 2: the row (guarded per lane), the iterate of the piece and its rim, the right-hand side
    unsigned wo[S / 2];
    double v[S];
    {
        const unsigned* eo = eoff + ((size_t)e0 >> 1) + lane;
        const double* ev = eval + (size_t)e0 + lane;
#pragma unroll
        for (int bt = 0; bt < NBMAX; bt++) {
            if (bt < nbl) {
#pragma unroll
                for (int t = bt * 4; t < bt * 4 + 4; t++) wo[t] = eo[t * WGS_ROWS];
#pragma unroll
                for (int t = bt * 8; t < bt * 8 + 8; t++) v[t] = ev[t * WGS_ROWS];
            } else {
#pragma unroll
                for (int t = bt * 4; t < bt * 4 + 4; t++) wo[t] = 0u;
#pragma unroll
                for (int t = bt * 8; t < bt * 8 + 8; t++) v[t] = 0.0;
            }
        }
    }
    double own[KB], bv[KB], rv[RIMI][KB];
    gather_kb<KB, double>(u + (size_t)(gr >= 0 ? gr : 0) * ld, gr >= 0, own);
#pragma unroll
    for (int i = 0; i < RIMI; i++) gather_kb<KB, double>(u + (size_t)rg[i] * ld, true, rv[i]);
    gather_kb<KB, double>(b + (size_t)(gr >= 0 ? gr : 0) * ld, gr >= 0, bv);
#pragma unroll
    for (int c = 0; c < KB; c++) xsd[lane * KB + c] = own[c];
#pragma unroll
    for (int i = 0; i < RIMI; i++)
#pragma unroll
        for (int c = 0; c < KB; c++) xsd[(WGS_ROWS + i * WGS_ROWS + lane) * KB + c] = rv[i][c];
    __builtin_amdgcn_wave_barrier();      // one wave: LDS runs its instructions in order -- the image is complete for every later read
    // ---- the phases: rows that read none of each other, all earlier neighbours in earlier phases.  Batch bt + 2's operands are requested while batch
    // bt is added up (ascending column of the wgs order, separate multiply and add; padding: +0.0 times the row's own, finite, value).
    for (int p = 0; p < nph; p++) {
        if (ph == p) {
            double acc[KB];
#pragma unroll
            for (int c = 0; c < KB; c++) acc[c] = 0.0;
            double xa[2][WGS_BATCH][KB];
            auto request = [&](const int bt, double (&x)[WGS_BATCH][KB]) {
#pragma unroll
                for (int j = 0; j < WGS_BATCH; j++) {
                    const unsigned word = wo[(bt * WGS_BATCH + j) >> 1];
                    const unsigned off = ((j & 1) ? (word >> 16) : (word & 0xffffu)) * KB;
                    const double* xp = reinterpret_cast<const double*>(xs + off);
#pragma unroll
                    for (int c = 0; c < KB; c++) x[j][c] = xp[c];
                }
            };
            request(0, xa[0]);
            if constexpr (NBMAX > 1) request(1, xa[1]);
#pragma unroll
            for (int bt = 0; bt < NBMAX; bt++) {
                if (bt < nbl) {
#pragma unroll
                    for (int j = 0; j < WGS_BATCH; j++)
#pragma unroll
                        for (int c = 0; c < KB; c++) acc[c] += v[bt * WGS_BATCH + j] * xa[bt & 1][j][c];
                }
                if (bt + 2 < NBMAX) request(bt + 2, xa[bt & 1]);
            }
            double* up = u + (size_t)gr * ld;
#pragma unroll
            for (int c = 0; c < KB; c++) {
                const double out = (bv[c] - acc[c]) / dg;
                xsd[lane * KB + c] = out;
                if (!stop) up[c] = out;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// the pieces [q_begin, q_end) -- one piece colour -- of one sweep, in place on u (row-major n x k).  Up to 7 columns: one or two launches (groups of <= 4
// columns per lane); 8 and more: groups of 4 in the grid's second dimension, one launch (+ one for k % 4 columns).  The order of the sweep does not depend
// on k: a column-sharded solve (smg_solve_sharded) iterates bit for bit like the fused one.
hipError_t launch_wgs(const WgsDev& P, int q_begin, int q_end, const double* b, double* u, int k, const Ctrl* ctrl, hipStream_t st)
{
    if (q_end <= q_begin) return hipSuccess;
    if (k < 1 || k > 4 * 65535 || P.nb_max < 1 || P.nb_max > WGS_MAX_BATCHES) return hipErrorInvalidValue;
    if (P.rim_pitch != 2 * WGS_ROWS && P.rim_pitch != 4 * WGS_ROWS && P.rim_pitch != 7 * WGS_ROWS) return hipErrorInvalidValue;     // wgs_rim_pitch()
    const int* done = ctrl ? &ctrl->done : never_done();
    const int n_wg = q_end - q_begin;
    static const int dbg = getenv("SMG_DEBUG_WGS_PHASES") ? atoi(getenv("SMG_DEBUG_WGS_PHASES")) : 1 << 20;   // timing probe (wrong results)
    // Columns per lane: on a latency-bound level (few pieces per launch) a phase costs its instruction count, and four columns per lane are four times the
    // multiply-adds of one -- groups of 2 side by side in the grid's second dimension (ogre.obj, 5 038 rows: k = 4 in 367 us per cycle as one group of 4,
    // as two groups of 2 like k = 2); a big level re-reads its rows per group, so it takes groups of 4.
    static const int small_pieces = getenv("SMG_WGS_KB2_MAX_PIECES") ? atoi(getenv("SMG_WGS_KB2_MAX_PIECES")) : 1600;
    static const int kb_small = getenv("SMG_WGS_KB_SMALL") ? atoi(getenv("SMG_WGS_KB_SMALL")) : 1, kb_big = getenv("SMG_WGS_KB_BIG") ? atoi(getenv("SMG_WGS_KB_BIG")) : 4;
    const int kbp = P.n_pieces <= small_pieces ? kb_small : kb_big;
    for (int c0 = 0; c0 < k;) {
        int kb = k - c0, groups = 1;
        if (kb >= kbp) { groups = kb / kbp; kb = kbp; }
#define SMG_WGS_LAUNCH(NB, KB, RI) hipLaunchKernelGGL((k_wgs<NB, KB, RI>), dim3((unsigned)n_wg, (unsigned)groups), dim3(64), 0, st, P.hdr, P.grow, P.meta, P.diag, P.rim, P.eoff, P.eval, q_begin, n_wg, P.rim_pitch, b + c0, u + c0, k, done, dbg)
#define SMG_WGS_RI(NB, KB) do { if (P.rim_pitch == 2 * WGS_ROWS) SMG_WGS_LAUNCH(NB, KB, 2); else if (P.rim_pitch == 4 * WGS_ROWS) SMG_WGS_LAUNCH(NB, KB, 4); else SMG_WGS_LAUNCH(NB, KB, 7); } while (0)
#define SMG_WGS_NB(KB) do { if (P.nb_max <= 3) SMG_WGS_RI(3, KB); else if (P.nb_max <= 5) SMG_WGS_RI(5, KB); else SMG_WGS_RI(8, KB); } while (0)
        if (kb == 1) SMG_WGS_NB(1);
        else if (kb == 2) SMG_WGS_NB(2);
        else if (kb == 3) SMG_WGS_NB(3);
        else SMG_WGS_NB(4);
#undef SMG_WGS_NB
#undef SMG_WGS_RI
#undef SMG_WGS_LAUNCH
        c0 += kb * groups;
    }
    return hipGetLastError();
}

}  // namespace smg
