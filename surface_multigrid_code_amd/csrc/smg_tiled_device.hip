// smg_tiled_device.hip -- relax(iters) of a latency-bound level in ONE launch: overlapped tiling of the multi-colour Gauss-Seidel
// sweeps (plan and rationale: smg_tiled.hpp).  One workgroup per tile; the iterate of the extended tile (tile + halo rings) lives in
// LDS, the phases (sweep, colour) are separated by workgroup barriers, every row update is the expression of k_sell<SELL_GS> on the
// same operands in the same order -- bit-identical results.  fp64, one column per launch.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "smg_device.hpp"
#include "smg_device_inl.hpp"
#include "smg_tiled.hpp"

namespace smg {

constexpr int TILED_NT = TILED_THREADS;   // most threads of a tile's workgroup (the plan says how many: TiledDev::threads)

// x -> y (x != y).  ld: columns of the row-major blocks; the launch handles KB adjacent columns (pointers are offset to the first).
// Thread t owns row t of every colour's panel (the plan guarantees m_c <= threads): the matrix rows are loaded ONCE, all requests
// in flight together, and stay in registers for all sweeps -- after the prologue the phases touch nothing but LDS.
template <int NC, int W, int KB>
__global__ __launch_bounds__(TILED_NT) void k_tiled_gs(const int* __restrict__ hdr, const int* __restrict__ ext_rows, const int* __restrict__ pcol,
                                                       const double* __restrict__ pval, const int* __restrict__ prow, const double* __restrict__ pdiag, const double* __restrict__ b,
                                                       const double* __restrict__ x, double* __restrict__ y, int ld, int nc, int sweeps, const int* done, int dbg_phases)
{
    extern __shared__ double xs[];      // the extended tile's iterate: n_ext x KB, row-major like the global blocks
    __shared__ int Hs[TILED_HDR];
    const int tid = threadIdx.x;
    if (tid < TILED_HDR) Hs[tid] = hdr[(size_t)blockIdx.x * TILED_HDR + tid];
    const int stop = load_flag(done);
    __syncthreads();
    const int ext_off = Hs[0], n_ext = Hs[1], w = Hs[2];
    const int P = sweeps * nc;
    // The prologue is two round trips behind the header, whatever the size of the tile: (A) every index the workgroup needs -- the rows of the image (U per
    // thread at a time), the panels' rows, columns and values -- requested together; (B) the gathers they address: the image's x, the right-hand sides.
    // (Until round 6 this was a loop of "index, then gather" per 512 rows of the image, then the panel loads, then the right-hand sides: 2 x (n_ext / 512) + 3
    // dependent round trips in front of the first phase.)
    constexpr int U = 4;
    const int nt = (int)blockDim.x;
    int er[U];
    // (unconditional loads at clamped -- valid -- indices: a predicated vector load becomes a branch, and a branch between two requests a wait)
    auto image_indices = [&](const int i0) {
#pragma unroll
        for (int t = 0; t < U; t++) { const int i = i0 + t * nt; er[t] = ext_rows[ext_off + (i < n_ext ? i : n_ext - 1)]; }
    };
    auto image_gather = [&](const int i0) {
        double g[U][KB];
#pragma unroll
        for (int t = 0; t < U; t++) gather_kb<KB, double>(x + (size_t)er[t] * ld, true, g[t]);
#pragma unroll
        for (int t = 0; t < U; t++) {
            const int i = i0 + t * nt;
            if (i < n_ext) {
#pragma unroll
                for (int q = 0; q < KB; q++) xs[i * KB + q] = g[t][q];
            }
        }
    };
    // (A)
    image_indices(tid);
    int cR[NC][W], gR[NC];
    double vR[NC][W], bR[NC][KB], dR[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        gR[c] = -1; dR[c] = 1.0;
#pragma unroll
        for (int j = 0; j < W; j++) { cR[c][j] = 0; vR[c][j] = 0.0; }      // slots beyond the tile's width, lanes without a row: +0.0 times the image's first (finite) value
        if (c < nc) {
            const int* C = Hs + 4 + c * TILED_CSTRIDE;
            const int pan = C[0], m = C[1];
            if (tid < m) {
                gR[c] = prow[C[2] + tid];
                dR[c] = pdiag[C[2] + tid];
#pragma unroll
                for (int j = 0; j < W; j++)
                    if (j < w) { cR[c][j] = pcol[(size_t)pan + (size_t)j * m + tid]; vR[c][j] = pval[(size_t)pan + (size_t)j * m + tid]; }
            }
        }
    }
    // (B)
    image_gather(tid);
#pragma unroll
    for (int c = 0; c < NC; c++) gather_kb<KB, double>(b + (size_t)(gR[c] >= 0 ? gR[c] : 0) * ld, gR[c] >= 0, bR[c]);
    for (int i0 = tid + U * nt; i0 < n_ext; i0 += U * nt) { image_indices(i0); image_gather(i0); }      // extended tiles of more than U x threads rows: two more round trips per chunk
    // The slots' positions in the image, scaled to KB columns HERE, behind every request of the prologue: left to itself the compiler scales each column index
    // where it is loaded -- under the per-column `j < w` branch, i.e. one wait per panel column (12 x 4 dependent round trips: k = 3 went from 52 to 69 us per
    // level visit when the loop that used to sort the diagonals out, and with it this anchor, was removed).
    if constexpr (KB > 1) {
#pragma unroll
        for (int c = 0; c < NC; c++)
#pragma unroll
            for (int j = 0; j < W; j++) { int v = cR[c][j]; asm volatile("" : "+v"(v)); cR[c][j] = v * KB; }
    }
    // Branch-free phases: the diagonal has left the row at plan time (its slot, like every padding slot, holds +0.0 at the row's own local index: a sum that
    // starts at +0 is not changed by adding +-0 -- it never holds -0 --, so the bits are those of the sum that skips these slots), and the phase loop is
    // W straight-line LDS reads and multiply-adds per column.
    __syncthreads();
    for (int s = 0; s < sweeps; s++) {
#pragma unroll
        for (int c = 0; c < NC; c++) {
            if (c < nc && s * nc + c < dbg_phases) {       // uniform
                const int p = s * nc + c + 1;
                const int* C = Hs + 4 + c * TILED_CSTRIDE;
                const int lrow = C[3] + tid, cnt = C[4 + (P - p)];
                if (tid < cnt) {
                    double acc[KB];
#pragma unroll
                    for (int q = 0; q < KB; q++) acc[q] = 0.0;
#pragma unroll
                    for (int j = 0; j < W; j++) {
                        const double v = vR[c][j];
                        const int at = cR[c][j];      // (already scaled by KB)
#pragma unroll
                        for (int q = 0; q < KB; q++) acc[q] += v * xs[at + q];
                    }
#pragma unroll
                    for (int q = 0; q < KB; q++) xs[lrow * KB + q] = (bR[c][q] - acc[q]) / dR[c];      // rows of one colour never read each other
                }
                __syncthreads();
            }
        }
    }
    if (!stop) {
#pragma unroll
        for (int c = 0; c < NC; c++)
            if (c < nc) {
                const int* C = Hs + 4 + c * TILED_CSTRIDE;
                if (tid < C[4]) {      // the owned rows lead the colour's panel
#pragma unroll
                    for (int q = 0; q < KB; q++) y[(size_t)gR[c] * ld + q] = xs[(C[3] + tid) * KB + q];
                }
            }
    }
}

template <int NC, int W>
static void launch_tiled_one(const TiledDev& Tl, const double* x, const double* b, double* y, int ld, int kb, const int* done, hipStream_t st)
{
    const size_t lds = (size_t)Tl.max_ext * kb * sizeof(double);
    static const int dbg = getenv("SMG_DEBUG_TILED_PHASES") ? atoi(getenv("SMG_DEBUG_TILED_PHASES")) : 1 << 20;   // timing probe (wrong results)
#define SMG_TILED_LAUNCH(KB) hipLaunchKernelGGL((k_tiled_gs<NC, W, KB>), dim3(Tl.n_tiles), dim3(Tl.threads), lds, st, Tl.hdr, Tl.ext_rows, Tl.pcol, Tl.pval, Tl.prow, Tl.pdiag, b, x, y, ld, Tl.nc, Tl.sweeps, done, dbg)
    if (kb == 1) SMG_TILED_LAUNCH(1);
    else if (kb == 2) SMG_TILED_LAUNCH(2);
    else SMG_TILED_LAUNCH(3);
#undef SMG_TILED_LAUNCH
}

// columns go through in groups of up to 3 (the mean-curvature-flow callers' k = 3 is one launch)
hipError_t launch_tiled_gs(const TiledDev& Tl, const double* x, const double* b, double* y, int k, const Ctrl* ctrl, hipStream_t st)
{
    // dynamic LDS (the extended tile's iterate) + the kernel's static header words must fit the 64 KB a workgroup gets without further ado
    if (Tl.n_tiles <= 0 || k < 1 || (size_t)Tl.max_ext * (k < 3 ? k : 3) * sizeof(double) + TILED_LDS_STATIC > 64 * 1024) return hipErrorInvalidValue;
    const int* done = ctrl ? &ctrl->done : never_done();
    for (int c = 0; c < k; c += 3) {
        const int kb = k - c < 3 ? k - c : 3;
        const bool w8 = Tl.w_max <= 8;
        if (Tl.nc <= 3) { if (w8) launch_tiled_one<3, 8>(Tl, x + c, b + c, y + c, k, kb, done, st); else launch_tiled_one<3, 12>(Tl, x + c, b + c, y + c, k, kb, done, st); }
        else if (Tl.nc == 4) { if (w8) launch_tiled_one<4, 8>(Tl, x + c, b + c, y + c, k, kb, done, st); else launch_tiled_one<4, 12>(Tl, x + c, b + c, y + c, k, kb, done, st); }
        else { if (w8) launch_tiled_one<5, 8>(Tl, x + c, b + c, y + c, k, kb, done, st); else launch_tiled_one<5, 12>(Tl, x + c, b + c, y + c, k, kb, done, st); }
    }
    return hipGetLastError();
}

hipError_t tiled_gs_prepare(int max_ext)
{
    // the plans are built with at most 8192 local rows: 64 KB of LDS, what a workgroup may ask for without further ado
    return (size_t)max_ext * sizeof(double) + TILED_LDS_STATIC <= 64 * 1024 ? hipSuccess : hipErrorInvalidValue;
}

}  // namespace smg
