// smg_bsr3_device.hip -- gfx950 kernels of the block (3 degrees of freedom per vertex) variant.
// Layout and rationale: smg_bsr3.hpp.  One wavefront = one slice of 64 VERTICES, one lane = one vertex = three matrix rows
// (3v, 3v+1, 3v+2); a wave's load of one value plane of one panel column is 512 contiguous bytes, of the block columns 256.
// HBM-bound like the scalar kernels (76 bytes per 3 x 3 block against 18 flops): no MFMA.
//
// Arithmetic: per row the products are added sequentially in ascending column order (block by block, inside a block by column),
// multiply and add separate (-ffp-contract=off) -- the reference's Eigen kernels on the scalar 3n x 3n matrix, bit for bit; a
// Gauss-Seidel lane finishes row 3v before it starts row 3v+1, which reads the new value of 3v (the reference's lexicographic
// sweep, src/mg_VCycle.cpp:146-160, on the colour-major vertex numbering).
#include <hip/hip_runtime.h>

#include "smg_device.hpp"
#include "smg_device_inl.hpp"

namespace smg {

template <bool LD1, typename T>
__device__ __forceinline__ void gather3(const T* x, int c, int ld, bool use, T (&out)[3])
{
    if constexpr (LD1) {
        gather_kb<3, T>(x + (size_t)3 * (size_t)(c < 0 ? 0 : c), use, out);
    } else {
        const size_t o = (size_t)3 * (size_t)(c < 0 ? 0 : c) * (size_t)ld;
        out[0] = use ? x[o] : (T)0;
        out[1] = use ? x[o + (size_t)ld] : (T)0;
        out[2] = use ? x[o + 2 * (size_t)ld] : (T)0;
    }
}

// MODE: SELL_AX, SELL_RESID, SELL_RESID_SS, SELL_RESID_BOTH, SELL_GS, SELL_JACOBI, SELL_CHEBY (smg_device.hpp; same meaning per scalar row).
// x / b / y / dvec: row-major (3 n_v) x ld blocks, already offset to the column this launch handles.
// T = double: the reference's arithmetic; T = float: the fp32 image of the mixed-precision cycle (same order of operations in fp32).
template <int MODE, bool LD1, typename T>
__global__ __launch_bounds__(256) void k_bsr3(const int* a_col, const T* a_val, const int* a_slice_off, const int* a_slice_row, const int* a_slice_w,
                                              const int* a_order, int s_begin, int s_end, int n_blocks, int use_order, const T* x, const T* b,
                                              T* y, int ld, const int* done, double* partials, double omega_d, double c1_d, T* dvec)
{
    constexpr bool GS = MODE == SELL_GS, JAC = MODE == SELL_JACOBI, CHEB = MODE == SELL_CHEBY, SS = MODE == SELL_RESID_SS, BOTH = MODE == SELL_RESID_BOTH;
    const T omega = (T)omega_d, c1 = (T)c1_d;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bid = xcd_remap(blockIdx.x, n_blocks);
    const int ls = __builtin_amdgcn_readfirstlane(s_begin + bid * 4 + wave);
    double ss = 0.0;   // (always fp64)
    int stop = 0;
    if (ls < s_end) {
        const int s = (!GS && use_order) ? a_order[ls] : ls;
        stop = load_flag(done);
        const int off = a_slice_off[s], w = a_slice_w[s];
        const int row0 = a_slice_row[s], nrow = a_slice_row[s + 1] - row0;
        const int vtx = row0 + lane;
        const bool live = lane < nrow;
        const int* cp = a_col + (size_t)off * 64 + lane;
        const T* vp = a_val + (size_t)off * 9 * 64 + lane;
        const size_t o0 = (size_t)3 * (size_t)vtx * (size_t)ld;
        T bv[3] = {(T)0, (T)0, (T)0}, dold[3] = {(T)0, (T)0, (T)0};
        if (MODE != SELL_AX && live) {
#pragma unroll
            for (int d = 0; d < 3; d++) bv[d] = b[o0 + (size_t)d * ld];
        }
        if (CHEB && live && c1 != (T)0) {
#pragma unroll
            for (int d = 0; d < 3; d++) dold[d] = dvec[o0 + (size_t)d * ld];
        }
        T out[3] = {(T)0, (T)0, (T)0};
        if constexpr (!GS) {
            // the three rows of a vertex are independent: one pass over the block row
            constexpr int U = 4;
            T acc[3] = {(T)0, (T)0, (T)0}, diag[3] = {(T)1, (T)1, (T)1}, xi[3] = {(T)0, (T)0, (T)0};
            for (int j0 = 0; j0 < w; j0 += U) {
                int c[U];
                T v[U][9], xg[U][3];
#pragma unroll
                for (int t = 0; t < U; t++) {
                    if (j0 + t < w) {   // wave-uniform
                        c[t] = cp[(size_t)(j0 + t) * 64];
#pragma unroll
                        for (int e = 0; e < 9; e++) v[t][e] = vp[((size_t)(j0 + t) * 9 + e) * 64];
                    } else {
                        c[t] = -1;
#pragma unroll
                        for (int e = 0; e < 9; e++) v[t][e] = (T)0;
                    }
                }
#pragma unroll
                for (int t = 0; t < U; t++) gather3<LD1, T>(x, c[t], ld, c[t] >= 0, xg[t]);
#pragma unroll
                for (int t = 0; t < U; t++) {
                    if (c[t] >= 0) {
                        const bool own = c[t] == vtx;
#pragma unroll
                        for (int d = 0; d < 3; d++)
#pragma unroll
                            for (int e = 0; e < 3; e++) {
                                if ((JAC || CHEB) && own && e == d) { diag[d] = v[t][3 * d + e]; xi[d] = xg[t][e]; }
                                else acc[d] += v[t][3 * d + e] * xg[t][e];
                            }
                    }
                }
            }
#pragma unroll
            for (int d = 0; d < 3; d++) {
                if (MODE == SELL_AX) out[d] = acc[d];
                else if (MODE == SELL_RESID) out[d] = bv[d] - acc[d];
                else if (BOTH) { out[d] = bv[d] - acc[d]; if (live) { const double t = (double)out[d]; ss += t * t; } }
                else if (JAC) { const T t = (bv[d] - acc[d]) / diag[d]; out[d] = xi[d] + omega * (t - xi[d]); }
                else if (CHEB) {
                    const T t = (bv[d] - acc[d]) / diag[d];
                    const T r = t - xi[d];
                    const T dn = c1 != (T)0 ? c1 * dold[d] + omega * r : omega * r;
                    out[d] = xi[d] + dn;
                    dold[d] = dn;
                }
                else if (live) { const double t = (double)(bv[d] - acc[d]); ss += t * t; }
            }
        } else {
            // Gauss-Seidel: row 3v, then 3v+1 with the new value of 3v, then 3v+2 with both.  The block columns and the gathered values of
            // the first WR panel columns (all of them on mesh matrices: a vertex has 7 - 8 blocks) are fetched ONCE and stay in registers for
            // the three rows; each row streams its own three value planes, the next row's planes requested while this one is summed.
            // Panel columns beyond WR (wide Galerkin rows) repeat their gathers per row.
            constexpr int WR = 8;
            int c[WR];
            T xg[WR][3];
#pragma unroll
            for (int t = 0; t < WR; t++) c[t] = t < w ? cp[(size_t)t * 64] : -1;
            T v[WR][3], vn[WR][3];
#pragma unroll
            for (int t = 0; t < WR; t++)
#pragma unroll
                for (int e = 0; e < 3; e++) v[t][e] = t < w ? vp[((size_t)t * 9 + e) * 64] : (T)0;
#pragma unroll
            for (int t = 0; t < WR; t++) gather3<LD1, T>(x, c[t], ld, c[t] >= 0, xg[t]);
#pragma unroll
            for (int d = 0; d < 3; d++) {
                if (d < 2) {
#pragma unroll
                    for (int t = 0; t < WR; t++)
#pragma unroll
                        for (int e = 0; e < 3; e++) vn[t][e] = t < w ? vp[((size_t)t * 9 + 3 * (d + 1) + e) * 64] : (T)0;
                }
                T acc = (T)0, diag = (T)1;
#pragma unroll
                for (int t = 0; t < WR; t++) {
                    if (c[t] >= 0) {
                        const bool own = c[t] == vtx;
#pragma unroll
                        for (int e = 0; e < 3; e++) {
                            if (own && e == d) diag = v[t][e];
                            else acc += v[t][e] * ((own && e < d) ? out[e] : xg[t][e]);
                        }
                    }
                }
                for (int j0 = WR; j0 < w; j0 += 4) {      // the tail of a wide block row
                    int c2[4];
                    T v2[4][3], x2[4][3];
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        if (j0 + t < w) {
                            c2[t] = cp[(size_t)(j0 + t) * 64];
#pragma unroll
                            for (int e = 0; e < 3; e++) v2[t][e] = vp[((size_t)(j0 + t) * 9 + 3 * d + e) * 64];
                        } else {
                            c2[t] = -1;
#pragma unroll
                            for (int e = 0; e < 3; e++) v2[t][e] = (T)0;
                        }
                    }
#pragma unroll
                    for (int t = 0; t < 4; t++) gather3<LD1, T>(x, c2[t], ld, c2[t] >= 0, x2[t]);
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        if (c2[t] >= 0) {
                            const bool own = c2[t] == vtx;
#pragma unroll
                            for (int e = 0; e < 3; e++) {
                                if (own && e == d) diag = v2[t][e];
                                else acc += v2[t][e] * ((own && e < d) ? out[e] : x2[t][e]);
                            }
                        }
                    }
                }
                out[d] = (bv[d] - acc) / diag;
#pragma unroll
                for (int t = 0; t < WR; t++)
#pragma unroll
                    for (int e = 0; e < 3; e++) v[t][e] = vn[t][e];
            }
        }
        if (live && !stop && !SS) {
#pragma unroll
            for (int d = 0; d < 3; d++) {
                y[o0 + (size_t)d * ld] = out[d];
                if (CHEB) dvec[o0 + (size_t)d * ld] = dold[d];
            }
        }
        if (SS && stop) ss = 0.0;
    }
    if constexpr (SS || BOTH) {
        __shared__ double red[4];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_down(ss, o, 64);
        if (lane == 0) red[wave] = ss;
        __syncthreads();
        if (threadIdx.x == 0 && !stop) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

int bsr3_blocks(int n_slices) { return (n_slices + 3) / 4; }

// The block image of B = A(perm3, perm3) -- perm3 the DOF numbering a vertex numbering `perm` (new -> old) induces -- or of B^T (A
// structurally symmetric), written on the device from A's scalar CSR arrays in the caller's numbering (Bsr3Buf of a layout made on the
// host from the block-row lengths): one lane per vertex.  A block's panel column is the rank of its new vertex index among the block row's
// (gptr / gcol: the n_v x n_v pattern of the blocks, caller numbering) -- block columns ascend, the host's build_bsr3 order: the same image.
// The panels must hold col = -1, val = 0 on entry.
__global__ __launch_bounds__(256) void k_bsr3_fill(const int* __restrict__ ptr, const int* __restrict__ col, const double* __restrict__ val,
                                                   const int* __restrict__ gptr, const int* __restrict__ gcol, const int* __restrict__ perm,
                                                   const int* __restrict__ iperm, const int* __restrict__ slice_row, const int* __restrict__ slice_off,
                                                   int n_slices, int transposed, int* b_col, double* b_val)
{
    const int lane = threadIdx.x & 63, s = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (s >= n_slices) return;
    const int row0 = slice_row[s], nrow = slice_row[s + 1] - row0;
    if (lane >= nrow) return;
    const int v = perm[row0 + lane];
    const int g0 = gptr[v], g1 = gptr[v + 1];
    const size_t off = (size_t)slice_off[s];
    for (int q = g0; q < g1; q++) {        // the block columns of the row, each at its rank
        const int Jn = iperm[gcol[q]];
        int rank = 0;
        for (int t = g0; t < g1; t++) rank += iperm[gcol[t]] < Jn ? 1 : 0;
        b_col[(off + (size_t)rank) * 64 + lane] = Jn;
    }
    for (int d = 0; d < 3; d++)
        for (int p = ptr[3 * v + d]; p < ptr[3 * v + d + 1]; p++) {
            const int c = col[p], Jo = c / 3, e = c - 3 * Jo;
            const int Jn = iperm[Jo];
            int rank = 0;
            for (int t = g0; t < g1; t++) rank += iperm[gcol[t]] < Jn ? 1 : 0;
            int from = p;
            if (transposed) {              // B^T(3v+d, 3J+e) = A(3J+e, 3v+d), found by bisection (the caller has checked that it is stored)
                const int want = 3 * v + d;
                int lo = ptr[c], hi = ptr[c + 1];
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (col[mid] < want) lo = mid + 1; else hi = mid; }
                from = lo;
            }
            b_val[((off + (size_t)rank) * 9 + (size_t)(3 * d + e)) * 64 + lane] = val[from];
        }
}
hipError_t launch_bsr3_fill(const int* ptr, const int* col, const double* val, const int* gptr, const int* gcol, const int* perm, const int* iperm, const Bsr3Dev& B,
                            size_t panel_cols, bool transposed, hipStream_t st)
{
    hipError_t e = hipMemsetAsync(const_cast<int*>(B.col), 0xFF, panel_cols * 64 * sizeof(int), st);       // -1
    if (e == hipSuccess) e = hipMemsetAsync(const_cast<double*>(B.val), 0, panel_cols * 64 * 9 * sizeof(double), st);
    if (e != hipSuccess || B.n_slices <= 0) return e;
    hipLaunchKernelGGL(k_bsr3_fill, dim3((B.n_slices + 3) / 4), dim3(256), 0, st, ptr, col, val, gptr, gcol, perm, iperm, B.slice_row, B.slice_off, B.n_slices,
                       transposed ? 1 : 0, const_cast<int*>(B.col), const_cast<double*>(B.val));
    return hipGetLastError();
}

template <int MODE, typename T>
static hipError_t launch_bsr3_mode(const Bsr3Dev& A, const T* vals, int s_begin, int s_end, const T* x, const T* b, T* y, int k, const Ctrl* ctrl,
                                   double* partials, int* n_blocks, hipStream_t st, double omega, double c1, T* dvec)
{
    const int ns = s_end - s_begin;
    if (n_blocks) *n_blocks = 0;
    if (ns <= 0) return hipSuccess;
    const int nb = bsr3_blocks(ns);
    const int* done = ctrl ? &ctrl->done : never_done();
    const int use_order = (A.order && s_begin == 0 && s_end == A.n_slices) ? 1 : 0;
    for (int c = 0; c < k; c++) {
        const T* xx = x ? x + c : nullptr;
        const T* bb = b ? b + c : nullptr;
        T* yy = y ? y + c : nullptr;
        T* dd = dvec ? dvec + c : nullptr;
        double* pp = partials ? partials + (size_t)c * nb : nullptr;
        if (k == 1)
            hipLaunchKernelGGL((k_bsr3<MODE, true, T>), dim3(nb), dim3(256), 0, st, A.col, vals, A.slice_off, A.slice_row, A.slice_w, A.order, s_begin, s_end, nb,
                               use_order, xx, bb, yy, k, done, pp, omega, c1, dd);
        else
            hipLaunchKernelGGL((k_bsr3<MODE, false, T>), dim3(nb), dim3(256), 0, st, A.col, vals, A.slice_off, A.slice_row, A.slice_w, A.order, s_begin, s_end, nb,
                               use_order, xx, bb, yy, k, done, pp, omega, c1, dd);
    }
    if (n_blocks) *n_blocks = nb * k;
    return hipGetLastError();
}

hipError_t launch_bsr3(SellMode mode, const Bsr3Dev& A, int s_begin, int s_end, const double* x, const double* b, double* y, int k, const Ctrl* ctrl,
                       double* partials, int* n_blocks, hipStream_t st, double omega, double c1, double* dvec)
{
    switch (mode) {
        case SELL_AX: return launch_bsr3_mode<SELL_AX, double>(A, A.val, s_begin, s_end, x, b, y, k, ctrl, partials, n_blocks, st, omega, c1, dvec);
        case SELL_RESID: return launch_bsr3_mode<SELL_RESID, double>(A, A.val, s_begin, s_end, x, b, y, k, ctrl, partials, n_blocks, st, omega, c1, dvec);
        case SELL_RESID_SS: return launch_bsr3_mode<SELL_RESID_SS, double>(A, A.val, s_begin, s_end, x, b, y, k, ctrl, partials, n_blocks, st, omega, c1, dvec);
        case SELL_RESID_BOTH: return launch_bsr3_mode<SELL_RESID_BOTH, double>(A, A.val, s_begin, s_end, x, b, y, k, ctrl, partials, n_blocks, st, omega, c1, dvec);
        case SELL_GS: return launch_bsr3_mode<SELL_GS, double>(A, A.val, s_begin, s_end, x, b, y, k, ctrl, partials, n_blocks, st, omega, c1, dvec);
        case SELL_JACOBI: return launch_bsr3_mode<SELL_JACOBI, double>(A, A.val, s_begin, s_end, x, b, y, k, ctrl, partials, n_blocks, st, omega, c1, dvec);
        case SELL_CHEBY: return launch_bsr3_mode<SELL_CHEBY, double>(A, A.val, s_begin, s_end, x, b, y, k, ctrl, partials, n_blocks, st, omega, c1, dvec);
        default: return hipErrorInvalidValue;
    }
}
// the fp32 image (Bsr3Dev::valf): what the V-cycle of the mixed-precision mode needs
hipError_t launch_bsr3_f32(SellMode mode, const Bsr3Dev& A, int s_begin, int s_end, const float* x, const float* b, float* y, int k, const Ctrl* ctrl, hipStream_t st,
                           double omega, double c1, float* dvec)
{
    if (!A.valf) return hipErrorInvalidValue;
    switch (mode) {
        case SELL_AX: return launch_bsr3_mode<SELL_AX, float>(A, A.valf, s_begin, s_end, x, b, y, k, ctrl, nullptr, nullptr, st, omega, c1, dvec);
        case SELL_RESID: return launch_bsr3_mode<SELL_RESID, float>(A, A.valf, s_begin, s_end, x, b, y, k, ctrl, nullptr, nullptr, st, omega, c1, dvec);
        case SELL_GS: return launch_bsr3_mode<SELL_GS, float>(A, A.valf, s_begin, s_end, x, b, y, k, ctrl, nullptr, nullptr, st, omega, c1, dvec);
        case SELL_JACOBI: return launch_bsr3_mode<SELL_JACOBI, float>(A, A.valf, s_begin, s_end, x, b, y, k, ctrl, nullptr, nullptr, st, omega, c1, dvec);
        case SELL_CHEBY: return launch_bsr3_mode<SELL_CHEBY, float>(A, A.valf, s_begin, s_end, x, b, y, k, ctrl, nullptr, nullptr, st, omega, c1, dvec);
        default: return hipErrorInvalidValue;
    }
}

// Gershgorin bound of D^-1 A over the scalar rows of the block matrix (see k_gershgorin): explicit zeros add |0| = 0.
__global__ __launch_bounds__(256) void k_bsr3_gershgorin(const int* a_col, const double* a_val, const int* a_slice_off, const int* a_slice_row, const int* a_slice_w,
                                                         int n_slices, unsigned long long* out)
{
    const int lane = threadIdx.x & 63, s = blockIdx.x * 4 + (threadIdx.x >> 6);
    double ratio = 0.0;
    if (s < n_slices) {
        const int row0 = a_slice_row[s], nrow = a_slice_row[s + 1] - row0, w = a_slice_w[s], off = a_slice_off[s];
        const int* cp = a_col + (size_t)off * 64 + lane;
        const double* vp = a_val + (size_t)off * 9 * 64 + lane;
        double sum[3] = {0.0, 0.0, 0.0}, diag[3] = {0.0, 0.0, 0.0};
        for (int j = 0; j < w; j++) {
            const int c = cp[(size_t)j * 64];
            if (c < 0) continue;
#pragma unroll
            for (int d = 0; d < 3; d++)
#pragma unroll
                for (int e = 0; e < 3; e++) {
                    const double v = vp[((size_t)j * 9 + 3 * d + e) * 64];
                    sum[d] += fabs(v);
                    if (c == row0 + lane && e == d) diag[d] = v;
                }
        }
        if (lane < nrow)
#pragma unroll
            for (int d = 0; d < 3; d++) if (diag[d] > 0.0) ratio = fmax(ratio, sum[d] / diag[d]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ratio = fmax(ratio, __shfl_down(ratio, o, 64));
    if (lane == 0 && ratio > 0.0) atomicMax(out, (unsigned long long)__double_as_longlong(ratio));
}

hipError_t launch_bsr3_gershgorin(const Bsr3Dev& A, double* out, hipStream_t st)
{
    hipError_t e = hipMemsetAsync(out, 0, sizeof(double), st);
    if (e != hipSuccess || A.n_slices <= 0) return e;
    hipLaunchKernelGGL(k_bsr3_gershgorin, dim3((A.n_slices + 3) / 4), dim3(256), 0, st, A.col, A.val, A.slice_off, A.slice_row, A.slice_w, A.n_slices,
                       (unsigned long long*)out);
    return hipGetLastError();
}

}  // namespace smg
