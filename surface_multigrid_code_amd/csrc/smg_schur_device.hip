// smg_schur_device.hip -- device side of the Schur-complement coarse solver (plan and algebra: smg_schur.hpp).
//
// factor (every re-precompute):  scatter the values into the arena -> per block: D_i^-1 in LDS, W_i = D_i^-1 P_i, P_i^T W_i -> S -= the
//                                products (fixed lists) -> S^-1 by the blocked Gauss-Jordan of the dense path (launch_spd_inverse)
// solve  (every V-cycle):        g = b_S - sum W_i^T b_i  ->  x_S = S^-1 g (the dense path's products)  ->  u += [D_i^-1 b_i - W_i x_S ; x_S]
// Two shapes of the solve kernels: k < 64 columns (lanes = the 64 rows of a block, eight columns in registers per pass) and k >= 64 (lanes = columns).
// Bound: latency at these sizes (3 952 unknowns: 64 blocks, 1 688 separator rows) -- the factorisation is ~60 launches of a few us each, chained through the 64 x 64 pivot inversions.
#include <hip/hip_runtime.h>

#include "smg_device.hpp"
#include "smg_device_inl.hpp"
#include "smg_gj_inl.hpp"

namespace smg {

namespace {

template <typename T>
__device__ __forceinline__ T wave_sum(T v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);      // a fixed butterfly: the same sum on every run
    return v;
}

__global__ void k_schur_scatter(double* __restrict__ arena, const long long* __restrict__ pos, const long long* __restrict__ pos2, const double* __restrict__ val, int nnz,
                                const long long* __restrict__ ones, int n_ones)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nnz) {
        const double v = val[t];
        const long long p = pos[t], q = pos2[t];
        if (p >= 0) arena[p] = v;
        if (q >= 0) arena[q] = v;
    }
    if (t < n_ones) arena[ones[t]] = 1.0;
}

// one workgroup per interior block: D_i -> D_i^-1 (exactly symmetric: the lower triangle mirrored), W_i^T = (D_i^-1 P_i)^T, C_i = P_i^T W_i (lower triangle).
// The two products run on the fp64 matrix cores (v_mfma_f64_16x16x4: A operand lane -> A[lane & 15][lane >> 4], B operand lane -> B[lane >> 4][lane & 15],
// register r of the result -> row (lane >> 4) + 4 r, column lane & 15), 16 x 16 tiles, the 64-deep sum in 16 steps:
//   W (64 x m)  = D^-1 (LDS) x P_i      (P_i[r][c] = P^T[c][r]: a lane's 16 B operands of a tile column are loaded once and meet the four row tiles)
//   C (m x m)   = P_i^T x W             (A operand from P^T in memory, once per tile row; B operand from the image of W^T in LDS, pitch 65)
// With lanes as rows / columns and the other operand handed round by v_readlane the two phases took 13 + 18 of the kernel's 75 us (instruction issue:
// two v_readlane per multiply-add); phases of block 0 of C3's plan before that, by wall_clock64: load 4.3, inversion 15.6, mirror + store 1.9, image 5.8.
// LDS: the block and the inversion's two panels first; once W_i is out, the same 50 KB hold W_i^T (96 columns at a time).
constexpr int SCHUR_WL = 96;                                     // columns of W_i^T the LDS image holds (96 x 65 <= 64 x 65 + 16 x 65 + 64 x 17)
__global__ __launch_bounds__(256) void k_schur_blocks(double* arena, long long off_D, long long off_P, long long off_W, long long off_C, const int* __restrict__ sptr,
                                                      const long long* __restrict__ coff)
{
    __shared__ double lds[GJ_NB * (GJ_NB + 1) + 16 * (GJ_NB + 1) + GJ_NB * 17];
    double (*a)[GJ_NB + 1] = reinterpret_cast<double (*)[GJ_NB + 1]>(lds);
    double (*Rb)[GJ_NB + 1] = reinterpret_cast<double (*)[GJ_NB + 1]>(lds + GJ_NB * (GJ_NB + 1));
    double (*Cb)[17] = reinterpret_cast<double (*)[17]>(lds + GJ_NB * (GJ_NB + 1) + 16 * (GJ_NB + 1));
    static_assert(SCHUR_WL * 65 <= GJ_NB * (GJ_NB + 1) + 16 * (GJ_NB + 1) + GJ_NB * 17 && SCHUR_WL % 16 == 0, "the image of W does not fit the inversion's LDS");
    const int i = blockIdx.x, t = threadIdx.x;
    double* D = arena + off_D + (size_t)i * (GJ_NB * GJ_NB);
    {
        double v[16];
#pragma unroll
        for (int e = 0; e < 16; e++) v[e] = D[t + 256 * e];
#pragma unroll
        for (int e = 0; e < 16; e++) a[(t + 256 * e) >> 6][t & 63] = v[e];
    }
    const int s0 = sptr[i], m = sptr[i + 1] - s0;
    const double* P = arena + off_P + (size_t)64 * s0;
    double* W = arena + off_W + (size_t)64 * s0;
    const int lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int lr = lane >> 4, lc = lane & 15;
    const int T = (m + 15) / 16;                                  // 16-column tiles of the panel
    // the B operands of this wave's first tile column of W: requested now, they arrive behind the inversion
    double bv[16];
    {
        const int c = 16 * w + lc;
#pragma unroll
        for (int q = 0; q < 16; q++) bv[q] = (w < T && c < m) ? P[64 * c + 4 * q + lr] : 0.0;
    }
    __syncthreads();
    gj_invert64(a, Rb, Cb);
    for (int e = t; e < GJ_NB * GJ_NB; e += 256) { const int r = e >> 6, c = e & 63; if (c > r) a[r][c] = a[c][r]; }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 16; e++) D[t + 256 * e] = a[(t + 256 * e) >> 6][t & 63];
    for (int tj = w; tj < T; tj += 4) {
        const int c = 16 * tj + lc;
        double bn[16];                                            // the next tile column's operands travel while this one is multiplied
        {
            const int cn = c + 64;
#pragma unroll
            for (int q = 0; q < 16; q++) bn[q] = (tj + 4 < T && cn < m) ? P[64 * cn + 4 * q + lr] : 0.0;
        }
#pragma unroll
        for (int ti = 0; ti < 4; ti++) {
            v4f64 acc = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 16; q++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[16 * ti + lc][4 * q + lr], bv[q], acc, 0, 0, 0);
            if (c < m) {
#pragma unroll
                for (int r = 0; r < 4; r++) W[64 * c + 16 * ti + lr + 4 * r] = acc[r];
            }
        }
#pragma unroll
        for (int q = 0; q < 16; q++) bv[q] = bn[q];
    }
    __syncthreads();     // (orders the workgroup's own stores to W before its loads below; an agent-scope fence here wrote back the L2 the preceding memsets had dirtied: 40 us)
    double* C = arena + off_C + coff[i];
    for (int b0 = 0; b0 < m; b0 += SCHUR_WL) {           // (one pass unless the block touches more than 96 separator rows)
        const int nb = min(SCHUR_WL, m - b0);
        if (b0) __syncthreads();
        for (int e0 = t; e0 < nb * 64; e0 += 256 * 8) {
            double v[8];
#pragma unroll
            for (int q = 0; q < 8; q++) { const int e = e0 + 256 * q; v[q] = e < nb * 64 ? W[64 * b0 + e] : 0.0; }
#pragma unroll
            for (int q = 0; q < 8; q++) { const int e = e0 + 256 * q; if (e < nb * 64) lds[(e >> 6) * 65 + (e & 63)] = v[q]; }
        }
        for (int e = nb * 64 + t; e < ((nb + 15) / 16) * 16 * 64; e += 256) lds[(e >> 6) * 65 + (e & 63)] = 0.0;      // the image's last tile, beyond the panel
        __syncthreads();
        const int t2a = b0 / 16, t2b = (b0 + nb + 15) / 16;       // tile columns of C this image serves
        for (int t1 = t2a + w; t1 < T; t1 += 4) {                  // C[c1][c2], c2 <= c1: tile rows over the waves
            const int c1l = 16 * t1 + lc;
            double av[16];
#pragma unroll
            for (int q = 0; q < 16; q++) av[q] = c1l < m ? P[64 * c1l + 4 * q + lr] : 0.0;
            for (int t2 = t2a; t2 < t2b && t2 <= t1; t2++) {
                const double* wl = lds + (16 * (t2 - t2a) + lc) * 65 + lr;
                v4f64 acc = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int q = 0; q < 16; q++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], wl[4 * q], acc, 0, 0, 0);
                const int c2 = 16 * t2 + lc;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int c1 = 16 * t1 + lr + 4 * r;
                    if (c1 < m && c2 <= c1) C[(size_t)c1 * m + c2] = acc[r];
                }
            }
        }
    }
}

__global__ void k_schur_reduce(double* arena, long long off_C, const long long* __restrict__ rdst, const long long* __restrict__ rdst2, const int* __restrict__ rptr,
                               const long long* __restrict__ rsrc, int n_red)
{
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n_red) return;
    double s = arena[rdst[d]];
    const double* C = arena + off_C;
    for (int q = rptr[d]; q < rptr[d + 1]; q++) s -= C[rsrc[q]];
    arena[rdst[d]] = s;
    if (rdst2[d] >= 0) arena[rdst2[d]] = s;
}

// ---- solve, k < 64 columns: lanes = the 64 rows of a block, up to 8 columns in registers (blockIdx.y: groups of 8 columns) ------------------
constexpr int SCHUR_KC = 8;

// row i of the k = 1 product with S^-1 out of the partial products launch_sym_gemv_tiles left: the shares of the nt tiles of its block row, ascending,
// eight loads in flight (a plain loop waits for every share before it asks for the next)
template <typename T>
__device__ __forceinline__ T part_sum(const T* __restrict__ part, int nt, int i)
{
    const T* p = part + (size_t)(i >> 6) * nt * 64 + (i & 63);
    T s = (T)0;
    for (int b0 = 0; b0 < nt; b0 += 8) {
        T v[8];
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = b0 + e < nt ? p[(size_t)(b0 + e) * 64] : (T)0;
#pragma unroll
        for (int e = 0; e < 8; e++) s += v[e];
    }
    return s;
}

// g_j = b[srow_j] - sum over the blocks that touch j of W_i^T[c] . b_i: one wave per separator row
template <typename T>
__global__ __launch_bounds__(256) void k_schur_g(const T* __restrict__ W, const int* __restrict__ irow, const int* __restrict__ srow, const int* __restrict__ aptr,
                                                 const int* __restrict__ ablk, const int* __restrict__ apan, int ns, int ns_pad, const T* __restrict__ b, int k,
                                                 T* __restrict__ g, T* __restrict__ xs, const int* done)
{
    if (load_flag(done)) return;
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int c0 = blockIdx.y * SCHUR_KC, kc = min(SCHUR_KC, k - c0);
    if (j >= ns_pad) return;
    if (j >= ns) { if (lane < kc) { g[(size_t)j * k + c0 + lane] = (T)0; xs[(size_t)j * k + c0 + lane] = (T)0; } return; }
    T acc[SCHUR_KC];
#pragma unroll
    for (int c = 0; c < SCHUR_KC; c++) acc[c] = (T)0;
    const int q1 = aptr[j + 1];
    for (int q = aptr[j]; q < q1; q++) {
        const T w = W[(size_t)64 * apan[q] + lane];
        const int row = irow[ablk[q] * 64 + lane];
        if (row >= 0) {
            const T* br = b + (size_t)row * k + c0;
#pragma unroll
            for (int c = 0; c < SCHUR_KC; c++) if (c < kc) acc[c] += w * br[c];
        }
    }
    T out = (T)0;
#pragma unroll
    for (int c = 0; c < SCHUR_KC; c++) if (c < kc) { const T s = wave_sum(acc[c]); if (lane == c) out = s; }
    if (lane < kc) { g[(size_t)j * k + c0 + lane] = b[(size_t)srow[j] * k + c0 + lane] - out; xs[(size_t)j * k + c0 + lane] = (T)0; }
}

// u[I_i] += D_i^-1 b_i - W_i x_S[S_i] (one workgroup per block: the sums over the 64 + m_i terms in four quarters, combined in a fixed order);
// the workgroups behind the blocks: u[srow_j] += x_S[j].  PARTS (k = 1): x_S is still the partial products of launch_sym_gemv_tiles -- summed here,
// one launch less on the cycle's serial path.
template <typename T, bool PARTS>
__global__ __launch_bounds__(256) void k_schur_x(const T* __restrict__ D, const T* __restrict__ W, const int* __restrict__ sptr, const int* __restrict__ sidx,
                                                 const int* __restrict__ irow, const int* __restrict__ srow, int nb, int ns, const T* __restrict__ b, const T* __restrict__ xs, int nt,
                                                 int k, T* u, const int* done)
{
    if (load_flag(done)) return;
    const int t = threadIdx.x;
    const int c0 = blockIdx.y * SCHUR_KC, kc = min(SCHUR_KC, k - c0);
    if ((int)blockIdx.x >= nb) {
        const int e = ((int)blockIdx.x - nb) * 256 + t;
        if (e < ns * kc) {
            const int j = e / kc, c = c0 + (e - j * kc);
            u[(size_t)srow[j] * k + c] += PARTS ? part_sum(xs, nt, j) : xs[(size_t)j * k + c];
        }
        return;
    }
    const int i = blockIdx.x, lane = t & 63, part = t >> 6;
    const int s0 = sptr[i], m = sptr[i + 1] - s0;
    __shared__ T bL[64][SCHUR_KC + 1], xL[SCHUR_M_MAX_DEV][SCHUR_KC + 1], red[4][64][SCHUR_KC + 1];
    // the block's coefficients first: their loads travel while b_i and x_S[S_i] are staged
    T d[16], wv[SCHUR_M_MAX_DEV / 4];
    const T* Di = D + (size_t)i * 4096 + 64 * (16 * part) + lane;    // D^-1 is stored exactly symmetric: column `lane` of rows 16 part ..
#pragma unroll
    for (int e = 0; e < 16; e++) d[e] = Di[64 * e];
    const int q4 = (m + 3) / 4, sa = part * q4, sb = min(m, sa + q4);
    const T* Wi = W + (size_t)64 * (s0 + sa) + lane;
#pragma unroll
    for (int e = 0; e < SCHUR_M_MAX_DEV / 4; e++) wv[e] = sa + e < sb ? Wi[64 * e] : (T)0;
    // staging, every thread one row with all its columns at once: threads 0 .. 63 the block's rows (they also fetch the u they will add to),
    // threads 64 .. 191 the separator rows the block touches
    T uo[SCHUR_KC];
    int row = -1;
    if (t < 64) {
        row = irow[i * 64 + t];
        T v[SCHUR_KC];
#pragma unroll
        for (int c = 0; c < SCHUR_KC; c++) { v[c] = (row >= 0 && c < kc) ? b[(size_t)row * k + c0 + c] : (T)0; uo[c] = (row >= 0 && c < kc) ? u[(size_t)row * k + c0 + c] : (T)0; }
#pragma unroll
        for (int c = 0; c < SCHUR_KC; c++) bL[t][c] = v[c];
    } else if (t - 64 < m) {
        const int s = t - 64, sid = sidx[s0 + s];
        if (PARTS) xL[s][0] = part_sum(xs, nt, sid);
        else {
            T v[SCHUR_KC];
#pragma unroll
            for (int c = 0; c < SCHUR_KC; c++) v[c] = c < kc ? xs[(size_t)sid * k + c0 + c] : (T)0;
#pragma unroll
            for (int c = 0; c < SCHUR_KC; c++) xL[s][c] = v[c];
        }
    }
    __syncthreads();
    T acc[SCHUR_KC];
#pragma unroll
    for (int c = 0; c < SCHUR_KC; c++) acc[c] = (T)0;
#pragma unroll
    for (int e = 0; e < 16; e++) {
#pragma unroll
        for (int c = 0; c < SCHUR_KC; c++) if (c < kc) acc[c] += d[e] * bL[16 * part + e][c];
    }
#pragma unroll
    for (int e = 0; e < SCHUR_M_MAX_DEV / 4; e++) {
        if (sa + e < sb) {
#pragma unroll
            for (int c = 0; c < SCHUR_KC; c++) if (c < kc) acc[c] -= wv[e] * xL[sa + e][c];
        }
    }
    if (part != 0) {
#pragma unroll
        for (int c = 0; c < SCHUR_KC; c++) if (c < kc) red[part][lane][c] = acc[c];
    }
    __syncthreads();
    if (part == 0 && row >= 0) {
#pragma unroll
        for (int c = 0; c < SCHUR_KC; c++)
            if (c < kc) u[(size_t)row * k + c0 + c] = uo[c] + ((acc[c] + red[1][lane][c]) + (red[2][lane][c] + red[3][lane][c]));
    }
}

// ---- solve, k >= 64 columns: lanes = columns (blockIdx.y: groups of 64 columns) ---------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_schur_g_wide(const T* __restrict__ W, const int* __restrict__ irow, const int* __restrict__ bsize, const int* __restrict__ srow,
                                                      const int* __restrict__ aptr, const int* __restrict__ ablk, const int* __restrict__ apan, int ns, int ns_pad,
                                                      const T* __restrict__ b, int k, T* __restrict__ g, T* __restrict__ xs, const int* done)
{
    if (load_flag(done)) return;
    const int lane = threadIdx.x & 63;
    const int j = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    const int col = blockIdx.y * 64 + lane;
    if (j >= ns_pad || col >= k) return;
    if (j >= ns) { g[(size_t)j * k + col] = (T)0; xs[(size_t)j * k + col] = (T)0; return; }
    T acc = (T)0;
    const int q1 = aptr[j + 1];
    for (int q = aptr[j]; q < q1; q++) {
        const int i = ablk[q];
        const T* wp = W + (size_t)64 * apan[q];
        const int* ir = irow + i * 64;
        const int nr = bsize[i];
        int r = 0;
        for (; r + 8 <= nr; r += 8) {                    // eight rows of b in flight
            T x[8];
#pragma unroll
            for (int e = 0; e < 8; e++) x[e] = b[(size_t)ir[r + e] * k + col];
#pragma unroll
            for (int e = 0; e < 8; e++) acc += wp[r + e] * x[e];
        }
        for (; r < nr; r++) acc += wp[r] * b[(size_t)ir[r] * k + col];
    }
    g[(size_t)j * k + col] = b[(size_t)srow[j] * k + col] - acc;
    xs[(size_t)j * k + col] = (T)0;
}

// a wave owns SCHUR_XR rows of a block (blockIdx.x -> block, group of 4 waves): every row of b_i and of x_S[S_i] it reads meets SCHUR_XR accumulators,
// eight rows in flight.  (16 rows per wave left one wave per CU on the chip: 32 us at 64 columns, a chain of ~25 dependent batches.)
constexpr int SCHUR_XR = 4;
template <typename T>
__global__ __launch_bounds__(256) void k_schur_x_wide(const T* __restrict__ D, const T* __restrict__ W, const int* __restrict__ sptr, const int* __restrict__ sidx,
                                                      const int* __restrict__ irow, const int* __restrict__ bsize, const int* __restrict__ srow, int nb, int ns,
                                                      const T* __restrict__ b, const T* __restrict__ xs, int k, T* u, const int* done)
{
    if (load_flag(done)) return;
    constexpr int G = 64 / (4 * SCHUR_XR);               // workgroups per block
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = blockIdx.y * 64 + lane;
    if (col >= k) return;
    if ((int)blockIdx.x >= nb * G) {
        const int j = ((int)blockIdx.x - nb * G) * 4 + w;
        if (j < ns) u[(size_t)srow[j] * k + col] += xs[(size_t)j * k + col];
        return;
    }
    const int i = blockIdx.x / G;
    const int s0 = sptr[i], m = sptr[i + 1] - s0, nr = bsize[i];
    const int r0 = SCHUR_XR * (4 * ((int)blockIdx.x % G) + w);
    if (r0 >= nr) return;
    T acc[SCHUR_XR];
#pragma unroll
    for (int e = 0; e < SCHUR_XR; e++) acc[e] = (T)0;
    const T* Di = D + (size_t)i * 4096 + r0;            // D^-1 exactly symmetric: D[rp][r0 ..] = the wave's rows against column rp
    const int* ir = irow + i * 64;
    int rp = 0;
    for (; rp + 8 <= nr; rp += 8) {
        T x[8];
#pragma unroll
        for (int q = 0; q < 8; q++) x[q] = b[(size_t)ir[rp + q] * k + col];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const T* d = Di + 64 * (rp + q);
#pragma unroll
            for (int e = 0; e < SCHUR_XR; e++) acc[e] += d[e] * x[q];
        }
    }
    for (; rp < nr; rp++) {
        const T x = b[(size_t)ir[rp] * k + col];
        const T* d = Di + 64 * rp;
#pragma unroll
        for (int e = 0; e < SCHUR_XR; e++) acc[e] += d[e] * x;
    }
    const T* Wi = W + (size_t)64 * s0 + r0;
    const int* sx = sidx + s0;
    int s = 0;
    for (; s + 8 <= m; s += 8) {
        T x[8];
#pragma unroll
        for (int q = 0; q < 8; q++) x[q] = xs[(size_t)sx[s + q] * k + col];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const T* wv = Wi + 64 * (s + q);
#pragma unroll
            for (int e = 0; e < SCHUR_XR; e++) acc[e] -= wv[e] * x[q];
        }
    }
    for (; s < m; s++) {
        const T x = xs[(size_t)sx[s] * k + col];
        const T* wv = Wi + 64 * s;
#pragma unroll
        for (int e = 0; e < SCHUR_XR; e++) acc[e] -= wv[e] * x;
    }
#pragma unroll
    for (int e = 0; e < SCHUR_XR; e++)
        if (r0 + e < nr) { const size_t a = (size_t)ir[r0 + e] * k + col; u[a] += acc[e]; }
}

template <typename T> struct SchurPtrs;
template <> struct SchurPtrs<double> {
    static const double* arena(const SchurDev& F) { return F.arena; }
    static double* g(const SchurDev& F) { return F.g; }
    static double* xs(const SchurDev& F) { return F.xs; }
    static hipError_t dense(const SchurDev& F, int k, const Ctrl* ctrl, hipStream_t st)
    { return launch_dense_gemv_add(F.arena + F.off_S, F.ns, F.ns_pad, F.g, F.xs, k, k, ctrl, st, nullptr); }
    static hipError_t tiles(const SchurDev& F, hipStream_t st) { return launch_sym_gemv_tiles(F.arena + F.off_S, F.ns_pad, F.g, F.sym_work, st); }
};
template <> struct SchurPtrs<float> {
    static const float* arena(const SchurDev& F) { return F.arena32; }
    static float* g(const SchurDev& F) { return F.g32; }
    static float* xs(const SchurDev& F) { return F.xs32; }
    static hipError_t dense(const SchurDev& F, int k, const Ctrl* ctrl, hipStream_t st)
    { return launch_dense_gemv_add_f32(F.arena32 + F.off_S, F.ns, F.ns_pad, F.g32, F.xs32, k, k, ctrl, st, nullptr); }
    static hipError_t tiles(const SchurDev& F, hipStream_t st) { return launch_sym_gemv_tiles_f32(F.arena32 + F.off_S, F.ns_pad, F.g32, (float*)F.sym_work, st); }
};

template <typename T>
hipError_t schur_solve(const SchurDev& F, const T* b, T* u, int k, const Ctrl* ctrl, hipStream_t st)
{
    if (k <= 0 || F.nb <= 0) return hipSuccess;
    const int* done = ctrl ? &ctrl->done : never_done();
    const T* A = SchurPtrs<T>::arena(F);
    T* g = SchurPtrs<T>::g(F);
    T* xs = SchurPtrs<T>::xs(F);
    if (!A || !g || !xs) return hipErrorInvalidValue;
    const T *D = A + F.off_D, *W = A + F.off_W;
    if (k < 64) {
        const int gy = (k + SCHUR_KC - 1) / SCHUR_KC;
        hipLaunchKernelGGL((k_schur_g<T>), dim3((F.ns_pad + 3) / 4, gy), dim3(256), 0, st, W, F.irow, F.srow, F.aptr, F.ablk, F.apan, F.ns, F.ns_pad, b, k, g, xs, done);
        if (k == 1 && F.sym_work) {
            // x_S = S^-1 g through the lower triangle of S^-1 (half the bytes); its 64-row shares are summed by the last kernel
            hipError_t e = SchurPtrs<T>::tiles(F, st);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((k_schur_x<T, true>), dim3(F.nb + (F.ns + 255) / 256, 1), dim3(256), 0, st, D, W, F.sptr, F.sidx, F.irow, F.srow, F.nb, F.ns, b,
                               (const T*)F.sym_work, F.ns_pad / 64, k, u, done);
            return hipGetLastError();
        }
        hipError_t e = SchurPtrs<T>::dense(F, k, ctrl, st);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_schur_x<T, false>), dim3(F.nb + (F.ns * SCHUR_KC + 255) / 256, gy), dim3(256), 0, st, D, W, F.sptr, F.sidx, F.irow, F.srow, F.nb, F.ns, b,
                           (const T*)xs, 0, k, u, done);
    } else {
        const int gy = (k + 63) / 64;
        hipLaunchKernelGGL((k_schur_g_wide<T>), dim3((F.ns_pad + 3) / 4, gy), dim3(256), 0, st, W, F.irow, F.bsize, F.srow, F.aptr, F.ablk, F.apan, F.ns, F.ns_pad, b, k, g, xs,
                           done);
        hipError_t e = SchurPtrs<T>::dense(F, k, ctrl, st);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_schur_x_wide<T>), dim3(F.nb * (64 / (4 * SCHUR_XR)) + (F.ns + 3) / 4, gy), dim3(256), 0, st, D, W, F.sptr, F.sidx, F.irow, F.bsize, F.srow, F.nb, F.ns, b, (const T*)xs, k, u,
                           done);
    }
    return hipGetLastError();
}

}  // namespace

hipError_t launch_schur_factor(const SchurDev& F, const double* vals, hipStream_t st)
{
    if (F.nb <= 0) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(F.arena + F.off_D, 0, (size_t)(F.off_W - F.off_D) * sizeof(double), st);      // D and P: the entries land on zeros
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(F.arena + F.off_S, 0, (size_t)F.ns_pad * F.ns_pad * sizeof(double), st);
    if (e != hipSuccess) return e;
    const int nmax = F.nnz > F.n_ones ? F.nnz : F.n_ones;
    hipLaunchKernelGGL(k_schur_scatter, dim3((nmax + 255) / 256), dim3(256), 0, st, F.arena, F.pos, F.pos2, vals, F.nnz, F.ones, F.n_ones);
    hipLaunchKernelGGL(k_schur_blocks, dim3(F.nb), dim3(256), 0, st, F.arena, F.off_D, F.off_P, F.off_W, F.off_C, F.sptr, F.coff);
    if (F.n_red > 0) hipLaunchKernelGGL(k_schur_reduce, dim3((F.n_red + 255) / 256), dim3(256), 0, st, F.arena, F.off_C, F.rdst, F.rdst2, F.rptr, F.rsrc, F.n_red);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_spd_inverse(F.arena + F.off_S, F.ns_pad, F.gj_work, st);
}

// The inverse of a symmetric positive definite matrix has a positive diagonal: after launch_schur_factor every diagonal entry of the inverted interior blocks
// and of S^-1 must be finite and > 0.  The elimination runs without pivoting and without a failure signal of its own (like the dense path); a coarsest
// matrix that is not SPD shows here -- *flag is raised -- instead of as Inf / NaN corrections in some later solve.
__global__ __launch_bounds__(256) void k_schur_check(const double* __restrict__ arena, long long off_D, int nb, long long off_S, int ns_pad, int* flag)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long nd = (long long)nb * 64;
    double x;
    if (t < nd) x = arena[off_D + (t / 64) * 4096 + (t % 64) * 65];
    else if (t < nd + ns_pad) { const long long j = t - nd; x = arena[off_S + j * ns_pad + j]; }
    else return;
    if (!(x > 0.0) || !(x < 1.0e300)) *flag = 1;
}
hipError_t launch_schur_check(const SchurDev& F, int* d_flag, hipStream_t st)
{
    if (F.nb <= 0 || !d_flag) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(d_flag, 0, sizeof(int), st);
    if (e != hipSuccess) return e;
    const long long tot = (long long)F.nb * 64 + F.ns_pad;
    hipLaunchKernelGGL(k_schur_check, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, F.arena, F.off_D, F.nb, F.off_S, F.ns_pad, d_flag);
    return hipGetLastError();
}

hipError_t launch_schur_solve(const SchurDev& F, const double* b, double* u, int k, const Ctrl* ctrl, hipStream_t st) { return schur_solve<double>(F, b, u, k, ctrl, st); }
hipError_t launch_schur_solve_f32(const SchurDev& F, const float* b, float* u, int k, const Ctrl* ctrl, hipStream_t st) { return schur_solve<float>(F, b, u, k, ctrl, st); }

}  // namespace smg
