// smg_gj_inl.hpp -- the in-LDS inverse of one 64 x 64 SPD block, shared by the blocked Gauss-Jordan inversion (smg_device.hip) and the
// interior blocks of the Schur-complement coarse solver (smg_schur_device.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace smg {

typedef double v4f64 __attribute__((ext_vector_type(4)));
constexpr int GJ_NB = 64;

// In-place inverse of a 64 x 64 (SPD) block held in LDS, by a workgroup of 256 threads: Gauss-Jordan in four steps of 16 -- the
// 16 x 16 pivot is inverted by one wave on its own, in registers (16 eliminations, no workgroup barrier, no LDS), the row panel and the
// rank-16 update run on the matrix cores -- 12 workgroup barriers instead of the 128 of the element-wise elimination
// (35 us -> 17 us for the stand-alone kernel with the pivot in LDS; what remains is the chain of 64 dependent eliminations).
// Must be entered by all threads, with `a` complete (barrier before the call is the caller's).
__device__ __forceinline__ void gj_invert64(double (*a)[GJ_NB + 1], double (*Rb)[GJ_NB + 1], double (*Cb)[17])
{
    const int t = threadIdx.x;
    for (int kk = 0; kk < 4; kk++) {
        const int P = 16 * kk;
        if (t < 64) {
            // the 16 x 16 pivot in the registers of one wave: lane (jj, ig) holds rows ig + 4 q of column jj; what an elimination needs of other
            // lanes -- row p and column p -- comes through the cross-lane network (ds_bpermute / v_readlane), not through LDS stores, fences and
            // wave barriers: 16 eliminations 3.2 -> ~1.2 us, and they are the serial chain of every Gauss-Jordan step on a small matrix
            const int jj = t & 15, ig = t >> 4;           // rows ig + 4 q of the pivot, column jj
            double cur[4];
#pragma unroll
            for (int q = 0; q < 4; q++) cur[q] = a[P + ig + 4 * q][P + jj];
#pragma unroll
            for (int p = 0; p < 16; p++) {
                // reciprocal of the pivot: hardware estimate + two Newton steps (the IEEE division sequence is several times longer
                // and sits on the one serial chain of the whole inversion)
                const int src = p + 16 * (p & 3);                                  // the lane that holds a[p][p], in cur[p >> 2]
                const double piv = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(cur[p >> 2]), src), __builtin_amdgcn_readlane(__double2loint(cur[p >> 2]), src));
                const double r = __shfl(cur[p >> 2], jj + 16 * (p & 3));           // a[p][jj]
                double f[4];
#pragma unroll
                for (int q = 0; q < 4; q++) f[q] = __shfl(cur[q], p + 16 * ig);    // a[ig + 4 q][p]
                double d = __builtin_amdgcn_rcp(piv);
                d = __builtin_fma(d, __builtin_fma(-piv, d, 1.0), d);
                d = __builtin_fma(d, __builtin_fma(-piv, d, 1.0), d);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int i = ig + 4 * q;
                    double val;
                    if (i == p) val = (jj == p) ? d : r * d;
                    else val = (jj == p) ? -(f[q] * d) : cur[q] - f[q] * (r * d);
                    cur[q] = val;
                }
            }
#pragma unroll
            for (int q = 0; q < 4; q++) a[P + ig + 4 * q][P + jj] = cur[q];
        }
        __syncthreads();
        // row panel D^-1 a[P.., :] (one 16 x 16 tile per wave; the inverse itself in the pivot columns) and a copy of the column panel
        const int w = t >> 6, lane = t & 63, lr = lane >> 4, lc = lane & 15;
        {
            if (w == kk) {
#pragma unroll
                for (int r = 0; r < 4; r++) Rb[lr + 4 * r][16 * w + lc] = a[P + lr + 4 * r][16 * w + lc];
            } else {
                v4f64 acc = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int q = 0; q < 4; q++)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[P + lc][P + 4 * q + lr], a[P + 4 * q + lr][16 * w + lc], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; r++) Rb[lr + 4 * r][16 * w + lc] = acc[r];
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) { const int e = t + 256 * q; Cb[e >> 4][e & 15] = a[e >> 4][P + (e & 15)]; }
        __syncthreads();
        // rank-16 update, four 16 x 16 tiles (tix >> 2, tix & 3) per wave; the pivot rows take the row panel
#pragma unroll
        for (int tix = w; tix < 16; tix += 4) {
            const int ti = tix >> 2, tj = tix & 3;
            if (ti == kk) {
#pragma unroll
                for (int r = 0; r < 4; r++) a[P + lr + 4 * r][16 * tj + lc] = Rb[lr + 4 * r][16 * tj + lc];
            } else {
                v4f64 acc;
#pragma unroll
                for (int r = 0; r < 4; r++) acc[r] = tj == kk ? 0.0 : a[16 * ti + lr + 4 * r][16 * tj + lc];
#pragma unroll
                for (int q = 0; q < 4; q++)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-Cb[16 * ti + lc][4 * q + lr], Rb[4 * q + lr][16 * tj + lc], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; r++) a[16 * ti + lr + 4 * r][16 * tj + lc] = acc[r];
            }
        }
        __syncthreads();
    }
}

}  // namespace smg
