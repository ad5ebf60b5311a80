// smg_schur.hpp -- coarse solver for the upper half of the dense range: one level of exact block elimination (Schur complement).
//
// solver.compute(Ac) / solver.solve (reference src/min_quad_with_fixed_mg.cpp:47-48, :253-254; src/mg_VCycle.cpp:181-201) on the coarsest
// matrix.  The dense inverse (smg_device.hip, blocked Gauss-Jordan) spends n^3 flops on a matrix with ~18 entries per row: 2.4 ms at the
// 3 952 unknowns of C3, every time the values change (the time-stepping callers 05 / 06 re-factor at every step).  Here the rows are cut
// into compact blocks of <= 64 (recursive breadth-first bisection, smg_tiled.cpp) and a vertex cover of the edges between different blocks
// is taken out as the SEPARATOR S; what is left of the blocks -- the INTERIORS I_1 .. I_p -- is coupled through S only:
//
//      A = [ D   P ]      D = diag(A_11 .. A_pp)  (<= 64 x 64 each),   P_i = A(I_i, S_i)  (S_i: the m_i separator rows block i touches)
//          [ P^T C ]
//
//      factor:  D_i^-1 (one workgroup per block, in LDS);  W_i = D_i^-1 P_i;  S = C - sum_i P_i^T W_i  (ns x ns, dense);  S^-1 by the
//               blocked Gauss-Jordan of the dense path -- (ns / n)^3 of its work: ns ~ 0.42 n on the Galerkin operators of a surface mesh
//      solve:   g = b_S - sum_i W_i^T b_i;   x_S = S^-1 g;   x_i = D_i^-1 b_i - W_i x_S
//
// Everything is a fixed sequence of sums (lists built here, ascending block), so results are bit-identical from run to run.  The
// triangle of A that counts is the lower one in the caller's numbering (SimplicialLDLT's convention, like the dense path).
#pragma once
#include <vector>

#include "smg_sparse.hpp"

namespace smg {

constexpr int SCHUR_B = 64;        // rows of an interior block at most
constexpr int SCHUR_M_MAX = 128;   // separator rows a block may touch at most (more: no plan, the dense inverse is used)
constexpr int SCHUR_NS_MAX = 24576; // separator rows at most (its inverse is dense: 4.8 GB there); more: no plan

struct SchurPlan {
    int n = 0, nb = 0;                     // unknowns; interior blocks
    int ns = 0, ns_pad = 0;                // separator rows; rounded up to a multiple of 64 (unit diagonal on the padding)
    std::vector<int> irow;                 // nb x 64: the row behind slot r of block i (-1: padding, unit diagonal)
    std::vector<int> bsize;                // nb: rows of block i (its slots 0 .. bsize[i] - 1)
    std::vector<int> srow;                 // ns: the row behind separator index j
    std::vector<int> sptr, sidx;           // block i touches separator indices sidx[sptr[i] .. sptr[i + 1]) (ascending): its local columns
    std::vector<int> aptr, ablk, apan;     // separator index j is panel row apan[q] (= sptr[i] + its local column) of block i = ablk[q], q in [aptr[j], aptr[j + 1]) (ascending block)
    // ---- one arena of doubles: [D: nb x 64 x 64][P: 64 sptr[nb], block i at 64 sptr[i], stored TRANSPOSED, P_i^T[c][r]][W: like P, W_i^T[c][r]]
    //      [S: ns_pad x ns_pad][C: block i's P_i^T W_i, m_i x m_i (lower triangle used) at coff[i]]
    long long off_D = 0, off_P = 0, off_W = 0, off_S = 0, off_C = 0, total = 0;
    std::vector<long long> coff;           // nb + 1
    std::vector<long long> pos, pos2;      // per stored entry of A: where its value goes in the arena (-1: nowhere; pos2: the mirror image)
    std::vector<long long> ones;           // arena positions of the unit diagonal of padding rows
    // S[rdst[d]] (and its mirror image S[rdst2[d]]) -= sum of arena[off_C + rsrc[q]], q in [rptr[d], rptr[d + 1])
    std::vector<long long> rdst, rdst2, rsrc;
    std::vector<int> rptr;
    bool empty() const { return nb == 0; }
};

// A: square, structurally symmetric, rows sorted, diagonal stored.  Empty plan when a block touches more than SCHUR_M_MAX separator rows,
// the separator is more than 0.7 n (nothing gained) or more than SCHUR_NS_MAX rows (its dense inverse would not be small).
SchurPlan build_schur(const Csr& A, int block_rows = SCHUR_B);

}  // namespace smg
