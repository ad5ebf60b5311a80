// smg_bsr3.hpp -- the block (3 degrees of freedom per vertex) variant of the hot path.
//
// The reference's mg_precompute_block / get_prolong_block (src/mg_precompute_block.cpp:23-95, src/get_prolong.cpp:59-115) build
// P (x) I_3 with DOF index 3 v + d, and its caller (06_example_balloon_sim/sim_utils/implicit_euler_mg_balloon.h:63-76) hands
// min_quad_with_fixed_mg_precompute a 3n x 3n elasticity system whose entries come in 3 x 3 blocks per (vertex, neighbour) pair.
// The reference runs its scalar kernels on that matrix.  libsmg stores the level matrices of such a hierarchy in 3 x 3 blocks
// (76 bytes per block instead of 9 x 12 = 108 for nine scalar entries) and colours VERTICES: one lane per vertex updates its three
// DOFs 3v, 3v+1, 3v+2 in order -- which IS the reference's lexicographic sweep (src/mg_VCycle.cpp:146-160) on the colour-major
// vertex numbering, because the three rows of a vertex are consecutive and vertices of one colour do not touch each other.
// Launches per sweep = vertex colours (the scalar path needs at least three times as many: the DOFs of a vertex are coupled).
// P (x) I_3 is never stored on the device: the transfer kernels run on the vertex-level P with 3 k columns, which is the same
// arithmetic (row 3r+d of P (x) I_3 holds P(r, c) at column 3c+d).
#pragma once
#include <vector>

#include "smg_sparse.hpp"

namespace smg {

// SELL-64 over VERTICES with 3 x 3 blocks: slice s covers vertices [slice_row[s], slice_row[s+1]) (<= 64, never straddling a colour
// boundary), one lane per vertex; panel column j of the slice holds for every lane the j-th block of its block row in ascending
// block-column order: the block column (a vertex number, -1 = padding) and nine value planes, plane e = 3 * (row inside the block) +
// (column inside the block), each plane 64 contiguous doubles.  An entry the scalar matrix does not store is an explicit 0.0 in its
// block: a row sum then contains products 0 * x at the positions in between -- the same bits for finite x (a sum that starts at +0
// never becomes -0, so adding +-0 changes nothing).
struct Bsr3Sell {
    int n_vert = 0, n_slices = 0, w_max = 0;
    std::vector<int> slice_row;        // n_slices + 1 (vertex offsets)
    std::vector<int> slice_off;        // n_slices + 1, in panel columns
    std::vector<int> slice_w;          // n_slices
    raw_vector<int> col;               // 64 * slice_off.back()
    raw_vector<double> val;            // 9 * 64 * slice_off.back():  val[((off + j) * 9 + e) * 64 + lane]
    raw_vector<int> entry;             // like val: index of the scalar CSR entry the slot holds, -1 = explicit zero / padding (empty unless asked for)
    std::vector<int> color_slice_ptr;  // n_colors + 1 slice offsets
    std::vector<int> region_order;     // launch order of whole-matrix kernels (see Sell::region_order), or empty
    long nnz_scalar = 0;               // stored entries of the scalar matrix
    long n_blocks = 0;                 // stored 3 x 3 blocks
};

// A: 3 n_v x 3 n_v, rows / columns numbered 3 v + d in the INTERNAL vertex numbering, entries ascending inside a row.
// vertex_breaks: optional ascending vertex offsets (the vertex colouring's colour_ptr) at which a new slice starts.
// with_entry: also fill Bsr3Sell::entry (what the value-only recipes are built from; a third of the image's bytes)
Bsr3Sell build_bsr3(const Csr& A, const std::vector<int>* vertex_breaks, bool region_order, bool with_entry = true);

// The layout alone (slices, widths, offsets, colour / region tables) from the block-row lengths in the internal vertex numbering: col / val / entry stay
// empty -- the device fills the panels (launch_bsr3_fill).
Bsr3Sell bsr3_layout(const std::vector<int>& block_row_len, const std::vector<int>* vertex_breaks, bool region_order, long nnz_scalar, long n_blocks);

// n_v x n_v pattern of the 3 x 3 blocks of A (values 1.0); *n_blocks receives their number.  A.nr must be a multiple of 3.
Csr block_pattern3(const Csr& A);

// Is P = Pv (x) I_3 (row 3r+d holds exactly the entries Pv(r, c) at columns 3c+d, c ascending)?  On success Pv receives the factor.
bool kron3_factor(const Csr& P, Csr& Pv);
// Pv (x) I_3
Csr kron3(const Csr& Pv);

}  // namespace smg
