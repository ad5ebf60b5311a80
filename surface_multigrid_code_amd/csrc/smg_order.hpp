// smg_order.hpp -- device-side numbering of one multigrid level and the SELL-64-sigma container.
//
// The reference smoother is a forward *lexicographic* Gauss-Seidel sweep (src/mg_VCycle.cpp:146-160),
// inherently sequential.  libsmg renumbers the unknowns of every level colour-major (greedy colouring of
// the pattern of A_l; reverse-Cuthill-McKee rank inside a colour for gather locality; rows bucketed by nnz
// inside sigma-row windows to keep SELL padding small).  A lexicographic sweep in THAT numbering visits
// colour 0, then colour 1, ... and rows of one colour never reference each other, so one kernel launch per
// colour reproduces the reference sweep on the renumbered system exactly (bit for bit, given the same
// ascending-column accumulation order per row).
#pragma once
#include <cstdint>
#include <vector>

#include "smg_sparse.hpp"

namespace smg {

struct Ordering {
    std::vector<int> perm;       // internal (new) -> caller (old)
    std::vector<int> iperm;      // caller (old) -> internal (new)
    std::vector<int> color_ptr;  // n_colors + 1 row offsets in the internal numbering
    std::vector<int> color_of;   // colour of every vertex, caller numbering (feeds the next finer level)
    int n_colors() const { return (int)color_ptr.size() - 1; }
};

std::vector<int> rcm_order(const Csr& A);                    // returns new -> old
std::vector<int> rcm_order_arrays(int n, const int* ptr, const int* col);   // the same on raw CSR arrays (rows sorted)
// Locality order of a fine level induced by its coarse level: fine vertex i goes where its parent -- the column of the largest
// weight in row i of P (fine x coarse) -- sits in the coarse order (coarse_rank: old -> position), children of one parent
// together.  O(nnz(P)), against a sequential breadth-first search over the fine matrix for RCM.
std::vector<int> induced_order(const Csr& P, const std::vector<int>& coarse_rank);
// A: square, structurally symmetric.  rcm: a precomputed rcm_order(A) (lets callers run the per-level RCMs concurrently).
// preset_final: the preset IS colours_for_ordering(A, rcm), computed ahead on another thread -- used as it stands, so that the numbering is the one
// make_ordering(A, sigma, nullptr, rcm) builds.
Ordering make_ordering(const Csr& A, int sigma = 512, const std::vector<int>* preset_colors = nullptr, const std::vector<int>* rcm = nullptr, bool preset_final = false);
// the from-scratch colouring make_ordering computes when no colours are handed in (rcm: rcm_order(A))
std::vector<int> colours_for_ordering(const Csr& A, const std::vector<int>& rcm);
// 4-colouring of a mid-point-subdivided level from a 4-colouring of its parent; false if P / A do not fit
bool subdivision_colors(const Csr& P, const std::vector<int>& coarse_color, const Csr& A, std::vector<int>& out);
Ordering identity_ordering(int n);                           // single "colour" (debug / non-smoothed levels)

constexpr int SELL_C = 64;  // slice height = one wavefront, one row per lane (two rows per lane / C = 128 measured slower)

// SELL-C-sigma, C = 64: slice s covers rows [slice_row[s], slice_row[s+1]) (<= 64 of them; a slice never
// straddles a colour boundary); its entries are stored column-major in a 64-wide panel starting at element
// 64 * slice_off[s]; panel width slice_w[s] = longest row of the slice.  Padding has col = -1.  Inside a row the
// stored order is ascending column index of the *internal* numbering.
// Fixed stride: when the widest slice is not much wider than the average (all mesh operators here), every panel gets
// `stride` columns of room, slice_off[s] = s * stride, and a kernel can address a panel -- and load its first w_lo
// columns, which (nearly) every slice has, the others hold padding there -- without first reading any per-slice table: one dependent memory round trip less per
// launch.  stride = 0: compact panels, slice_off is a prefix sum of the widths.
struct Sell {
    int C = SELL_C;                    // slice height: 64 (one row per lane) or 128 (two adjacent rows per lane)
    int n_rows = 0, n_cols = 0, n_slices = 0;
    std::vector<int> slice_row;        // n_slices + 1
    std::vector<int> slice_off;        // n_slices + 1, in units of C entries (panel columns)
    std::vector<int> slice_w;          // n_slices: panel width actually used by the slice
    int stride = 0;                    // > 0: slice_off[s] = s * stride
    int w_lo = 0;                      // columns to request ahead of the table: covers 90% of the slices (0 when stride == 0)
    std::vector<int> col;              // 64 * slice_off[n_slices]
    std::vector<double> val;
    std::vector<int> color_slice_ptr;  // n_colors + 1 slice offsets (single range when uncoloured)
    long nnz = 0;                      // stored (unpadded) entries
    long padded() const { return (long)C * (slice_off.empty() ? 0 : slice_off.back()); }   // allocated slots
    long used() const { long t = 0; for (int w : slice_w) t += w; return (long)C * t; }          // slots the kernels read
    // Launch order of the slices for whole-matrix kernels: sorted by relative position inside the colour block,
    // so that consecutive logical blocks (= one XCD's share) cover ONE mesh region across all colours.
    std::vector<int> region_order;     // n_slices, or empty (identity)
    std::vector<int> entry;            // per stored slot: index of the CSR entry it holds (-1 = padding); value refresh map
};

// row_breaks: optional ascending row offsets (e.g. Ordering::color_ptr) at which a new slice must start.
Sell build_sell(const Csr& A, const std::vector<int>* row_breaks, int C = SELL_C, bool region_order = false, int pitch_policy = -1);
// the same container from the row lengths alone: slice tables, pitch, launch order -- col / val / entry stay empty (the panels are then
// filled on the device, smg_device.hpp: launch_sell_fill)
// pitch_policy: 1 fixed panel pitch where it costs <= 2.5 x (the default), 0 compact panels + slice_off table (smg_hierarchy_set_memory_lean), -1: SMG_SELL_STRIDE or 1
Sell sell_layout(const std::vector<int>& row_len, int n_cols, long nnz, const std::vector<int>* row_breaks, int C = SELL_C, bool region_order = false, int pitch_policy = -1);

}  // namespace smg
