// smg_tiled.hpp -- relax(iters) of a latency-bound level as ONE launch: overlapped (temporal) tiling of the multi-colour Gauss-Seidel
// sweeps.
//
// The reference's relax() (src/mg_VCycle.cpp:113-178) on the colour-major numbering is iters x n_colours dependent phases; as one
// launch per phase a level of 16 k - 250 k rows costs ~3.2 us per phase of pure launch + dependent-load latency (DESIGN.md section 3:
// 173 of the 347 us of a C3 cycle).  Here a workgroup owns a TILE of rows (a compact part of the level's graph, found by recursive
// breadth-first bisection) and computes every phase itself, on the tile plus a halo: phase p of
// P = iters x n_colours needs correct neighbours one ring further out than what it updates, so the workgroup carries the rows within
// P - p rings of its tile through phase p, redundantly with the neighbouring tiles, and reads the rows within P rings as input.
// The iterate of the extended tile lives in LDS; phases are separated by workgroup barriers; only the owned rows are written back.
// Every row update is the same expression on the same operands in the same order (ascending column of the internal numbering) as the
// per-colour launch performs: the results are bit-identical -- the redundancy is in WHO computes a value, not in what is computed.
// Out of place (x -> y): a tile reads halo rows that another tile owns and may already have finished.
#pragma once
#include <vector>

#include "smg_sparse.hpp"

namespace smg {

constexpr int TILED_NCMAX = 5;     // colours
constexpr int TILED_PMAX = 15;     // phases = sweeps x colours
constexpr int TILED_WMAX = 12;     // stored entries per row
constexpr int TILED_THREADS = 512; // threads of a tile's workgroup = most rows a colour may have in a tile's panel (one row per thread)
// per tile: [0] ext_off  [1] n_ext  [2] W  [3] reserved, then per colour c (stride 4 + TILED_PMAX + 1):
//   [0] panel offset (entries)  [1] rows in the panel m_c (those within P - c - 1 rings: what the colour's first phase updates)  [2] row offset (prow)
//   [3] first local index of the colour's rows
//   [4 + d] rows of the colour within d rings of the tile, d = 0 .. TILED_PMAX
constexpr int TILED_CSTRIDE = 4 + TILED_PMAX + 1;
constexpr int TILED_HDR = 4 + TILED_NCMAX * TILED_CSTRIDE;
constexpr int TILED_LDS_STATIC = 512;   // bytes of static LDS of k_tiled_gs (the tile header, TILED_HDR ints) on top of the dynamic iterate: counted in every LDS bound

struct TiledGs {
    int n_tiles = 0, nc = 0, sweeps = 0, P = 0, max_ext = 0;
    long updates = 0;                // row updates per relax() over all tiles and phases (redundancy = updates / (sweeps * n))
    std::vector<int> hdr;            // n_tiles * TILED_HDR
    std::vector<int> ext_rows;       // global row of every local index, tile after tile
    // panels, column-major per (tile, colour): local column + value.  The DIAGONAL has left the row: its slot holds +0.0 at the row's own local index, like every
    // padding slot (a sum that starts at +0 is not changed by adding +-0: the bits are those of the sum that skips these slots), and a_ii sits in pdiag --
    // the kernel's phases are W straight-line LDS reads and multiply-adds, and nothing is sorted out per launch (round 6: that cost 2 us of every launch).
    std::vector<int> pcol;
    std::vector<double> pval;
    std::vector<int> pentry;         // like pval: index of the entry of G the slot holds (-1: padding / the diagonal's slot): value refresh
    std::vector<int> prow;           // global row of every panel row
    std::vector<double> pdiag;       // like prow: a_ii (1.0 for a row without a stored diagonal)
    std::vector<int> pdentry;        // like prow: index of the diagonal entry of G (-1: none): value refresh
    bool empty() const { return n_tiles == 0; }
};

// G: the matrix the smoother streams (A, or A^T where A is not bit-symmetric), internal numbering, structurally symmetric.
// Returns an empty plan when the level does not qualify (too many colours / phases, a row wider than TILED_WMAX, halo too large
// for max_ext local rows or a colour's panel for max_panel_rows).
TiledGs build_tiled_gs(const Csr& G, const std::vector<int>& color_ptr, int sweeps, int tile_rows, int max_ext, int max_panel_rows = TILED_THREADS);

}  // namespace smg
