// smg_bgs.hpp -- relax() for MANY right-hand sides (k a multiple of 16): block Gauss-Seidel with the block's iterate in LDS.
//
// The reference's relax() with k > 1 (src/mg_VCycle.cpp:161-177) is k independent lexicographic sweeps.  With one lane per COLUMN a
// wavefront works on one row at a time (k_sell_wide, KW = 64) and the multi-colour order of the narrow kernels costs this path 5.4 n k 8
// bytes per sweep instead of the 3 n k 8 a sweep moves algorithmically: every colour launch streams the iterate of the three other colours
// (512 B per row and column block: no cache holds that across launches; measured, profiles/r04_pmc_summary_C3_k64.json: 698 MB per colour
// launch at 5.2 TB/s, i.e. AT the memory's rate).
//
// Here the level is cut into compact BLOCKS of <= 64 rows (recursive breadth-first bisection, smg_tiled.cpp), the blocks are coloured
// (blocks of one colour share no matrix entry), and a sweep is one launch per BLOCK colour in which a wavefront owns (block, 16 columns):
// the block's rows AND its rim (the rows of other blocks it reads: ~0.65 per row) are read once, 128 B per row, into 16 KB of LDS; then
// the lanes are 16 ROWS x 4 columns -- a lane holds its row's entries (local indices into the LDS image, values) in registers, like the
// one-lane-per-row kernels of the k < 8 path, and re-uses them for all 16 columns -- and the block's rows are updated UNIT by unit, in place
// in LDS: a unit is <= 16 rows none of which reads another (level scheduling of the block's rows in the bgs order: a row joins the first
// unit after those of the earlier rows it reads), so every later row finds old or new neighbours as the order requires; results go
// straight to memory.  No barrier, no wave-uniform bookkeeping per row: that is what sank the first two designs
// (profiles/r04_bgs_experiments.txt).  Per sweep the iterate is read ~1.65 times instead of 3.
//
// This IS the reference's lexicographic sweep on the numbering "block colour, block, vertex colour, row" (bgs order; the units only group rows of that order that do not read each other): per row the products
// are added in ascending column of THAT numbering, so the oracle on the permuted system reproduces it bit for bit (tests/test_gpu_bgs.py).
// It is another valid Gauss-Seidel order than the multi-colour one of the other kernels: iterates differ between the two paths,
// converged solutions do not.
//
// (Round 4 first built the walk with lanes across the 64 COLUMNS and a wave handling one row at a time -- one wave per block with a ring of
// fresh values, then a workgroup per block with vertex-colour phases: bit-exact both, and 1.6 - 2 x SLOWER than the colour launches,
// ~1 us of wave-uniform control per row.)
#pragma once
#include <vector>

#include "smg_sparse.hpp"

namespace smg {

constexpr int BGS_ROWS = 64;          // rows of a block at most
constexpr int BGS_RIM_GOAL = 64;      // blocks with a larger rim are cut in two by the plan (the LDS image of every block is sized by the level's largest rim)
constexpr int BGS_RIM_MAX = 192;      // rows of other blocks a block may read at most (its rim); a level with a larger rim keeps the colour launches
constexpr int BGS_COLS = 16;          // columns per wavefront: a wave owns (block, 16 columns); 16 x xrows doubles of LDS (16 KB at a rim of 64)
constexpr int BGS_UROWS = 16;         // rows per unit: lanes = 16 rows x 4 columns, four passes for the wave's 16 columns
constexpr int BGS_BATCH = 8;          // entry slots per batch (a row holds NB batches, NB the same for all rows of a block)
constexpr int BGS_MAX_BATCHES = 2;    // per row: levels with rows of more than 16 off-diagonal entries keep the multi-colour launches
constexpr int BGS_HDR = 5;            // ints per block header: first unit, units, batches per row NB, first entry slot, local rows in use (multiple of 64)

struct BgsPlan {
    int n = 0, n_blocks = 0, n_colors = 0;
    int xrows = 0;                    // rows of a block's extended iterate in LDS: BGS_ROWS + the largest rim of the level, rounded up to a multiple of 128;
                                      // local index l < BGS_ROWS: own row, else rim row l - BGS_ROWS
    std::vector<int> color_ptr;       // blocks of colour c: [color_ptr[c], color_ptr[c + 1])
    std::vector<int> blk_ptr;         // rows of block b: positions [blk_ptr[b], blk_ptr[b + 1]) of `rows`
    std::vector<int> rows;            // position in the bgs order -> row (internal numbering)
    // ---- what the kernel reads.  A unit = up to 16 rows of a block that read none of each other (level scheduling over the block's rows in
    // the bgs order); row slots without a row of their own repeat the unit's first row (the same value is computed and stored twice: harmless).
    std::vector<int> hdr;             // per block BGS_HDR ints
    std::vector<int> xrow;            // xrows per block: the row behind local index l (unused slots: the block's first row)
    std::vector<int> ugrow, ulrow;    // 16 per unit: row, local index of the row
    std::vector<double> udiag;        // 16 per unit: a_ii
    std::vector<int> eidx;            // 16 x 8 NB per unit: local index of the entry's column (padding: 0 with value +0.0)
    std::vector<double> eval;         //   ... its value; off-diagonal entries in ascending column of the bgs order
    std::vector<int> eentry, dentry;  // like eval / udiag: index of the entry of G the slot holds (-1: padding) -- value refresh
    double rim = 0.0;                 // (distinct (block, foreign row) pairs) / n: what a sweep gathers beyond the iterate itself
    double fill = 0.0;                // n / row slots: the share of the walk's row updates that are not repeats
    bool empty() const { return n_blocks == 0; }
};

// G: the matrix the smoother streams (A, or A^T where A is not bit-symmetric), internal numbering, structurally symmetric, diagonal stored.
// color_ptr: the vertex colours of the level's numbering (rows of colour c: [color_ptr[c], color_ptr[c + 1]); rows of a colour share no entry).
// Returns an empty plan when a row has no stored diagonal or more than BGS_MAX_BATCHES * BGS_BATCH off-diagonal entries, or a block reads
// more than BGS_RIM_MAX rows of other blocks.
BgsPlan build_bgs(const Csr& G, const std::vector<int>& color_ptr, int block_rows = BGS_ROWS);

// compact parts of <= tile_rows rows (smg_tiled.cpp)
std::vector<int> partition_tiles(const Csr& G, int tile_rows, int* n_tiles);

}  // namespace smg
