// smg_bgs.hpp -- relax() for MANY right-hand sides (k a multiple of 64): block-sequential Gauss-Seidel.
//
// The reference's relax() with k > 1 (src/mg_VCycle.cpp:161-177) is k independent lexicographic sweeps.  With one lane per COLUMN a
// wavefront works on one row at a time (k_sell_wide, KW = 64), so nothing forces the rows a wave handles to be mutually independent: the
// multi-colour order of the narrow kernels costs this path 5.4 n k 8 bytes per sweep instead of the 3 n k 8 a sweep moves algorithmically --
// every colour launch streams the iterate of the three other colours (512 B per row and column block: no cache holds the reuse across
// launches; measured, profiles/r04_pmc_summary_C3_k64.json: 698 MB per colour launch at 5.2 TB/s, i.e. AT the memory's rate).
//
// Here the level is cut into compact BLOCKS of <= 64 rows (recursive breadth-first bisection, smg_tiled.cpp), the blocks are coloured
// (blocks of one colour share no matrix entry), and a sweep is one launch per BLOCK colour in which a wavefront walks its block row by row:
//   * a neighbour inside the block that was updated a few rows ago comes out of a ring of the last BGS_RING new values in LDS;
//   * every other neighbour (other blocks: not touched by this launch; later rows of the block: still old; earlier rows beyond the
//     ring: stored before the gather is issued, same lane, program order) is gathered from memory, one row ahead of the arithmetic.
// Per sweep the iterate is read once plus the blocks' rims (~0.56 n rows at 60-row blocks) instead of three times.
//
// This IS the reference's lexicographic sweep on the numbering "block colour, block, position in the block" (bgs order): per row the
// products are added in ascending column of THAT numbering, so the oracle on the permuted system reproduces it bit for bit
// (tests/test_gpu_bgs.py).  It is another valid Gauss-Seidel order than the multi-colour one of the k < 64 kernels: iterates differ
// between the two paths, converged solutions do not.
#pragma once
#include <vector>

#include "smg_sparse.hpp"

namespace smg {

constexpr int BGS_RING = 16;          // new values of the last BGS_RING rows of a block live in LDS
constexpr int BGS_BATCH = 8;          // entries per batch (a row holds a whole number of batches)
constexpr int BGS_MAX_BATCHES = 2;    // per row: levels with rows of more than 16 entries keep the multi-colour launches
constexpr int BGS_PAD = -1;           // entry column codes below 0: padding,
constexpr int BGS_DIAG = -2;          //   the row's diagonal,
constexpr int BGS_RING0 = -3;         //   ring slot s as BGS_RING0 - s

struct BgsPlan {
    int n = 0, n_blocks = 0, n_colors = 0;
    std::vector<int> color_ptr;       // blocks of colour c: [color_ptr[c], color_ptr[c + 1])
    std::vector<int> blk_ptr;         // rows of block b: positions [blk_ptr[b], blk_ptr[b + 1]) of `rows`
    std::vector<int> rows;            // position in the bgs order -> row (internal numbering)
    // ---- what the kernel reads.  A wavefront takes its block in CHUNKS of 64 entry slots = 8 rows of one batch (4 rows of two: every
    // row of a block holds the same number of batches, so a row's entries are found from its position alone); a block whose row count is
    // no multiple of that is padded with copies of its LAST row -- updating a row again with unchanged neighbours reproduces its value,
    // so the copies are harmless and the walk needs no tail code.
    std::vector<int> hdr;             // per block 4 ints: offset into prow, rows of the block (without copies), first chunk, batches per row
    std::vector<int> prow;            // rows of the blocks, padded per block to whole chunks
    std::vector<int> ecol;            // chunks * 64 entry codes: >= 0 row to gather, else BGS_*
    std::vector<double> eval;
    std::vector<int> eentry;          // like eval: index of the entry of G the slot holds (-1: padding) -- value refresh
    double rim = 0.0;                 // (distinct (block, foreign row) pairs) / n: what a sweep gathers beyond the iterate itself
    double ring_hits = 0.0;           // share of the in-block earlier neighbours served by the ring
    bool empty() const { return n_blocks == 0; }
};

// G: the matrix the smoother streams (A, or A^T where A is not bit-symmetric), internal numbering, structurally symmetric, diagonal stored.
// Returns an empty plan when a row has no stored diagonal or more than BGS_MAX_BATCHES * BGS_BATCH entries.
BgsPlan build_bgs(const Csr& G, int block_rows = 64);

// compact parts of <= tile_rows rows (smg_tiled.cpp)
std::vector<int> partition_tiles(const Csr& G, int tile_rows, int* n_tiles);

}  // namespace smg
