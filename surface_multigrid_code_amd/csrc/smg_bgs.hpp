// smg_bgs.hpp -- relax() for MANY right-hand sides (k a multiple of 64): block Gauss-Seidel with the block's iterate in LDS.
//
// The reference's relax() with k > 1 (src/mg_VCycle.cpp:161-177) is k independent lexicographic sweeps.  With one lane per COLUMN a
// wavefront works on one row at a time (k_sell_wide, KW = 64) and the multi-colour order of the narrow kernels costs this path 5.4 n k 8
// bytes per sweep instead of the 3 n k 8 a sweep moves algorithmically: every colour launch streams the iterate of the three other colours
// (512 B per row and column block: no cache holds that across launches; measured, profiles/r04_pmc_summary_C3_k64.json: 698 MB per colour
// launch at 5.2 TB/s, i.e. AT the memory's rate).
//
// Here the level is cut into compact BLOCKS of <= 64 rows (recursive breadth-first bisection, smg_tiled.cpp), the blocks are coloured
// (blocks of one colour share no matrix entry), and a sweep is one launch per BLOCK colour in which a workgroup of 4 waves owns a block:
// its 64 x 64 iterate values (rows x columns) are read ONCE into LDS, the block's rows are updated vertex colour by vertex colour (the
// colours of the level's numbering; a workgroup barrier between them, the rows of a colour dealt round-robin to the waves), every
// neighbour inside the block comes out of LDS -- old or new, whatever the order requires, because LDS is updated in place -- and only
// the block's rim (~0.65 n rows at 60-row blocks) is gathered from memory.  Per sweep the iterate is read ~1.65 times instead of 3.
//
// This IS the reference's lexicographic sweep on the numbering "block colour, block, vertex colour, row" (bgs order): per row the products
// are added in ascending column of THAT numbering, so the oracle on the permuted system reproduces it bit for bit (tests/test_gpu_bgs.py).
// It is another valid Gauss-Seidel order than the multi-colour one of the k < 64 kernels: iterates differ between the two paths,
// converged solutions do not.
//
// (Round 4 first built the walk as ONE wave per block, row after row, the fresh neighbours in a ring in LDS: bit-exact and slower than the
// colour launches -- 64 dependent rows of ~250 instructions per wave, 80 us per launch whatever the prefetch depth.  The vertex colours
// inside the block cut that chain to ~20 rows per wave.)
#pragma once
#include <vector>

#include "smg_sparse.hpp"

namespace smg {

constexpr int BGS_ROWS = 64;          // rows of a block at most (their iterate: 32 KB of LDS per workgroup)
constexpr int BGS_WAVES = 4;          // waves of the workgroup that owns a block
constexpr int BGS_BATCH = 8;          // entry slots per batch (a row holds NB batches, NB the same for all rows of a block)
constexpr int BGS_MAX_BATCHES = 2;    // per row: levels with rows of more than 16 entries keep the multi-colour launches
constexpr int BGS_LP_MAX = 8;         // rows per (block, vertex colour, wave) at most (8 rows x 8 slots = the 64 lanes of a metadata load)
constexpr int BGS_HDR = 5;            // ints per block header
constexpr int BGS_PAD = -1;           // entry codes below 0: padding,
constexpr int BGS_DIAG = -2;          //   the row's diagonal,
constexpr int BGS_LOCAL0 = -3;        //   row l of the block's LDS iterate as BGS_LOCAL0 - l

struct BgsPlan {
    int n = 0, n_blocks = 0, n_colors = 0;
    int lp = 0;                       // rows per (block, vertex colour, wave), padded: the same for the whole level (4 .. BGS_LP_MAX)
    std::vector<int> color_ptr;       // blocks of colour c: [color_ptr[c], color_ptr[c + 1])
    std::vector<int> blk_ptr;         // rows of block b: positions [blk_ptr[b], blk_ptr[b + 1]) of `rows`
    std::vector<int> rows;            // position in the bgs order -> row (internal numbering)
    // ---- what the kernel reads.  A unit = one (block, phase, wave): lp row slots, 64 NB entry slots (slot = row slot * 8 NB + entry; row
    // slots beyond lp and entries beyond the row's are padding).  A row slot the wave has no row for repeats a row of the SAME phase of the
    // block (updating a row twice within a phase reproduces its value: its neighbours belong to other phases) -- no tail code.
    std::vector<int> hdr;             // per block BGS_HDR ints: first unit, phases (vertex colours present), rows of the block, batches per row NB, first entry slot
    std::vector<int> brow;            // BGS_ROWS per block: row of local index l (beyond the block's rows: its first row again)
    std::vector<int> urow;            // 16 per unit: rows of the lp row slots [0, 8), their local indices [8, 16)
    std::vector<int> ecol;            // 64 NB per unit: >= 0 row to gather, else BGS_*
    std::vector<double> eval;
    std::vector<int> eentry;          // like eval: index of the entry of G the slot holds (-1: padding) -- value refresh
    double rim = 0.0;                 // (distinct (block, foreign row) pairs) / n: what a sweep gathers beyond the iterate itself
    double fill = 0.0;                // n / row slots: the share of the walk's row updates that are not repeats
    bool empty() const { return n_blocks == 0; }
};

// G: the matrix the smoother streams (A, or A^T where A is not bit-symmetric), internal numbering, structurally symmetric, diagonal stored.
// color_ptr: the vertex colours of the level's numbering (rows of colour c: [color_ptr[c], color_ptr[c + 1]); rows of a colour share no entry).
// Returns an empty plan when a row has no stored diagonal or more than BGS_MAX_BATCHES * BGS_BATCH entries, or a block holds more than
// BGS_WAVES * BGS_LP_MAX rows of one colour.
BgsPlan build_bgs(const Csr& G, const std::vector<int>& color_ptr, int block_rows = BGS_ROWS);

// compact parts of <= tile_rows rows (smg_tiled.cpp)
std::vector<int> partition_tiles(const Csr& G, int tile_rows, int* n_tiles);

}  // namespace smg
