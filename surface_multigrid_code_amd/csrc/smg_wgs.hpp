// smg_wgs.hpp -- relax() on the Galerkin levels of the REFERENCE's own hierarchies (mg_precompute: SSP decimation, 3 entries per row of P,
// A_l = PT A P with 18 - 30 entries per row): Gauss-Seidel in a two-level order, one wavefront per PIECE of the level.
//
// The reference's relax() (src/mg_VCycle.cpp:113-178) is a lexicographic sweep; on the colour-major numbering one launch per colour
// reproduces it (smg_order.hpp).  A Galerkin level of a decimated hierarchy couples every vertex to its two-ring: 11 - 15 colours, rows of
// up to 50 entries -- a sweep is 11 - 15 launches of 5 - 7 us each whatever the size of the level (profiles/r05_dec_baseline.txt: the 15 804-row
// level of the decimated C3 hierarchy costs 426 us per visit, the 1 011 330-row level 264), and the halo of the overlapped tiling
// (smg_tiled.hpp: one ring per phase) would be the level itself.
//
// Here the level is cut into compact PIECES of <= 64 rows (the recursive bisection of smg_tiled.cpp), the piece graph is coloured (4 - 6
// colours: it is the map of a surface, whatever the degree of the rows), and a sweep is ONE LAUNCH PER PIECE COLOUR in which a wavefront owns a piece:
// lane = row.  The lane keeps its whole row -- values, and the positions of its columns in the piece's LDS image -- in registers (requested at
// once: one round trip, not one per batch of 8 as in the panel kernels); the image holds the piece's rows and its RIM (the rows of other pieces
// it reads, which no piece of this colour writes).  Inside the piece the rows are updated PHASE by phase: a phase is a set of rows that read
// none of each other and all of whose earlier neighbours (in the order below) sit in earlier phases -- the local colour classes of the piece,
// compressed by level scheduling.  A phase costs LDS latency, not a launch: ~0.2 us instead of ~5.
//
// This IS the reference's lexicographic sweep on the numbering "piece colour, piece, local colour, row" (wgs order): per row the products
// are added in ascending column of THAT numbering with separate multiply and add, so the oracle on the permuted system reproduces it bit for
// bit (tests/test_gpu_wgs.py; the order is exposed by smg_level_get_wave_gs_order).  It is another valid Gauss-Seidel order than the
// multi-colour one: iterates differ between the two paths, converged solutions and cycle counts do not.
#pragma once
#include <vector>

#include "smg_sparse.hpp"

namespace smg {

constexpr int WGS_ROWS = 64;          // rows of a piece at most: one per lane
constexpr int WGS_RIM_MAX = 448;      // rows of other pieces a piece may read at most (its rim): the LDS image has <= 512 rows
constexpr int WGS_BATCH = 8;          // entry slots per batch; a piece's rows all hold NB batches (NB = what its widest row needs)
constexpr int WGS_MAX_BATCHES = 8;    // rows of more than 64 off-diagonal entries: the level keeps the colour launches
constexpr int WGS_HDR = 8;            // ints per piece: [0] first entry slot  [1] NB  [2] rim rows  [3] phases  [4] first rim slot  [5] rows  [6] reserved  [7] reserved

inline int wgs_rim_pitch(int max_rim) { return max_rim <= 2 * WGS_ROWS ? 2 * WGS_ROWS : max_rim <= 4 * WGS_ROWS ? 4 * WGS_ROWS : 7 * WGS_ROWS; }

struct WgsPlan {
    int n = 0, n_pieces = 0, n_colors = 0;
    int rim_pitch = 0;                // rim slots per piece: wgs_rim_pitch(the level's largest rim) = 128, 256 or 448 (the kernel's variants); unused slots repeat the piece's first row
    std::vector<int> color_ptr;       // pieces of colour c: [color_ptr[c], color_ptr[c + 1])
    std::vector<int> piece_ptr;       // rows of piece q: positions [piece_ptr[q], piece_ptr[q + 1]) of `rows`
    std::vector<int> rows;            // position in the wgs order -> row (internal numbering)
    // ---- what the kernel reads
    std::vector<int> hdr;             // WGS_HDR per piece
    std::vector<int> grow;            // 64 per piece: the lane's row (-1: the lane has none)
    std::vector<int> meta;            // 64 per piece: the phase the lane's row is updated in (bits 0 - 15; lanes without a row: 0xffff) | batches of 8 entry slots its row needs << 16
    std::vector<double> diag;         // 64 per piece: a_ii (lanes without a row: 1.0)
    std::vector<int> rim;             // rim_pitch per piece: the rows behind local indices 64, 65, ...
    std::vector<unsigned> eoff;       // per piece 64 x 4 NB words: the BYTE offsets (8 x local index) of entry slots 2 t and 2 t + 1 of the lane's row in the
                                      // piece's one-column image, packed 16 + 16 bits; word (t, lane) at [first slot / 2 + 64 t + lane]
    std::vector<double> eval;         // per piece 64 x 8 NB: the values, slot (t, lane) at [first slot + 64 t + lane]; off-diagonal entries in ascending column of the wgs order,
                                      // padding: +0.0 at the lane's own local index
    std::vector<int> eentry, dentry;  // like eval / diag: index of the entry of G the slot holds (-1: padding) -- value refresh
    double rim_ratio = 0.0;           // (distinct (piece, foreign row) pairs) / n: what a sweep gathers beyond the iterate itself
    double phases_mean = 0.0;         // phases per piece (the critical path of a launch is that of its deepest piece)
    int phases_max = 0;
    int nb_max = 0;                   // batches per row of the level's widest row
    bool empty() const { return n_pieces == 0; }
};

// G: the matrix the smoother streams (A, or A^T where A is not bit-symmetric), internal numbering, structurally symmetric, diagonal stored.
// mode 0: compact pieces (partition_tiles); 1: pieces along the breadth-first level sets of G (thin bands: fewer local colours, larger rims).
// Returns an empty plan when a row has no stored diagonal or more than WGS_MAX_BATCHES * WGS_BATCH off-diagonal entries, or a piece cannot be
// brought below WGS_RIM_MAX rim rows.
WgsPlan build_wgs(const Csr& G, int piece_rows = WGS_ROWS, int mode = 0);

// pieces of <= piece_rows rows cut along the breadth-first level sets of G (smg_wgs.cpp)
// hint (optional): a colour 0 .. 3 per piece -- 2 x (parity of the level set) + (parity of the run inside it)
std::vector<int> partition_bands(const Csr& G, int piece_rows, int* n_pieces, std::vector<int>* hint = nullptr);

// The plan executed on the host the way k_wgs executes it (one column, in place on u) -- the checker of the plan's bookkeeping (tests, CPU lane).
void wgs_sweep_host(const WgsPlan& P, const double* b, double* u);

}  // namespace smg
