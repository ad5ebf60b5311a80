// smg_tiled.cpp -- host side of the overlapped tiling of the Gauss-Seidel sweeps (smg_tiled.hpp): tiles, halo rings, tile-local panels.
#include "smg_tiled.hpp"
#include "smg_bgs.hpp"

#include <algorithm>
#include <array>

namespace smg {

namespace {
struct TileData {
    std::vector<int> ext;                                 // local -> global
    std::array<int, TILED_NCMAX> n_loc{}, m{};            // rows of the colour in the extended tile / in the panel
    std::array<std::array<int, TILED_PMAX + 1>, TILED_NCMAX> cnt{};
    int W = 0;
    std::array<std::vector<int>, TILED_NCMAX> pcol, pentry, prow, pdentry;
    std::array<std::vector<double>, TILED_NCMAX> pval, pdiag;
    long updates = 0;
    bool ok = true;
};
}  // namespace

// Compact tiles: recursive bisection of the level's graph by breadth-first distance from a pseudo-peripheral vertex of the part (an
// elongated part is cut across its long axis, so the parts come out roundish: the halo of P rings around a tile of N rows then has
// ~ P * sqrt(N) rows, not ~ P * N / band width as for a range of the locality order, which is a thin band of the mesh).
// Returns the tile of every row; *n_tiles receives their number.  O(n log(n / tile_rows)).
std::vector<int> partition_tiles(const Csr& G, int tile_rows, int* n_tiles)
{
    const int n = G.nr;
    std::vector<int> part((size_t)n, 0), order((size_t)n), dist((size_t)n, -1);
    for (int i = 0; i < n; i++) order[(size_t)i] = i;
    struct Seg { int b, e; };
    std::vector<Seg> cur{{0, n}}, done;
    std::vector<int> stamp((size_t)n, -1);       // stamp[v] = id of the segment v currently belongs to
    int id_base = 0;
    // Depth by depth, the segments of a depth side by side on the host threads: a segment's cut reads and writes the entries of its own
    // vertices only (order / dist / stamp are indexed by position resp. vertex), so the result does not depend on the schedule.  (At a
    // million rows the 14 depths of sequential breadth-first passes were 0.4 of the 0.7 s a block Gauss-Seidel plan took to build.)
    while (!cur.empty()) {
        std::vector<Seg> left(cur.size()), right(cur.size());
        std::vector<char> split(cur.size(), 0);
        parallel_for((long)cur.size(), 1, [&](long s0, long s1) {
            std::vector<int> queue;
            for (long s = s0; s < s1; s++) {
                const Seg sg = cur[(size_t)s];
                if (sg.e - sg.b <= tile_rows) continue;
                const int id = id_base + (int)s;
                // (stamp is the one array threads read outside their own segment -- a neighbour across the cut -- while its owner may be restamping it:
                //  relaxed atomics; a foreign stamp never equals this segment's id, old or new)
                for (int i = sg.b; i < sg.e; i++) { __atomic_store_n(&stamp[(size_t)order[(size_t)i]], id, __ATOMIC_RELAXED); dist[(size_t)order[(size_t)i]] = -1; }
                // breadth-first order of the segment from `root`, unreached vertices (other components) appended; returns the last vertex reached
                auto bfs = [&](int root) {
                    queue.clear();
                    for (int i = sg.b; i < sg.e; i++) dist[(size_t)order[(size_t)i]] = -1;
                    size_t head = 0;
                    int scan = sg.b;
                    int cur_root = root;
                    while ((int)queue.size() < sg.e - sg.b) {
                        if (head == queue.size()) {      // start (or another component)
                            while (cur_root < 0 || dist[(size_t)cur_root] >= 0) { cur_root = order[(size_t)scan]; scan++; }
                            dist[(size_t)cur_root] = queue.empty() ? 0 : dist[(size_t)queue.back()] + 1;
                            queue.push_back(cur_root);
                            cur_root = -1;
                        }
                        const int v = queue[head++];
                        for (int p = G.ptr[(size_t)v]; p < G.ptr[(size_t)v + 1]; p++) {
                            const int q = G.col[(size_t)p];
                            if (__atomic_load_n(&stamp[(size_t)q], __ATOMIC_RELAXED) == id && dist[(size_t)q] < 0) { dist[(size_t)q] = dist[(size_t)v] + 1; queue.push_back(q); }
                        }
                    }
                    return queue.back();
                };
                const int far1 = bfs(order[(size_t)sg.b]);
                bfs(far1);                               // from the far end: the cut runs across the long axis
                for (int i = 0; i < sg.e - sg.b; i++) order[(size_t)sg.b + i] = queue[(size_t)i];
                const int mid = sg.b + (sg.e - sg.b) / 2;
                left[(size_t)s] = {sg.b, mid};
                right[(size_t)s] = {mid, sg.e};
                split[(size_t)s] = 1;
            }
        });
        id_base += (int)cur.size();
        std::vector<Seg> nxt;
        for (size_t s = 0; s < cur.size(); s++) {
            if (split[s]) { nxt.push_back(left[s]); nxt.push_back(right[s]); }
            else done.push_back(cur[s]);
        }
        cur.swap(nxt);
    }
    // tiles numbered along the final order: neighbours in the numbering are neighbours in the bisection tree, i.e. in space
    std::sort(done.begin(), done.end(), [](const Seg& a, const Seg& c) { return a.b < c.b; });
    *n_tiles = (int)done.size();
    for (size_t t = 0; t < done.size(); t++)
        for (int i = done[t].b; i < done[t].e; i++) part[(size_t)order[(size_t)i]] = (int)t;
    return part;
}

TiledGs build_tiled_gs(const Csr& G, const std::vector<int>& cp, int sweeps, int tile_rows, int max_ext, int max_panel_rows)
{
    TiledGs R;
    const int nc = (int)cp.size() - 1, n = G.nr, P = sweeps * nc;
    if (nc < 2 || nc > TILED_NCMAX || sweeps < 1 || P > TILED_PMAX || n == 0 || tile_rows < nc) return R;
    for (int r = 0; r < n; r++) if (G.ptr[(size_t)r + 1] - G.ptr[(size_t)r] > TILED_WMAX) return R;
    std::vector<signed char> color_of((size_t)n);
    for (int c = 0; c < nc; c++) for (int r = cp[(size_t)c]; r < cp[(size_t)c + 1]; r++) color_of[(size_t)r] = (signed char)c;
    int T = 0;
    const std::vector<int> part = partition_tiles(G, tile_rows, &T);
    std::vector<int> tile_ptr((size_t)T + 1, 0), tile_rows_list((size_t)n);
    for (int r = 0; r < n; r++) tile_ptr[(size_t)part[(size_t)r] + 1]++;
    for (int t = 0; t < T; t++) tile_ptr[(size_t)t + 1] += tile_ptr[(size_t)t];
    {
        std::vector<int> fill(tile_ptr.begin(), tile_ptr.end() - 1);
        for (int r = 0; r < n; r++) tile_rows_list[(size_t)fill[(size_t)part[(size_t)r]]++] = r;     // ascending inside a tile
    }
    std::vector<TileData> tiles((size_t)T);
    parallel_for(T, 1, [&](long t0, long t1) {
        std::vector<int> dist((size_t)n, -1), loc((size_t)n, -1), frontier, next, touched;
        for (long t = t0; t < t1; t++) {
            TileData& D = tiles[(size_t)t];
            // owned rows: one compact part of the level's graph (all colours)
            frontier.clear(); touched.clear();
            for (int i = tile_ptr[(size_t)t]; i < tile_ptr[(size_t)t + 1]; i++) {
                const int r = tile_rows_list[(size_t)i];
                dist[(size_t)r] = 0; frontier.push_back(r); touched.push_back(r);
            }
            std::array<std::array<std::vector<int>, TILED_PMAX + 1>, TILED_NCMAX> bucket;
            for (int r : frontier) bucket[(size_t)color_of[(size_t)r]][0].push_back(r);
            for (int d = 1; d <= P; d++) {
                next.clear();
                for (int r : frontier)
                    for (int p = G.ptr[(size_t)r]; p < G.ptr[(size_t)r + 1]; p++) {
                        const int q = G.col[(size_t)p];
                        if (dist[(size_t)q] < 0) { dist[(size_t)q] = d; next.push_back(q); touched.push_back(q); bucket[(size_t)color_of[(size_t)q]][(size_t)d].push_back(q); }
                    }
                frontier.swap(next);
            }
            if ((int)touched.size() > max_ext) D.ok = false;
            if (D.ok) {
                // local numbering: colour-major, inside a colour by ring, inside a ring by global row (locality of the x loads)
                D.ext.reserve(touched.size());
                for (int c = 0; c < nc; c++) {
                    int run = 0;
                    for (int d = 0; d <= P; d++) {
                        std::vector<int>& bk = bucket[(size_t)c][(size_t)d];
                        std::sort(bk.begin(), bk.end());
                        for (int r : bk) { loc[(size_t)r] = (int)D.ext.size(); D.ext.push_back(r); }
                        run += (int)bk.size();
                        D.cnt[(size_t)c][(size_t)d] = run;
                    }
                    for (int d = P + 1; d <= TILED_PMAX; d++) D.cnt[(size_t)c][(size_t)d] = run;
                    D.n_loc[(size_t)c] = run;
                    D.m[(size_t)c] = D.cnt[(size_t)c][(size_t)(P - c - 1)];      // the colour's first phase is phase c + 1: rows within P - (c + 1) rings; later phases fewer
                }
                for (int c = 0; c < nc; c++) if (D.m[(size_t)c] > max_panel_rows) D.ok = false;
            }
            if (D.ok) {
                int W = 0, base = 0;
                for (int c = 0; c < nc; c++) {
                    for (int i = 0; i < D.m[(size_t)c]; i++) { const int r = D.ext[(size_t)base + i]; W = std::max(W, G.ptr[(size_t)r + 1] - G.ptr[(size_t)r]); }
                    base += D.n_loc[(size_t)c];
                }
                D.W = W;
                base = 0;
                for (int c = 0; c < nc; c++) {
                    const int m = D.m[(size_t)c];
                    D.pcol[(size_t)c].assign((size_t)W * m, 0);
                    D.pentry[(size_t)c].assign((size_t)W * m, -1);
                    D.pval[(size_t)c].assign((size_t)W * m, 0.0);
                    D.prow[(size_t)c].resize((size_t)m);
                    D.pdiag[(size_t)c].assign((size_t)m, 1.0);
                    D.pdentry[(size_t)c].assign((size_t)m, -1);
                    for (int i = 0; i < m; i++) {
                        const int r = D.ext[(size_t)base + i];
                        D.prow[(size_t)c][(size_t)i] = r;
                        for (int j = 0; j < W; j++) D.pcol[(size_t)c][(size_t)j * m + i] = base + i;      // padding: the row's own local index, value +0.0
                        int j = 0;
                        for (int p = G.ptr[(size_t)r]; p < G.ptr[(size_t)r + 1]; p++, j++) {   // ascending column of the internal numbering: the order of the sums
                            if (G.col[(size_t)p] == r) { D.pdiag[(size_t)c][(size_t)i] = G.val[(size_t)p]; D.pdentry[(size_t)c][(size_t)i] = p; continue; }      // (its slot stays padding)
                            D.pcol[(size_t)c][(size_t)j * m + i] = loc[(size_t)G.col[(size_t)p]];
                            D.pval[(size_t)c][(size_t)j * m + i] = G.val[(size_t)p];
                            D.pentry[(size_t)c][(size_t)j * m + i] = p;
                        }
                    }
                    base += D.n_loc[(size_t)c];
                }
                for (int p = 1; p <= P; p++) D.updates += D.cnt[(size_t)((p - 1) % nc)][(size_t)(P - p)];
            }
            for (int r : touched) { dist[(size_t)r] = -1; loc[(size_t)r] = -1; }
        }
    });
    for (const TileData& D : tiles) if (!D.ok) return R;
    R.n_tiles = T; R.nc = nc; R.sweeps = sweeps; R.P = P;
    R.hdr.assign((size_t)T * TILED_HDR, 0);
    for (int t = 0; t < T; t++) {
        const TileData& D = tiles[(size_t)t];
        int* H = R.hdr.data() + (size_t)t * TILED_HDR;
        H[0] = (int)R.ext_rows.size(); H[1] = (int)D.ext.size(); H[2] = D.W; H[3] = 0;
        R.max_ext = std::max(R.max_ext, (int)D.ext.size());
        R.updates += D.updates;
        R.ext_rows.insert(R.ext_rows.end(), D.ext.begin(), D.ext.end());
        int lbase = 0;
        for (int c = 0; c < nc; c++) {
            int* C = H + 4 + c * TILED_CSTRIDE;
            C[0] = (int)R.pcol.size(); C[1] = D.m[(size_t)c]; C[2] = (int)R.prow.size(); C[3] = lbase;
            for (int d = 0; d <= TILED_PMAX; d++) C[4 + d] = D.cnt[(size_t)c][(size_t)d];
            R.pcol.insert(R.pcol.end(), D.pcol[(size_t)c].begin(), D.pcol[(size_t)c].end());
            R.pval.insert(R.pval.end(), D.pval[(size_t)c].begin(), D.pval[(size_t)c].end());
            R.pentry.insert(R.pentry.end(), D.pentry[(size_t)c].begin(), D.pentry[(size_t)c].end());
            R.prow.insert(R.prow.end(), D.prow[(size_t)c].begin(), D.prow[(size_t)c].end());
            R.pdiag.insert(R.pdiag.end(), D.pdiag[(size_t)c].begin(), D.pdiag[(size_t)c].end());
            R.pdentry.insert(R.pdentry.end(), D.pdentry[(size_t)c].begin(), D.pdentry[(size_t)c].end());
            lbase += D.n_loc[(size_t)c];
        }
    }
    return R;
}

}  // namespace smg
