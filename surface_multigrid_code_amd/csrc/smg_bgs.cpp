// smg_bgs.cpp -- host side of the block Gauss-Seidel sweep for many right-hand sides (smg_bgs.hpp): blocks, block colours,
// the order inside a block, the per-row entry batches in ascending column of the bgs order.
#include "smg_bgs.hpp"

#include <algorithm>
#include <numeric>
#include <queue>

namespace smg {

BgsPlan build_bgs(const Csr& G, const std::vector<int>& vcp, int block_rows)
{
    BgsPlan R;
    const int n = G.nr;
    if (n == 0 || block_rows < 2 || block_rows > BGS_ROWS || vcp.size() < 2 || vcp.back() != n) return R;
    int nb = 0;
    std::vector<int> part = partition_tiles(G, block_rows, &nb);
    // A block whose rim (the distinct rows of other blocks it reads) exceeds BGS_RIM_GOAL is cut in two (first / second half of a
    // breadth-first order of its rows): the LDS image of EVERY block of the level is sized by the largest rim, and the few elongated or
    // irregular blocks with a rim of 100+ rows would cost all the others a third of their occupancy.
    for (int pass = 0; pass < 4; pass++) {
        std::vector<std::vector<int>> mem((size_t)nb);
        for (int i = 0; i < n; i++) mem[(size_t)part[(size_t)i]].push_back(i);
        std::vector<int> fat;
        for (int b = 0; b < nb; b++) {
            std::vector<int> rim;
            for (int i : mem[(size_t)b])
                for (int p = G.ptr[(size_t)i]; p < G.ptr[(size_t)i + 1]; p++) if (part[(size_t)G.col[(size_t)p]] != b) rim.push_back(G.col[(size_t)p]);
            std::sort(rim.begin(), rim.end());
            if ((int)(std::unique(rim.begin(), rim.end()) - rim.begin()) > BGS_RIM_GOAL && mem[(size_t)b].size() >= 8) fat.push_back(b);
        }
        if (fat.empty()) break;
        for (int b : fat) {
            const std::vector<int>& M = mem[(size_t)b];
            // breadth-first order inside the block from its first row (other components appended)
            std::vector<int> order;
            std::vector<char> seen(M.size(), 0);
            auto loc = [&](int row) { return (int)(std::lower_bound(M.begin(), M.end(), row) - M.begin()); };
            for (size_t s0 = 0; s0 < M.size(); s0++) {
                if (seen[s0]) continue;
                seen[s0] = 1; order.push_back((int)s0);
                for (size_t head = order.size() - 1; head < order.size(); head++) {
                    const int v = M[(size_t)order[head]];
                    for (int p = G.ptr[(size_t)v]; p < G.ptr[(size_t)v + 1]; p++) {
                        const int w = G.col[(size_t)p];
                        if (part[(size_t)w] != b) continue;
                        const int lw = loc(w);
                        if (!seen[(size_t)lw]) { seen[(size_t)lw] = 1; order.push_back(lw); }
                    }
                }
            }
            for (size_t t = order.size() / 2; t < order.size(); t++) part[(size_t)M[(size_t)order[t]]] = nb;
            nb++;
        }
    }
    // members of every block (ascending row)
    std::vector<int> mptr((size_t)nb + 1, 0), members((size_t)n);
    for (int i = 0; i < n; i++) mptr[(size_t)part[(size_t)i] + 1]++;
    for (int b = 0; b < nb; b++) mptr[(size_t)b + 1] += mptr[(size_t)b];
    {
        std::vector<int> fill(mptr.begin(), mptr.end() - 1);
        for (int i = 0; i < n; i++) members[(size_t)fill[(size_t)part[(size_t)i]]++] = i;
    }
    // block adjacency
    std::vector<std::vector<int>> adj((size_t)nb);
    parallel_for(nb, 64, [&](long b0, long b1) {
        for (long b = b0; b < b1; b++) {
            std::vector<int>& a = adj[(size_t)b];
            for (int m = mptr[(size_t)b]; m < mptr[(size_t)b + 1]; m++) {
                const int i = members[(size_t)m];
                for (int p = G.ptr[(size_t)i]; p < G.ptr[(size_t)i + 1]; p++) {
                    const int ob = part[(size_t)G.col[(size_t)p]];
                    if (ob != (int)b) a.push_back(ob);
                }
            }
            std::sort(a.begin(), a.end());
            a.erase(std::unique(a.begin(), a.end()), a.end());
        }
    });
    // Colouring of the block graph (blocks of one colour share no entry of G): DSATUR -- always the block that sees the most colours, ties
    // by degree, then index -- followed by attempts to empty the smallest class (a block moves to another colour none of its neighbours
    // has).  Every colour is a launch with its own ramp and tail, and first-fit left a fifth and a sixth class of a few blocks each.
    std::vector<int> colour((size_t)nb, -1);
    int ncol = 0;
    {
        std::vector<unsigned> seen((size_t)nb, 0u);       // bit c: a neighbour has colour c (c < 32)
        std::vector<int> nsat((size_t)nb, 0);
        std::vector<char> done((size_t)nb, 0);
        // buckets by saturation would be O(n); with ~16 k blocks a scan per step of a priority queue with lazy deletion is plenty
        struct E { int sat, deg, b; bool operator<(const E& o) const { return sat != o.sat ? sat < o.sat : (deg != o.deg ? deg < o.deg : b > o.b); } };
        std::priority_queue<E> pq;
        for (int b = 0; b < nb; b++) pq.push({0, (int)adj[(size_t)b].size(), b});
        while (!pq.empty()) {
            const E e = pq.top();
            pq.pop();
            if (done[(size_t)e.b] || e.sat != nsat[(size_t)e.b]) continue;
            int c = 0;
            while (c < 31 && (seen[(size_t)e.b] >> c & 1u)) c++;
            colour[(size_t)e.b] = c;
            done[(size_t)e.b] = 1;
            ncol = std::max(ncol, c + 1);
            for (int o : adj[(size_t)e.b])
                if (!done[(size_t)o] && !(seen[(size_t)o] >> c & 1u)) { seen[(size_t)o] |= 1u << c; nsat[(size_t)o]++; pq.push({nsat[(size_t)o], (int)adj[(size_t)o].size(), o}); }
        }
        if (ncol >= 31) return BgsPlan();
        // empty the smallest class while that works
        for (int guard = 0; guard < 8 && ncol > 3; guard++) {
            std::vector<int> cnt((size_t)ncol, 0);
            for (int b = 0; b < nb; b++) cnt[(size_t)colour[(size_t)b]]++;
            const int small = (int)(std::min_element(cnt.begin(), cnt.end()) - cnt.begin());
            bool all_moved = true;
            for (int b = 0; b < nb; b++) {
                if (colour[(size_t)b] != small) continue;
                unsigned used = 0u;
                for (int o : adj[(size_t)b]) used |= 1u << colour[(size_t)o];
                int c = -1;
                for (int t = 0; t < ncol; t++) if (t != small && !(used >> t & 1u)) { c = t; break; }
                if (c < 0) {     // one exchange deep: a neighbour of the only blocking colour may itself move elsewhere
                    for (int t = 0; t < ncol && c < 0; t++) {
                        if (t == small) continue;
                        int blocker = -1, nblock = 0;
                        for (int o : adj[(size_t)b]) if (colour[(size_t)o] == t) { blocker = o; nblock++; }
                        if (nblock != 1) continue;
                        unsigned u2 = 1u << small;
                        for (int o : adj[(size_t)blocker]) u2 |= 1u << colour[(size_t)o];
                        for (int t2 = 0; t2 < ncol; t2++) if (t2 != t && !(u2 >> t2 & 1u)) { colour[(size_t)blocker] = t2; c = t; break; }
                    }
                }
                if (c >= 0) colour[(size_t)b] = c; else all_moved = false;
            }
            if (!all_moved) break;
            for (int b = 0; b < nb; b++) if (colour[(size_t)b] > small) colour[(size_t)b]--;
            ncol--;
        }
    }
    // blocks in the order (colour, bisection id): neighbours in space stay neighbours in the launch (shared rims meet in one L2)
    std::vector<int> blocks((size_t)nb);
    std::iota(blocks.begin(), blocks.end(), 0);
    std::stable_sort(blocks.begin(), blocks.end(), [&](int a, int b) { return colour[(size_t)a] < colour[(size_t)b]; });
    R.color_ptr.assign((size_t)ncol + 1, 0);
    for (int b = 0; b < nb; b++) R.color_ptr[(size_t)colour[(size_t)b] + 1]++;
    for (int c = 0; c < ncol; c++) R.color_ptr[(size_t)c + 1] += R.color_ptr[(size_t)c];
    R.blk_ptr.assign((size_t)nb + 1, 0);
    for (int q = 0; q < nb; q++) R.blk_ptr[(size_t)q + 1] = R.blk_ptr[(size_t)q] + (mptr[(size_t)blocks[(size_t)q] + 1] - mptr[(size_t)blocks[(size_t)q]]);
    // order inside a block: vertex colour by vertex colour, ascending row inside a colour (the numbering is colour-major: ascending row
    // IS colour-major) -- `members` is ascending already
    R.rows.assign((size_t)n, 0);
    std::vector<int> pos((size_t)n, 0);          // row -> position in the bgs order
    for (int q = 0; q < nb; q++) {
        const int b = blocks[(size_t)q], m0 = mptr[(size_t)b], cnt = mptr[(size_t)b + 1] - m0, base = R.blk_ptr[(size_t)q];
        for (int t = 0; t < cnt; t++) { R.rows[(size_t)base + t] = members[(size_t)m0 + t]; pos[(size_t)members[(size_t)m0 + t]] = base + t; }
    }
    // per block: its rim (rows of other blocks it reads, ascending), units, batches per row.
    // Units: <= 16 rows that are updated at once from the image as it stands and written back together.  What that needs: no two rows of a unit
    // share an entry, and for every entry (i, j) inside the block with i before j in the bgs order, i sits in an EARLIER unit than j (j must see
    // i's new value, i must see j's old one).  Rows are placed in bgs order into the first unit after all their earlier neighbours that still
    // has room: the four vertex colours give a critical path of four units, and rows of a later colour whose earlier neighbours are all done
    // fill the slots a colour class of 17 - 20 rows would otherwise waste in a second, nearly empty unit (64 -> ~5 units per block).
    struct Blk { int nb = 1; std::vector<int> rim; std::vector<std::vector<int>> units; };      // units: positions relative to the block
    std::vector<Blk> info((size_t)nb);
    std::vector<char> bad((size_t)nb, 0);
    parallel_for(nb, 32, [&](long q0, long q1) {
        for (long q = q0; q < q1; q++) {
            Blk& I = info[(size_t)q];
            const int base = R.blk_ptr[(size_t)q], end = R.blk_ptr[(size_t)q + 1], m = end - base;
            if (m > BGS_ROWS) { bad[(size_t)q] = 1; continue; }
            int wmax = 1;
            int unit_of[BGS_ROWS];
            for (int t = 0; t < m; t++) {
                const int i = R.rows[(size_t)base + t];
                int w = 0, lb = 0;
                bool diag = false;
                for (int p = G.ptr[(size_t)i]; p < G.ptr[(size_t)i + 1]; p++) {
                    const int j = G.col[(size_t)p];
                    if (j == i) { diag = true; continue; }
                    w++;
                    const int pj = pos[(size_t)j];
                    if (pj < base || pj >= end) I.rim.push_back(j);
                    else if (pj < base + t) lb = std::max(lb, unit_of[pj - base] + 1);
                }
                size_t un = (size_t)lb;
                while (un < I.units.size() && (int)I.units[un].size() >= BGS_UROWS) un++;
                if (un >= I.units.size()) I.units.resize(un + 1);
                I.units[un].push_back(t);
                unit_of[t] = (int)un;
                if (!diag) bad[(size_t)q] = 1;
                wmax = std::max(wmax, w);
            }
            std::sort(I.rim.begin(), I.rim.end());
            I.rim.erase(std::unique(I.rim.begin(), I.rim.end()), I.rim.end());
            I.nb = (wmax + BGS_BATCH - 1) / BGS_BATCH;
            if (I.nb > BGS_MAX_BATCHES || (int)I.rim.size() > BGS_RIM_MAX) bad[(size_t)q] = 1;
        }
    });
    for (char c : bad) if (c) return BgsPlan();
    int max_rim = 0;
    for (int q = 0; q < nb; q++) max_rim = std::max(max_rim, (int)info[(size_t)q].rim.size());
    const int XR = (BGS_ROWS + max_rim + 127) / 128 * 128;
    R.xrows = XR;
    R.hdr.assign((size_t)nb * BGS_HDR, 0);
    R.xrow.assign((size_t)nb * XR, 0);
    std::vector<long> unit0((size_t)nb + 1, 0), ent0((size_t)nb + 1, 0);
    for (int q = 0; q < nb; q++) {
        const Blk& I = info[(size_t)q];
        const long nu = (long)I.units.size();
        unit0[(size_t)q + 1] = unit0[(size_t)q] + nu;
        ent0[(size_t)q + 1] = ent0[(size_t)q] + nu * BGS_UROWS * BGS_BATCH * I.nb;
        R.hdr[(size_t)q * BGS_HDR + 0] = (int)unit0[(size_t)q]; R.hdr[(size_t)q * BGS_HDR + 1] = (int)nu;
        R.hdr[(size_t)q * BGS_HDR + 2] = I.nb; R.hdr[(size_t)q * BGS_HDR + 3] = (int)ent0[(size_t)q];
        R.hdr[(size_t)q * BGS_HDR + 4] = BGS_ROWS + ((int)I.rim.size() + 63) / 64 * 64;
    }
    if (ent0[(size_t)nb] > 0x7fffffffl) return BgsPlan();
    const size_t NU = (size_t)unit0[(size_t)nb];
    R.ugrow.assign(NU * BGS_UROWS, 0); R.ulrow.assign(NU * BGS_UROWS, 0); R.udiag.assign(NU * BGS_UROWS, 1.0); R.dentry.assign(NU * BGS_UROWS, -1);
    R.eidx.assign((size_t)ent0[(size_t)nb], 0);
    R.eval.assign((size_t)ent0[(size_t)nb], 0.0);
    R.eentry.assign((size_t)ent0[(size_t)nb], -1);
    parallel_for(nb, 32, [&](long q0, long q1) {
        std::vector<std::pair<int, int>> ent;     // (position of the column, entry of G)
        for (long q = q0; q < q1; q++) {
            const Blk& I = info[(size_t)q];
            const int base = R.blk_ptr[(size_t)q], end = R.blk_ptr[(size_t)q + 1], m = end - base;
            int* X = &R.xrow[(size_t)q * XR];
            for (int l = 0; l < XR; l++) X[l] = R.rows[(size_t)base];
            for (int l = 0; l < m; l++) X[l] = R.rows[(size_t)base + l];
            for (size_t z = 0; z < I.rim.size(); z++) X[BGS_ROWS + z] = I.rim[z];
            const size_t S = (size_t)I.nb * BGS_BATCH;
            for (size_t un = 0; un < I.units.size(); un++) {
                const size_t u = (size_t)unit0[(size_t)q] + un;
                for (int r = 0; r < BGS_UROWS; r++) {
                    const int loc = I.units[un][r < (int)I.units[un].size() ? (size_t)r : 0];        // a slot without a row of its own: the unit's first row again
                    const int i = R.rows[(size_t)base + loc];
                    R.ugrow[u * BGS_UROWS + r] = i;
                    R.ulrow[u * BGS_UROWS + r] = loc;
                    ent.clear();
                    for (int pp = G.ptr[(size_t)i]; pp < G.ptr[(size_t)i + 1]; pp++) {
                        if (G.col[(size_t)pp] == i) { R.udiag[u * BGS_UROWS + r] = G.val[(size_t)pp]; R.dentry[u * BGS_UROWS + r] = pp; }
                        else ent.emplace_back(pos[(size_t)G.col[(size_t)pp]], pp);
                    }
                    std::sort(ent.begin(), ent.end());
                    size_t sl = (size_t)ent0[(size_t)q] + (un * BGS_UROWS + r) * S;
                    for (const auto& e : ent) {
                        const int pj = e.first, j = G.col[(size_t)e.second];
                        int l;
                        if (pj >= base && pj < end) l = pj - base;
                        else l = BGS_ROWS + (int)(std::lower_bound(I.rim.begin(), I.rim.end(), j) - I.rim.begin());
                        R.eidx[sl] = l; R.eval[sl] = G.val[(size_t)e.second]; R.eentry[sl] = e.second;
                        sl++;
                    }
                }
            }
        }
    });
    long rim = 0;
    for (int q = 0; q < nb; q++) rim += (long)info[(size_t)q].rim.size();
    R.n = n; R.n_blocks = nb; R.n_colors = ncol;
    R.rim = (double)rim / n;
    R.fill = (double)n / ((double)NU * BGS_UROWS);
    return R;
}

}  // namespace smg
