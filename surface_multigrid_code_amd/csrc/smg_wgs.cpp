// smg_wgs.cpp -- host side of the wave Gauss-Seidel sweep (smg_wgs.hpp): pieces, piece colours, the order and the phases inside a piece,
// the per-lane entry slots in ascending column of the wgs order.
#include "smg_wgs.hpp"
#include "smg_bgs.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <queue>

namespace smg {

namespace {

// DSATUR colouring of a small or large graph given as adjacency lists, followed by attempts to empty the smallest class (every colour of the
// piece graph is a launch; every colour inside a piece a phase).  Returns the number of colours (< 0: more than 30).
int colour_graph(const std::vector<std::vector<int>>& adj, std::vector<int>& colour)
{
    const int nb = (int)adj.size();
    colour.assign((size_t)nb, -1);
    int ncol = 0;
    std::vector<unsigned> seen((size_t)nb, 0u);
    std::vector<int> nsat((size_t)nb, 0);
    std::vector<char> done((size_t)nb, 0);
    struct E { int sat, deg, b; bool operator<(const E& o) const { return sat != o.sat ? sat < o.sat : (deg != o.deg ? deg < o.deg : b > o.b); } };
    std::priority_queue<E> pq;
    for (int b = 0; b < nb; b++) pq.push({0, (int)adj[(size_t)b].size(), b});
    while (!pq.empty()) {
        const E e = pq.top();
        pq.pop();
        if (done[(size_t)e.b] || e.sat != nsat[(size_t)e.b]) continue;
        int c = 0;
        while (c < 31 && (seen[(size_t)e.b] >> c & 1u)) c++;
        if (c >= 31) return -1;
        colour[(size_t)e.b] = c;
        done[(size_t)e.b] = 1;
        ncol = std::max(ncol, c + 1);
        for (int o : adj[(size_t)e.b])
            if (!done[(size_t)o] && !(seen[(size_t)o] >> c & 1u)) { seen[(size_t)o] |= 1u << c; nsat[(size_t)o]++; pq.push({nsat[(size_t)o], (int)adj[(size_t)o].size(), o}); }
    }
    for (int guard = 0; guard < 8 && ncol > 2; guard++) {
        std::vector<int> cnt((size_t)ncol, 0);
        for (int b = 0; b < nb; b++) cnt[(size_t)colour[(size_t)b]]++;
        const int small = (int)(std::min_element(cnt.begin(), cnt.end()) - cnt.begin());
        std::vector<int> saved = colour;
        bool all_moved = true;
        for (int b = 0; b < nb && all_moved; b++) {
            if (colour[(size_t)b] != small) continue;
            unsigned used = 0u;
            for (int o : adj[(size_t)b]) used |= 1u << colour[(size_t)o];
            int c = -1;
            for (int t = 0; t < ncol; t++) if (t != small && !(used >> t & 1u)) { c = t; break; }
            if (c < 0) {     // one exchange deep: the only neighbour of a colour may itself move elsewhere
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
        if (!all_moved) { colour.swap(saved); break; }
        for (int b = 0; b < nb; b++) if (colour[(size_t)b] > small) colour[(size_t)b]--;
        ncol--;
    }
    return ncol;
}

}  // namespace

// Pieces along the breadth-first level sets of G: a level set is a closed band of the surface one graph hop wide, its rows read only their own
// and the two neighbouring level sets.  Every connected part of a level set is walked from one of its ends and cut into runs of <= piece_rows rows.
std::vector<int> partition_bands(const Csr& G, int piece_rows, int* n_pieces, std::vector<int>* hint)
{
    const int n = G.nr;
    std::vector<int> dist((size_t)n, -1), order;
    order.reserve((size_t)n);
    // breadth-first numbering of every component from a pseudo-peripheral vertex
    std::vector<int> queue;
    auto bfs = [&](int root, std::vector<int>& d, int stamp_base) {
        queue.clear();
        queue.push_back(root);
        d[(size_t)root] = stamp_base;
        for (size_t head = 0; head < queue.size(); head++) {
            const int v = queue[head];
            for (int p = G.ptr[(size_t)v]; p < G.ptr[(size_t)v + 1]; p++) {
                const int q = G.col[(size_t)p];
                if (d[(size_t)q] < 0) { d[(size_t)q] = d[(size_t)v] + 1; queue.push_back(q); }
            }
        }
        return queue.back();
    };
    int base = 0;
    std::vector<int> tmp((size_t)n, -1);
    for (int s = 0; s < n; s++) {
        if (dist[(size_t)s] >= 0) continue;
        const int far1 = bfs(s, tmp, 0);
        for (int v : queue) tmp[(size_t)v] = -1;
        const int far2 = bfs(far1, tmp, 0);
        for (int v : queue) tmp[(size_t)v] = -1;
        bfs(far2, dist, base);
        int dmax = base;
        for (int v : queue) dmax = std::max(dmax, dist[(size_t)v]);
        base = dmax + 2;          // the next component's level sets share no index with this one's
    }
    // level sets -> connected parts -> runs
    std::vector<int> by_level((size_t)n);
    std::iota(by_level.begin(), by_level.end(), 0);
    std::stable_sort(by_level.begin(), by_level.end(), [&](int a, int b) { return dist[(size_t)a] < dist[(size_t)b]; });
    std::vector<int> part((size_t)n, -1);
    std::vector<char> seen((size_t)n, 0);
    int np = 0;
    std::vector<int> comp, walk;
    auto walk_from = [&](int root, int d, std::vector<int>& out, char mark) {      // breadth-first inside the level set d
        out.clear();
        out.push_back(root);
        seen[(size_t)root] = mark;
        for (size_t head = 0; head < out.size(); head++) {
            const int v = out[head];
            for (int p = G.ptr[(size_t)v]; p < G.ptr[(size_t)v + 1]; p++) {
                const int q = G.col[(size_t)p];
                if (dist[(size_t)q] == d && seen[(size_t)q] != mark && part[(size_t)q] < 0) { seen[(size_t)q] = mark; out.push_back(q); }
            }
        }
    };
    for (int i = 0; i < n; i++) {
        const int s = by_level[(size_t)i];
        if (part[(size_t)s] >= 0) continue;
        const int d = dist[(size_t)s];
        walk_from(s, d, comp, 1);
        const int end1 = comp.back();
        walk_from(end1, d, walk, 2);                 // from a far end: the walk runs along the band
        const int m = (int)walk.size();
        // an EVEN number of runs where there is more than one: the runs of a closed band then alternate all the way round, and with the parity of the
        // level set that is a 4-colouring of the pieces by construction (hint; build_wgs verifies it and repairs what a branching level set breaks)
        int runs = (m + piece_rows - 1) / piece_rows;
        if (runs > 1 && (runs & 1)) runs++;
        const int len = (m + runs - 1) / runs;
        for (int t = 0; t < m; t++) part[(size_t)walk[(size_t)t]] = np + t / len;
        const int made = (m + len - 1) / len;
        if (hint) for (int r = 0; r < made; r++) hint->push_back(2 * (d & 1) + (r & 1));
        np += made;
        for (int v : walk) seen[(size_t)v] = 0;
    }
    *n_pieces = np;
    return part;
}

WgsPlan build_wgs(const Csr& G, int piece_rows, int mode)
{
    WgsPlan R;
    const int n = G.nr;
    const bool tm_on = std::getenv("SMG_TIMING_WGS") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) { if (!tm_on) return; auto t = std::chrono::steady_clock::now(); std::fprintf(stderr, "[wgs plan] %-36s %7.1f ms\n", what, 1e3 * std::chrono::duration<double>(t - t_last).count()); t_last = t; };
    if (n == 0 || piece_rows < 2 || piece_rows > WGS_ROWS) return R;
    // every row needs its diagonal and fits the register image
    {
        bool ok = true;
        for (int i = 0; i < n && ok; i++) {
            bool diag = false;
            for (int p = G.ptr[(size_t)i]; p < G.ptr[(size_t)i + 1]; p++) if (G.col[(size_t)p] == i) diag = true;
            if (!diag || G.ptr[(size_t)i + 1] - G.ptr[(size_t)i] - 1 > WGS_MAX_BATCHES * WGS_BATCH) ok = false;
        }
        if (!ok) return R;
    }
    int np = 0;
    std::vector<int> hint;
    lap("row check");
    std::vector<int> part = mode == 1 ? partition_bands(G, piece_rows, &np, &hint) : partition_tiles(G, piece_rows, &np);
    lap("partition");
    // a piece whose rim exceeds the image is cut in two (first / second half of a breadth-first order of its rows)
    for (int pass = 0; pass < 6; pass++) {
        std::vector<std::vector<int>> mem((size_t)np);
        for (int i = 0; i < n; i++) mem[(size_t)part[(size_t)i]].push_back(i);
        std::vector<int> fat;
        std::vector<char> is_fat((size_t)np, 0);      // (the pieces side by side on the host threads: a quarter of the plan's time as one loop)
        parallel_for(np, 32, [&](long b0, long b1) {
            std::vector<int> rim;
            for (long b = b0; b < b1; b++) {
                rim.clear();
                for (int i : mem[(size_t)b])
                    for (int p = G.ptr[(size_t)i]; p < G.ptr[(size_t)i + 1]; p++) if (part[(size_t)G.col[(size_t)p]] != (int)b) rim.push_back(G.col[(size_t)p]);
                std::sort(rim.begin(), rim.end());
                if ((int)(std::unique(rim.begin(), rim.end()) - rim.begin()) > WGS_RIM_MAX) is_fat[(size_t)b] = mem[(size_t)b].size() < 2 ? 2 : 1;
            }
        });
        for (int b = 0; b < np; b++) { if (is_fat[(size_t)b] == 2) return R; if (is_fat[(size_t)b]) fat.push_back(b); }
        if (fat.empty()) break;
        if (pass == 5) return R;
        for (int b : fat) {
            const std::vector<int>& M = mem[(size_t)b];
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
            for (size_t t = order.size() / 2; t < order.size(); t++) part[(size_t)M[(size_t)order[t]]] = np;
            np++;
            if (!hint.empty()) hint.push_back(-1);
        }
    }
    lap("rim check / fat pieces");
    // members of every piece (ascending row)
    std::vector<int> mptr((size_t)np + 1, 0), members((size_t)n);
    for (int i = 0; i < n; i++) mptr[(size_t)part[(size_t)i] + 1]++;
    for (int b = 0; b < np; b++) mptr[(size_t)b + 1] += mptr[(size_t)b];
    {
        std::vector<int> fill(mptr.begin(), mptr.end() - 1);
        for (int i = 0; i < n; i++) members[(size_t)fill[(size_t)part[(size_t)i]]++] = i;
    }
    for (int b = 0; b < np; b++) if (mptr[(size_t)b + 1] - mptr[(size_t)b] > WGS_ROWS) return R;
    // piece adjacency, piece colours
    std::vector<std::vector<int>> adj((size_t)np);
    parallel_for(np, 64, [&](long b0, long b1) {
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
    lap("members, piece adjacency");
    std::vector<int> colour;
    int ncol = colour_graph(adj, colour);
    if (ncol < 1) return R;
    if ((int)hint.size() == np) {
        // the constructed colouring of band pieces (level-set parity, run parity): kept where it is valid; a piece in conflict (branching level sets,
        // pieces cut again for their rim) takes the first colour none of its neighbours has.  Used when it needs fewer colours than DSATUR found --
        // or as many, but without a nearly empty class (every class is a launch per sweep)
        std::vector<int> hc = hint;
        std::vector<int> redo;
        for (int b = 0; b < np; b++) {
            bool clash = hc[(size_t)b] < 0;
            for (int o : adj[(size_t)b]) if (o < b && hc[(size_t)o] == hc[(size_t)b]) clash = true;
            if (clash) { hc[(size_t)b] = -1; redo.push_back(b); }
        }
        int hcol = 4;
        bool ok = true;
        for (int b : redo) {
            unsigned used = 0u;
            for (int o : adj[(size_t)b]) if (hc[(size_t)o] >= 0) used |= 1u << hc[(size_t)o];
            int c = 0;
            while (c < 31 && (used >> c & 1u)) c++;
            if (c >= 31) { ok = false; break; }
            hc[(size_t)b] = c;
            hcol = std::max(hcol, c + 1);
        }
        if (ok) {
            auto smallest = [&](const std::vector<int>& col, int nc) { std::vector<int> cnt((size_t)nc, 0); for (int c : col) cnt[(size_t)c]++; return *std::min_element(cnt.begin(), cnt.end()); };
            if (hcol < ncol || (hcol == ncol && smallest(hc, hcol) > smallest(colour, ncol))) { colour.swap(hc); ncol = hcol; }
        }
    }
    lap("piece colours");
    // pieces in the order (colour, partition id): neighbours in space stay neighbours in the launch
    std::vector<int> pieces((size_t)np);
    std::iota(pieces.begin(), pieces.end(), 0);
    std::stable_sort(pieces.begin(), pieces.end(), [&](int a, int b) { return colour[(size_t)a] < colour[(size_t)b]; });
    R.color_ptr.assign((size_t)ncol + 1, 0);
    for (int b = 0; b < np; b++) R.color_ptr[(size_t)colour[(size_t)b] + 1]++;
    for (int c = 0; c < ncol; c++) R.color_ptr[(size_t)c + 1] += R.color_ptr[(size_t)c];
    R.piece_ptr.assign((size_t)np + 1, 0);
    for (int q = 0; q < np; q++) R.piece_ptr[(size_t)q + 1] = R.piece_ptr[(size_t)q] + (mptr[(size_t)pieces[(size_t)q] + 1] - mptr[(size_t)pieces[(size_t)q]]);
    // order inside a piece: local colour by local colour (DSATUR on the piece's own graph), ascending row inside a colour
    R.rows.assign((size_t)n, 0);
    std::vector<int> pos((size_t)n, 0);
    std::vector<char> bad((size_t)np, 0);
    parallel_for(np, 16, [&](long q0, long q1) {
        std::vector<std::vector<int>> ladj;
        std::vector<int> lcol, idx;
        for (long q = q0; q < q1; q++) {
            const int b = pieces[(size_t)q], m0 = mptr[(size_t)b], m = mptr[(size_t)b + 1] - m0, base = R.piece_ptr[(size_t)q];
            ladj.assign((size_t)m, std::vector<int>());
            for (int t = 0; t < m; t++) {
                const int i = members[(size_t)m0 + t];
                for (int p = G.ptr[(size_t)i]; p < G.ptr[(size_t)i + 1]; p++) {
                    const int j = G.col[(size_t)p];
                    if (j == i || part[(size_t)j] != b) continue;
                    ladj[(size_t)t].push_back((int)(std::lower_bound(members.begin() + m0, members.begin() + m0 + m, j) - (members.begin() + m0)));
                }
            }
            if (colour_graph(ladj, lcol) < 1) { bad[(size_t)q] = 1; continue; }
            idx.resize((size_t)m);
            std::iota(idx.begin(), idx.end(), 0);
            std::stable_sort(idx.begin(), idx.end(), [&](int a, int c) { return lcol[(size_t)a] < lcol[(size_t)c]; });
            for (int t = 0; t < m; t++) R.rows[(size_t)base + t] = members[(size_t)m0 + idx[(size_t)t]];
        }
    });
    for (char c : bad) if (c) return WgsPlan();
    lap("local colours");
    for (int t = 0; t < n; t++) pos[(size_t)R.rows[(size_t)t]] = t;
    // per piece: rim, phases (level scheduling in the wgs order), batches per row
    struct Pc { int nb = 1, nph = 0; std::vector<int> rim; int ph[WGS_ROWS]; };
    std::vector<Pc> info((size_t)np);
    parallel_for(np, 32, [&](long q0, long q1) {
        for (long q = q0; q < q1; q++) {
            Pc& I = info[(size_t)q];
            const int base = R.piece_ptr[(size_t)q], end = R.piece_ptr[(size_t)q + 1], m = end - base;
            int wmax = 1;
            for (int t = 0; t < m; t++) {
                const int i = R.rows[(size_t)base + t];
                int w = 0, lb = 0;
                for (int p = G.ptr[(size_t)i]; p < G.ptr[(size_t)i + 1]; p++) {
                    const int j = G.col[(size_t)p];
                    if (j == i) continue;
                    w++;
                    const int pj = pos[(size_t)j];
                    if (pj < base || pj >= end) I.rim.push_back(j);
                    else if (pj < base + t) lb = std::max(lb, I.ph[pj - base] + 1);
                }
                I.ph[t] = lb;
                I.nph = std::max(I.nph, lb + 1);
                wmax = std::max(wmax, w);
            }
            std::sort(I.rim.begin(), I.rim.end());
            I.rim.erase(std::unique(I.rim.begin(), I.rim.end()), I.rim.end());
            I.nb = (wmax + WGS_BATCH - 1) / WGS_BATCH;
            if (I.nb > WGS_MAX_BATCHES || (int)I.rim.size() > WGS_RIM_MAX) bad[(size_t)q] = 1;
        }
    });
    for (char c : bad) if (c) return WgsPlan();
    lap("rims, phases");
    int max_rim = 1;
    for (int q = 0; q < np; q++) max_rim = std::max(max_rim, (int)info[(size_t)q].rim.size());
    const int RP = wgs_rim_pitch(max_rim);
    R.rim_pitch = RP;
    R.hdr.assign((size_t)np * WGS_HDR, 0);
    std::vector<long> ent0((size_t)np + 1, 0);
    for (int q = 0; q < np; q++) {
        const Pc& I = info[(size_t)q];
        ent0[(size_t)q + 1] = ent0[(size_t)q] + (long)WGS_ROWS * WGS_BATCH * I.nb;
        int* H = &R.hdr[(size_t)q * WGS_HDR];
        H[0] = (int)ent0[(size_t)q]; H[1] = I.nb; H[2] = (int)I.rim.size(); H[3] = I.nph; H[4] = q * RP; H[5] = R.piece_ptr[(size_t)q + 1] - R.piece_ptr[(size_t)q];
    }
    if (ent0[(size_t)np] > 0x7fffffffl) return WgsPlan();
    const size_t NL = (size_t)np * WGS_ROWS;
    R.grow.assign(NL, -1); R.meta.assign(NL, 0xffff); R.diag.assign(NL, 1.0); R.dentry.assign(NL, -1);
    R.rim.assign((size_t)np * RP, 0);
    R.eoff.assign((size_t)ent0[(size_t)np] / 2, 0u);
    R.eval.assign((size_t)ent0[(size_t)np], 0.0);
    R.eentry.assign((size_t)ent0[(size_t)np], -1);
    parallel_for(np, 32, [&](long q0, long q1) {
        std::vector<std::pair<int, int>> ent;     // (position of the column, entry of G)
        for (long q = q0; q < q1; q++) {
            const Pc& I = info[(size_t)q];
            const int base = R.piece_ptr[(size_t)q], end = R.piece_ptr[(size_t)q + 1], m = end - base;
            for (int z = 0; z < RP; z++) R.rim[(size_t)q * RP + z] = z < (int)I.rim.size() ? I.rim[(size_t)z] : R.rows[(size_t)base];
            const int S = I.nb * WGS_BATCH;
            const size_t e0 = (size_t)ent0[(size_t)q];
            for (int t = 0; t < WGS_ROWS; t++) {
                const size_t w = (size_t)q * WGS_ROWS + t;
                // a lane without a row: every slot points at local row 0 with a zero (never active, nothing stored)
                unsigned offs[WGS_MAX_BATCHES * WGS_BATCH];
                for (int s = 0; s < S; s++) offs[s] = 8u * (unsigned)(t < m ? t : 0);
                if (t < m) {
                    const int i = R.rows[(size_t)base + t];
                    R.grow[w] = i;
                    ent.clear();
                    for (int pp = G.ptr[(size_t)i]; pp < G.ptr[(size_t)i + 1]; pp++) {
                        if (G.col[(size_t)pp] == i) { R.diag[w] = G.val[(size_t)pp]; R.dentry[w] = pp; }
                        else ent.emplace_back(pos[(size_t)G.col[(size_t)pp]], pp);
                    }
                    std::sort(ent.begin(), ent.end());
                    R.meta[w] = I.ph[t] | (std::max(1, ((int)ent.size() + WGS_BATCH - 1) / WGS_BATCH) << 16);
                    int s = 0;
                    for (const auto& e : ent) {
                        const int pj = e.first, j = G.col[(size_t)e.second];
                        const int l = (pj >= base && pj < end) ? pj - base : WGS_ROWS + (int)(std::lower_bound(I.rim.begin(), I.rim.end(), j) - I.rim.begin());
                        offs[s] = 8u * (unsigned)l;
                        R.eval[e0 + (size_t)s * WGS_ROWS + t] = G.val[(size_t)e.second];
                        R.eentry[e0 + (size_t)s * WGS_ROWS + t] = e.second;
                        s++;
                    }
                }
                for (int s = 0; s < S; s += 2) R.eoff[e0 / 2 + (size_t)(s / 2) * WGS_ROWS + t] = offs[s] | (offs[s + 1] << 16);
            }
        }
    });
    lap("slots");
    long rim = 0, phs = 0;
    for (int q = 0; q < np; q++) { R.nb_max = std::max(R.nb_max, info[(size_t)q].nb); rim += (long)info[(size_t)q].rim.size(); phs += info[(size_t)q].nph; R.phases_max = std::max(R.phases_max, info[(size_t)q].nph); }
    R.n = n; R.n_pieces = np; R.n_colors = ncol;
    R.rim_ratio = (double)rim / n;
    R.phases_mean = (double)phs / np;
    return R;
}

void wgs_sweep_host(const WgsPlan& P, const double* b, double* u)
{
    std::vector<double> xs((size_t)WGS_ROWS + P.rim_pitch);
    for (int c = 0; c < P.n_colors; c++)
        for (int q = P.color_ptr[(size_t)c]; q < P.color_ptr[(size_t)c + 1]; q++) {
            const int* H = &P.hdr[(size_t)q * WGS_HDR];
            const size_t e0 = (size_t)H[0];
            const int S = H[1] * WGS_BATCH, nrim = H[2], nph = H[3];
            for (int t = 0; t < WGS_ROWS; t++) { const int g = P.grow[(size_t)q * WGS_ROWS + t]; xs[(size_t)t] = g >= 0 ? u[(size_t)g] : 0.0; }
            for (int z = 0; z < nrim; z++) xs[(size_t)WGS_ROWS + z] = u[(size_t)P.rim[(size_t)H[4] + z]];
            for (int ph = 0; ph < nph; ph++) {
                double out[WGS_ROWS];
                for (int t = 0; t < WGS_ROWS; t++) {
                    const size_t w = (size_t)q * WGS_ROWS + t;
                    if ((P.meta[w] & 0xffff) != ph) continue;
                    double acc = 0.0;
                    const int Sl = (P.meta[w] >> 16) * WGS_BATCH;     // the lane's own batches (the rest of the piece's pitch is padding the kernel never requests)
                    if (Sl > S) { u[0] = 0.0 / 0.0; return; }
                    for (int s = 0; s < Sl; s++) {
                        const unsigned word = P.eoff[e0 / 2 + (size_t)(s / 2) * WGS_ROWS + t];
                        const unsigned off = (s & 1) ? (word >> 16) : (word & 0xffffu);
                        acc += P.eval[e0 + (size_t)s * WGS_ROWS + t] * xs[(size_t)(off / 8u)];
                    }
                    out[t] = (b[(size_t)P.grow[w]] - acc) / P.diag[w];
                }
                for (int t = 0; t < WGS_ROWS; t++) {      // the rows of a phase read none of each other: all of them see the image as it stood
                    const size_t w = (size_t)q * WGS_ROWS + t;
                    if ((P.meta[w] & 0xffff) != ph) continue;
                    xs[(size_t)t] = out[t];
                    u[(size_t)P.grow[w]] = out[t];
                }
            }
        }
}

}  // namespace smg
