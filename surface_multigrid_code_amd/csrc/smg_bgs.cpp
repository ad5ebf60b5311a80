// smg_bgs.cpp -- host side of the block-sequential Gauss-Seidel sweep for many right-hand sides (smg_bgs.hpp): blocks, block colours,
// the order inside a block, the per-row entry batches in ascending column of the bgs order.
#include "smg_bgs.hpp"

#include <algorithm>
#include <numeric>

namespace smg {

BgsPlan build_bgs(const Csr& G, const std::vector<int>& vcp, int block_rows)
{
    BgsPlan R;
    const int n = G.nr;
    if (n == 0 || block_rows < 2 || block_rows > BGS_ROWS || vcp.size() < 2 || vcp.back() != n) return R;
    int nb = 0;
    const std::vector<int> part = partition_tiles(G, block_rows, &nb);
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
    // greedy colouring, blocks of many neighbours first (ties: bisection order); blocks of one colour share no entry of G
    std::vector<int> order((size_t)nb), colour((size_t)nb, -1);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return adj[(size_t)a].size() > adj[(size_t)b].size(); });
    int ncol = 0;
    {
        std::vector<int> mark;
        for (int b : order) {
            mark.assign((size_t)ncol + 1, 0);
            for (int o : adj[(size_t)b]) if (colour[(size_t)o] >= 0) mark[(size_t)colour[(size_t)o]] = 1;
            int c = 0;
            while (mark[(size_t)c]) c++;
            colour[(size_t)b] = c;
            ncol = std::max(ncol, c + 1);
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
    auto vcolour = [&](int row) { return (int)(std::upper_bound(vcp.begin(), vcp.end(), row) - vcp.begin()) - 1; };
    // per block: phases = the vertex colours present, rows of a phase dealt round-robin to the waves; batches per row
    struct Blk { int nph = 0, nb = 1; std::vector<int> ph_ptr; };      // ph_ptr: positions (relative to the block) where the phases start
    std::vector<Blk> info((size_t)nb);
    int lp = 1;
    bool bad = false;
    for (int q = 0; q < nb; q++) {
        Blk& I = info[(size_t)q];
        const int base = R.blk_ptr[(size_t)q], m = R.blk_ptr[(size_t)q + 1] - base;
        if (m > BGS_ROWS) { bad = true; break; }
        int wmax = 1, last = -1;
        for (int t = 0; t < m; t++) {
            const int i = R.rows[(size_t)base + t], c = vcolour(i);
            if (c != last) { I.ph_ptr.push_back(t); last = c; }
            wmax = std::max(wmax, G.ptr[(size_t)i + 1] - G.ptr[(size_t)i]);
        }
        I.ph_ptr.push_back(m);
        I.nph = (int)I.ph_ptr.size() - 1;
        I.nb = (wmax + BGS_BATCH - 1) / BGS_BATCH;
        if (I.nb > BGS_MAX_BATCHES) { bad = true; break; }
        for (int p = 0; p < I.nph; p++) lp = std::max(lp, (I.ph_ptr[(size_t)p + 1] - I.ph_ptr[(size_t)p] + BGS_WAVES - 1) / BGS_WAVES);
    }
    if (bad || lp > BGS_LP_MAX) return BgsPlan();
    lp = std::max(lp, 4);
    if (lp == 7) lp = 8;      // (the kernel is instantiated for 4, 5, 6 and 8 row slots)
    R.lp = lp;
    R.hdr.assign((size_t)nb * BGS_HDR, 0);
    R.brow.assign((size_t)nb * BGS_ROWS, 0);
    std::vector<long> unit0((size_t)nb + 1, 0), ent0((size_t)nb + 1, 0);
    for (int q = 0; q < nb; q++) {
        const Blk& I = info[(size_t)q];
        unit0[(size_t)q + 1] = unit0[(size_t)q] + (long)I.nph * BGS_WAVES;
        ent0[(size_t)q + 1] = ent0[(size_t)q] + (long)I.nph * BGS_WAVES * 64 * I.nb;
        R.hdr[(size_t)q * BGS_HDR + 0] = (int)unit0[(size_t)q]; R.hdr[(size_t)q * BGS_HDR + 1] = I.nph;
        R.hdr[(size_t)q * BGS_HDR + 2] = R.blk_ptr[(size_t)q + 1] - R.blk_ptr[(size_t)q]; R.hdr[(size_t)q * BGS_HDR + 3] = I.nb;
        R.hdr[(size_t)q * BGS_HDR + 4] = (int)ent0[(size_t)q];
    }
    if (ent0[(size_t)nb] > 0x7fffffffl) return BgsPlan();
    R.urow.assign((size_t)unit0[(size_t)nb] * 16, 0);
    R.ecol.assign((size_t)ent0[(size_t)nb], BGS_PAD);
    R.eval.assign((size_t)ent0[(size_t)nb], 0.0);
    R.eentry.assign((size_t)ent0[(size_t)nb], -1);
    std::vector<long> rim_cnt((size_t)nb, 0);
    std::vector<char> no_diag((size_t)nb, 0);
    parallel_for(nb, 32, [&](long q0, long q1) {
        std::vector<std::pair<int, int>> ent;     // (position of the column, entry of G)
        std::vector<int> foreign;
        for (long q = q0; q < q1; q++) {
            foreign.clear();
            const Blk& I = info[(size_t)q];
            const int base = R.blk_ptr[(size_t)q], end = R.blk_ptr[(size_t)q + 1], m = end - base;
            for (int l = 0; l < BGS_ROWS; l++) R.brow[(size_t)q * BGS_ROWS + l] = R.rows[(size_t)base + (l < m ? l : 0)];
            const size_t S = (size_t)I.nb * BGS_BATCH;
            for (int p = 0; p < I.nph; p++) {
                const int t0 = I.ph_ptr[(size_t)p], cnt = I.ph_ptr[(size_t)p + 1] - t0;
                for (int w = 0; w < BGS_WAVES; w++) {
                    const size_t u = (size_t)unit0[(size_t)q] + (size_t)p * BGS_WAVES + w;
                    const size_t e0 = (size_t)ent0[(size_t)q] + ((size_t)p * BGS_WAVES + w) * 64 * I.nb;
                    for (int r = 0; r < lp; r++) {
                        // the wave's r-th row of the phase: local w + 4 r; beyond the phase's rows: one of its rows again
                        int t = w + BGS_WAVES * r;
                        const bool repeat = t >= cnt;
                        if (repeat) t = t % cnt;
                        const int loc = t0 + t, i = R.rows[(size_t)base + loc];
                        R.urow[u * 16 + r] = i;
                        R.urow[u * 16 + 8 + r] = loc;
                        ent.clear();
                        for (int pp = G.ptr[(size_t)i]; pp < G.ptr[(size_t)i + 1]; pp++) ent.emplace_back(pos[(size_t)G.col[(size_t)pp]], pp);
                        std::sort(ent.begin(), ent.end());
                        size_t s = e0 + (size_t)r * S;
                        bool diag = false;
                        for (const auto& e : ent) {
                            const int pj = e.first, j = G.col[(size_t)e.second];
                            int code;
                            if (j == i) { code = BGS_DIAG; diag = true; }
                            else if (pj >= base && pj < end) code = BGS_LOCAL0 - (pj - base);
                            else { code = j; if (!repeat) foreign.push_back(j); }
                            R.ecol[s] = code; R.eval[s] = G.val[(size_t)e.second]; R.eentry[s] = e.second;
                            s++;
                        }
                        if (!diag) no_diag[(size_t)q] = 1;
                    }
                    for (int r = lp; r < 8; r++) { R.urow[u * 16 + r] = R.urow[u * 16]; R.urow[u * 16 + 8 + r] = R.urow[u * 16 + 8]; }
                }
            }
            std::sort(foreign.begin(), foreign.end());
            rim_cnt[(size_t)q] = (long)(std::unique(foreign.begin(), foreign.end()) - foreign.begin());
        }
    });
    for (char c : no_diag) if (c) return BgsPlan();
    long rim = 0;
    for (int q = 0; q < nb; q++) rim += rim_cnt[(size_t)q];
    R.n = n; R.n_blocks = nb; R.n_colors = ncol;
    R.rim = (double)rim / n;
    R.fill = (double)n / ((double)unit0[(size_t)nb] * lp);
    return R;
}

}  // namespace smg
