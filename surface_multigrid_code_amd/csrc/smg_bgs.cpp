// smg_bgs.cpp -- host side of the block-sequential Gauss-Seidel sweep for many right-hand sides (smg_bgs.hpp): blocks, block colours,
// the order inside a block, the per-row entry batches in ascending column of the bgs order.
#include "smg_bgs.hpp"

#include <algorithm>
#include <numeric>

namespace smg {

BgsPlan build_bgs(const Csr& G, int block_rows)
{
    BgsPlan R;
    const int n = G.nr;
    if (n == 0 || block_rows < 2) return R;
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
    // order inside a block: breadth-first from a far end of the block (fronts of a roundish block of 60 rows are <= ~10 rows wide, so
    // the earlier neighbours of a row sit within the last two fronts: the ring)
    R.rows.assign((size_t)n, 0);
    std::vector<int> pos((size_t)n, 0);          // row -> position in the bgs order
    parallel_for(nb, 32, [&](long q0, long q1) {
        std::vector<int> queue, local;
        std::vector<int> seen;
        for (long q = q0; q < q1; q++) {
            const int b = blocks[(size_t)q], m0 = mptr[(size_t)b], m1 = mptr[(size_t)b + 1], cnt = m1 - m0;
            // local index of a member by binary search in the (ascending) member list
            auto loc = [&](int row) { return (int)(std::lower_bound(members.begin() + m0, members.begin() + m1, row) - (members.begin() + m0)); };
            auto bfs = [&](int root_local) {
                queue.clear();
                seen.assign((size_t)cnt, 0);
                int next_root = root_local, scan = 0;
                size_t head = 0;
                while ((int)queue.size() < cnt) {
                    if (head == queue.size()) {           // start, or another component of the block
                        while (next_root < 0 || seen[(size_t)next_root]) next_root = scan++;
                        seen[(size_t)next_root] = 1;
                        queue.push_back(next_root);
                        next_root = -1;
                    }
                    const int v = members[(size_t)m0 + queue[head++]];
                    for (int p = G.ptr[(size_t)v]; p < G.ptr[(size_t)v + 1]; p++) {
                        const int w = G.col[(size_t)p];
                        if (part[(size_t)w] != b) continue;
                        const int lw = loc(w);
                        if (!seen[(size_t)lw]) { seen[(size_t)lw] = 1; queue.push_back(lw); }
                    }
                }
                return queue.back();
            };
            const int far = bfs(0);
            bfs(far);
            const int base = R.blk_ptr[(size_t)q];
            for (int t = 0; t < cnt; t++) {
                const int row = members[(size_t)m0 + queue[(size_t)t]];
                R.rows[(size_t)base + t] = row;
                pos[(size_t)row] = base + t;
            }
        }
    });
    // batches per row: the same for all rows of a block; chunks of 64 entry slots
    R.hdr.assign((size_t)nb * 4, 0);
    std::vector<int> first_chunk((size_t)nb + 1, 0);
    {
        int prow_off = 0;
        for (int q = 0; q < nb; q++) {
            int wmax = 1;
            for (int t = R.blk_ptr[(size_t)q]; t < R.blk_ptr[(size_t)q + 1]; t++) {
                const int i = R.rows[(size_t)t];
                wmax = std::max(wmax, G.ptr[(size_t)i + 1] - G.ptr[(size_t)i]);
            }
            const int per_row = (wmax + BGS_BATCH - 1) / BGS_BATCH;
            if (per_row > BGS_MAX_BATCHES) return BgsPlan();
            const int m = R.blk_ptr[(size_t)q + 1] - R.blk_ptr[(size_t)q], rows_per_chunk = 8 / per_row;
            const int chunks = (m + rows_per_chunk - 1) / rows_per_chunk;
            R.hdr[(size_t)q * 4 + 0] = prow_off; R.hdr[(size_t)q * 4 + 1] = m;
            R.hdr[(size_t)q * 4 + 2] = first_chunk[(size_t)q]; R.hdr[(size_t)q * 4 + 3] = per_row;
            first_chunk[(size_t)q + 1] = first_chunk[(size_t)q] + chunks;
            prow_off += chunks * rows_per_chunk;
        }
        R.prow.assign((size_t)prow_off, 0);
        const size_t slots = (size_t)first_chunk[(size_t)nb] * 64;
        R.ecol.assign(slots, BGS_PAD);
        R.eval.assign(slots, 0.0);
        R.eentry.assign(slots, -1);
    }
    std::vector<long> rim_cnt((size_t)nb, 0), in_early((size_t)nb, 0), in_ring((size_t)nb, 0);
    std::vector<char> no_diag((size_t)nb, 0);
    parallel_for(nb, 32, [&](long q0, long q1) {
        std::vector<std::pair<int, int>> ent;     // (position of the column, entry of G)
        std::vector<int> foreign;
        for (long q = q0; q < q1; q++) {
            foreign.clear();
            const int base = R.blk_ptr[(size_t)q], end = R.blk_ptr[(size_t)q + 1];
            for (int t = base; t < end; t++) {
                const int i = R.rows[(size_t)t];
                ent.clear();
                for (int p = G.ptr[(size_t)i]; p < G.ptr[(size_t)i + 1]; p++) ent.emplace_back(pos[(size_t)G.col[(size_t)p]], p);
                std::sort(ent.begin(), ent.end());
                const int per_row = R.hdr[(size_t)q * 4 + 3];
                size_t s = (size_t)R.hdr[(size_t)q * 4 + 2] * 64 + (size_t)(t - base) * per_row * BGS_BATCH;
                R.prow[(size_t)R.hdr[(size_t)q * 4 + 0] + (t - base)] = i;
                bool diag = false;
                for (const auto& e : ent) {
                    const int pj = e.first, j = G.col[(size_t)e.second];
                    int code;
                    if (j == i) { code = BGS_DIAG; diag = true; }
                    else if (pj >= base && pj < t) {           // earlier row of this block
                        in_early[(size_t)q]++;
                        if (t - pj <= BGS_RING) { code = BGS_RING0 - ((pj - base) % BGS_RING); in_ring[(size_t)q]++; }
                        else code = j;
                    } else {
                        code = j;
                        if (pj < base || pj >= end) foreign.push_back(j);
                    }
                    R.ecol[s] = code; R.eval[s] = G.val[(size_t)e.second]; R.eentry[s] = e.second;
                    s++;
                }
                if (!diag) no_diag[(size_t)q] = 1;
            }
            {   // copies of the last row up to a whole chunk
                const int per_row = R.hdr[(size_t)q * 4 + 3], m = end - base, rows_per_chunk = 8 / per_row;
                const int padded = (m + rows_per_chunk - 1) / rows_per_chunk * rows_per_chunk;
                const size_t e0 = (size_t)R.hdr[(size_t)q * 4 + 2] * 64, w = (size_t)per_row * BGS_BATCH;
                for (int t = m; t < padded; t++) {
                    R.prow[(size_t)R.hdr[(size_t)q * 4 + 0] + t] = R.rows[(size_t)end - 1];
                    for (size_t z = 0; z < w; z++) {
                        R.ecol[e0 + (size_t)t * w + z] = R.ecol[e0 + (size_t)(m - 1) * w + z];
                        R.eval[e0 + (size_t)t * w + z] = R.eval[e0 + (size_t)(m - 1) * w + z];
                        R.eentry[e0 + (size_t)t * w + z] = R.eentry[e0 + (size_t)(m - 1) * w + z];
                    }
                }
            }
            std::sort(foreign.begin(), foreign.end());
            rim_cnt[(size_t)q] = (long)(std::unique(foreign.begin(), foreign.end()) - foreign.begin());
        }
    });
    for (char c : no_diag) if (c) return BgsPlan();
    long rim = 0, early = 0, ring = 0;
    for (int q = 0; q < nb; q++) { rim += rim_cnt[(size_t)q]; early += in_early[(size_t)q]; ring += in_ring[(size_t)q]; }
    R.n = n; R.n_blocks = nb; R.n_colors = ncol;
    R.rim = (double)rim / n;
    R.ring_hits = early ? (double)ring / (double)early : 1.0;
    return R;
}

}  // namespace smg
