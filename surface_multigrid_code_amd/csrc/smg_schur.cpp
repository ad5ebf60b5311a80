// smg_schur.cpp -- the plan of the Schur-complement coarse solver (smg_schur.hpp): blocks, separator, arena layout, scatter and sum lists.
#include "smg_schur.hpp"

#include <algorithm>
#include <queue>
#include <tuple>

#include "smg_bgs.hpp"   // partition_tiles

namespace smg {

SchurPlan build_schur(const Csr& A, int block_rows)
{
    SchurPlan Pn;
    const int n = A.nr;
    if (n <= 0 || A.nc != n || block_rows < 1 || block_rows > SCHUR_B) return Pn;
    int n_parts = 0;
    const std::vector<int> part = partition_tiles(A, block_rows, &n_parts);
    // ---- separator: a vertex cover of the edges that join different parts, greedily by the number of such edges still uncovered
    // (ties: the smaller row -- the plan is a function of the matrix alone)
    std::vector<int> deg((size_t)n, 0);
    for (int u = 0; u < n; u++)
        for (int p = A.ptr[u]; p < A.ptr[u + 1]; p++) if (part[(size_t)A.col[p]] != part[(size_t)u]) deg[(size_t)u]++;
    std::vector<char> in_s((size_t)n, 0);
    {
        typedef std::pair<int, int> Key;                      // (uncovered cut edges, -row): the largest first
        std::priority_queue<Key> heap;
        for (int u = 0; u < n; u++) if (deg[(size_t)u] > 0) heap.push(Key(deg[(size_t)u], -u));
        while (!heap.empty()) {
            const Key k = heap.top(); heap.pop();
            const int u = -k.second;
            if (in_s[(size_t)u] || k.first != deg[(size_t)u] || k.first == 0) continue;      // stale entry
            in_s[(size_t)u] = 1;
            deg[(size_t)u] = 0;
            for (int p = A.ptr[u]; p < A.ptr[u + 1]; p++) {
                const int v = A.col[p];
                if (part[(size_t)v] == part[(size_t)u] || in_s[(size_t)v]) continue;
                if (--deg[(size_t)v] > 0) heap.push(Key(deg[(size_t)v], -v));
            }
        }
    }
    // ---- numbering: separator rows by (part, row); the interiors of the parts, parts without one dropped
    std::vector<int> sep_of((size_t)n, -1), blk_of((size_t)n, -1), slot_of((size_t)n, -1);
    {
        std::vector<std::vector<int>> rows_of((size_t)n_parts);
        for (int u = 0; u < n; u++) rows_of[(size_t)part[(size_t)u]].push_back(u);
        for (int q = 0; q < n_parts; q++)
            for (int u : rows_of[(size_t)q]) if (in_s[(size_t)u]) { sep_of[(size_t)u] = (int)Pn.srow.size(); Pn.srow.push_back(u); }
        for (int q = 0; q < n_parts; q++) {
            int cnt = 0;
            for (int u : rows_of[(size_t)q]) if (!in_s[(size_t)u]) cnt++;
            if (cnt == 0) continue;
            if (cnt > SCHUR_B) return SchurPlan();
            const int i = Pn.nb++;
            Pn.irow.resize((size_t)Pn.nb * SCHUR_B, -1);
            int r = 0;
            for (int u : rows_of[(size_t)q]) if (!in_s[(size_t)u]) { blk_of[(size_t)u] = i; slot_of[(size_t)u] = r; Pn.irow[(size_t)i * SCHUR_B + r] = u; r++; }
            Pn.bsize.push_back(cnt);
        }
    }
    Pn.n = n;
    Pn.ns = (int)Pn.srow.size();
    Pn.ns_pad = (Pn.ns + 63) / 64 * 64;
    if (Pn.nb == 0 || Pn.ns == 0 || (double)Pn.ns > 0.7 * n || Pn.ns > SCHUR_NS_MAX) return SchurPlan();
    // ---- the separator rows every block touches (its local columns), and the inverse lists
    Pn.sptr.assign((size_t)Pn.nb + 1, 0);
    {
        std::vector<std::vector<int>> touch((size_t)Pn.nb);
        for (int u = 0; u < n; u++) {
            const int i = blk_of[(size_t)u];
            if (i < 0) continue;
            for (int p = A.ptr[u]; p < A.ptr[u + 1]; p++) {
                const int v = A.col[p];
                if (sep_of[(size_t)v] >= 0) touch[(size_t)i].push_back(sep_of[(size_t)v]);
                else if (blk_of[(size_t)v] != i) return SchurPlan();       // not structurally symmetric: an edge the cover never saw
            }
        }
        for (int i = 0; i < Pn.nb; i++) {
            std::vector<int>& t = touch[(size_t)i];
            std::sort(t.begin(), t.end());
            t.erase(std::unique(t.begin(), t.end()), t.end());
            if ((int)t.size() > SCHUR_M_MAX) return SchurPlan();
            Pn.sptr[(size_t)i + 1] = Pn.sptr[(size_t)i] + (int)t.size();
            Pn.sidx.insert(Pn.sidx.end(), t.begin(), t.end());
        }
    }
    Pn.aptr.assign((size_t)Pn.ns + 1, 0);
    for (int j : Pn.sidx) Pn.aptr[(size_t)j + 1]++;
    for (int j = 0; j < Pn.ns; j++) Pn.aptr[(size_t)j + 1] += Pn.aptr[(size_t)j];
    Pn.ablk.resize(Pn.sidx.size());
    Pn.apan.resize(Pn.sidx.size());
    {
        std::vector<int> fill(Pn.aptr.begin(), Pn.aptr.end() - 1);
        for (int i = 0; i < Pn.nb; i++)
            for (int q = Pn.sptr[(size_t)i]; q < Pn.sptr[(size_t)i + 1]; q++) {
                const int w = fill[(size_t)Pn.sidx[(size_t)q]]++;
                Pn.ablk[(size_t)w] = i; Pn.apan[(size_t)w] = q;
            }
    }
    // ---- arena
    const long long panel = (long long)SCHUR_B * Pn.sptr[(size_t)Pn.nb];
    Pn.off_D = 0;
    Pn.off_P = (long long)Pn.nb * SCHUR_B * SCHUR_B;
    Pn.off_W = Pn.off_P + panel;
    Pn.off_S = Pn.off_W + panel;
    Pn.off_C = Pn.off_S + (long long)Pn.ns_pad * Pn.ns_pad;
    Pn.coff.assign((size_t)Pn.nb + 1, 0);
    for (int i = 0; i < Pn.nb; i++) { const long long m = Pn.sptr[(size_t)i + 1] - Pn.sptr[(size_t)i]; Pn.coff[(size_t)i + 1] = Pn.coff[(size_t)i] + m * m; }
    Pn.total = Pn.off_C + Pn.coff[(size_t)Pn.nb];
    // ---- where the entries of A go (lower triangle of the caller's numbering, mirrored)
    Pn.pos.assign((size_t)A.nnz(), -1);
    Pn.pos2.assign((size_t)A.nnz(), -1);
    auto local_col = [&](int i, int j) {
        const int* b = Pn.sidx.data() + Pn.sptr[(size_t)i];
        const int* e = Pn.sidx.data() + Pn.sptr[(size_t)i + 1];
        return (int)(std::lower_bound(b, e, j) - b);
    };
    for (int r = 0; r < n; r++)
        for (int p = A.ptr[r]; p < A.ptr[r + 1]; p++) {
            const int c = A.col[p];
            if (c > r) continue;
            const int ir = blk_of[(size_t)r], ic = blk_of[(size_t)c];
            if (ir >= 0 && ic >= 0) {
                const long long base = Pn.off_D + (long long)ir * SCHUR_B * SCHUR_B;
                Pn.pos[(size_t)p] = base + (long long)slot_of[(size_t)r] * SCHUR_B + slot_of[(size_t)c];
                if (r != c) Pn.pos2[(size_t)p] = base + (long long)slot_of[(size_t)c] * SCHUR_B + slot_of[(size_t)r];
            } else if (ir >= 0 || ic >= 0) {
                const int u = ir >= 0 ? r : c, s = ir >= 0 ? c : r;      // interior row, separator row
                const int i = blk_of[(size_t)u];
                Pn.pos[(size_t)p] = Pn.off_P + (long long)SCHUR_B * Pn.sptr[(size_t)i] + (long long)local_col(i, sep_of[(size_t)s]) * SCHUR_B + slot_of[(size_t)u];
            } else {
                const long long a = sep_of[(size_t)r], b = sep_of[(size_t)c];
                Pn.pos[(size_t)p] = Pn.off_S + a * Pn.ns_pad + b;
                if (a != b) Pn.pos2[(size_t)p] = Pn.off_S + b * Pn.ns_pad + a;
            }
        }
    for (int i = 0; i < Pn.nb; i++)
        for (int r = 0; r < SCHUR_B; r++)
            if (Pn.irow[(size_t)i * SCHUR_B + r] < 0) Pn.ones.push_back(Pn.off_D + (long long)i * SCHUR_B * SCHUR_B + (long long)r * SCHUR_B + r);
    for (long long j = Pn.ns; j < Pn.ns_pad; j++) Pn.ones.push_back(Pn.off_S + j * Pn.ns_pad + j);
    // ---- S -= sum_i P_i^T W_i: per entry of S (lower triangle) the products that land on it, ascending block
    {
        typedef std::tuple<long long, int, long long> Term;   // (entry of S, block, where the product is)
        std::vector<Term> terms;
        for (int i = 0; i < Pn.nb; i++) {
            const int m = Pn.sptr[(size_t)i + 1] - Pn.sptr[(size_t)i];
            const int* sx = Pn.sidx.data() + Pn.sptr[(size_t)i];
            for (int c1 = 0; c1 < m; c1++)
                for (int c2 = 0; c2 <= c1; c2++)
                    terms.emplace_back((long long)sx[c1] * Pn.ns_pad + sx[c2], i, Pn.coff[(size_t)i] + (long long)c1 * m + c2);
        }
        std::sort(terms.begin(), terms.end());
        Pn.rptr.push_back(0);
        for (size_t t = 0; t < terms.size(); t++) {
            const long long key = std::get<0>(terms[t]);
            if (t == 0 || key != std::get<0>(terms[t - 1])) {
                if (t) Pn.rptr.push_back((int)t);
                const long long a = key / Pn.ns_pad, b = key % Pn.ns_pad;
                Pn.rdst.push_back(Pn.off_S + key);
                Pn.rdst2.push_back(a != b ? Pn.off_S + b * Pn.ns_pad + a : -1);
            }
            Pn.rsrc.push_back(std::get<2>(terms[t]));
        }
        Pn.rptr.push_back((int)terms.size());
        if (terms.empty()) Pn.rptr.assign(1, 0);
    }
    return Pn;
}

}  // namespace smg
