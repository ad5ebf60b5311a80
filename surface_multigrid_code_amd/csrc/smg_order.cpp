// smg_order.cpp -- see smg_order.hpp.
#include "smg_order.hpp"

#include <algorithm>
#include <numeric>

namespace smg {

std::vector<int> rcm_order(const Csr& A)
{
    int n = A.nr;
    std::vector<int> order;
    order.reserve(n);
    std::vector<char> seen(n, 0);
    std::vector<int> deg(n);
    for (int i = 0; i < n; i++) deg[i] = A.ptr[i + 1] - A.ptr[i];
    // component seeds in ascending degree (cheap stand-in for a pseudo-peripheral search)
    std::vector<int> seeds(n);
    std::iota(seeds.begin(), seeds.end(), 0);
    std::stable_sort(seeds.begin(), seeds.end(), [&](int a, int b) { return deg[a] < deg[b]; });
    std::vector<int> nb;
    for (int s : seeds) {
        if (seen[s]) continue;
        // one BFS to find a far vertex, then the real BFS from there (George-Liu style, one round)
        size_t mark = order.size();
        int start = s;
        for (int round = 0; round < 2; round++) {
            order.resize(mark);
            size_t head = mark;
            order.push_back(start);
            seen[start] = 1;
            while (head < order.size()) {
                int v = order[head++];
                nb.clear();
                for (int p = A.ptr[v]; p < A.ptr[v + 1]; p++) {
                    int w = A.col[p];
                    if (w != v && !seen[w]) { seen[w] = 1; nb.push_back(w); }
                }
                std::sort(nb.begin(), nb.end(), [&](int a, int b) { return deg[a] != deg[b] ? deg[a] < deg[b] : a < b; });
                order.insert(order.end(), nb.begin(), nb.end());
            }
            if (round == 0) {
                int far = order.back();
                for (size_t t = mark; t < order.size(); t++) seen[order[t]] = 0;
                start = far;
            }
        }
    }
    std::reverse(order.begin(), order.end());
    return order;
}

Ordering identity_ordering(int n)
{
    Ordering o;
    o.perm.resize(n);
    std::iota(o.perm.begin(), o.perm.end(), 0);
    o.iperm = o.perm;
    o.color_ptr = {0, n};
    return o;
}

Ordering make_ordering(const Csr& A, int sigma)
{
    int n = A.nr;
    Ordering o;
    std::vector<int> rcm = rcm_order(A);  // new -> old
    // greedy first-fit colouring, visiting vertices in RCM order
    std::vector<int> color(n, -1);
    std::vector<int> forbid;  // forbid[c] == v  <=> colour c used by a neighbour of v
    int ncol = 0;
    for (int t = 0; t < n; t++) {
        int v = rcm[t];
        for (int p = A.ptr[v]; p < A.ptr[v + 1]; p++) {
            int w = A.col[p];
            if (w != v && color[w] >= 0) {
                if ((int)forbid.size() <= color[w]) forbid.resize(color[w] + 1, -1);
                forbid[color[w]] = v;
            }
        }
        int c = 0;
        while (c < (int)forbid.size() && forbid[c] == v) c++;
        if ((int)forbid.size() <= c) forbid.resize(c + 1, -1);
        color[v] = c;
        ncol = std::max(ncol, c + 1);
    }
    // colour-major, RCM rank inside a colour (counting sort keeps the RCM order stable)
    o.color_ptr.assign(ncol + 1, 0);
    for (int v = 0; v < n; v++) o.color_ptr[color[v] + 1]++;
    for (int c = 0; c < ncol; c++) o.color_ptr[c + 1] += o.color_ptr[c];
    o.perm.resize(n);
    {
        std::vector<int> next(o.color_ptr.begin(), o.color_ptr.end() - 1);
        for (int t = 0; t < n; t++) { int v = rcm[t]; o.perm[next[color[v]]++] = v; }
    }
    // rows bucketed by nnz: inside each sigma-row window of a colour, longest rows first (stable)
    if (sigma > 1) {
        for (int c = 0; c < ncol; c++)
            for (int b = o.color_ptr[c]; b < o.color_ptr[c + 1]; b += sigma) {
                int e = std::min(b + sigma, o.color_ptr[c + 1]);
                std::stable_sort(o.perm.begin() + b, o.perm.begin() + e, [&](int x, int y) {
                    return (A.ptr[x + 1] - A.ptr[x]) > (A.ptr[y + 1] - A.ptr[y]);
                });
            }
    }
    o.iperm.resize(n);
    for (int i = 0; i < n; i++) o.iperm[o.perm[i]] = i;
    if (n == 0) o.color_ptr = {0, 0};
    return o;
}

Sell build_sell(const Csr& A, const std::vector<int>* row_breaks)
{
    Sell S;
    S.n_rows = A.nr; S.n_cols = A.nc; S.nnz = A.nnz();
    std::vector<int> breaks;
    if (row_breaks) breaks = *row_breaks;
    else breaks = {0, A.nr};
    S.slice_row.push_back(0);
    S.slice_off.push_back(0);
    S.color_slice_ptr.push_back(0);
    for (size_t c = 0; c + 1 < breaks.size(); c++) {
        for (int r0 = breaks[c]; r0 < breaks[c + 1]; r0 += SELL_C) {
            int r1 = std::min(r0 + SELL_C, breaks[c + 1]);
            int w = 0;
            for (int r = r0; r < r1; r++) w = std::max(w, A.ptr[r + 1] - A.ptr[r]);
            S.slice_row.push_back(r1);
            S.slice_off.push_back(S.slice_off.back() + w);
        }
        S.color_slice_ptr.push_back((int)S.slice_row.size() - 1);
    }
    S.n_slices = (int)S.slice_row.size() - 1;
    size_t tot = (size_t)SELL_C * (size_t)S.slice_off.back();
    S.col.assign(tot, -1);
    S.val.assign(tot, 0.0);
    for (int s = 0; s < S.n_slices; s++) {
        size_t base = (size_t)SELL_C * (size_t)S.slice_off[s];
        for (int r = S.slice_row[s]; r < S.slice_row[s + 1]; r++) {
            int lane = r - S.slice_row[s];
            int j = 0;
            for (int p = A.ptr[r]; p < A.ptr[r + 1]; p++, j++) {
                S.col[base + (size_t)j * SELL_C + lane] = A.col[p];
                S.val[base + (size_t)j * SELL_C + lane] = A.val[p];
            }
        }
    }
    return S;
}

}  // namespace smg
