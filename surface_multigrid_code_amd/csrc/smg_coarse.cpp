// smg_coarse.cpp -- host side of the sparse coarse solver (smg_coarse.hpp): nested-dissection ordering, up-looking Cholesky.
#include "smg_coarse.hpp"

#include <algorithm>
#include <cmath>

namespace smg {

std::vector<int> nested_dissection_order(const Csr& A, int leaf)
{
    const int n = A.nr;
    std::vector<int> order((size_t)n), stamp((size_t)n, -1), dist((size_t)n, -1), queue;
    for (int i = 0; i < n; i++) order[(size_t)i] = i;
    struct Seg { int b, e; };
    std::vector<Seg> work{{0, n}};
    int next_id = 0;
    while (!work.empty()) {
        const Seg sg = work.back();
        work.pop_back();
        const int len = sg.e - sg.b;
        if (len <= leaf) continue;
        const int id = next_id++;
        for (int i = sg.b; i < sg.e; i++) stamp[(size_t)order[(size_t)i]] = id;
        // level structure of the segment from `root` (other components continue the level count); returns the last vertex reached
        auto bfs = [&](int root) {
            queue.clear();
            for (int i = sg.b; i < sg.e; i++) dist[(size_t)order[(size_t)i]] = -1;
            size_t head = 0;
            int scan = sg.b, cur = root;
            while ((int)queue.size() < len) {
                if (head == queue.size()) {
                    while (cur < 0 || dist[(size_t)cur] >= 0) { cur = order[(size_t)scan]; scan++; }
                    dist[(size_t)cur] = queue.empty() ? 0 : dist[(size_t)queue.back()] + 1;
                    queue.push_back(cur);
                    cur = -1;
                }
                const int v = queue[head++];
                for (int p = A.ptr[(size_t)v]; p < A.ptr[(size_t)v + 1]; p++) {
                    const int q = A.col[(size_t)p];
                    if (q != v && stamp[(size_t)q] == id && dist[(size_t)q] < 0) { dist[(size_t)q] = dist[(size_t)v] + 1; queue.push_back(q); }
                }
            }
            return queue.back();
        };
        const int far1 = bfs(order[(size_t)sg.b]);
        bfs(far1);
        // separator = the level that holds the median vertex of the breadth-first order; A = the levels before it, B = those after
        const int m = dist[(size_t)queue[(size_t)len / 2]];
        int a0 = 0, a1 = 0;     // queue[a0, a1) = level m (the queue is sorted by level)
        while (a0 < len && dist[(size_t)queue[(size_t)a0]] < m) a0++;
        a1 = a0;
        while (a1 < len && dist[(size_t)queue[(size_t)a1]] == m) a1++;
        const int nA = a0, nS = a1 - a0, nB = len - a1;
        if (nA == 0 || nB == 0) {      // no proper split (a clique-like part): keep the breadth-first order
            for (int i = 0; i < len; i++) order[(size_t)sg.b + i] = queue[(size_t)i];
            continue;
        }
        int w = sg.b;
        for (int i = 0; i < nA; i++) order[(size_t)w++] = queue[(size_t)i];
        for (int i = a1; i < len; i++) order[(size_t)w++] = queue[(size_t)i];
        for (int i = a0; i < a1; i++) order[(size_t)w++] = queue[(size_t)i];
        (void)nS;
        work.push_back({sg.b, sg.b + nA});
        work.push_back({sg.b + nA, sg.b + nA + nB});
    }
    return order;
}

bool sparse_cholesky(const Csr& A, SparseChol& F, bool reuse_symbolic)
{
    const int n = A.nr;
    if (!reuse_symbolic || (int)F.perm.size() != n) F.perm = nested_dissection_order(A);
    F.n = n;
    const Csr B = permute(A, F.perm, F.perm);
    // elimination tree (Liu), from the strict lower triangle of the rows
    F.parent.assign((size_t)n, -1);
    {
        std::vector<int> anc((size_t)n, -1);
        for (int k = 0; k < n; k++)
            for (int p = B.ptr[(size_t)k]; p < B.ptr[(size_t)k + 1]; p++) {
                int i = B.col[(size_t)p];
                while (i != -1 && i < k) {
                    const int nx = anc[(size_t)i];
                    anc[(size_t)i] = k;
                    if (nx == -1) F.parent[(size_t)i] = k;
                    i = nx;
                }
            }
    }
    // pattern of row k of L = the nodes reached in the tree from the entries of row k of B (topological order on `stack`)
    std::vector<int> w((size_t)n, -1), stack((size_t)n), s((size_t)n);
    auto ereach = [&](int k) {
        int top = n;
        w[(size_t)k] = k;
        for (int p = B.ptr[(size_t)k]; p < B.ptr[(size_t)k + 1]; p++) {
            int i = B.col[(size_t)p];
            if (i >= k) continue;
            int len = 0;
            for (; w[(size_t)i] != k; i = F.parent[(size_t)i]) { s[(size_t)len++] = i; w[(size_t)i] = k; }
            while (len > 0) stack[(size_t)--top] = s[(size_t)--len];
        }
        return top;
    };
    // symbolic: column counts
    std::vector<int> cnt((size_t)n, 0);
    for (int k = 0; k < n; k++) for (int t = ereach(k); t < n; t++) cnt[(size_t)stack[(size_t)t]]++;
    F.cptr.assign((size_t)n + 1, 0);
    for (int i = 0; i < n; i++) F.cptr[(size_t)i + 1] = F.cptr[(size_t)i] + cnt[(size_t)i];
    const size_t nnz = (size_t)F.cptr[(size_t)n];
    F.crow.assign(nnz, 0); F.cval.assign(nnz, 0.0); F.diag.assign((size_t)n, 0.0);
    std::fill(w.begin(), w.end(), -1);
    // numeric, up-looking: row k of L from a sparse triangular solve with the rows above
    std::vector<int> cpos(F.cptr.begin(), F.cptr.end() - 1);
    std::vector<double> x((size_t)n, 0.0);
    for (int k = 0; k < n; k++) {
        const int top = ereach(k);
        double d = 0.0;
        for (int p = B.ptr[(size_t)k]; p < B.ptr[(size_t)k + 1]; p++) {
            const int i = B.col[(size_t)p];
            if (i < k) x[(size_t)i] = B.val[(size_t)p];
            else if (i == k) d = B.val[(size_t)p];
        }
        for (int t = top; t < n; t++) {
            const int i = stack[(size_t)t];
            const double lki = x[(size_t)i] / F.diag[(size_t)i];
            x[(size_t)i] = 0.0;
            for (int p = F.cptr[(size_t)i]; p < cpos[(size_t)i]; p++) x[(size_t)F.crow[(size_t)p]] -= F.cval[(size_t)p] * lki;
            d -= lki * lki;
            F.crow[(size_t)cpos[(size_t)i]] = k;
            F.cval[(size_t)cpos[(size_t)i]] = lki;
            cpos[(size_t)i]++;
        }
        if (!(d > 0.0) || !std::isfinite(d)) return false;
        F.diag[(size_t)k] = std::sqrt(d);
    }
    // the same entries by rows (columns ascend inside a row: the columns were filled in row order)
    F.rptr.assign((size_t)n + 1, 0);
    for (size_t p = 0; p < nnz; p++) F.rptr[(size_t)F.crow[p] + 1]++;
    for (int i = 0; i < n; i++) F.rptr[(size_t)i + 1] += F.rptr[(size_t)i];
    F.rcol.assign(nnz, 0); F.rval.assign(nnz, 0.0);
    std::vector<int> rpos(F.rptr.begin(), F.rptr.end() - 1);
    for (int j = 0; j < n; j++)
        for (int p = F.cptr[(size_t)j]; p < F.cptr[(size_t)j + 1]; p++) {
            const int r = F.crow[(size_t)p];
            F.rcol[(size_t)rpos[(size_t)r]] = j;
            F.rval[(size_t)rpos[(size_t)r]] = F.cval[(size_t)p];
            rpos[(size_t)r]++;
        }
    return true;
}

}  // namespace smg
