// smg_sparse.cpp -- host-side sparse kernels of libsmg (see smg_sparse.hpp).
#include "smg_sparse.hpp"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <numeric>
#include <thread>
#include <utility>

namespace smg {

// ---------------------------------------------------------------------------------------------- host parallelism
int host_threads()
{
    static const int n = [] {
        const char* v = std::getenv("SMG_HOST_THREADS");
        int t = v && *v ? std::atoi(v) : (int)std::min(32u, std::max(1u, std::thread::hardware_concurrency()));
        return std::max(1, t);
    }();
    return n;
}
static thread_local bool tl_inside_parallel = false;

void parallel_for(long n, long grain, const std::function<void(long, long)>& fn)
{
    if (n <= 0) return;
    const int T = (int)std::min<long>(host_threads(), (n + grain - 1) / std::max<long>(grain, 1));
    if (T <= 1 || tl_inside_parallel) { fn(0, n); return; }
    std::vector<std::thread> th;
    th.reserve(T - 1);
    auto body = [&](int t) {
        tl_inside_parallel = true;
        const long b = n * t / T, e = n * (t + 1) / T;
        if (e > b) fn(b, e);
        tl_inside_parallel = false;
    };
    for (int t = 1; t < T; t++) th.emplace_back(body, t);
    body(0);
    for (auto& x : th) x.join();
}

void parallel_tasks(const std::vector<std::function<void()>>& tasks)
{
    if (tasks.empty()) return;
    if (tasks.size() == 1 || host_threads() <= 1 || tl_inside_parallel) { for (auto& f : tasks) f(); return; }
    // tasks keep the right to parallel_for themselves: they run on plain threads, not flagged as "inside"
    std::atomic<size_t> next{0};
    const int T = (int)std::min<size_t>(tasks.size(), (size_t)host_threads());
    auto worker = [&] { for (size_t i; (i = next.fetch_add(1)) < tasks.size();) tasks[i](); };
    std::vector<std::thread> th;
    for (int t = 1; t < T; t++) th.emplace_back(worker);
    worker();
    for (auto& x : th) x.join();
}

Csr csr_from_arrays(int nr, int nc, const int* ptr, const int* col, const double* val)
{
    Csr A;
    A.nr = nr; A.nc = nc;
    A.ptr.assign(nr + 1, 0);
    long nnz = ptr[nr];
    A.col.reserve(nnz); A.val.reserve(nnz);
    std::vector<std::pair<int, double>> row;
    for (int i = 0; i < nr; i++) {
        int b = ptr[i], e = ptr[i + 1];
        bool sorted = true;
        for (int p = b + 1; p < e; p++) if (col[p] <= col[p - 1]) { sorted = false; break; }
        if (sorted) {
            for (int p = b; p < e; p++) { A.col.push_back(col[p]); A.val.push_back(val[p]); }
        } else {
            row.clear();
            for (int p = b; p < e; p++) row.emplace_back(col[p], val[p]);
            std::stable_sort(row.begin(), row.end(),
                             [](const std::pair<int, double>& a, const std::pair<int, double>& b) { return a.first < b.first; });
            for (size_t t = 0; t < row.size(); t++) {
                if (t > 0 && row[t].first == A.col.back() && (long)A.col.size() > A.ptr[i]) A.val.back() += row[t].second;
                else { A.col.push_back(row[t].first); A.val.push_back(row[t].second); }
            }
        }
        A.ptr[i + 1] = (int)A.col.size();
    }
    return A;
}

Csr csr_from_csc_arrays(int nr, int nc, const int* colptr, const int* rowidx, const double* val)
{
    // the CSC arrays of an nr x nc matrix are the CSR arrays of its nc x nr transpose
    Csr T = csr_from_arrays(nc, nr, colptr, rowidx, val);
    return transpose(T);
}

Csr transpose(const Csr& A, std::vector<int>* src)
{
    Csr T;
    if (src) src->assign(A.nnz(), 0);
    T.nr = A.nc; T.nc = A.nr;
    long nnz = A.nnz();
    T.ptr.assign(T.nr + 1, 0);
    T.col.resize(nnz); T.val.resize(nnz);
    for (long p = 0; p < nnz; p++) T.ptr[A.col[p] + 1]++;
    for (int i = 0; i < T.nr; i++) T.ptr[i + 1] += T.ptr[i];
    std::vector<int> next(T.ptr.begin(), T.ptr.end() - 1);
    for (int i = 0; i < A.nr; i++)
        for (int p = A.ptr[i]; p < A.ptr[i + 1]; p++) {
            int q = next[A.col[p]]++;
            T.col[q] = i;
            T.val[q] = A.val[p];
            if (src) (*src)[q] = p;
        }
    return T;
}

Csr spgemm(const Csr& A, const Csr& B)
{
    Csr C;
    C.nr = A.nr; C.nc = B.nc;
    C.ptr.assign(C.nr + 1, 0);
    // symbolic pass (row sizes), rows in parallel: every thread owns a marker array
    parallel_for(A.nr, 4096, [&](long r0, long r1) {
        std::vector<int> mark(B.nc, -1);
        for (long i = r0; i < r1; i++) {
            int cnt = 0;
            for (int pa = A.ptr[i]; pa < A.ptr[i + 1]; pa++) {
                const int k = A.col[pa];
                for (int pb = B.ptr[k]; pb < B.ptr[k + 1]; pb++) {
                    const int j = B.col[pb];
                    if (mark[j] != (int)i) { mark[j] = (int)i; cnt++; }
                }
            }
            C.ptr[i + 1] = cnt;
        }
    });
    for (int i = 0; i < C.nr; i++) C.ptr[i + 1] += C.ptr[i];
    C.col.resize(C.ptr[C.nr]); C.val.resize(C.ptr[C.nr]);
    // numeric pass: per row the same ascending-k accumulation as before
    parallel_for(A.nr, 4096, [&](long r0, long r1) {
        std::vector<int> mark(B.nc, -1);
        std::vector<double> acc(B.nc, 0.0);
        std::vector<int> idx;
        idx.reserve(64);
        for (long i = r0; i < r1; i++) {
            idx.clear();
            for (int pa = A.ptr[i]; pa < A.ptr[i + 1]; pa++) {
                const int k = A.col[pa];
                const double a = A.val[pa];
                for (int pb = B.ptr[k]; pb < B.ptr[k + 1]; pb++) {
                    const int j = B.col[pb];
                    if (mark[j] != (int)i) { mark[j] = (int)i; acc[j] = a * B.val[pb]; idx.push_back(j); }
                    else acc[j] += a * B.val[pb];
                }
            }
            std::sort(idx.begin(), idx.end());
            const int base = C.ptr[i];
            for (size_t t = 0; t < idx.size(); t++) { C.col[base + t] = idx[t]; C.val[base + t] = acc[idx[t]]; }
        }
    });
    return C;
}

Csr slice(const Csr& X, const std::vector<int>* rows, const std::vector<int>* cols, std::vector<int>* src)
{
    Csr Y;
    Y.nr = rows ? (int)rows->size() : X.nr;
    Y.nc = cols ? (int)cols->size() : X.nc;
    Y.ptr.assign(Y.nr + 1, 0);
    std::vector<int> cmap;
    if (cols) {
        cmap.assign(X.nc, -1);
        for (int j = 0; j < Y.nc; j++) cmap[(*cols)[j]] = j;
    }
    // pass 1: row sizes
    parallel_for(Y.nr, 16384, [&](long r0, long r1) {
        for (long i = r0; i < r1; i++) {
            const int r = rows ? (*rows)[i] : (int)i;
            int cnt = 0;
            if (!cols) cnt = X.ptr[r + 1] - X.ptr[r];
            else for (int p = X.ptr[r]; p < X.ptr[r + 1]; p++) cnt += cmap[X.col[p]] >= 0;
            Y.ptr[i + 1] = cnt;
        }
    });
    for (int i = 0; i < Y.nr; i++) Y.ptr[i + 1] += Y.ptr[i];
    const long nnz = Y.ptr[Y.nr];
    Y.col.resize(nnz); Y.val.resize(nnz);
    if (src) src->assign(nnz, 0);
    // pass 2: fill, rows sorted by new column
    parallel_for(Y.nr, 16384, [&](long r0, long r1) {
        std::vector<std::pair<int, int>> row;  // (new column, index into X)
        for (long i = r0; i < r1; i++) {
            const int r = rows ? (*rows)[i] : (int)i;
            row.clear();
            for (int p = X.ptr[r]; p < X.ptr[r + 1]; p++) {
                const int j = cols ? cmap[X.col[p]] : X.col[p];
                if (j >= 0) row.emplace_back(j, p);
            }
            if (cols) std::sort(row.begin(), row.end(),
                                [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.first < b.first; });
            long o = Y.ptr[i];
            for (auto& e : row) { Y.col[o] = e.first; Y.val[o] = X.val[e.second]; if (src) (*src)[o] = e.second; o++; }
        }
    });
    return Y;
}

Csr permute(const Csr& A, const std::vector<int>& rperm, const std::vector<int>& cperm, std::vector<int>* src)
{
    return slice(A, &rperm, &cperm, src);
}

void spgemm_recipe(const Csr& A, const Csr& B, bool coef_from_A, const Csr& C, Recipe& R)
{
    // position of column j inside row i of C
    std::vector<int> pos(C.nc, -1);
    R.ptr.assign(C.nnz() + 1, 0);
    for (int i = 0; i < A.nr; i++) {
        for (int e = C.ptr[i]; e < C.ptr[i + 1]; e++) pos[C.col[e]] = e;
        for (int pa = A.ptr[i]; pa < A.ptr[i + 1]; pa++) {
            const int k = A.col[pa];
            for (int pb = B.ptr[k]; pb < B.ptr[k + 1]; pb++) R.ptr[pos[B.col[pb]] + 1]++;
        }
    }
    for (long e = 0; e < C.nnz(); e++) R.ptr[e + 1] += R.ptr[e];
    R.idx.resize(R.ptr[C.nnz()]);
    R.coef.resize(R.ptr[C.nnz()]);
    std::vector<int> next(R.ptr.begin(), R.ptr.end() - 1);
    for (int i = 0; i < A.nr; i++) {
        for (int e = C.ptr[i]; e < C.ptr[i + 1]; e++) pos[C.col[e]] = e;
        for (int pa = A.ptr[i]; pa < A.ptr[i + 1]; pa++) {  // ascending k: the accumulation order of spgemm()
            const int k = A.col[pa];
            for (int pb = B.ptr[k]; pb < B.ptr[k + 1]; pb++) {
                const int t = next[pos[B.col[pb]]]++;
                if (coef_from_A) { R.coef[t] = A.val[pa]; R.idx[t] = pb; }
                else { R.coef[t] = B.val[pb]; R.idx[t] = pa; }
            }
        }
    }
}

std::vector<double> diagonal(const Csr& A)
{
    std::vector<double> d(A.nr, 0.0);
    for (int i = 0; i < A.nr; i++)
        for (int p = A.ptr[i]; p < A.ptr[i + 1]; p++)
            if (A.col[p] == i) d[i] = A.val[p];
    return d;
}

void spmv_host(const Csr& A, const double* x, double* y)
{
    for (int i = 0; i < A.nr; i++) {
        double s = 0.0;
        for (int p = A.ptr[i]; p < A.ptr[i + 1]; p++) s += A.val[p] * x[A.col[p]];
        y[i] = s;
    }
}

}  // namespace smg
