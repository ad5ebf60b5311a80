// smg_sparse.cpp -- host-side sparse kernels of libsmg (see smg_sparse.hpp).
#include "smg_sparse.hpp"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <exception>
#include <memory>
#include <mutex>
#include <cstdint>
#include <cstdlib>
#include <numeric>
#include <cmath>
#include <cstdio>
#include <string>
#include <thread>
#include <utility>

namespace smg {

// ---------------------------------------------------------------------------------------------- host parallelism
// CPUs this process may actually use: the hardware threads, cut down to the cgroup's CPU quota where there is one (a container on a
// 256-thread host with cpu.max = "1600000 100000" gets 16 CPUs' worth of time per period: 32 runnable threads spend their budget in half
// a period and are then ALL stalled for the other half).
static int usable_cpus()
{
    int n = (int)std::max(1u, std::thread::hardware_concurrency());
    auto read_pair = [](const char* path, double& a, double& b) {
        std::FILE* f = std::fopen(path, "r");
        if (!f) return false;
        char s1[64] = {0}, s2[64] = {0};
        const int got = std::fscanf(f, "%63s %63s", s1, s2);
        std::fclose(f);
        if (got < 1 || std::string(s1) == "max") return false;
        a = std::atof(s1); b = got >= 2 ? std::atof(s2) : 0.0;
        return a > 0.0;
    };
    double quota = 0.0, period = 0.0;
    if (read_pair("/sys/fs/cgroup/cpu.max", quota, period) && period > 0.0) n = std::min(n, std::max(1, (int)std::ceil(quota / period)));      // cgroup v2
    else {
        double q = 0.0, p = 0.0, dummy = 0.0;
        if (read_pair("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", q, dummy) && read_pair("/sys/fs/cgroup/cpu/cpu.cfs_period_us", p, dummy) && p > 0.0)
            n = std::min(n, std::max(1, (int)std::ceil(q / p)));                                                                             // cgroup v1
    }
    return n;
}

int host_threads()
{
    static const int n = [] {
        const char* v = std::getenv("SMG_HOST_THREADS");
        int t = v && *v ? std::atoi(v) : std::min(32, usable_cpus());
        return std::max(1, t);
    }();
    return n;
}
static thread_local bool tl_inside_parallel = false;

// A persistent pool: the precompute calls parallel_for dozens of times, and 31 thread creations + joins per call were a good part of
// the short ones.  A call posts one ticket per chunk beyond its own first; whoever holds a ticket claims the next unclaimed chunk of
// that job.  The caller claims chunks as well until none is left, then waits for the ones in flight -- it never depends on a worker
// being free (nested calls, a child process after fork() without workers: the caller simply ends up doing the chunks itself).
namespace {
struct Job {
    std::function<void(int)> run;     // chunk index -> work
    int total = 0;
    std::atomic<int> next{0}, done{0};
    std::atomic<bool> failed{false};
    std::exception_ptr err;           // the first exception of a chunk, rethrown by the caller of run()
};
class Pool {
public:
    static Pool& get() { static Pool p; return p; }
    void run(int chunks, std::function<void(int)> body)
    {
        auto job = std::make_shared<Job>();
        job->run = std::move(body);
        job->total = chunks;
        start_workers();
        {
            std::lock_guard<std::mutex> g(m_);
            for (int i = 1; i < chunks; i++) tickets_.push_back(job);
        }
        if (chunks > 2) cv_.notify_all(); else cv_.notify_one();
        work_on(*job);
        if (job->done.load(std::memory_order_acquire) < job->total) {
            std::unique_lock<std::mutex> g(m_);
            done_cv_.wait(g, [&] { return job->done.load(std::memory_order_acquire) >= job->total; });
        }
        if (job->failed.load(std::memory_order_acquire)) std::rethrow_exception(job->err);
    }
    ~Pool()
    {
        { std::lock_guard<std::mutex> g(m_); stop_ = true; }
        cv_.notify_all();
        for (auto& t : th_) if (t.joinable()) t.join();
    }
private:
    void work_on(Job& j)
    {
        for (int c; (c = j.next.fetch_add(1, std::memory_order_relaxed)) < j.total;) {
            try { if (!j.failed.load(std::memory_order_relaxed)) j.run(c); }
            catch (...) { if (!j.failed.exchange(true)) j.err = std::current_exception(); }
            if (j.done.fetch_add(1, std::memory_order_acq_rel) + 1 == j.total) {
                std::lock_guard<std::mutex> g(m_);     // (the waiter checks under this lock: no lost wake-up)
                done_cv_.notify_all();
            }
        }
    }
    void start_workers()
    {
        if (started_.load(std::memory_order_acquire)) return;
        std::lock_guard<std::mutex> g(m_);
        if (started_.load(std::memory_order_relaxed)) return;
        const int n = host_threads() - 1;
        for (int i = 0; i < n; i++) th_.emplace_back([this] { loop(); });
        started_.store(true, std::memory_order_release);
    }
    void loop()
    {
        for (;;) {
            std::shared_ptr<Job> j;
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&] { return stop_ || !tickets_.empty(); });
                if (stop_) return;
                j = std::move(tickets_.front());
                tickets_.pop_front();
            }
            work_on(*j);
        }
    }
    std::mutex m_;
    std::condition_variable cv_, done_cv_;
    std::deque<std::shared_ptr<Job>> tickets_;
    std::vector<std::thread> th_;
    std::atomic<bool> started_{false};
    bool stop_ = false;
};
}  // namespace

void parallel_for(long n, long grain, const std::function<void(long, long)>& fn)
{
    if (n <= 0) return;
    const int T = (int)std::min<long>(host_threads(), (n + grain - 1) / std::max<long>(grain, 1));
    if (T <= 1 || tl_inside_parallel) { fn(0, n); return; }
    Pool::get().run(T, [&fn, n, T](int t) {
        const bool was = tl_inside_parallel;
        tl_inside_parallel = true;
        const long b = n * t / T, e = n * (t + 1) / T;
        if (e > b) fn(b, e);
        tl_inside_parallel = was;
    });
}

void parallel_tasks(const std::vector<std::function<void()>>& tasks)
{
    if (tasks.empty()) return;
    if (tasks.size() == 1 || host_threads() <= 1 || tl_inside_parallel) { for (auto& f : tasks) f(); return; }
    // tasks keep the right to parallel_for themselves: they are not flagged as "inside"
    Pool::get().run((int)tasks.size(), [&tasks](int i) { tasks[(size_t)i](); });
}

bool rows_strictly_ascending(int nr, const int* ptr, const int* col)
{
    std::atomic<int> bad{0};
    parallel_for(nr, 65536, [&](long r0, long r1) {
        for (long i = r0; i < r1 && !bad.load(std::memory_order_relaxed); i++)
            for (int p = ptr[i] + 1; p < ptr[i + 1]; p++) if (col[p] <= col[p - 1]) { bad.store(1, std::memory_order_relaxed); break; }
    });
    return bad.load() == 0;
}

Csr csr_from_arrays(int nr, int nc, const int* ptr, const int* col, const double* val)
{
    Csr A;
    A.nr = nr; A.nc = nc;
    long nnz = ptr[nr];
    if (ptr[0] == 0 && rows_strictly_ascending(nr, ptr, col)) {      // already canonical: a plain (parallel) copy
        A.ptr.assign(ptr, ptr + nr + 1);
        A.col.resize((size_t)nnz); A.val.resize((size_t)nnz);
        parallel_for(nnz, 1 << 18, [&](long a, long b) {
            std::copy(col + a, col + b, A.col.begin() + a);
            std::copy(val + a, val + b, A.val.begin() + a);
        });
        return A;
    }
    A.ptr.assign(nr + 1, 0);
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

Csr copy_of(const Csr& A)
{
    Csr B;
    B.nr = A.nr; B.nc = A.nc;
    B.ptr.resize(A.ptr.size()); B.col.resize(A.col.size()); B.val.resize(A.val.size());
    parallel_for((long)A.ptr.size(), 1 << 18, [&](long a, long b) { std::copy(A.ptr.begin() + a, A.ptr.begin() + b, B.ptr.begin() + a); });
    parallel_for((long)A.col.size(), 1 << 18, [&](long a, long b) {
        std::copy(A.col.begin() + a, A.col.begin() + b, B.col.begin() + a);
        std::copy(A.val.begin() + a, A.val.begin() + b, B.val.begin() + a);
    });
    return B;
}

Csr transpose(const Csr& A, std::vector<int>* src)
{
    Csr T;
    if (src) src->assign(A.nnz(), 0);
    T.nr = A.nc; T.nc = A.nr;
    long nnz = A.nnz();
    T.ptr.assign((size_t)T.nr + 1, 0);
    T.col.resize(nnz); T.val.resize(nnz);
    // big matrices: row chunks counted and scattered by separate threads; within an output row the entries keep the order of the
    // sequential pass (chunk by chunk, row by row: ascending column), so the result is the same
    const int chunks = (int)std::min<long>(std::min(16, host_threads()), std::max<long>(1, (8L << 20) / std::max(T.nr, 1)));
    if (nnz >= 200000 && chunks > 1 && A.nr >= chunks) {
        std::vector<raw_vector<int>> cnt((size_t)chunks);
        auto row0 = [&](int c) { return (int)((long)A.nr * c / chunks); };
        std::vector<std::function<void()>> tasks;
        for (int c = 0; c < chunks; c++)
            tasks.push_back([&, c] {
                cnt[(size_t)c].assign((size_t)T.nr, 0);
                for (long p = A.ptr[row0(c)]; p < A.ptr[row0(c + 1)]; p++) cnt[(size_t)c][(size_t)A.col[p]]++;
            });
        parallel_tasks(tasks);
        // cnt[c][j] -> where chunk c starts writing in output row j
        parallel_for(T.nr, 1 << 15, [&](long j0, long j1) {
            for (long j = j0; j < j1; j++) {
                int run = 0;
                for (int c = 0; c < chunks; c++) { const int k = cnt[(size_t)c][(size_t)j]; cnt[(size_t)c][(size_t)j] = run; run += k; }
                T.ptr[(size_t)j + 1] = run;
            }
        });
        for (int j = 0; j < T.nr; j++) T.ptr[(size_t)j + 1] += T.ptr[(size_t)j];
        tasks.clear();
        for (int c = 0; c < chunks; c++)
            tasks.push_back([&, c] {
                raw_vector<int>& off = cnt[(size_t)c];
                for (int i = row0(c); i < row0(c + 1); i++)
                    for (int p = A.ptr[i]; p < A.ptr[i + 1]; p++) {
                        const int j = A.col[p];
                        const int q = T.ptr[(size_t)j] + off[(size_t)j]++;
                        T.col[(size_t)q] = i;
                        T.val[(size_t)q] = A.val[p];
                        if (src) (*src)[(size_t)q] = p;
                    }
            });
        parallel_tasks(tasks);
        return T;
    }
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
    // One pass, rows in parallel.  A row's products are accumulated in a small open-addressing table keyed by the output column (the
    // rows of Galerkin products hold tens of entries: the table lives in L1, where marker / accumulator arrays of B.nc entries per
    // thread cost more to allocate and clear than the whole product).  Per output entry the products are added in the order they are
    // met -- ascending k -- exactly as a dense accumulator would: same bits.  Each chunk of rows writes its own buffers, which are
    // then moved to their place in C.
    Csr C;
    C.nr = A.nr; C.nc = B.nc;
    C.ptr.assign((size_t)C.nr + 1, 0);
    struct Chunk { long r0 = 0, r1 = 0; std::vector<int> col; std::vector<double> val; };
    std::vector<Chunk> chunks;
    std::mutex chunks_m;
    parallel_for(A.nr, 2048, [&](long r0, long r1) {
        Chunk ch;
        ch.r0 = r0; ch.r1 = r1;
        std::vector<int> key(64, -1);
        std::vector<double> acc(64, 0.0);
        std::vector<uint64_t> used;      // (column << 32) | slot
        {
            // room for every product of the chunk (an upper bound of its entries; pages that stay unused are never touched): no regrowth
            long all = 0;
            for (long i = r0; i < r1; i++)
                for (int pa = A.ptr[i]; pa < A.ptr[i + 1]; pa++) all += B.ptr[(size_t)A.col[pa] + 1] - B.ptr[A.col[pa]];
            ch.col.reserve((size_t)all); ch.val.reserve((size_t)all);
        }
        for (long i = r0; i < r1; i++) {
            // (the rows of B a row of A asks for lie all over B: request them before they are needed -- the row pointers of the next
            //  row's, the entries of this row's -- or the product is one cache miss after the other)
            if (i + 1 < r1) for (int pa = A.ptr[i + 1]; pa < A.ptr[i + 2]; pa++) __builtin_prefetch(&B.ptr[A.col[pa]]);
            long upper = 0;
            for (int pa = A.ptr[i]; pa < A.ptr[i + 1]; pa++) {
                const int b0 = B.ptr[A.col[pa]], b1 = B.ptr[(size_t)A.col[pa] + 1];
                upper += b1 - b0;
                __builtin_prefetch(&B.col[b0]); __builtin_prefetch(&B.val[b0]);
                if (b1 - b0 > 8) { __builtin_prefetch(&B.col[b0] + 16); __builtin_prefetch(&B.val[b0] + 8); }
            }
            int lg = 4;
            while ((1L << lg) < 2 * upper) lg++;
            const uint32_t cap = 1u << lg, mask = cap - 1;
            if (key.size() < cap) { key.assign(cap, -1); acc.assign(cap, 0.0); }      // (all slots are free between rows)
            used.clear();
            for (int pa = A.ptr[i]; pa < A.ptr[i + 1]; pa++) {
                const int k = A.col[pa];
                const double a = A.val[pa];
                for (int pb = B.ptr[k]; pb < B.ptr[k + 1]; pb++) {
                    const int j = B.col[pb];
                    uint32_t s = ((uint32_t)j * 2654435761u) >> (32 - lg);
                    while (key[s] != j && key[s] != -1) s = (s + 1) & mask;
                    if (key[s] == -1) { key[s] = j; acc[s] = a * B.val[pb]; used.push_back(((uint64_t)(uint32_t)j << 32) | s); }
                    else acc[s] += a * B.val[pb];
                }
            }
            std::sort(used.begin(), used.end());
            for (uint64_t u : used) {
                const uint32_t s = (uint32_t)u;
                ch.col.push_back((int)(u >> 32));
                ch.val.push_back(acc[s]);
                key[s] = -1;
            }
            C.ptr[(size_t)i + 1] = (int)used.size();
        }
        std::lock_guard<std::mutex> g(chunks_m);
        chunks.push_back(std::move(ch));
    });
    for (int i = 0; i < C.nr; i++) C.ptr[(size_t)i + 1] += C.ptr[i];
    C.col.resize((size_t)C.ptr[C.nr]); C.val.resize((size_t)C.ptr[C.nr]);
    std::vector<std::function<void()>> moves;
    for (Chunk& ch : chunks)
        moves.push_back([&C, &ch] {
            std::copy(ch.col.begin(), ch.col.end(), C.col.begin() + C.ptr[(size_t)ch.r0]);
            std::copy(ch.val.begin(), ch.val.end(), C.val.begin() + C.ptr[(size_t)ch.r0]);
        });
    parallel_tasks(moves);
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
    // Rows in parallel: the terms of an output entry all come from its own row of A.  Where a column of B lands inside row i of C is
    // found by bisection (rows of C are sorted and short).
    auto entry_of = [&](long i, int j) {
        const int* b = C.col.data() + C.ptr[i];
        const int* e = C.col.data() + C.ptr[i + 1];
        return (long)(std::lower_bound(b, e, j) - C.col.data());
    };
    R.ptr.assign((size_t)C.nnz() + 1, 0);
    parallel_for(A.nr, 2048, [&](long r0, long r1) {
        for (long i = r0; i < r1; i++)
            for (int pa = A.ptr[i]; pa < A.ptr[i + 1]; pa++) {
                const int k = A.col[pa];
                for (int pb = B.ptr[k]; pb < B.ptr[k + 1]; pb++) R.ptr[(size_t)entry_of(i, B.col[pb]) + 1]++;
            }
    });
    for (long e = 0; e < C.nnz(); e++) R.ptr[(size_t)e + 1] += R.ptr[(size_t)e];
    R.idx.resize((size_t)R.ptr[(size_t)C.nnz()]);
    R.coef.resize((size_t)R.ptr[(size_t)C.nnz()]);
    parallel_for(A.nr, 2048, [&](long r0, long r1) {
        std::vector<int> next;      // write cursors of the entries of the row at hand
        for (long i = r0; i < r1; i++) {
            const long e0 = C.ptr[i];
            next.assign(R.ptr.begin() + e0, R.ptr.begin() + C.ptr[i + 1]);
            for (int pa = A.ptr[i]; pa < A.ptr[i + 1]; pa++) {  // ascending k: the accumulation order of spgemm()
                const int k = A.col[pa];
                for (int pb = B.ptr[k]; pb < B.ptr[k + 1]; pb++) {
                    const int t = next[(size_t)(entry_of(i, B.col[pb]) - e0)]++;
                    if (coef_from_A) { R.coef[(size_t)t] = A.val[pa]; R.idx[(size_t)t] = pb; }
                    else { R.coef[(size_t)t] = B.val[pb]; R.idx[(size_t)t] = pa; }
                }
            }
        }
    });
}

std::vector<double> diagonal(const Csr& A)
{
    std::vector<double> d(A.nr, 0.0);
    parallel_for(A.nr, 1 << 16, [&](long r0, long r1) {
        for (long i = r0; i < r1; i++)
            for (int p = A.ptr[i]; p < A.ptr[i + 1]; p++)
                if (A.col[p] == i) d[i] = A.val[p];
    });
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
