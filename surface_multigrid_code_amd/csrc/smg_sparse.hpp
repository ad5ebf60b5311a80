// smg_sparse.hpp -- host-side sparse containers and kernels of the product library (libsmg).
//
// CSR, int32 indices, fp64 values: the row-major twin of the reference's Eigen::SparseMatrix<double>
// (column-major, int32) that mg_data carries (reference src/mg_data.h:13-19).  For the symmetric system
// matrices CSR == CSC; P and PT are both kept explicitly, as the reference does.
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <functional>
#include <memory>
#include <type_traits>
#include <utility>
#include <vector>

namespace smg {

// The arrays of a Csr: std::vector with an allocator whose resize(n) leaves new elements UNINITIALISED.  The big ones (tens of MB at a
// million rows) are always filled right after they are sized, by several threads: with the usual value-initialisation one thread would
// first touch -- page-fault in -- all of it, which costs more than the fill (input copy of C3: 20 ms of 23).  resize(n, v) / assign(n, v)
// still initialise.
template <class T>
struct NoInitAlloc : std::allocator<T> {
    template <class U> struct rebind { using other = NoInitAlloc<U>; };
    NoInitAlloc() = default;
    template <class U> NoInitAlloc(const NoInitAlloc<U>&) noexcept {}
    template <class U> void construct(U* p) noexcept(std::is_nothrow_default_constructible<U>::value) { ::new (static_cast<void*>(p)) U; }
    template <class U, class... Args> void construct(U* p, Args&&... args) { ::new (static_cast<void*>(p)) U(std::forward<Args>(args)...); }
};
template <class T> using raw_vector = std::vector<T, NoInitAlloc<T>>;

struct Csr {
    int nr = 0, nc = 0;
    raw_vector<int> ptr;     // nr + 1
    raw_vector<int> col;     // nnz, ascending inside a row
    raw_vector<double> val;  // nnz
    long nnz() const { return ptr.empty() ? 0 : (long)ptr.back(); }
    bool empty() const { return nr == 0 && nc == 0; }
};

// Build from raw arrays; sorts every row by column and sums duplicate (row,col) pairs
// (Eigen setFromTriplets semantics).  Explicit zeros are kept.
Csr csr_from_arrays(int nr, int nc, const int* ptr, const int* col, const double* val);
bool rows_strictly_ascending(int nr, const int* ptr, const int* col);   // every row sorted, no duplicates: the arrays are canonical already
// Interpret (ptr,idx,val) as compressed *columns* of an nr x nc matrix and return its CSR.
Csr csr_from_csc_arrays(int nr, int nc, const int* colptr, const int* rowidx, const double* val);

Csr copy_of(const Csr& A);   // a copy made by several threads
Csr transpose(const Csr& A, std::vector<int>* src = nullptr);  // src[e]: index in A.val of output entry e

// C = A * B.  Row i of C accumulates  A(i,k) * B(k,:)  over the stored k of row i in ascending order;
// the first touch of an output entry assigns.  Entry-wise this is the same sequence of additions as
// Eigen's column-major product (ascending k), so Galerkin operators are bit-identical to the
// reference's  PT * A * P  (reference src/min_quad_with_fixed_mg.cpp:25, :227).  No pruning.
Csr spgemm(const Csr& A, const Csr& B);

// Y(i,j) = X(rows[i], cols[j])  (igl::slice).  A null pointer means "all, in order".  Indices unique.
Csr slice(const Csr& X, const std::vector<int>* rows, const std::vector<int>* cols, std::vector<int>* src = nullptr);

// B(i,j) = A(rperm[i], cperm[j])  with both lists permutations (new -> old).
Csr permute(const Csr& A, const std::vector<int>& rperm, const std::vector<int>& cperm, std::vector<int>* src = nullptr);

std::vector<double> diagonal(const Csr& A);

// Numeric phase of C = A * B for a FIXED sparsity: out[e] = sum_t coef[t] * src[idx[t]], t in [ptr[e], ptr[e+1]), the terms
// of an output entry listed in ascending k (the order spgemm accumulates them in), so replaying the recipe with new
// values of the variable factor reproduces spgemm bit for bit.  coef_from_A: A is the constant factor (its values are
// baked into coef, idx points into B.val); otherwise B is constant and idx points into A.val.
struct Recipe {
    std::vector<int> ptr;
    raw_vector<int> idx;
    raw_vector<double> coef;
};
void spgemm_recipe(const Csr& A, const Csr& B, bool coef_from_A, const Csr& C, Recipe& R);

// Host parallelism for the precompute (the reference's precompute is single-threaded Eigen; here it must not dwarf a solve
// that takes milliseconds).  fn(begin, end) is called on disjoint chunks of [0, n) from up to host_threads() threads; chunks are
// contiguous and their outputs position-independent, so every result is identical to the sequential one.
int host_threads();   // SMG_HOST_THREADS, default min(hardware threads, 32)
void parallel_for(long n, long grain, const std::function<void(long, long)>& fn);
// run independent tasks concurrently (each may itself call parallel_for: nested calls run inline)
void parallel_tasks(const std::vector<std::function<void()>>& tasks);

// std::sort of a big array on the host threads: chunks sorted side by side, then merged pairwise (log2(chunks) passes of std::inplace_merge, the pairs of a pass
// side by side).  cmp must be a strict total order for the result to be unique (it is for every caller: ties are broken by ids) -- then the outcome is
// std::sort's, whatever the number of threads.
template <class T, class Cmp>
void parallel_sort(std::vector<T>& v, Cmp cmp)
{
    const size_t n = v.size();
    int chunks = 1;
    while (chunks * 2 <= host_threads() && chunks < 64 && n / (size_t)(chunks * 2) >= (size_t)1 << 15) chunks *= 2;
    if (chunks == 1) { std::sort(v.begin(), v.end(), cmp); return; }
    std::vector<size_t> cut((size_t)chunks + 1);
    for (int c = 0; c <= chunks; c++) cut[(size_t)c] = n * (size_t)c / (size_t)chunks;
    {
        std::vector<std::function<void()>> tasks;
        for (int c = 0; c < chunks; c++) tasks.push_back([&, c] { std::sort(v.begin() + (long)cut[(size_t)c], v.begin() + (long)cut[(size_t)c + 1], cmp); });
        parallel_tasks(tasks);
    }
    for (int w = 1; w < chunks; w *= 2) {
        std::vector<std::function<void()>> tasks;
        for (int c = 0; c + w < chunks; c += 2 * w)
            tasks.push_back([&, c, w] { std::inplace_merge(v.begin() + (long)cut[(size_t)c], v.begin() + (long)cut[(size_t)c + w], v.begin() + (long)cut[(size_t)std::min(c + 2 * w, chunks)], cmp); });
        parallel_tasks(tasks);
    }
}

// y = A x for dense column-major blocks (host; used only by precompute-time checks and tools)
void spmv_host(const Csr& A, const double* x, double* y);

}  // namespace smg
