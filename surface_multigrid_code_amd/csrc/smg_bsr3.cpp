// smg_bsr3.cpp -- host side of the block (3-DOF) variant: block patterns, the Kronecker factor of P (x) I_3, the block SELL image.
// See smg_bsr3.hpp.
#include "smg_bsr3.hpp"

#include <algorithm>
#include <atomic>
#include <utility>

namespace smg {

Csr block_pattern3(const Csr& A)
{
    Csr G;
    const int nv = A.nr / 3;
    G.nr = nv; G.nc = A.nc / 3;
    G.ptr.assign((size_t)nv + 1, 0);
    // block row I = union of the block columns of scalar rows 3I .. 3I+2 (each ascending): a three-way merge per vertex
    std::vector<int> cnt((size_t)nv, 0);
    auto merge_row = [&](int I, int* out) {
        int p[3], e[3];
        for (int d = 0; d < 3; d++) { p[d] = A.ptr[3 * I + d]; e[d] = A.ptr[3 * I + d + 1]; }
        int n = 0, last = -1;
        while (true) {
            int best = -1;
            for (int d = 0; d < 3; d++) if (p[d] < e[d]) { const int c = A.col[p[d]] / 3; if (best < 0 || c < best) best = c; }
            if (best < 0) break;
            if (best != last) { if (out) out[n] = best; n++; last = best; }
            for (int d = 0; d < 3; d++) while (p[d] < e[d] && A.col[p[d]] / 3 == best) p[d]++;
        }
        return n;
    };
    parallel_for(nv, 4096, [&](long a, long b) { for (long I = a; I < b; I++) cnt[(size_t)I] = merge_row((int)I, nullptr); });
    for (int I = 0; I < nv; I++) G.ptr[(size_t)I + 1] = G.ptr[(size_t)I] + cnt[(size_t)I];
    G.col.resize((size_t)G.ptr[(size_t)nv]);
    G.val.assign(G.col.size(), 1.0);
    parallel_for(nv, 4096, [&](long a, long b) { for (long I = a; I < b; I++) merge_row((int)I, G.col.data() + G.ptr[(size_t)I]); });
    return G;
}

bool kron3_factor(const Csr& P, Csr& Pv)
{
    if (P.nr % 3 || P.nc % 3) return false;
    const int nr = P.nr / 3;
    std::atomic<int> bad{0};
    parallel_for(nr, 1 << 15, [&](long r0, long r1) {
        for (long r = r0; r < r1 && !bad.load(std::memory_order_relaxed); r++) {
            const int p0 = P.ptr[3 * r], n = P.ptr[3 * r + 1] - p0;
            for (int d = 0; d < 3; d++) {
                const int pd = P.ptr[3 * r + d];
                if (P.ptr[3 * r + d + 1] - pd != n) { bad.store(1, std::memory_order_relaxed); return; }
                for (int t = 0; t < n; t++) {
                    if (P.col[pd + t] % 3 != d || P.col[pd + t] / 3 != P.col[p0 + t] / 3) { bad.store(1, std::memory_order_relaxed); return; }
                    // the three copies must be the same BITS (the device applies one of them to all three components)
                    if (!(P.val[pd + t] == P.val[p0 + t]) && !(P.val[pd + t] != P.val[pd + t] && P.val[p0 + t] != P.val[p0 + t])) { bad.store(1, std::memory_order_relaxed); return; }
                }
            }
        }
    });
    if (bad.load()) return false;
    Pv = Csr();
    Pv.nr = nr; Pv.nc = P.nc / 3;
    Pv.ptr.resize((size_t)nr + 1);
    Pv.ptr[0] = 0;
    for (int r = 0; r < nr; r++) Pv.ptr[(size_t)r + 1] = Pv.ptr[(size_t)r] + (P.ptr[3 * (size_t)r + 1] - P.ptr[3 * (size_t)r]);
    Pv.col.resize((size_t)Pv.ptr[(size_t)nr]); Pv.val.resize((size_t)Pv.ptr[(size_t)nr]);
    parallel_for(nr, 1 << 15, [&](long r0, long r1) {
        for (long r = r0; r < r1; r++) {
            int o = Pv.ptr[(size_t)r];
            for (int p = P.ptr[3 * r]; p < P.ptr[3 * r + 1]; p++, o++) { Pv.col[(size_t)o] = P.col[p] / 3; Pv.val[(size_t)o] = P.val[p]; }
        }
    });
    return true;
}

Csr kron3(const Csr& P)
{
    Csr B;
    B.nr = 3 * P.nr; B.nc = 3 * P.nc;
    B.ptr.resize((size_t)B.nr + 1);
    B.col.resize((size_t)3 * P.nnz()); B.val.resize((size_t)3 * P.nnz());
    parallel_for(P.nr, 1 << 15, [&](long r0, long r1) {
        for (long r = r0; r < r1; r++) {
            const int len = P.ptr[r + 1] - P.ptr[r];
            for (int d = 0; d < 3; d++) {   // row 3r+d holds P(r,c) at column 3c+d  (reference src/get_prolong.cpp:108-110)
                int q = 3 * P.ptr[r] + d * len;
                B.ptr[(size_t)3 * r + d] = q;
                for (int p = P.ptr[r]; p < P.ptr[r + 1]; p++, q++) { B.col[(size_t)q] = 3 * P.col[p] + d; B.val[(size_t)q] = P.val[p]; }
            }
        }
    });
    B.ptr[(size_t)B.nr] = (int)(3 * P.nnz());
    return B;
}

Bsr3Sell bsr3_layout(const std::vector<int>& row_len, const std::vector<int>* vertex_breaks, bool region_order, long nnz_scalar, long n_blocks)
{
    constexpr int C = 64;
    Bsr3Sell S;
    const int nv = (int)row_len.size();
    S.n_vert = nv;
    S.nnz_scalar = nnz_scalar;
    S.n_blocks = n_blocks;
    std::vector<int> breaks;
    if (vertex_breaks) breaks = *vertex_breaks; else breaks = {0, nv};
    S.slice_row.push_back(0);
    S.color_slice_ptr.push_back(0);
    for (size_t c = 0; c + 1 < breaks.size(); c++) {
        for (int r0 = breaks[c]; r0 < breaks[c + 1]; r0 += C) {
            const int r1 = std::min(r0 + C, breaks[c + 1]);
            int w = 0;
            for (int r = r0; r < r1; r++) w = std::max(w, row_len[(size_t)r]);
            S.slice_row.push_back(r1);
            S.slice_w.push_back(w);
            S.w_max = std::max(S.w_max, w);
        }
        S.color_slice_ptr.push_back((int)S.slice_row.size() - 1);
    }
    S.n_slices = (int)S.slice_row.size() - 1;
    S.slice_off.assign((size_t)S.n_slices + 1, 0);
    for (int s = 0; s < S.n_slices; s++) S.slice_off[(size_t)s + 1] = S.slice_off[(size_t)s] + S.slice_w[(size_t)s];
    if (region_order && S.color_slice_ptr.size() > 2) {
        std::vector<std::pair<double, int>> key((size_t)S.n_slices);
        for (size_t c = 0; c + 1 < S.color_slice_ptr.size(); c++) {
            const int b = S.color_slice_ptr[c], e = S.color_slice_ptr[c + 1];
            for (int s = b; s < e; s++) key[(size_t)s] = {(s - b + 0.5) / (double)(e - b), s};
        }
        std::stable_sort(key.begin(), key.end(), [](const std::pair<double, int>& x, const std::pair<double, int>& y) { return x.first < y.first; });
        S.region_order.resize((size_t)S.n_slices);
        for (int i = 0; i < S.n_slices; i++) S.region_order[(size_t)i] = key[(size_t)i].second;
    }
    return S;
}

Bsr3Sell build_bsr3(const Csr& A, const std::vector<int>* vertex_breaks, bool region_order, bool with_entry)
{
    constexpr int C = 64;
    const int nv = A.nr / 3;
    const Csr G = block_pattern3(A);
    std::vector<int> row_len((size_t)nv);
    for (int r = 0; r < nv; r++) row_len[(size_t)r] = G.ptr[(size_t)r + 1] - G.ptr[(size_t)r];
    Bsr3Sell S = bsr3_layout(row_len, vertex_breaks, region_order, A.nnz(), G.nnz());
    const size_t cols = (size_t)C * (size_t)S.slice_off.back();
    // (sized without initialisation: every slice clears and fills its own part, so the hundreds of MB are first touched by many threads)
    S.col.resize(cols);
    S.val.resize(cols * 9);
    if (with_entry) S.entry.resize(cols * 9);
    parallel_for(S.n_slices, 256, [&](long s0, long s1) {
        for (long s = s0; s < s1; s++) {
            const size_t off = (size_t)S.slice_off[(size_t)s];
            {
                const size_t c0 = off * C, c1 = (size_t)S.slice_off[(size_t)s + 1] * C;
                std::fill(S.col.begin() + c0, S.col.begin() + c1, -1);
                std::fill(S.val.begin() + c0 * 9, S.val.begin() + c1 * 9, 0.0);
                if (with_entry) std::fill(S.entry.begin() + c0 * 9, S.entry.begin() + c1 * 9, -1);
            }
            for (int r = S.slice_row[(size_t)s]; r < S.slice_row[(size_t)s + 1]; r++) {
                const int lane = r - S.slice_row[(size_t)s];
                const int g0 = G.ptr[(size_t)r], gw = G.ptr[(size_t)r + 1] - g0;
                for (int j = 0; j < gw; j++) S.col[(off + (size_t)j) * C + lane] = G.col[(size_t)g0 + j];
                for (int d = 0; d < 3; d++) {
                    int j = 0;   // block columns ascend with the scalar columns: one forward walk per row
                    for (int p = A.ptr[(size_t)3 * r + d]; p < A.ptr[(size_t)3 * r + d + 1]; p++) {
                        const int J = A.col[p] / 3, e = A.col[p] % 3;
                        while (G.col[(size_t)g0 + j] != J) j++;
                        const size_t at = ((off + (size_t)j) * 9 + (size_t)(3 * d + e)) * C + lane;
                        S.val[at] = A.val[p];
                        if (with_entry) S.entry[at] = p;
                    }
                }
            }
        }
    });
    return S;
}

}  // namespace smg
