// smg_order.cpp -- see smg_order.hpp.
#include "smg_order.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstdint>
#include <numeric>
#include <queue>
#include <utility>

namespace smg {

std::vector<int> rcm_order(const Csr& A) { return rcm_order_arrays(A.nr, A.ptr.data(), A.col.data()); }

std::vector<int> rcm_order_arrays(int n, const int* Aptr, const int* Acol)
{
    std::vector<int> order;
    order.reserve(n);
    std::vector<char> seen(n, 0);
    std::vector<int> deg(n);
    for (int i = 0; i < n; i++) deg[i] = Aptr[i + 1] - Aptr[i];
    // component seeds in ascending degree (cheap stand-in for a pseudo-peripheral search)
    // (a counting sort: the same sequence a stable sort by degree gives, without its 1 M-element merge passes)
    std::vector<int> seeds(n);
    {
        int dmax = 0;
        for (int i = 0; i < n; i++) dmax = std::max(dmax, deg[i]);
        std::vector<int> start((size_t)dmax + 2, 0);
        for (int i = 0; i < n; i++) start[(size_t)deg[i] + 1]++;
        for (int d = 0; d <= dmax; d++) start[(size_t)d + 1] += start[(size_t)d];
        for (int i = 0; i < n; i++) seeds[(size_t)start[(size_t)deg[i]]++] = i;
    }
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
                // the queue is known a few vertices ahead: ask for their rows now (the search is a chain of cache misses otherwise)
                if (head + 8 < order.size()) __builtin_prefetch(&Aptr[order[head + 8]]);
                if (head + 4 < order.size()) __builtin_prefetch(&Acol[Aptr[order[head + 4]]]);
                int v = order[head++];
                nb.clear();
                for (int p = Aptr[v]; p < Aptr[v + 1]; p++) {
                    int w = Acol[p];
                    if (w != v && !seen[w]) { seen[w] = 1; nb.push_back(w); }
                }
                // (the first round only looks for a far vertex: any breadth-first order ends in the last level)
                if (round == 1) std::sort(nb.begin(), nb.end(), [&](int a, int b) { return deg[a] != deg[b] ? deg[a] < deg[b] : a < b; });
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

// ---- colouring -------------------------------------------------------------------------------------------
// Every colour is one kernel launch per Gauss-Seidel sweep, on every level, and small levels are pure launch
// latency: fewer colours = fewer launches.  Triangle-mesh graphs are 4-colourable in principle; plain first-fit
// in BFS order gives 5-6 with one or two almost empty classes.  Pipeline: DSATUR, then try to dissolve the
// smallest classes by local recolouring, then iterated greedy (Culberson) which can only lower the count.

static int count_colors(const std::vector<int>& color)
{
    int nc = 0;
    for (int c : color) nc = std::max(nc, c + 1);
    return nc;
}

// first-fit greedy over a given visiting order; returns the number of colours
static int greedy_color(const Csr& A, const std::vector<int>& order, std::vector<int>& color)
{
    const int n = A.nr;
    color.assign(n, -1);
    std::vector<int> forbid;
    int ncol = 0;
    for (int t = 0; t < n; t++) {
        const int v = order[t];
        for (int p = A.ptr[v]; p < A.ptr[v + 1]; p++) {
            const int w = A.col[p];
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
    return ncol;
}

// DSATUR (Brelaz): always colour the vertex that sees the most distinct colours; ties by degree, then index.
static int dsatur_color(const Csr& A, std::vector<int>& color)
{
    const int n = A.nr;
    color.assign(n, -1);
    std::vector<uint64_t> seen(n, 0);  // bit c set <=> a neighbour has colour c (c < 64)
    std::vector<int> sat(n, 0), deg(n);
    for (int i = 0; i < n; i++) deg[i] = A.ptr[i + 1] - A.ptr[i];
    struct E { int sat, deg, v; bool operator<(const E& o) const { return sat != o.sat ? sat < o.sat : (deg != o.deg ? deg < o.deg : v > o.v); } };
    std::priority_queue<E> pq;
    for (int i = 0; i < n; i++) pq.push({0, deg[i], i});
    int ncol = 0;
    while (!pq.empty()) {
        const E e = pq.top();
        pq.pop();
        const int v = e.v;
        if (color[v] >= 0 || e.sat != sat[v]) continue;
        int c = 0;
        while (c < 64 && ((seen[v] >> c) & 1)) c++;
        if (c >= 63) return INT_MAX;   // the 64-bit colour sets of this routine are exhausted: leave dense graphs to plain greedy
        color[v] = c;
        ncol = std::max(ncol, c + 1);
        for (int p = A.ptr[v]; p < A.ptr[v + 1]; p++) {
            const int w = A.col[p];
            if (w == v || color[w] >= 0) continue;
            if (!((seen[w] >> c) & 1)) { seen[w] |= (1ull << c); sat[w]++; pq.push({sat[w], deg[w], w}); }
        }
    }
    return ncol;
}

// Try to empty the highest class K.  Each of its vertices v is moved to a lower colour, by (1) a free colour,
// (2) first moving the neighbours that hold some colour c to another free colour, (3) a Kempe-chain swap a<->b on
// the (a,b)-components hanging off v's a-neighbours when those do not reach a b-neighbour of v (component size
// capped, total work budgeted), (4) for the last stragglers an exhaustive re-colouring of a small ball around v.
// Every step keeps the colouring valid; returns true when class K ended up empty.
struct Recolor {
    const Csr& A;
    std::vector<int>& color;
    int K;
    std::vector<char> mark;
    long budget;
    long ball_budget;   // elementary steps all ball searches of this object may take together
    Recolor(const Csr& A_, std::vector<int>& c, int K_) : A(A_), color(c), K(K_), mark(A_.nr, 0), budget(40L * A_.nr + 100000), ball_budget(400L * A_.nr + 20000000L) {}

    int free_color(int u, int avoid) const
    {
        uint64_t used = 0;
        for (int p = A.ptr[u]; p < A.ptr[u + 1]; p++) { const int w = A.col[p]; if (w != u && color[w] >= 0) used |= 1ull << color[w]; }
        for (int c = 0; c < K; c++) if (c != avoid && !((used >> c) & 1)) return c;
        return -1;
    }
    bool exchange(int v)
    {
        for (int c = 0; c < K; c++) {
            std::vector<std::pair<int, int>> undo;
            bool ok = true;
            for (int p = A.ptr[v]; p < A.ptr[v + 1] && ok; p++) {
                const int u = A.col[p];
                if (u == v || color[u] != c) continue;
                const int d = free_color(u, c);
                if (d < 0) ok = false;
                else { undo.emplace_back(u, color[u]); color[u] = d; }
            }
            if (ok) { color[v] = c; return true; }
            for (auto it = undo.rbegin(); it != undo.rend(); ++it) color[it->first] = it->second;
        }
        return false;
    }
    bool kempe(int v, int cap)
    {
        std::vector<int> comp, stack;
        for (int a = 0; a < K; a++)
            for (int b = 0; b < K; b++) {
                if (a == b || budget <= 0) continue;
                comp.clear(); stack.clear();
                bool blocked = false;
                auto push = [&](int u) { if (!mark[u]) { mark[u] = 1; comp.push_back(u); stack.push_back(u); } };
                for (int p = A.ptr[v]; p < A.ptr[v + 1]; p++) { const int w = A.col[p]; if (w != v && color[w] == a) push(w); }
                while (!stack.empty() && !blocked) {
                    const int u = stack.back();
                    stack.pop_back();
                    for (int p = A.ptr[u]; p < A.ptr[u + 1]; p++) {
                        const int w = A.col[p];
                        if (w == u || w == v || (color[w] != a && color[w] != b)) continue;
                        push(w);
                    }
                    if ((int)comp.size() > cap) blocked = true;
                }
                budget -= (long)comp.size() * 8;
                if (!blocked)
                    for (int p = A.ptr[v]; p < A.ptr[v + 1]; p++) { const int w = A.col[p]; if (w != v && color[w] == b && mark[w]) { blocked = true; break; } }
                if (!blocked) for (int u : comp) color[u] = (color[u] == a) ? b : a;
                for (int u : comp) mark[u] = 0;
                if (!blocked) { color[v] = a; return true; }
            }
        return false;
    }
    // exhaustive K-colouring of the ball of given radius around v, colours outside the ball fixed
    bool ball(int v, int radius, long node_limit)
    {
        std::vector<int> B{v}, dist{0};
        mark[v] = 1;
        for (size_t h = 0; h < B.size(); h++) {
            if (dist[h] == radius) continue;
            const int u = B[h];
            for (int p = A.ptr[u]; p < A.ptr[u + 1]; p++) { const int w = A.col[p]; if (!mark[w]) { mark[w] = 1; B.push_back(w); dist.push_back(dist[h] + 1); } }
        }
        // the search below costs O(|ball| * degree) per node: it is meant for mesh neighbourhoods (a few dozen vertices), not for
        // the balls of an expander, which hold most of the graph after three hops
        if (B.size() > 192 || ball_budget <= 0) { for (int w : B) mark[w] = 0; return false; }
        std::vector<int> saved(B.size());
        for (size_t i = 0; i < B.size(); i++) { saved[i] = color[B[i]]; color[B[i]] = -1; }
        long nodes = 0;
        std::vector<int> stackv, stackc;  // iterative DFS: chosen vertex, next colour to try
        auto avail = [&](int u) { uint64_t used = 0; for (int p = A.ptr[u]; p < A.ptr[u + 1]; p++) { const int w = A.col[p]; if (w != u && color[w] >= 0) used |= 1ull << color[w]; } return (~used) & ((1ull << K) - 1); };
        auto pick = [&]() {  // most constrained uncoloured ball vertex
            int best = -1, bc = 99;
            for (int u : B) if (color[u] < 0) { const int c = __builtin_popcountll(avail(u)); if (c < bc) { bc = c; best = u; } }
            return best;
        };
        bool ok = false;
        int u = pick();
        stackv.push_back(u); stackc.push_back(0);
        while (!stackv.empty()) {
            if (++nodes > node_limit) break;
            ball_budget -= (long)B.size();
            if (ball_budget <= 0) break;
            const int cu = stackv.back();
            int& next = stackc.back();
            color[cu] = -1;
            const uint64_t av = avail(cu);
            int c = next;
            while (c < K && !((av >> c) & 1)) c++;
            if (c >= K) { stackv.pop_back(); stackc.pop_back(); continue; }
            next = c + 1;
            color[cu] = c;
            const int nu = pick();
            if (nu < 0) { ok = true; break; }
            stackv.push_back(nu); stackc.push_back(0);
        }
        if (!ok) for (size_t i = 0; i < B.size(); i++) color[B[i]] = saved[i];
        for (int w : B) mark[w] = 0;
        return ok;
    }
    // The stragglers' last resort: a random walk of Kempe interchanges (the heuristic of Morgenstern & Shapiro for planar graphs).  The vertex
    // stays uncoloured while a two-colour component hanging off one of its neighbours is swapped -- the rest stays properly coloured whatever the
    // component -- until one of the K colours is missing around it.  kempe() above only takes the swaps that free a colour at once; the walk takes
    // those that do not, and gets there a few dozen swaps later.  Sequential, fixed-seed generator: the colouring depends on the matrix alone.
    uint64_t rng_state = 0x9E3779B97F4A7C15ull;
    uint32_t rnd(uint32_t m) { rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)((rng_state >> 33) % m); }
    long walk_budget = 0;
    bool walk(int v, int max_swaps, int cap)
    {
        std::vector<int> comp, stack, cand;
        for (int it = 0; it < max_swaps && walk_budget > 0; it++) {
            const int d = free_color(v, -1);
            if (d >= 0) { color[v] = d; return true; }
            const int a = (int)rnd((uint32_t)K);
            int b = (int)rnd((uint32_t)(K - 1));
            if (b >= a) b++;
            cand.clear();
            for (int p = A.ptr[v]; p < A.ptr[v + 1]; p++) { const int w = A.col[p]; if (w != v && color[w] == a) cand.push_back(w); }
            walk_budget -= A.ptr[v + 1] - A.ptr[v];
            if (cand.empty()) continue;
            const int w0 = cand[rnd((uint32_t)cand.size())];
            comp.clear(); stack.clear();
            mark[w0] = 1; comp.push_back(w0); stack.push_back(w0);
            bool too_big = false;
            while (!stack.empty()) {
                const int u = stack.back();
                stack.pop_back();
                for (int p = A.ptr[u]; p < A.ptr[u + 1]; p++) {
                    const int w = A.col[p];
                    if (w == u || w == v || mark[w] || (color[w] != a && color[w] != b)) continue;
                    mark[w] = 1; comp.push_back(w); stack.push_back(w);
                }
                if ((int)comp.size() > cap) { too_big = true; break; }
            }
            walk_budget -= 8L * (long)comp.size();
            if (!too_big) for (int u : comp) color[u] = (color[u] == a) ? b : a;
            for (int u : comp) mark[u] = 0;
        }
        const int d = free_color(v, -1);
        if (d >= 0) { color[v] = d; return true; }
        return false;
    }
    bool run()
    {
        const bool tm_on = std::getenv("SMG_TIMING_COLOR") != nullptr;
        auto t_last = std::chrono::steady_clock::now();
        auto lap = [&](const char* what, size_t left_now) { if (!tm_on) return; auto t = std::chrono::steady_clock::now(); std::fprintf(stderr, "[colour]     K %d: %-34s %7.1f ms  (%zu left)\n", K, what, 1e3 * std::chrono::duration<double>(t - t_last).count(), left_now); t_last = t; };
        std::vector<int> todo, left;
        for (int v = 0; v < A.nr; v++) if (color[v] == K) todo.push_back(v);
        for (int v : todo) {
            const int d = free_color(v, -1);
            if (d >= 0) { color[v] = d; continue; }
            if (exchange(v)) continue;
            left.push_back(v);
        }
        lap("free colour / exchange", left.size());
        // Kempe passes with growing component caps: the expensive caps only ever see the few survivors
        // small graphs (coarse levels) get an all-out search: a clean 4-colouring there is inherited by every finer
        // subdivision level for free (subdivision_colors), and costs little in absolute terms
        const bool small = A.nr <= 70000 && K == 4;   // only the 5 -> 4 step is worth an all-out search (the 4 -> 3 step of a regular mesh gets the cheap passes: 0.39 s on a 51 200-row torus otherwise)
        const int caps_small[] = {256, 2048, 16384, 1 << 30};
        const int caps_big[] = {256, 2048, 16384};
        const int* caps = small ? caps_small : caps_big;
        const int ncaps = small ? 4 : 3;
        // (on a big mesh level the walk below does the cheap Kempe pass's work at a third of its price: 1 011 330 rows, 5 054 stragglers: pass with cap 256
        //  0.36 s for 3 064 of them + walk 0.17 s for the rest, against the walk alone 0.40 s)
        const bool walk_first = K == 4 && A.nr > 100000;
        auto balls = [&]() {
            todo.swap(left); left.clear();
            for (int v : todo) {
                if (color[v] != K) continue;
                const int d = free_color(v, -1);
                if (d >= 0) { color[v] = d; continue; }
                bool ok = false;
                for (int rad = 2; rad <= (small ? 5 : 3) && !ok; rad++) ok = ball(v, rad, 8000L * rad);
                if (!ok) left.push_back(v);
            }
            lap("ball searches", left.size());
        };
        for (int ci = 0; ci < ncaps; ci++) {
            const int cap = caps[ci];
            if (left.empty()) break;
            todo.swap(left); left.clear();
            budget = (small ? 600L : 40L) * A.nr + 1000000;
            for (int v : todo) {
                if (color[v] != K) continue;
                const int d = free_color(v, -1);   // earlier swaps may have freed a colour
                if (d >= 0) { color[v] = d; continue; }
                if (exchange(v)) continue;
                if (budget > 0 && !(ci == 0 && walk_first) && kempe(v, cap)) continue;
                left.push_back(v);
            }
            if (tm_on) { char nm[64]; std::snprintf(nm, sizeof nm, "Kempe pass, cap %d", cap); lap(nm, left.size()); }
            if (ci > 0) continue;
            // after the cheapest Kempe pass the survivors first get the small exhaustive ball searches: on meshes those settle
            // nearly all of them in about a millisecond, where the passes with large component caps walk half the graph per
            // attempt (C3's 15.8 k-vertex level: 18 stragglers, 137 ms of large-cap passes without a single success)
            if (left.size() <= 256) balls();
            // ... and then the random walk, BEFORE the passes with large caps (1 011 330-row mesh level: 1 990 stragglers after the pass with cap 256;
            // caps 2 048 and 16 384 settled 197 and 4 of them in 0.38 + 0.45 s, the walk all 1 789 that were left in 0.17 s).
            // Only the 5 -> 4 step (where a colour is a launch per sweep; the wide Galerkin levels of decimated hierarchies are swept piece-wise), or a
            // class of a handful of rows.  Component caps grow like those of the Kempe passes: small components are cheap to swap and settle nearly
            // everybody (252 834-row mesh level: 5 542 stragglers -> 17 at cap 128), the few survivors get the big ones.
            walk_budget = 2000L * A.nr + 4000000L;
            if (!left.empty() && (K == 4 || left.size() <= 64)) {
                const int caps_walk[] = {128, 512, 2048, 8192, 32768, 1 << 30};
                for (int wi = 0; wi < 6 && !left.empty() && walk_budget > 0; wi++) {
                    if (caps_walk[wi] > 32768 && A.nr > 70000) break;
                    todo.swap(left); left.clear();
                    for (int v : todo) {
                        if (color[v] != K) continue;
                        if (walk_budget <= 0 || !walk(v, 64, caps_walk[wi])) left.push_back(v);
                    }
                }
                lap("random walk of Kempe interchanges", left.size());
            }
        }
        if (!left.empty() && left.size() <= 256) balls();
        for (int v = 0; v < A.nr; v++) if (color[v] == K) return false;
        return true;
    }
};

static bool dissolve_top_class(const Csr& A, std::vector<int>& color, int K)
{
    Recolor r(A, color, K);
    return r.run();
}

static void compact_colors(std::vector<int>& color)
{
    // renumber classes by decreasing size (largest first: big launches first, stragglers last)
    const int nc = count_colors(color);
    std::vector<long> size(nc, 0);
    for (int c : color) size[c]++;
    std::vector<int> idx(nc);
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return size[a] > size[b]; });
    std::vector<int> remap(nc, -1);
    int k = 0;
    for (int c : idx) if (size[c] > 0) remap[c] = k++;
    for (int& c : color) c = remap[c];
}

// A vertex whose neighbours form closed cycles (every neighbour has exactly two neighbours inside the neighbourhood: a closed
// fan of triangles) of odd total length sits at the hub of an odd wheel, which needs 4 colours.  Almost every triangle mesh has
// one (any interior vertex of odd valence), and then trying to dissolve the fourth class is a search that cannot succeed.
static bool has_odd_wheel(const Csr& A)
{
    std::vector<char> in(A.nr, 0);
    for (int v = 0; v < A.nr; v++) {
        int d = 0;
        for (int p = A.ptr[v]; p < A.ptr[v + 1]; p++) if (A.col[p] != v) { in[A.col[p]] = 1; d++; }
        bool wheel = (d % 2 == 1) && d >= 3;
        if (wheel)
            for (int p = A.ptr[v]; p < A.ptr[v + 1] && wheel; p++) {
                const int u = A.col[p];
                if (u == v) continue;
                int ring = 0;
                for (int q = A.ptr[u]; q < A.ptr[u + 1]; q++) if (A.col[q] != u && A.col[q] != v && in[A.col[q]]) ring++;
                if (ring != 2) wheel = false;
            }
        for (int p = A.ptr[v]; p < A.ptr[v + 1]; p++) in[A.col[p]] = 0;
        if (wheel) return true;
    }
    return false;
}

static bool coloring_is_valid(const Csr& A, const std::vector<int>& color)
{
    for (int i = 0; i < A.nr; i++)
        for (int p = A.ptr[i]; p < A.ptr[i + 1]; p++)
            if (A.col[p] != i && color[A.col[p]] == color[i]) return false;
    return true;
}

static std::vector<int> color_graph(const Csr& A, const std::vector<int>& rcm)
{
    const bool tm_on = std::getenv("SMG_TIMING_COLOR") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what, int nc) { if (!tm_on) return; auto t = std::chrono::steady_clock::now(); std::fprintf(stderr, "[colour] %-28s %7.1f ms  (%d colours)\n", what, 1e3 * std::chrono::duration<double>(t - t_last).count(), nc); t_last = t; };
    std::vector<int> best, cur;
    int nbest = greedy_color(A, rcm, best);
    if (A.nr == 0) return best;
    // The refinement below works with 64-bit colour sets and local searches whose cost grows with the degree: it is for
    // mesh-like graphs.  Dense or high-chromatic graphs (Galerkin operators of aggressive aggregations, say) keep the plain
    // first-fit colouring, which is always valid.
    if (nbest > 32 || A.nnz() > 64L * A.nr) return best;
    const std::vector<int> fallback = best;
    lap("greedy", nbest);
    // Big graphs of high degree (the scalar image of a 3-DOF system: 21 neighbours per row, 3 M rows) get first-fit plus one round of
    // iterated greedy only: DSATUR and the class-dissolving searches are sequential, cost seconds there (13 s + 3 s at 3 M rows -- of the
    // 18 s that first precompute took in round 3) and found nothing iterated greedy had not (15 colours either way).  Mesh levels
    // (7 per row) and the small, wide Galerkin levels of decimated hierarchies, where a colour less is a launch less per sweep, keep
    // the full treatment.
    const bool heavy = A.nr > 100000 && A.nnz() > 12L * A.nr;
    // (DSATUR is a sequential search with a priority queue: a second on a million rows, where it found nothing first-fit + iterated greedy + the
    //  class-dissolving passes below do not -- 6 colours, then 5, then 4 either way on the 1 011 330-row mesh level; below that size it is cheap and
    //  lowers the count the later passes start from on the wide Galerkin levels.)
    if (!heavy && A.nr <= 400000) {
        if (dsatur_color(A, cur) <= nbest) best = cur;
        compact_colors(best);
        nbest = count_colors(best);
        lap("dsatur", nbest);
    }
    // two rounds of iterated greedy (Culberson): revisit class by class, can only lower the count
    for (int it = 0; it < (heavy ? 1 : 2); it++) {
        const int nc = count_colors(best);
        std::vector<std::vector<int>> cls(nc);
        for (int t = 0; t < A.nr; t++) cls[best[rcm[t]]].push_back(rcm[t]);
        std::vector<int> order;
        order.reserve(A.nr);
        for (int c = nc - 1; c >= 0; c--) order.insert(order.end(), cls[c].begin(), cls[c].end());
        if (greedy_color(A, order, cur) <= nc) { best = cur; compact_colors(best); }
    }
    nbest = count_colors(best);
    lap("iterated greedy", nbest);
    if (heavy) return coloring_is_valid(A, best) ? best : fallback;
    // Wide graphs of any size (the Galerkin levels of the reference's own hierarchies: 18 - 30 entries per row, 10 - 15 colours) stop here too: such a
    // level is swept piece-wise (csrc/smg_wgs.hpp), where a colour is not a launch, and the class-dissolving searches below cost more than everything
    // else in the first precompute of a small mesh (ogre.obj: 47 of 96 ms for taking the 5 038-row level from 11 colours to 10).
    if (A.nnz() > 12L * A.nr) return coloring_is_valid(A, best) ? best : fallback;
    // dissolve the smallest class while that succeeds (3 colours is the floor for any mesh with a triangle, 4 with an odd wheel)
    const int floor_colors = (nbest > 3 && has_odd_wheel(A)) ? 4 : 3;
    for (int guard = 0; guard < 6 && nbest > floor_colors; guard++) {
        // below four colours only small graphs are searched: a mesh without an odd wheel (a regular torus) may be 3-colourable, but two of three colours
        // are two thirds of the vertices -- every Kempe component is the whole graph -- and the attempt cost 1.5 s on a 901 120-row torus, mostly in vain
        if (nbest <= 4 && A.nr > 70000) break;
        cur = best;
        const bool emptied = dissolve_top_class(A, cur, nbest - 1);
        best = cur;  // partial progress is kept: the colouring stays valid
        compact_colors(best);
        nbest = count_colors(best);
        lap("dissolve top class", nbest);
        if (!emptied) break;
    }
    return coloring_is_valid(A, best) ? best : fallback;   // belt and braces: a wrong colouring would be a data race in the sweep
}

// Colouring of a mid-point-subdivided mesh from a proper 4-colouring of its parent (Tait's construction): the old
// vertices are pairwise non-adjacent in the subdivided graph (every old edge was split) -> colour 0; the mid-point of
// the old edge (a,b) gets c_a XOR c_b in {1,2,3}: the three edges of an old face get three different values, and
// mid-points are adjacent exactly when their edges share a face.  P must have the subdivision structure (rows with one
// entry 1.0 or two entries 0.5); the result is validated against the pattern of A and rejected otherwise.
bool subdivision_colors(const Csr& P, const std::vector<int>& coarse_color, const Csr& A, std::vector<int>& out)
{
    if (P.nr != A.nr || (int)coarse_color.size() != P.nc) return false;
    for (int c : coarse_color) if (c < 0 || c > 3) return false;
    out.assign(P.nr, -1);
    std::atomic<int> bad{0};
    parallel_for(P.nr, 1 << 15, [&](long r0, long r1) {
        for (long i = r0; i < r1; i++) {
            const int b = P.ptr[i], e = P.ptr[i + 1];
            if (e - b == 1 && P.val[b] == 1.0) out[i] = 0;
            else if (e - b == 2 && P.val[b] == 0.5 && P.val[b + 1] == 0.5) {
                const int x = coarse_color[P.col[b]] ^ coarse_color[P.col[b + 1]];
                if (x == 0) { bad.store(1, std::memory_order_relaxed); return; }
                out[i] = x;
            } else { bad.store(1, std::memory_order_relaxed); return; }
        }
    });
    if (bad.load()) return false;
    parallel_for(A.nr, 1 << 15, [&](long r0, long r1) {
        for (long i = r0; i < r1 && !bad.load(std::memory_order_relaxed); i++)
            for (int p = A.ptr[i]; p < A.ptr[i + 1]; p++)
                if (A.col[p] != i && out[A.col[p]] == out[i]) { bad.store(1, std::memory_order_relaxed); break; }
    });
    return bad.load() == 0;
}

std::vector<int> colours_for_ordering(const Csr& A, const std::vector<int>& rcm) { return color_graph(A, rcm); }

Ordering make_ordering(const Csr& A, int sigma, const std::vector<int>* preset_colors, const std::vector<int>* rcm_in, bool preset_final)
{
    int n = A.nr;
    Ordering o;
    std::vector<int> rcm = rcm_in ? *rcm_in : rcm_order(A);  // new -> old
    std::vector<int> color = preset_colors ? *preset_colors : color_graph(A, rcm);
    if (preset_colors && !preset_final) compact_colors(color);
    int ncol = count_colors(color);
    o.color_of = color;
    // colour-major, RCM rank inside a colour (counting sort keeps the RCM order stable)
    o.color_ptr.assign(ncol + 1, 0);
    for (int v = 0; v < n; v++) o.color_ptr[color[v] + 1]++;
    for (int c = 0; c < ncol; c++) o.color_ptr[c + 1] += o.color_ptr[c];
    o.perm.resize(n);
    {
        // a stable partition of the RCM sequence by colour; big levels: chunks of the sequence counted, then placed, side by side
        const int chunks = n >= (1 << 17) ? std::min(16, host_threads()) : 1;
        std::vector<std::vector<int>> cnt((size_t)chunks, std::vector<int>((size_t)ncol, 0));
        auto t0 = [&](int c) { return (int)((long)n * c / chunks); };
        parallel_for(chunks, 1, [&](long c0, long c1) {
            for (long c = c0; c < c1; c++) for (int t = t0((int)c); t < t0((int)c + 1); t++) cnt[(size_t)c][(size_t)color[rcm[t]]]++;
        });
        for (int k = 0; k < ncol; k++) {
            int run = o.color_ptr[k];
            for (int c = 0; c < chunks; c++) { const int m = cnt[(size_t)c][(size_t)k]; cnt[(size_t)c][(size_t)k] = run; run += m; }
        }
        parallel_for(chunks, 1, [&](long c0, long c1) {
            for (long c = c0; c < c1; c++) {
                std::vector<int>& next = cnt[(size_t)c];
                for (int t = t0((int)c); t < t0((int)c + 1); t++) { const int v = rcm[t]; o.perm[next[(size_t)color[v]]++] = v; }
            }
        });
    }
    // rows bucketed by nnz: inside each sigma-row window of a colour, longest rows first (stable)
    if (sigma > 1) {
        std::vector<std::pair<int, int>> win;      // the windows are independent
        for (int c = 0; c < ncol; c++)
            for (int b = o.color_ptr[c]; b < o.color_ptr[c + 1]; b += sigma) win.emplace_back(b, std::min(b + sigma, o.color_ptr[c + 1]));
        parallel_for((long)win.size(), 64, [&](long w0, long w1) {
            for (long w = w0; w < w1; w++)
                std::stable_sort(o.perm.begin() + win[(size_t)w].first, o.perm.begin() + win[(size_t)w].second, [&](int x, int y) {
                    return (A.ptr[x + 1] - A.ptr[x]) > (A.ptr[y + 1] - A.ptr[y]);
                });
        });
    }
    o.iperm.resize(n);
    parallel_for(n, 1 << 16, [&](long a, long b) { for (long i = a; i < b; i++) o.iperm[(size_t)o.perm[(size_t)i]] = (int)i; });
    if (n == 0) o.color_ptr = {0, 0};
    return o;
}

std::vector<int> induced_order(const Csr& P, const std::vector<int>& coarse_rank)
{
    const int n = P.nr, nc = P.nc;
    std::vector<int> key(n);
    parallel_for(n, 65536, [&](long r0, long r1) {
        for (long i = r0; i < r1; i++) {
            int par = -1;
            double best = -1.0;
            for (int p = P.ptr[i]; p < P.ptr[i + 1]; p++) {
                const double w = std::fabs(P.val[p]);
                if (w > best) { best = w; par = P.col[p]; }
            }
            key[i] = par >= 0 ? coarse_rank[par] : nc;   // rows without a parent go last
        }
    });
    // stable counting sort by parent position
    std::vector<int> start(nc + 2, 0);
    for (int i = 0; i < n; i++) start[key[i] + 1]++;
    for (int k = 0; k <= nc; k++) start[k + 1] += start[k];
    std::vector<int> order(n);
    for (int i = 0; i < n; i++) order[start[key[i]]++] = i;
    return order;
}

Sell sell_layout(const std::vector<int>& row_len, int n_cols, long nnz, const std::vector<int>* row_breaks, int C, bool region_order, int pitch_policy)
{
    Sell S;
    S.C = C;
    const int n_rows = (int)row_len.size();
    S.n_rows = n_rows; S.n_cols = n_cols; S.nnz = nnz;
    std::vector<int> breaks;
    if (row_breaks) breaks = *row_breaks;
    else breaks = {0, n_rows};
    S.slice_row.push_back(0);
    S.color_slice_ptr.push_back(0);
    long sum_w = 0;
    int wmax = 0, wmin = 1 << 30;
    for (size_t c = 0; c + 1 < breaks.size(); c++) {
        for (int r0 = breaks[c]; r0 < breaks[c + 1]; r0 += C) {
            int r1 = std::min(r0 + C, breaks[c + 1]);
            int w = 0;
            for (int r = r0; r < r1; r++) w = std::max(w, row_len[(size_t)r]);
            S.slice_row.push_back(r1);
            S.slice_w.push_back(w);
            sum_w += w; wmax = std::max(wmax, w); wmin = std::min(wmin, w);
        }
        S.color_slice_ptr.push_back((int)S.slice_row.size() - 1);
    }
    S.n_slices = (int)S.slice_row.size() - 1;
    // fixed stride unless a few very wide slices would blow the storage up (then: compact panels, table-driven addressing)
    const int allow_stride = pitch_policy >= 0 ? pitch_policy : std::getenv("SMG_SELL_STRIDE") ? std::atoi(std::getenv("SMG_SELL_STRIDE")) : 1;   // A/B knob, and the tests' way to the compact layout
    if (allow_stride && S.n_slices > 0 && wmax > 0 && (long)wmax * S.n_slices <= (5 * sum_w) / 2 + 64) {
        // columns requested before a slice's width is known: the smallest W that covers 90% of the slices (narrower slices read
        // padding there, which the stride guarantees to exist; wider ones continue table-driven)
        S.stride = wmax;
        std::vector<int> hist(wmax + 1, 0);
        for (int w : S.slice_w) hist[w]++;
        int cum = 0, W = wmin;
        for (int w = 0; w <= wmax; w++) { cum += hist[w]; if (10 * (long)cum >= 9 * (long)S.n_slices) { W = w; break; } }
        S.w_lo = std::max(W, wmin);
    }
    S.slice_off.assign(S.n_slices + 1, 0);
    for (int s = 0; s < S.n_slices; s++) S.slice_off[s + 1] = S.stride ? (s + 1) * S.stride : S.slice_off[s] + S.slice_w[s];
    if (region_order && S.color_slice_ptr.size() > 2) {
        // key = position of the slice inside its colour block, in [0,1): rows of a colour are in RCM order, so equal
        // keys across colours are the same region of the mesh
        std::vector<std::pair<double, int>> key(S.n_slices);
        for (size_t c = 0; c + 1 < S.color_slice_ptr.size(); c++) {
            const int b = S.color_slice_ptr[c], e = S.color_slice_ptr[c + 1];
            for (int s = b; s < e; s++) key[s] = {(s - b + 0.5) / (double)(e - b), s};
        }
        std::stable_sort(key.begin(), key.end(), [](const std::pair<double, int>& x, const std::pair<double, int>& y) { return x.first < y.first; });
        S.region_order.resize(S.n_slices);
        for (int i = 0; i < S.n_slices; i++) S.region_order[i] = key[i].second;
    }
    return S;
}

Sell build_sell(const Csr& A, const std::vector<int>* row_breaks, int C, bool region_order, int pitch_policy)
{
    std::vector<int> row_len((size_t)A.nr);
    for (int r = 0; r < A.nr; r++) row_len[(size_t)r] = A.ptr[(size_t)r + 1] - A.ptr[(size_t)r];
    Sell S = sell_layout(row_len, A.nc, A.nnz(), row_breaks, C, region_order, pitch_policy);
    size_t tot = (size_t)C * (size_t)S.slice_off.back();
    S.col.assign(tot, -1);
    S.val.assign(tot, 0.0);
    S.entry.assign(tot, -1);
    parallel_for(S.n_slices, 512, [&](long s0, long s1) {
        for (long s = s0; s < s1; s++) {
            size_t base = (size_t)C * (size_t)S.slice_off[s];
            for (int r = S.slice_row[s]; r < S.slice_row[s + 1]; r++) {
                int lane = r - S.slice_row[s];
                int j = 0;
                for (int p = A.ptr[r]; p < A.ptr[r + 1]; p++, j++) {
                    S.col[base + (size_t)j * C + lane] = A.col[p];
                    S.val[base + (size_t)j * C + lane] = A.val[p];
                    S.entry[base + (size_t)j * C + lane] = p;
                }
            }
        }
    });
    return S;
}

}  // namespace smg
