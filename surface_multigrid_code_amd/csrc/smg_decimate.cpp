// smg_decimate.cpp -- host hierarchy builder behind smg_mg_precompute: one coarsening step
// (the role of the reference's get_prolong(), src/get_prolong.cpp:3-57:  decimate to tarF faces, push every
// fine vertex through the collapse history to barycentric coordinates on the coarse mesh, assemble P with
// exactly three stored entries per row).
//
// The construction follows the reference's (src/SSP_midpoint.cpp, src/SSP_collapse_edge.cpp, src/joint_lscm.cpp,
// src/query_fine_to_coarse.cpp), written from its formulation on own data structures (no libigl edge-flap arrays):
//   * the boundary is closed by a vertex at infinity: every boundary edge gets a phantom face to it (igl::connect_boundary_to_infinity,
//     SSP_midpoint.cpp:31); edges to that vertex cost infinity and are never collapsed, phantom faces are removed at the end and do
//     not count towards the target (max_faces_stopping_condition, :53);
//   * greedy loop (SSP_midpoint.cpp:188-220, SSP_collapse_edge.cpp:401-533): cheapest edge first (dec_type 1: length, merged vertex at
//     the mid-point -- ALSO for boundary vertices, as in the reference), lazy deletion by time stamps; an edge whose collapse is refused
//     leaves the queue (cost infinity there) and comes back only when a successful collapse next to it re-costs the edges of the
//     merged vertex's new star (:482-520) -- no other retry, no cap on how much a vertex may absorb (an absorption cap is an opt-in
//     argument, see decimate_level);
//   * validity (igl::edge_collapse_is_valid, SSP_collapse_edge.cpp:57): exactly two common neighbours in the closed mesh, not the
//     edge of a single tetrahedron; patches of <= 2 faces are refused (:188-195);
//   * per collapse the 1-rings before and after are flattened JOINTLY by least-squares conformal maps (joint_lscm.cpp:483-555 flatten():
//     Q = -L_pre + 2 A_pre - L_post + 2 A_post, dense constrained minimisation), in one of three set-ups (joint_lscm.cpp:226-241):
//       case 0  both end points interior: the merged vertex is an extra unknown; pins uv(vi) = (0,0), uv(vj) = (1,0)      (:557-651)
//       case 1  one end point on the boundary: the merged vertex shares that end point's unknown                           (:653-749)
//       case 2  a boundary edge: three candidates -- merged vertex snapped onto vi, onto vj (the far boundary neighbour of the other
//               end point pinned onto the line through vi, vj), or free on that line together with both boundary neighbours -- and the
//               one with the smallest quasi-conformal error wins                                                            (:750-1131)
//     boundary cases first check the 3D quality of the post-collapse triangles, 4 sqrt(3) area / sum l^2 >= 0.3 (:91-117);
//   * the flattening is refused on NaN, flipped faces (signed area < 1e-10), fold-over around the collapsing / merged vertices (angle
//     sum > 2 pi + 1e-10) or UV slivers (quality < 0.01) (check_valid_UV_lscm, :243-481);
//   * every fine point of the old 1-ring is located in the flattened post patch by the largest-minimum-barycentric rule, clamped and
//     renormalised (query_fine_to_coarse.cpp:93-116).
// Not restated: libigl's edge numbering (ties between equal costs may be broken differently), the randomised variants, the
// coarse-to-fine queries of the remeshing demos.  dec_type 0 / 2 reuse the same machinery with libsmg's own cost / placement (quadric
// error metric; end-point placement) -- the reference's versions of those two are built on libigl's quadric callbacks.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <queue>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "smg_mesh.hpp"

namespace smg {
namespace {

struct V3 { double x, y, z; };
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double norm(V3 a) { return std::sqrt(dot(a, a)); }

// ---- joint conformal flattening of the pre- and post-collapse 1-rings -------------------------------------------------------
//   minimise  E = sum over {pre, post}  1/2 (u^T K u + v^T K v) - Area(u, v),   K = cotangent stiffness of the 3D patch,
//   subject to pinned coordinates.  Unknowns: the local vertices of the patch; the merged vertex is an extra unknown (`lm` = n - 1)
//   or shares the unknown of an end point (cases 1 and 2-snap), in which case the post patch sees that index at the merged position.
struct Patch {
    int n = 0;                                  // local unknown vertices
    int la = -1, lb = -1, lm = -1;              // end points of the edge, merged vertex (may equal la or lb)
    std::vector<std::array<int, 3>> pre, post;  // local faces
    std::vector<V3> P, Ppost;                   // positions seen by the pre / post patch (they differ at a shared merged index)
    std::vector<double> U, Vv;                  // solution
};
struct Pin { int idx; double val; };            // idx in [0, 2n): u block first, then v

// The energy's matrix Q (2n x 2n, u block first) is never formed: the constrained system only needs its free x free part -- of which
// the Cholesky factorisation reads the lower triangle -- and, for the right-hand side, the columns of the pins with a non-zero value.
// `sym(I, J, w)` stands for the pair Q[I][J] += w, Q[J][I] += w; every entry receives its terms in the order the full matrix would.
template <class Sym, class Diag>
static inline void add_conformal_energy(int n, const std::vector<V3>& X, const std::vector<std::array<int, 3>>& F, Sym&& sym, Diag&& diag)
{
    for (const auto& f : F) {
        for (int c = 0; c < 3; c++) {
            const int i = f[c], j = f[(c + 1) % 3], k = f[(c + 2) % 3];   // edge (i,j), opposite corner k
            const V3 e1 = X[i] - X[k], e2 = X[j] - X[k];
            const double cr = norm(cross(e1, e2));
            const double w = cr > 0 ? 0.5 * dot(e1, e2) / cr : 0.0;       // 1/2 cot(angle at k)
            for (int d = 0; d < 2; d++) {                                   // K (+)= w (x_i - x_j)^2 for u and v
                const int I = i + d * n, J = j + d * n;
                diag(I, w); diag(J, w);
                sym(I, J, -w);
            }
            // - 2 S:  Area = 1/2 sum over directed face edges (u_i v_j - u_j v_i)
            sym(i, j + n, -0.5);
            sym(j, i + n, 0.5);
        }
    }
}

// quasi-conformal error of a flattened patch: per face the ratio of the singular values of the map UV -> 3D (Sander et al., "Texture
// Mapping Progressive Meshes"; the reference's selection criterion between the case-2 candidates, src/quasi_conformal_error.cpp),
// 2-norm over the faces
static double qc_error_norm(const std::vector<V3>& X, const std::vector<std::array<int, 3>>& F, const std::vector<double>& U, const std::vector<double>& W)
{
    double ss = 0.0;
    for (const auto& f : F) {
        const double s1 = U[f[0]], s2 = U[f[1]], s3 = U[f[2]], t1 = W[f[0]], t2 = W[f[1]], t3 = W[f[2]];
        const double A = ((s2 - s1) * (t3 - t1) - (s3 - s1) * (t2 - t1)) / 2.0;
        const V3 q1 = X[f[0]], q2 = X[f[1]], q3 = X[f[2]];
        const V3 Ss = (1.0 / (2.0 * A)) * ((t2 - t3) * q1 + (t3 - t1) * q2 + (t1 - t2) * q3);
        const V3 St = (1.0 / (2.0 * A)) * ((s3 - s2) * q1 + (s1 - s3) * q2 + (s2 - s1) * q3);
        const double a = dot(Ss, Ss), b = dot(Ss, St), c = dot(St, St);
        const double disc = std::sqrt((a - c) * (a - c) + 4.0 * b * b);
        const double sigma = std::sqrt((a + c + disc) / 2.0), gamma = std::sqrt((a + c - disc) / 2.0);
        const double e = sigma / gamma;
        ss += e * e;
    }
    return std::sqrt(ss);
}

// solves the constrained minimisation and runs the validity checks of check_valid_UV_lscm; false = refuse the collapse
static bool solve_joint_flattening(Patch& pt, const std::vector<Pin>& pins)
{
    const int n = pt.n, N = 2 * n;
    // (scratch vectors live across calls: a collapse is a few microseconds of arithmetic, allocation would be a good part of it)
    static thread_local std::vector<double> M, rhs, x, qpin;
    static thread_local std::vector<int> freei, fpos, pslot;
    fpos.assign(N, 0);                         // >= 0: position among the free unknowns; -1: pinned
    pslot.assign(N, -1);                       // pinned with a non-zero value: its column of Q is kept (qpin)
    x.assign(N, 0.0);
    int n_slots = 0;
    for (const Pin& p : pins) {
        if (fpos[p.idx] < 0) { x[p.idx] = p.val; continue; }
        fpos[p.idx] = -1; x[p.idx] = p.val;
    }
    for (const Pin& p : pins) if (p.val != 0.0 && pslot[p.idx] < 0) pslot[p.idx] = n_slots++;
    freei.clear();
    for (int i = 0; i < N; i++) if (fpos[i] >= 0) { fpos[i] = (int)freei.size(); freei.push_back(i); }
    const int m = (int)freei.size();
    M.assign((size_t)m * m, 0.0); rhs.resize(m);
    qpin.assign((size_t)n_slots * m, 0.0);
    {
        double* Mp = M.data();
        double* qp = qpin.data();
        const int* fp = fpos.data();
        const int* ps = pslot.data();
        auto sym = [&](int I, int J, double w) {
            const int a = fp[I], b = fp[J];
            if (a >= 0 && b >= 0) { if (a >= b) Mp[(size_t)a * m + b] += w; else Mp[(size_t)b * m + a] += w; }
            else if (a >= 0) { if (ps[J] >= 0) qp[(size_t)ps[J] * m + a] += w; }
            else if (b >= 0) { if (ps[I] >= 0) qp[(size_t)ps[I] * m + b] += w; }
        };
        auto diag = [&](int I, double w) { const int a = fp[I]; if (a >= 0) Mp[(size_t)a * m + a] += w; };
        add_conformal_energy(n, pt.P, pt.pre, sym, diag);
        add_conformal_energy(n, pt.Ppost, pt.post, sym, diag);
    }
    for (int r = 0; r < m; r++) {
        double s = 0.0;
        for (const Pin& p : pins) if (p.val != 0.0) s += qpin[(size_t)pslot[p.idx] * m + r] * p.val;
        rhs[r] = -s;
    }
    // dense Cholesky (the pinned conformal energy is positive definite on a valid patch)
    for (int j = 0; j < m; j++) {
        double d = M[(size_t)j * m + j];
        for (int k = 0; k < j; k++) d -= M[(size_t)j * m + k] * M[(size_t)j * m + k];
        if (!(d > 1e-14)) return false;
        d = std::sqrt(d);
        M[(size_t)j * m + j] = d;
        // (four rows side by side: each entry's sum keeps its order, the four chains of dependent subtractions overlap)
        const double* Mj = &M[(size_t)j * m];
        int i = j + 1;
        for (; i + 3 < m; i += 4) {
            double* r0 = &M[(size_t)i * m];
            double *r1 = r0 + m, *r2 = r1 + m, *r3 = r2 + m;
            double s0 = r0[j], s1 = r1[j], s2 = r2[j], s3 = r3[j];
            for (int k = 0; k < j; k++) { const double t = Mj[k]; s0 -= r0[k] * t; s1 -= r1[k] * t; s2 -= r2[k] * t; s3 -= r3[k] * t; }
            r0[j] = s0 / d; r1[j] = s1 / d; r2[j] = s2 / d; r3[j] = s3 / d;
        }
        for (; i < m; i++) {
            double sx = M[(size_t)i * m + j];
            for (int k = 0; k < j; k++) sx -= M[(size_t)i * m + k] * Mj[k];
            M[(size_t)i * m + j] = sx / d;
        }
    }
    for (int i = 0; i < m; i++) { double sx = rhs[i]; for (int k = 0; k < i; k++) sx -= M[(size_t)i * m + k] * rhs[k]; rhs[i] = sx / M[(size_t)i * m + i]; }
    for (int i = m - 1; i >= 0; i--) { double sx = rhs[i]; for (int k = i + 1; k < m; k++) sx -= M[(size_t)k * m + i] * rhs[k]; rhs[i] = sx / M[(size_t)i * m + i]; }
    for (int r = 0; r < m; r++) x[freei[r]] = rhs[r];
    pt.U.assign(x.begin(), x.begin() + n);
    pt.Vv.assign(x.begin() + n, x.end());
    for (double v : x) if (!(v == v)) return false;
    // every flattened face of both patches must keep its orientation (check_valid_UV_lscm: signed area >= 1e-10)
    auto oriented = [&](const std::vector<std::array<int, 3>>& F) {
        for (const auto& f : F) {
            const double ar = (pt.U[f[1]] - pt.U[f[0]]) * (pt.Vv[f[2]] - pt.Vv[f[0]]) - (pt.Vv[f[1]] - pt.Vv[f[0]]) * (pt.U[f[2]] - pt.U[f[0]]);
            if (!(ar >= 1e-10)) return false;
        }
        return true;
    };
    if (!(oriented(pt.pre) && oriented(pt.post))) return false;
    // no fold-over: the flattened angles around the collapsing vertices / the merged vertex must not exceed 2 pi
    // (src/joint_lscm.cpp:330-392), and no flattened sliver: quality 4 sqrt(3) area / (l0^2 + l1^2 + l2^2) >= 0.01 (:394-478)
    auto checks = [&](const std::vector<std::array<int, 3>>& F, int v0, int v1) {
        double ang0 = 0, ang1 = 0;
        for (const auto& f : F) {
            double l2[3], ar = 0;
            for (int c = 0; c < 3; c++) {
                const int i = f[c], j = f[(c + 1) % 3];
                const double dx = pt.U[i] - pt.U[j], dy = pt.Vv[i] - pt.Vv[j];
                l2[c] = dx * dx + dy * dy;
            }
            ar = 0.5 * ((pt.U[f[1]] - pt.U[f[0]]) * (pt.Vv[f[2]] - pt.Vv[f[0]]) - (pt.Vv[f[1]] - pt.Vv[f[0]]) * (pt.U[f[2]] - pt.U[f[0]]));
            const double q = 4.0 * std::sqrt(3.0) * ar / (l2[0] + l2[1] + l2[2]);
            if (!(q >= 0.01)) return false;
            for (int c = 0; c < 3; c++) {
                if (f[c] != v0 && f[c] != v1) continue;
                const int i = f[c], j = f[(c + 1) % 3], k = f[(c + 2) % 3];
                const double ax = pt.U[j] - pt.U[i], ay = pt.Vv[j] - pt.Vv[i], bx = pt.U[k] - pt.U[i], by = pt.Vv[k] - pt.Vv[i];
                const double ang = std::atan2(ax * by - ay * bx, ax * bx + ay * by);
                (f[c] == v0 ? ang0 : ang1) += ang;
            }
        }
        const double two_pi = 6.283185307179586;
        return ang0 - two_pi <= 1e-10 && ang1 - two_pi <= 1e-10;
    };
    // pre: around vi and vj; post: around the merged vertex (the reference tests the indices vi, vj in both patches, :334-392; vj does
    // not occur in the post patch, and with a shared unknown the merged vertex IS la or lb)
    return checks(pt.pre, pt.la, pt.lb) && checks(pt.post, pt.lm, -1);
}

struct QEntry {
    double cost;
    int a, b, va, vb;
    bool operator<(const QEntry& o) const
    {   // std::priority_queue is a max-heap: invert; ties broken deterministically
        if (cost != o.cost) return cost > o.cost;
        if (a != o.a) return a > o.a;
        return b > o.b;
    }
};

struct Decimator {
    DecimationLog* log = nullptr;            // optional: the record of every collapse (query_coarse_to_fine)
    long refuse_reason[32] = {0};   // statistics (SMG_DEC_STATS=1): refusals by the `return` that issued them, in source order
    std::vector<V3> pos;
    std::vector<char> valive;
    std::vector<int> version;
    std::vector<std::array<int, 3>> faces;
    std::vector<char> falive;
    std::vector<std::vector<int>> vfaces;   // incident faces (may hold dead ones; filtered on access)
    std::vector<std::vector<int>> fpoints;  // fine points living on each face
    std::vector<int> pface;
    std::vector<std::array<double, 3>> pbary;
    // Edges created or re-offered during the loop.  One heap over all of them grows to millions of entries at a million vertices (lazy
    // deletion: ten fresh entries per collapse, the stale ones leave only when they surface), and every pop walks twenty levels of cache
    // misses.  Costs rise as the mesh coarsens, so entries are kept by cost CLASS (the bit pattern of the non-negative double, 8 classes per
    // octave): classes up to `cur_class` live in the heap `pq`, the rest wait in unsorted buckets and enter the heap when the front of the
    // queue reaches their class.  The heap then holds the entries within a tenth of the current cost; the pop sequence is the one heap's.
    std::priority_queue<QEntry> pq;
    std::vector<std::pair<uint32_t, std::vector<QEntry>>> far;   // (class, entries), classes ascending
    uint32_t cur_class = 0;
    size_t n_far = 0;
    static uint32_t cost_class(double c)
    {
        if (!(c > 0.0)) return 0;                 // zero, negative (rounding of a quadric cost) or NaN: always in the heap
        uint64_t b; std::memcpy(&b, &c, 8);
        return (uint32_t)(b >> 49) + 1;            // sign 0: monotone in c
    }
    void offer(const QEntry& e)
    {
        const uint32_t k = cost_class(e.cost);
        if (k <= cur_class) { pq.push(e); return; }
        auto it = std::lower_bound(far.begin(), far.end(), k, [](const std::pair<uint32_t, std::vector<QEntry>>& x, uint32_t key) { return x.first < key; });
        if (it == far.end() || it->first != k) it = far.insert(it, {k, {}});
        it->second.push_back(e);
        n_far++;
    }
    // makes sure the smallest live entry is at one of the two fronts: while the heap is empty, or its top lies beyond the current class,
    // -- and the sorted front does not come first anyway -- the next class moves in
    void settle()
    {
        while (!far.empty()) {
            const bool init_left = ihead < initial.size();
            const uint32_t next = far.front().first;
            const bool heap_ok = !pq.empty() && cost_class(pq.top().cost) < next;
            const bool init_ok = init_left && cost_class(initial[ihead].cost) < next;
            if (heap_ok || init_ok) return;        // an entry of a class below every waiting one exists: it is the minimum's class
            cur_class = next;
            for (const QEntry& e : far.front().second) pq.push(e);
            n_far -= far.front().second.size();
            far.erase(far.begin());
        }
    }
    // The edges of the input mesh -- most of what the queue ever holds -- are sorted once and consumed front to back; the heap only
    // takes the edges created or re-offered later.  pop_next() returns the better of the two fronts: the same sequence one heap
    // holding everything would produce (the order of QEntry is total), at a fraction of the cache misses.
    std::vector<QEntry> initial;
    size_t ihead = 0;
    bool filling = true;
    void seal_initial()
    {
        parallel_sort(initial, [](const QEntry& x, const QEntry& y) { return y < x; });   // best first (3 #V entries: on the host threads -- a fifth of the level's time as one std::sort)
        filling = false;
    }
    bool queue_empty() const { return pq.empty() && ihead == initial.size() && n_far == 0; }
    QEntry pop_next()
    {
        settle();
        if (pq.empty() || (ihead < initial.size() && !(initial[ihead] < pq.top()))) {
            // the sorted front is consumed in order: what the collapses a few places down will touch is requested now (the loop is a
            // chain of cache misses otherwise) -- list headers first, a few pops later the lists, then the faces and their point lists
            const size_t n = initial.size();
            if (ihead + 64 < n) {
                const QEntry& e = initial[ihead + 64];
                __builtin_prefetch(&vfaces[e.a]); __builtin_prefetch(&vfaces[e.b]);
                __builtin_prefetch(&pos[e.a]); __builtin_prefetch(&pos[e.b]);
                __builtin_prefetch(&version[e.a]); __builtin_prefetch(&version[e.b]);
                __builtin_prefetch(&valive[e.a]); __builtin_prefetch(&valive[e.b]);
            }
            if (ihead + 48 < n) {
                const QEntry& e = initial[ihead + 48];
                __builtin_prefetch(vfaces[e.a].data()); __builtin_prefetch(vfaces[e.b].data());
            }
            if (ihead + 32 < n) {
                const QEntry& e = initial[ihead + 32];
                if (valive[e.a] && valive[e.b] && version[e.a] == e.va && version[e.b] == e.vb)
                    for (int v : {e.a, e.b})
                        for (int f : vfaces[v]) { __builtin_prefetch(&faces[f]); __builtin_prefetch(&fpoints[f]); __builtin_prefetch(&falive[f]); }
            }
            return initial[ihead++];
        }
        QEntry e = pq.top();
        pq.pop();
        return e;
    }
    int n_alive_faces = 0;                   // REAL faces alive (phantom faces to the vertex at infinity do not count)
    int dec_type = 1;
    int nF_real = 0;                         // faces [nF_real, ...) are phantom: (b, a, inf) for every boundary edge a -> b
    int vinf = -1;                           // the vertex at infinity (-1: closed mesh)
    std::unordered_set<uint64_t> refused;    // edges whose collapse was refused (cost infinity in the reference's queue)
    std::vector<int> n_refused;              // per vertex: refused edges at it (the set is only consulted when both ends have some)
    void refuse(int a, int b) { if (refused.insert(edge_key(a, b)).second) { n_refused[a]++; n_refused[b]++; } }
    bool unrefuse(int a, int b)
    {
        if (n_refused[a] == 0 || n_refused[b] == 0) return false;
        if (!refused.erase(edge_key(a, b))) return false;
        n_refused[a]--; n_refused[b]--;
        return true;
    }
    bool phantom(int f) const { return f >= nF_real; }
    static uint64_t edge_key(int a, int b) { if (a > b) std::swap(a, b); return ((uint64_t)(uint32_t)a << 32) | (uint32_t)b; }
    // dec_type 0 (the reference's "qslim", src/SSP_qslim.cpp): quadric error metric -- every vertex carries the area-weighted
    // sum of the squared distances to the planes of its input faces, Q(v) = v^T A v + 2 b^T v + c stored as
    // (a11 a12 a13 a22 a23 a33 b1 b2 b3 c); an edge costs the minimum of Q_a + Q_b and the merged vertex goes to the minimiser
    // (Garland & Heckbert).  Boundary edges add a plane through the edge perpendicular to their face, so borders keep their shape
    // (the reference reaches the same end by connecting the boundary to a vertex at infinity).
    std::vector<std::array<double, 10>> quad;
    std::vector<int> locmap;                 // scratch of collapse(): global vertex -> local index of the current patch (-1 outside)

    static void add_plane(std::array<double, 10>& q, V3 n, double d, double w)
    {
        q[0] += w * n.x * n.x; q[1] += w * n.x * n.y; q[2] += w * n.x * n.z;
        q[3] += w * n.y * n.y; q[4] += w * n.y * n.z; q[5] += w * n.z * n.z;
        q[6] += w * d * n.x; q[7] += w * d * n.y; q[8] += w * d * n.z; q[9] += w * d * d;
    }
    static double qeval(const std::array<double, 10>& q, V3 v)
    {
        return q[0] * v.x * v.x + 2 * q[1] * v.x * v.y + 2 * q[2] * v.x * v.z + q[3] * v.y * v.y + 2 * q[4] * v.y * v.z + q[5] * v.z * v.z +
               2 * (q[6] * v.x + q[7] * v.y + q[8] * v.z) + q[9];
    }
    void init_quadrics()
    {
        quad.assign(pos.size(), std::array<double, 10>{});
        for (size_t f = 0; f < (size_t)nF_real; f++) {
            const auto& fc = faces[f];
            const V3 p0 = pos[fc[0]], p1 = pos[fc[1]], p2 = pos[fc[2]];
            const V3 cr = cross(p1 - p0, p2 - p0);
            const double l = norm(cr);
            if (!(l > 0)) continue;
            const V3 n = (1.0 / l) * cr;
            for (int c = 0; c < 3; c++) add_plane(quad[fc[c]], n, -dot(n, p0), 0.5 * l);
            for (int c = 0; c < 3; c++) {   // boundary edges: a constraint plane through the edge, perpendicular to the face
                const int a = fc[c], b = fc[(c + 1) % 3];
                int ef[3];
                if (edge_faces(a, b, ef) != 1) continue;
                const V3 e = pos[b] - pos[a];
                const V3 nb0 = cross(e, n);
                const double lb = norm(nb0);
                if (!(lb > 0)) continue;
                const V3 nb = (1.0 / lb) * nb0;
                const double w = 10.0 * dot(e, e);   // length^2: the scale of an area, weighted up so that the border wins
                add_plane(quad[a], nb, -dot(nb, pos[a]), w);
                add_plane(quad[b], nb, -dot(nb, pos[a]), w);
            }
        }
    }
    // minimiser and minimum of Q_a + Q_b; falls back to the better of the end points / the mid-point when the 3 x 3 system is
    // (nearly) singular -- flat or straight neighbourhoods -- or the minimiser runs away from the edge
    double qem(int a, int b, V3* where) const
    {
        std::array<double, 10> q;
        for (int i = 0; i < 10; i++) q[i] = quad[a][i] + quad[b][i];
        const double a11 = q[0], a12 = q[1], a13 = q[2], a22 = q[3], a23 = q[4], a33 = q[5];
        const double c11 = a22 * a33 - a23 * a23, c12 = a13 * a23 - a12 * a33, c13 = a12 * a23 - a13 * a22;
        const double det = a11 * c11 + a12 * c12 + a13 * c13;
        const double tr = (a11 + a22 + a33) / 3.0;
        const V3 mid = 0.5 * (pos[a] + pos[b]);
        V3 best = mid;
        double cost = qeval(q, mid);
        bool have_opt = false;
        if (std::fabs(det) > 1e-6 * tr * tr * tr && tr > 0) {
            const double c22 = a11 * a33 - a13 * a13, c23 = a12 * a13 - a11 * a23, c33 = a11 * a22 - a12 * a12;
            const V3 v = {-(c11 * q[6] + c12 * q[7] + c13 * q[8]) / det, -(c12 * q[6] + c22 * q[7] + c23 * q[8]) / det,
                          -(c13 * q[6] + c23 * q[7] + c33 * q[8]) / det};
            if (norm(v - mid) <= 2.0 * norm(pos[a] - pos[b])) { best = v; cost = qeval(q, v); have_opt = true; }
        }
        if (!have_opt) {
            const double ca = qeval(q, pos[a]), cb = qeval(q, pos[b]);
            if (ca < cost) { cost = ca; best = pos[a]; }
            if (cb < cost) { cost = cb; best = pos[b]; }
        }
        if (where) *where = best;
        return cost;
    }

    void clean(int v)
    {
        auto& l = vfaces[v];
        l.erase(std::remove_if(l.begin(), l.end(), [&](int f) { return !falive[f]; }), l.end());
    }
    bool has(const std::array<int, 3>& f, int v) const { return f[0] == v || f[1] == v || f[2] == v; }

    // faces containing both a and b
    int edge_faces(int a, int b, int* out)
    {
        int n = 0;
        for (int f : vfaces[a]) if (falive[f] && !phantom(f) && has(faces[f], b)) { if (n < 3) out[n] = f; n++; }   // real faces only
        return n;
    }
    void push_edge(int a, int b)
    {
        if (a == vinf || b == vinf) return;   // infinite cost: never collapsed (SSP_midpoint.cpp:196-200)
        if (a > b) std::swap(a, b);
        unrefuse(a, b);
        const QEntry e{dec_type == 0 ? qem(a, b, nullptr) : norm(pos[a] - pos[b]), a, b, version[a], version[b]};
        if (filling) initial.push_back(e); else offer(e);
    }
    void push_star(int v)
    {
        static thread_local std::vector<int> nb;
        nb.clear();
        for (int f : vfaces[v]) if (falive[f]) for (int c = 0; c < 3; c++) if (faces[f][c] != v) nb.push_back(faces[f][c]);
        std::sort(nb.begin(), nb.end());
        nb.erase(std::unique(nb.begin(), nb.end()), nb.end());
        for (int w : nb) push_edge(v, w);
    }
    // try to collapse (a,b), a < b: b merges into a.  Returns true on success.
    bool collapse(int a, int b)
    {
        clean(a); clean(b);
        // ---- neighbourhoods in the CLOSED mesh (phantom faces and the vertex at infinity included)
        static thread_local std::vector<int> na, nb, common;
        na.clear(); nb.clear(); common.clear();
        for (int f : vfaces[a]) for (int c = 0; c < 3; c++) if (faces[f][c] != a) na.push_back(faces[f][c]);
        for (int f : vfaces[b]) for (int c = 0; c < 3; c++) if (faces[f][c] != b) nb.push_back(faces[f][c]);
        std::sort(na.begin(), na.end()); na.erase(std::unique(na.begin(), na.end()), na.end());
        std::sort(nb.begin(), nb.end()); nb.erase(std::unique(nb.begin(), nb.end()), nb.end());
        if (!std::binary_search(na.begin(), na.end(), b)) return (refuse_reason[1]++, false);          // not an edge (any more)
        // igl::edge_collapse_is_valid (SSP_collapse_edge.cpp:55-60): the end points share exactly two neighbours, and the edge is not
        // an edge of a single tetrahedron (both end points of valence 3 on the same two neighbours)
        std::set_intersection(na.begin(), na.end(), nb.begin(), nb.end(), std::back_inserter(common));
        if ((int)common.size() != 2) return (refuse_reason[2]++, false);
        if (na.size() == 3 && nb.size() == 3) return (refuse_reason[3]++, false);
        const bool ba = vinf >= 0 && std::binary_search(na.begin(), na.end(), vinf);   // on the boundary <=> adjacent to infinity
        const bool bb = vinf >= 0 && std::binary_search(nb.begin(), nb.end(), vinf);
        const int kase = (ba ? 1 : 0) + (bb ? 1 : 0);
        // the faces on the edge: two in the closed mesh; a boundary edge has one real and one phantom face
        int ef[4], nef_all = 0, nef = 0;
        for (int f : vfaces[a]) if (has(faces[f], b)) { if (nef_all < 4) ef[nef_all] = f; nef_all++; if (!phantom(f)) nef++; }
        if (nef_all != 2 || nef < 1) return (refuse_reason[4]++, false);
        // both end points on the boundary: with two common neighbours the edge itself is a boundary edge (a chord between two boundary
        // vertices has three: its two opposite vertices and infinity -- the reference's `isFlap` refusal, joint_lscm.cpp:60-81)
        if (kase == 2 && nef != 1) return (refuse_reason[5]++, false);
        // ---- placement (dec_type 1: the mid-point, for boundary vertices too: shortest_edge_and_midpoint, SSP_midpoint.cpp:52)
        V3 m;
        if (dec_type == 1) m = 0.5 * (pos[a] + pos[b]);
        else if (ba != bb) m = ba ? pos[a] : pos[b];          // libsmg's own types 0 / 2: a boundary vertex stays where it is
        else if (dec_type == 2) m = pos[a];
        else (void)qem(a, b, &m);
        // libsmg's own types 0 / 2 additionally refuse collapses that fold a surviving face over in 3D or leave a sliver (the reference has
        // these tests commented out for its mid-point type, SSP_collapse_edge.cpp:196-236; a quadric-optimal or end-point placement needs them)
        if (dec_type != 1) {
            for (int pass = 0; pass < 2; pass++) {
                const int v = pass == 0 ? a : b;
                for (int f : vfaces[v]) {
                    if (phantom(f)) continue;
                    const auto& fc = faces[f];
                    if (has(fc, a) && has(fc, b)) continue;
                    V3 p0 = pos[fc[0]], p1 = pos[fc[1]], p2 = pos[fc[2]];
                    V3 n0 = cross(p1 - p0, p2 - p0);
                    V3 q0 = (fc[0] == v) ? m : p0, q1 = (fc[1] == v) ? m : p1, q2 = (fc[2] == v) ? m : p2;
                    V3 n1 = cross(q1 - q0, q2 - q0);
                    const double l0 = norm(n0), l1 = norm(n1);
                    if (!(l1 > 1e-14 * (1.0 + l0))) return (refuse_reason[6]++, false);
                    if (dot(n0, n1) < 0.2 * l0 * l1) return (refuse_reason[7]++, false);
                    const double e = std::max(norm(q1 - q0), std::max(norm(q2 - q1), norm(q0 - q2)));
                    if (l1 < 0.02 * e * e) return (refuse_reason[8]++, false);
                }
            }
        }
        // ---- local patch: ring vertices, a, b (and the merged vertex, where it is its own unknown)
        static thread_local Patch patch;
        static thread_local std::vector<int> pre_gid, post_gid;              // global face ids of patch.pre / patch.post
        patch.pre.clear(); patch.post.clear(); patch.P.clear(); patch.Ppost.clear(); patch.U.clear(); patch.Vv.clear(); patch.n = 0;
        pre_gid.clear(); post_gid.clear();
        if (locmap.size() != pos.size()) locmap.assign(pos.size(), -1);
        static thread_local std::vector<int> loc_touched;
        loc_touched.clear();
        struct LocGuard {
            std::vector<int>& m; std::vector<int>& touched;
            int& operator[](int v) { if (m[v] < 0) touched.push_back(v); return m[v]; }
            ~LocGuard() { for (int v : touched) m[v] = -1; }
        } loc{locmap, loc_touched};
        for (int v : na) if (v != b && v != vinf) { loc[v] = (int)patch.P.size(); patch.P.push_back(pos[v]); }
        for (int v : nb) if (v != a && v != vinf && !std::binary_search(na.begin(), na.end(), v)) { loc[v] = (int)patch.P.size(); patch.P.push_back(pos[v]); }
        const int la = (int)patch.P.size(); patch.P.push_back(pos[a]);
        const int lb = la + 1; patch.P.push_back(pos[b]);
        loc[a] = la; loc[b] = lb;
        const int n_shared = (int)patch.P.size();       // unknowns without an extra merged vertex
        // boundary neighbours of a boundary edge (case 2): pv = the neighbour of a along the boundary other than b, nv = that of b
        int pv = -1, nv = -1;
        if (kase == 2) {
            for (int f : vfaces[a]) if (phantom(f) && !has(faces[f], b)) for (int c = 0; c < 3; c++) if (faces[f][c] != a && faces[f][c] != vinf) pv = faces[f][c];
            for (int f : vfaces[b]) if (phantom(f) && !has(faces[f], a)) for (int c = 0; c < 3; c++) if (faces[f][c] != b && faces[f][c] != vinf) nv = faces[f][c];
            if (pv < 0 || nv < 0 || pv == nv || locmap[pv] < 0 || locmap[nv] < 0) return (refuse_reason[9]++, false);   // a boundary loop of three edges
        }
        // builds the face lists for a given index of the merged vertex
        auto build_faces = [&](int lm) {
            patch.pre.clear(); patch.post.clear(); pre_gid.clear(); post_gid.clear();
            auto add = [&](int v) {
                for (int f : vfaces[v]) {
                    if (phantom(f)) continue;
                    const auto& fc = faces[f];
                    if (v == b && has(fc, a)) continue;   // edge faces already taken from a's star
                    pre_gid.push_back(f);
                    patch.pre.push_back({loc[fc[0]], loc[fc[1]], loc[fc[2]]});
                    if (!(has(fc, a) && has(fc, b))) {
                        post_gid.push_back(f);
                        std::array<int, 3> g = {loc[fc[0]], loc[fc[1]], loc[fc[2]]};
                        for (int c2 = 0; c2 < 3; c2++) if (g[c2] == la || g[c2] == lb) g[c2] = lm;
                        patch.post.push_back(g);
                    }
                }
            };
            add(a); add(b);
        };
        // sets the patch up for one flattening: where the merged vertex lives, what the post patch sees there
        auto setup = [&](int lm_shared /* la, lb or -1 = own unknown */) {
            patch.P.resize(n_shared);
            int lm = lm_shared;
            if (lm < 0) { lm = n_shared; patch.P.push_back(m); }
            patch.n = (int)patch.P.size();
            patch.Ppost = patch.P;
            patch.Ppost[lm] = m;
            patch.la = la; patch.lb = lb; patch.lm = lm;
            build_faces(lm);
        };
        // When the merged position IS an end point (end-point placement) it shares that end point's unknown even for interior edges:
        // otherwise the surviving vertex's own record -- the fine vertex sitting exactly on the coarse vertex -- would be re-located
        // into the interior of a face (coarse vertices nobody interpolates from were the symptom).
        const bool m_is_a = (m.x == pos[a].x && m.y == pos[a].y && m.z == pos[a].z);
        const bool m_is_b = !m_is_a && (m.x == pos[b].x && m.y == pos[b].y && m.z == pos[b].z);
        const int own = m_is_a ? la : (m_is_b ? lb : -1);
        setup(kase == 1 ? (ba ? la : lb) : own);
        if ((int)patch.pre.size() <= 2) return (refuse_reason[10]++, false);                           // SSP_collapse_edge.cpp:188-195
        // boundary cases: 3D quality of the post-collapse triangles (joint_lscm.cpp:91-117)
        if (kase > 0) {
            for (const auto& g : patch.post) {
                const double l0 = norm(patch.Ppost[g[0]] - patch.Ppost[g[1]]), l1 = norm(patch.Ppost[g[1]] - patch.Ppost[g[2]]), l2 = norm(patch.Ppost[g[2]] - patch.Ppost[g[0]]);
                const double xs = (l0 + l1 + l2) / 2.0;
                const double delta = std::sqrt(xs * (xs - l0) * (xs - l1) * (xs - l2));
                const double q = 4.0 * std::sqrt(3.0) * delta / (l0 * l0 + l1 * l1 + l2 * l2);
                if (!(q >= 0.3)) return (refuse_reason[11]++, false);
            }
        }
        static thread_local std::vector<Pin> pins;
        auto base_pins = [&](int n) { pins.clear(); pins.push_back({la, 0.0}); pins.push_back({lb, 1.0}); pins.push_back({la + n, 0.0}); pins.push_back({lb + n, 0.0}); };
        if (kase < 2) {
            base_pins(patch.n);                                                 // uv(vi) = (0,0), uv(vj) = (1,0)
            if (!solve_joint_flattening(patch, pins)) return (refuse_reason[12]++, false);
        } else {
            // case 2 (joint_lscm.cpp:750-829): three candidates, the smallest quasi-conformal error wins (ties: snap vi, snap vj, free)
            struct Cand { bool ok = false; double err = 0; std::vector<double> U, Vv; int lm = -1; };
            static thread_local Cand cand[3];
            const int lpv = locmap[pv], lnv = locmap[nv];
            for (int t = 0; t < 3; t++) {
                setup(t == 0 ? la : (t == 1 ? lb : own));
                const int n = patch.n;
                base_pins(n);
                if (t == 0) pins.push_back({lnv + n, 0.0});                     // snapped onto vi: vi -- vj -- nv on one line (:845-905)
                else if (t == 1) pins.push_back({lpv + n, 0.0});                // snapped onto vj: pv -- vi -- vj on one line
                else {                                                          // free on the line through pv, vi, vj, nv (:1060-1085)
                    pins.push_back({lpv + n, 0.0}); pins.push_back({lnv + n, 0.0});
                    if (patch.lm != la && patch.lm != lb) pins.push_back({patch.lm + n, 0.0});
                }
                // the reference flattens all three, picks by error and validates the winner only (joint_lscm.cpp:241): solve without
                // refusing here, remember whether this candidate would pass
                cand[t].ok = solve_joint_flattening(patch, pins);
                cand[t].U = patch.U; cand[t].Vv = patch.Vv; cand[t].lm = patch.lm;
                double e = std::numeric_limits<double>::quiet_NaN();
                if ((int)patch.U.size() == n) e = qc_error_norm(patch.P, patch.pre, patch.U, patch.Vv) + qc_error_norm(patch.Ppost, patch.post, patch.U, patch.Vv);
                cand[t].err = (e == e) ? e : 2147483647.0;                      // NaN -> INT_MAX (:775-776)
            }
            int best = 2;
            if (cand[0].err <= cand[1].err && cand[0].err <= cand[2].err) best = 0;
            else if (cand[1].err <= cand[0].err && cand[1].err <= cand[2].err) best = 1;
            if (!cand[best].ok) return (refuse_reason[13]++, false);
            setup(best == 0 ? la : (best == 1 ? lb : own));
            patch.U = cand[best].U; patch.Vv = cand[best].Vv;
        }
        if (log) {   // decInfo.push_back(data); decIM[FIdx_onering_pre(ii)].push_back(...)  (SSP_collapse_edge.cpp:452-459)
            DecimationLog::Rec r;
            r.first_face = (int)log->face_id.size(); r.n_faces = (int)pre_gid.size();
            r.first_uv = (int)log->U.size(); r.n_loc = patch.n;
            r.la = la; r.lb = lb; r.lm = patch.lm;
            const int k = (int)log->rec.size();
            for (size_t t = 0; t < pre_gid.size(); t++) {
                log->face_id.push_back(pre_gid[t]);
                log->tri.push_back(patch.pre[t]);
                log->face_recs[pre_gid[t]].push_back(k);
            }
            log->U.insert(log->U.end(), patch.U.begin(), patch.U.begin() + patch.n);
            log->V.insert(log->V.end(), patch.Vv.begin(), patch.Vv.begin() + patch.n);
            log->rec.push_back(r);
        }
        // ---- gather the fine points of the pre-collapse 1-ring with their positions in the flattening
        static thread_local std::vector<int> pts;
        static thread_local std::vector<std::array<double, 2>> puv;
        pts.clear(); puv.clear();
        auto take = [&](int f) {
            if (phantom(f)) return;
            for (int p : fpoints[f]) {
                pts.push_back(p);
                const auto& fc = faces[f];
                const auto& w = pbary[p];
                const int l0 = loc[fc[0]], l1 = loc[fc[1]], l2 = loc[fc[2]];
                puv.push_back({w[0] * patch.U[l0] + w[1] * patch.U[l1] + w[2] * patch.U[l2],
                               w[0] * patch.Vv[l0] + w[1] * patch.Vv[l1] + w[2] * patch.Vv[l2]});
            }
            fpoints[f].clear();   // (capacity kept: re-homed points come straight back to the surviving faces)
        };
        for (int f : vfaces[a]) take(f);
        for (int f : vfaces[b]) if (!has(faces[f], a)) take(f);
        // UV of the post patch: the pre UV with the merged vertex at its own / shared unknown (UV_post.row(vi) = UVjoint.row(vi_post))
        // ---- connectivity surgery: b -> a, the two faces on the edge die (a phantom one among them on the boundary)
        for (int i = 0; i < 2; i++) { falive[ef[i]] = 0; if (!phantom(ef[i])) n_alive_faces--; }
        for (int f : vfaces[b]) {
            if (!falive[f]) continue;
            for (int c = 0; c < 3; c++) if (faces[f][c] == b) faces[f][c] = a;
            vfaces[a].push_back(f);
        }
        vfaces[b].clear();
        valive[b] = 0;
        pos[a] = m;
        if (dec_type == 0) for (int i = 0; i < 10; i++) quad[a][i] += quad[b][i];
        version[a]++; version[b]++;
        clean(a);
        for (int w : common) if (w != vinf) clean(w);
        // ---- re-home the points: the face of the flattened post patch in which the smallest barycentric coordinate is largest,
        // clamped to >= 0 and renormalised (src/query_fine_to_coarse.cpp:93-116)
        for (size_t i = 0; i < pts.size(); i++) {
            double bestmin = -1e300, bw[3] = {1, 0, 0};
            int bl = -1;
            for (size_t t = 0; t < patch.post.size(); t++) {
                const auto& g = patch.post[t];
                const double x0 = patch.U[g[0]], y0 = patch.Vv[g[0]], x1 = patch.U[g[1]], y1 = patch.Vv[g[1]], x2 = patch.U[g[2]], y2 = patch.Vv[g[2]];
                const double det = (x1 - x0) * (y2 - y0) - (x2 - x0) * (y1 - y0);
                const double w1 = ((puv[i][0] - x0) * (y2 - y0) - (x2 - x0) * (puv[i][1] - y0)) / det;
                const double w2 = ((x1 - x0) * (puv[i][1] - y0) - (puv[i][0] - x0) * (y1 - y0)) / det;
                const double w0 = 1.0 - w1 - w2, mn = std::min(w0, std::min(w1, w2));
                if (mn > bestmin) { bestmin = mn; bl = (int)t; bw[0] = w0; bw[1] = w1; bw[2] = w2; }
            }
            double sw = 0;
            for (int c2 = 0; c2 < 3; c2++) { bw[c2] = std::max(bw[c2], 0.0); sw += bw[c2]; }
            // patch.post[bl] lists the local vertices in the order of faces[post_gid[bl]] (with a/b -> merged): same slots
            const int p = pts[i], gf = post_gid[bl];
            pface[p] = gf;
            pbary[p] = {bw[0] / sw, bw[1] / sw, bw[2] / sw};
            fpoints[gf].push_back(p);
        }
        // ---- re-cost the edges of the merged vertex's new star (SSP_collapse_edge.cpp:482-520): its own edges get fresh entries, the
        // ring edges opposite to it come back if they had been refused before
        push_star(a);
        for (int f : vfaces[a]) {
            if (!falive[f]) continue;
            for (int c = 0; c < 3; c++) {
                const int x = faces[f][c], y = faces[f][(c + 1) % 3];
                if (x == a || y == a || x == vinf || y == vinf) continue;
                if (unrefuse(x, y)) push_edge(x, y);
            }
        }
        return true;
    }
};

}  // namespace

// absorption_cap_tenths: 0 (default) = the reference's plain greedy order.  > 0: opt-in departure -- a surviving vertex may stand for at
// most (cap / 10) x (#F / tarF) input vertices; edges that would exceed that wait, and the bound doubles only when nothing else can be
// collapsed.  (Shortest-edge-first decimation coarsens densely sampled parts of a mesh far beyond the requested ratio before it
// touches the rest; on ogre.obj the V-cycle factor goes from 0.6 to 0.3 with a cap of 20.)
int decimate_level(const Mesh& fine, int tarF, int dec_type, int absorption_cap_tenths, Mesh& coarse, Csr& P, std::string& err, DecimationLog* log)
{
    const int nV = fine.nV(), nF = fine.nF();
    if (dec_type < 0 || dec_type > 2) { err = "dec_type must be 0 (qslim), 1 (mid-point) or 2 (vertex removal)"; return -1; }
    if (nV < 4 || nF < 4) { err = "mesh too small to decimate"; return -1; }
    Decimator D;
    D.dec_type = dec_type;
    if (log) {
        *log = DecimationLog();
        log->face_recs.assign((size_t)fine.nF(), {});
        D.log = log;
    }
    D.nF_real = nF;
    D.pos.resize(nV);
    for (int i = 0; i < nV; i++) D.pos[i] = {fine.V[3 * i], fine.V[3 * i + 1], fine.V[3 * i + 2]};
    D.faces.resize(nF);
    D.vfaces.assign(nV, {});
    {
        std::vector<int> deg(nV, 0);
        for (size_t t = 0; t < (size_t)nF * 3; t++) {
            const int v = fine.F[t];
            if (v < 0 || v >= nV) { err = "face index out of range"; return -1; }
            deg[v]++;
        }
        for (int v = 0; v < nV; v++) D.vfaces[v].reserve((size_t)deg[v] + 3);   // one allocation per list (+ room for a phantom face / a first merge)
    }
    for (int f = 0; f < nF; f++) {
        for (int c = 0; c < 3; c++) {
            int v = fine.F[3 * f + c];
            D.faces[f][c] = v;
            D.vfaces[v].push_back(f);
        }
        if (D.faces[f][0] == D.faces[f][1] || D.faces[f][1] == D.faces[f][2] || D.faces[f][0] == D.faces[f][2]) { err = "degenerate face"; return -1; }
    }
    // manifoldness (the reference bails out on non-manifold input, src/SSP_decimate.cpp:20-23) and the boundary edges
    std::vector<std::array<int, 2>> bedges;   // directed as in their face
    std::vector<uint64_t> ekeys;
    {
        // The undirected edges in ascending (min, max) order, enumerated at their smaller end point from that vertex's face list -- blocks of
        // vertices side by side on the host threads, outputs concatenated in block order (a global sort of the 3 #F directed edges cost a
        // tenth of the level's time).  An edge met more than twice, or twice in the same direction, is an error; once = boundary.
        const int n_blocks = (int)std::min<long>(256, std::max<long>(1, nV / 4096));
        struct Blk { std::vector<uint64_t> keys; std::vector<std::array<int, 2>> bedges; int bad = 0; };
        std::vector<Blk> blk((size_t)n_blocks);
        parallel_for(n_blocks, 1, [&](long b0, long b1) {
            std::vector<std::array<int, 2>> nb;   // (other end point, 0: the face runs a -> b, 1: b -> a)
            for (long bi = b0; bi < b1; bi++) {
                Blk& B = blk[(size_t)bi];
                const int v0 = (int)((long)nV * bi / n_blocks), v1 = (int)((long)nV * (bi + 1) / n_blocks);
                B.keys.reserve((size_t)(v1 - v0) * 3 + 16);
                for (int a = v0; a < v1; a++) {
                    nb.clear();
                    for (int f : D.vfaces[a]) {
                        const auto& fc = D.faces[f];
                        const int c = fc[0] == a ? 0 : (fc[1] == a ? 1 : 2);
                        const int nx = fc[(c + 1) % 3], pv = fc[(c + 2) % 3];
                        if (nx > a) nb.push_back({nx, 0});
                        if (pv > a) nb.push_back({pv, 1});
                    }
                    std::sort(nb.begin(), nb.end());
                    for (size_t i = 0; i < nb.size();) {
                        size_t j = i;
                        while (j < nb.size() && nb[j][0] == nb[i][0]) j++;
                        if (j - i > 2) { if (!B.bad) B.bad = 1; }
                        else if (j - i == 2 && nb[i][1] == nb[i + 1][1]) { if (!B.bad) B.bad = 2; }
                        if (j - i == 1) B.bedges.push_back(nb[i][1] == 0 ? std::array<int, 2>{a, nb[i][0]} : std::array<int, 2>{nb[i][0], a});
                        B.keys.push_back(((uint64_t)(uint32_t)a << 32) | (uint32_t)nb[i][0]);
                        i = j;
                    }
                }
            }
        });
        size_t nk = 0;
        for (const Blk& B : blk) {
            if (B.bad) { err = B.bad == 1 ? "input mesh is not edge-manifold" : "input mesh is not consistently oriented"; return -1; }
            nk += B.keys.size();
        }
        ekeys.reserve(nk);
        for (const Blk& B : blk) { ekeys.insert(ekeys.end(), B.keys.begin(), B.keys.end()); bedges.insert(bedges.end(), B.bedges.begin(), B.bedges.end()); }
    }
    // close the boundary with a vertex at infinity (igl::connect_boundary_to_infinity, SSP_midpoint.cpp:31): boundary edge (a -> b) of
    // a face gets the phantom face (b, a, inf)
    if (!bedges.empty()) {
        D.vinf = nV;
        const double inf = std::numeric_limits<double>::infinity();
        D.pos.push_back({inf, inf, inf});
        D.vfaces.push_back({});
        std::vector<int> outdeg(nV, 0);
        for (const auto& e : bedges) {
            if (++outdeg[e[0]] > 1) { err = "input mesh has a non-manifold boundary vertex"; return -1; }
            const int f = (int)D.faces.size();
            D.faces.push_back({e[1], e[0], D.vinf});
            D.vfaces[e[1]].push_back(f); D.vfaces[e[0]].push_back(f); D.vfaces[D.vinf].push_back(f);
        }
    }
    const int nVall = (int)D.pos.size(), nFall = (int)D.faces.size();
    D.valive.assign(nVall, 1);
    D.version.assign(nVall, 0);
    D.n_refused.assign(nVall, 0);
    D.falive.assign(nFall, 1);
    D.fpoints.assign(nFall, {});
    D.n_alive_faces = nF;
    if (dec_type == 0) D.init_quadrics();
    for (uint64_t k : ekeys) D.push_edge((int)(k >> 32), (int)(k & 0xffffffffu));
    D.seal_initial();
    // every fine vertex starts as a one-hot barycentric point on one of its faces (src/get_prolong.cpp:23-39)
    D.pface.assign(nV, -1);
    D.pbary.assign(nV, {0, 0, 0});
    for (int f = 0; f < nF; f++)
        for (int c = 0; c < 3; c++) {
            int v = D.faces[f][c];
            if (D.pface[v] < 0) { D.pface[v] = f; D.pbary[v] = {0, 0, 0}; D.pbary[v][c] = 1.0; D.fpoints[f].push_back(v); }
        }
    for (int v = 0; v < nV; v++) if (D.pface[v] < 0) { err = "unreferenced vertex in input mesh"; return -1; }
    // greedy loop (src/SSP_midpoint.cpp:188-220): pop the cheapest live edge; a refused edge leaves the queue until a collapse next to
    // it brings it back (Decimator::collapse); stop as soon as the number of real faces is <= tarF, or when nothing is left.
    const bool capped = absorption_cap_tenths > 0;
    int cap = capped ? std::max(3, (int)std::lround(0.1 * absorption_cap_tenths * (double)nF / (double)std::max(tarF, 1))) : (1 << 30);
    std::vector<int> weight(capped ? nVall : 0, 1);
    std::vector<QEntry> waiting;   // edges held back by the opt-in cap
    while (D.n_alive_faces > tarF) {
        if (D.queue_empty()) {
            if (!capped || waiting.empty() || cap >= (1 << 29)) break;
            cap *= 2;   // nothing else can be collapsed under the current cap: relax it
            for (const QEntry& e : waiting)
                if (D.valive[e.a] && D.valive[e.b]) D.push_edge(e.a, e.b);
            waiting.clear();
            continue;
        }
        QEntry e = D.pop_next();
        if (!D.valive[e.a] || !D.valive[e.b] || D.version[e.a] != e.va || D.version[e.b] != e.vb) continue;
        if (capped && weight[e.a] + weight[e.b] > cap) { waiting.push_back(e); continue; }
        if (D.collapse(e.a, e.b)) { if (capped) weight[e.a] += weight[e.b]; }   // b merges into a
        else D.refuse(e.a, e.b);                     // cost infinity until re-costed (SSP_collapse_edge.cpp:522-531)
    }
    if (std::getenv("SMG_DEC_STATS")) {
        std::fprintf(stderr, "[smg decimate] faces %d -> %d; refusals by site:", nF, D.n_alive_faces);
        for (int i = 0; i < 32; i++) if (D.refuse_reason[i]) std::fprintf(stderr, " #%d:%ld", i, D.refuse_reason[i]);
        std::fprintf(stderr, "\n");
    }
    // compact: drop the vertex at infinity and the phantom faces (SSP_midpoint.cpp:65-70)
    std::vector<int> vmap(nVall, -1);
    int nVc = 0;
    for (int v = 0; v < nV; v++) {
        if (!D.valive[v]) continue;
        D.clean(v);
        bool real = false;
        for (int f : D.vfaces[v]) if (!D.phantom(f)) { real = true; break; }
        if (!real) continue;
        vmap[v] = nVc++;
    }
    coarse.V.resize((size_t)nVc * 3);
    for (int v = 0; v < nV; v++) if (vmap[v] >= 0) { coarse.V[3 * vmap[v]] = D.pos[v].x; coarse.V[3 * vmap[v] + 1] = D.pos[v].y; coarse.V[3 * vmap[v] + 2] = D.pos[v].z; }
    coarse.F.clear();
    for (int f = 0; f < nF; f++) if (D.falive[f]) for (int c = 0; c < 3; c++) coarse.F.push_back(vmap[D.faces[f][c]]);
    if (log) for (int f = 0; f < nF; f++) if (D.falive[f]) log->coarse_face.push_back(f);
    // Every coarse vertex must be interpolated from by somebody: a column of P without a positive entry is a zero row and column
    // of the Galerkin operator, and the smoother divides by its diagonal.  (Possible in principle with mid-point placement: all
    // points of the incident faces may sit on the opposite edges.)  Repair, not in the reference: the fine point of an incident face
    // that lies closest to the orphaned vertex is snapped onto it -- never a point that is the last supporter of another vertex; the
    // supporter counts are kept up to date, and the pass repeats until nothing changes.
    {
        std::vector<int> support(nVall, 0);
        for (int p = 0; p < nV; p++)
            for (int c = 0; c < 3; c++) if (D.pbary[p][c] > 1e-12) support[D.faces[D.pface[p]][c]]++;
        for (int round = 0; round < 8; round++) {
            bool changed = false;
            for (int v = 0; v < nV; v++) {
                if (vmap[v] < 0 || support[v] > 0) continue;
                int best = -1, bf = -1, bc = -1;
                double bd = 1e300;
                for (int f : D.vfaces[v]) {
                    if (!D.falive[f] || D.phantom(f)) continue;
                    int cv = 0;
                    while (D.faces[f][cv] != v) cv++;
                    for (int p : D.fpoints[f]) {
                        if (D.pface[p] != f) continue;
                        V3 q = {0, 0, 0};
                        for (int c = 0; c < 3; c++) q = q + D.pbary[p][c] * D.pos[D.faces[f][c]];
                        const double d = norm(q - D.pos[v]);
                        bool sole = false;   // would moving this point orphan a vertex it supports now?
                        for (int c = 0; c < 3; c++) if (D.pbary[p][c] > 1e-12 && support[D.faces[f][c]] <= 1) sole = true;
                        if (!sole && d < bd) { bd = d; best = p; bf = f; bc = cv; }
                    }
                }
                if (best >= 0) {
                    for (int c = 0; c < 3; c++) if (D.pbary[best][c] > 1e-12) support[D.faces[bf][c]]--;
                    D.pbary[best] = {0, 0, 0}; D.pbary[best][bc] = 1.0;
                    support[v]++;
                    changed = true;
                }
            }
            if (!changed) break;
        }
        for (int v = 0; v < nV; v++)
            if (vmap[v] >= 0 && support[v] == 0) { err = "a coarse vertex is left without any fine vertex interpolating from it"; return -1; }
    }
    // P: three stored entries per row (explicit zeros kept), src/get_prolong.cpp:45-56
    std::vector<int> ptr(nV + 1), col((size_t)nV * 3);
    std::vector<double> val((size_t)nV * 3);
    for (int p = 0; p < nV; p++) {
        ptr[p] = 3 * p;
        const auto& f = D.faces[D.pface[p]];
        double s = 0;
        double w[3];
        for (int c = 0; c < 3; c++) { w[c] = std::max(D.pbary[p][c], 0.0); s += w[c]; }   // clamp + renormalise
        for (int c = 0; c < 3; c++) { col[3 * p + c] = vmap[f[c]]; val[3 * p + c] = w[c] / s; }  // (src/query_fine_to_coarse.cpp:113-116)
    }
    ptr[nV] = 3 * nV;
    P = csr_from_arrays(nV, nVc, ptr.data(), col.data(), val.data());
    return 0;
}

// The reference's query_coarse_to_fine (src/query_coarse_to_fine.cpp:41-140): for the face the point lives in, the latest collapse not
// yet undone whose one-ring held that face; the point's position in that collapse's flattening from the POST one-ring (the face's
// corners, end points standing at the merged vertex); its barycentric coordinates in every face of the PRE one-ring
// (compute_barycentric.cpp: the dot-product form); the face where the smallest coordinate is largest wins (the first one on ties),
// coordinates clamped to >= 0 and renormalised; on to that face's earlier collapses.
void query_coarse_to_fine(const DecimationLog& L, int n, const int* face, const double* bary, int* out_face, double* out_bary)
{
    for (int q = 0; q < n; q++) {
        int f = L.coarse_face[(size_t)face[q]];
        double w[3] = {bary[3 * q], bary[3 * q + 1], bary[3 * q + 2]};
        int upper = (int)L.rec.size();
        while (true) {
            const std::vector<int>& lst = L.face_recs[(size_t)f];
            auto it = std::lower_bound(lst.begin(), lst.end(), upper);
            if (it == lst.begin()) break;
            const int k = *(--it);
            upper = k;
            const DecimationLog::Rec& R = L.rec[(size_t)k];
            const int* fid = &L.face_id[(size_t)R.first_face];
            const std::array<int, 3>* tri = &L.tri[(size_t)R.first_face];
            const double* U = &L.U[(size_t)R.first_uv];
            const double* V = &L.V[(size_t)R.first_uv];
            int t0 = -1;
            for (int t = 0; t < R.n_faces; t++) if (fid[t] == f) { t0 = t; break; }
            if (t0 < 0) break;   // (cannot happen: the record lists the face)
            double pu = 0.0, pv = 0.0;
            for (int c = 0; c < 3; c++) {
                int l = tri[t0][c];
                if (l == R.la || l == R.lb) l = R.lm;   // UV_post: both end points are the merged vertex
                pu += w[c] * U[l]; pv += w[c] * V[l];
            }
            double best = 1.0, bw[3] = {w[0], w[1], w[2]};
            int bt = -1;
            for (int t = 0; t < R.n_faces; t++) {
                const double ax = U[tri[t][0]], ay = V[tri[t][0]];
                const double v0x = U[tri[t][1]] - ax, v0y = V[tri[t][1]] - ay, v1x = U[tri[t][2]] - ax, v1y = V[tri[t][2]] - ay;
                const double v2x = -ax + pu, v2y = -ay + pv;
                const double d00 = v0x * v0x + v0y * v0y, d01 = v0x * v1x + v0y * v1y, d11 = v1x * v1x + v1y * v1y;
                const double d20 = v2x * v0x + v2y * v0y, d21 = v2x * v1x + v2y * v1y;
                const double denom = d00 * d11 - d01 * d01;
                if (!(denom != 0.0)) continue;   // a degenerate flattened triangle (zero area or NaN) locates nothing
                const double bv = (d11 * d20 - d01 * d21) / denom, bwt = (d00 * d21 - d01 * d20) / denom;
                const double bu = 1.0 - (bv + bwt);
                const double dist = -std::min(bu, std::min(bv, bwt));
                if (dist < best) { best = dist; bt = t; bw[0] = bu; bw[1] = bv; bw[2] = bwt; }
            }
            if (bt < 0) break;   // every face is further than a whole triangle away (the reference leaves this case undefined)
            double sw = 0.0;
            for (int c = 0; c < 3; c++) { bw[c] = std::max(0.0, bw[c]); sw += bw[c]; }
            if (!(sw > 0.0)) break;   // nothing to renormalise by: the point stays where it is, with the weights it came with
            for (int c = 0; c < 3; c++) w[c] = bw[c] / sw;
            f = fid[bt];
        }
        out_face[q] = f;
        out_bary[3 * q] = w[0]; out_bary[3 * q + 1] = w[1]; out_bary[3 * q + 2] = w[2];
    }
}

// The reference's query_fine_to_coarse (src/query_fine_to_coarse.cpp:27-120), the other direction: collapses first to last; the point's
// position from the PRE one-ring, located in the POST one-ring (the pre faces without the two on the edge, end points standing at the
// merged vertex) by the same rule.  out_face: face of the coarse mesh.
void query_fine_to_coarse(const DecimationLog& L, int n, const int* face, const double* bary, int* out_face, double* out_bary)
{
    std::vector<int> coarse_of(L.face_recs.size(), -1);
    for (size_t c = 0; c < L.coarse_face.size(); c++) coarse_of[(size_t)L.coarse_face[c]] = (int)c;
    for (int q = 0; q < n; q++) {
        int f = face[q];
        double w[3] = {bary[3 * q], bary[3 * q + 1], bary[3 * q + 2]};
        int lower = -1;
        while (true) {
            const std::vector<int>& lst = L.face_recs[(size_t)f];
            auto it = std::upper_bound(lst.begin(), lst.end(), lower);
            if (it == lst.end()) break;
            const int k = *it;
            lower = k;
            const DecimationLog::Rec& R = L.rec[(size_t)k];
            const int* fid = &L.face_id[(size_t)R.first_face];
            const std::array<int, 3>* tri = &L.tri[(size_t)R.first_face];
            const double* U = &L.U[(size_t)R.first_uv];
            const double* V = &L.V[(size_t)R.first_uv];
            int t0 = -1;
            for (int t = 0; t < R.n_faces; t++) if (fid[t] == f) { t0 = t; break; }
            if (t0 < 0) break;
            double pu = 0.0, pv = 0.0;
            for (int c = 0; c < 3; c++) { pu += w[c] * U[tri[t0][c]]; pv += w[c] * V[tri[t0][c]]; }
            double best = 1.0, bw[3] = {w[0], w[1], w[2]};
            int bt = -1;
            for (int t = 0; t < R.n_faces; t++) {
                int g[3];
                int ends = 0;
                for (int c = 0; c < 3; c++) { g[c] = tri[t][c]; if (g[c] == R.la || g[c] == R.lb) { g[c] = R.lm; ends++; } }
                if (ends == 2) continue;   // one of the two faces on the edge: gone after the collapse
                const double ax = U[g[0]], ay = V[g[0]];
                const double v0x = U[g[1]] - ax, v0y = V[g[1]] - ay, v1x = U[g[2]] - ax, v1y = V[g[2]] - ay;
                const double v2x = -ax + pu, v2y = -ay + pv;
                const double d00 = v0x * v0x + v0y * v0y, d01 = v0x * v1x + v0y * v1y, d11 = v1x * v1x + v1y * v1y;
                const double d20 = v2x * v0x + v2y * v0y, d21 = v2x * v1x + v2y * v1y;
                const double denom = d00 * d11 - d01 * d01;
                if (!(denom != 0.0)) continue;   // a degenerate flattened triangle (zero area or NaN) locates nothing
                const double bv = (d11 * d20 - d01 * d21) / denom, bwt = (d00 * d21 - d01 * d20) / denom;
                const double bu = 1.0 - (bv + bwt);
                const double dist = -std::min(bu, std::min(bv, bwt));
                if (dist < best) { best = dist; bt = t; bw[0] = bu; bw[1] = bv; bw[2] = bwt; }
            }
            if (bt < 0) break;
            double sw = 0.0;
            for (int c = 0; c < 3; c++) { bw[c] = std::max(0.0, bw[c]); sw += bw[c]; }
            if (!(sw > 0.0)) break;   // nothing to renormalise by: the point stays where it is, with the weights it came with
            for (int c = 0; c < 3; c++) w[c] = bw[c] / sw;
            f = fid[bt];
        }
        out_face[q] = coarse_of[(size_t)f];
        out_bary[3 * q] = w[0]; out_bary[3 * q + 1] = w[1]; out_bary[3 * q + 2] = w[2];
    }
}

}  // namespace smg
