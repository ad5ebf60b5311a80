// smg_decimate.cpp -- host hierarchy builder behind smg_mg_precompute: one coarsening step
// (the role of the reference's get_prolong(), src/get_prolong.cpp:3-57:  decimate to tarF faces, push every
// fine vertex through the collapse history to barycentric coordinates on the coarse mesh, assemble P with
// exactly three stored entries per row).
//
// Same contract as the reference (greedy collapse: shortest edge first with mid-point placement, dec_type 1, or end-point
// placement, dec_type 2; smallest quadric error first with the quadric's minimiser as placement, dec_type 0 "qslim"; link-condition and fold-over rejection; every fine vertex carried along as (face, barycentric);
// P with exactly three stored entries per row, non-negative, rows summing to 1).  The per-collapse re-parameterisation:
//   * the reference's construction -- the 1-rings before and after the collapse are flattened JOINTLY by least-squares
//     conformal maps with a shared boundary ring (src/joint_lscm.cpp), a collapse whose flattening flips, folds over or
//     degenerates is rejected (check_valid_UV_lscm), and every point of the 1-ring is located in the flattened post patch
//     by the largest-minimum-barycentric rule (src/query_fine_to_coarse.cpp:93-116).  Written from the formulation (energy,
//     pins, checks), not from the reference's code; there is no reference binary to compare with, so it is validated by
//     invariants and by the V-cycle convergence it yields;
//   * collapses touching the boundary go through the same flattening on the open 1-ring (natural boundary conditions); the
//     reference closes the boundary with an "infinity vertex" and has two more LSCM cases for it (src/joint_lscm.cpp:
//     642-1131), which are not restated;
//   * the libigl-internal edge-flap bookkeeping, the randomised variants and the coarse-to-fine queries of the
//     remeshing demos are not restated.
#include <algorithm>
#include <cstdlib>
#include <array>
#include <cmath>
#include <cstdint>
#include <queue>
#include <string>
#include <unordered_map>
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

// ---- joint conformal flattening of the pre- and post-collapse 1-rings (interior collapses) -------------------------------
// The reference's successive self-parameterisation maps a point through a collapse by flattening the 1-ring before and
// after the collapse JOINTLY (shared boundary ring, least-squares conformal energy of both patches, two pins) and locating
// the point's UV position in the post patch (src/joint_lscm.cpp:483-555 flatten(), :557-640 interior case;
// src/query_fine_to_coarse.cpp:85-116).  This is the same construction, written from the formulation:
//   minimise  E = sum over {pre, post}  1/2 (u^T K u + v^T K v) - Area(u, v),   K = cotangent stiffness of the 3D patch,
//   subject to  uv(a) = (0,0), uv(b) = (1,0);   unknowns: ring vertices (shared), a, b (pre only), m (post only).
struct Patch {
    int n = 0;                                // local vertices: ring..., a = n-3, b = n-2, m = n-1
    std::vector<std::array<int, 3>> pre, post;  // local faces
    std::vector<V3> P;                        // local positions (a, b at their pre positions, m at the merged position)
    std::vector<double> U, Vv;                // solution
};

static void add_conformal_energy(const Patch& pt, const std::vector<std::array<int, 3>>& F, std::vector<double>& Q)
{
    const int n = pt.n, N = 2 * n;
    for (const auto& f : F) {
        for (int c = 0; c < 3; c++) {
            const int i = f[c], j = f[(c + 1) % 3], k = f[(c + 2) % 3];   // edge (i,j), opposite corner k
            const V3 e1 = pt.P[i] - pt.P[k], e2 = pt.P[j] - pt.P[k];
            const double cr = norm(cross(e1, e2));
            const double w = cr > 0 ? 0.5 * dot(e1, e2) / cr : 0.0;       // 1/2 cot(angle at k)
            for (int d = 0; d < 2; d++) {                                   // K (+)= w (x_i - x_j)^2 for u and v
                const int I = i + d * n, J = j + d * n;
                Q[(size_t)I * N + I] += w; Q[(size_t)J * N + J] += w;
                Q[(size_t)I * N + J] -= w; Q[(size_t)J * N + I] -= w;
            }
            // - 2 S:  Area = 1/2 sum over directed face edges (u_i v_j - u_j v_i)
            Q[(size_t)i * N + (j + n)] -= 0.5; Q[(size_t)(j + n) * N + i] -= 0.5;
            Q[(size_t)j * N + (i + n)] += 0.5; Q[(size_t)(i + n) * N + j] += 0.5;
        }
    }
}

static bool solve_joint_flattening(Patch& pt)
{
    const int n = pt.n, N = 2 * n;
    // (scratch vectors live across calls: a collapse is a few microseconds of arithmetic, allocation would be a good part of it)
    static thread_local std::vector<double> Q, M, rhs, x;
    static thread_local std::vector<int> freei;
    Q.assign((size_t)N * N, 0.0);
    add_conformal_energy(pt, pt.pre, Q);
    add_conformal_energy(pt, pt.post, Q);
    const int a = n - 3, b = n - 2;
    // pins: u_a = 0, v_a = 0, u_b = 1, v_b = 0
    freei.clear();
    for (int i = 0; i < N; i++) if (i != a && i != b && i != a + n && i != b + n) freei.push_back(i);
    const int m = (int)freei.size();
    M.resize((size_t)m * m); rhs.resize(m);
    for (int r = 0; r < m; r++) {
        for (int c = 0; c < m; c++) M[(size_t)r * m + c] = Q[(size_t)freei[r] * N + freei[c]];
        rhs[r] = -Q[(size_t)freei[r] * N + b] * 1.0;   // only u_b = 1 is non-zero among the pins
    }
    // dense Cholesky (the pinned conformal energy is positive definite on a valid patch)
    for (int j = 0; j < m; j++) {
        double d = M[(size_t)j * m + j];
        for (int k = 0; k < j; k++) d -= M[(size_t)j * m + k] * M[(size_t)j * m + k];
        if (!(d > 1e-14)) return false;
        d = std::sqrt(d);
        M[(size_t)j * m + j] = d;
        for (int i = j + 1; i < m; i++) {
            double sx = M[(size_t)i * m + j];
            for (int k = 0; k < j; k++) sx -= M[(size_t)i * m + k] * M[(size_t)j * m + k];
            M[(size_t)i * m + j] = sx / d;
        }
    }
    for (int i = 0; i < m; i++) { double sx = rhs[i]; for (int k = 0; k < i; k++) sx -= M[(size_t)i * m + k] * rhs[k]; rhs[i] = sx / M[(size_t)i * m + i]; }
    for (int i = m - 1; i >= 0; i--) { double sx = rhs[i]; for (int k = i + 1; k < m; k++) sx -= M[(size_t)k * m + i] * rhs[k]; rhs[i] = sx / M[(size_t)i * m + i]; }
    x.assign(N, 0.0);
    x[b] = 1.0;
    for (int r = 0; r < m; r++) x[freei[r]] = rhs[r];
    pt.U.assign(x.begin(), x.begin() + n);
    pt.Vv.assign(x.begin() + n, x.end());
    for (double v : x) if (!(v == v)) return false;
    // every flattened face of both patches must keep its orientation (reference check_valid_UV_lscm: signed area >= 1e-10)
    auto oriented = [&](const std::vector<std::array<int, 3>>& F) {
        for (const auto& f : F) {
            const double ar = (pt.U[f[1]] - pt.U[f[0]]) * (pt.Vv[f[2]] - pt.Vv[f[0]]) - (pt.Vv[f[1]] - pt.Vv[f[0]]) * (pt.U[f[2]] - pt.U[f[0]]);
            if (!(ar > 1e-10)) return false;
        }
        return true;
    };
    if (!(oriented(pt.pre) && oriented(pt.post))) return false;
    // no fold-over: the flattened angles around the collapsing vertices / the merged vertex must not exceed 2 pi
    // (reference check_valid_UV_lscm, src/joint_lscm.cpp:330-392), and no flattened sliver: quality
    // 4 sqrt(3) area / (l0^2 + l1^2 + l2^2) >= 0.01 (:394-478)
    const int la = n - 3, lb = n - 2, lm = n - 1;
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
    return checks(pt.pre, la, lb) && checks(pt.post, lm, -1);
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
    std::vector<V3> pos;
    std::vector<char> valive;
    std::vector<int> version;
    std::vector<std::array<int, 3>> faces;
    std::vector<char> falive;
    std::vector<std::vector<int>> vfaces;   // incident faces (may hold dead ones; filtered on access)
    std::vector<std::vector<int>> fpoints;  // fine points living on each face
    std::vector<int> pface;
    std::vector<std::array<double, 3>> pbary;
    std::priority_queue<QEntry> pq;
    // The edges of the input mesh -- most of what the queue ever holds -- are sorted once and consumed front to back; the heap only
    // takes the edges created or re-offered later.  pop_next() returns the better of the two fronts: the same sequence one heap
    // holding everything would produce (the order of QEntry is total), at a fraction of the cache misses.
    std::vector<QEntry> initial;
    size_t ihead = 0;
    bool filling = true;
    void seal_initial()
    {
        std::sort(initial.begin(), initial.end(), [](const QEntry& x, const QEntry& y) { return y < x; });   // best first
        filling = false;
    }
    bool queue_empty() const { return pq.empty() && ihead == initial.size(); }
    QEntry pop_next()
    {
        if (pq.empty() || (ihead < initial.size() && !(initial[ihead] < pq.top()))) return initial[ihead++];
        QEntry e = pq.top();
        pq.pop();
        return e;
    }
    int n_alive_faces = 0;
    int dec_type = 1;
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
        for (size_t f = 0; f < faces.size(); f++) {
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
        for (int f : vfaces[a]) if (falive[f] && has(faces[f], b)) { if (n < 3) out[n] = f; n++; }
        return n;
    }
    bool on_boundary_big(int v)   // valence > 64
    {
        std::unordered_map<int, int> cnt;
        for (int f : vfaces[v]) if (falive[f]) for (int c = 0; c < 3; c++) if (faces[f][c] != v) cnt[faces[f][c]]++;
        for (auto& kv : cnt) if (kv.second == 1) return true;
        return false;
    }
    bool on_boundary(int v)
    {
        // v is a boundary vertex iff one of its edges has a single incident face
        // (a one-ring holds a dozen vertices: a flat list beats a hash map, and this runs twice per attempted collapse)
        int nbv[64], nbc[64], nn = 0;
        for (int f : vfaces[v]) {
            if (!falive[f]) continue;
            for (int c = 0; c < 3; c++) {
                const int w = faces[f][c];
                if (w == v) continue;
                int i = 0;
                while (i < nn && nbv[i] != w) i++;
                if (i == nn) { if (nn == 64) return on_boundary_big(v); nbv[nn] = w; nbc[nn] = 0; nn++; }
                nbc[i]++;
            }
        }
        for (int i = 0; i < nn; i++) if (nbc[i] == 1) return true;
        return false;
    }
    void push_edge(int a, int b)
    {
        if (a > b) std::swap(a, b);
        const QEntry e{dec_type == 0 ? qem(a, b, nullptr) : norm(pos[a] - pos[b]), a, b, version[a], version[b]};
        if (filling) initial.push_back(e); else pq.push(e);
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
    // try to collapse (a,b); returns true on success
    bool collapse(int a, int b)
    {
        clean(a); clean(b);
        int ef[3];
        const int nef = edge_faces(a, b, ef);
        if (nef < 1 || nef > 2) return false;
        const bool edge_is_boundary = (nef == 1);
        const bool ba = on_boundary(a), bb = on_boundary(b);
        if (ba && bb && !edge_is_boundary) return false;  // interior chord between two boundary vertices
        // link condition: common neighbours == vertices opposite to the edge
        static thread_local std::vector<int> na, nb, common;
        na.clear(); nb.clear(); common.clear();
        for (int f : vfaces[a]) for (int c = 0; c < 3; c++) if (faces[f][c] != a) na.push_back(faces[f][c]);
        for (int f : vfaces[b]) for (int c = 0; c < 3; c++) if (faces[f][c] != b) nb.push_back(faces[f][c]);
        std::sort(na.begin(), na.end()); na.erase(std::unique(na.begin(), na.end()), na.end());
        std::sort(nb.begin(), nb.end()); nb.erase(std::unique(nb.begin(), nb.end()), nb.end());
        std::set_intersection(na.begin(), na.end(), nb.begin(), nb.end(), std::back_inserter(common));
        if ((int)common.size() != nef) return false;
        for (int i = 0; i < nef; i++) {
            const auto& f = faces[ef[i]];
            int opp = f[0] != a && f[0] != b ? f[0] : (f[1] != a && f[1] != b ? f[1] : f[2]);
            if (!std::binary_search(common.begin(), common.end(), opp)) return false;
        }
        if ((int)na.size() + (int)nb.size() - 2 - nef < 3) return false;  // would leave a vertex of valence < 3
        // placement
        V3 m;
        if (dec_type == 2) m = pos[a];                       // vertex removal: keep an end point
        else if (dec_type == 0 && ba == bb) (void)qem(a, b, &m);  // quadric-optimal placement (boundary: see below)
        else if (ba && !bb) m = pos[a];                      // keep the boundary where it is
        else if (bb && !ba) m = pos[b];
        else m = 0.5 * (pos[a] + pos[b]);                    // mid-point (dec_type 1)
        // fold-over / degeneracy test on the surviving faces
        for (int pass = 0; pass < 2; pass++) {
            const int v = pass == 0 ? a : b;
            for (int f : vfaces[v]) {
                const auto& fc = faces[f];
                if (has(fc, a) && has(fc, b)) continue;
                V3 p0 = pos[fc[0]], p1 = pos[fc[1]], p2 = pos[fc[2]];
                V3 n0 = cross(p1 - p0, p2 - p0);
                V3 q0 = (fc[0] == v) ? m : p0, q1 = (fc[1] == v) ? m : p1, q2 = (fc[2] == v) ? m : p2;
                V3 n1 = cross(q1 - q0, q2 - q0);
                const double l0 = norm(n0), l1 = norm(n1);
                if (!(l1 > 1e-14 * (1.0 + l0))) return false;
                if (dot(n0, n1) < 0.2 * l0 * l1) return false;
                // reject slivers: height/longest-edge quality
                const double e = std::max(norm(q1 - q0), std::max(norm(q2 - q1), norm(q0 - q2)));
                if (l1 < 0.02 * e * e) return false;
            }
        }
        // ---- interior collapse: joint conformal flattening of the 1-ring before / after (reject the collapse if invalid)
        // (collapses touching the boundary use the same construction on the open 1-ring: the conformal energy has natural
        //  boundary conditions; when one end point is a boundary vertex the merged vertex sits on that end point)
        static thread_local Patch patch;
        static thread_local std::vector<int> pre_gid, post_gid;              // global face ids of patch.pre / patch.post
        patch.pre.clear(); patch.post.clear(); patch.P.clear(); patch.U.clear(); patch.Vv.clear(); patch.n = 0;
        pre_gid.clear(); post_gid.clear();
        // global vertex -> local: a scratch array over all vertices, touched entries reset on every way out of this function
        if (locmap.size() != pos.size()) locmap.assign(pos.size(), -1);
        static thread_local std::vector<int> loc_touched;
        loc_touched.clear();
        struct LocGuard {
            std::vector<int>& m; std::vector<int>& touched;
            int& operator[](int v) { if (m[v] < 0) touched.push_back(v); return m[v]; }
            ~LocGuard() { for (int v : touched) m[v] = -1; }
        } loc{locmap, loc_touched};
        {
            static thread_local std::vector<int> ringv;
            ringv.clear();
            for (int v : na) if (v != b) ringv.push_back(v);
            for (int v : nb) if (v != a && !std::binary_search(na.begin(), na.end(), v)) ringv.push_back(v);
            for (int v : ringv) { loc[v] = (int)patch.P.size(); patch.P.push_back(pos[v]); }
            const int la = (int)patch.P.size(); patch.P.push_back(pos[a]);
            const int lb = la + 1; patch.P.push_back(pos[b]);
            // The merged vertex is its own unknown of the flattening when it is a new point (mid-point placement).  When it IS one of
            // the end points (vertex removal; a boundary vertex that stays put) it shares that end point's unknown: otherwise the
            // two would be flattened to different places and the surviving vertex's own record -- the fine vertex sitting exactly
            // on the coarse vertex -- would be re-located into the interior of a face (coarse vertices nobody interpolates from,
            // i.e. zero rows in the Galerkin operator, were the symptom).
            const bool m_is_a = (m.x == pos[a].x && m.y == pos[a].y && m.z == pos[a].z);
            const bool m_is_b = !m_is_a && (m.x == pos[b].x && m.y == pos[b].y && m.z == pos[b].z);
            int lm = lb + 1;
            if (m_is_a) lm = la;
            else if (m_is_b) lm = lb;
            else patch.P.push_back(m);
            patch.n = (int)patch.P.size();
            loc[a] = la; loc[b] = lb;
            auto add_faces = [&](int v) {
                for (int f : vfaces[v]) {
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
            add_faces(a); add_faces(b);
            if (!solve_joint_flattening(patch)) return false;
        }
        // ---- gather the fine points of the pre-collapse 1-ring with their positions
        static thread_local std::vector<int> pts;
        static thread_local std::vector<std::array<double, 2>> puv;   // position in the joint flattening (interior collapses)
        pts.clear(); puv.clear();
        auto take = [&](int f) {
            for (int p : fpoints[f]) {
                pts.push_back(p);
                {
                    const auto& fc = faces[f];
                    const auto& w = pbary[p];
                    const int l0 = loc[fc[0]], l1 = loc[fc[1]], l2 = loc[fc[2]];
                    puv.push_back({w[0] * patch.U[l0] + w[1] * patch.U[l1] + w[2] * patch.U[l2],
                                   w[0] * patch.Vv[l0] + w[1] * patch.Vv[l1] + w[2] * patch.Vv[l2]});
                }
            }
            fpoints[f].clear();   // (capacity kept: re-homed points come straight back to the surviving faces)
        };
        for (int f : vfaces[a]) take(f);
        for (int f : vfaces[b]) if (!has(faces[f], a)) take(f);
        // ---- connectivity surgery: b -> a, the edge faces die
        for (int i = 0; i < nef; i++) { falive[ef[i]] = 0; n_alive_faces--; }
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
        for (int w : common) clean(w);
        // ---- re-home the points on the post-collapse star of a (closest-point re-parameterisation)
        {
            // locate every point in the flattened post patch: the face in which its smallest barycentric coordinate is
            // largest, clamped to >= 0 and renormalised (src/query_fine_to_coarse.cpp:93-116)
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
                // patch.post[bl] lists the local vertices in the order of faces[post_gid[bl]] (with a/b -> m): same slots
                const int p = pts[i], gf = post_gid[bl];
                pface[p] = gf;
                pbary[p] = {bw[0] / sw, bw[1] / sw, bw[2] / sw};
                fpoints[gf].push_back(p);
            }
        }
        push_star(a);
        return true;
    }
};

}  // namespace

int decimate_level(const Mesh& fine, int tarF, int dec_type, Mesh& coarse, Csr& P, std::string& err)
{
    const int nV = fine.nV(), nF = fine.nF();
    if (dec_type < 0 || dec_type > 2) { err = "dec_type must be 0 (qslim), 1 (mid-point) or 2 (vertex removal)"; return -1; }
    if (nV < 4 || nF < 4) { err = "mesh too small to decimate"; return -1; }
    Decimator D;
    D.dec_type = dec_type;
    D.pos.resize(nV);
    for (int i = 0; i < nV; i++) D.pos[i] = {fine.V[3 * i], fine.V[3 * i + 1], fine.V[3 * i + 2]};
    D.valive.assign(nV, 1);
    D.version.assign(nV, 0);
    D.faces.resize(nF);
    D.falive.assign(nF, 1);
    D.vfaces.assign(nV, {});
    D.fpoints.assign(nF, {});
    D.n_alive_faces = nF;
    for (int f = 0; f < nF; f++) {
        for (int c = 0; c < 3; c++) {
            int v = fine.F[3 * f + c];
            if (v < 0 || v >= nV) { err = "face index out of range"; return -1; }
            D.faces[f][c] = v;
            D.vfaces[v].push_back(f);
        }
        if (D.faces[f][0] == D.faces[f][1] || D.faces[f][1] == D.faces[f][2] || D.faces[f][0] == D.faces[f][2]) { err = "degenerate face"; return -1; }
    }
    if (dec_type == 0) D.init_quadrics();
    // manifoldness (the reference bails out on non-manifold input, src/SSP_decimate.cpp:20-23)
    {
        // sorted edge keys: equal keys are adjacent (a hash map over 3 #F edges cost a tenth of the whole decimation)
        std::vector<uint64_t> ekeys;
        ekeys.reserve((size_t)nF * 3);
        for (int f = 0; f < nF; f++)
            for (int c = 0; c < 3; c++) {
                int a = D.faces[f][c], b = D.faces[f][(c + 1) % 3];
                ekeys.push_back(((uint64_t)(uint32_t)std::min(a, b) << 32) | (uint32_t)std::max(a, b));
            }
        std::sort(ekeys.begin(), ekeys.end());
        for (size_t i = 0; i < ekeys.size();) {
            size_t j = i;
            while (j < ekeys.size() && ekeys[j] == ekeys[i]) j++;
            if (j - i > 2) { err = "input mesh is not edge-manifold"; return -1; }
            D.push_edge((int)(ekeys[i] >> 32), (int)(ekeys[i] & 0xffffffffu));
            i = j;
        }
    }
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
    // greedy loop (src/SSP_midpoint.cpp:188-220): pop the cheapest valid edge until #faces <= tarF
    // Rejected edges are parked and offered again once the queue runs dry after at least one success (their
    // validity can change when a neighbouring collapse rewires the link).
    // Absorption cap (not in the reference): shortest-edge-first decimation coarsens the densely sampled parts of a mesh far
    // beyond the global ratio before it touches the rest (ogre.obj: coarse triangles holding 50 fine vertices next to regions
    // left untouched), and a smooth error bump inside such a triangle is invisible to the coarse level: the V-cycle stalls at
    // 0.6 per cycle there.  A surviving vertex may therefore stand for at most `cap` input vertices (twice the average of the
    // requested ratio); edges that would exceed it wait, and the cap is relaxed only when nothing else is left to collapse.
    // SMG_DECIMATE_CAP=0 switches it off (the reference's behaviour); another value sets the factor in tenths (default 20 = 2.0).
    const int cap_mode = [] { const char* v = std::getenv("SMG_DECIMATE_CAP"); return v && *v ? std::atoi(v) : 20; }();
    int cap = cap_mode ? std::max(3, (int)std::lround(0.1 * cap_mode * (double)nF / (double)std::max(tarF, 1))) : (1 << 30);
    std::vector<int> weight(nV, 1);
    std::vector<QEntry> parked;
    bool progressed = false;
    while (D.n_alive_faces > tarF) {
        if (D.queue_empty()) {
            if (parked.empty()) break;
            if (!progressed) {
                if (cap >= (1 << 29)) break;
                cap *= 2;   // nothing moved under the current cap: relax it
            }
            for (const QEntry& e : parked)
                if (D.valive[e.a] && D.valive[e.b]) D.push_edge(e.a, e.b);
            parked.clear();
            progressed = false;
            continue;
        }
        QEntry e = D.pop_next();
        if (!D.valive[e.a] || !D.valive[e.b] || D.version[e.a] != e.va || D.version[e.b] != e.vb) continue;
        if (weight[e.a] + weight[e.b] > cap) { parked.push_back(e); continue; }
        if (D.collapse(e.a, e.b)) { progressed = true; weight[e.a] += weight[e.b]; }   // b merges into a
        else parked.push_back(e);
    }
    // compact
    std::vector<int> vmap(nV, -1);
    int nVc = 0;
    for (int v = 0; v < nV; v++) {
        if (!D.valive[v]) continue;
        D.clean(v);
        if (D.vfaces[v].empty()) continue;
        vmap[v] = nVc++;
    }
    coarse.V.resize((size_t)nVc * 3);
    for (int v = 0; v < nV; v++) if (vmap[v] >= 0) { coarse.V[3 * vmap[v]] = D.pos[v].x; coarse.V[3 * vmap[v] + 1] = D.pos[v].y; coarse.V[3 * vmap[v] + 2] = D.pos[v].z; }
    coarse.F.clear();
    for (int f = 0; f < nF; f++) if (D.falive[f]) for (int c = 0; c < 3; c++) coarse.F.push_back(vmap[D.faces[f][c]]);
    // Every coarse vertex must be interpolated from by somebody: a column of P without a positive entry is a zero row and column
    // of the Galerkin operator, and the smoother divides by its diagonal.  (Possible in principle with mid-point placement: all
    // points of the incident faces may sit on the opposite edges.)  Repair: the fine point of an incident face that lies closest
    // to the orphaned vertex is snapped onto it.
    {
        std::vector<char> used(nV, 0);
        for (int p = 0; p < nV; p++)
            for (int c = 0; c < 3; c++) if (D.pbary[p][c] > 1e-12) used[D.faces[D.pface[p]][c]] = 1;
        for (int v = 0; v < nV; v++) {
            if (vmap[v] < 0 || used[v]) continue;
            int best = -1, bf = -1, bc = -1;
            double bd = 1e300;
            for (int f : D.vfaces[v]) {
                if (!D.falive[f]) continue;
                int cv = 0;
                while (D.faces[f][cv] != v) cv++;
                for (int p : D.fpoints[f]) {
                    // position of the point on the coarse face
                    V3 q = {0, 0, 0};
                    for (int c = 0; c < 3; c++) q = q + D.pbary[p][c] * D.pos[D.faces[f][c]];
                    const double d = norm(q - D.pos[v]);
                    // do not orphan another vertex: the point must not be the only user of a vertex it currently leans on fully
                    bool sole = false;
                    for (int c = 0; c < 3; c++) if (D.pbary[p][c] > 0.999999) sole = true;
                    if (!sole && d < bd) { bd = d; best = p; bf = f; bc = cv; }
                }
            }
            if (best >= 0) { D.pface[best] = bf; D.pbary[best] = {0, 0, 0}; D.pbary[best][bc] = 1.0; used[v] = 1; }
        }
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

}  // namespace smg
