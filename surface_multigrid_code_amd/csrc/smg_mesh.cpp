// smg_mesh.cpp -- see smg_mesh.hpp.
#include "smg_mesh.hpp"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <unordered_map>

namespace smg {

static bool ends_with(const std::string& s, const char* suf)
{
    size_t n = std::strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

static bool read_smgm(const std::string& path, Mesh& m)
{
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    char magic[4];
    uint32_t ver = 0;
    int32_t nv = 0, nf = 0;
    bool ok = std::fread(magic, 1, 4, f) == 4 && std::memcmp(magic, "SMGM", 4) == 0 &&
              std::fread(&ver, 4, 1, f) == 1 && ver == 1 && std::fread(&nv, 4, 1, f) == 1 &&
              std::fread(&nf, 4, 1, f) == 1 && nv >= 0 && nf >= 0;
    if (ok) {
        m.V.resize((size_t)nv * 3);
        m.F.resize((size_t)nf * 3);
        ok = std::fread(m.V.data(), 8, m.V.size(), f) == m.V.size() &&
             std::fread(m.F.data(), 4, m.F.size(), f) == m.F.size();
    }
    std::fclose(f);
    return ok;
}

bool write_smgm(const std::string& path, const Mesh& m)
{
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    uint32_t ver = 1;
    int32_t nv = m.nV(), nf = m.nF();
    std::fwrite("SMGM", 1, 4, f);
    std::fwrite(&ver, 4, 1, f);
    std::fwrite(&nv, 4, 1, f);
    std::fwrite(&nf, 4, 1, f);
    std::fwrite(m.V.data(), 8, m.V.size(), f);
    std::fwrite(m.F.data(), 4, m.F.size(), f);
    return std::fclose(f) == 0;
}

static bool read_obj(const std::string& path, Mesh& m)
{
    FILE* f = std::fopen(path.c_str(), "r");
    if (!f) return false;
    m.V.clear(); m.F.clear();
    char line[4096];
    std::vector<int> poly;
    while (std::fgets(line, sizeof(line), f)) {
        if (line[0] == 'v' && (line[1] == ' ' || line[1] == '\t')) {
            double x, y, z;
            if (std::sscanf(line + 2, "%lf %lf %lf", &x, &y, &z) == 3) { m.V.push_back(x); m.V.push_back(y); m.V.push_back(z); }
        } else if (line[0] == 'f' && (line[1] == ' ' || line[1] == '\t')) {
            poly.clear();
            char* p = line + 2;
            while (*p) {
                while (*p == ' ' || *p == '\t') p++;
                if (*p == '\0' || *p == '\n' || *p == '\r') break;
                char* end = nullptr;
                long idx = std::strtol(p, &end, 10);
                if (end == p) break;
                if (idx < 0) idx = (long)m.nV() + idx + 1;  // relative indices
                poly.push_back((int)idx - 1);
                p = end;
                while (*p && *p != ' ' && *p != '\t' && *p != '\n' && *p != '\r') p++;  // skip /vt/vn
            }
            for (size_t k = 1; k + 1 < poly.size(); k++) { m.F.push_back(poly[0]); m.F.push_back(poly[k]); m.F.push_back(poly[k + 1]); }
        }
    }
    std::fclose(f);
    return m.nV() > 0;
}

bool read_mesh(const std::string& path, Mesh& m)
{
    if (ends_with(path, ".smgm")) return read_smgm(path, m);
    return read_obj(path, m);
}

static inline void sub3(const double* a, const double* b, double* o) { o[0] = a[0] - b[0]; o[1] = a[1] - b[1]; o[2] = a[2] - b[2]; }
static inline double norm3(const double* a) { return std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }

std::vector<double> doublearea(const Mesh& m)
{
    int nF = m.nF();
    std::vector<double> dA(nF);
    for (int f = 0; f < nF; f++) {
        const double* a = &m.V[3 * (size_t)m.F[3 * f]];
        const double* b = &m.V[3 * (size_t)m.F[3 * f + 1]];
        const double* c = &m.V[3 * (size_t)m.F[3 * f + 2]];
        double u[3], v[3], w[3];
        sub3(b, a, u); sub3(c, a, v);
        w[0] = u[1] * v[2] - u[2] * v[1];
        w[1] = u[2] * v[0] - u[0] * v[2];
        w[2] = u[0] * v[1] - u[1] * v[0];
        dA[f] = norm3(w);
    }
    return dA;
}

void normalize_unit_area(Mesh& m)
{
    std::vector<double> dA = doublearea(m);
    double s = 0.0;
    for (double a : dA) s += a;
    double scale = std::sqrt(s / 2);
    int n = m.nV();
    for (double& v : m.V) v /= scale;
    double mx = 0, my = 0, zmin = n ? m.V[2] : 0.0;
    for (int i = 0; i < n; i++) { mx += m.V[3 * i]; my += m.V[3 * i + 1]; zmin = std::min(zmin, m.V[3 * i + 2]); }
    mx /= n; my /= n;
    for (int i = 0; i < n; i++) { m.V[3 * i] -= mx; m.V[3 * i + 1] -= my; m.V[3 * i + 2] -= zmin; }
}

static void edge_lengths(const Mesh& m, int f, double* l)
{
    const double* a = &m.V[3 * (size_t)m.F[3 * f]];
    const double* b = &m.V[3 * (size_t)m.F[3 * f + 1]];
    const double* c = &m.V[3 * (size_t)m.F[3 * f + 2]];
    double d[3];
    sub3(b, c, d); l[0] = norm3(d);
    sub3(c, a, d); l[1] = norm3(d);
    sub3(a, b, d); l[2] = norm3(d);
}

Csr cotmatrix(const Mesh& m)
{
    int n = m.nV(), nF = m.nF();
    std::vector<double> dA = doublearea(m);
    // triplets (row, col, val) in a per-row bucket: count first
    std::vector<int> cnt(n + 1, 0);
    for (int f = 0; f < nF; f++)
        for (int c = 0; c < 3; c++) cnt[m.F[3 * f + c] + 1] += 4;  // each corner is in 2 edges, each edge adds 2 entries to its row
    for (int i = 0; i < n; i++) cnt[i + 1] += cnt[i];
    std::vector<int> tc(cnt[n]);
    std::vector<double> tv(cnt[n]);
    std::vector<int> next(cnt.begin(), cnt.end() - 1);
    static const int es[3] = {1, 2, 0}, ed[3] = {2, 0, 1};
    for (int f = 0; f < nF; f++) {
        double l[3];
        edge_lengths(m, f, l);
        double l2[3] = {l[0] * l[0], l[1] * l[1], l[2] * l[2]};
        double C[3] = {(l2[1] + l2[2] - l2[0]) / dA[f] / 4.0, (l2[2] + l2[0] - l2[1]) / dA[f] / 4.0,
                       (l2[0] + l2[1] - l2[2]) / dA[f] / 4.0};
        for (int e = 0; e < 3; e++) {
            int s = m.F[3 * f + es[e]], d = m.F[3 * f + ed[e]];
            int q;
            q = next[s]++; tc[q] = d; tv[q] = C[e];
            q = next[d]++; tc[q] = s; tv[q] = C[e];
            q = next[s]++; tc[q] = s; tv[q] = -C[e];
            q = next[d]++; tc[q] = d; tv[q] = -C[e];
        }
    }
    return csr_from_arrays(n, n, cnt.data(), tc.data(), tv.data());
}

std::vector<double> massmatrix_diag(const Mesh& m, MassType t)
{
    int n = m.nV(), nF = m.nF();
    std::vector<double> M(n, 0.0), dA = doublearea(m);
    for (int f = 0; f < nF; f++) {
        double q[3];
        if (t == MASS_BARYCENTRIC) {
            q[0] = q[1] = q[2] = dA[f] / 6.0;
        } else {
            double l[3];
            edge_lengths(m, f, l);
            double cs[3] = {(l[2] * l[2] + l[1] * l[1] - l[0] * l[0]) / (l[1] * l[2] * 2.0),
                            (l[0] * l[0] + l[2] * l[2] - l[1] * l[1]) / (l[2] * l[0] * 2.0),
                            (l[1] * l[1] + l[0] * l[0] - l[2] * l[2]) / (l[0] * l[1] * 2.0)};
            double b[3] = {cs[0] * l[0], cs[1] * l[1], cs[2] * l[2]};
            double bs = b[0] + b[1] + b[2];
            double p[3] = {b[0] / bs * (dA[f] * 0.5), b[1] / bs * (dA[f] * 0.5), b[2] / bs * (dA[f] * 0.5)};
            q[0] = (p[1] + p[2]) * 0.5; q[1] = (p[2] + p[0]) * 0.5; q[2] = (p[0] + p[1]) * 0.5;
            for (int c = 0; c < 3; c++)
                if (cs[c] < 0)
                    for (int cc = 0; cc < 3; cc++) q[cc] = (cc == c ? 0.25 : 0.125) * dA[f];
        }
        for (int c = 0; c < 3; c++) M[m.F[3 * f + c]] += q[c];
    }
    return M;
}

std::vector<int> boundary_loop(const Mesh& m)
{
    int nF = m.nF();
    std::unordered_map<uint64_t, int> he;
    he.reserve((size_t)nF * 3);
    auto key = [](int a, int b) { return ((uint64_t)(uint32_t)a << 32) | (uint32_t)b; };
    for (int f = 0; f < nF; f++)
        for (int c = 0; c < 3; c++) he[key(m.F[3 * f + c], m.F[3 * f + (c + 1) % 3])] = 1;
    std::map<int, int> nxt;
    for (auto& kv : he) {
        int a = (int)(kv.first >> 32), b = (int)(kv.first & 0xffffffffu);
        if (!he.count(key(b, a))) nxt[a] = b;
    }
    std::vector<int> best;
    std::unordered_map<int, char> seen;
    for (auto& kv : nxt) {
        int s = kv.first;
        if (seen.count(s)) continue;
        std::vector<int> loop;
        int v = s;
        while (!seen.count(v)) {
            seen[v] = 1;
            loop.push_back(v);
            auto it = nxt.find(v);
            if (it == nxt.end()) break;
            v = it->second;
        }
        if (loop.size() > best.size()) best.swap(loop);
    }
    return best;
}

void midpoint_upsample(int nV, const std::vector<int>& F, Csr& S, std::vector<int>& NF)
{
    int nF = (int)(F.size() / 3);
    std::vector<uint64_t> keys((size_t)nF * 3);
    auto key = [](int a, int b) { int lo = std::min(a, b), hi = std::max(a, b); return ((uint64_t)(uint32_t)lo << 32) | (uint32_t)hi; };
    for (int i = 0; i < 3; i++)
        for (int f = 0; f < nF; f++) keys[(size_t)i * nF + f] = key(F[3 * f + i], F[3 * f + (i + 1) % 3]);
    std::vector<uint64_t> uniq(keys);
    std::sort(uniq.begin(), uniq.end());
    uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
    int nE = (int)uniq.size();
    auto eidx = [&](uint64_t k) { return (int)(std::lower_bound(uniq.begin(), uniq.end(), k) - uniq.begin()); };
    NF.resize((size_t)nF * 12);
    for (int f = 0; f < nF; f++) {
        int m01 = nV + eidx(keys[f]), m12 = nV + eidx(keys[(size_t)nF + f]), m20 = nV + eidx(keys[(size_t)2 * nF + f]);
        int f0 = F[3 * f], f1 = F[3 * f + 1], f2 = F[3 * f + 2];
        int* a = &NF[3 * (size_t)f];               a[0] = f0;  a[1] = m01; a[2] = m20;
        int* b = &NF[3 * ((size_t)nF + f)];        b[0] = f1;  b[1] = m12; b[2] = m01;
        int* c = &NF[3 * ((size_t)2 * nF + f)];    c[0] = f2;  c[1] = m20; c[2] = m12;
        int* d = &NF[3 * ((size_t)3 * nF + f)];    d[0] = m12; d[1] = m20; d[2] = m01;
    }
    S.nr = nV + nE; S.nc = nV;
    S.ptr.resize(S.nr + 1);
    S.col.resize((size_t)nV + 2 * (size_t)nE);
    S.val.resize(S.col.size());
    for (int v = 0; v < nV; v++) { S.ptr[v] = v; S.col[v] = v; S.val[v] = 1.0; }
    for (int e = 0; e < nE; e++) {
        int q = nV + 2 * e;
        S.ptr[nV + e] = q;
        S.col[q] = (int)(uniq[e] >> 32);       S.val[q] = 0.5;
        S.col[q + 1] = (int)(uniq[e] & 0xffffffffu); S.val[q + 1] = 0.5;
    }
    S.ptr[S.nr] = nV + 2 * nE;
}

void subdivide(Mesh& m, int n_sub, std::vector<Csr>& Ps)
{
    std::vector<Csr> ops;
    for (int s = 0; s < n_sub; s++) {
        Csr S;
        std::vector<int> NF;
        midpoint_upsample(m.nV(), m.F, S, NF);
        std::vector<double> NV((size_t)S.nr * 3);
        for (int i = 0; i < S.nr; i++)
            for (int c = 0; c < 3; c++) {
                double acc = 0.0;
                for (int p = S.ptr[i]; p < S.ptr[i + 1]; p++) acc += S.val[p] * m.V[3 * (size_t)S.col[p] + c];
                NV[3 * (size_t)i + c] = acc;
            }
        m.V.swap(NV);
        m.F.swap(NF);
        ops.push_back(std::move(S));
    }
    Ps.clear();
    for (int s = n_sub - 1; s >= 0; s--) Ps.push_back(std::move(ops[s]));
}

AssemblyPlan make_assembly_plan(const std::vector<int>& F, int nV)
{
    AssemblyPlan P;
    const int nF = (int)(F.size() / 3);
    P.nV = nV; P.nF = nF;
    // same bucketing and insertion order as cotmatrix()
    std::vector<int> cnt(nV + 1, 0);
    for (int f = 0; f < nF; f++) for (int c = 0; c < 3; c++) cnt[F[3 * f + c] + 1] += 4;
    for (int i = 0; i < nV; i++) cnt[i + 1] += cnt[i];
    struct Trip { int col, term; signed char sgn; };
    std::vector<Trip> tr(cnt[nV]);
    std::vector<int> next(cnt.begin(), cnt.end() - 1);
    static const int es[3] = {1, 2, 0}, ed[3] = {2, 0, 1};
    for (int f = 0; f < nF; f++)
        for (int e = 0; e < 3; e++) {
            const int s = F[3 * f + es[e]], d = F[3 * f + ed[e]], t = 3 * f + e;
            tr[next[s]++] = {d, t, 1};
            tr[next[d]++] = {s, t, 1};
            tr[next[s]++] = {s, t, -1};
            tr[next[d]++] = {d, t, -1};
        }
    P.pattern.nr = P.pattern.nc = nV;
    P.pattern.ptr.assign(nV + 1, 0);
    for (int i = 0; i < nV; i++) {
        std::stable_sort(tr.begin() + cnt[i], tr.begin() + cnt[i + 1], [](const Trip& a, const Trip& b) { return a.col < b.col; });
        for (int p = cnt[i]; p < cnt[i + 1]; p++) {
            if (p == cnt[i] || tr[p].col != tr[p - 1].col) {   // new stored entry (csr_from_arrays merges duplicates in this order)
                P.l_ptr.push_back((int)P.l_idx.size());
                P.pattern.col.push_back(tr[p].col);
                P.diag_of.push_back(tr[p].col == i ? i : -1);
            }
            P.l_idx.push_back(tr[p].term);
            P.l_sgn.push_back(tr[p].sgn);
        }
        P.pattern.ptr[i + 1] = (int)P.pattern.col.size();
    }
    P.l_ptr.push_back((int)P.l_idx.size());
    P.pattern.val.assign(P.pattern.col.size(), 0.0);
    // mass: massmatrix_diag() adds the corner terms in face order
    P.m_ptr.assign(nV + 1, 0);
    for (int f = 0; f < nF; f++) for (int c = 0; c < 3; c++) P.m_ptr[F[3 * f + c] + 1]++;
    for (int i = 0; i < nV; i++) P.m_ptr[i + 1] += P.m_ptr[i];
    P.m_idx.resize(P.m_ptr[nV]);
    std::vector<int> mn(P.m_ptr.begin(), P.m_ptr.end() - 1);
    for (int f = 0; f < nF; f++) for (int c = 0; c < 3; c++) P.m_idx[mn[F[3 * f + c]]++] = 3 * f + c;
    return P;
}

Mesh make_torus(int nu, int nv, double R, double r)
{
    Mesh m;
    const double PI = 3.14159265358979323846;
    m.V.resize((size_t)nu * nv * 3);
    for (int i = 0; i < nu; i++)
        for (int j = 0; j < nv; j++) {
            double u = i * (2 * PI / nu), w = j * (2 * PI / nv);
            double* p = &m.V[3 * ((size_t)i * nv + j)];
            p[0] = (R + r * std::cos(w)) * std::cos(u);
            p[1] = (R + r * std::cos(w)) * std::sin(u);
            p[2] = r * std::sin(w);
        }
    auto idx = [&](int i, int j) { return (i % nu) * nv + (j % nv); };
    for (int i = 0; i < nu; i++)
        for (int j = 0; j < nv; j++) {
            int a = idx(i, j), b = idx(i + 1, j), c = idx(i + 1, j + 1), d = idx(i, j + 1);
            m.F.push_back(a); m.F.push_back(b); m.F.push_back(c);
            m.F.push_back(a); m.F.push_back(c); m.F.push_back(d);
        }
    return m;
}

}  // namespace smg
