// examples/08_subdiv_remesh.cpp -- the reference's 08_subdiv_remesh/main.cpp:113-166 on libsmg (no viewer): decimate the mesh to 500
// faces by mid-point collapse keeping the record of the collapses, mid-point-upsample the coarse mesh twice, carry every vertex of the
// upsampled mesh back onto the input surface through the bijection of the successive self-parameterisation, and write the three meshes.
// Host only: no GPU is touched.  The reference checks the three files it writes in (08_subdiv_remesh/output_s{0,1,2}.obj);
// tests/test_host_logic.py compares what this program writes with them (tests/golden/bunny_remesh_500.npz).
//
//   g++ -std=c++17 -O2 examples/08_subdiv_remesh.cpp -Iinclude -Lsurface_multigrid_code_amd/lib -lsmg -o 08_subdiv_remesh
//   ./08_subdiv_remesh tests/golden/meshes/bunny.smgm <output directory>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "smg.h"

// igl::upsample's numbering (the reference's main.cpp:65-72 calls it): the new vertex of edge (F(i,j), F(i,j+1)) gets the next free index
// the first time a face i = 0, 1, ... / corner j = 0, 1, 2 meets that edge, and face i becomes (v0,m0,m2), (v1,m1,m0), (m0,m1,m2), (m1,v2,m2).
// (The convention is the one the faces of the reference's output_s1/_s2.obj follow -- tests/golden/bunny_remesh_500.npz;
// smg_mesh_midpoint_upsample numbers the new vertices by sorted edges instead, which gives the same meshes in another order.)
static void upsample_igl(int nV, const std::vector<int>& F, std::vector<std::pair<int, int>>& edge_of_new, std::vector<int>& NF)
{
    std::map<std::pair<int, int>, int> ids;
    edge_of_new.clear();
    NF.clear();
    for (size_t i = 0; i < F.size() / 3; i++) {
        int v[3] = {F[3 * i], F[3 * i + 1], F[3 * i + 2]}, m[3];
        for (int j = 0; j < 3; j++) {
            const int a = v[j], b = v[(j + 1) % 3];
            const std::pair<int, int> key(a < b ? a : b, a < b ? b : a);
            auto it = ids.find(key);
            if (it == ids.end()) { it = ids.emplace(key, nV + (int)edge_of_new.size()).first; edge_of_new.push_back(key); }
            m[j] = it->second;
        }
        const int sub[12] = {v[0], m[0], m[2], v[1], m[1], m[0], m[0], m[1], m[2], m[1], v[2], m[2]};
        NF.insert(NF.end(), sub, sub + 12);
    }
}

#define CHECK(call) do { if ((call) != SMG_OK) { std::fprintf(stderr, "%s: %s\n", #call, smg_last_error()); return 1; } } while (0)

int main(int argc, char* argv[])
{
    const char* path = argc > 1 ? argv[1] : "tests/golden/meshes/bunny.smgm";
    const std::string outdir = argc > 2 ? argv[2] : ".";
    double* VO = nullptr; int* FO = nullptr; int nVO = 0, nFO = 0;
    CHECK(smg_mesh_read(path, &VO, &nVO, &FO, &nFO));
    std::printf("original mesh: |V| %d, |F|: %d\n", nVO, nFO);

    // decimate the input mesh using SSP (main.cpp:126-136: tarF = 500, dec_type = 1 mid-point)
    const int tarF = 500, dec_type = 1;
    smg_hierarchy* mg = nullptr;
    CHECK(smg_mg_precompute_logged(VO, nVO, FO, nFO, (float)tarF / (float)nFO, /*nVCoarsest=*/200, dec_type, 0.0f, /*keep_log=*/1, &mg));
    int nV = 0, nF = 0;
    CHECK(smg_level_get_mesh(mg, 1, &nV, &nF, nullptr, nullptr));
    std::vector<double> V((size_t)nV * 3); std::vector<int> F((size_t)nF * 3);
    CHECK(smg_level_get_mesh(mg, 1, &nV, &nF, V.data(), F.data()));
    std::printf("coarse mesh: |V| %d, |F|: %d\n", nV, nF);

    // upsample the coarse mesh (main.cpp:45-111): per vertex of the upsampled mesh its barycentric coordinates in a coarse face
    const int num_subdivs = 2;
    std::vector<std::vector<int>> level_F(1, F);                      // faces after 0, 1, 2 upsamplings
    std::vector<int> level_nV(1, nV);
    std::vector<std::map<int, double>> rows((size_t)nV);              // S: upsampled vertex -> {coarse vertex: weight}
    for (int v = 0; v < nV; v++) rows[v][v] = 1.0;
    for (int it = 0; it < num_subdivs; it++) {
        const std::vector<int> Fc = level_F.back();
        const int nVc = level_nV.back();
        std::vector<std::pair<int, int>> edges;
        std::vector<int> NF;
        upsample_igl(nVc, Fc, edges, NF);
        for (const auto& e : edges) {   // the mid-point: half of each end point's row
            std::map<int, double> r;
            for (const auto& kv : rows[e.first]) r[kv.first] += 0.5 * kv.second;
            for (const auto& kv : rows[e.second]) r[kv.first] += 0.5 * kv.second;
            rows.push_back(r);
        }
        level_F.push_back(NF);
        level_nV.push_back(nVc + (int)edges.size());
    }
    const int nQ = (int)rows.size();
    std::vector<int> qface(nQ);
    std::vector<double> qbary((size_t)nQ * 3, 0.0);
    for (int q = 0; q < nQ; q++) {
        int found = -1;   // the first coarse face that holds all the vertices this point depends on (find_row_with_elements)
        for (int f = 0; f < nF && found < 0; f++) {
            bool all = true;
            for (const auto& kv : rows[q]) all = all && (F[3 * f] == kv.first || F[3 * f + 1] == kv.first || F[3 * f + 2] == kv.first);
            if (all) found = f;
        }
        if (found < 0) { std::fprintf(stderr, "no coarse face for point %d\n", q); return 1; }
        qface[q] = found;
        for (int c = 0; c < 3; c++) { auto it = rows[q].find(F[3 * found + c]); if (it != rows[q].end()) qbary[3 * q + c] = it->second; }
    }

    // query_coarse_to_fine (main.cpp:146) and the subdivided vertex locations (:148-154)
    std::vector<int> bf(nQ);
    std::vector<double> bc((size_t)nQ * 3);
    CHECK(smg_query_coarse_to_fine(mg, 1, nQ, qface.data(), qbary.data(), bf.data(), bc.data()));
    std::vector<double> SV((size_t)nQ * 3, 0.0);
    for (int q = 0; q < nQ; q++)
        for (int c = 0; c < 3; c++)
            for (int d = 0; d < 3; d++) SV[3 * q + d] += bc[3 * q + c] * VO[3 * FO[3 * bf[q] + c] + d];

    // split the subdivided meshes into levels (:156-165)
    for (int it = 0; it <= num_subdivs; it++) {
        const std::string name = outdir + "/output_s" + std::to_string(it) + ".obj";
        FILE* fp = std::fopen(name.c_str(), "w");
        if (!fp) { std::fprintf(stderr, "cannot write %s\n", name.c_str()); return 1; }
        for (int v = 0; v < level_nV[it]; v++) std::fprintf(fp, "v %.15g %.15g %.15g\n", SV[3 * v], SV[3 * v + 1], SV[3 * v + 2]);
        const std::vector<int>& Fi = level_F[it];
        for (size_t f = 0; f < Fi.size() / 3; f++) std::fprintf(fp, "f %d %d %d\n", Fi[3 * f] + 1, Fi[3 * f + 1] + 1, Fi[3 * f + 2] + 1);
        std::fclose(fp);
        std::printf("%s: |V| %d, |F|: %d\n", name.c_str(), level_nV[it], (int)Fi.size() / 3);
    }
    smg_hierarchy_destroy(mg);
    smg_free(VO); smg_free(FO);
    return 0;
}
