// smg_mesh.hpp -- caller-side mesh numerics (host C++): what the reference demos do around the solve
// with libigl (03_mg_solver/main.cpp:29-61, 04_mg_solver_nobd/main.cpp:40-94,
// 05_example_mean_curvature_flow/main.cpp:57-69) plus the mid-point upsampling operator of
// 09_random_subdiv_remesh/main.cpp:46-140 used to generate the ~1M / ~4M-vertex benchmark meshes.
// libigl is not part of the reference checkout (empty submodule): semantics follow SURVEY.md Appendix A.
#pragma once
#include <array>
#include <string>
#include <vector>

#include "smg_sparse.hpp"

namespace smg {

struct Mesh {
    std::vector<double> V;  // nV x 3, row-major
    std::vector<int> F;     // nF x 3, row-major, 0-based
    int nV() const { return (int)(V.size() / 3); }
    int nF() const { return (int)(F.size() / 3); }
};

// igl::read_triangle_mesh for .obj (v / f records; "a", "a/b", "a//c", "a/b/c"; polygons fan-triangulated)
// and the repo's binary .smgm fixtures.  Returns false on failure.
bool read_mesh(const std::string& path, Mesh& m);
bool write_smgm(const std::string& path, const Mesh& m);

std::vector<double> doublearea(const Mesh& m);                 // igl::doublearea
void normalize_unit_area(Mesh& m);                             // src/normalize_unit_area.cpp:3-25
Csr cotmatrix(const Mesh& m);                                  // igl::cotmatrix (negative semi-definite)
enum MassType { MASS_BARYCENTRIC = 0, MASS_VORONOI = 1 };
std::vector<double> massmatrix_diag(const Mesh& m, MassType t); // igl::massmatrix (lumped, diagonal)
std::vector<int> boundary_loop(const Mesh& m);                 // igl::boundary_loop(F, VectorXi): longest loop

// Mid-point upsampling with Loop connectivity.  S is (#V + #E) x #V with NV = S * V; old vertices keep
// their index; new vertex nV + e for the e-th edge of the lexicographically sorted unique (min,max) list.
void midpoint_upsample(int nV, const std::vector<int>& F, Csr& S, std::vector<int>& NF);
// Apply n_sub upsamplings in place; Ps[l-1] (l = 1..n_sub) maps level l -> level l-1, level 0 finest.
void subdivide(Mesh& m, int n_sub, std::vector<Csr>& Ps);

Mesh make_torus(int nu, int nv, double R, double r);           // BASELINE config C5 base mesh

// Assembly plan for a FIXED connectivity: the sparsity of cotmatrix(V, F) and, per stored
// entry, the ordered list of per-face cotangent terms (+/- C(f,e)) that cotmatrix() sums into it; per vertex the ordered
// list of per-corner mass terms.  Replaying the plan on new vertex positions reproduces cotmatrix()/massmatrix_diag()
// bit for bit; the device kernels in smg_device.hip do exactly that.
struct AssemblyPlan {
    int nV = 0, nF = 0;
    Csr pattern;                  // values unused
    std::vector<int> l_ptr;       // nnz + 1
    std::vector<int> l_idx;       // term -> 3*f + e   (index into the per-face cot array)
    std::vector<signed char> l_sgn;
    std::vector<int> m_ptr;       // nV + 1
    std::vector<int> m_idx;       // term -> 3*f + corner (index into the per-face mass array)
    std::vector<int> diag_of;     // per entry: the vertex whose diagonal it is, or -1
};
AssemblyPlan make_assembly_plan(const std::vector<int>& F, int nV);

// What one coarsening step keeps about its collapses when asked to (the reference's decInfo / decIM, src/single_collapse_data.h and
// src/SSP_collapse_edge.cpp:452-459): per successful collapse the faces of the pre-collapse one-ring with their flattened positions.
// The post-collapse one-ring is the same flattening with the two end points standing at the merged vertex (local index lm) and the two
// faces on the edge gone.  Consumed by query_coarse_to_fine (the reference's src/query_coarse_to_fine.cpp).
struct DecimationLog {
    struct Rec { int first_face, n_faces, first_uv, n_loc, la, lb, lm; };
    std::vector<Rec> rec;
    std::vector<int> face_id;                   // concatenated over the records: input-mesh ids of the pre one-ring faces ...
    std::vector<std::array<int, 3>> tri;        // ... and their local vertex triples, corner by corner in the order of the face
    std::vector<double> U, V;                   // concatenated: flattened position of every local vertex
    std::vector<std::vector<int>> face_recs;    // per input face: the records whose pre one-ring held it, ascending (decIM)
    std::vector<int> coarse_face;               // face of the coarse mesh -> input face it descends from (same corner order)
};
// Maps n points of the coarse mesh -- (coarse face, barycentric coordinates) -- onto the mesh the step started from by undoing the
// collapses last to first.  out_face / out_bary: face of the fine mesh and coordinates with respect to its corners.
void query_coarse_to_fine(const DecimationLog& log, int n, const int* face, const double* bary, int* out_face, double* out_bary);
// The other direction (the reference's src/query_fine_to_coarse.cpp): points of the fine mesh onto the coarse mesh; out_face < 0 if the
// walk ends in a face that did not survive (cannot happen for a consistent log).
void query_fine_to_coarse(const DecimationLog& log, int n, const int* face, const double* bary, int* out_face, double* out_bary);

}  // namespace smg
