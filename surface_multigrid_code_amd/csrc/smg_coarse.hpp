// smg_coarse.hpp -- coarseSolve() for coarsest levels too large for a dense inverse (reference src/mg_VCycle.cpp:181-201,
// src/min_quad_with_fixed_mg.cpp:47-48, :253-254: Eigen::SimplicialLDLT factors whatever size mg_precompute's nVCoarsest left).
//
// Up to smg_hierarchy_set_coarse_dense_max unknowns (default 16384) the coarsest matrix is inverted on the device and the solve is a bandwidth-bound
// dense product (smg_device.hip: launch_spd_inverse, k_sym_gemv_*): 8 n^2 bytes.  Above, the reference's own method: a sparse Cholesky
// factorisation  P A P^T = L L^T  -- nested-dissection ordering by breadth-first bisection, up-looking numeric factorisation, both on the
// host as part of the precompute (the reference factors on the host, too) -- and the two triangular solves on the device, each ONE launch:
// one wavefront per row in dependency order, a row waits for the rows it reads by polling their values in HBM (bounded spins; rows only wait
// for rows of lower launch index, which the dispatcher has started before them), products summed in a fixed order: deterministic.
// Memory: 2 x 12 bytes per entry of L (rows for the forward solve, columns for the backward one): O(n log n) on meshes.
#pragma once
#include <vector>

#include "smg_sparse.hpp"

namespace smg {

struct SparseChol {
    int n = 0;
    std::vector<int> perm;        // new -> old (fill-reducing order)
    std::vector<int> parent;      // elimination tree
    // strict lower triangle of L by rows (forward solve) and by columns (backward solve), indices ascending; diagonal apart
    std::vector<int> rptr, rcol, cptr, crow;
    std::vector<double> rval, cval, diag;
    long nnzL() const { return rptr.empty() ? 0 : (long)rptr.back(); }
};

// Nested-dissection order of the pattern of A (square, structurally symmetric): new -> old.
std::vector<int> nested_dissection_order(const Csr& A, int leaf = 64);
// Symbolic + numeric Cholesky of the symmetric positive definite A (both triangles stored).  With reuse_symbolic (F.perm / F.parent of a
// previous call on the same pattern) only the numeric phase runs.  Returns false when a pivot is not positive.
bool sparse_cholesky(const Csr& A, SparseChol& F, bool reuse_symbolic = false);

}  // namespace smg
