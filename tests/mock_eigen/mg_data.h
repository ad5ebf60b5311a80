// tests/mock_eigen/mg_data.h -- stand-in for the reference's src/mg_data.h in the adapter's syntax / smoke check: the per-level
// container with the fields the solve path uses (reference src/mg_data.h:11-19; the dead colouring fields are left out).
// Written for the check; an integrator compiles examples/smg_eigen_adapter.cpp against the reference's own header.
#pragma once
#include <Eigen/Core>
#include <Eigen/Sparse>
struct mg_data {
    Eigen::MatrixXd V;
    Eigen::MatrixXi F;
    Eigen::SparseMatrix<double> P_full, A;
    Eigen::VectorXd A_diag;
    Eigen::SparseMatrix<double> P, PT;
};
