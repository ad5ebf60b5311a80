// tests/mock_eigen/min_quad_with_fixed_mg.h -- stand-in for the reference's src/min_quad_with_fixed_mg.h in the adapter's syntax /
// smoke check: only the data struct (reference src/min_quad_with_fixed_mg.h:22-29).  The function declarations come from the
// adapter's own definitions here; an integrator compiles the adapter against the reference's own header instead.
#pragma once
#include <vector>

#include <Eigen/Core>
#include <Eigen/Sparse>

#include "mg_data.h"
struct min_quad_with_fixed_mg_data {
    int n;
    Eigen::VectorXi known, unknown;
    Eigen::SparseMatrix<double> LHS, Auk;
};
