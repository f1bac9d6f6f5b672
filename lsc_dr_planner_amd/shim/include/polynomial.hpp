// The part of the reference's include/polynomial.hpp the trajectory path uses: binomials, the Bernstein basis and the
// Bernstein->monomial matrix B that TrajPlanner builds and hands to TrajOptimizer (src/traj_planner.cpp:10,27).
#pragma once
#include <cmath>
#include <eigen_standin.hpp>
#include <sp_const.hpp>

namespace DynamicPlanning {
static inline int nChoosek(int n, int k) {  // include/polynomial.hpp:9-20
    if (k > n) return 0;
    if (k * 2 > n) k = n - k;
    if (k == 0) return 1;
    int result = n;
    for (int i = 2; i <= k; i++) {
        result *= (n - i + 1);
        result /= i;
    }
    return result;
}

static inline double getBernsteinBasis(int n, int i, double t_normalized) {  // :22-24
    return nChoosek(n, i) * std::pow(t_normalized, i) * std::pow(1 - t_normalized, n - i);
}

static inline void buildBernsteinBasis(int n, Eigen::MatrixXd& B, Eigen::MatrixXd& B_inv) {  // :281-294
    B = Eigen::MatrixXd::Zero(n + 1, n + 1);
    for (int i = 0; i < n + 1; i++)
        for (int j = 0; j < n + 1; j++)
            B(i, j) = (j >= i) ? nChoosek(n, i) * nChoosek(n - i, n - j) * std::pow(-1, j - i) : 0;
    B_inv = B.inverse();
}
}  // namespace DynamicPlanning
