// Minimal Eigen::MatrixXd stand-in so that the reference's public signature
//     TrajOptimizer(const Param&, const Mission&, const Eigen::MatrixXd& B)      (include/traj_optimizer.hpp:25)
// can be kept where Eigen is absent (this image, the GPU box).  If the real Eigen is on the include path it is used.
#pragma once
#if __has_include(<Eigen/Dense>)
#include <Eigen/Dense>
#else
#include <cstddef>
#include <vector>
namespace Eigen {
class MatrixXd {
public:
    MatrixXd() : r_(0), c_(0) {}
    MatrixXd(int rows, int cols) : r_(rows), c_(cols), d_((size_t)rows * cols, 0.0) {}
    static MatrixXd Zero(int rows, int cols) { return MatrixXd(rows, cols); }
    int rows() const { return r_; }
    int cols() const { return c_; }
    double& operator()(int i, int j) { return d_[(size_t)i * c_ + j]; }
    double operator()(int i, int j) const { return d_[(size_t)i * c_ + j]; }
    MatrixXd transpose() const {
        MatrixXd t(c_, r_);
        for (int i = 0; i < r_; i++)
            for (int j = 0; j < c_; j++) t(j, i) = (*this)(i, j);
        return t;
    }
    MatrixXd operator*(const MatrixXd& o) const {
        MatrixXd p(r_, o.c_);
        for (int i = 0; i < r_; i++)
            for (int j = 0; j < o.c_; j++) {
                double s = 0;
                for (int k = 0; k < c_; k++) s += (*this)(i, k) * o(k, j);
                p(i, j) = s;
            }
        return p;
    }
    MatrixXd inverse() const {  // Gauss-Jordan with partial pivoting (small matrices only)
        int n = r_;
        MatrixXd a = *this, inv(n, n);
        for (int i = 0; i < n; i++) inv(i, i) = 1.0;
        for (int c = 0; c < n; c++) {
            int p = c;
            for (int r = c + 1; r < n; r++)
                if ((a(r, c) < 0 ? -a(r, c) : a(r, c)) > (a(p, c) < 0 ? -a(p, c) : a(p, c))) p = r;
            for (int j = 0; j < n; j++) {
                double t = a(c, j); a(c, j) = a(p, j); a(p, j) = t;
                t = inv(c, j); inv(c, j) = inv(p, j); inv(p, j) = t;
            }
            double piv = a(c, c);
            for (int j = 0; j < n; j++) { a(c, j) /= piv; inv(c, j) /= piv; }
            for (int r = 0; r < n; r++) {
                if (r == c) continue;
                double f = a(r, c);
                if (f == 0) continue;
                for (int j = 0; j < n; j++) { a(r, j) -= f * a(c, j); inv(r, j) -= f * inv(c, j); }
            }
        }
        return inv;
    }

private:
    int r_, c_;
    std::vector<double> d_;
};
}  // namespace Eigen
#endif
