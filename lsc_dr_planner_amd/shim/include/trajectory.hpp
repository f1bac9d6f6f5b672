// M-segment Bernstein trajectory container with the reference's interface (include/trajectory.hpp:9-70,
// src/trajectory.cpp).  Output type of TrajOptimizer::solve, input type `initial_traj`.  No ROS markers.
#pragma once
#include <polynomial.hpp>
#include <sp_const.hpp>

namespace DynamicPlanning {
template <typename T>
class Segment {
public:
    std::vector<T> control_points;
    double segment_time = 0;
    T startPoint() const { return control_points[0]; }
    T lastPoint() const { return control_points.back(); }
    T operator[](int idx) const { return control_points[idx]; }
    T& operator[](int idx) { return control_points[idx]; }
    // Control points of the piece [t_normalized_0, t_normalized_f] of this segment (src/trajectory.cpp:15-49): the row vectors of
    // the control points times B A B^-1, t -> a t + b, evaluated in double and rounded by T's constructor; segment_time scales with
    // a.  (The reference's generic template logs "Wrong usage"; like there, only 3-component point types are meaningful.)
    Segment<T> subSegment(double t_normalized_0, double t_normalized_f, const Eigen::MatrixXd& B, const Eigen::MatrixXd& B_inv) const {
        const double b = t_normalized_0, a = t_normalized_f - t_normalized_0;
        const int n = (int)control_points.size() - 1;
        Eigen::MatrixXd A = Eigen::MatrixXd::Zero(n + 1, n + 1);
        for (int i = 0; i < n + 1; i++)
            for (int j = 0; j < i + 1; j++) A(i, j) = nChoosek(i, j) * std::pow(a, j) * std::pow(b, i - j);
        Segment<T> sub_segment;
        sub_segment.segment_time = segment_time * a;
        sub_segment.control_points.resize(n + 1);
        std::vector<double> out(3 * (n + 1));
        for (int k = 0; k < 3; k++) {
            std::vector<double> c(n + 1), t1(n + 1), t2(n + 1);
            for (int i = 0; i < n + 1; i++) c[i] = control_points[i](k);
            for (int j = 0; j < n + 1; j++) {
                t1[j] = 0;
                for (int l = 0; l < n + 1; l++) t1[j] += c[l] * B(l, j);
            }
            for (int j = 0; j < n + 1; j++) {
                t2[j] = 0;
                for (int l = 0; l < n + 1; l++) t2[j] += t1[l] * A(l, j);
            }
            for (int j = 0; j < n + 1; j++) {
                double v = 0;
                for (int l = 0; l < n + 1; l++) v += t2[l] * B_inv(l, j);
                out[3 * j + k] = v;
            }
        }
        for (int i = 0; i < n + 1; i++) sub_segment.control_points[i] = T((float)out[3 * i], (float)out[3 * i + 1], (float)out[3 * i + 2]);
        return sub_segment;
    }
};

template <typename T>
class Trajectory {
public:
    Trajectory() : M(0), n(0) {}
    Trajectory(size_t _M, size_t _n, double dt) : M(_M), n(_n) {  // src/trajectory.cpp:68-76
        segments.resize(M);
        for (size_t m = 0; m < M; m++) {
            segments[m].control_points.resize(n + 1);
            segments[m].segment_time = dt;
        }
    }
    void planConstVelTraj(T current_state, T velocity) {  // :79-91
        if (n == 0 || segments.empty()) throw std::invalid_argument("[Trajectory] n = 0");
        double time = 0;
        for (size_t m = 0; m < M; m++)
            for (size_t i = 0; i < n + 1; i++) {
                segments[m][i] = current_state + velocity * (float)time;
                time += segments[m].segment_time / n;
            }
    }
    int size() const { return (int)segments.size(); }
    bool empty() const { return segments.empty(); }
    void clear() { M = 0; n = 0; segments.clear(); }
    // Bernstein evaluation with the reference's conventions (src/trajectory.cpp:111-148): times past the horizon by
    // less than SP_EPSILON_FLOAT clamp to the end, anything else out of range yields the default point; the sum is
    // accumulated in T (float32 for point3d) exactly like `point + control_point * b_i_n`.
    T getPointAt(double time) const {
        int m;
        double u;
        if (!locate(time, m, u)) return T();
        T acc;
        for (size_t i = 0; i <= n; i++) acc = acc + segments[m].control_points[i] * (float)getBernsteinBasis((int)n, (int)i, u);
        return acc;
    }
    State getStateAt(double time) const {  // :156-170
        State state;
        state.position = getPointAt(time);
        Trajectory<T> dtraj = derivative();
        state.velocity = dtraj.getPointAt(time);
        Trajectory<T> ddtraj = dtraj.derivative();
        state.acceleration = ddtraj.getPointAt(time);
        return state;
    }
    T startPoint() const { return segments[0][0]; }
    T lastPoint() const { return segments[M - 1][n]; }
    Trajectory<T> derivative() const {  // :183-199
        Trajectory<T> dtraj;
        dtraj.M = M;
        dtraj.n = n - 1;
        dtraj.segments.resize(M);
        for (size_t m = 0; m < M; m++) {
            dtraj.segments[m].segment_time = segments[m].segment_time;
            dtraj.segments[m].control_points.resize(n + 1);
            for (size_t i = 0; i < n; i++)
                dtraj.segments[m].control_points[i] =
                    (segments[m].control_points[i + 1] - segments[m].control_points[i]) * (float)(n / segments[m].segment_time);
        }
        return dtraj;
    }
    Trajectory<T> coordinateTransform(double downwash) {  // :207-219
        Trajectory<T> t = *this;
        for (size_t m = 0; m < M; m++)
            for (size_t i = 0; i < n + 1; i++) t.segments[m].control_points[i].z() /= (float)downwash;
        return t;
    }
    Segment<T> operator[](int idx) const { return segments[idx]; }
    Segment<T>& operator[](int idx) { return segments[idx]; }

private:
    // segment index and normalised time of `time`; false when out of range
    bool locate(double time, int& m, double& u) const {
        if (time < 0) return false;
        double end = 0;
        for (size_t idx = 0; idx < M; idx++) {
            end += segments[idx].segment_time;
            if (time < end) {
                m = (int)idx;
                u = 1 - (end - time) / segments[idx].segment_time;
                return true;
            }
        }
        if (time < end + SP_EPSILON_FLOAT) {
            m = (int)M - 1;
            u = 1.0;
            return true;
        }
        return false;
    }
    size_t M;
    size_t n;
    std::vector<Segment<T>> segments;
};

typedef Trajectory<point3d> traj_t;
}  // namespace DynamicPlanning
