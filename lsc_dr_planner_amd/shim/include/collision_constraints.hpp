// The container half of the reference's CollisionConstraints (include/collision_constraints.hpp:17-46,98-176,
// src/collision_constraints.cpp:5-8,32-59,385-394,482-543): storage lscs[oi][m][i], sfcs[m], setters used by the LSC
// generators (src/traj_planner.cpp:581-736) and the getters the QP builder reads.  SFC expansion over the octomap EDT,
// RViz markers and convex-hull meshing are outside the trajectory-QP path.
#pragma once
#include <param.hpp>
#include <mission.hpp>
#include <set>
#include <sp_const.hpp>
#include <trajectory.hpp>

namespace DynamicPlanning {
// LSC = {c | (c - c_obs).dot(normal_vector) - d > 0}
class LSC {
public:
    LSC() = default;
    LSC(const point3d& _obs_control_point, const point3d& _normal_vector, double _d)
        : obs_control_point(_obs_control_point), normal_vector(_normal_vector), d(_d) {}
    point3d obs_control_point;
    point3d normal_vector;
    double d = 0;
};
typedef std::vector<LSC> LSCs;

class Box {
public:
    point3d box_min;
    point3d box_max;
    Box() = default;
    Box(const point3d& _box_min, const point3d& _box_max) : box_min(_box_min), box_max(_box_max) {}
    LSCs convertToLSCs(int dim) const {  // src/collision_constraints.cpp:37-59
        point3d zero_point(0, 0, 0);
        LSCs lscs;
        lscs.resize(2 * dim);
        for (int i = 0; i < dim; i++) {
            point3d normal_vector_min = zero_point, normal_vector_max = zero_point;
            normal_vector_min(i) = 1;
            normal_vector_max(i) = -1;
            lscs[2 * i] = LSC(zero_point, normal_vector_min, box_min(i));
            lscs[2 * i + 1] = LSC(zero_point, normal_vector_max, -box_max(i));
        }
        return lscs;
    }
    bool isPointInBox(const point3d& point) const {
        return point.x() > box_min.x() - SP_EPSILON_FLOAT && point.y() > box_min.y() - SP_EPSILON_FLOAT &&
               point.z() > box_min.z() - SP_EPSILON_FLOAT && point.x() < box_max.x() + SP_EPSILON_FLOAT &&
               point.y() < box_max.y() + SP_EPSILON_FLOAT && point.z() < box_max.z() + SP_EPSILON_FLOAT;
    }
};

typedef std::vector<std::vector<std::vector<LSC>>> RSFCs;  // [obs_idx][segment_idx][control_point_idx]
typedef std::vector<Box> SFCs;                             // [segment_idx]

class CollisionConstraints {
public:
    CollisionConstraints(const Param& _param, const Mission& _mission) : mission(_mission), param(_param) {
        sfcs.resize(param.M);
    }
    void initializeLSC(size_t N_obs) {  // src/collision_constraints.cpp:385-394
        lscs.clear();
        lscs.resize(N_obs);
        for (size_t oi = 0; oi < N_obs; oi++) {
            lscs[oi].resize(param.M);
            for (int m = 0; m < param.M; m++) lscs[oi][m].resize(param.n + 1);
        }
    }
    // Getter (:482-504)
    LSC getLSC(int oi, int m, int i) const { return lscs[oi][m][i]; }
    Box getSFC(int m) const { return sfcs[m]; }
    size_t getObsSize() const { return lscs.size(); }
    std::set<int> getDynamicObstacles() const { return dynamic_obstacle_indices; }
    bool isDynamicObstacle(int oi) const { return dynamic_obstacle_indices.find(oi) != dynamic_obstacle_indices.end(); }
    bool slackObstaclesEmpty() const { return dynamic_obstacle_indices.empty(); }
    // Setter (:514-543)
    void setLSC(int oi, int m, const points_t& obs_control_points, const vector3d& normal_vector, const std::vector<double>& ds) {
        for (int i = 0; i < param.n + 1; i++) lscs[oi][m][i] = LSC(obs_control_points[i], normal_vector, ds[i]);
    }
    void setLSC(int oi, int m, const points_t& obs_control_points, const vector3d& normal_vector, double d) {
        for (int i = 0; i < param.n + 1; i++) lscs[oi][m][i] = LSC(obs_control_points[i], normal_vector, d);
    }
    void setLSC(int oi, int m, const point3d& obs_point, const vector3d& normal_vector, double d) {
        for (int i = 0; i < param.n + 1; i++) lscs[oi][m][i] = LSC(obs_point, normal_vector, d);
    }
    void setSFC(int m, const Box& sfc) { sfcs[m] = sfc; }

private:
    Mission mission;
    Param param;
    RSFCs lscs;
    SFCs sfcs;
    std::set<int> dynamic_obstacle_indices;
};
}  // namespace DynamicPlanning
