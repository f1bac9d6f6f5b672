// The container half of the reference's CollisionConstraints (include/collision_constraints.hpp:17-46,98-176,
// src/collision_constraints.cpp:5-8,32-59,385-394,482-543): storage lscs[oi][m][i], sfcs[m], setters used by the LSC
// generators (src/traj_planner.cpp:581-736) and the getters the QP builder reads.  SFC expansion over the octomap EDT,
// RViz markers and convex-hull meshing are outside the trajectory-QP path.
#pragma once
#include <param.hpp>
#include <mission.hpp>
#include <lscqp.h>

#include <memory>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>
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

// Stands where the reference holds std::shared_ptr<DynamicEDTOctomap> (include/collision_constraints.hpp:134,168;
// built by MapManager, src/map_manager.cpp:13-14,74-77): the voxel map of include/lscqp.h, resident on the GPU.
class DistanceMap {
public:
    // the world file MapManager::updateOctreeFromCSV reads (src/map_manager.cpp:262-305)
    DistanceMap(const std::string& world_csv, const point3d& world_min, const point3d& world_max, double resolution, double maxdist = 1.0) {
        const double lo[3] = {world_min.x(), world_min.y(), world_min.z()}, hi[3] = {world_max.x(), world_max.y(), world_max.z()};
        if (lscqp_map_create_from_csv(world_csv.c_str(), lo, hi, resolution, maxdist, &map_) != LSCQP_OK)
            throw std::runtime_error(std::string("[DistanceMap] ") + lscqp_last_error());
    }
    // obstacle boxes given directly: rows of {centre x, y, z, size x, y, z}
    DistanceMap(const std::vector<double>& boxes, const point3d& world_min, const point3d& world_max, double resolution, double maxdist = 1.0) {
        const double lo[3] = {world_min.x(), world_min.y(), world_min.z()}, hi[3] = {world_max.x(), world_max.y(), world_max.z()};
        if (lscqp_map_create(boxes.data(), (int64_t)(boxes.size() / 6), lo, hi, resolution, maxdist, &map_) != LSCQP_OK)
            throw std::runtime_error(std::string("[DistanceMap] ") + lscqp_last_error());
    }
    ~DistanceMap() { lscqp_map_destroy(map_); }
    DistanceMap(const DistanceMap&) = delete;
    DistanceMap& operator=(const DistanceMap&) = delete;
    lscqp_map handle() const { return map_; }
    // optional, at set-up time (before the planners' threads start): the free-space table for agents up to max_radius -- the corridor
    // tests in open space then pass without sampling, the boxes stay the same (lscqp_map_prepare)
    void prepare(double max_radius) {
        if (lscqp_map_prepare(map_, max_radius) != LSCQP_OK) throw std::runtime_error(std::string("[DistanceMap] ") + lscqp_last_error());
    }

private:
    lscqp_map map_ = nullptr;
};

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
    // (additive: the reference declares the set but nothing ever inserts into it, src/collision_constraints.cpp:495-503)
    void markDynamicObstacle(int oi) { dynamic_obstacle_indices.insert(oi); }
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
    void setDistmap(std::shared_ptr<DistanceMap> distmap_ptr_) { distmap_ptr = std::move(distmap_ptr_); }  // :506-508

    // Corridor construction (src/collision_constraints.cpp:366-436), one agent = a batch of one on the device.
    void initializeSFC(const point3d& agent_position, double agent_radius) {  // :366-384
        if (!constructOnDevice(LSCQP_SFC_INIT, agent_position, agent_position, agent_position, agent_radius))
            throw std::invalid_argument("[CollisionConstraints] Invalid initial SFC");
    }
    void constructSFCFromPoint(const point3d& point, const point3d& goal_point, double agent_radius) {  // :396-412
        constructOnDevice(LSCQP_SFC_FROM_POINT, point, goal_point, goal_point, agent_radius);
    }
    // convex_hull = {last point of the initial trajectory, current goal point} (src/traj_planner.cpp:742-745)
    void constructSFCFromConvexHull(const points_t& convex_hull, const point3d& next_waypoint, double agent_radius) {  // :414-436
        if (convex_hull.size() != 2) throw std::invalid_argument("[CollisionConstraints] convex hull of two points expected");
        constructOnDevice(LSCQP_SFC_FROM_HULL, convex_hull[0], convex_hull[1], next_waypoint, agent_radius);
    }

private:
    // returns false where the reference keeps the previous box ("Cannot find proper SFC, use previous one") or, for the
    // initial corridor, throws
    bool constructOnDevice(int mode, const point3d& a, const point3d& b, const point3d& c, double radius) {
        if (!distmap_ptr) throw std::runtime_error("[CollisionConstraints] setDistmap() first");
        const double pts[9] = {a.x(), a.y(), a.z(), b.x(), b.y(), b.z(), c.x(), c.y(), c.z()};
        std::vector<lscqp_box> boxes((size_t)param.M);
        for (int m = 0; m < param.M; m++)
            for (int k = 0; k < 3; k++) boxes[m].bmin[k] = sfcs[m].box_min(k), boxes[m].bmax[k] = sfcs[m].box_max(k);
        int32_t status = 0;
        if (lscqp_construct_sfc(distmap_ptr->handle(), mode, param.M, 1, pts, &radius, boxes.data(), &status) != LSCQP_OK)
            throw std::runtime_error(std::string("[CollisionConstraints] ") + lscqp_last_error());
        for (int m = 0; m < param.M; m++)
            sfcs[m] = Box(point3d((float)boxes[m].bmin[0], (float)boxes[m].bmin[1], (float)boxes[m].bmin[2]),
                          point3d((float)boxes[m].bmax[0], (float)boxes[m].bmax[1], (float)boxes[m].bmax[2]));
        return status == 1;
    }

    Mission mission;
    Param param;
    std::shared_ptr<DistanceMap> distmap_ptr;
    RSFCs lscs;
    SFCs sfcs;
    std::set<int> dynamic_obstacle_indices;
};
}  // namespace DynamicPlanning
