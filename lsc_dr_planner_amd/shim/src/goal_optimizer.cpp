// GoalOptimizer over the C ABI of the HIP library (lscqp_optimize_goal).  Uses only members of Param / Mission / Agent /
// CollisionConstraints that exist in the reference, like traj_optimizer.cpp.
#include <goal_optimizer.hpp>

#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace DynamicPlanning {

GoalOptimizer::GoalOptimizer(const Param& _param, const Mission& _mission) : param(_param), mission(_mission) {
    lscqp_class_desc d;
    std::memset(&d, 0, sizeof d);
    d.M = param.M;
    d.n = param.n;
    d.phi = param.phi;
    d.phi_n = param.phi_n;
    d.dim = param.world_dimension;
    d.planner_mode = (param.planner_mode == PlannerMode::LSC) ? LSCQP_PLANNER_LSC
                     : (param.planner_mode == PlannerMode::BVC) ? LSCQP_PLANNER_BVC : LSCQP_PLANNER_DLSC;
    d.use_sfc = param.world_use_octomap ? 1 : 0;
    d.dt = param.dt;
    d.control_input_weight = param.control_input_weight;
    d.terminal_weight = param.terminal_weight;
    d.communication_range = param.communication_range;
    for (int k = 0; k < 3; k++) {
        d.world_min[k] = mission.world_min(k);
        d.world_max[k] = mission.world_max(k);
    }
    const int rc = lscqp_create(&d, &handle);
    if (rc == LSCQP_ERR_INVALID_ARGUMENT || rc == LSCQP_ERR_UNSUPPORTED) throw std::invalid_argument(lscqp_last_error());
    if (rc != LSCQP_OK) throw std::runtime_error(lscqp_last_error());
}

GoalOptimizer::~GoalOptimizer() {
    if (handle) lscqp_destroy(handle);
}

point3d GoalOptimizer::solve(const Agent& agent, const CollisionConstraints& constraints, const point3d& current_goal_point,
                             const point3d& next_waypoint) {
    const int M = param.M, n = param.n, dim = param.world_dimension;
    lscqp_header h;
    std::memset(&h, 0, sizeof h);
    for (int k = 0; k < 3; k++) {
        h.p0[k] = agent.current_state.position(k);
        h.goal[k] = current_goal_point(k);
        h.next_waypoint[k] = next_waypoint(k);
    }
    const size_t N_obs = constraints.getObsSize();
    h.n_obs = (int32_t)N_obs;
    // only getLSC(oi, M-1, n) enters the model (src/goal_optimizer.cpp:140); the other slots of the packed layout stay zero
    std::vector<lscqp_row> rows(N_obs * M * (n + 1) + 1);
    std::memset(rows.data(), 0, sizeof(lscqp_row) * rows.size());
    for (size_t oi = 0; oi < N_obs; oi++) {
        LSC lsc = constraints.getLSC((int)oi, M - 1, n);
        lscqp_row& r = rows[(oi * M + (M - 1)) * (n + 1) + n];
        r.nx = lsc.normal_vector.x();
        r.ny = lsc.normal_vector.y();
        r.nz = (dim == 3) ? lsc.normal_vector.z() : 0.0;
        r.b = lsc.d + r.nx * (double)lsc.obs_control_point.x() + r.ny * (double)lsc.obs_control_point.y() +
              r.nz * (double)lsc.obs_control_point.z();
        if (lsc.normal_vector.norm() < SP_EPSILON_FLOAT) r.nx = r.ny = r.nz = 0.0;  // :142-144, point3d's float norm
    }
    std::vector<uint64_t> off = {0, (uint64_t)(N_obs * M * (n + 1))};
    std::vector<lscqp_box> boxes;
    if (param.world_use_octomap)
        for (int m = 0; m < M; m++) {
            Box sfc = constraints.getSFC(m);
            lscqp_box b;
            for (int k = 0; k < 3; k++) {
                b.bmin[k] = sfc.box_min(k);
                b.bmax[k] = sfc.box_max(k);
            }
            boxes.push_back(b);
        }
    int32_t status = -1;
    const int rc = lscqp_optimize_goal(handle, 1, &h, rows.data(), off.data(), boxes.empty() ? nullptr : boxes.data(), &status);
    if (rc != LSCQP_OK) throw std::runtime_error(std::string("[GoalOptimizer] ") + lscqp_last_error());
    if (status != LSCQP_STATUS_OPTIMAL) throw PlanningReport::QPFAILED;  // src/goal_optimizer.cpp:57-69
    return point3d((float)h.goal[0], (float)h.goal[1], (float)h.goal[2]);
}

}  // namespace DynamicPlanning
