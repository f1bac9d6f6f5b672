// TrajOptimizer over the C ABI of the HIP solver.  Uses only members of Param / Mission / Agent /
// CollisionConstraints that exist in the reference, so the same file builds inside the reference tree
//  and against the stand-in headers of this directory.
#include <traj_optimizer.hpp>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>

namespace DynamicPlanning {

TrajOptimizer::TrajOptimizer(const Param& _param, const Mission& _mission, const Eigen::MatrixXd& _B)
    : param(_param), mission(_mission), B(_B) {
    dim = param.world_dimension;
    M = param.M;
    n = param.n;
    phi = param.phi;
    dt = param.dt;
    configure();
}

TrajOptimizer::~TrajOptimizer() {
    if (handle) lscqp_destroy(handle);
}

TrajOptimizer::TrajOptimizer(const TrajOptimizer& o)
    : param(o.param), mission(o.mission), B(o.B), M(o.M), n(o.n), phi(o.phi), dim(o.dim), dt(o.dt), handle(nullptr), raw_x(o.raw_x),
      last_iterations(o.last_iterations), last_conflict(o.last_conflict), last_devices_used(o.last_devices_used) {
    configure();  // its own handle
}

TrajOptimizer& TrajOptimizer::operator=(const TrajOptimizer& o) {
    if (this == &o) return *this;
    param = o.param, mission = o.mission, B = o.B;
    M = o.M, n = o.n, phi = o.phi, dim = o.dim, dt = o.dt;
    raw_x = o.raw_x, last_iterations = o.last_iterations, last_conflict = o.last_conflict, last_devices_used = o.last_devices_used;
    configure();  // lscqp_update on the handle this object already owns
    return *this;
}

// Everything buildQBase/buildAeqBase precomputed in the reference (src/traj_optimizer.cpp:13-15) happens inside
// lscqp_create; the (n, phi) != (5, 3) and dim > 3 cases surface as std::invalid_argument exactly as there (:200, :249).
void TrajOptimizer::configure() {
    lscqp_class_desc d;
    std::memset(&d, 0, sizeof d);
    d.M = param.M;
    d.n = param.n;
    d.phi = param.phi;
    d.phi_n = param.phi_n;
    d.dim = param.world_dimension;
    d.planner_mode = (param.planner_mode == PlannerMode::LSC)   ? LSCQP_PLANNER_LSC
                     : (param.planner_mode == PlannerMode::BVC) ? LSCQP_PLANNER_BVC
                     : (param.planner_mode == PlannerMode::RECIPROCALRSFC) ? LSCQP_PLANNER_RSFC : LSCQP_PLANNER_DLSC;
    d.use_sfc = param.world_use_octomap ? 1 : 0;
    d.dt = param.dt;
    d.control_input_weight = param.control_input_weight;
    d.terminal_weight = param.terminal_weight;
    d.communication_range = param.communication_range;
    for (int k = 0; k < 3; k++) {
        d.world_min[k] = mission.world_min(k);
        d.world_max[k] = mission.world_max(k);
    }
    int rc = handle ? lscqp_update(handle, &d) : lscqp_create(&d, &handle);
    if (rc == LSCQP_ERR_INVALID_ARGUMENT || rc == LSCQP_ERR_UNSUPPORTED) throw std::invalid_argument(lscqp_last_error());
    if (rc != LSCQP_OK) throw std::runtime_error(lscqp_last_error());
}

void TrajOptimizer::updateParam(const Param& _param) {  // src/traj_optimizer.cpp:158-160
    param = _param;
    configure();
}

// src/traj_optimizer.cpp:530-538 with point3d's float32 arithmetic
int TrajOptimizer::getTerminalSegments_old(const Agent& agent) const {
    double ideal_flight_time = (agent.current_goal_point - agent.current_state.position).norm() / agent.nominal_velocity;
    int terminal_segments = std::max(static_cast<int>((M * param.dt - ideal_flight_time + SP_EPSILON) / param.dt), 1);
    return terminal_segments;
}

void TrajOptimizer::pack(const Agent& agent, const CollisionConstraints& constraints, lscqp_header& h,
                         std::vector<lscqp_row>& rows, std::vector<lscqp_box>& boxes) const {
    std::memset(&h, 0, sizeof h);
    for (int k = 0; k < 3; k++) {
        h.p0[k] = agent.current_state.position(k);
        h.v0[k] = agent.current_state.velocity(k);
        h.a0[k] = agent.current_state.acceleration(k);
        h.goal[k] = agent.current_goal_point(k);
        h.next_waypoint[k] = agent.next_waypoint(k);
        h.vmax[k] = agent.max_vel[k < (int)agent.max_vel.size() ? k : 0];
        h.amax[k] = agent.max_acc[k < (int)agent.max_acc.size() ? k : 0];
    }
    h.radius = agent.radius;
    h.nominal_velocity = agent.nominal_velocity;
    const size_t N_obs = constraints.getObsSize();
    h.n_obs = (int32_t)N_obs;
    h.terminal_segments = getTerminalSegments_old(agent);
    // LSC{p, nrm, d}: nrm.(c - p) - d >= 0  ->  nrm.c >= d + nrm.p   (src/traj_optimizer.cpp:413-429), order [oi][m][i]
    for (size_t oi = 0; oi < N_obs; oi++)
        for (int m = 0; m < M; m++)
            for (int i = 0; i < n + 1; i++) {
                LSC lsc = constraints.getLSC((int)oi, m, i);
                lscqp_row r;
                r.nx = lsc.normal_vector.x();
                r.ny = lsc.normal_vector.y();
                r.nz = lsc.normal_vector.z();
                r.b = lsc.d + r.nx * (double)lsc.obs_control_point.x() + r.ny * (double)lsc.obs_control_point.y() +
                      (dim == 3 ? r.nz * (double)lsc.obs_control_point.z() : 0.0);
                if (dim != 3) r.nz = 0.0;
                // the reference tests ||normal|| < SP_EPSILON_FLOAT with point3d's float norm (:409); make the same
                // decision here so borderline rows agree bit for bit
                if (lsc.normal_vector.norm() < SP_EPSILON_FLOAT) r.nx = r.ny = r.nz = 0.0;
                // An obstacle in constraints' dynamic-obstacle set gets a slack variable eps in (-inf, 0] on its rows
                // (src/traj_optimizer.cpp:272-283, 423-425) that appears in NO cost term (:285-316): such a row can always be
                // satisfied by its own slack, i.e. it never binds.  Dropping the row (zero normal: the solver skips it like
                // :409-411) is the same QP.  (The set is never populated in the reference, src/collision_constraints.cpp:495-503.)
                // slack_mode COLLISIONCONSTRAINT (the RECIPROCALRSFC planner, src/param.cpp:157-161) gives EVERY obstacle such a slack.
                if (param.slack_mode == SlackMode::COLLISIONCONSTRAINT || constraints.isDynamicObstacle((int)oi)) {
                    r.nx = r.ny = r.nz = 0.0;
                    r.b = -1.0;
                }
                rows.push_back(r);
            }
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
}

TrajOptResult TrajOptimizer::unpack(const double* x, double obj) const {
    TrajOptResult result;
    result.desired_traj = Trajectory<point3d>(M, n, dt);
    const int offset_seg = n + 1, offset_dim = M * (n + 1);
    for (int m = 0; m < M; m++)
        for (int i = 0; i < n + 1; i++) {  // src/traj_optimizer.cpp:71-83: double -> float32, z := world_z_2d in 2-D
            if (dim == 3)
                result.desired_traj[m][i] = point3d((float)x[0 * offset_dim + m * offset_seg + i], (float)x[1 * offset_dim + m * offset_seg + i],
                                                    (float)x[2 * offset_dim + m * offset_seg + i]);
            else
                result.desired_traj[m][i] = point3d((float)x[0 * offset_dim + m * offset_seg + i], (float)x[1 * offset_dim + m * offset_seg + i],
                                                    (float)param.world_z_2d);
        }
    result.total_qp_cost = obj;
    return result;
}

void TrajOptimizer::solveBatch(const std::vector<BatchItem>& items, std::vector<TrajOptResult>& results, std::vector<bool>& ok) {
    const size_t nq = items.size();
    std::vector<lscqp_header> hdr(nq);
    std::vector<lscqp_row> rows;
    std::vector<lscqp_box> boxes;
    std::vector<uint64_t> off(nq + 1, 0);
    for (size_t q = 0; q < nq; q++) {
        pack(*items[q].agent, *items[q].constraints, hdr[q], rows, boxes);
        off[q + 1] = rows.size();
    }
    const int nv = dim * M * (n + 1);
    // initial_traj (src/traj_planner.cpp:399-411) as the primal start, if every item carries one of the right shape
    std::vector<double> x_init;
    bool warm = nq > 0;
    for (size_t q = 0; q < nq && warm; q++) warm = items[q].initial_traj && items[q].initial_traj->size() == M;
    if (warm) {
        x_init.resize(nq * nv);
        for (size_t q = 0; q < nq; q++)
            for (int k = 0; k < dim; k++)
                for (int m = 0; m < M; m++)
                    for (int i = 0; i < n + 1; i++) x_init[q * nv + (k * M + m) * (n + 1) + i] = (*items[q].initial_traj)[m][i](k);
    }
    raw_x.assign(nq * nv, 0.0);
    std::vector<double> obj(nq);
    std::vector<int32_t> status(nq);
    std::vector<lscqp_info> info(nq);
    if (rows.empty()) rows.resize(1);
    int rc;
    if (communicator()) {
        int32_t used = 1;
        rc = lscqp_solve_batch_sharded(handle, communicator(), (int64_t)nq, hdr.data(), rows.data(), off.data(),
                                       boxes.empty() ? nullptr : boxes.data(), warm ? x_init.data() : nullptr, raw_x.data(), obj.data(),
                                       status.data(), info.data(), &used);
        last_devices_used = used;
    } else {
        rc = lscqp_solve_batch(handle, (int64_t)nq, hdr.data(), rows.data(), off.data(), boxes.empty() ? nullptr : boxes.data(),
                               warm ? x_init.data() : nullptr, raw_x.data(), obj.data(), status.data(), info.data());
        last_devices_used = 1;
    }
    if (rc != LSCQP_OK) throw std::runtime_error(std::string("[TrajOptimizer] ") + lscqp_last_error());
    results.resize(nq);
    ok.assign(nq, false);
    last_conflict.clear();
    const std::string QPmodel_path = param.package_path + "/log/QPmodel_trajOpt.lp";  // src/traj_optimizer.cpp:45
    for (size_t q = 0; q < nq; q++) {
        ok[q] = (status[q] == LSCQP_STATUS_OPTIMAL);
        results[q] = unpack(&raw_x[q * nv], obj[q]);
        last_iterations = info[q].iterations;
        const lscqp_row* rq = rows.data() + off[q];
        const lscqp_box* bq = boxes.empty() ? nullptr : boxes.data() + q * M;
        // the reference exports the model with param.log_solver (:47-48) and, whatever that flag says, when the solve fails (:103, 147)
        if (param.log_solver || !ok[q]) (void)lscqp_dump_instance(handle, &hdr[q], rq, bq, QPmodel_path.c_str());
        if (!ok[q]) {
            // ... and names the rows that cannot hold together (conflict refiner, :105-135).  Here: every row of the model evaluated
            // on the iterate the interior-point method stopped at -- the rows it could not satisfy are the members of the conflict.
            lscqp_diag dg;
            const uint64_t off1[2] = {0, (uint64_t)hdr[q].n_obs * (uint64_t)(M * (n + 1))};
            if (lscqp_diagnose(handle, 1, &hdr[q], rq, off1, bq, &raw_x[q * nv], 1e-6, &dg) == LSCQP_OK) {
                char buf[320];
                const bool infeasible = status[q] == LSCQP_STATUS_INFEASIBLE;
                if (dg.family >= 0 && dg.violation > 1e-6)
                    std::snprintf(buf, sizeof buf,
                                  "[TrajOptimizer] %s at mav %d (status %d, %d iterations). Conflict: %s row, oi: %d, m: %d, i: %d, axis: %d, "
                                  "violated by %g; violated rows SFC %d, LSC %d, vel %d, acc %d, comm %d",
                                  infeasible ? "No solution" : "Solver failure", items[q].agent->id, (int)status[q], (int)info[q].iterations,
                                  lscqp_row_family_name(dg.family), (int)dg.obstacle, (int)dg.segment, (int)dg.point, (int)dg.axis, dg.violation,
                                  (int)dg.violated[LSCQP_ROW_SFC], (int)dg.violated[LSCQP_ROW_LSC], (int)dg.violated[LSCQP_ROW_VEL],
                                  (int)dg.violated[LSCQP_ROW_ACC], (int)(dg.violated[LSCQP_ROW_COMM_PAIR] + dg.violated[LSCQP_ROW_COMM_WAYPOINT]));
                else
                    std::snprintf(buf, sizeof buf, "[TrajOptimizer] Solver failure at mav %d (status %d, %d iterations); no row of the last iterate is violated",
                                  items[q].agent->id, (int)status[q], (int)info[q].iterations);
                last_conflict = buf;
                std::fprintf(stderr, "%s\n", buf);  // ROS_ERROR_STREAM in the reference
            }
        }
    }
}

TrajOptResult TrajOptimizer::solve(const Agent& agent, const CollisionConstraints& constraints, const traj_t& initial_traj,
                                   bool /*use_primal_algorithm*/) {
    // initial_traj is unused by the reference's solve (only dead code reads it, :516-528); here it is the primal start of the
    // interior-point iteration.  use_primal_algorithm selected CPLEX's primal simplex (:36-39) and has no meaning here.
    std::vector<BatchItem> one(1);
    one[0].agent = &agent;
    one[0].constraints = &constraints;
    one[0].initial_traj = &initial_traj;
    std::vector<TrajOptResult> res;
    std::vector<bool> ok;
    solveBatch(one, res, ok);
    if (!ok[0]) throw PlanningReport::QPFAILED;  // src/traj_optimizer.cpp:143,152
    return res[0];
}

}  // namespace DynamicPlanning
