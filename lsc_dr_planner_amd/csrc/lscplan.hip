// lscplan.hip — one replan of a whole batch of agents as ONE chain of device work, gfx950 only: the device analogue of
//   TrajPlanner::plan / planImpl          reference src/traj_planner.cpp:33-60, 117-139
//   MultiSyncSimulator's per-agent loop   src/multi_sync_simulator.cpp:305-352 (who is whose obstacle), :354-400 (plan, doStep)
// planImpl's steps map onto the kernels of this library, all enqueued on one stream without a host round trip:
//   obstaclePrediction + initialTrajPlanning (PrevSol)  -> lscqp_shift_traj(_partial)_device over every agent's previous plan
//   broadcastMsgs' range filter                         -> lscqp_select_neighbours_device
//   constructLSC                                        -> lscqp_generate_constraints_device (LSC / CLSC / BVC)
//   constructSFC                                        -> lscqp_construct_sfc_device (initializeSFC on the first replan)
//   goalPlanning (grid-based planner goal mode)         -> lscqp_optimize_goal_device, result held as point3d (float32)
//   trajOptimization + failsafe (:755-803)              -> lscqp_solve_batch_device_ex (+ device-side second pass), failed QPs
//                                                          keep the initial trajectory
//   prev_traj = desired_traj; AgentManager::doStep      -> the plan buffer is updated in place; lscqp_validate_step_device
// Two small kernels of this file glue them: `prepare` (headers, corridor seed points, the initial trajectory in the solver's
// layout) and `commit` (failsafe, goal point).  Because nothing in the chain synchronises, allocates or copies through the
// host, the whole replan can be captured once in a hipGraph and replayed (lscqp_plan_step_graph): one graph launch instead of
// ten kernel launches per replan.  What stays on the host is what is out of scope: the waypoints of the
// grid planner / MAPF layer, written into the plan's waypoint buffer before each step.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/lscqp.h"

extern "C" int lscqp_set_error_(int code, const char* msg);
extern "C" const lscqp_class_desc* lscqp_class_desc_of_(lscqp_handle h);
extern "C" uint64_t lscqp_handle_generation_(lscqp_handle h);
extern "C" int lscqp_map_device_(lscqp_map mp);
extern "C" int lscqp_commit_validate_raw_(int M, int dim, int use_sfc, double dt, int64_t n, double time_step, double z_2d, const int32_t* d_qp_status,
                                          const double* d_x_new, const double* d_x_init, double* d_x_plan, double* d_goal, const lscqp_header* d_hdr,
                                          const lscqp_box* d_sfc, int32_t* d_valid, double* d_state, void* stream);
extern "C" int lscqp_optimize_goal_fin_device_(lscqp_handle h, int64_t n, lscqp_header* d_hdr, const lscqp_row* d_rows, const uint64_t* d_row_offsets,
                                               const lscqp_box* d_sfc, int32_t* d_status_out, double fin_dt, void* stream);
extern "C" uint64_t lscqp_map_generation_(lscqp_map mp);
extern "C" int lscqp_generate_constraints_own_(lscqp_handle h, int32_t mode, int64_t n_agents, int32_t n_obs, int64_t first_agent,
                                               const double* d_traj, const double* d_own_traj, const int32_t* d_neighbours, const double* d_radius,
                                               const double* d_downwash, const double* d_goal_all, lscqp_row* d_rows_out, int32_t n_obs_total,
                                               int32_t slot0, void* stream);

namespace lscplan {

// A plan carries the work order of its QP launch from replan to replan when the launch exceeds what the chip works on at once
// (lscqp_launch_capacity: 256 instances of the M = 10 class, 1024 of M = 5), and that of its corridor launch from this many agents on
// (the throughput build of the corridor kernel keeps four agents per CU)
constexpr size_t kSfcOrderMin = 1024;
constexpr int kThreads = 64;
static_assert(kThreads == 64, "prepare_kernel hands data between the lanes of ONE wavefront (initial trajectory read before the prediction overwrites it)");

struct Shape {
    int M, dim, nv, n_obs;
    int64_t n_agents, n_total, first_agent;
    double z_2d, dt;
    int prediction_mode, initial_traj_mode;  // LSCQP_TRAJ_FROM_*
    double reset_threshold;                  // checkObstacleDisturbance, <= 0: off
};

// Trajectory::planConstVelTraj (src/trajectory.cpp:79-91): control point (m, i) = current_state + velocity * time in point3d (float)
// arithmetic, `time` a double that advances by segment_time / n after EVERY control point -- n + 1 steps per segment, so segment m
// starts at 1.2 m dt: the reference's own quirk, kept -- and is narrowed to float by Vector3::operator*(float).
__device__ __forceinline__ double const_vel_point(double p, double v, int idx, double dt) {
#pragma clang fp contract(off)
    double time = 0;
    for (int q = 0; q < idx; q++) time += dt / 5;
    const float step = (float)v * (float)time;
    return (double)((float)p + step);
}

// One wavefront per agent of the mission.  For every agent: its position (range filter) and its PREDICTED trajectory as the others'
// obstacle -- obstaclePrediction (src/traj_planner.cpp:228-253): the shifted previous plan that lscqp_shift_traj left in `traj`
// (PREVIOUSSOLUTION; constant velocity while planner_seq < 2, :276-279), or planConstVelTraj from the current state (POSITION /
// VELOCITY), then checkObstacleDisturbance (:312-319): a prediction that starts further than reset_threshold from where the agent is
// becomes "stays where it is".  For the local agents also: header, corridor seed points, and the INITIAL trajectory
// (initialTrajPlanning :360-423, same three sources but never reset: AgentManager's is_disturbed stays false) in the generator's layout
// (`own`) and in the solver's (`x_init`).  The initial trajectory is taken before the prediction overwrites the agent's entry of `traj`.
// x_shift != NULL: the agent's entry of `traj` is first made from the previous plans -- lscqp_shift_traj's whole-segment shift
// (lscgen.hip shift_traj_kernel: segment m := previous segment m + 1, the last one := the previous plan's last point, values rounded to
// float32, z := z_2d in the plane), one graph node less per replan.
__global__ __launch_bounds__(kThreads) void prepare_kernel(Shape s, int first_replan, const double* __restrict__ x_shift, const double* __restrict__ state,
                                                            const double* __restrict__ waypoint, const double* __restrict__ goal,
                                                            double* traj, const lscqp_agent_param* __restrict__ par,
                                                            double* __restrict__ pos, lscqp_header* __restrict__ hdr,
                                                            double* __restrict__ points, double* __restrict__ x_init, double* __restrict__ own) {
    const int64_t g = blockIdx.x;  // global agent id
    const int lane = threadIdx.x;
    if (g >= s.n_total) return;
    if (lane < 3) pos[g * 3 + lane] = state[g * 9 + lane];
    const int64_t t = g - s.first_agent;  // local id
    const bool local = t >= 0 && t < s.n_agents;
    double* tr = traj + g * s.M * 18;
    const int P18 = s.M * 18;
    if (x_shift != nullptr) {
        const double* x = x_shift + g * s.dim * s.M * 6;
        for (int e = lane; e < s.M * 6; e += kThreads) {
            const int m = e / 6, i = e - 6 * m;
            const int ms = (m + 1 < s.M) ? m + 1 : s.M - 1, is = (m + 1 < s.M) ? i : 5;
            double* o = tr + e * 3;
            o[0] = (double)(float)x[(0 * s.M + ms) * 6 + is];
            o[1] = (double)(float)x[(1 * s.M + ms) * 6 + is];
            o[2] = (s.dim == 3) ? (double)(float)x[(2 * s.M + ms) * 6 + is] : (double)(float)s.z_2d;
        }
        __syncthreads();  // (one wavefront per agent: its lanes read each other's entries below)
    }
    if (local) {
        const bool prev = s.initial_traj_mode == LSCQP_TRAJ_FROM_PREVIOUS_SOLUTION && !first_replan;
        const bool still = s.initial_traj_mode == LSCQP_TRAJ_FROM_POSITION;
        auto init_at = [&](int r, int k) -> double {  // control point r = 6 m + i, axis k
            return prev ? tr[r * 3 + k] : const_vel_point(state[g * 9 + k], still ? 0.0 : state[g * 9 + 3 + k], r, s.dt);
        };
        if (lane == 0) {
            lscqp_header H;
            memset(&H, 0, sizeof H);
            const lscqp_agent_param A = par[g];
            for (int k = 0; k < 3; k++) {
                H.p0[k] = state[g * 9 + k];
                H.v0[k] = state[g * 9 + 3 + k];
                H.a0[k] = state[g * 9 + 6 + k];
                H.goal[k] = goal[g * 3 + k];
                H.next_waypoint[k] = waypoint[t * 3 + k];
                H.vmax[k] = A.max_vel[k];
                H.amax[k] = A.max_acc[k];
            }
            H.radius = A.radius;
            H.nominal_velocity = A.nominal_velocity;
            H.n_obs = s.n_obs;
            H.terminal_segments = 0;  // set by finalize_goal_kernel once the goal LP has moved the goal
            hdr[t] = H;
        }
        if (lane < 9) {
            // generateSFC (src/traj_planner.cpp:738-753): the agent's position on the first replan, afterwards the hull
            // {initial_traj.lastPoint(), current_goal_point} and the next waypoint
            const int which = lane / 3, k = lane - 3 * which;
            const double p0k = state[g * 9 + k];
            const double v = which == 0 ? init_at((s.M - 1) * 6 + 5, k) : (which == 1 ? goal[g * 3 + k] : waypoint[t * 3 + k]);
            points[t * 9 + lane] = first_replan ? p0k : v;
        }
        double* xi = x_init + t * s.nv;
        for (int e = lane; e < s.nv; e += kThreads) {  // x_init[k][m][i] = initial_traj[m][i][k]
            const int k = e / (6 * s.M), r = e - k * 6 * s.M;
            xi[e] = init_at(r, k);
        }
        double* ow = own + t * P18;
        for (int e = lane; e < P18; e += kThreads) ow[e] = init_at(e / 3, e % 3);
    }
    // the agent as an obstacle of the others.  Every lane has finished reading tr[] (init_at above: x_init, own, the corridor seed
    // points) before any lane overwrites it with the prediction below.
    __syncthreads();
    const bool keep = s.prediction_mode == LSCQP_TRAJ_FROM_PREVIOUS_SOLUTION && !first_replan;
    bool reset = false;
    if (s.reset_threshold > 0) {
#pragma clang fp contract(off)
        // (startPoint() - position).norm(): float differences, float sum of squares, sqrt in double (octomath::Vector3)
        const float dx = (float)(keep ? tr[0] : state[g * 9 + 0]) - (float)state[g * 9 + 0];
        const float dy = (float)(keep ? tr[1] : state[g * 9 + 1]) - (float)state[g * 9 + 1];
        const float dz = (float)(keep ? tr[2] : state[g * 9 + 2]) - (float)state[g * 9 + 2];
        reset = sqrt((double)(dx * dx + dy * dy + dz * dz)) > s.reset_threshold;
    }
    if (!keep || reset) {
        const bool still = reset || s.prediction_mode == LSCQP_TRAJ_FROM_POSITION;
        for (int e = lane; e < P18; e += kThreads) {
            const int k = e % 3;
            tr[e] = const_vel_point(state[g * 9 + k], still ? 0.0 : state[g * 9 + 3 + k], e / 3, s.dt);
        }
    }
}

// after the goal LP: agent.current_goal_point is a point3d (GoalOptimizer::solve returns one, src/goal_optimizer.cpp:7-55), and
// getTerminalSegments_old (src/traj_optimizer.cpp:530-538) reads it with octomap's float32 vector arithmetic
__global__ __launch_bounds__(kThreads) void finalize_goal_kernel(Shape s, lscqp_header* __restrict__ hdr) {
#pragma clang fp contract(off)
    const int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (t >= s.n_agents) return;
    lscqp_header* H = hdr + t;
    float d[3];
    for (int k = 0; k < 3; k++) {
        const float gk = (float)H->goal[k];
        H->goal[k] = (double)gk;
        d[k] = gk - (float)H->p0[k];
    }
    const float nsq = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    const double ideal_flight_time = sqrt((double)nsq) / H->nominal_velocity;
    int ts = (int)((s.M * s.dt - ideal_flight_time + 1e-9) / s.dt);
    H->terminal_segments = ts > 1 ? ts : 1;
}

}  // namespace lscplan

struct lscqp_plan_s {
    lscqp_handle hq = nullptr;  // the handle the QP solves run on: the caller's, or a private clone of its class in LSCQP_WARM_TIGHT mode
    bool own_hq = false;
    lscqp_handle h = nullptr;
    lscqp_map map = nullptr;
    lscqp_plan_desc d;
    lscplan::Shape s;
    int device = 0;
    bool first = true;
    int64_t steps = 0;
    void* buf[LSCQP_PLAN_BUF_COUNT] = {};
    size_t bytes[LSCQP_PLAN_BUF_COUNT] = {};
    // private buffers
    lscqp_agent_param* par = nullptr;
    double *radius = nullptr, *downwash = nullptr, *traj = nullptr, *pos = nullptr, *points = nullptr, *x_init = nullptr, *x_new = nullptr, *own = nullptr;
    int32_t* nbr = nullptr;
    uint64_t* off = nullptr;
    int32_t* order = nullptr;  // work order of the next solve (only where the launch exceeds lscqp_launch_capacity)
    int32_t* sfc_order = nullptr;  // ... and of the next corridor launch, from the costs recorded by this one
    uint32_t* sfc_cost = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipStream_t cap = nullptr;
    // what the captured graph (and the tight-warm-start clone) was derived from: a later lscqp_update / lscqp_map_prepare bumps these
    uint64_t h_gen = 0, map_gen = 0;
    std::vector<void*> owned;
};

namespace {

int hip_fail(hipError_t e, const char* what) {
    return lscqp_set_error_(LSCQP_ERR_HIP, (std::string(what) + ": " + hipGetErrorString(e)).c_str());
}

template <class T>
int dalloc(lscqp_plan_s* p, T** out, size_t count) {
    void* ptr = nullptr;
    const size_t b = (count ? count : 1) * sizeof(T);
    hipError_t e = hipMalloc(&ptr, b);
    if (e != hipSuccess) return hip_fail(e, "hipMalloc(plan buffer)");
    e = hipMemset(ptr, 0, b);
    if (e != hipSuccess) return hip_fail(e, "hipMemset(plan buffer)");
    p->owned.push_back(ptr);
    *out = (T*)ptr;
    return LSCQP_OK;
}

template <class T>
int dalloc_pub(lscqp_plan_s* p, int which, size_t count) {
    T* ptr = nullptr;
    const int rc = dalloc(p, &ptr, count);
    if (rc != LSCQP_OK) return rc;
    p->buf[which] = ptr;
    p->bytes[which] = count * sizeof(T);
    return LSCQP_OK;
}

#define PLAN_TRY(call)                 \
    do {                               \
        const int rc_ = (call);        \
        if (rc_ != LSCQP_OK) return rc_; \
    } while (0)

// the whole replan on `stream`; nothing here synchronises, allocates or touches host memory (capturable)
int enqueue(lscqp_plan_s* p, bool first_replan, hipStream_t stream) {
    const lscplan::Shape& s = p->s;
    if (s.n_agents == 0) return LSCQP_OK;  // an empty block of a sharded mission (9 agents over 4 devices: 3, 3, 3, 0): nothing to replan
    lscqp_handle h = p->h;
    double* state = (double*)p->buf[LSCQP_PLAN_BUF_STATE];
    double* waypoint = (double*)p->buf[LSCQP_PLAN_BUF_WAYPOINT];
    double* x_plan = (double*)p->buf[LSCQP_PLAN_BUF_PLAN];
    double* goal = (double*)p->buf[LSCQP_PLAN_BUF_GOAL];
    lscqp_header* hdr = (lscqp_header*)p->buf[LSCQP_PLAN_BUF_HEADER];
    lscqp_row* rows = (lscqp_row*)p->buf[LSCQP_PLAN_BUF_ROWS];
    lscqp_box* sfc = (lscqp_box*)p->buf[LSCQP_PLAN_BUF_SFC];
    int32_t* status = (int32_t*)p->buf[LSCQP_PLAN_BUF_STATUS];
    int32_t* goal_status = (int32_t*)p->buf[LSCQP_PLAN_BUF_GOAL_STATUS];
    int32_t* sfc_status = (int32_t*)p->buf[LSCQP_PLAN_BUF_SFC_STATUS];
    int32_t* valid = (int32_t*)p->buf[LSCQP_PLAN_BUF_VALID];
    int32_t* count = (int32_t*)p->buf[LSCQP_PLAN_BUF_IN_RANGE];
    double* state_out = (double*)p->buf[LSCQP_PLAN_BUF_NEXT_STATE];
    double* obj = (double*)p->buf[LSCQP_PLAN_BUF_OBJECTIVE];
    lscqp_info* info = (lscqp_info*)p->buf[LSCQP_PLAN_BUF_INFO];
    const double fraction = p->d.time_step / s.dt;
    // obstaclePredictionWithPrevSol / initialTrajPlanningPrevSol for every agent of the mission (:273-310, 399-423)
    const bool from_plans = !first_replan && (s.prediction_mode == LSCQP_TRAJ_FROM_PREVIOUS_SOLUTION || s.initial_traj_mode == LSCQP_TRAJ_FROM_PREVIOUS_SOLUTION);
    const bool whole_shift = from_plans && fraction >= 1.0 - 1e-9;  // (done by prepare_kernel itself)
    if (from_plans && !whole_shift) PLAN_TRY(lscqp_shift_traj_partial_device(h, s.n_total, fraction, s.z_2d, x_plan, p->traj, stream));
    hipLaunchKernelGGL(lscplan::prepare_kernel, dim3((unsigned)s.n_total), dim3(lscplan::kThreads), 0, stream, s, first_replan ? 1 : 0,
                       whole_shift ? (const double*)x_plan : (const double*)nullptr, state, waypoint, goal, p->traj, p->par, p->pos, hdr, p->points, p->x_init, p->own);
    if (p->map) {
        // (large plans: most expensive corridor of the previous replan first, the costs of this one recorded for the next)
        const int32_t* sfc_order = nullptr;
        if (p->sfc_order && !first_replan) {
            PLAN_TRY(lscqp_order_by_cost_device(s.n_agents, p->sfc_cost, p->sfc_order, stream));
            sfc_order = p->sfc_order;
        }
        PLAN_TRY(lscqp_construct_sfc_device_ordered(h, p->map, first_replan ? LSCQP_SFC_INIT : p->d.sfc_mode, s.n_agents, p->points,
                                                    p->radius + s.first_agent, sfc, sfc_status, sfc_order, p->sfc_cost, stream));
    }
    PLAN_TRY(lscqp_select_neighbours_device(h, s.n_agents, s.first_agent, s.n_total, s.n_obs, lscqp_class_desc_of_(h)->communication_range, p->pos, p->nbr, count,
                                            stream));
    if (s.n_obs > 0)
        PLAN_TRY(lscqp_generate_constraints_own_(h, p->d.constraint_mode, s.n_agents, s.n_obs, s.first_agent, p->traj, p->own, p->nbr, p->radius,
                                                 p->downwash, goal, rows, s.n_obs, 0, stream));
    if (p->d.optimize_goal) {  // (the goal LP's kernel finishes the headers itself: one node less)
        PLAN_TRY(lscqp_optimize_goal_fin_device_(h, s.n_agents, hdr, rows, p->off, p->map ? sfc : nullptr, goal_status, s.dt, stream));
    } else {
        const unsigned nb = (unsigned)((s.n_agents + lscplan::kThreads - 1) / lscplan::kThreads);
        hipLaunchKernelGGL(lscplan::finalize_goal_kernel, dim3(nb), dim3(lscplan::kThreads), 0, stream, s, hdr);
    }
    // Work order of the solve (include/lscqp.h): from the second replan on, the agents whose previous QP took the most iterations go
    // first -- the info records of the previous replan are still in place here.  Only where a launch can have a tail: more agents than
    // a few rounds of workgroups (below that every QP starts at once).
    const int32_t* order = nullptr;
    if (p->order && !first_replan) {
        PLAN_TRY(lscqp_order_by_work_device(s.n_agents, info, p->order, stream));
        order = p->order;
    }
    // retry = 3: the reference tries a failed QP again on the spot (src/traj_planner.cpp:763-766) -- here a second pass from the default start and
    // the RESCUE pass behind it are always enqueued (each returns per instance on OPTIMAL: two near-empty launches when nothing failed)
    PLAN_TRY(lscqp_solve_batch_device_ordered(p->hq, s.n_agents, s.n_obs, hdr, rows, p->off, p->map ? sfc : nullptr, p->x_init, p->x_new, obj,
                                              status, info, 3, order, stream));
    {   // commit (failsafe of trajOptimization, prev_traj = desired_traj, the goal point carried over) + isSolValid + doStep: one launch
        const lscqp_class_desc* cd = lscqp_class_desc_of_(h);
        if (cd->use_sfc && !p->map) return lscqp_set_error_(LSCQP_ERR_INVALID_ARGUMENT, "the class has corridor rows but the plan has no map");
        PLAN_TRY(lscqp_commit_validate_raw_(s.M, s.dim, cd->use_sfc, s.dt, s.n_agents, p->d.time_step, s.z_2d, status, p->x_new, p->x_init,
                                            x_plan + s.first_agent * s.nv, goal + s.first_agent * 3, hdr, p->map ? sfc : nullptr, valid, state_out, stream));
    }
    if (p->d.safety_samples > 0)  // MultiSyncSimulator::update's safety ratio / excess ratios over the step just planned (:486-577)
        PLAN_TRY(lscqp_safety_metrics_device(h, s.n_agents, s.first_agent, s.n_total, p->d.safety_samples, p->d.record_time_step, s.z_2d, x_plan,
                                             p->radius, p->downwash, hdr, (lscqp_safety*)p->buf[LSCQP_PLAN_BUF_SAFETY], stream));
    // (closed loop: LSCQP_PLAN_BUF_NEXT_STATE is the local agents' slice of the state buffer itself, so doStep's result is the next
    // replan's current state without a copy)
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "HIP launch failed (plan step)");
    return LSCQP_OK;
}

void drop_graph(lscqp_plan_s* p) {
    if (p->exec) (void)hipGraphExecDestroy(p->exec);
    if (p->graph) (void)hipGraphDestroy(p->graph);
    p->exec = nullptr;
    p->graph = nullptr;
}

// The class constants travel to the kernels BY VALUE and the map view (table pointer + margin) likewise, so a captured graph -- and
// the private WARM_TIGHT clone of the handle -- are snapshots.  lscqp_update(h) (TrajOptimizer::updateParam, src/traj_optimizer.cpp:158-160)
// or lscqp_map_prepare after the capture would be silently ignored by the replayed graph while the eager chain sees the new values:
// compare the generation counters before every step, drop the graph and re-derive the clone when they moved.
int refresh(lscqp_plan_s* p) {
    const uint64_t hg = lscqp_handle_generation_(p->h), mg = p->map ? lscqp_map_generation_(p->map) : 0;
    if (hg == p->h_gen && mg == p->map_gen) return LSCQP_OK;
    drop_graph(p);
    if (hg != p->h_gen) {
        // the buffers and the launch shapes of prepare / commit were sized at lscqp_plan_create: an update that changes the SHAPE of the
        // class (segments, dimension, corridor rows on / off, row format, a kernel capacity below the plan's neighbour slots) cannot be
        // followed -- refuse the step instead of reading and writing x_new / x_init / rows out of bounds.  Such a plan has to be
        // destroyed and created again; every other field of the class (weights, dt-independent limits, modes) is followed.
        const lscqp_class_desc* now = lscqp_class_desc_of_(p->h);
        const int M = lscqp_num_segments(p->h), nv = lscqp_num_variables(p->h);
        if (M != p->s.M || nv != p->s.nv || (lscqp_uses_sfc(p->h) != 0) != (p->map != nullptr) || lscqp_max_obstacles(p->h) < p->s.n_obs ||
            lscqp_row_bytes(p->h) != (int)sizeof(lscqp_row) || !(p->d.time_step <= now->dt * (1 + 1e-9)))
            return lscqp_set_error_(LSCQP_ERR_INVALID_ARGUMENT,
                                    "lscqp_update changed the shape of the class (segments / dimension / corridor rows / row format / neighbour "
                                    "capacity / dt below the plan's time_step) under a live plan: destroy the plan and create it again");
        p->s.dt = now->dt;
        lscqp_class_desc cd = *now;
        if (p->d.tight_warm_start && cd.warm_start != LSCQP_WARM_TIGHT) {
            cd.warm_start = LSCQP_WARM_TIGHT;
            if (p->own_hq) {
                const int rc = lscqp_update(p->hq, &cd);
                if (rc != LSCQP_OK) return rc;
            } else {  // h was WARM_TIGHT itself when the plan was made and no longer is: the clone becomes necessary now
                lscqp_handle hq = nullptr;
                const int rc = lscqp_create(&cd, &hq);
                if (rc != LSCQP_OK) return rc;
                p->hq = hq;
                p->own_hq = true;
            }
        } else if (p->own_hq) {  // h became WARM_TIGHT itself: the clone only has to follow it
            const int rc = lscqp_update(p->hq, &cd);
            if (rc != LSCQP_OK) return rc;
        }
    }
    p->h_gen = hg;
    p->map_gen = mg;
    return LSCQP_OK;
}

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev);
        else prev = -1;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

}  // namespace

extern "C" {

int lscqp_plan_create(lscqp_handle h, lscqp_map map, const lscqp_plan_desc* desc, const lscqp_agent_param* agents, lscqp_plan* out) {
    if (!h || !desc || !agents || !out) return lscqp_set_error_(LSCQP_ERR_INVALID_ARGUMENT, "null argument");
    // (n_agents == 0 is a legal plan: the empty last block of a mission sharded over more devices than ceil(N / G) blocks fill --
    // it holds the all-agent buffers, takes part in lscqp_plan_group_step's exchange and replans nothing)
    if (desc->n_agents < 0 || desc->n_total <= 0 || desc->first_agent < 0 || desc->n_total < desc->first_agent + desc->n_agents)
        return lscqp_set_error_(LSCQP_ERR_INVALID_ARGUMENT, "inconsistent sizes (n_agents >= 0, n_total > 0, n_total >= first_agent + n_agents required)");
    if (desc->n_obs < 0) return lscqp_set_error_(LSCQP_ERR_INVALID_ARGUMENT, "negative n_obs");
    if (desc->n_obs > lscqp_max_obstacles(h))
        return lscqp_set_error_(LSCQP_ERR_UNSUPPORTED, "n_obs exceeds the largest compiled kernel instance of the shape (lscqp_max_obstacles)");
    if (desc->constraint_mode != LSCQP_GEN_LSC && desc->constraint_mode != LSCQP_GEN_CLSC && desc->constraint_mode != LSCQP_GEN_BVC)
        return lscqp_set_error_(LSCQP_ERR_INVALID_ARGUMENT, "constraint_mode must be LSCQP_GEN_LSC, LSCQP_GEN_CLSC or LSCQP_GEN_BVC");
    if (desc->sfc_mode != LSCQP_SFC_FROM_HULL && desc->sfc_mode != LSCQP_SFC_FROM_POINT)
        return lscqp_set_error_(LSCQP_ERR_INVALID_ARGUMENT, "sfc_mode must be LSCQP_SFC_FROM_HULL or LSCQP_SFC_FROM_POINT");
    for (const int32_t mode : {desc->prediction_mode, desc->initial_traj_mode})
        if (mode != LSCQP_TRAJ_FROM_PREVIOUS_SOLUTION && mode != LSCQP_TRAJ_FROM_POSITION && mode != LSCQP_TRAJ_FROM_VELOCITY)
            return lscqp_set_error_(LSCQP_ERR_INVALID_ARGUMENT, "prediction_mode / initial_traj_mode must be LSCQP_TRAJ_FROM_PREVIOUS_SOLUTION, _FROM_POSITION or _FROM_VELOCITY");
    if (!(desc->reset_threshold == desc->reset_threshold)) return lscqp_set_error_(LSCQP_ERR_INVALID_ARGUMENT, "reset_threshold is NaN");
    const int uses_sfc = lscqp_uses_sfc(h);
    if ((uses_sfc != 0) != (map != nullptr))
        return lscqp_set_error_(LSCQP_ERR_INVALID_ARGUMENT, "a map is required exactly when the solver class uses corridors (use_sfc)");
    const int M = lscqp_num_segments(h), nv = lscqp_num_variables(h);
    const double dt_probe = lscqp_class_desc_of_(h)->dt;
    if (!(desc->time_step > 0) || desc->time_step > dt_probe * (1 + 1e-9))
        return lscqp_set_error_(LSCQP_ERR_INVALID_ARGUMENT, "0 < time_step <= dt required (multisim_time_step, src/traj_planner.cpp:401-421)");
    if (desc->safety_samples < 0 || (desc->safety_samples > 0 && (desc->n_agents != desc->n_total || !(desc->record_time_step > 0))))
        return lscqp_set_error_(LSCQP_ERR_INVALID_ARGUMENT,
                                "safety_samples needs every agent's new plan on this device (n_agents == n_total) and record_time_step > 0");
    if (lscqp_row_bytes(h) != (int)sizeof(lscqp_row))
        return lscqp_set_error_(LSCQP_ERR_UNSUPPORTED, "the plan chain uses 32-byte rows (row_format = LSCQP_ROWS_F64)");
    if (map) {  // the corridor kernel's free-space table for the agents of this mission (no-op if the map already has one that serves them)
        double rmax = 0;
        for (int64_t i = 0; i < desc->n_total; i++) rmax = agents[i].radius > rmax ? agents[i].radius : rmax;
        if (rmax > 0) {
            const int rc = lscqp_map_prepare(map, rmax);
            if (rc != LSCQP_OK && rc != LSCQP_ERR_UNSUPPORTED) return rc;  // (a map too large for the table: the corridors work without it)
        }
    }
    int cur_dev = 0;
    if (hipGetDevice(&cur_dev) != hipSuccess) return lscqp_set_error_(LSCQP_ERR_NO_DEVICE, "no HIP device: lscqp has no CPU fallback");
    if (map && lscqp_map_device_(map) != cur_dev)
        return lscqp_set_error_(LSCQP_ERR_INVALID_ARGUMENT,
                                "the map lives on another device than the plan (create the map with the plan's device current: the corridor "
                                "kernel dereferences the map's grids)");
    lscqp_plan_s* p = new lscqp_plan_s();
    p->h = h;
    p->hq = h;
    if (desc->tight_warm_start && lscqp_class_desc_of_(h)->warm_start != LSCQP_WARM_TIGHT) {
        lscqp_class_desc cd = *lscqp_class_desc_of_(h);
        cd.warm_start = LSCQP_WARM_TIGHT;
        const int rc_clone = lscqp_create(&cd, &p->hq);
        if (rc_clone != LSCQP_OK) {
            delete p;
            return rc_clone;  // (lscqp_create has set the message)
        }
        p->own_hq = true;
    }
    p->h_gen = lscqp_handle_generation_(h);
    p->map_gen = map ? lscqp_map_generation_(map) : 0;
    p->map = map;
    p->d = *desc;
    p->s.M = M;
    p->s.dim = nv / (6 * M);
    p->s.nv = nv;
    p->s.n_obs = desc->n_obs;
    p->s.n_agents = desc->n_agents;
    p->s.n_total = desc->n_total;
    p->s.first_agent = desc->first_agent;
    p->s.z_2d = desc->z_2d;
    p->s.dt = dt_probe;
    p->s.prediction_mode = desc->prediction_mode;
    p->s.initial_traj_mode = desc->initial_traj_mode;
    p->s.reset_threshold = desc->reset_threshold;
    p->device = cur_dev;
    const size_t n = (size_t)desc->n_agents, nt = (size_t)desc->n_total, P = (size_t)M * 6, no = (size_t)desc->n_obs;
    int rc = LSCQP_OK;
    auto ok = [&](int r) { return rc == LSCQP_OK && (rc = r) == LSCQP_OK; };
    // (the three buffers a group exchanges carry room for LSCQP_PLAN_EXCHANGE_PAD agents behind the mission: a ragged split is then one
    // in-place all-gather of full blocks as well, include/lscqp.h; the public size stays the mission's)
    const size_t ntp = nt + LSCQP_PLAN_EXCHANGE_PAD;
    ok(dalloc_pub<double>(p, LSCQP_PLAN_BUF_STATE, ntp * 9)) && ok(dalloc_pub<double>(p, LSCQP_PLAN_BUF_WAYPOINT, n * 3)) &&
        ok(dalloc_pub<double>(p, LSCQP_PLAN_BUF_PLAN, ntp * nv)) && ok(dalloc_pub<double>(p, LSCQP_PLAN_BUF_GOAL, ntp * 3)) &&
        ok(dalloc_pub<lscqp_header>(p, LSCQP_PLAN_BUF_HEADER, n)) && ok(dalloc_pub<lscqp_row>(p, LSCQP_PLAN_BUF_ROWS, n * no * P)) &&
        ok(dalloc_pub<lscqp_box>(p, LSCQP_PLAN_BUF_SFC, n * M)) && ok(dalloc_pub<int32_t>(p, LSCQP_PLAN_BUF_STATUS, n)) &&
        ok(dalloc_pub<int32_t>(p, LSCQP_PLAN_BUF_GOAL_STATUS, n)) && ok(dalloc_pub<int32_t>(p, LSCQP_PLAN_BUF_SFC_STATUS, n)) &&
        ok(dalloc_pub<int32_t>(p, LSCQP_PLAN_BUF_VALID, n)) && ok(dalloc_pub<int32_t>(p, LSCQP_PLAN_BUF_IN_RANGE, n)) &&
        (desc->closed_loop ? true : ok(dalloc_pub<double>(p, LSCQP_PLAN_BUF_NEXT_STATE, n * 9))) && ok(dalloc_pub<double>(p, LSCQP_PLAN_BUF_OBJECTIVE, n)) &&
        ok(dalloc_pub<lscqp_info>(p, LSCQP_PLAN_BUF_INFO, n)) && ok(dalloc_pub<lscqp_safety>(p, LSCQP_PLAN_BUF_SAFETY, n)) && ok(dalloc(p, &p->par, nt)) && ok(dalloc(p, &p->radius, nt)) &&
        ok(dalloc(p, &p->downwash, nt)) && ok(dalloc(p, &p->traj, nt * P * 3)) && ok(dalloc(p, &p->pos, nt * 3)) &&
        ok(dalloc(p, &p->points, n * 9)) && ok(dalloc(p, &p->own, n * P * 3)) && ok(dalloc(p, &p->x_init, n * nv)) && ok(dalloc(p, &p->x_new, n * nv)) &&
        ok(dalloc(p, &p->nbr, n * no)) && ok(dalloc(p, &p->off, n + 1)) && ((int64_t)n > lscqp_launch_capacity(h, (int64_t)n, desc->n_obs) ? ok(dalloc(p, &p->order, n)) : true) &&
        ((map && n > lscplan::kSfcOrderMin) ? (ok(dalloc(p, &p->sfc_order, n)) && ok(dalloc(p, &p->sfc_cost, n))) : true);
    if (rc == LSCQP_OK) {
        p->bytes[LSCQP_PLAN_BUF_STATE] = nt * 9 * sizeof(double);
        p->bytes[LSCQP_PLAN_BUF_PLAN] = nt * nv * sizeof(double);
        p->bytes[LSCQP_PLAN_BUF_GOAL] = nt * 3 * sizeof(double);
    }
    if (rc == LSCQP_OK && desc->closed_loop) {
        p->buf[LSCQP_PLAN_BUF_NEXT_STATE] = (double*)p->buf[LSCQP_PLAN_BUF_STATE] + desc->first_agent * 9;
        p->bytes[LSCQP_PLAN_BUF_NEXT_STATE] = n * 9 * sizeof(double);
    }
    if (rc == LSCQP_OK) {
        std::vector<double> r(nt), w(nt);
        std::vector<uint64_t> off(n + 1);
        for (size_t i = 0; i < nt; i++) {
            r[i] = agents[i].radius;
            w[i] = agents[i].downwash;
        }
        for (size_t i = 0; i <= n; i++) off[i] = (uint64_t)(i * no * P);
        hipError_t e = hipMemcpy(p->par, agents, nt * sizeof(lscqp_agent_param), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(p->radius, r.data(), nt * sizeof(double), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(p->downwash, w.data(), nt * sizeof(double), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(p->off, off.data(), (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&p->cap, hipStreamNonBlocking);
        if (e != hipSuccess) rc = hip_fail(e, "plan set-up");
    }
    if (rc != LSCQP_OK) {
        lscqp_plan_destroy(p);
        return rc;
    }
    *out = p;
    return LSCQP_OK;
}

void lscqp_plan_destroy(lscqp_plan p) {
    if (!p) return;
    DeviceGuard g(p->device);
    (void)hipDeviceSynchronize();
    drop_graph(p);
    if (p->cap) (void)hipStreamDestroy(p->cap);
    for (void* q : p->owned) (void)hipFree(q);
    if (p->own_hq) lscqp_destroy(p->hq);
    delete p;
}

int lscqp_plan_reset(lscqp_plan p, const double* start_positions, const double* goal_points) {
    if (!p || !start_positions) return lscqp_set_error_(LSCQP_ERR_INVALID_ARGUMENT, "null argument");
    DeviceGuard g(p->device);
    const lscplan::Shape& s = p->s;
    const size_t nt = (size_t)s.n_total, P = (size_t)s.M * 6;
    std::vector<double> st(nt * 9, 0.0), x(nt * s.nv), gl(nt * 3);
    for (size_t a = 0; a < nt; a++) {
        for (int k = 0; k < 3; k++) {
            // State holds point3d; a 2-D mission flies at z = world_z_2d (src/agent_manager.cpp:40-42)
            const double v = (k < s.dim) ? (double)(float)start_positions[a * 3 + k] : (double)(float)s.z_2d;
            st[a * 9 + k] = v;
            gl[a * 3 + k] = goal_points ? (double)(float)goal_points[a * 3 + k] : v;  // AgentManager ctor: current_goal_point = start
        }
        for (int k = 0; k < s.dim; k++)
            for (size_t j = 0; j < P; j++) x[a * s.nv + k * P + j] = st[a * 9 + k];
    }
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(p->buf[LSCQP_PLAN_BUF_STATE], st.data(), st.size() * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(p->buf[LSCQP_PLAN_BUF_PLAN], x.data(), x.size() * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(p->buf[LSCQP_PLAN_BUF_GOAL], gl.data(), gl.size() * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(p->buf[LSCQP_PLAN_BUF_WAYPOINT], gl.data() + s.first_agent * 3, (size_t)s.n_agents * 3 * sizeof(double), hipMemcpyHostToDevice);
    if (e != hipSuccess) return hip_fail(e, "plan reset");
    p->first = true;
    p->steps = 0;
    return LSCQP_OK;
}

void* lscqp_plan_buffer(lscqp_plan p, int32_t which, uint64_t* bytes_out) {
    if (!p || which < 0 || which >= LSCQP_PLAN_BUF_COUNT) {
        lscqp_set_error_(LSCQP_ERR_INVALID_ARGUMENT, "unknown plan buffer");
        return nullptr;
    }
    if (bytes_out) *bytes_out = p->bytes[which];
    return p->buf[which];
}

int lscqp_plan_upload(lscqp_plan p, int32_t which, const void* host, uint64_t offset, uint64_t bytes) {
    if (!p || !host || which < 0 || which >= LSCQP_PLAN_BUF_COUNT) return lscqp_set_error_(LSCQP_ERR_INVALID_ARGUMENT, "bad argument");
    if (offset + bytes > p->bytes[which]) return lscqp_set_error_(LSCQP_ERR_INVALID_ARGUMENT, "range exceeds the buffer");
    DeviceGuard g(p->device);
    const hipError_t e = hipMemcpy((char*)p->buf[which] + offset, host, bytes, hipMemcpyHostToDevice);
    return e == hipSuccess ? LSCQP_OK : hip_fail(e, "plan upload");
}

int lscqp_plan_download(lscqp_plan p, int32_t which, void* host, uint64_t offset, uint64_t bytes) {
    if (!p || !host || which < 0 || which >= LSCQP_PLAN_BUF_COUNT) return lscqp_set_error_(LSCQP_ERR_INVALID_ARGUMENT, "bad argument");
    if (offset + bytes > p->bytes[which]) return lscqp_set_error_(LSCQP_ERR_INVALID_ARGUMENT, "range exceeds the buffer");
    DeviceGuard g(p->device);
    const hipError_t e = hipMemcpy(host, (const char*)p->buf[which] + offset, bytes, hipMemcpyDeviceToHost);  // waits for the device
    return e == hipSuccess ? LSCQP_OK : hip_fail(e, "plan download");
}

int lscqp_plan_step(lscqp_plan p, void* stream) {
    if (!p) return lscqp_set_error_(LSCQP_ERR_INVALID_ARGUMENT, "null plan");
    DeviceGuard g(p->device);
    int rc = refresh(p);
    if (rc != LSCQP_OK) return rc;
    rc = enqueue(p, p->first, (hipStream_t)stream);
    if (rc != LSCQP_OK) return rc;
    p->first = false;
    p->steps++;
    return LSCQP_OK;
}

int lscqp_plan_step_graph(lscqp_plan p, void* stream) {
    if (!p) return lscqp_set_error_(LSCQP_ERR_INVALID_ARGUMENT, "null plan");
    // the first replan differs (initializeSFC) and it also warms the kernels' one-time function attributes up: eager
    if (p->first || p->s.n_agents == 0) return lscqp_plan_step(p, stream);
    DeviceGuard g(p->device);
    {
        const int rc_ = refresh(p);
        if (rc_ != LSCQP_OK) return rc_;
    }
    if (!p->exec) {
        hipError_t e = hipStreamBeginCapture(p->cap, hipStreamCaptureModeThreadLocal);
        if (e != hipSuccess) return hip_fail(e, "hipStreamBeginCapture");
        const int rc = enqueue(p, false, p->cap);
        hipGraph_t gr = nullptr;
        e = hipStreamEndCapture(p->cap, &gr);
        if (rc != LSCQP_OK) {
            if (gr) (void)hipGraphDestroy(gr);
            return rc;
        }
        if (e != hipSuccess) return hip_fail(e, "hipStreamEndCapture");
        p->graph = gr;
        e = hipGraphInstantiate(&p->exec, gr, nullptr, nullptr, 0);
        if (e != hipSuccess) {
            drop_graph(p);
            return hip_fail(e, "hipGraphInstantiate");
        }
    }
    const hipError_t e = hipGraphLaunch(p->exec, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "hipGraphLaunch");
    p->steps++;
    return LSCQP_OK;
}

const lscqp_plan_desc* lscqp_plan_desc_of_(lscqp_plan p) { return &p->d; }  // (library-internal: lscqp_comm.hip)
int lscqp_plan_device_(lscqp_plan p) { return p->device; }

int64_t lscqp_plan_graph_nodes(lscqp_plan p) {
    if (!p || !p->graph) return 0;
    size_t n = 0;
    if (hipGraphGetNodes(p->graph, nullptr, &n) != hipSuccess) return -1;
    return (int64_t)n;
}

}  // extern "C"
