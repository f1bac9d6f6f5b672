// lscpost.hip — post-solve epilogue on the device, gfx950 only: what the planner does with a solved
// trajectory before the next replan.
//
// Replaces, for a batch of solved agents,
//   TrajPlanner::isSolValid          reference src/traj_planner.cpp:990-1045  (SFC containment + dynamic limits at the
//                                    simulation step; its LSC check is commented out in the reference and stays out)
//   Trajectory::getStateAt           src/trajectory.cpp:156-170 (getPointAt :111-153, derivative :180-199)
//   AgentManager::doStep             src/agent_manager.cpp:29-50  (the agent's next state = ideal future state)
//   MultiSyncSimulator::update       src/multi_sync_simulator.cpp:486-577 (safety ratio between agents, velocity /
//                                    acceleration excess ratios: the numbers of the reference's summary CSV)
// Control points are first truncated to float32 exactly as TrajOptResult::desired_traj holds them
// (src/traj_optimizer.cpp:71-83); in 2-D missions z := world_z_2d.  One wavefront per agent (validation), one lane per agent pair tile (safety).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/lscqp.h"

namespace lscpost {

constexpr int kThreads = 64;
constexpr double kEpsFloat = 1e-5;  // SP_EPSILON_FLOAT, Box::isPointInBox (src/collision_constraints.cpp:81-88)

// sum_i cp[i] C(n,i) t^i (1-t)^(n-i)
template <int N>
__device__ __forceinline__ double bern(const double (&cp)[6], double t) {
    constexpr int binom[6][6] = {{1, 0, 0, 0, 0, 0}, {1, 1, 0, 0, 0, 0}, {1, 2, 1, 0, 0, 0}, {1, 3, 3, 1, 0, 0}, {1, 4, 6, 4, 1, 0}, {1, 5, 10, 10, 5, 1}};
    double s = 0, ti = 1;
    double omt[6];
    omt[0] = 1;
#pragma unroll
    for (int i = 1; i <= N; i++) omt[i] = omt[i - 1] * (1 - t);
#pragma unroll
    for (int i = 0; i <= N; i++) {
        s += cp[i] * binom[N][i] * ti * omt[N - i];
        ti *= t;
    }
    return s;
}

// one wavefront per agent: the lanes share the (segment, control point) pairs of the corridor test, lanes 0..2 evaluate one axis of the
// state each (a lane per agent walked ~60 dependent loads: 19 us for 10 agents in the replan chain)
// The replan chain (lscplan.hip) hands over its COMMIT as well (qp_status != NULL): desired_traj = the new plan where the QP is OPTIMAL, else
// the initial trajectory (trajOptimization's failsafe, src/traj_planner.cpp:796-797), stored as the agent's plan for the next replan together
// with the goal point the planner carries over -- and validated from where it came from, so that nothing is read back (one graph node less).
struct Commit {
    const int32_t* qp_status;  // NULL: validate `x` as it is
    const double* x_new;       // [n][nv]
    const double* x_init;      // [n][nv]
    double* x_plan;            // [n][nv] the local agents' block of the mission's plans
    double* goal;              // [n][3] ... of the goal points
};
__global__ __launch_bounds__(kThreads) void validate_step_kernel(int M, int dim, int use_sfc, double dt, int64_t n, double time_step, double z_2d,
                                                                 const double* __restrict__ x, const lscqp_header* __restrict__ hdr,
                                                                 const lscqp_box* __restrict__ sfc, int32_t* __restrict__ valid,
                                                                 double* __restrict__ state, Commit cm) {
    const int64_t q = blockIdx.x;
    if (q >= n) return;
    const int lane = threadIdx.x;
    const int P = 6 * M;
    const double* xq = x + q * dim * P;
    if (cm.qp_status != nullptr) {
        const int nv = dim * P;
        xq = (cm.qp_status[q] == LSCQP_STATUS_OPTIMAL ? cm.x_new : cm.x_init) + q * nv;
        for (int j = lane; j < nv; j += kThreads) cm.x_plan[q * nv + j] = xq[j];
        if (lane < 3) cm.goal[q * 3 + lane] = hdr[q].goal[lane];
    }
    auto cp = [&](int k, int m, int i) -> double {  // desired_traj[m][i](k), float32
        return (k < dim) ? (double)(float)xq[k * P + 6 * m + i] : (double)(float)z_2d;
    };
    bool ok = true;
    if (use_sfc) {  // :992-1010: segment 0 from control point phi = 3 on, whole segments afterwards
        for (int e = lane; e < P; e += kThreads) {
            const int m = e / 6, i = e - 6 * m;
            if (m == 0 && i < 3) continue;
            const lscqp_box* B = &sfc[q * M + m];
            for (int k = 0; k < 3; k++) {
                const double c = cp(k, m, i);
                ok = ok && (c > (double)(float)B->bmin[k] - kEpsFloat) && (c < (double)(float)B->bmax[k] + kEpsFloat);
            }
        }
    }
    // getPointAt's segment search (:121-136)
    int ms = -1;
    double tn = 0, end = 0;
    for (int idx = 0; idx < M; idx++) {
        end += dt;
        if (time_step < end) {
            ms = idx;
            tn = 1 - (end - time_step) / dt;
            break;
        }
    }
    if (ms < 0) {
        ms = M - 1;
        tn = 1.0;
    }
    if (lane < 3) {
        const int k = lane;
        const lscqp_header* H = hdr + q;
        double* S = state + q * 9;
        double c[6], d1[6] = {0, 0, 0, 0, 0, 0}, d2[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 6; i++) c[i] = cp(k, ms, i);
        for (int i = 0; i < 5; i++) d1[i] = (c[i + 1] - c[i]) * (5.0 / dt);   // derivative(), :183-199
        for (int i = 0; i < 4; i++) d2[i] = (d1[i + 1] - d1[i]) * (4.0 / dt);
        const double pos = bern<5>(c, tn), vel = bern<4>(d1, tn), acc = bern<3>(d2, tn);
        // State holds point3d (float32); doStep: 2-D missions pin z to world_z_2d (src/agent_manager.cpp:40-42)
        const double sp = (k < dim) ? (double)(float)pos : (double)(float)z_2d;
        const double sv = (k < dim) ? (double)(float)vel : 0.0, sa = (k < dim) ? (double)(float)acc : 0.0;
        S[k] = sp;
        S[3 + k] = sv;
        S[6 + k] = sa;
        if (k < dim) {  // :1030-1041, 1 % tolerance
            const double vm = k == 0 ? H->vmax[0] : (k == 1 ? H->vmax[1] : H->vmax[2]), am = k == 0 ? H->amax[0] : (k == 1 ? H->amax[1] : H->amax[2]);
            ok = ok && !(fabs(sv) > vm * 1.01) && !(fabs(sa) > am * 1.01);
        }
    }
    const bool all_ok = __builtin_amdgcn_ballot_w64(!ok) == 0;
    if (lane == 0) valid[q] = all_ok ? 1 : 0;
}

// ---- safety metrics of MultiSyncSimulator::update (reference src/multi_sync_simulator.cpp:486-577) --------------------
// For every sample time t_s = s * record_time_step of the step just planned and every local agent i:
//   safety ratio  min_j |E (p_i - p_j)| / (r_i + r_j),  E = diag(1, 1, 1/downwash_ij)   (ellipsoidalDistance, include/util.hpp:155-159)
//   excess ratios (v_k - vmax_k) / vmax_k and (a_k - amax_k) / amax_k where positive       (:560-572; signed, like the reference)
// An all-pairs pass: 32 local agents x 8 slices of the j range per block; the block evaluates a tile of 256 trajectories
// at t_s into LDS (about 2 flop per pair of redundant work, no scratch buffer in HBM) and every lane scans its slice of it.
constexpr int kSafI = 32, kSafS = 8, kSafT = kSafI * kSafS;

__device__ __forceinline__ double post_rcp(double d) {  // 1/d: v_rcp_f64 + two Newton steps
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double post_sqrt(double x) {  // sqrt(x), x >= 0: v_rsq_f64 + two coupled Newton steps
    if (!(x > 0.0)) return 0.0;
    const double r = __builtin_amdgcn_rsq(x);
    double g = x * r, h = 0.5 * r;
    double e = fma(-h, g, 0.5);
    g = fma(g, e, g);
    h = fma(h, e, h);
    e = fma(-h, g, 0.5);
    g = fma(g, e, g);
    h = fma(h, e, h);
    return fma(fma(-g, g, x), h, g);  // final correction: g + (x - g^2) * h
}

// position (float32, as State holds it) of the trajectory xq at time t
__device__ __forceinline__ void position_at(int M, int dim, double dt, double t, double z_2d, const double* xq, float (&pos)[3]) {
    const int P = 6 * M;
    int ms = -1;
    double tn = 0, end = 0;
    for (int idx = 0; idx < M; idx++) {  // getPointAt's segment search (src/trajectory.cpp:121-136)
        end += dt;
        if (t < end) {
            ms = idx;
            tn = 1 - (end - t) / dt;
            break;
        }
    }
    if (ms < 0) {
        ms = M - 1;
        tn = 1.0;
    }
    for (int k = 0; k < 3; k++) {
        double c[6];
        for (int i = 0; i < 6; i++) c[i] = (k < dim) ? (double)(float)xq[k * P + 6 * ms + i] : (double)(float)z_2d;
        pos[k] = (float)bern<5>(c, tn);
    }
}

__global__ __launch_bounds__(kSafT) void safety_metrics_kernel(int M, int dim, double dt, int64_t n_agents, int64_t first_agent,
                                                               int64_t n_total, int n_samples, double record_time_step, double z_2d,
                                                               const double* __restrict__ x_all, const double* __restrict__ radius,
                                                               const double* __restrict__ downwash,
                                                               const lscqp_header* __restrict__ hdr, lscqp_safety* __restrict__ out) {
    __shared__ float tp[kSafT][3];
    __shared__ double tr[kSafT], tdr[kSafT];  // r_j, downwash_j * r_j
    __shared__ double red_v[kSafT];
    __shared__ int64_t red_k[kSafT];
    const int li = threadIdx.x % kSafI, slice = threadIdx.x / kSafI;
    const int64_t a = (int64_t)blockIdx.x * kSafI + li;
    const bool live = a < n_agents;
    const int64_t gi = first_agent + (live ? a : 0);
    const int nv = dim * 6 * M;
    const double ri = radius[gi], dri = downwash[gi] * ri;
    double best = 1e300;
    int64_t best_key = INT64_MAX;  // (sample, j): the reference keeps the first strict minimum in this order
    double vex[3] = {0, 0, 0}, aex[3] = {0, 0, 0};
    for (int s = 0; s < n_samples; s++) {
        const double t = s * record_time_step;
        float pi[3];
        position_at(M, dim, dt, t, z_2d, x_all + gi * nv, pi);
        if (slice == 0 && live) {  // velocity / acceleration excess at t (:560-572)
            const double* xq = x_all + gi * nv;
            const int P = 6 * M;
            int ms = M - 1;
            double tn = 1.0, end = 0;
            for (int idx = 0; idx < M; idx++) {
                end += dt;
                if (t < end) {
                    ms = idx;
                    tn = 1 - (end - t) / dt;
                    break;
                }
            }
            for (int k = 0; k < dim; k++) {
                double c[6], d1[6] = {0, 0, 0, 0, 0, 0}, d2[6] = {0, 0, 0, 0, 0, 0};
                for (int i = 0; i < 6; i++) c[i] = (double)(float)xq[k * P + 6 * ms + i];
                for (int i = 0; i < 5; i++) d1[i] = (c[i + 1] - c[i]) * (5.0 / dt);
                for (int i = 0; i < 4; i++) d2[i] = (d1[i + 1] - d1[i]) * (4.0 / dt);
                const double vel = (double)(float)bern<4>(d1, tn), acc = (double)(float)bern<3>(d2, tn);
                const double ve = (vel - hdr[a].vmax[k]) / hdr[a].vmax[k], ae = (acc - hdr[a].amax[k]) / hdr[a].amax[k];
                vex[k] = (ve > 0 && ve > vex[k]) ? ve : vex[k];
                aex[k] = (ae > 0 && ae > aex[k]) ? ae : aex[k];
            }
        }
        for (int64_t tile = 0; tile < n_total; tile += kSafT) {
            __syncthreads();
            const int64_t gj = tile + threadIdx.x;
            if (gj < n_total) {
                float pj[3];
                position_at(M, dim, dt, t, z_2d, x_all + gj * nv, pj);
                tp[threadIdx.x][0] = pj[0], tp[threadIdx.x][1] = pj[1], tp[threadIdx.x][2] = pj[2];
                tr[threadIdx.x] = radius[gj];
                tdr[threadIdx.x] = downwash[gj] * tr[threadIdx.x];
            }
            __syncthreads();
            const int cnt = (int)((n_total - tile < kSafT) ? n_total - tile : kSafT);
            for (int jj = slice; jj < cnt; jj += kSafS) {
                const int64_t j = tile + jj;
                // 16.8 M pairs at 4096 agents: the two divisions and the square root per pair go through v_rcp_f64 / v_rsq_f64
                // plus Newton steps (full fp64 accuracy to the last bit or two) instead of the IEEE sequences
                const double rsum = ri + tr[jj];
                const double irs = post_rcp(rsum);
                const double dwn = (dri + tdr[jj]) * irs;  // :505-507
                // point3d delta = p_i - p_j (float); delta.z /= downwash; norm() = sqrt of the float sum of squares
                const float dx = pi[0] - tp[jj][0], dy = pi[1] - tp[jj][1];
                const float dz = (float)((double)(pi[2] - tp[jj][2]) * post_rcp(dwn));
                float nsq;
                {
#pragma clang fp contract(off)
                    nsq = dx * dx + dy * dy + dz * dz;
                }
                const double ratio = post_sqrt((double)nsq) * irs;
                const int64_t key = (int64_t)s * n_total + j;
                const bool take = (j != gi) && (ratio < best || (ratio == best && key < best_key));
                best = take ? ratio : best;
                best_key = take ? key : best_key;
            }
        }
    }
    red_v[threadIdx.x] = best;
    red_k[threadIdx.x] = best_key;
    __syncthreads();
    if (slice == 0 && live) {
        for (int q = 1; q < kSafS; q++) {
            const double v = red_v[q * kSafI + li];
            const int64_t k = red_k[q * kSafI + li];
            const bool take = v < best || (v == best && k < best_key);
            best = take ? v : best;
            best_key = take ? k : best_key;
        }
        lscqp_safety R;
        const bool any = best_key != INT64_MAX;
        R.safety_ratio = any ? best : INFINITY;  // SP_INFINITY when there is no other agent
        R.closest_agent = any ? (int32_t)(best_key % n_total) : -1;
        R.sample = any ? (int32_t)(best_key / n_total) : -1;
        for (int k = 0; k < 3; k++) {
            R.vel_excess_ratio[k] = vex[k];
            R.acc_excess_ratio[k] = aex[k];
        }
        out[a] = R;
    }
}

// The obstacle leg of the same loop (src/multi_sync_simulator.cpp:527-557).  A mission has a handful of obstacles (mission.on) against
// up to thousands of agents: one lane per agent, every lane walks the samples of its own plan and the obstacle table (uniform
// addresses: scalar loads).  IEEE division and square root (few pairs: the agent-agent kernel's fast
// reciprocals buy nothing here), so the figures are bit-for-bit the double arithmetic of the reference's expression.
constexpr int kObsT = 256;
__global__ __launch_bounds__(kObsT) void safety_obstacles_kernel(int M, int dim, double dt, int64_t n_agents, int64_t first_agent, int n_samples,
                                                                 double record_time_step, double z_2d, const double* __restrict__ x_all,
                                                                 const double* __restrict__ radius, const double* __restrict__ downwash,
                                                                 int n_obstacles, const lscqp_obstacle* __restrict__ obs,
                                                                 lscqp_safety_obs* __restrict__ out) {
    const int64_t a = (int64_t)blockIdx.x * kObsT + threadIdx.x;
    if (a >= n_agents) return;
    const int64_t gi = first_agent + a;
    const int nv = dim * 6 * M;
    const double ri = radius[gi], dri = ri * downwash[gi];
    double best = INFINITY;
    int bo = -1, bs = -1;
    for (int s = 0; s < n_samples; s++) {
        float pi[3];
        position_at(M, dim, dt, s * record_time_step, z_2d, x_all + gi * nv, pi);
        for (int o = 0; o < n_obstacles; o++) {
            if (obs[o].type == LSCQP_OBSTACLE_REAL) continue;  // :531-532
            const double ro = obs[o].radius;
            const double dwn = (ro * obs[o].downwash + dri) / (ri + ro);  // :538-540
            const float dx = pi[0] - (float)obs[o].position[0], dy = pi[1] - (float)obs[o].position[1];
            const float dz = (float)((double)(pi[2] - (float)obs[o].position[2]) / dwn);
            float nsq;
            {
#pragma clang fp contract(off)
                nsq = dx * dx + dy * dy + dz * dz;
            }
            const double ratio = sqrt((double)nsq) / (ri + ro);
            if (ratio < best) {  // the reference's strict <: first (sample, obstacle) attaining the minimum
                best = ratio;
                bo = o;
                bs = s;
            }
        }
    }
    lscqp_safety_obs R;
    R.safety_ratio_obs = best;
    R.closest_obstacle = bo;
    R.sample = bs;
    out[a] = R;
}

}  // namespace lscpost

extern "C" int lscqp_set_error_(int code, const char* msg);

extern "C" int lscqp_safety_obstacles_raw_(int M, int dim, double dt, int64_t n_agents, int64_t first_agent, int n_samples, double record_time_step,
                                           double z_2d, const double* d_x_all, const double* d_radius, const double* d_downwash, int n_obstacles,
                                           const lscqp_obstacle* d_obstacles, lscqp_safety_obs* d_out, void* stream) {
    if (n_agents == 0) return LSCQP_OK;
    const unsigned blocks = (unsigned)((n_agents + lscpost::kObsT - 1) / lscpost::kObsT);
    hipLaunchKernelGGL(lscpost::safety_obstacles_kernel, dim3(blocks), dim3(lscpost::kObsT), 0, (hipStream_t)stream, M, dim, dt, n_agents, first_agent,
                       n_samples, record_time_step, z_2d, d_x_all, d_radius, d_downwash, n_obstacles, d_obstacles, d_out);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return lscqp_set_error_(LSCQP_ERR_HIP, (std::string("HIP launch failed: ") + hipGetErrorString(e)).c_str());
    return LSCQP_OK;
}

extern "C" int lscqp_safety_metrics_raw_(int M, int dim, double dt, int64_t n_agents, int64_t first_agent, int64_t n_total, int n_samples,
                                         double record_time_step, double z_2d, const double* d_x_all, const double* d_radius,
                                         const double* d_downwash, const lscqp_header* d_hdr, lscqp_safety* d_out, void* stream) {
    if (n_agents == 0) return LSCQP_OK;
    const unsigned blocks = (unsigned)((n_agents + lscpost::kSafI - 1) / lscpost::kSafI);
    hipLaunchKernelGGL(lscpost::safety_metrics_kernel, dim3(blocks), dim3(lscpost::kSafT), 0, (hipStream_t)stream, M, dim, dt, n_agents,
                       first_agent, n_total, n_samples, record_time_step, z_2d, d_x_all, d_radius, d_downwash, d_hdr, d_out);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return lscqp_set_error_(LSCQP_ERR_HIP, (std::string("HIP launch failed: ") + hipGetErrorString(e)).c_str());
    return LSCQP_OK;
}

extern "C" int lscqp_validate_step_raw_(int M, int dim, int use_sfc, double dt, int64_t n, double time_step, double z_2d, const double* d_x,
                                        const lscqp_header* d_hdr, const lscqp_box* d_sfc, int32_t* d_valid, double* d_state,
                                        void* stream) {
    if (n == 0) return LSCQP_OK;
    hipLaunchKernelGGL(lscpost::validate_step_kernel, dim3((unsigned)n), dim3(lscpost::kThreads), 0, (hipStream_t)stream, M, dim, use_sfc, dt, n,
                       time_step, z_2d, d_x, d_hdr, d_sfc, d_valid, d_state, lscpost::Commit{nullptr, nullptr, nullptr, nullptr, nullptr});
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return lscqp_set_error_(LSCQP_ERR_HIP, (std::string("HIP launch failed: ") + hipGetErrorString(e)).c_str());
    return LSCQP_OK;
}

// (library-internal, lscplan.hip) commit + isSolValid + doStep of the local agents in one launch; d_x_plan / d_goal: the local block
extern "C" int lscqp_commit_validate_raw_(int M, int dim, int use_sfc, double dt, int64_t n, double time_step, double z_2d, const int32_t* d_qp_status,
                                          const double* d_x_new, const double* d_x_init, double* d_x_plan, double* d_goal, const lscqp_header* d_hdr,
                                          const lscqp_box* d_sfc, int32_t* d_valid, double* d_state, void* stream) {
    if (n == 0) return LSCQP_OK;
    hipLaunchKernelGGL(lscpost::validate_step_kernel, dim3((unsigned)n), dim3(lscpost::kThreads), 0, (hipStream_t)stream, M, dim, use_sfc, dt, n,
                       time_step, z_2d, d_x_new, d_hdr, d_sfc, d_valid, d_state, lscpost::Commit{d_qp_status, d_x_new, d_x_init, d_x_plan, d_goal});
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return lscqp_set_error_(LSCQP_ERR_HIP, (std::string("HIP launch failed: ") + hipGetErrorString(e)).c_str());
    return LSCQP_OK;
}
