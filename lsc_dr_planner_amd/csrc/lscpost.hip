// lscpost.hip — post-solve epilogue on the device, gfx950 only: what the planner does with a solved
// trajectory before the next replan.
//
// Replaces, for a batch of solved agents,
//   TrajPlanner::isSolValid          reference src/traj_planner.cpp:990-1045  (SFC containment + dynamic limits at the
//                                    simulation step; its LSC check is commented out in the reference and stays out)
//   Trajectory::getStateAt           src/trajectory.cpp:156-170 (getPointAt :111-153, derivative :180-199)
//   AgentManager::doStep             src/agent_manager.cpp:29-50  (the agent's next state = ideal future state)
// Control points are first truncated to float32 exactly as TrajOptResult::desired_traj holds them
// (src/traj_optimizer.cpp:71-83); in 2-D missions z := world_z_2d.  One lane per agent.
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

__global__ __launch_bounds__(kThreads) void validate_step_kernel(int M, int dim, int use_sfc, double dt, int64_t n, double time_step, double z_2d,
                                                                 const double* __restrict__ x, const lscqp_header* __restrict__ hdr,
                                                                 const lscqp_box* __restrict__ sfc, int32_t* __restrict__ valid,
                                                                 double* __restrict__ state) {
    const int64_t q = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (q >= n) return;
    const int P = 6 * M;
    const double* xq = x + q * dim * P;
    auto cp = [&](int k, int m, int i) -> double {  // desired_traj[m][i](k), float32
        return (k < dim) ? (double)(float)xq[k * P + 6 * m + i] : (double)(float)z_2d;
    };
    bool ok = true;
    if (use_sfc) {  // :992-1010: segment 0 from control point phi = 3 on, whole segments afterwards
        for (int m = 0; m < M; m++) {
            const lscqp_box B = sfc[q * M + m];
            for (int i = (m == 0 ? 3 : 0); i < 6; i++)
                for (int k = 0; k < 3; k++) {
                    const double c = cp(k, m, i);
                    ok = ok && (c > (double)(float)B.bmin[k] - kEpsFloat) && (c < (double)(float)B.bmax[k] + kEpsFloat);
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
    const lscqp_header* H = hdr + q;
    double* S = state + q * 9;
    for (int k = 0; k < 3; k++) {
        double c[6], d1[6] = {0, 0, 0, 0, 0, 0}, d2[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 6; i++) c[i] = cp(k, ms, i);
        for (int i = 0; i < 5; i++) d1[i] = (c[i + 1] - c[i]) * (5.0 / dt);   // derivative(), :183-199
        for (int i = 0; i < 4; i++) d2[i] = (d1[i + 1] - d1[i]) * (4.0 / dt);
        const double pos = bern<5>(c, tn), vel = bern<4>(d1, tn), acc = bern<3>(d2, tn);
        // State holds point3d (float32); doStep: 2-D missions pin z to world_z_2d (src/agent_manager.cpp:40-42)
        S[k] = (k < dim) ? (double)(float)pos : (double)(float)z_2d;
        S[3 + k] = (k < dim) ? (double)(float)vel : 0.0;
        S[6 + k] = (k < dim) ? (double)(float)acc : 0.0;
        if (k < dim) {  // :1030-1041, 1 % tolerance
            ok = ok && !(fabs(S[3 + k]) > H->vmax[k] * 1.01) && !(fabs(S[6 + k]) > H->amax[k] * 1.01);
        }
    }
    valid[q] = ok ? 1 : 0;
}

}  // namespace lscpost

extern "C" int lscqp_set_error_(int code, const char* msg);

extern "C" int lscqp_validate_step_raw_(int M, int dim, int use_sfc, double dt, int64_t n, double time_step, double z_2d, const double* d_x,
                                        const lscqp_header* d_hdr, const lscqp_box* d_sfc, int32_t* d_valid, double* d_state,
                                        void* stream) {
    if (n == 0) return LSCQP_OK;
    const unsigned blocks = (unsigned)((n + lscpost::kThreads - 1) / lscpost::kThreads);
    hipLaunchKernelGGL(lscpost::validate_step_kernel, dim3(blocks), dim3(lscpost::kThreads), 0, (hipStream_t)stream, M, dim, use_sfc, dt, n,
                       time_step, z_2d, d_x, d_hdr, d_sfc, d_valid, d_state);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return lscqp_set_error_(LSCQP_ERR_HIP, (std::string("HIP launch failed: ") + hipGetErrorString(e)).c_str());
    return LSCQP_OK;
}
