// lscgoal.hip — the goal LP of the reference in closed form, on the device, gfx950 only.
//
// Replaces GoalOptimizer::solve (reference src/goal_optimizer.cpp:7-70; model built by populatebyrow, :72-147), the
// second CPLEX call site of the planner: one variable t in [0, 1 + SP_EPSILON_FLOAT],
//     min t   s.t.   n_r . ((g - w) t + w - p_r) - d_r >= 0
// over the SFC faces of the LAST segment (:122-137, Box::convertToLSCs, src/collision_constraints.cpp:37-59) and the
// LSC rows (oi, M-1, n) of every obstacle (:140-155; normals shorter than SP_EPSILON_FLOAT skipped), and the goal is
// (g - w) t* + w (:55).  A one-variable LP needs no solver: every row a_r t + c_r >= 0 is a lower bound on t (a_r > 0),
// an upper bound (a_r < 0) or a feasibility condition (a_r = 0); t* is the largest lower bound.
// One wavefront per agent; the rows are read where the QP's ABI keeps them (packed: n.c >= b, b = d + n.p).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/lscqp.h"

namespace lscgoal {

constexpr int kThreads = 256;  // four agents per workgroup
constexpr double kEpsFloat = 1e-5;  // SP_EPSILON_FLOAT (reference include/sp_const.hpp)
constexpr double kFeasTol = 1e-9;   // slack allowed on a_r = 0 rows and on L <= U (CPLEX's own LP tolerance is 1e-6)

// One WAVEFRONT per agent, a lane per row (round 3; a lane per agent walked its rows one dependent load after the other: 15 us for 20
// neighbours).  Every row is a bound of its own and the LP's value is a maximum of lower bounds against a minimum of upper ones --
// order-independent, so the result is the sequential one bit for bit.
// fin_dt > 0 (lscqp_plan's chain): the agent's header is finished here as well -- the goal held as a point3d (float32) and
// terminal_segments from it, what lscplan.hip's finalize_goal_kernel does when the goal LP is off (one graph node less per replan).
__global__ __launch_bounds__(kThreads) void goal_kernel(int M, int dim, int use_sfc, int rows_f32, double fin_dt, int64_t n, lscqp_header* __restrict__ hdr,
                                                        const lscqp_row* __restrict__ rows, const uint64_t* __restrict__ row_offsets,
                                                        const lscqp_box* __restrict__ sfc, int32_t* __restrict__ status) {
    const int64_t q = (int64_t)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (q >= n) return;
    lscqp_header* H = hdr + q;
    double g[3], w[3], dgw[3];
    double dist2 = 0;
    for (int k = 0; k < 3; k++) {
        g[k] = H->goal[k];
        w[k] = H->next_waypoint[k];
        dgw[k] = g[k] - w[k];
        dist2 += dgw[k] * dgw[k];
    }
    // what the agent's header is left with: the new goal (the old one if the LP is infeasible) and the status
    auto finish = [&](double g0, double g1, double g2, int st) {
        if (fin_dt > 0.0) {  // agent.current_goal_point is a point3d; getTerminalSegments_old (src/traj_optimizer.cpp:530-538) in float32
#pragma clang fp contract(off)
            const float f0 = (float)g0, f1 = (float)g1, f2 = (float)g2;
            g0 = (double)f0, g1 = (double)f1, g2 = (double)f2;
            const float d0 = f0 - (float)H->p0[0], d1 = f1 - (float)H->p0[1], d2 = f2 - (float)H->p0[2];
            const float nsq = d0 * d0 + d1 * d1 + d2 * d2;
            const double ideal_flight_time = sqrt((double)nsq) / H->nominal_velocity;
            const int ts = (int)((M * fin_dt - ideal_flight_time + 1e-9) / fin_dt);
            if (lane == 0) H->terminal_segments = ts > 1 ? ts : 1;
        }
        if (lane < 3) H->goal[lane] = lane == 0 ? g0 : (lane == 1 ? g1 : g2);
        if (lane == 0) status[q] = st;
    };
    if (sqrt(dist2) < kEpsFloat) {  // :12-14
        finish(w[0], w[1], w[2], LSCQP_STATUS_OPTIMAL);
        return;
    }
    double lo = 0.0, hi = 1.0 + kEpsFloat;  // variable bounds (:112)
    bool bad = false;
    auto row = [&](double a, double c) {  // a t + c >= 0
        if (a > 0.0)
            lo = fmax(lo, -c / a);
        else if (a < 0.0)
            hi = fmin(hi, -c / a);
        else
            bad = bad || (c < -kFeasTol);
    };
    const int n_obs = H->n_obs;
    const uint64_t roff = n_obs > 0 ? row_offsets[q] : 0;
    const int n_faces = use_sfc ? 2 * dim : 0;
    for (int r = lane; r < n_faces + n_obs; r += 64) {
        if (r < n_faces) {  // faces of the last segment's box: +e_k . c - bmin_k >= 0, -e_k . c + bmax_k >= 0
            const lscqp_box* B = sfc + q * M + (M - 1);
            const int k = r >> 1;
            const double dk = k == 0 ? dgw[0] : (k == 1 ? dgw[1] : dgw[2]), wk = k == 0 ? w[0] : (k == 1 ? w[1] : w[2]);
            if ((r & 1) == 0)
                row(dk, wk - B->bmin[k]);
            else
                row(-dk, B->bmax[k] - wk);
            continue;
        }
        const int o = r - n_faces;
        const size_t ri = roff + ((size_t)o * M + (M - 1)) * 6 + 5;  // getLSC(oi, M-1, n)
        lscqp_row rw;
        if (rows_f32) {
            const lscqp_row_f32 f = reinterpret_cast<const lscqp_row_f32*>(rows)[ri];
            rw = lscqp_row{(double)f.nx, (double)f.ny, (double)f.nz, (double)f.b};
        } else {
            rw = rows[ri];
        }
        if (sqrt(rw.nx * rw.nx + rw.ny * rw.ny + rw.nz * rw.nz) < kEpsFloat) continue;  // :142-144
        double a = rw.nx * dgw[0] + rw.ny * dgw[1], c = rw.nx * w[0] + rw.ny * w[1];
        if (dim == 3) {
            a += rw.nz * dgw[2];
            c += rw.nz * w[2];
        }
        row(a, c - rw.b);
    }
    for (int d = 32; d >= 1; d >>= 1) {  // (no NaN can arise: a != 0 on the paths that divide)
        lo = fmax(lo, __shfl_xor(lo, d));
        hi = fmin(hi, __shfl_xor(hi, d));
    }
    bad = __builtin_amdgcn_ballot_w64(bad) != 0;
    if (bad || lo > hi + kFeasTol) {  // reference: CPLEX reports infeasible -> throw PlanningReport::QPFAILED (:57-69)
        finish(g[0], g[1], g[2], LSCQP_STATUS_INFEASIBLE);
        return;
    }
    const double t = fmin(lo, hi);
    finish(dgw[0] * t + w[0], dgw[1] * t + w[1], dgw[2] * t + w[2], LSCQP_STATUS_OPTIMAL);
}

}  // namespace lscgoal

extern "C" int lscqp_set_error_(int code, const char* msg);

extern "C" int lscqp_goal_fin_raw_(int M, int dim, int use_sfc, int rows_f32, double fin_dt, int64_t n, lscqp_header* d_hdr, const lscqp_row* d_rows,
                                   const uint64_t* d_row_offsets, const lscqp_box* d_sfc, int32_t* d_status, void* stream) {
    if (n == 0) return LSCQP_OK;
    const int64_t per = lscgoal::kThreads / 64;
    const unsigned blocks = (unsigned)((n + per - 1) / per);
    hipLaunchKernelGGL(lscgoal::goal_kernel, dim3(blocks), dim3(lscgoal::kThreads), 0, (hipStream_t)stream, M, dim, use_sfc, rows_f32, fin_dt, n, d_hdr, d_rows,
                       d_row_offsets, d_sfc, d_status);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return lscqp_set_error_(LSCQP_ERR_HIP, (std::string("HIP launch failed: ") + hipGetErrorString(e)).c_str());
    return LSCQP_OK;
}

extern "C" int lscqp_goal_raw_(int M, int dim, int use_sfc, int rows_f32, int64_t n, lscqp_header* d_hdr, const lscqp_row* d_rows,
                               const uint64_t* d_row_offsets, const lscqp_box* d_sfc, int32_t* d_status, void* stream) {
    return lscqp_goal_fin_raw_(M, dim, use_sfc, rows_f32, 0.0, n, d_hdr, d_rows, d_row_offsets, d_sfc, d_status, stream);
}
