// lscgoal.hip — the goal LP of the reference in closed form, on the device, gfx950 only.
//
// Replaces GoalOptimizer::solve (reference src/goal_optimizer.cpp:7-70; model built by populatebyrow, :72-147), the
// second CPLEX call site of the planner: one variable t in [0, 1 + SP_EPSILON_FLOAT],
//     min t   s.t.   n_r . ((g - w) t + w - p_r) - d_r >= 0
// over the SFC faces of the LAST segment (:122-137, Box::convertToLSCs, src/collision_constraints.cpp:37-59) and the
// LSC rows (oi, M-1, n) of every obstacle (:140-155; normals shorter than SP_EPSILON_FLOAT skipped), and the goal is
// (g - w) t* + w (:55).  A one-variable LP needs no solver: every row a_r t + c_r >= 0 is a lower bound on t (a_r > 0),
// an upper bound (a_r < 0) or a feasibility condition (a_r = 0); t* is the largest lower bound.
// One lane per agent; the rows are read where the QP's ABI keeps them (packed: n.c >= b, b = d + n.p).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/lscqp.h"

namespace lscgoal {

constexpr int kThreads = 64;
constexpr double kEpsFloat = 1e-5;  // SP_EPSILON_FLOAT (reference include/sp_const.hpp)
constexpr double kFeasTol = 1e-9;   // slack allowed on a_r = 0 rows and on L <= U (CPLEX's own LP tolerance is 1e-6)

__global__ __launch_bounds__(kThreads) void goal_kernel(int M, int dim, int use_sfc, int rows_f32, int64_t n, lscqp_header* __restrict__ hdr,
                                                        const lscqp_row* __restrict__ rows, const uint64_t* __restrict__ row_offsets,
                                                        const lscqp_box* __restrict__ sfc, int32_t* __restrict__ status) {
    const int64_t q = (int64_t)blockIdx.x * kThreads + threadIdx.x;
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
    if (sqrt(dist2) < kEpsFloat) {  // :12-14
        for (int k = 0; k < 3; k++) H->goal[k] = w[k];
        status[q] = LSCQP_STATUS_OPTIMAL;
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
    if (use_sfc) {  // faces of the last segment's box: +e_k . c - bmin_k >= 0, -e_k . c + bmax_k >= 0
        const lscqp_box* B = sfc + q * M + (M - 1);
        for (int k = 0; k < dim; k++) {
            row(dgw[k], w[k] - B->bmin[k]);
            row(-dgw[k], B->bmax[k] - w[k]);
        }
    }
    const int n_obs = H->n_obs;
    const uint64_t roff = n_obs > 0 ? row_offsets[q] : 0;
    for (int o = 0; o < n_obs; o++) {
        const size_t ri = roff + ((size_t)o * M + (M - 1)) * 6 + 5;  // getLSC(oi, M-1, n)
        lscqp_row r;
        if (rows_f32) {
            const lscqp_row_f32 f = reinterpret_cast<const lscqp_row_f32*>(rows)[ri];
            r = lscqp_row{(double)f.nx, (double)f.ny, (double)f.nz, (double)f.b};
        } else {
            r = rows[ri];
        }
        if (sqrt(r.nx * r.nx + r.ny * r.ny + r.nz * r.nz) < kEpsFloat) continue;  // :142-144
        double a = r.nx * dgw[0] + r.ny * dgw[1], c = r.nx * w[0] + r.ny * w[1];
        if (dim == 3) {
            a += r.nz * dgw[2];
            c += r.nz * w[2];
        }
        row(a, c - r.b);
    }
    if (bad || lo > hi + kFeasTol) {  // reference: CPLEX reports infeasible -> throw PlanningReport::QPFAILED (:57-69)
        status[q] = LSCQP_STATUS_INFEASIBLE;
        return;
    }
    const double t = fmin(lo, hi);
    for (int k = 0; k < 3; k++) H->goal[k] = dgw[k] * t + w[k];
    status[q] = LSCQP_STATUS_OPTIMAL;
}

}  // namespace lscgoal

extern "C" int lscqp_set_error_(int code, const char* msg);

extern "C" int lscqp_goal_raw_(int M, int dim, int use_sfc, int rows_f32, int64_t n, lscqp_header* d_hdr, const lscqp_row* d_rows,
                               const uint64_t* d_row_offsets, const lscqp_box* d_sfc, int32_t* d_status, void* stream) {
    if (n == 0) return LSCQP_OK;
    const unsigned blocks = (unsigned)((n + lscgoal::kThreads - 1) / lscgoal::kThreads);
    hipLaunchKernelGGL(lscgoal::goal_kernel, dim3(blocks), dim3(lscgoal::kThreads), 0, (hipStream_t)stream, M, dim, use_sfc, rows_f32, n, d_hdr, d_rows,
                       d_row_offsets, d_sfc, d_status);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return lscqp_set_error_(LSCQP_ERR_HIP, (std::string("HIP launch failed: ") + hipGetErrorString(e)).c_str());
    return LSCQP_OK;
}
