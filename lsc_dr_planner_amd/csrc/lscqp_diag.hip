// lscqp_diag.hip — what the reference tells its user when a QP fails (include/lscqp.h, "failure diagnostics").
//
// On a solver failure the reference (1) exports the model as a CPLEX LP file and runs the conflict refiner to name the rows that
// cannot hold together (src/traj_optimizer.cpp:103-137, and :45-52 with param.log_solver), and (2) its caller walks over every SFC
// and LSC row of `initial_traj` -- the trajectory it falls back to -- and prints the violated ones with obstacle, segment, control
// point and margin (src/traj_planner.cpp:767-797).  Here:
//   lscqp_diagnose[_device]  evaluates EVERY row of the reference's model (populatebyrow, src/traj_optimizer.cpp:238-511) on a given
//                            trajectory, one workgroup per instance: per row family the largest violation and the number of
//                            violated rows, and the most violated row by name (family, obstacle, segment, control point, axis).
//                            Run on initial_traj it is the caller's debug loop; run on a solver's last iterate it names the rows an
//                            infeasible instance cannot satisfy (the rows an interior-point iteration leaves violated are the
//                            members of the conflict).
//   lscqp_dump_instance      writes one instance as a CPLEX LP file, row for row and name for name as populatebyrow builds it
//                            (variables x_m_i / y_m_i / z_m_i, :238-283), host side, no device needed.
// Neither touches the solve path.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/lscqp.h"

extern "C" int lscqp_set_error_(int code, const char* msg);
extern "C" const lscqp_class_desc* lscqp_class_desc_of_(lscqp_handle h);

namespace {

int fail(int code, const std::string& m) { return lscqp_set_error_(code, m.c_str()); }

struct DiagClass {
    int M, dim, es, use_sfc, rsfc, rows_f32;
    double dt, comm_range;
    double world_min[3], world_max[3];
};

struct Worst {  // running argmax of one lane
    double v;
    int fam, obs, seg, pt, axis;
};

__device__ __forceinline__ void take(Worst& w, double v, int fam, int obs, int seg, int pt, int axis) {
    if (v > w.v) {
        w.v = v;
        w.fam = fam;
        w.obs = obs;
        w.seg = seg;
        w.pt = pt;
        w.axis = axis;
    }
}

// One wavefront per instance.  x: control points in the reference variable order x[k*P + 6m + i], world frame.
__global__ __launch_bounds__(64) void diagnose_kernel(DiagClass c, int64_t n, const lscqp_header* __restrict__ hdr, const lscqp_row* __restrict__ rows,
                                                      const uint64_t* __restrict__ row_offsets, const lscqp_box* __restrict__ sfc,
                                                      const double* __restrict__ x, double tol, lscqp_diag* __restrict__ out) {
    const int64_t q = blockIdx.x;
    if (q >= n) return;
    const int lane = threadIdx.x;
    const int M = c.M, dim = c.dim, P = 6 * M;
    const lscqp_header* H = hdr + q;
    const double* X = x + q * (int64_t)dim * P;
    auto at = [&](int k, int m, int i) -> double { return X[k * P + 6 * m + i]; };
    double fam_worst[LSCQP_ROW_FAMILIES];
    int fam_cnt[LSCQP_ROW_FAMILIES];
#pragma unroll
    for (int f = 0; f < LSCQP_ROW_FAMILIES; f++) {
        fam_worst[f] = -1e300;
        fam_cnt[f] = 0;
    }
    Worst w{-1e300, -1, -1, -1, -1, -1};
    auto row = [&](int fam, double v, int obs, int seg, int pt, int axis) {
#pragma unroll
        for (int f = 0; f < LSCQP_ROW_FAMILIES; f++)
            if (f == fam) {
                fam_worst[f] = fmax(fam_worst[f], v);
                fam_cnt[f] += (v > tol) ? 1 : 0;
            }
        take(w, v, fam, obs, seg, pt, axis);
    };
    // variable bounds (:252-265) and corridor faces (:372-397): every control point but the initial state
    for (int e = lane; e < dim * P; e += 64) {
        const int k = e / P, m = (e % P) / 6, i = e % 6;
        if (m == 0 && i < 3) continue;
        double lo = c.world_min[k], hi = c.world_max[k];
        if (c.rsfc && k == 2 && m == 0) {
            lo = -100.0;
            hi = 100.0;
        }
        const double v = at(k, m, i);
        row(LSCQP_ROW_BOUND, fmax(lo - v, v - hi), -1, m, i, k);
        if (c.use_sfc && sfc) {
            const lscqp_box& b = sfc[q * M + m];
            row(LSCQP_ROW_SFC, fmax(b.bmin[k] - v, v - b.bmax[k]), -1, m, i, k);
        }
    }
    // LSC / BVC half-spaces (:399-437): n.(c - p) - d >= 0 <=> n.c >= b; rows with a normal shorter than 1e-5 do not exist (:409-411)
    {
        const int n_obs = H->n_obs;
        const uint64_t r0 = (n_obs > 0 && row_offsets) ? row_offsets[q] : 0;
        for (int e = lane; e < n_obs * P; e += 64) {
            const int o = e / P, m = (e % P) / 6, i = e % 6;
            if (m == 0 && i < 3) continue;
            double nx, ny, nz, b;
            if (c.rows_f32) {
                const float4 f = reinterpret_cast<const float4*>(rows)[r0 + e];
                nx = f.x, ny = f.y, nz = f.z, b = f.w;
            } else {
                const lscqp_row r = rows[r0 + e];
                nx = r.nx, ny = r.ny, nz = r.nz, b = r.b;
            }
            if (sqrt(nx * nx + ny * ny + nz * nz) < 1e-5) continue;
            const double lhs = nx * at(0, m, i) + ny * at(1, m, i) + (dim == 3 ? nz * at(2, m, i) : 0.0);
            row(LSCQP_ROW_LSC, b - lhs, o, m, i, -1);
        }
    }
    // dynamic limits (:439-474), in the reference's units (m/s, m/s^2)
    const double sv = 5.0 / c.dt, sa = 20.0 / (c.dt * c.dt);
    for (int e = lane; e < dim * M * 5; e += 64) {
        const int k = e / (5 * M), m = (e % (5 * M)) / 5, i = e % 5;
        if (!(m == 0 && i < 2)) row(LSCQP_ROW_VEL, fabs(sv * (at(k, m, i + 1) - at(k, m, i))) - H->vmax[k], -1, m, i, k);
        if (i < 4 && !(m == 0 && i < 1))
            row(LSCQP_ROW_ACC, fabs(sa * (at(k, m, i + 2) - 2.0 * at(k, m, i + 1) + at(k, m, i))) - H->amax[k], -1, m, i, k);
    }
    // communication range (:476-500): pairs (mi <= m) and the waypoint rows
    if (c.comm_range > 0) {
        const double rho = 0.5 * c.comm_range - H->radius, rho_w = 0.5 * c.comm_range - 1e-5;
        for (int e = lane; e < dim * M * M; e += 64) {
            const int k = e / (M * M), mi = (e % (M * M)) / M, m = e % M;
            if (m >= mi) row(LSCQP_ROW_COMM_PAIR, fabs(at(k, m, 5) - at(k, mi, 0)) - rho, mi, m, 5, k);  // (obstacle field: mi)
        }
        for (int e = lane; e < dim * M; e += 64) {
            const int k = e / M, m = e % M;
            row(LSCQP_ROW_COMM_WAYPOINT, fabs(at(k, m, 5) - H->next_waypoint[k]) - rho_w, -1, m, 5, k);
        }
    }
    // equalities (:318-368, 502-511): |residual| of the row as the reference scales it
    for (int e = lane; e < dim * M; e += 64) {
        const int k = e / M, m = e % M;
        if (m == 0) {
            row(LSCQP_ROW_EQUALITY, fabs(at(k, 0, 0) - H->p0[k]), -1, 0, 0, k);
            row(LSCQP_ROW_EQUALITY, fabs(sv * (at(k, 0, 1) - at(k, 0, 0)) - H->v0[k]), -1, 0, 1, k);
            row(LSCQP_ROW_EQUALITY, fabs(sa * (at(k, 0, 2) - 2.0 * at(k, 0, 1) + at(k, 0, 0)) - H->a0[k]), -1, 0, 2, k);
        } else {
            // join (m-1) -> m.  The first join is unscaled (:341-352), the later ones carry dt^-j n.. factors (buildAeqBase, :180-214)
            const double f1 = (m == 1) ? 1.0 : sv, f2 = (m == 1) ? 1.0 : sa;
            row(LSCQP_ROW_EQUALITY, fabs(at(k, m - 1, 5) - at(k, m, 0)), -1, m, 0, k);
            row(LSCQP_ROW_EQUALITY, fabs(f1 * ((at(k, m, 1) - at(k, m, 0)) - (at(k, m - 1, 5) - at(k, m - 1, 4)))), -1, m, 1, k);
            row(LSCQP_ROW_EQUALITY,
                fabs(f2 * ((at(k, m, 2) - 2.0 * at(k, m, 1) + at(k, m, 0)) - (at(k, m - 1, 5) - 2.0 * at(k, m - 1, 4) + at(k, m - 1, 3)))), -1, m, 2, k);
        }
        if (c.es && m == M - 1) {
            row(LSCQP_ROW_EQUALITY, fabs(at(k, m, 5) - at(k, m, 4)), -1, m, 4, k);
            row(LSCQP_ROW_EQUALITY, fabs(at(k, m, 5) - at(k, m, 3)), -1, m, 3, k);
        }
    }
    // wave reduction: per family max / count, and the argmax (ties: the lowest lane, i.e. deterministic)
#pragma unroll
    for (int f = 0; f < LSCQP_ROW_FAMILIES; f++) {
        for (int d = 32; d >= 1; d >>= 1) {
            fam_worst[f] = fmax(fam_worst[f], __shfl_xor(fam_worst[f], d));
            fam_cnt[f] += __shfl_xor(fam_cnt[f], d);
        }
    }
    double best = w.v;
    for (int d = 32; d >= 1; d >>= 1) best = fmax(best, __shfl_xor(best, d));
    const unsigned long long owners = __ballot(w.v == best);
    const int owner = owners ? (int)__builtin_ctzll(owners) : 0;
    if (lane == owner) {
        lscqp_diag& D = out[q];
        D.violation = (w.fam >= 0) ? w.v : 0.0;
        D.family = w.fam;
        D.obstacle = w.obs;
        D.segment = w.seg;
        D.point = w.pt;
        D.axis = w.axis;
        D.reserved = 0;
#pragma unroll
        for (int f = 0; f < LSCQP_ROW_FAMILIES; f++) {
            D.worst[f] = fam_worst[f] > -1e299 ? fam_worst[f] : 0.0;
            D.violated[f] = fam_cnt[f];
        }
    }
}

DiagClass diag_class(const lscqp_class_desc* d) {
    DiagClass c;
    c.M = d->M;
    c.dim = d->dim;
    c.es = d->planner_mode == LSCQP_PLANNER_LSC;
    c.use_sfc = d->use_sfc;
    c.rsfc = d->planner_mode == LSCQP_PLANNER_RSFC;
    c.rows_f32 = d->row_format == LSCQP_ROWS_F32;
    c.dt = d->dt;
    c.comm_range = d->communication_range;
    for (int k = 0; k < 3; k++) {
        c.world_min[k] = d->world_min[k];
        c.world_max[k] = d->world_max[k];
    }
    return c;
}

}  // namespace

extern "C" {

int lscqp_diagnose_device(lscqp_handle h, int64_t n, const lscqp_header* d_hdr, const lscqp_row* d_rows, const uint64_t* d_row_offsets,
                          const lscqp_box* d_sfc, const double* d_x, double tol, lscqp_diag* d_out, void* stream) {
    if (!h) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null handle");
    if (n < 0) return fail(LSCQP_ERR_INVALID_ARGUMENT, "negative size");
    if (n == 0) return LSCQP_OK;
    const lscqp_class_desc* d = lscqp_class_desc_of_(h);
    if (!d_hdr || !d_x || !d_out || (d->use_sfc && !d_sfc)) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null buffer");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(LSCQP_ERR_NO_DEVICE, "no HIP device: lscqp has no CPU fallback");
    hipLaunchKernelGGL(diagnose_kernel, dim3((unsigned)n), dim3(64), 0, (hipStream_t)stream, diag_class(d), n, d_hdr, d_rows, d_row_offsets, d_sfc,
                       d_x, tol, d_out);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(LSCQP_ERR_HIP, std::string("diagnose kernel: ") + hipGetErrorString(e));
    return LSCQP_OK;
}

int lscqp_diagnose(lscqp_handle h, int64_t n, const lscqp_header* hdr, const lscqp_row* rows, const uint64_t* row_offsets, const lscqp_box* sfc,
                   const double* x, double tol, lscqp_diag* out) {
    if (!h) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null handle");
    if (n < 0) return fail(LSCQP_ERR_INVALID_ARGUMENT, "negative size");
    if (n == 0) return LSCQP_OK;
    const lscqp_class_desc* d = lscqp_class_desc_of_(h);
    if (!hdr || !x || !out || (d->use_sfc && !sfc)) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null buffer");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(LSCQP_ERR_NO_DEVICE, "no HIP device: lscqp has no CPU fallback");
    int n_obs_max = 0;
    for (int64_t q = 0; q < n; q++) n_obs_max = hdr[q].n_obs > n_obs_max ? hdr[q].n_obs : n_obs_max;
    if (n_obs_max > 0 && (!rows || !row_offsets)) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null row buffer");
    const size_t rb = d->row_format == LSCQP_ROWS_F32 ? sizeof(lscqp_row_f32) : sizeof(lscqp_row);
    const size_t nv = (size_t)d->dim * d->M * 6, n_rows = n_obs_max > 0 ? (size_t)row_offsets[n] : 0;
    const size_t b_hdr = sizeof(lscqp_header) * n, b_rows = rb * n_rows, b_off = sizeof(uint64_t) * (n + 1), b_sfc = d->use_sfc ? sizeof(lscqp_box) * n * d->M : 0,
                 b_x = sizeof(double) * n * nv, b_out = sizeof(lscqp_diag) * n;
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    char* dev = nullptr;
    const size_t total = al(b_hdr) + al(b_rows) + al(b_off) + al(b_sfc) + al(b_x) + al(b_out);
    if (hipMalloc(&dev, total) != hipSuccess) return fail(LSCQP_ERR_HIP, "hipMalloc failed");
    char* p = dev;
    hipError_t up = hipSuccess;  // the first upload that failed
    auto put = [&](const void* src, size_t b) -> char* {
        char* at = p;
        if (b && up == hipSuccess) up = hipMemcpy(at, src, b, hipMemcpyHostToDevice);
        p += al(b);
        return at;
    };
    const lscqp_header* d_hdr = (const lscqp_header*)put(hdr, b_hdr);
    const lscqp_row* d_rows = (const lscqp_row*)put(rows, b_rows);
    std::vector<uint64_t> zero_off;
    if (!row_offsets) zero_off.assign(n + 1, 0);
    const uint64_t* d_off = (const uint64_t*)put(row_offsets ? row_offsets : zero_off.data(), b_off);
    const lscqp_box* d_sfc = (const lscqp_box*)put(sfc, b_sfc);
    const double* d_x = (const double*)put(x, b_x);
    lscqp_diag* d_out = (lscqp_diag*)p;
    if (up != hipSuccess) {
        (void)hipFree(dev);
        return fail(LSCQP_ERR_HIP, std::string("hipMemcpy (H2D) failed: ") + hipGetErrorString(up));
    }
    int rc = lscqp_diagnose_device(h, n, d_hdr, d_rows, d_off, b_sfc ? d_sfc : nullptr, d_x, tol, d_out, nullptr);
    if (rc == LSCQP_OK && hipMemcpy(out, d_out, b_out, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(LSCQP_ERR_HIP, "hipMemcpy (D2H) failed");
    (void)hipFree(dev);
    return rc;
}

const char* lscqp_row_family_name(int32_t family) {
    static const char* const names[LSCQP_ROW_FAMILIES] = {"variable bound", "SFC", "LSC", "velocity limit", "acceleration limit",
                                                           "communication range (segment pair)", "communication range (waypoint)", "equality"};
    return (family >= 0 && family < LSCQP_ROW_FAMILIES) ? names[family] : "none";
}

// CPLEX LP file of one instance, as cplex.exportModel writes the model populatebyrow built (src/traj_optimizer.cpp:45-52, 103):
// same variables (x_m_i, y_m_i, z_m_i), same rows in the same order, coefficients as the reference computes them.
int lscqp_dump_instance(lscqp_handle h, const lscqp_header* hdr, const lscqp_row* rows, const lscqp_box* sfc, const char* path) {
    if (!h || !hdr || !path) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null argument");
    const lscqp_class_desc* d = lscqp_class_desc_of_(h);
    if (d->use_sfc && !sfc) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null sfc buffer");
    if (hdr->n_obs > 0 && !rows) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null row buffer");
    FILE* f = fopen(path, "w");
    if (!f) return fail(LSCQP_ERR_INVALID_ARGUMENT, std::string("cannot open ") + path);
    const int M = d->M, dim = d->dim;
    const double dt = d->dt;
    static const double kQ[6][6] = {{720, -1800, 1200, 0, 0, -120},  {-1800, 4800, -3600, 0, 600, 0}, {1200, -3600, 3600, -1200, 0, 0},
                                    {0, 0, -1200, 3600, -3600, 1200}, {0, 600, 0, -3600, 4800, -1800}, {-120, 0, 0, 1200, -1800, 720}};
    const char ax[3] = {'x', 'y', 'z'};
    auto var = [&](int k, int m, int i) -> std::string {
        char b[32];
        snprintf(b, sizeof b, "%c_%d_%d", ax[k], m, i);
        return b;
    };
    int ts = hdr->terminal_segments;
    if (ts <= 0) {
        double d2 = 0;
        for (int k = 0; k < 3; k++) d2 += (hdr->goal[k] - hdr->p0[k]) * (hdr->goal[k] - hdr->p0[k]);
        ts = (int)((M * dt - std::sqrt(d2) / hdr->nominal_velocity + 1e-9) / dt);
        if (ts < 1) ts = 1;
    }
    if (ts > M) ts = M;
    fprintf(f, "\\ lscqp_dump_instance: the trajectory QP of one agent, row for row as TrajOptimizer::populatebyrow builds it\n");
    fprintf(f, "\\ (reference src/traj_optimizer.cpp:216-514); M=%d dim=%d dt=%.17g n_obs=%d terminal_segments=%d\n", M, dim, dt, hdr->n_obs, ts);
    fprintf(f, "Minimize\n obj:");
    // linear + constant part of the terminal cost w_t (x - g)^2 (:301-316)
    double cst = 0;
    for (int m = M - ts; m < M; m++)
        for (int k = 0; k < dim; k++) {
            fprintf(f, " %+.17g %s", -2.0 * d->terminal_weight * hdr->goal[k], var(k, m, 5).c_str());
            cst += d->terminal_weight * hdr->goal[k] * hdr->goal[k];
        }
    fprintf(f, " %+.17g\n  + [", cst);
    const double sc = std::pow(dt, -5.0);
    for (int k = 0; k < dim; k++)
        for (int m = 0; m < M; m++) {
            for (int i = 0; i < 6; i++)
                for (int j = i; j < 6; j++) {
                    if (kQ[i][j] == 0 || d->control_input_weight == 0) continue;
                    // cost += w_c Q(i,j) x_i x_j over all (i,j) (:285-299): the pair (i,j), i != j, appears twice; inside [ ]/2 once more doubled
                    const double coef = d->control_input_weight * (kQ[i][j] * sc) * (i == j ? 2.0 : 4.0);
                    if (i == j)
                        fprintf(f, " %+.17g %s ^2", coef, var(k, m, i).c_str());
                    else
                        fprintf(f, " %+.17g %s * %s", coef, var(k, m, i).c_str(), var(k, m, j).c_str());
                }
            if (m >= M - ts) fprintf(f, " %+.17g %s ^2", 2.0 * d->terminal_weight, var(k, m, 5).c_str());
            fprintf(f, "\n   ");
        }
    fprintf(f, "] / 2\nSubject To\n");
    int ci = 0;
    const double sv = 5.0 / dt, sa = 20.0 / (dt * dt);
    for (int k = 0; k < dim; k++) {  // :318-353
        fprintf(f, " c%d: %s = %.17g\n", ++ci, var(k, 0, 0).c_str(), hdr->p0[k]);
        fprintf(f, " c%d: %s - %s = 0\n", ++ci, var(k, 0, 5).c_str(), var(k, 1, 0).c_str());
        fprintf(f, " c%d: %.17g %s - %.17g %s = %.17g\n", ++ci, sv, var(k, 0, 1).c_str(), sv, var(k, 0, 0).c_str(), hdr->v0[k]);
        fprintf(f, " c%d: %.17g %s - %.17g %s + %.17g %s = %.17g\n", ++ci, sa, var(k, 0, 2).c_str(), 2 * sa, var(k, 0, 1).c_str(), sa, var(k, 0, 0).c_str(),
                hdr->a0[k]);
        fprintf(f, " c%d: %s - %s - %s + %s = 0\n", ++ci, var(k, 1, 1).c_str(), var(k, 1, 0).c_str(), var(k, 0, 5).c_str(), var(k, 0, 4).c_str());
        fprintf(f, " c%d: %s - 2 %s + %s - %s + 2 %s - %s = 0\n", ++ci, var(k, 1, 2).c_str(), var(k, 1, 1).c_str(), var(k, 1, 0).c_str(),
                var(k, 0, 5).c_str(), var(k, 0, 4).c_str(), var(k, 0, 3).c_str());
    }
    for (int k = 0; k < dim; k++)  // Aeq_base (:180-214, 356-368): joins (m-1) -> m for m = 2 .. M-1, rows dt^-j nn_j (A_T - A_0)
        for (int m = 2; m < M; m++) {
            fprintf(f, " c%d: %s - %s = 0\n", ++ci, var(k, m - 1, 5).c_str(), var(k, m, 0).c_str());
            fprintf(f, " c%d: %.17g %s - %.17g %s - %.17g %s + %.17g %s = 0\n", ++ci, sv, var(k, m - 1, 5).c_str(), sv, var(k, m - 1, 4).c_str(), sv,
                    var(k, m, 1).c_str(), sv, var(k, m, 0).c_str());
            fprintf(f, " c%d: %.17g %s - %.17g %s + %.17g %s - %.17g %s + %.17g %s - %.17g %s = 0\n", ++ci, sa, var(k, m - 1, 5).c_str(), 2 * sa,
                    var(k, m - 1, 4).c_str(), sa, var(k, m - 1, 3).c_str(), sa, var(k, m, 2).c_str(), 2 * sa, var(k, m, 1).c_str(), sa, var(k, m, 0).c_str());
        }
    if (d->use_sfc)  // :370-397, faces in Box::convertToLSCs order (src/collision_constraints.cpp:37-59): per axis +e_k at box_min, then -e_k at box_max
        for (int m = 0; m < M; m++)
            for (int k = 0; k < dim; k++)
                for (int side = 0; side < 2; side++)
                    for (int j = 0; j < 6; j++) {
                        if (m == 0 && j < 3) continue;
                        if (side == 0)
                            fprintf(f, " c%d: %s >= %.17g\n", ++ci, var(k, m, j).c_str(), sfc[m].bmin[k]);
                        else
                            fprintf(f, " c%d: - %s >= %.17g\n", ++ci, var(k, m, j).c_str(), -sfc[m].bmax[k]);
                    }
    for (int oi = 0; oi < hdr->n_obs; oi++)  // :399-437
        for (int m = 0; m < M; m++)
            for (int i = 0; i < 6; i++) {
                if (m == 0 && i < 3) continue;
                double nx, ny, nz, b;
                const size_t e = (size_t)oi * 6 * M + 6 * m + i;
                if (d->row_format == LSCQP_ROWS_F32) {
                    const lscqp_row_f32& r = reinterpret_cast<const lscqp_row_f32*>(rows)[e];
                    nx = r.nx, ny = r.ny, nz = r.nz, b = r.b;
                } else {
                    nx = rows[e].nx, ny = rows[e].ny, nz = rows[e].nz, b = rows[e].b;
                }
                if (std::sqrt(nx * nx + ny * ny + nz * nz) < 1e-5) continue;
                fprintf(f, " c%d: %+.17g %s %+.17g %s", ++ci, nx, var(0, m, i).c_str(), ny, var(1, m, i).c_str());
                if (dim == 3) fprintf(f, " %+.17g %s", nz, var(2, m, i).c_str());
                fprintf(f, " >= %.17g\n", b);
            }
    for (int k = 0; k < dim; k++)  // :439-474
        for (int m = 0; m < M; m++) {
            for (int i = 0; i < 5; i++) {
                if (m == 0 && i < 2) continue;
                fprintf(f, " c%d: %.17g %s - %.17g %s <= %.17g\n", ++ci, sv, var(k, m, i + 1).c_str(), sv, var(k, m, i).c_str(), hdr->vmax[k]);
                fprintf(f, " c%d: - %.17g %s + %.17g %s <= %.17g\n", ++ci, sv, var(k, m, i + 1).c_str(), sv, var(k, m, i).c_str(), hdr->vmax[k]);
            }
            for (int i = 0; i < 4; i++) {
                if (m == 0 && i < 1) continue;
                fprintf(f, " c%d: %.17g %s - %.17g %s + %.17g %s <= %.17g\n", ++ci, sa, var(k, m, i + 2).c_str(), 2 * sa, var(k, m, i + 1).c_str(), sa,
                        var(k, m, i).c_str(), hdr->amax[k]);
                fprintf(f, " c%d: - %.17g %s + %.17g %s - %.17g %s <= %.17g\n", ++ci, sa, var(k, m, i + 2).c_str(), 2 * sa, var(k, m, i + 1).c_str(), sa,
                        var(k, m, i).c_str(), hdr->amax[k]);
            }
        }
    if (d->communication_range > 0) {  // :476-500
        const double rho = 0.5 * d->communication_range - hdr->radius, rho_w = 0.5 * d->communication_range - 1e-5;
        for (int k = 0; k < dim; k++)
            for (int mi = 0; mi < M; mi++)
                for (int m = mi; m < M; m++) {
                    fprintf(f, " c%d: %s - %s <= %.17g\n", ++ci, var(k, m, 5).c_str(), var(k, mi, 0).c_str(), rho);
                    fprintf(f, " c%d: - %s + %s <= %.17g\n", ++ci, var(k, m, 5).c_str(), var(k, mi, 0).c_str(), rho);
                }
        for (int k = 0; k < dim; k++)
            for (int m = 0; m < M; m++) {
                fprintf(f, " c%d: %s <= %.17g\n", ++ci, var(k, m, 5).c_str(), rho_w + hdr->next_waypoint[k]);
                fprintf(f, " c%d: - %s <= %.17g\n", ++ci, var(k, m, 5).c_str(), rho_w - hdr->next_waypoint[k]);
            }
    }
    if (d->planner_mode == LSCQP_PLANNER_LSC)  // :502-511
        for (int k = 0; k < dim; k++)
            for (int i = 1; i < 3; i++) fprintf(f, " c%d: %s - %s = 0\n", ++ci, var(k, M - 1, 5).c_str(), var(k, M - 1, 5 - i).c_str());
    fprintf(f, "Bounds\n");  // :238-270
    for (int k = 0; k < dim; k++)
        for (int m = 0; m < M; m++)
            for (int i = 0; i < 6; i++) {
                if (m == 0 && i < 3) {
                    fprintf(f, " %s free\n", var(k, m, i).c_str());
                    continue;
                }
                double lo = d->world_min[k], hi = d->world_max[k];
                if (k == 2 && m == 0 && d->planner_mode == LSCQP_PLANNER_RSFC) lo = -100, hi = 100;
                fprintf(f, " %.17g <= %s <= %.17g\n", lo, var(k, m, i).c_str(), hi);
            }
    fprintf(f, "End\n");
    const bool ok = fclose(f) == 0;
    return ok ? LSCQP_OK : fail(LSCQP_ERR_INVALID_ARGUMENT, std::string("write failed: ") + path);
}

}  // extern "C"

// ---- counter calibration (library-internal; tools/profile_round.py, bench.py --calibrate-counters) ---------------------------------------
// rocprofv3's FETCH_SIZE / WRITE_SIZE are derived from the L2's memory-side request counters and, on gfx950, tally wide coalesced reads at
// a fraction of their bytes (the guide: exactly 1/2 for 16 B/lane; "other access widths and WRITE_SIZE are uncalibrated: calibrate on a
// known byte count in your own access pattern").  These kernels ARE that known byte count, in the access patterns of the solver's row
// stream -- every lane reads LB contiguous bytes, a wavefront 64 x LB contiguous bytes, exactly as das_kernel fetches lscqp_row (LB = 32, two
// dwordx4 per lane) and lscqp_row_f32 (LB = 16) -- over a buffer far larger than the 256 MiB Infinity Cache, launched in the SAME profiled
// process as the workload so that the same counter session measures both.
namespace lscqp_calib {
template <int LB>
__global__ __launch_bounds__(256) void read_kernel(const char* __restrict__ buf, int64_t n_units, unsigned* __restrict__ sink) {
    unsigned acc = 0;
    for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < n_units; u += (int64_t)gridDim.x * blockDim.x) {
        const char* p = buf + u * LB;
        if constexpr (LB == 8) {
            const uint2 v = *reinterpret_cast<const uint2*>(p);
            acc ^= v.x ^ v.y;
        } else {
#pragma unroll
            for (int j = 0; j < LB / 16; j++) {
                const uint4 v = *reinterpret_cast<const uint4*>(p + 16 * j);
                acc ^= v.x ^ v.y ^ v.z ^ v.w;
            }
        }
    }
    if (acc == 0x9e3779b9u) sink[0] = acc;  // (keeps the loads alive; the buffer holds zeros)
}
template <int LB>
__global__ __launch_bounds__(256) void write_kernel(char* __restrict__ buf, int64_t n_units) {
    for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < n_units; u += (int64_t)gridDim.x * blockDim.x) {
        char* p = buf + u * LB;
#pragma unroll
        for (int j = 0; j < LB / 8; j++) *reinterpret_cast<uint2*>(p + 8 * j) = make_uint2((unsigned)u, (unsigned)j);
    }
}
}  // namespace lscqp_calib

// bytes: size of the streamed buffer (>= 512 MiB to defeat the Infinity Cache).  Launches, in this order and each over the WHOLE buffer:
// read_kernel<8>, <16>, <32>, write_kernel<8>, <16> -- every launch's byte count is exactly `bytes` (rounded down to a multiple of 32).
extern "C" int lscqp_debug_calibrate_(int64_t bytes, void* stream) {
    if (bytes < 4096) return fail(LSCQP_ERR_INVALID_ARGUMENT, "calibration buffer too small");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(LSCQP_ERR_NO_DEVICE, "no HIP device: lscqp has no CPU fallback");
    bytes = bytes / 32 * 32;
    char* buf = nullptr;
    unsigned* sink = nullptr;
    if (hipMalloc(&buf, (size_t)bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) {
        if (buf) (void)hipFree(buf);
        return fail(LSCQP_ERR_HIP, "calibration: hipMalloc failed");
    }
    hipStream_t st = (hipStream_t)stream;
    (void)hipMemsetAsync(buf, 0, (size_t)bytes, st);
    (void)hipMemsetAsync(sink, 0, 64, st);
    const dim3 grid(256 * 16), block(256);
    hipLaunchKernelGGL(lscqp_calib::read_kernel<8>, grid, block, 0, st, buf, bytes / 8, sink);
    hipLaunchKernelGGL(lscqp_calib::read_kernel<16>, grid, block, 0, st, buf, bytes / 16, sink);
    hipLaunchKernelGGL(lscqp_calib::read_kernel<32>, grid, block, 0, st, buf, bytes / 32, sink);
    hipLaunchKernelGGL(lscqp_calib::write_kernel<8>, grid, block, 0, st, buf, bytes / 8);
    hipLaunchKernelGGL(lscqp_calib::write_kernel<16>, grid, block, 0, st, buf, bytes / 16);
    const hipError_t e = hipStreamSynchronize(st);
    (void)hipFree(buf);
    (void)hipFree(sink);
    if (e != hipSuccess) return fail(LSCQP_ERR_HIP, std::string("calibration kernels: ") + hipGetErrorString(e));
    return LSCQP_OK;
}
