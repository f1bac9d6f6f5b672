// lscqp_kernel.hpp — batched primal-dual interior point for the LSC trajectory QP, gfx950 (CDNA4) only.
//
// Replaces the arithmetic of TrajOptimizer::solve (reference src/traj_optimizer.cpp:18-156: CPLEX) for the
// model TrajOptimizer::populatebyrow builds (src/traj_optimizer.cpp:216-514).  Not a translation of either:
//
//   * ONE WAVEFRONT (64 lanes) PER QP, one workgroup per wavefront.
//   * The equality rows (src/traj_optimizer.cpp:318-368, 502-511) are eliminated analytically: per axis the free
//     variables are z = (c3,c4,c5) of every segment (one scalar for the last segment under the LSC end stop);
//     (c0,c1,c2) of segment m+1 = TB (c3,c4,c5) of segment m, TB = [[0,0,1],[0,-1,2],[1,-4,4]].
//     nz = dim*(3M-2) <= 64, so lane r owns row r of the reduced KKT matrix.
//   * Inequalities are never formed as a matrix.  LSC rows (n.c >= b, one control point each) live in LDS as
//     SoA [obstacle][control point]; the per-axis structured rows (merged interval bounds from world box / SFC /
//     communication range, velocity and acceleration differences, communication pairs) are two-sided rows held in
//     registers, a few per lane.
//   * Each Mehrotra iteration: rows -> per-control-point 3x3 blocks S and x-space vectors in LDS -> every lane builds
//     its row of the reduced matrix from per-(axis,segment) 6x6 local blocks -> LDL^T entirely in registers,
//     pivot rows broadcast with v_readlane -> two solves -> step.
//   * All arithmetic fp64, in coordinates translated to the agent's position.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lscqp.h"

namespace lscqp {

struct DevClass {
    double dt, w_c, w_t, comm_range;
    double world_min[3], world_max[3];
    double Q2[36];  // 2 * w_c * Q_base (closed form of src/traj_optimizer.cpp:163-178)
    double dQ[36];  // Q_base as the reference rounds it (integer matrix * pow(dt,-5), per entry, in double)
                    // minus the exact product: restores the reference model's tiny non-translation-invariance
                    // in the reported objective
    double tol;
    int max_iter;
    int use_sfc;
    int n_obs_max;  // LDS is sized for this many obstacles per instance
    int pad;
};

__device__ __forceinline__ double bcast(double v, int lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double fast_rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;
}

template <int M_, int DIM_, bool ES_>
struct Cfg {
    static constexpr int M = M_, DIM = DIM_;
    static constexpr bool ES = ES_;
    static constexpr int P = 6 * M;
    static constexpr int NZA = 3 * (M - 1) + (ES ? 1 : 3);
    static constexpr int NZ = DIM * NZA;
    static constexpr int NX = DIM * P;  // x-space size
    // two-sided per-axis rows, uniform index space (rows that do not exist for m==0 stay infinite):
    //   [0,P) interval of cp | [P,P+5M) vel (m,i) | [..,+4M) acc (m,i) | comm pairs (u,up<u)
    static constexpr int OV = P, OA = P + 5 * M, OC = P + 9 * M, NRA = P + 9 * M + M * (M - 1) / 2;
    static constexpr int NR2 = DIM * NRA;
    static constexpr int RPL = (NR2 + 63) / 64;
    static constexpr int G = (64 / P) > 0 ? (64 / P) : 1;  // lane groups in the LSC pass
    static constexpr int LDH = NZ | 1;
    static_assert(NZ <= 64, "lane-per-row kernel needs dim*(3M-2) <= 64");
    static_assert(P <= 64, "M <= 10");
    static_assert(M >= 2, "the reference assumes M >= 2 (src/traj_optimizer.cpp:341-352)");
    // LDS carve (in doubles)
    static constexpr int o_c = 0;               // control points (translated)
    static constexpr int o_dca = o_c + NX;      // affine direction, x-space
    static constexpr int o_dc = o_dca + NX;     // final direction, x-space
    static constexpr int o_x0 = o_dc + NX;      // x-space accumulators (4)
    static constexpr int o_S = o_x0 + 4 * NX;   // per-cp 3x3 sym blocks [P][6]
    static constexpr int o_om = o_S + 6 * P;    // two-sided row weights [DIM][NRA]
    static constexpr int o_z = o_om + NR2;      // z, dz
    static constexpr int o_H = o_z + 2 * 64;    // per-lane scratch rows of the reduced matrix [NZ][LDH]
    static constexpr int o_rows = ((o_H + NZ * LDH + 1) / 2) * 2;  // LSC SoA: nx,ny,nz,b,s,lam each [n_obs_max*P]
    static size_t lds_bytes(int n_obs_max) { return sizeof(double) * ((size_t)o_rows + 6 * (size_t)n_obs_max * P); }
};

// TB rows: (c0,c1,c2) of the next segment in terms of (c3,c4,c5) of this one.
#define LSCQP_TB(i, j) ((i) == 0 ? ((j) == 2 ? 1.0 : 0.0) : (i) == 1 ? ((j) == 0 ? 0.0 : (j) == 1 ? -1.0 : 2.0) : ((j) == 0 ? 1.0 : (j) == 1 ? -4.0 : 4.0))

template <int M, int DIM, bool ES>
__global__ __launch_bounds__(64) void lscqp_pdip_kernel(DevClass cls, int64_t n, const lscqp_header* __restrict__ hdr,
                                                        const lscqp_row* __restrict__ rows,
                                                        const uint64_t* __restrict__ row_offsets,
                                                        const lscqp_box* __restrict__ sfc, double* __restrict__ x_out,
                                                        double* __restrict__ obj_out, int32_t* __restrict__ status_out,
                                                        lscqp_info* __restrict__ info_out) {
    using C = Cfg<M, DIM, ES>;
    constexpr int P = C::P, NZA = C::NZA, NZ = C::NZ, NX = C::NX, NRA = C::NRA, NR2 = C::NR2, RPL = C::RPL, G = C::G,
                  LDH = C::LDH;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* const c_ = smem + C::o_c;
    double* const dca_ = smem + C::o_dca;
    double* const dc_ = smem + C::o_dc;
    double* const XL = smem + C::o_x0;           // G' lambda          (x-space)
    double* const XA = smem + C::o_x0 + NX;      // G' q_aff  /  later G' q_corr
    double* const XB1 = smem + C::o_x0 + 2 * NX; // G' (1/s)
    double* const XB2 = smem + C::o_x0 + 3 * NX; // G' (-ds_a dl_a/s - w rp)
    double* const S_ = smem + C::o_S;
    double* const om_ = smem + C::o_om;
    double* const z_ = smem + C::o_z;
    double* const dz_ = smem + C::o_z + 64;
    double* const Hs = smem + C::o_H;
    const int nrow_max = cls.n_obs_max * P;
    double* const Rnx = smem + C::o_rows;
    double* const Rny = Rnx + nrow_max;
    double* const Rnz = Rny + nrow_max;
    double* const Rb = Rnz + nrow_max;
    double* const Rs = Rb + nrow_max;
    double* const Rl = Rs + nrow_max;

    const int lane = threadIdx.x;
    const int64_t q = blockIdx.x;
    if (q >= n) return;
    const lscqp_header* H = hdr + q;
    const int n_obs = H->n_obs < cls.n_obs_max ? H->n_obs : cls.n_obs_max;
    const int nrow = n_obs * P;
    const double dt = cls.dt;

    // ---- per-QP scalars (uniform) ------------------------------------------------------------------------
    double org[3], goal[3], wp[3], vlim[3], alim[3], cf1[3], cf2[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        org[k] = H->p0[k];
        goal[k] = H->goal[k] - org[k];
        wp[k] = H->next_waypoint[k] - org[k];
        vlim[k] = H->vmax[k] * dt * 0.2;            // |c_{i+1}-c_i| <= vmax dt/n      (src/traj_optimizer.cpp:448-453)
        alim[k] = H->amax[k] * dt * dt * 0.05;      // |c_{i+2}-2c_{i+1}+c_i| <= amax dt^2/(n(n-1))   (:462-471)
        cf1[k] = H->v0[k] * dt * 0.2;               // c1 - c0                         (:330-332)
        cf2[k] = H->a0[k] * dt * dt * 0.05 + 2.0 * cf1[k];  // c2 - c0                (:335-338)
    }
    int ts = H->terminal_segments;
    if (ts <= 0) {  // src/traj_optimizer.cpp:530-538 in fp64
        double d2 = 0;
#pragma unroll
        for (int k = 0; k < 3; k++) d2 += goal[k] * goal[k];
        ts = (int)((M * dt - sqrt(d2) / H->nominal_velocity + 1e-9) / dt);
        if (ts < 1) ts = 1;
    }
    if (ts > M) ts = M;
    const double rho_pair = 0.5 * cls.comm_range - H->radius;  // :484
    const double rho_wp = 0.5 * cls.comm_range - 1e-5;         // :495
    const bool comm_on = cls.comm_range > 0;

    // ---- lane roles ---------------------------------------------------------------------------------------
    // z lane: row r = lane of the reduced system, r = k*NZA + a
    const bool zl = lane < NZ;
    const int zk = zl ? lane / NZA : 0;
    const int za = zl ? lane % NZA : 0;
    const bool zlast = ES && (za == 3 * (M - 1));
    const int zm = zlast ? (M - 1) : za / 3;
    const int zj = zlast ? 0 : za % 3;
    // e[j']: which of (c3,c4,c5) of segment zm this variable drives
    const double e0 = (zlast || zj == 0) ? 1.0 : 0.0, e1 = (zlast || zj == 1) ? 1.0 : 0.0, e2 = (zlast || zj == 2) ? 1.0 : 0.0;
    // tb[i] = TB[i][zj] (zero for the end-stop variable: no next segment)
    const bool has_next = (zm + 1 < M);
    const double tb0 = has_next ? (zj == 2 ? 1.0 : 0.0) : 0.0;
    const double tb1 = has_next ? (zj == 0 ? 0.0 : zj == 1 ? -1.0 : 2.0) : 0.0;
    const double tb2 = has_next ? (zj == 0 ? 1.0 : zj == 1 ? -4.0 : 4.0) : 0.0;
    auto zidx = [](int m, int j) -> int { return (ES && m == M - 1) ? 3 * (M - 1) : 3 * m + j; };

    // ---- two-sided rows owned by this lane (registers) ----------------------------------------------------
    int r_ia[RPL], r_ib[RPL], r_ic[RPL];  // LDS indices into c_ (x-space), -1 = unused
    double r_cb[RPL];                     // middle coefficient (-2 for acc, +1/-1 otherwise handled by type)
    int r_ty[RPL];                        // 0 interval, 1 vel, 2 acc, 3 comm, -1 none
    double r_lo[RPL], r_hi[RPL], r_slo[RPL], r_shi[RPL], r_llo[RPL], r_lhi[RPL];
#pragma unroll
    for (int u = 0; u < RPL; u++) {
        const int t2 = lane + 64 * u;
        r_ty[u] = -1;
        r_ia[u] = r_ib[u] = r_ic[u] = 0;
        r_cb[u] = 0;
        r_lo[u] = -INFINITY;
        r_hi[u] = INFINITY;
        if (t2 < NR2) {
            const int k = t2 / NRA, t = t2 % NRA;
            const int base = k * P;
            if (t < C::OV) {  // interval on cp t
                const int m = t / 6, i = t % 6;
                if (!(m == 0 && i < 3)) {
                    r_ty[u] = 0;
                    r_ia[u] = base + t;
                    double lo = cls.world_min[k] - org[k], hi = cls.world_max[k] - org[k];  // :252-253,260-265
                    if (cls.use_sfc) {                                                      // :372-397
                        lo = fmax(lo, sfc[q * M + m].bmin[k] - org[k]);
                        hi = fmin(hi, sfc[q * M + m].bmax[k] - org[k]);
                    }
                    if (comm_on && i == 5) {  // pairs (m, mi=0) :482-487 and waypoint rows :494-497
                        lo = fmax(lo, fmax(-rho_pair, wp[k] - rho_wp));
                        hi = fmin(hi, fmin(rho_pair, wp[k] + rho_wp));
                    }
                    r_lo[u] = lo;
                    r_hi[u] = hi;
                }
            } else if (t < C::OA) {  // velocity (m,i): c[i+1]-c[i]
                const int v = t - C::OV, m = v / 5, i = v % 5;
                if (!(m == 0 && i < 2)) {
                    r_ty[u] = 1;
                    r_ia[u] = base + 6 * m + i;
                    r_ib[u] = base + 6 * m + i + 1;
                    r_lo[u] = -vlim[k];
                    r_hi[u] = vlim[k];
                }
            } else if (t < C::OC) {  // acceleration (m,i): c[i+2]-2c[i+1]+c[i]
                const int a = t - C::OA, m = a / 4, i = a % 4;
                if (!(m == 0 && i < 1)) {
                    r_ty[u] = 2;
                    r_ia[u] = base + 6 * m + i;
                    r_ib[u] = base + 6 * m + i + 1;
                    r_ic[u] = base + 6 * m + i + 2;
                    r_lo[u] = -alim[k];
                    r_hi[u] = alim[k];
                }
            } else if (comm_on) {  // pair (uu, up<uu): c[uu][5] - c[up+1][0]     (:482-487 with mi = up+1 >= 1)
                const int cidx = t - C::OC;
                int uu = 1;
                while (uu * (uu + 1) / 2 <= cidx) uu++;
                const int up = cidx - uu * (uu - 1) / 2;
                r_ty[u] = 3;
                r_ia[u] = base + 6 * (up + 1) + 0;
                r_ib[u] = base + 6 * uu + 5;
                r_lo[u] = -rho_pair;
                r_hi[u] = rho_pair;
            }
        }
    }
    // row value y = g.c  for the rows of this lane
    auto row_val = [&](const double* v, int u) -> double {
        const int ty = r_ty[u];
        double y = 0;
        if (ty == 0) y = v[r_ia[u]];
        else if (ty == 1 || ty == 3) y = v[r_ib[u]] - v[r_ia[u]];
        else if (ty == 2) y = v[r_ic[u]] - 2.0 * v[r_ib[u]] + v[r_ia[u]];
        return y;
    };
    // scatter val * g into an x-space accumulator (LDS atomics; few per lane)
    auto row_scatter = [&](double* X, int u, double val) {
        const int ty = r_ty[u];
        if (ty == 0) atomicAdd(&X[r_ia[u]], val);
        else if (ty == 1 || ty == 3) {
            atomicAdd(&X[r_ib[u]], val);
            atomicAdd(&X[r_ia[u]], -val);
        } else if (ty == 2) {
            atomicAdd(&X[r_ic[u]], val);
            atomicAdd(&X[r_ib[u]], -2.0 * val);
            atomicAdd(&X[r_ia[u]], val);
        }
    };

    // ---- stage LSC rows: HBM (AoS 32 B, [oi][m][i]) -> LDS SoA, translated to the agent's origin -----------
    {
        const lscqp_row* R = rows + row_offsets[q];
        for (int e = lane; e < nrow; e += 64) {
            const double4 v = *reinterpret_cast<const double4*>(&R[e]);
            const int cp = e % P;
            double nx = v.x, ny = v.y, nz = (DIM == 3) ? v.z : 0.0;
            double b = v.w - (v.x * org[0] + v.y * org[1] + (DIM == 3 ? v.z * org[2] : 0.0));
            // rows of the initial state (:404-406) and rows with ||normal|| < 1e-5 (:409-411) are dropped
            const bool dead = (cp < 3) || (sqrt(v.x * v.x + v.y * v.y + v.z * v.z) < 1e-5);
            if (dead) {
                nx = ny = nz = 0.0;
                b = -1.0;
            }
            Rnx[e] = nx;
            Rny[e] = ny;
            Rnz[e] = nz;
            Rb[e] = b;
        }
    }
    // ---- initial point: every free control point at c2 of the first segment -------------------------------
    for (int e = lane; e < NX; e += 64) {
        const int k = e / P, cp = e % P;
        c_[e] = (cp == 0) ? 0.0 : (cp == 1) ? cf1[k] : cf2[k];
    }
    if (zl) z_[lane] = cf2[zk];
    __syncthreads();

    int status = LSCQP_STATUS_ITER_LIMIT;
    double m_tot = 0;
    {
        double cnt = 0;
        bool bad = false;
#pragma unroll
        for (int u = 0; u < RPL; u++) {
            r_slo[u] = r_shi[u] = 1.0;
            r_llo[u] = r_lhi[u] = 0.0;
            if (r_ty[u] >= 0) {
                const double y = row_val(c_, u);
                if (r_lo[u] > r_hi[u]) bad = true;
                if (r_lo[u] > -INFINITY) {
                    r_slo[u] = fmax(y - r_lo[u], 1e-2);
                    r_llo[u] = 1.0;
                    cnt += 1;
                }
                if (r_hi[u] < INFINITY) {
                    r_shi[u] = fmax(r_hi[u] - y, 1e-2);
                    r_lhi[u] = 1.0;
                    cnt += 1;
                }
            }
        }
        for (int e = lane; e < nrow; e += 64) {
            const int cp = e % P;
            const double nx = Rnx[e], ny = Rny[e], nz = Rnz[e];
            const bool act = (nx != 0.0) || (ny != 0.0) || (nz != 0.0);
            double r = nx * c_[cp] + ny * c_[P + cp] - Rb[e];
            if (DIM == 3) r += nz * c_[2 * P + cp];
            Rs[e] = act ? fmax(r, 1e-2) : 1.0;
            Rl[e] = act ? 1.0 : 0.0;
            cnt += act ? 1.0 : 0.0;
        }
        m_tot = wave_sum(cnt);
        if (__any(bad)) status = LSCQP_STATUS_INFEASIBLE;  // empty interval: lo > hi
    }
    __syncthreads();

    // objective exactly as cplex.getObjValue() reports it (src/traj_optimizer.cpp:100): jerk cost
    //   x'(w_c Q)x == w_c * 3600 dt^-5 * sum_seg (D3 c)' MB (D3 c)   (third differences: no cancellation)
    // plus the terminal cost including its constant goal^2 term (:301-316).  Translation invariant.
    auto objective = [&](bool ref_rounding) -> double {
        double part = 0;
        if (lane < DIM * M) {
            const int k = lane / M, m = lane % M;
            const double* cc = &c_[k * P + 6 * m];
            const double j0 = (cc[3] - cc[0]) - 3.0 * (cc[2] - cc[1]);
            const double j1 = (cc[4] - cc[1]) - 3.0 * (cc[3] - cc[2]);
            const double j2 = (cc[5] - cc[2]) - 3.0 * (cc[4] - cc[3]);
            const double quad =
                0.2 * (j0 * j0 + j2 * j2) + (2.0 / 15.0) * j1 * j1 + 0.2 * (j0 * j1 + j1 * j2) + (1.0 / 15.0) * j0 * j2;
            const double dt2 = dt * dt;
            part = cls.w_c * 3600.0 / (dt2 * dt2 * dt) * quad;
            if (ref_rounding) {  // world-frame correction, O(1e-9): negligible rounding of its own
                double corr = 0;
#pragma unroll
                for (int i = 0; i < 6; i++) {
                    double r = 0;
#pragma unroll
                    for (int ip = 0; ip < 6; ip++) r += cls.dQ[i * 6 + ip] * (cc[ip] + org[k]);
                    corr += r * (cc[i] + org[k]);
                }
                part += cls.w_c * corr;
            }
            if (m >= M - ts) {
                const double dgoal = cc[5] - goal[k];
                part += cls.w_t * dgoal * dgoal;
            }
        }
        return wave_sum(part);
    };

    double A[NZ];  // row `lane` of the reduced KKT matrix, then its LDL^T factors
    double dinv_own = 0.0;
    double res_p = 0, res_d = 0, res_gap = 0;
    int it = 0;
    const double tol = cls.tol;

    if (status != LSCQP_STATUS_INFEASIBLE)
        for (it = 0; it < cls.max_iter; it++) {
            // ============ pass 1: residuals, weights, per-cp blocks =========================================
            for (int e = lane; e < 4 * NX + 6 * P; e += 64) smem[C::o_x0 + e] = 0.0;  // XL,XA,XB1,XB2,S
            __syncthreads();
            double sum_sl = 0, sum_pinf = 0, max_rp = 0;
#pragma unroll
            for (int u = 0; u < RPL; u++) {
                if (r_ty[u] >= 0) {
                    const double y = row_val(c_, u);
                    const double rplo = (y - r_lo[u]) - r_slo[u], rphi = (r_hi[u] - y) - r_shi[u];
                    const bool flo = r_lo[u] > -INFINITY, fhi = r_hi[u] < INFINITY;
                    const double wlo = flo ? r_llo[u] / r_slo[u] : 0.0, whi = fhi ? r_lhi[u] / r_shi[u] : 0.0;
                    if (flo) {
                        sum_sl += r_slo[u] * r_llo[u];
                        sum_pinf += r_llo[u] * fabs(rplo);
                        max_rp = fmax(max_rp, fabs(rplo));
                    }
                    if (fhi) {
                        sum_sl += r_shi[u] * r_lhi[u];
                        sum_pinf += r_lhi[u] * fabs(rphi);
                        max_rp = fmax(max_rp, fabs(rphi));
                    }
                    om_[lane + 64 * u] = wlo + whi;
                    row_scatter(XL, u, r_llo[u] - r_lhi[u]);
                    row_scatter(XA, u, -(flo ? wlo * rplo : 0.0) + (fhi ? whi * rphi : 0.0));
                } else if (lane + 64 * u < NR2) {
                    om_[lane + 64 * u] = 0.0;
                }
            }
            if (lane < G * P) {  // LSC rows: lane = g*P + cp, obstacles o = g, g+G, ...
                const int g = lane / P, cp = lane % P;
                double s00 = 0, s01 = 0, s02 = 0, s11 = 0, s12 = 0, s22 = 0, l0 = 0, l1 = 0, l2 = 0, a0 = 0, a1 = 0, a2 = 0;
                const double cx = c_[cp], cy = c_[P + cp], cz = (DIM == 3) ? c_[2 * P + cp] : 0.0;
                for (int o = g; o < n_obs; o += G) {
                    const int e = o * P + cp;
                    const double nx = Rnx[e], ny = Rny[e], nz = Rnz[e], s = Rs[e], lam = Rl[e];
                    const double rp = (nx * cx + ny * cy + nz * cz - Rb[e]) - s;
                    const double w = lam / s;
                    sum_sl += s * lam;
                    sum_pinf += lam * fabs(rp);
                    max_rp = fmax(max_rp, lam > 0.0 ? fabs(rp) : 0.0);
                    s00 += w * nx * nx; s01 += w * nx * ny; s02 += w * nx * nz;
                    s11 += w * ny * ny; s12 += w * ny * nz; s22 += w * nz * nz;
                    l0 += lam * nx; l1 += lam * ny; l2 += lam * nz;
                    const double qa = -w * rp;
                    a0 += qa * nx; a1 += qa * ny; a2 += qa * nz;
                }
                atomicAdd(&S_[cp * 6 + 0], s00); atomicAdd(&S_[cp * 6 + 1], s01); atomicAdd(&S_[cp * 6 + 3], s11);
                atomicAdd(&XL[cp], l0); atomicAdd(&XL[P + cp], l1);
                atomicAdd(&XA[cp], a0); atomicAdd(&XA[P + cp], a1);
                if (DIM == 3) {
                    atomicAdd(&S_[cp * 6 + 2], s02); atomicAdd(&S_[cp * 6 + 4], s12); atomicAdd(&S_[cp * 6 + 5], s22);
                    atomicAdd(&XL[2 * P + cp], l2);
                    atomicAdd(&XA[2 * P + cp], a2);
                }
            }
            sum_sl = wave_sum(sum_sl);
            sum_pinf = wave_sum(sum_pinf);
            max_rp = wave_max(max_rp);
            const double mu = sum_sl / m_tot;
            __syncthreads();

            // ============ cost gradient in x-space, z-space residual ========================================
            // gx[k][cp] = sum_i' Q2[i][i'] c[m][i'] + 2 w_t (c[m][5] - goal) [terminal segments]  (:285-316)
            auto cost_grad = [&](int k, int cp) -> double {
                const int m = cp / 6, i = cp % 6;
                const double* cc = &c_[k * P + 6 * m];
                double g = 0;
#pragma unroll
                for (int ip = 0; ip < 6; ip++) g += cls.Q2[i * 6 + ip] * cc[ip];
                if (i == 5 && m >= M - ts) g += 2.0 * cls.w_t * (cc[5] - goal[k]);
                return g;
            };
            // gather x-space vector -> own z component:  (T' v)_r
            auto gatherT = [&](auto&& xs) -> double {
                const int b0 = zk * P + 6 * zm;
                double v = e0 * xs(zk, 6 * zm + 3) + e1 * xs(zk, 6 * zm + 4) + e2 * xs(zk, 6 * zm + 5);
                (void)b0;
                if (has_next) v += tb0 * xs(zk, 6 * (zm + 1) + 0) + tb1 * xs(zk, 6 * (zm + 1) + 1) + tb2 * xs(zk, 6 * (zm + 1) + 2);
                return v;
            };
            double gcost = 0, gl = 0, ga = 0;
            if (zl) {
                gcost = gatherT([&](int k, int cp) { return cost_grad(k, cp); });
                gl = gatherT([&](int k, int cp) { return XL[k * P + cp]; });
                ga = gatherT([&](int k, int cp) { return XA[k * P + cp]; });
            }
            const double rd_own = zl ? (gcost - gl) : 0.0;
            const double rdn = wave_max(fabs(rd_own));
            const double gls = fmax(1.0, wave_max(fmax(fabs(gcost), fabs(gl))));
            res_p = max_rp;
            res_d = rdn / gls;
            const double objcur = objective(false);
            res_gap = (sum_sl + sum_pinf) / (1.0 + fabs(objcur));
            // stop: primal residual (metres), scaled stationarity, and duality gap + multiplier-weighted primal
            // residual in objective units (the latter is what bounds the objective error to first order)
            if (max_rp <= 1e-9 && rdn <= 10.0 * tol * gls && res_gap <= tol) {
                status = LSCQP_STATUS_OPTIMAL;
                break;
            }

            // ============ assemble own row of Hred = T'(H + G'WG)T into the LDS scratch row ==================
            if (zl) {
                double* hrow = &Hs[lane * LDH];
#pragma unroll
                for (int cidx = 0; cidx < NZ; cidx++) hrow[cidx] = 0.0;
                // local 6x6 block (upper triangle) of (axis zk, segment m)
                auto local_block = [&](int m, double (&B)[6][6]) {
#pragma unroll
                    for (int i = 0; i < 6; i++)
#pragma unroll
                        for (int ip = i; ip < 6; ip++) B[i][ip] = cls.Q2[i * 6 + ip];
                    if (m >= M - ts) B[5][5] += 2.0 * cls.w_t;
                    const double* omk = &om_[zk * NRA];
                    const int sd = (zk == 0) ? 0 : (zk == 1) ? 3 : 5;  // S diagonal entry of axis zk
#pragma unroll
                    for (int i = 0; i < 6; i++) B[i][i] += omk[6 * m + i] + S_[(6 * m + i) * 6 + sd];
#pragma unroll
                    for (int i = 0; i < 5; i++) {
                        const double w = omk[C::OV + 5 * m + i];
                        B[i][i] += w; B[i + 1][i + 1] += w; B[i][i + 1] -= w;
                    }
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const double w = omk[C::OA + 4 * m + i];
                        B[i][i] += w; B[i][i + 1] -= 2.0 * w; B[i][i + 2] += w;
                        B[i + 1][i + 1] += 4.0 * w; B[i + 1][i + 2] -= 2.0 * w; B[i + 2][i + 2] += w;
                    }
                };
                auto sym = [](const double (&B)[6][6], int i, int ip) -> double { return i <= ip ? B[i][ip] : B[ip][i]; };
                const double ej[3] = {e0, e1, e2};
                const double tbj[3] = {tb0, tb1, tb2};
                double Bm[6][6];
                local_block(zm, Bm);
                // own segment columns (zm, j')
#pragma unroll
                for (int jp = 0; jp < 3; jp++) {
                    double v = 0;
#pragma unroll
                    for (int j = 0; j < 3; j++) v += ej[j] * sym(Bm, 3 + j, 3 + jp);
                    hrow[zk * NZA + zidx(zm, jp)] += v;
                }
                // previous segment columns (zm-1, j'): c[zm][i] = sum_j' TB[i][j'] z(zm-1, j')
                if (zm >= 1) {
#pragma unroll
                    for (int jp = 0; jp < 3; jp++) {
                        double v = 0;
#pragma unroll
                        for (int i = 0; i < 3; i++) {
                            double bi = 0;
#pragma unroll
                            for (int j = 0; j < 3; j++) bi += ej[j] * sym(Bm, i, 3 + j);
                            v += LSCQP_TB(i, jp) * bi;
                        }
                        hrow[zk * NZA + zidx(zm - 1, jp)] += v;
                    }
                }
                if (has_next) {
                    double Bn[6][6];
                    local_block(zm + 1, Bn);
                    double ta[3];  // ta[i'] = sum_i tbj[i] Bn[i][i']
#pragma unroll
                    for (int ip = 0; ip < 3; ip++) {
                        ta[ip] = 0;
#pragma unroll
                        for (int i = 0; i < 3; i++) ta[ip] += tbj[i] * sym(Bn, i, ip);
                    }
#pragma unroll
                    for (int jp = 0; jp < 3; jp++) {
                        double v = 0;
#pragma unroll
                        for (int ip = 0; ip < 3; ip++) v += ta[ip] * LSCQP_TB(ip, jp);
                        hrow[zk * NZA + zidx(zm, jp)] += v;
                        double v2 = 0;
#pragma unroll
                        for (int i = 0; i < 3; i++) v2 += tbj[i] * sym(Bn, i, 3 + jp);
                        hrow[zk * NZA + zidx(zm + 1, jp)] += v2;
                    }
                }
                // cross-axis blocks come only from the LSC rows, block-diagonal in the segment index
#pragma unroll
                for (int l = 0; l < DIM; l++) {
                    if (l == zk) continue;
                    const int a_ = zk < l ? zk : l, b_ = zk < l ? l : zk;
                    const int so = (a_ == 0) ? b_ : 4;  // (0,1)->1 (0,2)->2 (1,2)->4
#pragma unroll
                    for (int jp = 0; jp < 3; jp++) {
                        double v = ej[jp] * S_[(6 * zm + 3 + jp) * 6 + so];
                        if (has_next) {
#pragma unroll
                            for (int i = 0; i < 3; i++) v += tbj[i] * LSCQP_TB(i, jp) * S_[(6 * (zm + 1) + i) * 6 + so];
                        }
                        hrow[l * NZA + zidx(zm, jp)] += v;
                    }
                }
                // communication pairs couple the c5 variables of one axis
                if (zlast || zj == 2) {
                    const double* omc = &om_[zk * NRA + C::OC];
                    for (int up = 0; up < M; up++) {
                        if (up == zm) continue;
                        const int hi_ = zm > up ? zm : up, lo_ = zm > up ? up : zm;
                        const double w = omc[hi_ * (hi_ - 1) / 2 + lo_];
                        hrow[lane] += w;
                        hrow[zk * NZA + zidx(up, 2)] -= w;
                    }
                }
#pragma unroll
                for (int cidx = 0; cidx < NZ; cidx++) A[cidx] = hrow[cidx];
            } else {
#pragma unroll
                for (int cidx = 0; cidx < NZ; cidx++) A[cidx] = 0.0;
            }

            // ============ LDL^T in registers: lane i holds row i; pivot row broadcast by v_readlane ==========
            bool pivot_bad = false;
#pragma unroll
            for (int j = 0; j < NZ; j++) {
                const double d = bcast(A[j], j);
                if (!(d > 1e-300)) pivot_bad = true;
                const double invd = fast_rcp(d);
                if (lane == j) dinv_own = invd;
                const double li = (lane > j) ? A[j] * invd : 0.0;
#pragma unroll
                for (int kk = j + 1; kk < NZ; kk++) A[kk] = fma(-li, bcast(A[kk], j), A[kk]);
                if (lane > j) A[j] = li;
            }
            if (pivot_bad) {
                status = LSCQP_STATUS_NUMERIC;
                break;
            }
            auto solve = [&](double b) -> double {
#pragma unroll
                for (int j = 0; j < NZ; j++) {  // L w = b (unit lower)
                    const double wj = bcast(b, j);
                    if (lane > j) b = fma(-A[j], wj, b);
                }
                double x = 0;
#pragma unroll
                for (int j = NZ - 1; j >= 0; j--) {  // (D L') x = w : row i of the upper factor is A[j>i] of lane i
                    const double xj = bcast(b * dinv_own, j);
                    if (lane == j) x = xj;
                    if (lane < j) b = fma(-A[j], xj, b);
                }
                return x;
            };

            // ============ predictor ========================================================================
            const double dza = solve(zl ? (-gcost + ga) : 0.0);
            if (zl) dz_[lane] = dza;
            __syncthreads();
            // x-space direction  dc = T dz
            auto expandT = [&](double* out) {
                for (int e = lane; e < NX; e += 64) {
                    const int k = e / P, cp = e % P, m = cp / 6, i = cp % 6;
                    double v = 0;
                    if (i >= 3) v = dz_[k * NZA + zidx(m, i - 3)];
                    else if (m >= 1) {
                        const double* zz = &dz_[k * NZA + 3 * (m - 1)];
                        v = LSCQP_TB(i, 0) * zz[0] + LSCQP_TB(i, 1) * zz[1] + LSCQP_TB(i, 2) * zz[2];
                    }
                    out[e] = v;
                }
            };
            expandT(dca_);
            __syncthreads();
            // ============ pass 2: affine step length, mu_aff, corrector right-hand side ======================
            double amin = 1e300, sA = 0, sB = 0;  // sum(s dl + l ds), sum(ds dl)
#pragma unroll
            for (int u = 0; u < RPL; u++) {
                if (r_ty[u] >= 0) {
                    const double y = row_val(c_, u), dy = row_val(dca_, u);
                    double t1 = 0, t2 = 0;
                    if (r_lo[u] > -INFINITY) {
                        const double s = r_slo[u], l = r_llo[u], rp = (y - r_lo[u]) - s, w = l / s;
                        const double ds = dy + rp, dl = -l - w * ds;
                        if (ds < 0) amin = fmin(amin, -s / ds);
                        if (dl < 0) amin = fmin(amin, -l / dl);
                        sA += s * dl + l * ds;
                        sB += ds * dl;
                        const double is = 1.0 / s;
                        t1 += is;
                        t2 += -ds * dl * is - w * rp;
                    }
                    if (r_hi[u] < INFINITY) {
                        const double s = r_shi[u], l = r_lhi[u], rp = (r_hi[u] - y) - s, w = l / s;
                        const double ds = -dy + rp, dl = -l - w * ds;
                        if (ds < 0) amin = fmin(amin, -s / ds);
                        if (dl < 0) amin = fmin(amin, -l / dl);
                        sA += s * dl + l * ds;
                        sB += ds * dl;
                        const double is = 1.0 / s;
                        t1 -= is;
                        t2 -= -ds * dl * is - w * rp;
                    }
                    row_scatter(XB1, u, t1);
                    row_scatter(XB2, u, t2);
                }
            }
            if (lane < G * P) {
                const int g = lane / P, cp = lane % P;
                double b10 = 0, b11 = 0, b12 = 0, b20 = 0, b21 = 0, b22 = 0;
                const double cx = c_[cp], cy = c_[P + cp], cz = (DIM == 3) ? c_[2 * P + cp] : 0.0;
                const double dx = dca_[cp], dy = dca_[P + cp], dzz = (DIM == 3) ? dca_[2 * P + cp] : 0.0;
                for (int o = g; o < n_obs; o += G) {
                    const int e = o * P + cp;
                    const double nx = Rnx[e], ny = Rny[e], nz = Rnz[e], s = Rs[e], l = Rl[e];
                    if (l > 0.0) {
                        const double rp = (nx * cx + ny * cy + nz * cz - Rb[e]) - s, w = l / s;
                        const double ds = (nx * dx + ny * dy + nz * dzz) + rp, dl = -l - w * ds;
                        if (ds < 0) amin = fmin(amin, -s / ds);
                        if (dl < 0) amin = fmin(amin, -l / dl);
                        sA += s * dl + l * ds;
                        sB += ds * dl;
                        const double is = 1.0 / s, t2 = -ds * dl * is - w * rp;
                        b10 += is * nx; b11 += is * ny; b12 += is * nz;
                        b20 += t2 * nx; b21 += t2 * ny; b22 += t2 * nz;
                    }
                }
                atomicAdd(&XB1[cp], b10); atomicAdd(&XB1[P + cp], b11);
                atomicAdd(&XB2[cp], b20); atomicAdd(&XB2[P + cp], b21);
                if (DIM == 3) {
                    atomicAdd(&XB1[2 * P + cp], b12);
                    atomicAdd(&XB2[2 * P + cp], b22);
                }
            }
            amin = wave_min(amin);
            sA = wave_sum(sA);
            sB = wave_sum(sB);
            const double a_aff = fmin(1.0, amin);
            const double mu_aff = (sum_sl + a_aff * sA + a_aff * a_aff * sB) / m_tot;
            double sigma = fmax(mu_aff, 0.0) / mu;
            sigma = sigma * sigma * sigma;
            const double smu = sigma * mu;
            __syncthreads();
            // ============ corrector solve ==================================================================
            double gb = 0;
            if (zl) gb = gatherT([&](int k, int cp) { return smu * XB1[k * P + cp] + XB2[k * P + cp]; });
            const double dzc = solve(zl ? (-gcost + gb) : 0.0);
            __syncthreads();  // everyone is done reading dz_ (expandT above) before it is overwritten
            if (zl) dz_[lane] = dzc;
            __syncthreads();
            expandT(dc_);
            __syncthreads();
            // ============ pass 3: step length ==============================================================
            amin = 1e300;
            // corrector direction of one row; returns ds, dl
            auto row_dir = [&](double s, double l, double rp, double dya, double dyc, double& ds, double& dl) {
                const double w = l / s;
                const double dsa = dya + rp, dla = -l - w * dsa;
                ds = dyc + rp;
                dl = (smu - dsa * dla) / s - l - w * ds;
            };
#pragma unroll
            for (int u = 0; u < RPL; u++) {
                if (r_ty[u] >= 0) {
                    const double y = row_val(c_, u), dya = row_val(dca_, u), dyc = row_val(dc_, u);
                    double ds, dl;
                    if (r_lo[u] > -INFINITY) {
                        row_dir(r_slo[u], r_llo[u], (y - r_lo[u]) - r_slo[u], dya, dyc, ds, dl);
                        if (ds < 0) amin = fmin(amin, -r_slo[u] / ds);
                        if (dl < 0) amin = fmin(amin, -r_llo[u] / dl);
                    }
                    if (r_hi[u] < INFINITY) {
                        row_dir(r_shi[u], r_lhi[u], (r_hi[u] - y) - r_shi[u], -dya, -dyc, ds, dl);
                        if (ds < 0) amin = fmin(amin, -r_shi[u] / ds);
                        if (dl < 0) amin = fmin(amin, -r_lhi[u] / dl);
                    }
                }
            }
            for (int e = lane; e < nrow; e += 64) {
                const int cp = e % P;
                const double nx = Rnx[e], ny = Rny[e], nz = Rnz[e], s = Rs[e], l = Rl[e];
                if (l > 0.0) {
                    double r = nx * c_[cp] + ny * c_[P + cp], da = nx * dca_[cp] + ny * dca_[P + cp],
                           dcv = nx * dc_[cp] + ny * dc_[P + cp];
                    if (DIM == 3) {
                        r += nz * c_[2 * P + cp];
                        da += nz * dca_[2 * P + cp];
                        dcv += nz * dc_[2 * P + cp];
                    }
                    double ds, dl;
                    row_dir(s, l, (r - Rb[e]) - s, da, dcv, ds, dl);
                    if (ds < 0) amin = fmin(amin, -s / ds);
                    if (dl < 0) amin = fmin(amin, -l / dl);
                }
            }
            amin = wave_min(amin);
            const double alpha = fmin(1.0, 0.995 * amin);
            // ============ update ===========================================================================
#pragma unroll
            for (int u = 0; u < RPL; u++) {
                if (r_ty[u] >= 0) {
                    const double y = row_val(c_, u), dya = row_val(dca_, u), dyc = row_val(dc_, u);
                    double ds, dl;
                    if (r_lo[u] > -INFINITY) {
                        row_dir(r_slo[u], r_llo[u], (y - r_lo[u]) - r_slo[u], dya, dyc, ds, dl);
                        r_slo[u] += alpha * ds;
                        r_llo[u] += alpha * dl;
                    }
                    if (r_hi[u] < INFINITY) {
                        row_dir(r_shi[u], r_lhi[u], (r_hi[u] - y) - r_shi[u], -dya, -dyc, ds, dl);
                        r_shi[u] += alpha * ds;
                        r_lhi[u] += alpha * dl;
                    }
                }
            }
            for (int e = lane; e < nrow; e += 64) {
                const int cp = e % P;
                const double nx = Rnx[e], ny = Rny[e], nz = Rnz[e], s = Rs[e], l = Rl[e];
                if (l > 0.0) {
                    double r = nx * c_[cp] + ny * c_[P + cp], da = nx * dca_[cp] + ny * dca_[P + cp],
                           dcv = nx * dc_[cp] + ny * dc_[P + cp];
                    if (DIM == 3) {
                        r += nz * c_[2 * P + cp];
                        da += nz * dca_[2 * P + cp];
                        dcv += nz * dc_[2 * P + cp];
                    }
                    double ds, dl;
                    row_dir(s, l, (r - Rb[e]) - s, da, dcv, ds, dl);
                    Rs[e] = s + alpha * ds;
                    Rl[e] = l + alpha * dl;
                }
            }
            __syncthreads();  // all rows have read c_, dca_, dc_
            if (zl) z_[lane] += alpha * dzc;
            __syncthreads();
            // c = c_fixed + T z, recomputed from z so the eliminated equalities hold to rounding every iteration
            for (int e = lane; e < NX; e += 64) {
                const int k = e / P, cp = e % P, m = cp / 6, i = cp % 6;
                double v;
                if (i >= 3) v = z_[k * NZA + zidx(m, i - 3)];
                else if (m >= 1) {
                    const double* zz = &z_[k * NZA + 3 * (m - 1)];
                    v = LSCQP_TB(i, 0) * zz[0] + LSCQP_TB(i, 1) * zz[1] + LSCQP_TB(i, 2) * zz[2];
                } else v = (cp == 0) ? 0.0 : (cp == 1) ? cf1[k] : cf2[k];
                c_[e] = v;
            }
            __syncthreads();
            if (!(alpha > 1e-12) || !(mu == mu)) {  // stalled or NaN
                status = LSCQP_STATUS_NUMERIC;
                break;
            }
        }
    if (status == LSCQP_STATUS_ITER_LIMIT && res_p > 1e-6) status = LSCQP_STATUS_INFEASIBLE;
    if (status == LSCQP_STATUS_NUMERIC && res_p > 1e-6) status = LSCQP_STATUS_INFEASIBLE;

    // ---- epilogue: objective, control points back in the world frame ---------------------------------------
    const double obj = objective(true);
    for (int e = lane; e < NX; e += 64) x_out[q * NX + e] = c_[e] + org[e / P];
    if (lane == 0) {
        obj_out[q] = obj;
        status_out[q] = status;
        if (info_out) {
            info_out[q].iterations = it;
            info_out[q].reserved = 0;
            info_out[q].res_primal = res_p;
            info_out[q].res_dual = res_d;
            info_out[q].gap = res_gap;
        }
    }
}

}  // namespace lscqp
