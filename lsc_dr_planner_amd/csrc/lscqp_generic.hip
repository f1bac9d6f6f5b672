// lscqp_generic.hip — the run-time-shaped instance of the batched trajectory-QP solver: any (M, dim, planner mode) the reference
// accepts and any number of obstacles the workgroup's registers hold, gfx950 only.
//
// The reference builds its QP for whatever param.M and constraints.getObsSize() are (src/traj_optimizer.cpp:4-16, 399-437;
// src/param.cpp:71, 128-163); the register-resident kernel of lscqp_kernel.hpp exists as a table of compiled (M, dim, end stop,
// slots, wavefronts) instances, chosen for speed.  THIS kernel serves everything the table does not: horizons without a compiled
// instance (M = 9; the end-stop-free DLSC / BVC / RSFC classes at M != 5, 10; reduced systems of more than 64 rows without the
// two equal blocks the nested dissection needs), and neighbour counts beyond a compiled instance's capacity.  Same model, same
// interior-point iteration, same stopping rules and statuses as lscqp_kernel.hpp (its header comment and DESIGN.md section 2 describe
// them; the numbered steps below cite it) -- organised for generality instead of issue rate:
//   * one workgroup of 256 threads per QP, M / dim / end stop / n_obs are run-time values;
//   * the reduced KKT matrix (nz = dim (3M - 2) or 3 dim M rows, <= 108) lives in LDS as a packed lower triangle and is factorised
//     there (LDL^T, right-looking, kept unscaled: one barrier per column); the substitutions run in one wavefront with the right-hand
//     side in registers and v_readlane broadcasts;
//   * LSC rows: thread = (group g, control point cp), obstacles o = g, g + G, ... with G = floor(256 / (6M - 3)); row CONSTANTS are
//     re-read from HBM / L2 in every pass (never staged), row STATE (s, lambda: 16 bytes per row) sits in LDS behind the matrix, sized
//     per launch from n_obs_max: the capacity is what the CU's 160 KB leave -- about 90 obstacles at M = 10 in 3-D, > 500 at M = 5;
//   * the structured two-sided rows (merged interval bounds, velocity / acceleration differences, communication pairs) are spread
//     over the threads by row index, three per thread; their x-space contributions are GATHERED per control point in a fixed order
//     (no atomics anywhere: results are bitwise reproducible).
// It is several times slower per QP than a compiled instance of the same shape (the matrix is not in registers, the rows are re-read)
// and is only selected when no instance serves the launch.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#ifndef LSCQP_SOC_GATE
#define LSCQP_SOC_GATE 0.3  // rescue pass: affine step length below which the corrector's second-order term is weighted with it
#endif
#include "lscqp_kernel.hpp"  // DevClass, KQ, TBc, fast_rcp (shared with the compiled instances)
#include "lscqp_launch.hpp"

namespace lscqp_generic {

using lscqp::DevClass;
using lscqp::KQ;
using lscqp::TBc;

constexpr int kT = 256;     // threads per QP
constexpr int kR2 = 3;      // two-sided rows per thread: ceil(NOM / 256) for every shape with LDS room
constexpr int kMaxM = 12;

struct Shape {
    int M, dim, es;
    int P, CP, NZA, NZ, NX, G;
    int NV, NA, NCP, NC, OV, OA, OC, NOM;
    // LDS carve, in doubles
    int o_c, o_dca, o_dc, o_x1, o_x2, o_S, o_B, o_om, o_rv1, o_rv2, o_z, o_dz, o_zs, o_rhs, o_gc, o_dinv, o_col, o_lx, o_red, o_goal, o_H, o_rs, o_rl, total;
    __host__ __device__ static Shape make(int M, int dim, int es, int n_obs_max) {
        Shape s;
        s.M = M, s.dim = dim, s.es = es;
        s.P = 6 * M, s.CP = s.P - 3;
        s.NZA = 3 * (M - 1) + (es ? 1 : 3);
        s.NZ = dim * s.NZA, s.NX = dim * s.P;
        s.G = kT / s.CP > 0 ? kT / s.CP : 1;
        s.NV = dim * 5 * M, s.NA = dim * 4 * M, s.NCP = M * (M - 1) / 2, s.NC = dim * s.NCP;
        s.OV = s.NX, s.OA = s.NX + s.NV, s.OC = s.OA + s.NA, s.NOM = s.OC + s.NC;
        int o = 0;
        auto take = [&](int n) {
            const int at = o;
            o += (n + 1) & ~1;
            return at;
        };
        s.o_c = take(s.NX), s.o_dca = take(s.NX), s.o_dc = take(s.NX), s.o_x1 = take(s.NX), s.o_x2 = take(s.NX);
        s.o_S = take(6 * s.P), s.o_B = take(dim * M * 36), s.o_om = take(s.NOM), s.o_rv1 = take(s.NOM), s.o_rv2 = take(s.NOM);
        s.o_z = take(s.NZ), s.o_dz = take(s.NZ), s.o_zs = take(s.NZ), s.o_rhs = take(s.NZ), s.o_gc = take(s.NZ), s.o_dinv = take(s.NZ), s.o_col = take(s.NZ);
        s.o_lx = take(2 * 3 * s.CP);  // LSC shares of two x-space vectors, per control point
        s.o_red = take(3 * kT + 8), s.o_goal = take(4);
        // the packed lower triangle of the reduced matrix; while the row passes run the same region holds the groups' partial sums
        const int tri = s.NZ * (s.NZ + 1) / 2, part = s.G * s.CP * 12;
        s.o_H = take(tri > part ? tri : part);
        s.o_rs = take(n_obs_max * s.CP), s.o_rl = take(n_obs_max * s.CP);  // LSC row state [obstacle][control point]
        s.total = o;
        return s;
    }
};

// Development aid (-DLSCQP_GEN_TIMING, tools/bench_generic.py --phases): cycles per phase, summed over the workgroups by thread 0.
#ifdef LSCQP_GEN_TIMING
__device__ unsigned long long gen_cycles[16];
#define GEN_T(slot)                                                        \
    do {                                                                   \
        __syncthreads();                                                   \
        const unsigned long long now_ = __builtin_readcyclecounter();      \
        if (tid == 0) atomicAdd(&gen_cycles[slot], now_ - tprev_);         \
        tprev_ = __builtin_readcyclecounter();                             \
    } while (0)
#else
#define GEN_T(slot) \
    do {            \
    } while (0)
#endif

__device__ __forceinline__ double rcp2(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;
}

__global__ __launch_bounds__(kT) void pdip_generic_kernel(DevClass cls, int M_, int dim_, int es_, int64_t n, const lscqp_header* __restrict__ hdr,
                                                          const lscqp_row* __restrict__ rows, const uint64_t* __restrict__ row_offsets,
                                                          const lscqp_box* __restrict__ sfc, const double* __restrict__ x_init,
                                                          double* __restrict__ x_out, double* __restrict__ obj_out,
                                                          int32_t* __restrict__ status_out, lscqp_info* __restrict__ info_out) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const Shape S = Shape::make(M_, dim_, es_, cls.n_obs_max);
    const int M = S.M, DIM = S.dim, P = S.P, CP = S.CP, NZA = S.NZA, NZ = S.NZ, NX = S.NX, G = S.G;
    const bool ES = S.es != 0;
    double* const c_ = smem + S.o_c;
    double* const dca_ = smem + S.o_dca;
    double* const dc_ = smem + S.o_dc;
    double* const X1 = smem + S.o_x1;  // pass 1: G' lambda            pass 2: G' (1/s)
    double* const X2 = smem + S.o_x2;  // pass 1: G' q_aff             pass 2: G' (-ds_a dl_a / s - w rp)
    double* const S_ = smem + S.o_S;
    double* const B_ = smem + S.o_B;
    double* const om_ = smem + S.o_om;
    double* const rv1_ = smem + S.o_rv1;
    double* const rv2_ = smem + S.o_rv2;
    double* const z_ = smem + S.o_z;
    double* const dz_ = smem + S.o_dz;
    double* const zs_ = smem + S.o_zs;
    double* const rhs_ = smem + S.o_rhs;
    double* const gc_ = smem + S.o_gc;
    double* const dinv_ = smem + S.o_dinv;
    double* const col_ = smem + S.o_col;
    double* const LX_ = smem + S.o_lx;
    double* const red_ = smem + S.o_red;
    double* const goal_ = smem + S.o_goal;
    double* const H_ = smem + S.o_H;
    double* const Rs_ = smem + S.o_rs;
    double* const Rl_ = smem + S.o_rl;
    const int tid = threadIdx.x;
    if ((int64_t)blockIdx.x >= n) return;
    const int64_t q = cls.order ? (int64_t)cls.order[blockIdx.x] : (int64_t)blockIdx.x;  // (lscqp_kernel.hpp: work order of the launch; no queue here)
#ifdef LSCQP_GEN_TIMING
    unsigned long long tprev_ = __builtin_readcyclecounter();
#endif
    const lscqp_header* Hd = hdr + q;
    int flags = 0, it_before = 0;
    // cls.repair == 2: the RESCUE pass (lscqp_api.hip) -- only instances that ran into the iteration limit or broke down numerically,
    // re-solved from the default start with the corrector's second-order term weighted (see pass 2)
    const bool rescue = cls.repair == 2;
    if (cls.repair) {
        const int st0 = status_out[q];
        if (st0 == LSCQP_STATUS_OPTIMAL || st0 == LSCQP_STATUS_CAPACITY) return;
        // (round 6: an instance the dual active-set phase PROVED infeasible -- INFEASIBLE with LSCQP_INFO_ACTIVE_SET -- is finished: the pass right
        // behind the phase takes only what the phase marked ITER_LIMIT; a later pass recognises the verdict by the flag)
        if (cls.repair == 3 && st0 != LSCQP_STATUS_ITER_LIMIT) return;
        if (st0 == LSCQP_STATUS_INFEASIBLE && info_out && (info_out[q].flags & LSCQP_INFO_ACTIVE_SET)) return;
        if (rescue && st0 != LSCQP_STATUS_ITER_LIMIT && st0 != LSCQP_STATUS_NUMERIC) return;
        if (cls.repair != 3) {  // (3: the FIRST interior-point pass, behind the dual active-set phase of lscqp_das.hip)
            flags |= LSCQP_INFO_REPAIRED | (rescue ? LSCQP_INFO_RESCUED : 0);
            if (info_out) it_before = info_out[q].iterations;
        }
    }
    const int n_obs = Hd->n_obs;
    if (n_obs > cls.n_obs_max) {  // refused, never truncated (see lscqp_kernel.hpp): the launch sized the row state for n_obs_max
        for (int e = tid; e < NX; e += kT) x_out[q * NX + e] = x_init ? x_init[q * NX + e] : Hd->p0[e / P];
        if (tid == 0) {
            obj_out[q] = 0.0;
            status_out[q] = LSCQP_STATUS_CAPACITY;
            if (info_out) {
                info_out[q].iterations = 0;
                info_out[q].flags = flags;
                info_out[q].res_primal = info_out[q].res_dual = info_out[q].gap = 0.0;
            }
        }
        return;
    }
    const double dt = cls.dt;
    const double org[3] = {Hd->p0[0], Hd->p0[1], Hd->p0[2]};
    auto orgk = [&](int k) -> double { return k == 0 ? org[0] : (k == 1 ? org[1] : org[2]); };
    const double g0 = Hd->goal[0] - org[0], g1 = Hd->goal[1] - org[1], g2 = Hd->goal[2] - org[2];
    if (tid < 3) goal_[tid] = Hd->goal[tid] - Hd->p0[tid];
    int ts = Hd->terminal_segments;
    if (ts <= 0) {
        ts = (int)((M * dt - sqrt(g0 * g0 + g1 * g1 + g2 * g2) / Hd->nominal_velocity + 1e-9) / dt);
        if (ts < 1) ts = 1;
    }
    if (ts > M) ts = M;
    const double q2s = cls.q2s, wt2 = 2.0 * cls.w_t;

    // ---- block reductions through LDS, fixed order (bitwise reproducible) -----------------------------------
    auto block_reduce3 = [&](double& a, double& b, double& c, bool max_a, bool max_b, bool max_c) {
        __syncthreads();
        red_[tid] = a;
        red_[kT + tid] = b;
        red_[2 * kT + tid] = c;
        __syncthreads();
        for (int w = kT / 2; w >= 1; w >>= 1) {
            if (tid < w) {
                red_[tid] = max_a ? fmax(red_[tid], red_[tid + w]) : red_[tid] + red_[tid + w];
                red_[kT + tid] = max_b ? fmax(red_[kT + tid], red_[kT + tid + w]) : red_[kT + tid] + red_[kT + tid + w];
                red_[2 * kT + tid] = max_c ? fmax(red_[2 * kT + tid], red_[2 * kT + tid + w]) : red_[2 * kT + tid] + red_[2 * kT + tid + w];
            }
            __syncthreads();
        }
        a = red_[0];
        b = red_[kT];
        c = red_[2 * kT];
    };
    auto block_sum = [&](double v) -> double {
        double b = 0, c = 0;
        block_reduce3(v, b, c, false, false, false);
        return v;
    };
    auto block_max = [&](double v) -> double {
        double b = 0, c = 0;
        block_reduce3(v, b, c, true, true, true);
        return v;
    };

    // ---- index helpers ------------------------------------------------------------------------------------------
    auto zidx = [&](int m, int j) -> int { return (ES && m == M - 1) ? 3 * (M - 1) : 3 * m + j; };
    // x-space vector = T * (z-space vector), all NX entries
    auto expandT = [&](const double* zsrc, double* out, bool keep_fixed) {
        for (int e = tid; e < NX; e += kT) {
            const int k = e / P, cp = e % P, m = cp / 6, i = cp % 6;
            double v;
            if (i >= 3) {
                v = zsrc[k * NZA + zidx(m, i - 3)];
            } else if (m >= 1) {
                const double* zz = &zsrc[k * NZA + 3 * (m - 1)];
                v = TBc(i, 0) * zz[0] + TBc(i, 1) * zz[1] + TBc(i, 2) * zz[2];
            } else {
                v = 0.0;
            }
            if (!(keep_fixed && cp < 3)) out[e] = v;
        }
    };
    // (T' X)_zi: the x entries a z variable drives (its own control point(s), and c0..c2 of the next segment through TB)
    auto gatherT = [&](const double* X, int zi) -> double {
        const int k = zi / NZA, a = zi % NZA;
        const bool zlast = ES && a == 3 * (M - 1);
        const int m = zlast ? M - 1 : a / 3, j = zlast ? 0 : a % 3;
        const double* xs = &X[k * P + 6 * m];
        if (zlast) return xs[3] + xs[4] + xs[5];
        double v = xs[3 + j];
        if (m + 1 < M) {
            const double* xn = &X[k * P + 6 * (m + 1)];
            v += TBc(0, j) * xn[0] + TBc(1, j) * xn[1] + TBc(2, j) * xn[2];
        }
        return v;
    };

    // ---- control points: fixed part from (p0, v0, a0) (:321-338), free part from the initial trajectory or "stay at c2" ------------
    for (int e = tid; e < NX; e += kT) {
        const int k = e / P, cp = e % P;
        const double cf1 = Hd->v0[k] * dt * 0.2;
        const double cf2 = Hd->a0[k] * dt * dt * 0.05 + 2.0 * cf1;
        c_[e] = (cp == 0) ? 0.0 : (cp == 1) ? cf1 : cf2;
        dca_[e] = 0.0;
        dc_[e] = 0.0;
    }
    for (int zi = tid; zi < NZ; zi += kT) {
        const int k = zi / NZA, a = zi % NZA;
        const bool zlast = ES && a == 3 * (M - 1);
        const int m = zlast ? M - 1 : a / 3, j = zlast ? 0 : a % 3;
        const double cf1 = Hd->v0[k] * dt * 0.2;
        double v = Hd->a0[k] * dt * dt * 0.05 + 2.0 * cf1;
        if (x_init) v = x_init[q * NX + k * P + 6 * m + (zlast ? 5 : 3 + j)] - orgk(k);
        z_[zi] = v;
    }
    __syncthreads();
    expandT(z_, c_, true);
    __syncthreads();

    // ---- two-sided rows: row r = tid + 256 u of [interval NX | velocity NV | acceleration NA | pair NC] -----------------------
    // a row's value on an x-space vector v is cf0 v[i0] + cf1 v[i1] + cf2 v[i2]
    int t_i0[kR2], t_i1[kR2], t_i2[kR2];
    double t_c0[kR2], t_c1[kR2], t_c2[kR2];
    bool t_on[kR2];
    double t_lo[kR2], t_hi[kR2], t_sl[kR2], t_sh[kR2], t_ll[kR2], t_lh[kR2];
    {
        const double rho_pair = 0.5 * cls.comm_range - Hd->radius;  // :484
        const double rho_wp = 0.5 * cls.comm_range - 1e-5;          // :495
        const bool comm_on = cls.comm_range > 0;
#pragma unroll
        for (int u = 0; u < kR2; u++) {
            const int r = tid + kT * u;
            int i0 = 0, i1 = 0, i2 = 0;
            double c0 = 0, c1 = 0, c2 = 0, lo = -1.0, hi = 1.0;
            bool on = false;
            if (r < S.OV) {  // interval on one control point: world box, corridor, (m, mi = 0) communication rows, waypoint rows
                const int k = r / P, cp = r % P, m = cp / 6;
                if (cp >= 3) {
                    on = true;
                    i0 = r, c0 = 1.0;
                    const double ok_ = orgk(k);
                    lo = cls.world_min[k] - ok_, hi = cls.world_max[k] - ok_;  // :252-253, 260-265
                    if (cls.rsfc && k == 2 && m == 0) lo = -100.0 - ok_, hi = 100.0 - ok_;  // :255-258
                    if (cls.use_sfc) {                                                  // :372-397
                        lo = fmax(lo, sfc[q * M + m].bmin[k] - ok_);
                        hi = fmin(hi, sfc[q * M + m].bmax[k] - ok_);
                    }
                    if (comm_on && cp % 6 == 5) {  // :482-487 with mi = 0, and :494-497
                        const double wpk = Hd->next_waypoint[k] - ok_;
                        lo = fmax(lo, fmax(-rho_pair, wpk - rho_wp));
                        hi = fmin(hi, fmin(rho_pair, wpk + rho_wp));
                    }
                }
            } else if (r < S.OA) {  // velocity (m, i): c[i+1] - c[i], |.| <= vmax dt / n   (:448-453)
                const int v = r - S.OV, k = v / (5 * M), rr = v % (5 * M), m = rr / 5, i = rr % 5;
                if (!(m == 0 && i < 2)) {
                    on = true;
                    i0 = k * P + 6 * m + i, i1 = i0 + 1, c0 = -1.0, c1 = 1.0;
                    hi = Hd->vmax[k] * dt * 0.2, lo = -hi;
                }
            } else if (r < S.OC) {  // acceleration (m, i): c[i+2] - 2 c[i+1] + c[i]   (:462-471)
                const int a = r - S.OA, k = a / (4 * M), rr = a % (4 * M), m = rr / 4, i = rr % 4;
                if (!(m == 0 && i < 1)) {
                    on = true;
                    i0 = k * P + 6 * m + i, i1 = i0 + 1, i2 = i0 + 2, c0 = 1.0, c1 = -2.0, c2 = 1.0;
                    hi = Hd->amax[k] * dt * dt * 0.05, lo = -hi;
                }
            } else if (r < S.NOM) {  // pair (uu, up < uu): c[uu][5] - c[up+1][0]   (:482-487 with mi = up + 1 >= 1)
                const int cc = r - S.OC, k = cc / S.NCP, ci = cc % S.NCP;
                if (comm_on) {
                    int uu = 1;
                    while (uu * (uu + 1) / 2 <= ci) uu++;
                    const int up = ci - uu * (uu - 1) / 2;
                    on = true;
                    i0 = k * P + 6 * (up + 1), i1 = k * P + 6 * uu + 5, c0 = -1.0, c1 = 1.0;
                    hi = rho_pair, lo = -rho_pair;
                }
            }
            t_i0[u] = i0, t_i1[u] = i1, t_i2[u] = i2, t_c0[u] = c0, t_c1[u] = c1, t_c2[u] = c2, t_on[u] = on, t_lo[u] = lo, t_hi[u] = hi;
        }
    }
    auto row_val = [&](const double* v, int u) -> double { return t_c0[u] * v[t_i0[u]] + t_c1[u] * v[t_i1[u]] + t_c2[u] * v[t_i2[u]]; };
    // x-space entry e of G2' rv for a per-row scalar array rv[NOM] of the two-sided rows, gathered in a fixed order
    auto gather_rows = [&](const double* rv, int e) -> double {
        const int k = e / P, cp = e % P, m = cp / 6, i = cp % 6;
        double v = rv[e];  // interval
        const double* rvV = &rv[S.OV + k * 5 * M + 5 * m];
        if (i >= 1) v += rvV[i - 1];
        if (i <= 4) v -= rvV[i];
        const double* rvA = &rv[S.OA + k * 4 * M + 4 * m];
        if (i >= 2) v += rvA[i - 2];
        if (i >= 1 && i <= 4) v -= 2.0 * rvA[i - 1];
        if (i <= 3) v += rvA[i];
        const double* rvC = &rv[S.OC + k * S.NCP];
        if (i == 5)
            for (int up = 0; up < m; up++) v += rvC[m * (m - 1) / 2 + up];
        if (i == 0 && m >= 1)
            for (int uu = m; uu < M; uu++) v -= rvC[uu * (uu - 1) / 2 + (m - 1)];
        return v;
    };

    // ---- LSC rows: thread (lg, lcp), obstacles o = lg + G u; constants from HBM / L2, state in registers ----------------------
    const bool ll = tid < G * CP;
    const int lg = ll ? tid / CP : 0, lcp = ll ? tid % CP : 0, lx = lcp + 3;
    const uint64_t r0 = (n_obs > 0 && row_offsets) ? row_offsets[q] : 0;
    struct Row {
        double nx, ny, nz, b;
    };
    auto load_row = [&](int o) -> Row {  // row of obstacle o on this thread's control point, translated to the agent's origin;
        Row R{0.0, 0.0, 0.0, -1.0};       // a dropped row (:409-411): n = 0, b = -1
        {
            double vx, vy, vz, vb;
            const uint64_t e = r0 + (uint64_t)(o * P + lx);
            if (cls.rows_f32) {
                const float4 f = reinterpret_cast<const float4*>(rows)[e];
                vx = f.x, vy = f.y, vz = f.z, vb = f.w;
            } else {
                const double4 v = *reinterpret_cast<const double4*>(&rows[e]);
                vx = v.x, vy = v.y, vz = v.z, vb = v.w;
            }
            if (!(sqrt(vx * vx + vy * vy + vz * vz) < 1e-5)) {  // (:409-411)
                R.nx = vx, R.ny = vy, R.nz = (DIM == 3) ? vz : 0.0;
                R.b = vb - (vx * org[0] + vy * org[1] + (DIM == 3 ? vz * org[2] : 0.0));
            }
        }
        return R;
    };
    const int o_end = ll ? n_obs : 0;  // this thread's rows: for (o = lg; o < o_end; o += G), state at [o * CP + lcp]

    // ---- centred start (lscqp_kernel.hpp "initial slacks / multipliers") ---------------------------------------------------------
    const double pull = (double)ts * cls.w_t * fmax(fabs(g0), fmax(fabs(g1), fabs(g2)));
    const double mu_scale = fmin(1e3, fmax(1.0, 0.25 * pull));
    const double MU0 = (x_init ? cls.warm_mu0 : LSCQP_COLD_MU0) * mu_scale, S0MIN = x_init ? cls.warm_s0 : 0.1;
    int status = LSCQP_STATUS_ITER_LIMIT;
    auto centre_rows = [&](double mu_c, double s_c, double& cnt, bool& bad) {
#pragma unroll
        for (int u = 0; u < kR2; u++) {
            double sl0 = 1.0, sh0 = 1.0, l0 = 0.0;
            if (t_on[u]) {
                const double y = row_val(c_, u);
                if (t_lo[u] > t_hi[u]) bad = true;
                sl0 = fmax(y - t_lo[u], s_c);
                sh0 = fmax(t_hi[u] - y, s_c);
                l0 = 1.0;
                cnt += 2.0;
            }
            t_sl[u] = sl0, t_sh[u] = sh0;
            t_ll[u] = l0 * mu_c * rcp2(sl0), t_lh[u] = l0 * mu_c * rcp2(sh0);
        }
        const double cx = c_[lx], cy = c_[P + lx], cz = (DIM == 3) ? c_[2 * P + lx] : 0.0;
        for (int o = lg; o < o_end; o += G) {
            const Row R = load_row(o);
            double s_init = 1.0, l_init = 0.0;
            if ((R.nx != 0.0) || (R.ny != 0.0) || (R.nz != 0.0)) {
                s_init = fmax(R.nx * cx + R.ny * cy + R.nz * cz - R.b, s_c);
                l_init = mu_c * rcp2(s_init);
                cnt += 1.0;
            }
            Rs_[o * CP + lcp] = s_init, Rl_[o * CP + lcp] = l_init;
        }
    };
    double m_tot = 0;
    {
        double cnt = 0;
        bool bad = false;
        centre_rows(MU0, S0MIN, cnt, bad);
        m_tot = block_sum(cnt);
        if (block_max(bad ? 1.0 : 0.0) > 0.0) status = LSCQP_STATUS_INFEASIBLE;
    }
    const double inv_m = 1.0 / m_tot;

    auto objective = [&](bool ref_rounding) -> double {  // cplex.getObjValue() (src/traj_optimizer.cpp:100), see lscqp_kernel.hpp
        double part = 0.0;
        if (tid < DIM * M) {
            const int k = tid / M, m = tid % M;
            const double* cc = &c_[k * P + 6 * m];
            const double j0 = (cc[3] - cc[0]) - 3.0 * (cc[2] - cc[1]);
            const double j1 = (cc[4] - cc[1]) - 3.0 * (cc[3] - cc[2]);
            const double j2 = (cc[5] - cc[2]) - 3.0 * (cc[4] - cc[3]);
            const double quad = 0.2 * (j0 * j0 + j2 * j2) + (2.0 / 15.0) * j1 * j1 + 0.2 * (j0 * j1 + j1 * j2) + (1.0 / 15.0) * j0 * j2;
            part = 0.5 * q2s * 3600.0 * quad;
            if (ref_rounding) {
                const double ok_ = Hd->p0[k];
                double corr = 0;
                for (int i = 0; i < 6; i++) {
                    double r = 0;
                    for (int ip = 0; ip < 6; ip++) r += cls.dQ[i * 6 + ip] * (cc[ip] + ok_);
                    corr += r * (cc[i] + ok_);
                }
                part += cls.w_c * corr;
            }
            const double dgoal = cc[5] - goal_[k];
            part += (m >= M - ts) ? cls.w_t * dgoal * dgoal : 0.0;
        }
        return block_sum(part);
    };

    // ---- the reduced matrix, packed lower triangle; x entries a z variable drives --------------------------------------------
    auto tri = [](int i, int j) -> int { return i * (i + 1) / 2 + j; };  // j <= i
    // (control points are carried as (axis, 6 m + i) pairs: an integer division by the run-time P costs ~40 instructions on this
    // hardware, and the assembly evaluated a dozen of them per K entry -- 280 k of its 317 k cycles at nz = 90)
    struct Drv {
        int k, cp[4];  // axis, control point index 6 m + i within the axis
        double w[4];
    };
    auto drive = [&](int k, int a) -> Drv {  // z variable (axis k, index a within the axis)
        Drv D;
        D.k = k;
        const bool zlast = ES && a == 3 * (M - 1);
        const int m = zlast ? M - 1 : a / 3, j = zlast ? 0 : a % 3;
        const int base = 6 * m;
        const bool nxt = !zlast && (m + 1 < M);
        D.cp[0] = zlast ? base + 3 : base + 3 + j, D.w[0] = 1.0;
#pragma unroll
        for (int t = 0; t < 3; t++) {
            D.cp[1 + t] = zlast ? (t < 2 ? base + 4 + t : base) : (nxt ? base + 6 + t : base);
            D.w[1 + t] = zlast ? (t < 2 ? 1.0 : 0.0) : (nxt ? TBc(t, j) : 0.0);
        }
        return D;
    };
    const bool comm_on_k = cls.comm_range > 0;
    // entry ((ka, ca), (kb, cb)) of K = H + G'WG in x-space; B_ holds the same-axis same-segment 6x6 blocks
    auto Kentry = [&](int ka, int ca, int kb, int cb) -> double {
        const int ma = ca / 6, ia = ca - 6 * ma, mb = cb / 6, ib = cb - 6 * mb;  // (division by a constant: a multiply)
        double v = 0.0;
        if (ka == kb) {
            if (ma == mb) v += B_[((ka * M + ma) * 6 + ia) * 6 + ib];
            if (comm_on_k) {  // off-diagonal entries of the communication pairs: (uu, 5) with (up + 1, 0), weight -w(uu, up)
                const double* omc = &om_[S.OC + ka * S.NCP];
                if (ia == 5 && ib == 0 && mb >= 1 && ma >= mb) v -= omc[ma * (ma - 1) / 2 + (mb - 1)];
                if (ib == 5 && ia == 0 && ma >= 1 && mb >= ma) v -= omc[mb * (mb - 1) / 2 + (ma - 1)];
            }
        } else if (ca == cb && ca >= 3) {  // cross-axis block of the LSC rows on one control point
            const int lo_ = ka < kb ? ka : kb, hi_ = ka < kb ? kb : ka;
            const int so = (lo_ == 0) ? hi_ : 4;  // (0,1) -> 1, (0,2) -> 2, (1,2) -> 4
            v += S_[ca * 6 + so];
        }
        return v;
    };

    double res_p = 0, res_d = 0, res_gap = 0, snap_p = 0, snap_d = 0, snap_gap = 0, obj_abs = 0;
    bool restore = false;
    int it = 0, near_cnt = 0, floor_cnt = 0;
    float rp_ref = 3.0e38f;
    int stalled = 0;
    float alpha_first = 1.0f;
    bool net_done = false;
    float gap_mark = 3.0e38f;
    int jam_since = 0;
    bool recentred = false;
    int shift_level = 0;  // (lscqp_kernel.hpp: a lost pivot repeats the iteration with 1e-14 / 1e-12 max|K| on the diagonal)
    bool repeated = false;  // (lscqp_kernel.hpp: a repeated iteration takes the point's tests and counters once, and is counted once)
    const double tol = cls.tol;

    if (status != LSCQP_STATUS_INFEASIBLE)
        for (it = 0; it < cls.max_iter; it++) {
            GEN_T(0);
            // ============ pass 1: residuals, weights, per-control-point blocks ===========================================
            double sum_sl = 0, sum_pinf = 0, max_rp = 0;
#pragma unroll
            for (int u = 0; u < kR2; u++) {
                const int r = tid + kT * u;
                const double y = row_val(c_, u);
                const double rpl = t_on[u] ? (y - t_lo[u]) - t_sl[u] : 0.0, rph = t_on[u] ? (t_hi[u] - y) - t_sh[u] : 0.0;
                const double wl = t_ll[u] * rcp2(t_sl[u]), wh = t_lh[u] * rcp2(t_sh[u]);
                sum_sl += t_sl[u] * t_ll[u] + t_sh[u] * t_lh[u];
                sum_pinf += t_ll[u] * fabs(rpl) + t_lh[u] * fabs(rph);
                max_rp = fmax(max_rp, fmax(fabs(rpl), fabs(rph)));
                if (r < S.NOM) {
                    om_[r] = wl + wh;
                    rv1_[r] = t_ll[u] - t_lh[u];
                    rv2_[r] = wh * rph - wl * rpl;
                }
            }
            {
                double s00 = 0, s01 = 0, s02 = 0, s11 = 0, s12 = 0, s22 = 0, l0 = 0, l1 = 0, l2 = 0, a0 = 0, a1 = 0, a2 = 0;
                const double cx = c_[lx], cy = c_[P + lx], cz = (DIM == 3) ? c_[2 * P + lx] : 0.0;
                for (int o = lg; o < o_end; o += G) {
                    const Row R = load_row(o);
                    const double s = Rs_[o * CP + lcp], lam = Rl_[o * CP + lcp];
                    const double rp = (R.nx * cx + R.ny * cy + R.nz * cz - R.b) - s;
                    const double w = lam * rcp2(s);
                    sum_sl += s * lam;
                    sum_pinf += lam * fabs(rp);
                    max_rp = fmax(max_rp, fabs(rp));
                    const double wx = w * R.nx, wy = w * R.ny, wz = w * R.nz;
                    s00 += wx * R.nx, s01 += wx * R.ny, s02 += wx * R.nz, s11 += wy * R.ny, s12 += wy * R.nz, s22 += wz * R.nz;
                    l0 += lam * R.nx, l1 += lam * R.ny, l2 += lam * R.nz;
                    const double qa = -w * rp;
                    a0 += qa * R.nx, a1 += qa * R.ny, a2 += qa * R.nz;
                }
                if (ll) {  // the groups' partial sums, combined below in group order
                    double* pp = &H_[(lg * CP + lcp) * 12];
                    pp[0] = s00, pp[1] = s01, pp[2] = s02, pp[3] = s11, pp[4] = s12, pp[5] = s22;
                    pp[6] = l0, pp[7] = l1, pp[8] = l2, pp[9] = a0, pp[10] = a1, pp[11] = a2;
                }
            }
            block_reduce3(sum_sl, sum_pinf, max_rp, false, false, true);  // (its barriers also publish om_, rv*, the partial sums)
            const double mu = sum_sl * inv_m;
            if (tid < CP) {
                double acc[12];
#pragma unroll
                for (int t = 0; t < 12; t++) acc[t] = 0.0;
                for (int g = 0; g < G; g++)
#pragma unroll
                    for (int t = 0; t < 12; t++) acc[t] += H_[(g * CP + tid) * 12 + t];
                const int cp6 = (tid + 3) * 6;
#pragma unroll
                for (int t = 0; t < 6; t++) S_[cp6 + t] = acc[t];
#pragma unroll
                for (int t = 0; t < 3; t++) LX_[t * CP + tid] = acc[6 + t], LX_[(3 + t) * CP + tid] = acc[9 + t];
            }
            if (tid < 18) S_[tid] = 0.0;  // the initial state carries no LSC rows
            __syncthreads();
            for (int e = tid; e < NX; e += kT) {
                const int k = e / P, cp = e % P;
                const double lsc1 = cp >= 3 ? LX_[k * CP + cp - 3] : 0.0, lsc2 = cp >= 3 ? LX_[(3 + k) * CP + cp - 3] : 0.0;
                X1[e] = gather_rows(rv1_, e) + lsc1;
                X2[e] = gather_rows(rv2_, e) + lsc2;
            }
            __syncthreads();

            GEN_T(1);
            // ============ cost gradient in z-space, residual norms, convergence ========================================
            double rdn = 0, gls = 0;
            for (int zi = tid; zi < NZ; zi += kT) {
                const int k = zi / NZA, a = zi % NZA;
                const bool zlast = ES && a == 3 * (M - 1);
                const int m = zlast ? M - 1 : a / 3, j = zlast ? 0 : a % 3;
                const double* cs = &c_[k * P + 6 * m];
                double gx[6];  // gradient of the cost w.r.t. this segment's control points (the ones this z drives)
                double gcost = 0;
#pragma unroll
                for (int i = 3; i < 6; i++) {
                    double gq = 0;
#pragma unroll
                    for (int ip = 0; ip < 6; ip++) gq += KQ(i, ip) * cs[ip];
                    gx[i] = q2s * gq;
                }
                gx[5] += (m >= M - ts) ? wt2 * (cs[5] - goal_[k]) : 0.0;
                if (zlast) {
                    gcost = gx[3] + gx[4] + gx[5];
                } else {
                    gcost = j == 0 ? gx[3] : (j == 1 ? gx[4] : gx[5]);
                    if (m + 1 < M) {
                        const double* cn = &c_[k * P + 6 * (m + 1)];
#pragma unroll
                        for (int r = 0; r < 3; r++) {
                            double gq = 0;
#pragma unroll
                            for (int ip = 0; ip < 6; ip++) gq += KQ(r, ip) * cn[ip];
                            gcost += TBc(r, j) * q2s * gq;
                        }
                    }
                }
                const double gl = gatherT(X1, zi), ga = gatherT(X2, zi);
                gc_[zi] = gcost;
                rhs_[zi] = -gcost + ga;  // predictor right-hand side
                rdn = fmax(rdn, fabs(gcost - gl));
                gls = fmax(gls, fmax(fabs(gcost), fabs(gl)));
            }
            {
                double dummy = 0;
                block_reduce3(rdn, gls, dummy, true, true, true);
            }
            gls = fmax(1.0, gls);
            res_p = max_rp;
            res_d = rdn / gls;
            if (!repeated && (it & 3) == 2) {  // infeasibility: a primal residual that stalls, or runaway multipliers (see lscqp_kernel.hpp)
                stalled = (it >= 10 && max_rp > 1e-4 && max_rp > 0.7 * (double)rp_ref) ? stalled + 1 : 0;
                if (stalled >= 2 || (it >= 10 && max_rp > 1e-5 && sum_pinf > 1e6)) {
                    status = LSCQP_STATUS_INFEASIBLE;
                    break;
                }
                rp_ref = (float)max_rp;
            }
            if (repeated) {
                // (nothing: see `repeated`)
            } else if (max_rp <= 1e-9 && rdn <= 1e-6 * gls) {
                obj_abs = fabs(objective(false));
                res_gap = (sum_sl + sum_pinf) / (1.0 + obj_abs);
                if (res_gap <= tol || (res_gap <= 10.0 * tol && rdn <= 1e-7 * gls)) {
                    // the BEST remembered point, not the latest (lscqp_kernel.hpp: gap target first, then the smaller stationarity)
                    const bool at_target = res_gap <= tol, had_target = snap_gap <= tol;
                    if (floor_cnt == 0 || (at_target && !had_target) || (at_target == had_target && res_d < snap_d)) {
                        for (int zi = tid; zi < NZ; zi += kT) zs_[zi] = z_[zi];
                        snap_p = max_rp, snap_d = res_d, snap_gap = res_gap;
                    }
                }
                if (res_gap <= tol) {
                    floor_cnt++;
                    if (rdn <= 1e-8 * gls) {
                        near_cnt++;
                        if (rdn <= 10.0 * tol * gls || near_cnt >= LSCQP_NEAR_CONFIRM) {
                            status = LSCQP_STATUS_OPTIMAL;
                            break;
                        }
                    }
                } else if (res_gap <= 10.0 * tol && rdn <= 1e-7 * gls) {
                    floor_cnt++;
                }
                if (res_gap > tol && res_gap <= 1e-4) {
                    if ((float)res_gap <= 0.1f * gap_mark) {
                        gap_mark = (float)res_gap;
                        jam_since = 0;
                    } else {
                        jam_since++;
                    }
                }
            } else
                res_gap = sum_sl + sum_pinf;
            repeated = false;
            const bool net = cls.warm_net > 0 && x_init != nullptr && it == 1 && !net_done && (double)alpha_first < cls.warm_net;
            if (net || (!recentred && jam_since >= 6 && floor_cnt == 0)) {
                double cnt_ = 0;
                bool bad_ = false;
                centre_rows(1e-3 * mu_scale, 0.03, cnt_, bad_);
                if (net) net_done = true; else recentred = true;
                rp_ref = 3.0e38f;
                near_cnt = 0;
                __syncthreads();
                continue;
            }

            GEN_T(2);
            double diag_shift = 0.0;
            if (shift_level > 0) {  // uniform; rare
                double big = 0.0;
                for (int e = tid; e < 6 * P; e += kT) big = fmax(big, fabs(S_[e]));
                for (int e = tid; e < S.NOM; e += kT) big = fmax(big, om_[e]);
                big = block_max(big);
                diag_shift = (shift_level == 1 ? 1e-14 : 1e-12) * fmax(big, 1.0);
            }
            // ============ assembly: same-axis same-segment blocks, then Hred = T'(H + G'WG)T, packed lower triangle ==========
            for (int e = tid; e < DIM * M * 36; e += kT) {
                const int km = e / 36, r36 = e - 36 * km, k = (km >= 2 * M) ? 2 : (km >= M ? 1 : 0), m = km - k * M, i = r36 / 6, ip = r36 - 6 * i;
                double v = q2s * KQ(i, ip);
                if (i == 5 && ip == 5 && m >= M - ts) v += wt2;
                const double* omi = &om_[k * P + 6 * m];
                const double* omv = &om_[S.OV + k * 5 * M + 5 * m];
                const double* oma = &om_[S.OA + k * 4 * M + 4 * m];
                const int d = ip - i;
                if (d == 0) {
                    v += omi[i] + S_[(6 * m + i) * 6 + (k == 0 ? 0 : (k == 1 ? 3 : 5))];
                    if (i <= 4) v += omv[i];
                    if (i >= 1) v += omv[i - 1];
                    if (i <= 3) v += oma[i];
                    if (i >= 1 && i <= 4) v += 4.0 * oma[i - 1];
                    if (i >= 2) v += oma[i - 2];
                    if (comm_on_k) {  // diagonal of the communication pairs
                        const double* omc = &om_[S.OC + k * S.NCP];
                        if (i == 5)
                            for (int up = 0; up < m; up++) v += omc[m * (m - 1) / 2 + up];
                        if (i == 0 && m >= 1)
                            for (int uu = m; uu < M; uu++) v += omc[uu * (uu - 1) / 2 + (m - 1)];
                    }
                } else if (d == 1 || d == -1) {
                    const int lo_ = i < ip ? i : ip;  // entries (lo, lo + 1)
                    v -= omv[lo_];
                    if (lo_ <= 3) v -= 2.0 * oma[lo_];
                    if (lo_ >= 1) v -= 2.0 * oma[lo_ - 1];
                } else if (d == 2 || d == -2) {
                    const int lo_ = i < ip ? i : ip;
                    v += oma[lo_];
                }
                B_[e] = v;
            }
            __syncthreads();
            for (int e = tid; e < NZ * (NZ + 1) / 2; e += kT) {
                int i = (int)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
                while (i * (i + 1) / 2 > e) i--;
                while ((i + 1) * (i + 2) / 2 <= e) i++;
                const int j = e - i * (i + 1) / 2;
                // Two z variables meet in K only if they drive a common control point (different axes: same segment), neighbouring
                // segments of one axis (the 6 x 6 blocks through TB), or both reach a c5 / c0 of one axis (communication pairs):
                // everything else -- 70 % of the triangle at M = 10 -- is a structural zero and skips the 16 K-entry evaluations.
                const int ki = (i >= 2 * NZA) ? 2 : (i >= NZA ? 1 : 0), ai = i - ki * NZA, kj = (j >= 2 * NZA) ? 2 : (j >= NZA ? 1 : 0), aj = j - kj * NZA;
                const bool li_ = ES && ai == 3 * (M - 1), lj_ = ES && aj == 3 * (M - 1);
                const int mi = li_ ? M - 1 : ai / 3, mj = lj_ ? M - 1 : aj / 3, ji = li_ ? 2 : ai % 3, jj = lj_ ? 2 : aj % 3;
                const int dm = mi > mj ? mi - mj : mj - mi;
                const bool meet = (ki != kj) ? (dm == 0) : (dm <= 1 || (comm_on_k && ji == 2 && jj == 2));
                double v = 0.0;
                if (meet) {
                    const Drv Di = drive(ki, ai), Dj = drive(kj, aj);
                    // (compile-time indices: a private array indexed at run time lives in scratch memory -- a memory round trip per access)
#pragma unroll
                    for (int a = 0; a < 4; a++)
#pragma unroll
                        for (int b = 0; b < 4; b++) {
                            const double w = Di.w[a] * Dj.w[b];
                            if (w != 0.0) v += w * Kentry(Di.k, Di.cp[a], Dj.k, Dj.cp[b]);
                        }
                }
                H_[e] = (i == j) ? v + diag_shift : v;
            }
            __syncthreads();

            GEN_T(3);
            // ============ LDL^T in LDS, right-looking: after step j column j holds L[.][j], the diagonal d_j ======================
            // The factor is kept UNSCALED: after step j column j still holds U[j][i] = d_j L[i][j] (nothing of step j writes column j), so
            // the step needs no column copy and ONE barrier; the substitutions apply 1 / d_j.  Thread t owns the entries k = t mod 4
            // (mod 4) of rows t / 4 and t / 4 + 64 for the whole factorisation, and issues its loads four at a time (with a load per
            // fused multiply-add the update was a chain of LDS round trips: 330 k cycles per factorisation at nz = 90).
            bool pivot_bad = false;
            {
                const int rsel = tid >> 2, part = tid & 3;
                for (int j = 0; j < NZ; j++) {
                    const double d = H_[tri(j, j)];
                    if (!(d > 1e-300)) {  // uniform: every thread reads the same entry
                        pivot_bad = true;
                        break;
                    }
                    const double invd = rcp2(d);
                    if (tid == 0) dinv_[j] = invd;
                    for (int i = rsel; i < NZ; i += 64) {
                        if (i <= j) continue;
                        const double li = H_[tri(i, j)] * invd;
                        double* const hrow = &H_[tri(i, 0)];
                        int k = j + 1 + ((part - (j + 1)) & 3);  // first column > j of this thread's phase
                        for (; k + 12 <= i; k += 16) {
                            const double u0 = H_[tri(k, j)], u1 = H_[tri(k + 4, j)], u2 = H_[tri(k + 8, j)], u3 = H_[tri(k + 12, j)];
                            const double h0 = hrow[k], h1 = hrow[k + 4], h2 = hrow[k + 8], h3 = hrow[k + 12];
                            hrow[k] = fma(-li, u0, h0), hrow[k + 4] = fma(-li, u1, h1), hrow[k + 8] = fma(-li, u2, h2), hrow[k + 12] = fma(-li, u3, h3);
                        }
                        for (; k <= i; k += 4) hrow[k] = fma(-li, H_[tri(k, j)], hrow[k]);
                    }
                    __syncthreads();
                }
            }
            if (pivot_bad && shift_level < 2) {  // repeat this iteration with a (larger) diagonal shift: rounding lost the pivot, not the matrix
                shift_level++;
                flags |= LSCQP_INFO_SHIFTED;
                __syncthreads();
                repeated = true;
                it--;  // (the loop's increment makes it this iteration again)
                continue;
            }
            if (pivot_bad) {
                status = (near_cnt > 0 || floor_cnt > 0) ? LSCQP_STATUS_OPTIMAL : LSCQP_STATUS_NUMERIC;
                restore = status == LSCQP_STATUS_OPTIMAL;
                break;
            }
            // solve Hred x = rhs_ in place with the unscaled factor U[j][i] = d_j L[i][j] (stored at H_[tri(i, j)], i > j):
            //   forward   y_j = (b_j - sum_{i<j} U[i][j] y_i) / d_j            (L D y' = b with y = y')
            //   backward  x_j = y_j - (sum_{i>j} U[j][i] x_i) / d_j
            // ONE wavefront does it, lane t holding rows t and t + 64 in registers; the value of step j reaches the other lanes through a
            // v_readlane, not through LDS and a barrier (2 nz barriers per solve before: 70 k cycles at nz = 90).
            auto solve = [&]() {
                __syncthreads();
                if (tid < 64) {
                    const int r0 = tid, r1 = tid + 64;
                    double b0 = r0 < NZ ? rhs_[r0] : 0.0, b1 = r1 < NZ ? rhs_[r1] : 0.0;
                    const double d0 = r0 < NZ ? dinv_[r0] : 0.0, d1 = r1 < NZ ? dinv_[r1] : 0.0;
                    auto lane_value = [](double v, int src) -> double {  // (src is uniform: the loop counter)
                        const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
                        return __hiloint2double(hi, lo);
                    };
                    // (the factor entries of four steps are loaded before the four dependent broadcast + multiply-add steps run)
                    const double* const h0r = &H_[tri(r0 < NZ ? r0 : 0, 0)];
                    const double* const h1r = &H_[tri(r1 < NZ ? r1 : 0, 0)];
                    for (int j4 = 0; j4 < NZ; j4 += 4) {
                        double u0[4], u1[4];
#pragma unroll
                        for (int t = 0; t < 4; t++) {
                            const int j = j4 + t;
                            u0[t] = (r0 > j && r0 < NZ) ? h0r[j] : 0.0;
                            u1[t] = (r1 > j && r1 < NZ) ? h1r[j] : 0.0;
                        }
#pragma unroll
                        for (int t = 0; t < 4; t++) {
                            const int j = j4 + t;  // (steps past nz - 1 multiply zeros)
                            const double yj = (j < 64) ? lane_value(b0 * d0, j) : lane_value(b1 * d1, (j - 64) & 63);
                            b0 = fma(-u0[t], yj, b0);
                            b1 = fma(-u1[t], yj, b1);
                        }
                    }
                    // b now holds d_j y_j in its own row; y = b * dinv
                    const double y0 = b0 * d0, y1 = b1 * d1;
                    double s0 = 0.0, s1 = 0.0;  // sum_{i>j} U[j][i] x_i of the lane's rows
                    for (int j4 = ((NZ + 3) & ~3) - 4; j4 >= 0; j4 -= 4) {
                        double u0[4], u1[4];
#pragma unroll
                        for (int t = 0; t < 4; t++) {
                            const int j = j4 + 3 - t;
                            u0[t] = (j < NZ && r0 < j) ? H_[tri(j < NZ ? j : 0, r0)] : 0.0;
                            u1[t] = (j < NZ && r1 < j) ? H_[tri(j < NZ ? j : 0, r1 < NZ ? r1 : 0)] : 0.0;
                        }
#pragma unroll
                        for (int t = 0; t < 4; t++) {
                            const int j = j4 + 3 - t;
                            const double xj = (j < 64) ? lane_value(fma(-d0, s0, y0), j) : lane_value(fma(-d1, s1, y1), (j - 64) & 63);
                            s0 = fma(u0[t], xj, s0);
                            s1 = fma(u1[t], xj, s1);
                        }
                    }
                    if (r0 < NZ) rhs_[r0] = fma(-d0, s0, y0);
                    if (r1 < NZ) rhs_[r1] = fma(-d1, s1, y1);
                }
                __syncthreads();
            };

            GEN_T(4);
            // ============ predictor =============================================================================================
            solve();
            for (int zi = tid; zi < NZ; zi += kT) dz_[zi] = rhs_[zi];
            __syncthreads();
            expandT(dz_, dca_, false);
            __syncthreads();
            GEN_T(5);
            // ============ pass 2: affine step length, mu_aff, corrector right-hand side =========================================
            double rmax = 1.0, sB = 0, dmy = 0;
            // WEIGHT OF THE SECOND-ORDER TERM, rescue pass only (end of round 4).  Mehrotra's corrector carries ds_a dl_a, the error a FULL
            // affine step would leave.  When the affine step is blocked early (alpha_aff < LSCQP_SOC_GATE) that estimate is far too large
            // and the corrected direction can be worse than none: a DLSC instance (M = 10, 3-D, 23 neighbours, cold start;
            // tests/golden/limit_cycle_dlsc.npz, found by tools/stress_parity.py --dlsc --seed0 400) CYCLES with period four -- alpha
            // 0.89 / 0.46 / 0.16 / 0.84, sigma 1e-3 / 0.97 / 0.61 / 0.97, mu going UP sixfold on the blocked steps, gap 6e-6 .. 6e-5 --
            // to the iteration limit, before and after its one re-centring, on every kernel and in the numpy prototype alike.  With the
            // term weighted by alpha_aff in those iterations (what Colombo & Gondzio's weighted correctors reduce to without a line
            // search) it converges in 11.  The weight is applied only once the primal residual is below 1e-6 m: before that a short
            // affine step is the infeasible start's, not bad centring, and the weighted iteration stalls there (an M = 7 instance of
            // tests/test_gpu_parity.py went INFEASIBLE; tools/proto_corrector.py: the feasible-phase gate changes no iteration count
            // on the bench shapes).  In the compiled instances the same sweep costs the 64-QP headline 3 % (measured, A/B) for a
            // one-in-30 000 event, so it lives here, behind the rescue pass, and the first pass is bit-identical to what it was.
            const double soc_gate = rescue ? LSCQP_SOC_GATE : 0.0;
            double som = 1.0;  // weight of ds_a dl_a in the corrector (uniform over the workgroup)
#pragma nounroll
            for (int rep2 = 0;; rep2++) {
            rmax = 1.0, sB = 0;
#pragma unroll
            for (int u = 0; u < kR2; u++) {
                const int r = tid + kT * u;
                const double y = row_val(c_, u), dy = row_val(dca_, u);
                const double rpl = (y - t_lo[u]) - t_sl[u], rph = (t_hi[u] - y) - t_sh[u];
                const double isl = rcp2(t_sl[u]), ish = rcp2(t_sh[u]);
                const double tl = (dy + rpl) * isl, th = (rph - dy) * ish;
                const double rr = fmax(fmax(-tl, 1.0 + tl), fmax(-th, 1.0 + th));
                rmax = fmax(rmax, t_on[u] ? rr : 1.0);
                const double pl = -(dy + rpl) * t_ll[u] * (1.0 + tl), ph = -(rph - dy) * t_lh[u] * (1.0 + th);
                sB += t_on[u] ? (pl + ph) : 0.0;
                if (r < S.NOM) {
                    rv1_[r] = t_on[u] ? isl - ish : 0.0;
                    rv2_[r] = t_on[u] ? (-som * pl - t_ll[u] * rpl) * isl - (-som * ph - t_lh[u] * rph) * ish : 0.0;
                }
            }
            {
                double b10 = 0, b11 = 0, b12 = 0, b20 = 0, b21 = 0, b22 = 0;
                const double cx = c_[lx], cy = c_[P + lx], cz = (DIM == 3) ? c_[2 * P + lx] : 0.0;
                const double dx = dca_[lx], dy = dca_[P + lx], dzz = (DIM == 3) ? dca_[2 * P + lx] : 0.0;
                for (int o = lg; o < o_end; o += G) {
                    const Row R = load_row(o);
                    const double s = Rs_[o * CP + lcp], l = Rl_[o * CP + lcp];
                    const double rp = (R.nx * cx + R.ny * cy + R.nz * cz - R.b) - s;
                    const double is = rcp2(s);
                    const double ds = (R.nx * dx + R.ny * dy + R.nz * dzz) + rp;
                    const double t = ds * is;
                    rmax = fmax(rmax, (l > 0.0) ? fmax(-t, 1.0 + t) : 1.0);
                    const double pa = -ds * l * (1.0 + t);
                    sB += pa;
                    const double t2 = (-som * pa - l * rp) * is;
                    b10 += is * R.nx, b11 += is * R.ny, b12 += is * R.nz;
                    b20 += t2 * R.nx, b21 += t2 * R.ny, b22 += t2 * R.nz;
                }
                // partial sums per (group, control point), combined in group order by the threads of group 0.  The factor occupies H_
                // during this pass, so the shares go through the reduction scratch (3 x 256 doubles): two rounds of three values.
                __syncthreads();
                for (int round = 0; round < 2; round++) {
                    if (ll) {
                        red_[tid] = round == 0 ? b10 : b20;
                        red_[kT + tid] = round == 0 ? b11 : b21;
                        red_[2 * kT + tid] = round == 0 ? b12 : b22;
                    }
                    __syncthreads();
                    if (tid < CP) {
                        double a0 = 0, a1 = 0, a2 = 0;
                        for (int g = 0; g < G; g++) a0 += red_[g * CP + tid], a1 += red_[kT + g * CP + tid], a2 += red_[2 * kT + g * CP + tid];
                        LX_[(3 * round + 0) * CP + tid] = a0, LX_[(3 * round + 1) * CP + tid] = a1, LX_[(3 * round + 2) * CP + tid] = a2;
                    }
                    __syncthreads();
                }
            }
            block_reduce3(rmax, sB, dmy, true, false, false);
            if (rep2 == 1 || !(rmax * soc_gate > 1.0 && max_rp <= 1e-6)) break;  // uniform; the repetition rewrites rv2_ / LX_ and reproduces rmax, sB
            som = rcp2(rmax);
            __syncthreads();
            }
            for (int e = tid; e < NX; e += kT) {
                const int k = e / P, cp = e % P;
                X1[e] = gather_rows(rv1_, e) + (cp >= 3 ? LX_[k * CP + cp - 3] : 0.0);
                X2[e] = gather_rows(rv2_, e) + (cp >= 3 ? LX_[(3 + k) * CP + cp - 3] : 0.0);
            }
            const double a_aff = rcp2(rmax);
            const double mu_aff = ((1.0 - a_aff) * sum_sl + a_aff * a_aff * sB) * inv_m;
            double sigma = fmax(mu_aff, 0.0) / mu;
            sigma = sigma * sigma * sigma;
            const double smu = fmax(sigma * mu, LSCQP_SMU_FLOOR * tol * (1.0 + obj_abs) * inv_m);  // (lscqp_kernel.hpp: never aim below the gap target)
            __syncthreads();
            GEN_T(6);
            // ============ corrector solve ======================================================================================
            for (int zi = tid; zi < NZ; zi += kT) rhs_[zi] = -gc_[zi] + smu * gatherT(X1, zi) + gatherT(X2, zi);
            __syncthreads();
            solve();
            for (int zi = tid; zi < NZ; zi += kT) dz_[zi] = rhs_[zi];
            __syncthreads();
            expandT(dz_, dc_, false);
            __syncthreads();
            GEN_T(7);
            // ============ pass 3: directions of the row state, step length ======================================================
            rmax = 0.0;
            double sdl = 0, sdd = 0;
            double t_ds[2 * kR2], t_dl[2 * kR2];
#pragma unroll
            for (int u = 0; u < kR2; u++) {
                const bool on = t_on[u];
                const double y = row_val(c_, u), dya = row_val(dca_, u), dyc = row_val(dc_, u);
                const double rpl = (y - t_lo[u]) - t_sl[u], rph = (t_hi[u] - y) - t_sh[u];
                const double isl = rcp2(t_sl[u]), ish = rcp2(t_sh[u]);
                const double tl = (dya + rpl) * isl, th = (rph - dya) * ish;
                const double pl = -(dya + rpl) * t_ll[u] * (1.0 + tl), ph = -(rph - dya) * t_lh[u] * (1.0 + th);
                const double dsl = dyc + rpl, dsh = rph - dyc;
                const double dll = (smu - som * pl) * isl - t_ll[u] - t_ll[u] * isl * dsl;
                const double dlh = (smu - som * ph) * ish - t_lh[u] - t_lh[u] * ish * dsh;
                const double ill = rcp2(on ? t_ll[u] : 1.0), ilh = rcp2(on ? t_lh[u] : 1.0);
                const double rr = fmax(fmax(-dsl * isl, -dsh * ish), fmax(-dll * ill, -dlh * ilh));
                rmax = fmax(rmax, on ? rr : 0.0);
                t_ds[2 * u] = on ? dsl : 0.0, t_ds[2 * u + 1] = on ? dsh : 0.0;
                t_dl[2 * u] = on ? dll : 0.0, t_dl[2 * u + 1] = on ? dlh : 0.0;
                sdl += t_sl[u] * t_dl[2 * u] + t_ll[u] * t_ds[2 * u] + t_sh[u] * t_dl[2 * u + 1] + t_lh[u] * t_ds[2 * u + 1];
                sdd += t_ds[2 * u] * t_dl[2 * u] + t_ds[2 * u + 1] * t_dl[2 * u + 1];
            }
            // The LSC rows' state lives in LDS and is only rewritten once the step length is final; their directions are recomputed
            // from (c, dc_aff, dc) wherever they are needed (no room to keep them): for the ratio test, per trial of the centrality
            // loop, and for the update.
            struct Dir {
                double s, l, ds, dl;
            };
            const double cx3 = c_[lx], cy3 = c_[P + lx], cz3 = (DIM == 3) ? c_[2 * P + lx] : 0.0;
            const double ax3 = dca_[lx], ay3 = dca_[P + lx], az3 = (DIM == 3) ? dca_[2 * P + lx] : 0.0;
            const double dx3 = dc_[lx], dy3 = dc_[P + lx], dz3 = (DIM == 3) ? dc_[2 * P + lx] : 0.0;
            auto lsc_dir = [&](int o) -> Dir {
                const Row R = load_row(o);
                Dir D;
                D.s = Rs_[o * CP + lcp], D.l = Rl_[o * CP + lcp];
                const double rp = (R.nx * cx3 + R.ny * cy3 + R.nz * cz3 - R.b) - D.s;
                const double is = rcp2(D.s);
                const double dsa = (R.nx * ax3 + R.ny * ay3 + R.nz * az3) + rp;
                const double pa = -dsa * D.l * (1.0 + dsa * is);
                const double ds = (R.nx * dx3 + R.ny * dy3 + R.nz * dz3) + rp;
                const bool act = D.l > 0.0;
                D.dl = act ? ((smu - som * pa) * is - D.l - D.l * is * ds) : 0.0;
                D.ds = act ? ds : 0.0;
                return D;
            };
            for (int o = lg; o < o_end; o += G) {
                const Dir D = lsc_dir(o);
                if (D.l > 0.0) rmax = fmax(rmax, fmax(-D.ds * rcp2(D.s), -D.dl * rcp2(D.l)));
                sdl += D.s * D.dl + D.l * D.ds;
                sdd += D.ds * D.dl;
            }
            block_reduce3(sdl, sdd, rmax, false, false, true);
            const double tau = (sigma < 1e-4) ? fmax(0.9995, 1.0 - mu) : 0.9995;
            const double irmax = rcp2(rmax);
            const double alpha_std = (rmax > 0.9995) ? 0.9995 * irmax : 1.0;
            double alpha = (rmax > tau) ? tau * irmax : 1.0;
            double applied = 0.0;
            for (int bt = 0;; bt++) {  // centrality safeguard (see lscqp_kernel.hpp): no product below GAMMA * mu(alpha)
                const double mu_a = (sum_sl + alpha * (sdl + alpha * sdd)) * inv_m;
                const double delta = alpha - applied;
                applied = alpha;
                double pmin = 1e300;
#pragma unroll
                for (int u = 0; u < kR2; u++) {
                    t_sl[u] = fma(delta, t_ds[2 * u], t_sl[u]), t_ll[u] = fma(delta, t_dl[2 * u], t_ll[u]);
                    t_sh[u] = fma(delta, t_ds[2 * u + 1], t_sh[u]), t_lh[u] = fma(delta, t_dl[2 * u + 1], t_lh[u]);
                    pmin = fmin(pmin, t_on[u] ? fmin(t_sl[u] * t_ll[u], t_sh[u] * t_lh[u]) : 1e300);
                }
                for (int o = lg; o < o_end; o += G) {
                    const Dir D = lsc_dir(o);
                    const double ns = fma(alpha, D.ds, D.s), nl = fma(alpha, D.dl, D.l);
                    pmin = fmin(pmin, (nl > 0.0) ? ns * nl : 1e300);
                }
                pmin = -block_max(-pmin);
                if (pmin >= LSCQP_CENTRALITY_GAMMA * mu_a || bt == 9) break;
                alpha = (bt == 0 && alpha_std < alpha) ? alpha_std : 0.7 * alpha;
            }
            for (int o = lg; o < o_end; o += G) {  // (each thread rewrites only its own rows: no barrier needed before the next pass)
                const Dir D = lsc_dir(o);
                Rs_[o * CP + lcp] = fma(alpha, D.ds, D.s);
                Rl_[o * CP + lcp] = fma(alpha, D.dl, D.l);
            }
            GEN_T(8);
            // ============ update of z and the control points ====================================================================
            if (it == 0) alpha_first = (float)alpha;
            for (int zi = tid; zi < NZ; zi += kT) z_[zi] += alpha * dz_[zi];
            __syncthreads();
            expandT(z_, c_, true);
            __syncthreads();
            if (!(alpha > 1e-12) || !(mu == mu)) {
                status = (near_cnt > 0 || floor_cnt > 0) ? LSCQP_STATUS_OPTIMAL : LSCQP_STATUS_NUMERIC;
                restore = status == LSCQP_STATUS_OPTIMAL;
                break;
            }
            GEN_T(9);
        }
    if (status == LSCQP_STATUS_ITER_LIMIT && (near_cnt > 0 || floor_cnt > 0)) {
        status = LSCQP_STATUS_OPTIMAL;
        restore = true;
    }
    if (status == LSCQP_STATUS_ITER_LIMIT && res_p > 1e-6) status = LSCQP_STATUS_INFEASIBLE;
    if (status == LSCQP_STATUS_NUMERIC && res_p > 1e-6) status = LSCQP_STATUS_INFEASIBLE;
    if (restore) {
        __syncthreads();
        for (int zi = tid; zi < NZ; zi += kT) z_[zi] = zs_[zi];
        __syncthreads();
        expandT(z_, c_, true);
        __syncthreads();
        res_p = snap_p, res_d = snap_d, res_gap = snap_gap;
        flags |= LSCQP_INFO_REMEMBERED;
        if (!(snap_d <= 1e-8 && snap_gap <= tol && snap_p <= 1e-9)) flags |= LSCQP_INFO_FLOOR_ACCEPTED;  // (lscqp_kernel.hpp: the stated deviation only)
    }
    if (recentred || net_done) flags |= LSCQP_INFO_RECENTRED;
    __syncthreads();
    const double obj = objective(true);
    for (int e = tid; e < NX; e += kT) x_out[q * NX + e] = c_[e] + Hd->p0[e / P];
    if (tid == 0) {
        obj_out[q] = obj;
        status_out[q] = status;
        if (info_out) {
            info_out[q].iterations = it + it_before;
            info_out[q].flags = flags;
            info_out[q].res_primal = res_p;
            info_out[q].res_dual = res_d;
            info_out[q].gap = res_gap;
        }
    }
}

}  // namespace lscqp_generic

// what lscqp_api.hip needs to know about the run-time-shaped instance
extern "C" int lscqp_generic_supports(int M, int dim, int es) {
    if (M < 2 || M > lscqp_generic::kMaxM || dim < 2 || dim > 3) return 0;
    const lscqp_generic::Shape s = lscqp_generic::Shape::make(M, dim, es, 0);
    if ((s.NOM + lscqp_generic::kT - 1) / lscqp_generic::kT > lscqp_generic::kR2) return 0;
    return sizeof(double) * (size_t)s.total <= lscqp::kMaxLdsBytes ? 1 : 0;
}
extern "C" size_t lscqp_generic_lds_bytes(int M, int dim, int es, int n_obs_max) {
    return sizeof(double) * (size_t)lscqp_generic::Shape::make(M, dim, es, n_obs_max).total;
}
extern "C" int lscqp_generic_max_obstacles(int M, int dim, int es) {  // what the CU's LDS leaves for the row state (16 bytes per row)
    if (!lscqp_generic_supports(M, dim, es)) return 0;
    const lscqp_generic::Shape s = lscqp_generic::Shape::make(M, dim, es, 0);
    const size_t left = lscqp::kMaxLdsBytes - sizeof(double) * (size_t)s.total;
    int cap = (int)(left / (16 * (size_t)s.CP));
    while (cap > 0 && lscqp_generic_lds_bytes(M, dim, es, cap) > lscqp::kMaxLdsBytes) cap--;  // (the carve rounds to even offsets)
    return cap;
}

extern "C" hipError_t lscqp_launch_generic(const lscqp::DevClass* cls, int M, int dim, int es, int64_t n, const lscqp_header* hdr, const lscqp_row* rows,
                                           const uint64_t* row_offsets, const lscqp_box* sfc, const double* x_init, double* x_out, double* obj_out,
                                           int32_t* status_out, lscqp_info* info_out, hipStream_t stream) {
    if (!lscqp_generic_supports(M, dim, es) || cls->n_obs_max > lscqp_generic_max_obstacles(M, dim, es)) return hipErrorInvalidValue;
    const size_t lds = lscqp_generic_lds_bytes(M, dim, es, cls->n_obs_max);
    static std::atomic<bool> attr_set[64];  // (zero-initialised; two host threads may launch for the first time at once: setting it twice is harmless)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lscqp_generic::pdip_generic_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)lscqp::kMaxLdsBytes);
        if (e != hipSuccess) return e;
        attr_set[dev].store(true, std::memory_order_release);
    }
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(lscqp_generic::pdip_generic_kernel, dim3((unsigned)n), dim3(lscqp_generic::kT), lds, stream, *cls, M, dim, es, n, hdr, rows,
                       row_offsets, sfc, x_init, x_out, obj_out, status_out, info_out);
    return hipGetLastError();
}

#ifdef LSCQP_GEN_TIMING
extern "C" int lscqp_generic_cycles(unsigned long long* out, int reset) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(lscqp_generic::gen_cycles), sizeof(unsigned long long) * 16);
    if (reset) {
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(lscqp_generic::gen_cycles), z, sizeof z);
    }
    return 0;
}
#endif
