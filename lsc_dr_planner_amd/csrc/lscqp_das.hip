// lscqp_das.hip — the DUAL ACTIVE SET phase of the batched trajectory-QP solver (round 5), gfx950 only.
//
// Why it exists.  The QP of TrajOptimizer::populatebyrow (reference src/traj_optimizer.cpp:216-514) has a CONSTANT Hessian: the jerk
// cost and the terminal pull depend on the class (dt, weights) and on the number of terminal segments only -- never on the agent's
// neighbours.  And a plan's optimum holds very few of its ~1000 rows: on the bench's own batches 61 of the 64 headline QPs
// (BASELINE configs[1]) have NO active row at all -- the optimum is the unconstrained minimiser -- and the other three hold one; the
// dense-maze class (configs[2]) holds <= 4, the 1024 x M10 x 40 class (configs[3]) <= 5 (tools/proto_gi.py, tools/proto_das.py,
// profiles/r05_proto_active_set.txt).  An interior-point method pays 3-13 full iterations (row passes over every row, assembly and
// LDL^T of the reduced system, two substitutions) to find that out.  The dual active-set method of Goldfarb and Idnani starts AT the
// unconstrained minimiser and adds violated rows one at a time:
//
//     min 1/2 c'Hx c + fx'c   over control points  c = cfix + T z  (the equality rows, eliminated as in lscqp_kernel.hpp),   a_i'c >= h_i
//     C = T (T'Hx T)^-1 T'    the COMPLIANCE of the plan: the displacement of every control point per unit multiplier on one of them.
//                             One symmetric P x P table per number of terminal segments, the same for every axis, built on the host
//                             in extended precision when the class is created (lscqp_das_build_tables) -- 7 KB at M = 5.
//     unconstrained optimum   c_u[k] = cfix[k] - c1_k U1 - c2_k U2 + 2 w_t goal_k G1      (three table vectors: no factorisation)
//     one step for row p      w_p = C a_p;   r = S^-1 A'w_p  (S = A'W, the active rows' small Gram matrix, Cholesky in LDS);
//                             dc = w_p - W r;   t = min( min_{r_j > 0} u_j / r_j ,  -slack_p / a_p'dc );   c += t dc,  u -= t r,  u_p += t
//                             t = the second: p joins the active set;  t = the first: row j leaves it and the step is repeated.
//
// The work per QP is one pass over the rows per step (the rows are read where they lie -- HBM the first time, L2 afterwards; nothing
// is staged) plus a handful of short vector operations: the phase is bound by memory latency and, for large batches, by HBM bandwidth
// -- the roofline north_star names.  What it returns is a KKT point of the reference's model: primal violation <= 1e-9 m on EVERY row
// (the last pass), multipliers >= 0, exact complementarity, and the reduced stationarity residual verified against the same scale the
// interior-point kernel uses (<= 1e-9) -- after a final "polish" that rebuilds the point from its multipliers and refines them once.
// An instance the phase does not finish (more active rows than its budget, more steps than its budget, a dependent active set, an
// infeasible row system, a failed verification) is LEFT to the interior-point kernel, which runs behind it over the same batch in
// "first pass after the active-set phase" mode (cls.repair == 3) and skips what is already OPTIMAL.  Nothing here is a CPU fallback and
// nothing is approximate: both methods return the optimum of the same strictly convex QP.
//
// Organisation: one workgroup (64 .. 256 threads) per QP, M / dim / end stop / n_obs are run-time values (one kernel for every class);
// row ids:  [LSC rows o*P + cp | interval lo/hi per (axis, cp) | velocity lo/hi | acceleration lo/hi | communication pairs lo/hi],
// selection = the most violated row (slack / |a|), lowest id on ties: results are reproducible bit for bit from run to run.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <cmath>
#include <mutex>
#include <vector>

#include "lscqp_kernel.hpp"  // DevClass, KQ
#include "lscqp_launch.hpp"

namespace lscqp_das {

using lscqp::DevClass;
using lscqp::KQ;

constexpr double kTolP = 1e-9;      // a row is violated below -1e-9 (normalised): the interior-point kernel's primal bar
constexpr double kTolD = 1e-9;      // accepted stationarity (scaled like lscqp_info.res_dual)
constexpr int kMaxK = 32;           // active rows the phase can hold (lanes of one wavefront own the rows of the small factor)

// ---- tables, per number of terminal segments ts = 1 .. M:  [U1 (P) | U2 (P) | G1 (P) | C (P x P, symmetric)] --------------------------
__host__ __device__ inline size_t table_stride(int M) { return (size_t)(3 + 6 * M) * (size_t)(6 * M); }

struct Layout {  // LDS carve of one QP, in doubles
    int P, NX, kmax;
    int o_hdr, o_sfc, o_c, o_cu, o_dc, o_wp, o_lo, o_hi, o_g, o_W, o_L, o_u, o_r, o_v, o_y, o_linv, o_arhs, o_acoef, o_red, o_sc, o_C, o_int, o_rows, n_stage, total;
    __host__ __device__ static Layout make(int M, int dim, int kmax, int cacheC, int stage_rows = 0) {
        Layout s;
        s.P = 6 * M, s.NX = dim * s.P, s.kmax = kmax;
        int o = 0;
        auto take = [&](int n) { const int at = o; o += (n + 1) & ~1; return at; };
        s.o_hdr = take(32);
        s.o_sfc = take(6 * M);
        s.o_c = take(s.NX), s.o_cu = take(s.NX), s.o_dc = take(s.NX), s.o_wp = take(s.NX), s.o_lo = take(s.NX), s.o_hi = take(s.NX), s.o_g = take(3 * s.NX);
        s.o_W = take(kmax * s.NX);
        s.o_L = take(kmax * (kmax + 1));
        s.o_u = take(kmax), s.o_r = take(kmax), s.o_v = take(kmax), s.o_y = take(kmax), s.o_linv = take(kmax), s.o_arhs = take(kmax + 1);
        s.o_acoef = take(3 * (kmax + 1));
        s.o_red = take(32);
        s.o_sc = take(16);  // org[3], goal[3], vmax dt/n [3], amax dt^2/(n(n-1)) [3]
        s.o_C = take(cacheC ? s.P * s.P : 0);
        s.o_int = take(4 * (kmax + 1) + 16);  // ints: per active row {id, idx0, idx1, idx2} (+ the candidate), control words
        s.n_stage = stage_rows;  // LSC rows of the instance kept in LDS after the first pass (SoA nx | ny | nz | b), 0: re-read from L2
        s.o_rows = take(4 * stage_rows);
        s.total = o;
        return s;
    }
};

// Wave reductions on the DPP network (lscqp_kernel.hpp: four row_shr steps, two row broadcasts, one v_readlane pair -- ~150 cycles per
// value against ~600 for a ds_bpermute butterfly on fp64).
__device__ __forceinline__ double wave_max(double v) { return lscqp::wave_max(v); }
__device__ __forceinline__ double wave_min(double v) { return -lscqp::wave_max(-v); }
__device__ __forceinline__ double wave_sum(double v) { return lscqp::wave_sum(v); }
__device__ __forceinline__ void wave_argmin(double& v, int& id) {  // lexicographic (value, id): every lane ends with the result
    const double vm = wave_min(v);
    const double cand = (v == vm) ? (double)id : 2147483647.0;  // (ids are < 2^31: exact in fp64)
    id = (int)wave_min(cand);
    v = vm;
}
// a / b for small non-negative integers (a < 2^20, b <= 2^10) through one fp32 multiplication: exact, and a handful of instructions where an
// integer division by a run-time value costs ~40
__device__ __forceinline__ int fdiv(int a, float inv_b) { return (int)(((float)a + 0.5f) * inv_b); }

// LDS hand-overs between the lanes of ONE wavefront (the small factor): LDS operations of a wavefront execute in order, the fence keeps
// the compiler from moving them across
#define LSCQP_DAS_WAVE_SYNC()                                   \
    do {                                                        \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  \
        __builtin_amdgcn_wave_barrier();                        \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  \
    } while (0)

// Development aid: per-phase cycle totals, compiled in only with -DLSCQP_DAS_TIMING (tools/das_timing.py)
#ifdef LSCQP_DAS_TIMING
__device__ unsigned long long das_cycles[16];
#define DAS_T(slot)                                                        \
    do {                                                                   \
        const unsigned long long now_ = __builtin_readcyclecounter();      \
        if (tid == 0) atomicAdd(&das_cycles[slot], now_ - tprev_);          \
        tprev_ = now_;                                                     \
    } while (0)
#else
#define DAS_T(slot) \
    do {            \
    } while (0)
#endif

// One row of the model as (<= 3 control-point entries, right-hand side): a'c >= h.
struct Row {
    int idx[3];
    double coef[3];
    double rhs;
};

__global__ __launch_bounds__(256) void das_kernel(DevClass cls, int M, int dim, int es, int cap, int kmax, int max_steps, int cacheC, int stage_rows,
                                                  const double* __restrict__ tab, int64_t n, const lscqp_header* __restrict__ hdr,
                                                  const lscqp_row* __restrict__ rows, const uint64_t* __restrict__ row_offsets,
                                                  const lscqp_box* __restrict__ sfc, const double* __restrict__ x_init, double* __restrict__ x_out,
                                                  double* __restrict__ obj_out, int32_t* __restrict__ status_out, lscqp_info* __restrict__ info_out) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int64_t k0 = blockIdx.x;
    if (k0 >= n) return;
    const int64_t q = cls.order ? (int64_t)cls.order[k0] : k0;
    const int tid = threadIdx.x, T = blockDim.x, lane = tid & 63, wv = tid >> 6, NW = T >> 6;
#ifdef LSCQP_DAS_TIMING
    unsigned long long tprev_ = __builtin_readcyclecounter();
#endif
    const Layout L = Layout::make(M, dim, kmax, cacheC, stage_rows);
    const int P = L.P, NX = L.NX;
    double* const H_ = smem + L.o_hdr;
    double* const sfc_ = smem + L.o_sfc;
    double* const c_ = smem + L.o_c;
    double* const cu_ = smem + L.o_cu;
    double* const dc_ = smem + L.o_dc;
    double* const wp_ = smem + L.o_wp;
    double* const lo_ = smem + L.o_lo;
    double* const hi_ = smem + L.o_hi;
    double* const g_ = smem + L.o_g;
    double* const W_ = smem + L.o_W;
    double* const Lm_ = smem + L.o_L;  // [kmax][kmax + 1] lower Cholesky factor of S = A'W
    double* const u_ = smem + L.o_u;
    double* const r_ = smem + L.o_r;
    double* const v_ = smem + L.o_v;
    double* const y_ = smem + L.o_y;
    double* const linv_ = smem + L.o_linv;
    double* const arhs_ = smem + L.o_arhs;    // [kmax + 1]: slot kmax = the candidate row
    double* const acoef_ = smem + L.o_acoef;  // [kmax + 1][3]
    double* const red_ = smem + L.o_red;
    double* const Cc_ = smem + L.o_C;
    int* const aint_ = reinterpret_cast<int*>(smem + L.o_int);  // [kmax + 1][4]: id, idx0, idx1, idx2
    int* const ctl_ = aint_ + 4 * (kmax + 1);                   // control words shared by the workgroup
    double* const Sx_ = smem + L.o_rows;                        // staged LSC rows: [nx | ny | nz | b] x stage_rows
    double* const Sy_ = Sx_ + stage_rows;
    double* const Sz_ = Sy_ + stage_rows;
    double* const Sb_ = Sz_ + stage_rows;
    const int LDL = kmax + 1;

    // ---- header, corridor boxes (and the instance's row offset: one memory round trip for all three) -------------------------------
    const uint64_t roff = row_offsets ? row_offsets[q] : 0;
    {
        const double* hsrc = reinterpret_cast<const double*>(hdr + q);
        const double* ssrc = reinterpret_cast<const double*>(sfc) + q * 6 * M;
        for (int e = tid; e < 32 + (cls.use_sfc ? 6 * M : 0); e += T) (e < 32 ? H_[e] : sfc_[e - 32]) = e < 32 ? hsrc[e] : ssrc[e - 32];
    }
    __syncthreads();
    DAS_T(0);  // header, boxes, row offset
    const lscqp_header* Hd = reinterpret_cast<const lscqp_header*>(H_);
    const lscqp_box* sfcl = reinterpret_cast<const lscqp_box*>(sfc_);
    const int n_obs = Hd->n_obs;
    // Handing an instance over: the interior-point kernel behind this phase solves whatever is not OPTIMAL (cls.repair == 3 there).
    auto hand_over = [&](int steps) {
        for (int e = tid; e < NX; e += T) x_out[q * NX + e] = x_init ? x_init[q * NX + e] : Hd->p0[e / P];
        if (tid == 0) {
            obj_out[q] = 0.0;
            status_out[q] = LSCQP_STATUS_ITER_LIMIT;
            if (info_out) {
                info_out[q].iterations = 0;
                info_out[q].flags = 0;
                info_out[q].res_primal = info_out[q].res_dual = 0.0;
                info_out[q].gap = (double)steps;  // (overwritten by the pass that solves the instance)
            }
        }
    };
    if (n_obs > cap || n_obs < 0) {  // the kernel instance behind this phase refuses it (LSCQP_STATUS_CAPACITY): its verdict, not ours
        hand_over(0);
        return;
    }
    const double dt = cls.dt;
    // per-axis scalars live in LDS: indexed with a run-time axis, a register array would be materialised in scratch memory (and a kernel
    // with a private segment costs tens of microseconds to launch)
    double* const org = smem + L.o_sc;
    double* const goal = org + 3;
    double* const Vk = org + 6;
    double* const Ak = org + 9;
    if (tid < 3) {
        org[tid] = Hd->p0[tid];
        goal[tid] = Hd->goal[tid] - Hd->p0[tid];
        Vk[tid] = Hd->vmax[tid] * dt * 0.2;
        Ak[tid] = Hd->amax[tid] * dt * dt * 0.05;
    }
    __syncthreads();
    int ts = Hd->terminal_segments;
    if (ts <= 0) {  // src/traj_optimizer.cpp:530-538 in fp64 (as lscqp_kernel.hpp)
        const double d2 = goal[0] * goal[0] + goal[1] * goal[1] + goal[2] * goal[2];
        ts = (int)((M * dt - sqrt(d2) / Hd->nominal_velocity + 1e-9) / dt);
        if (ts < 1) ts = 1;
    }
    if (ts > M) ts = M;
    const double q2s = cls.q2s, wt2 = 2.0 * cls.w_t;
    const double* const tb = tab + (size_t)(ts - 1) * table_stride(M);
    const double* const U1 = tb, * const U2 = tb + P, * const G1 = tb + 2 * P, * const Cg = tb + 3 * P;
    const bool comm_on = cls.comm_range > 0;
    const double rho_pair = 0.5 * cls.comm_range - Hd->radius;  // :484
    const double rho_wp = 0.5 * cls.comm_range - 1e-5;          // :495

    // ---- merged intervals (world box, corridor, communication rows on c[m][5]: as lscqp_kernel.hpp), unconstrained optimum --------
    bool empty = false;
    for (int e = tid; e < NX; e += T) {
        const int k = e / P, cp = e % P, m = cp / 6;
        const double ok_ = org[k];
        double lo = cls.world_min[k] - ok_, hi = cls.world_max[k] - ok_;  // :252-253,260-265
        if (cls.rsfc && k == 2 && m == 0) {                               // :255-258
            lo = -100.0 - ok_;
            hi = 100.0 - ok_;
        }
        if (cls.use_sfc) {  // :372-397
            lo = fmax(lo, sfcl[m].bmin[k] - ok_);
            hi = fmin(hi, sfcl[m].bmax[k] - ok_);
        }
        if (comm_on && cp % 6 == 5) {  // pairs (m, mi = 0) :482-487 and waypoint rows :494-497
            const double wpk = Hd->next_waypoint[k] - ok_;
            lo = fmax(lo, fmax(-rho_pair, wpk - rho_wp));
            hi = fmin(hi, fmin(rho_pair, wpk + rho_wp));
        }
        lo_[e] = lo;
        hi_[e] = hi;
        if (cp >= 3 && lo > hi) empty = true;
        const double c1 = Hd->v0[k] * dt * 0.2;
        const double c2 = Hd->a0[k] * dt * dt * 0.05 + 2.0 * c1;
        const double fixv = (cp == 1) ? c1 : (cp == 2) ? c2 : 0.0;
        const double cv = fixv - c1 * U1[cp] - c2 * U2[cp] + wt2 * goal[k] * G1[cp];
        c_[e] = cv;
        cu_[e] = cv;
    }
    if (tid == 0) ctl_[0] = 0;
    __syncthreads();
    if (empty) ctl_[0] = 1;  // (benign race: every writer stores 1)
    __syncthreads();
    if (ctl_[0]) {  // an empty interval: INFEASIBLE is the interior-point kernel's verdict to give
        hand_over(0);
        return;
    }

    DAS_T(1);  // intervals, unconstrained optimum (table vectors)
    // ---- row ids ------------------------------------------------------------------------------------------------------------------
    const int nL = n_obs * P;
    const int NCP = M * (M - 1) / 2;
    const int oB = nL, oV = oB + 2 * NX, oA = oV + 2 * dim * 5 * M, oC = oA + 2 * dim * 4 * M, nAll = oC + (comm_on ? 2 * dim * NCP : 0);
    const float iP = 1.0f / (float)P, i5M = 1.0f / (float)(5 * M), i4M = 1.0f / (float)(4 * M), iNCP = 1.0f / (float)(NCP > 0 ? NCP : 1);
    const bool staged = stage_rows > 0 && nL <= stage_rows;  // (uniform)
    bool rows_in_lds = false;                                // set after the first pass
    auto load_row = [&](int j, double& nx, double& ny, double& nz, double& b) -> bool {  // LSC row j of this instance, translated; false: dropped
        if (rows_in_lds) {  // (uniform) staged by the first pass: dropped rows hold (0, 0, 0 | -1)
            nx = Sx_[j], ny = Sy_[j], nz = Sz_[j], b = Sb_[j];
            return b != -1.0 || nx != 0.0 || ny != 0.0 || nz != 0.0;
        }
        double x, y, z, w;
        if (cls.rows_f32) {
            const float4 f = reinterpret_cast<const float4*>(rows)[roff + (uint64_t)j];
            x = f.x, y = f.y, z = f.z, w = f.w;
        } else {
            const double4 d = *reinterpret_cast<const double4*>(&rows[roff + (uint64_t)j]);
            x = d.x, y = d.y, z = d.z, w = d.w;
        }
        nx = x, ny = y, nz = (dim == 3) ? z : 0.0;
        b = w - (x * org[0] + y * org[1] + (dim == 3 ? z * org[2] : 0.0));
        return !(sqrt(x * x + y * y + z * z) < 1e-5) && (j - P * fdiv(j, iP)) >= 3;  // dropped like the reference does (:409-411, :404-406)
    };
    // the row with id `rid` as entries (uniform over the workgroup); false: the row does not exist
    auto decode = [&](int rid, Row& R) -> bool {
        R.idx[0] = R.idx[1] = R.idx[2] = 0;
        R.coef[0] = R.coef[1] = R.coef[2] = 0.0;
        if (rid < nL) {
            const int cp = rid - P * fdiv(rid, iP);
            double nx, ny, nz, b;
            const bool ok = load_row(rid, nx, ny, nz, b);
            R.idx[0] = cp, R.idx[1] = P + cp, R.idx[2] = (dim == 3 ? 2 * P : 0) + cp;
            R.coef[0] = nx, R.coef[1] = ny, R.coef[2] = (dim == 3) ? nz : 0.0;
            R.rhs = b;
            return ok;
        }
        if (rid < oV) {
            const int s = rid - oB, e = s >> 1;
            R.idx[0] = e;
            R.coef[0] = (s & 1) ? -1.0 : 1.0;
            R.rhs = (s & 1) ? -hi_[e] : lo_[e];
            return (e - P * fdiv(e, iP)) >= 3;
        }
        if (rid < oA) {
            const int s = rid - oV, r = s >> 1, k = fdiv(r, i5M), rr = r - k * 5 * M, m = rr / 5, i = rr % 5, e = k * P + 6 * m + i;
            const double sg = (s & 1) ? -1.0 : 1.0;
            R.idx[0] = e + 1, R.idx[1] = e;
            R.coef[0] = sg, R.coef[1] = -sg;
            R.rhs = -Vk[k];
            return !(m == 0 && i < 2);
        }
        if (rid < oC) {
            const int s = rid - oA, r = s >> 1, k = fdiv(r, i4M), rr = r - k * 4 * M, m = rr / 4, i = rr % 4, e = k * P + 6 * m + i;
            const double sg = (s & 1) ? -1.0 : 1.0;
            R.idx[0] = e + 2, R.idx[1] = e + 1, R.idx[2] = e;
            R.coef[0] = sg, R.coef[1] = -2.0 * sg, R.coef[2] = sg;
            R.rhs = -Ak[k];
            return !(m == 0 && i < 1);
        }
        {
            const int s = rid - oC, r = s >> 1, k = fdiv(r, iNCP), ci = r - k * NCP;
            int uu = 1;
            while (uu * (uu + 1) / 2 <= ci) uu++;
            const int up = ci - uu * (uu - 1) / 2;
            const double sg = (s & 1) ? -1.0 : 1.0;
            R.idx[0] = k * P + 6 * uu + 5, R.idx[1] = k * P + 6 * (up + 1);
            R.coef[0] = sg, R.coef[1] = -sg;
            R.rhs = -rho_pair;
            return true;
        }
    };
    auto row_dot = [&](const int* idx, const double* coef, const double* vec) -> double {
        return coef[0] * vec[idx[0]] + coef[1] * vec[idx[1]] + coef[2] * vec[idx[2]];
    };

    // ---- one pass over every row: the most violated one (normalised slack, lowest id on ties) and the largest raw violation ---------
    // The FIRST pass reads the LSC rows from HBM (four in flight per thread) and, in the staged form (small batches: LDS to spare), leaves
    // them translated in LDS; later passes read them from there, or from L2.
    auto pass = [&](double& best, int& bid, double& worst_raw) {
        double bv = 1e300, wr = 1e300;
        int bi = 0x7fffffff;
        auto see = [&](double slack, double inrm, int id) {
            const double v = slack * inrm;
            if (v < bv) bv = v, bi = id;  // (ids ascend within a thread: the first minimum is the lowest id)
            wr = fmin(wr, slack);
        };
        constexpr int U = 4;
        for (int j0 = tid; j0 < nL; j0 += U * T) {
            double rx[U], ry[U], rz[U], rb[U];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int j = j0 + u * T;
                ok[u] = (j < nL) ? load_row(j, rx[u], ry[u], rz[u], rb[u]) : false;
                if (j >= nL) rx[u] = ry[u] = rz[u] = 0.0, rb[u] = -1.0;
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int j = j0 + u * T;
                if (j < nL) {
                    if (staged && !rows_in_lds) {
                        Sx_[j] = ok[u] ? rx[u] : 0.0, Sy_[j] = ok[u] ? ry[u] : 0.0, Sz_[j] = ok[u] ? rz[u] : 0.0, Sb_[j] = ok[u] ? rb[u] : -1.0;
                    }
                    if (ok[u]) {
                        const int cp = j - P * fdiv(j, iP);
                        const double s = rx[u] * c_[cp] + ry[u] * c_[P + cp] + (dim == 3 ? rz[u] * c_[2 * P + cp] : 0.0) - rb[u];
                        const double n2 = rx[u] * rx[u] + ry[u] * ry[u] + rz[u] * rz[u];
                        see(s, n2 > 1e-20 ? rsqrt(n2) : 1.0, j);
                    }
                }
            }
        }
        // intervals
        for (int e = tid; e < NX; e += T) {
            if ((e - P * fdiv(e, iP)) >= 3) {
                const double cv = c_[e];
                see(cv - lo_[e], 1.0, oB + 2 * e);
                see(hi_[e] - cv, 1.0, oB + 2 * e + 1);
            }
        }
        // velocity: c[i+1] - c[i], |.| <= vmax dt / n   (:448-453)
        for (int r = tid; r < dim * 5 * M; r += T) {
            const int k = fdiv(r, i5M), rr = r - k * 5 * M, m = rr / 5, i = rr % 5;
            if (!(m == 0 && i < 2)) {
                const int e = k * P + 6 * m + i;
                const double d = c_[e + 1] - c_[e];
                see(d + Vk[k], 0.70710678118654752, oV + 2 * r);
                see(Vk[k] - d, 0.70710678118654752, oV + 2 * r + 1);
            }
        }
        // acceleration: c[i+2] - 2 c[i+1] + c[i]   (:462-471)
        for (int r = tid; r < dim * 4 * M; r += T) {
            const int k = fdiv(r, i4M), rr = r - k * 4 * M, m = rr / 4, i = rr % 4;
            if (!(m == 0 && i < 1)) {
                const int e = k * P + 6 * m + i;
                const double d = c_[e + 2] - 2.0 * c_[e + 1] + c_[e];
                see(d + Ak[k], 0.40824829046386302, oA + 2 * r);
                see(Ak[k] - d, 0.40824829046386302, oA + 2 * r + 1);
            }
        }
        // communication pairs (uu, up < uu): c[uu][5] - c[up+1][0]   (:482-487 with mi = up + 1 >= 1)
        if (comm_on) {
            for (int r = tid; r < dim * NCP; r += T) {
                const int k = fdiv(r, iNCP), ci = r - k * NCP;
                int uu = 1;
                while (uu * (uu + 1) / 2 <= ci) uu++;
                const int up = ci - uu * (uu - 1) / 2;
                const double d = c_[k * P + 6 * uu + 5] - c_[k * P + 6 * (up + 1)];
                see(d + rho_pair, 0.70710678118654752, oC + 2 * r);
                see(rho_pair - d, 0.70710678118654752, oC + 2 * r + 1);
            }
        }
        wave_argmin(bv, bi);
        wr = wave_min(wr);
        if (NW > 1) {
            __syncthreads();  // (red_ may still be read from the previous reduction)
            if (lane == 0) {
                red_[wv] = bv;
                red_[8 + wv] = wr;
                reinterpret_cast<int*>(red_ + 16)[wv] = bi;
            }
            __syncthreads();
            bv = red_[0], wr = red_[8], bi = reinterpret_cast<int*>(red_ + 16)[0];
            for (int w = 1; w < NW; w++) {
                const double ov = red_[w];
                const int oi = reinterpret_cast<int*>(red_ + 16)[w];
                if (ov < bv || (ov == bv && oi < bi)) bv = ov, bi = oi;
                wr = fmin(wr, red_[8 + w]);
            }
        } else if (staged && !rows_in_lds) {
            __syncthreads();  // the staged rows are read by other lanes from now on
        }
        if (staged) rows_in_lds = true;
        best = bv, bid = bi, worst_raw = wr;
    };
    auto block_max = [&](double v) -> double {
        v = wave_max(v);
        if (NW > 1) {
            __syncthreads();
            if (lane == 0) red_[wv] = v;
            __syncthreads();
            v = red_[0];
            for (int w = 1; w < NW; w++) v = fmax(v, red_[w]);
        }
        return v;
    };
    auto block_sum = [&](double v) -> double {  // fixed order: reproducible
        v = wave_sum(v);
        if (NW > 1) {
            __syncthreads();
            if (lane == 0) red_[wv] = v;
            __syncthreads();
            v = red_[0];
            for (int w = 1; w < NW; w++) v += red_[w];
        }
        return v;
    };

    // ---- the small factor: S = A'W (k x k, SPD), S = Lm Lm', rows owned by the lanes of wavefront 0 ------------------------------------
    // from scratch (after a row left the set): S_ij = a_i . W_j
    auto factor_scratch = [&](int k) -> bool {  // wavefront 0 only; returns false on a lost pivot (dependent rows)
        bool ok = true;
        if (wv == 0) {
            const int i = lane;
            double sdiag = 1.0;
            if (i < k) {
                for (int j = 0; j <= i; j++) Lm_[i * LDL + j] = row_dot(&aint_[4 * i + 1], &acoef_[3 * i], W_ + (size_t)j * NX);
                sdiag = Lm_[i * LDL + i];
            }
            LSCQP_DAS_WAVE_SYNC();
            for (int j = 0; j < k; j++) {
                const double d = Lm_[j * LDL + j];
                if (!(d > 1e-13 * __shfl(sdiag, j, 64))) ok = false;
                const double dj = sqrt(fmax(d, 1e-300));
                const double idj = 1.0 / dj;
                const double lij = (i > j && i < k) ? Lm_[i * LDL + j] * idj : 0.0;
                LSCQP_DAS_WAVE_SYNC();  // every lane has read the pivot before its owner overwrites it
                if (i == j) Lm_[j * LDL + j] = dj, linv_[j] = idj;
                if (i > j && i < k) Lm_[i * LDL + j] = lij;
                LSCQP_DAS_WAVE_SYNC();
                // trailing update of the own row: S_ic -= L_ij L_cj, c = j+1 .. i
                if (i > j && i < k) {
                    for (int cidx = j + 1; cidx <= i; cidx++) Lm_[i * LDL + cidx] -= lij * Lm_[cidx * LDL + j];
                }
                LSCQP_DAS_WAVE_SYNC();
            }
        }
        return ok;
    };
    // r = S^-1 v through the factor (wavefront 0; v_ in, y_ = Lm^-1 v and r_ out; lane j owns component j, the pivot component of a step
    // reaches the others with v_readlane).  With `ratio`: the dual step bound t1 = min over r_j > 0 of u_j / r_j and its row (lowest j on
    // ties) land in red_[24], red_[25] -- one division per lane instead of k per thread.
    auto solve_factor = [&](int k, bool ratio) {
        if (wv == 0) {
            double vi = (lane < k) ? v_[lane] : 0.0;
            for (int j = 0; j < k; j++) {  // forward: Lm y = v
                const int js = __builtin_amdgcn_readfirstlane(j);
                const double yj = lscqp::bcast(vi, js) * linv_[js];
                if (lane == js) y_[js] = yj;
                if (lane > js && lane < k) vi -= Lm_[lane * LDL + js] * yj;
            }
            LSCQP_DAS_WAVE_SYNC();
            double yi = (lane < k) ? y_[lane] : 0.0;
            double ri = 0.0;
            for (int j = k - 1; j >= 0; j--) {  // backward: Lm' r = y
                const int js = __builtin_amdgcn_readfirstlane(j);
                const double rj = lscqp::bcast(yi, js) * linv_[js];
                if (lane == js) r_[js] = rj, ri = rj;
                if (lane < js) yi -= Lm_[js * LDL + lane] * rj;
            }
            if (ratio) {
                double tr = (lane < k && ri > 0.0) ? u_[lane] / ri : 1e300;
                int tl = lane;
                wave_argmin(tr, tl);
                if (lane == 0) red_[24] = tr, red_[25] = (double)(tr < 1e299 ? tl : -1);
            }
        }
    };

    // ---- verification: reduced stationarity  T'(Hx c + fx - A'u), scaled as lscqp_info.res_dual (uniform result) -----------------------
    // g_[0 .. NX): Hx c + fx - A'u;  g_[NX ..): Hx c + fx;  g_[2 NX ..): Hx cfix + fx
    const int NZA = 3 * (M - 1) + (es ? 1 : 3);
    auto verify = [&](int k) -> double {
        __syncthreads();
        for (int e = tid; e < NX; e += T) {
            const int kx = e / P, cp = e % P, m = cp / 6, i = cp % 6;
            const double* cc = &c_[kx * P + 6 * m];
            double hx = 0.0;
#pragma unroll
            for (int j = 0; j < 6; j++) hx += q2s * KQ(i, j) * cc[j];
            const bool term = (i == 5 && m >= M - ts);
            const double fx = term ? -wt2 * goal[kx] : 0.0;
            if (term) hx += wt2 * cc[5];
            double h0 = fx;
            if (m == 0) {
                const double c1 = Hd->v0[kx] * dt * 0.2;
                const double c2 = Hd->a0[kx] * dt * dt * 0.05 + 2.0 * c1;
                h0 += q2s * (KQ(i, 1) * c1 + KQ(i, 2) * c2);
            }
            double lam = 0.0;
            for (int j = 0; j < k; j++) {
#pragma unroll
                for (int t_ = 0; t_ < 3; t_++)
                    if (aint_[4 * j + 1 + t_] == e) lam += u_[j] * acoef_[3 * j + t_];
            }
            g_[e] = hx + fx - lam;
            g_[NX + e] = hx + fx;
            g_[2 * NX + e] = h0;
        }
        __syncthreads();
        double rd = 0.0, gs = 0.0;
        for (int zi = tid; zi < dim * NZA; zi += T) {
            const int kx = zi / NZA, a = zi % NZA;
            const bool last = es && a == 3 * (M - 1);
            const int m = last ? M - 1 : a / 3, j = last ? 0 : a % 3;
#pragma unroll
            for (int w = 0; w < 3; w++) {
                const double* gg = g_ + w * NX + kx * P;
                double comp = last ? (gg[6 * m + 3] + gg[6 * m + 4] + gg[6 * m + 5]) : gg[6 * m + 3 + j];
                if (m + 1 < M) {  // (c0, c1, c2) of the next segment = TB (c3, c4, c5) of this one, TB = [[0,0,1],[0,-1,2],[1,-4,4]]
                    const double* gn = gg + 6 * (m + 1);
                    comp += (j == 0) ? gn[2] : (j == 1) ? (-gn[1] - 4.0 * gn[2]) : (gn[0] + 2.0 * gn[1] + 4.0 * gn[2]);
                }
                if (w == 0) rd = fmax(rd, fabs(comp));
                else gs = fmax(gs, fabs(comp));
            }
        }
        // (one reduction for both: the two maxima packed side by side)
        rd = wave_max(rd);
        gs = wave_max(gs);
        if (NW > 1) {
            __syncthreads();
            if (lane == 0) red_[wv] = rd, red_[8 + wv] = gs;
            __syncthreads();
            rd = red_[0], gs = red_[8];
            for (int w = 1; w < NW; w++) rd = fmax(rd, red_[w]), gs = fmax(gs, red_[8 + w]);
        }
        return rd / fmax(1.0, gs);
    };

    // ---- the loop ---------------------------------------------------------------------------------------------------------------------
    int k = 0, steps = 0;
    bool polished = false, solved = false, haveC = false;
    double res_p = 0.0, res_d = 0.0;
    const double* Cm = Cg;  // column cp = Cm + cp * P (symmetric); the LDS copy once a step needs it
    for (;;) {
        __syncthreads();  // c_ final
        double best, worst;
        int bid;
        pass(best, bid, worst);
        DAS_T(steps == 0 ? 2 : 3);  // first pass / later passes
        if (!(best < -kTolP)) {
            res_p = fmax(0.0, -worst);
            res_d = verify(k);
            DAS_T(4);  // verification
            if (res_d <= kTolD) {
                solved = true;
                break;
            }
            if (polished || k == 0) break;  // (never seen on the bench's classes; the interior-point kernel then solves the instance)
            // POLISH (a stationarity residual above the bar: rounding accumulated over many steps): the point rebuilt from its
            // multipliers, c = c_u + W u -- stationary up to the table's rounding -- and one refinement of the multipliers that puts the
            // active rows back at zero slack:  rho = h_A - A c,  du = S^-1 rho,  u += du,  c += W du.  Then every row is looked at again.
            __syncthreads();
            for (int e = tid; e < NX; e += T) {
                double a = cu_[e];
                for (int j = 0; j < k; j++) a += u_[j] * W_[(size_t)j * NX + e];
                c_[e] = a;
            }
            __syncthreads();
            if (tid < k) v_[tid] = arhs_[tid] - row_dot(&aint_[4 * tid + 1], &acoef_[3 * tid], c_);
            __syncthreads();
            solve_factor(k, false);
            __syncthreads();
            if (tid < k) u_[tid] += r_[tid];
            for (int e = tid; e < NX; e += T) {
                double a = c_[e];
                for (int j = 0; j < k; j++) a += r_[j] * W_[(size_t)j * NX + e];
                c_[e] = a;
            }
            polished = true;
            continue;
        }
        polished = false;
        if (k >= kmax) break;  // more active rows than this launch holds: the interior-point kernel's
        if (cacheC && !haveC) {  // the table of this instance's ts in LDS from the first step on (every step reads up to three of its columns)
            for (int e = tid; e < P * P; e += T) Cc_[e] = Cg[e];
            haveC = true;
            Cm = Cc_;
            __syncthreads();
        }
        // ---- the candidate row p = bid, slot kmax of the descriptors ----
        Row Rp;
        (void)decode(bid, Rp);
        if (tid == 0) {
            aint_[4 * kmax] = bid;
            for (int t = 0; t < 3; t++) aint_[4 * kmax + 1 + t] = Rp.idx[t], acoef_[3 * kmax + t] = Rp.coef[t];
            arhs_[kmax] = Rp.rhs;
        }
        // w_p = C a_p
        for (int e = tid; e < NX; e += T) {
            const int kx = fdiv(e, iP), cp = e - kx * P;
            double a = 0.0;
#pragma unroll
            for (int t = 0; t < 3; t++) {
                const int ix = Rp.idx[t], ik = fdiv(ix, iP);
                if (Rp.coef[t] != 0.0 && ik == kx) a += Rp.coef[t] * Cm[(size_t)(ix - ik * P) * P + cp];
            }
            wp_[e] = a;
        }
        __syncthreads();
        DAS_T(5);  // candidate: decode, table copy, w_p
        const double spp = row_dot(Rp.idx, Rp.coef, wp_);
        double up = 0.0;
        bool stop = false;
        for (;;) {  // partial steps until p has joined the set
            steps++;
            if (steps > max_steps) {
                stop = true;
                break;
            }
            // v = A'w_p, r = S^-1 v, dc = w_p - W r
            if (k > 0) {
                if (tid < k) v_[tid] = row_dot(&aint_[4 * tid + 1], &acoef_[3 * tid], wp_);
                __syncthreads();
                solve_factor(k, true);
                __syncthreads();
            }
            for (int e = tid; e < NX; e += T) {
                double a = wp_[e];
                for (int j = 0; j < k; j++) a -= r_[j] * W_[(size_t)j * NX + e];
                dc_[e] = a;
            }
            __syncthreads();
            const double curv = row_dot(Rp.idx, Rp.coef, dc_);
            const double sp = row_dot(Rp.idx, Rp.coef, c_) - Rp.rhs;
            const double t2 = (curv > 1e-12 * spp) ? -sp / curv : 1e300;
            // t1 = min over the active rows with r_j > 0 of u_j / r_j (lowest j on ties): computed by wavefront 0 with the solve
            const double t1 = (k > 0) ? red_[24] : 1e300;
            const int l = (k > 0) ? (int)red_[25] : -1;
            const double t = fmin(t1, t2);
            if (!(t < 1e299)) {  // no step at all: the rows admit no point (INFEASIBLE is the interior-point kernel's verdict to give)
                stop = true;
                break;
            }
            __syncthreads();  // everybody has read c_, u_, r_
            if (t2 < 1e299) {
                for (int e = tid; e < NX; e += T) c_[e] += t * dc_[e];
            }
            if (tid < k) u_[tid] = fmax(0.0, u_[tid] - t * r_[tid]);
            up += t;
            if (t2 <= t1) {
                // p joins: W, descriptor, multiplier, one more row of the factor (y = Lm^-1 v is in y_)
                double yy = 0.0;
                for (int j = 0; j < k; j++) yy += y_[j] * y_[j];
                const double dnew = spp - yy;
                if (!(dnew > 1e-13 * spp)) {  // dependent on the active rows after all (rounding): leave it
                    stop = true;
                    break;
                }
                for (int e = tid; e < NX; e += T) W_[(size_t)k * NX + e] = wp_[e];
                if (tid == 0) {
                    for (int t_ = 0; t_ < 4; t_++) aint_[4 * k + t_] = aint_[4 * kmax + t_];
                    for (int t_ = 0; t_ < 3; t_++) acoef_[3 * k + t_] = acoef_[3 * kmax + t_];
                    arhs_[k] = arhs_[kmax];
                    u_[k] = up;
                    const double dl = sqrt(dnew);
                    for (int j = 0; j < k; j++) Lm_[k * LDL + j] = y_[j];
                    Lm_[k * LDL + k] = dl;
                    linv_[k] = 1.0 / dl;
                }
                k++;
                break;
            }
            // row l leaves: close the gap in W, descriptors, multipliers; factor from scratch
            __syncthreads();
            for (int e = tid; e < NX; e += T) {
                for (int j = l; j + 1 < k; j++) W_[(size_t)j * NX + e] = W_[(size_t)(j + 1) * NX + e];
            }
            if (tid == 0) {
                for (int j = l; j + 1 < k; j++) {
                    for (int t_ = 0; t_ < 4; t_++) aint_[4 * j + t_] = aint_[4 * (j + 1) + t_];
                    for (int t_ = 0; t_ < 3; t_++) acoef_[3 * j + t_] = acoef_[3 * (j + 1) + t_];
                    arhs_[j] = arhs_[j + 1];
                    u_[j] = u_[j + 1];
                }
            }
            k--;
            __syncthreads();
            const bool okf = factor_scratch(k);
            if (wv == 0 && lane == 0) ctl_[1] = okf ? 0 : 1;
            __syncthreads();
            if (ctl_[1]) {
                stop = true;
                break;
            }
        }
        DAS_T(6);  // the partial steps of the candidate
        if (stop) break;
    }
    __syncthreads();
    if (!solved) {
        hand_over(steps);
        return;
    }

    // ---- epilogue: objective exactly as cplex.getObjValue() reports it (as lscqp_kernel.hpp), control points in the world frame ------------
    double part = 0.0;
    for (int lv = tid; lv < dim * M; lv += T) {
        const int kx = lv / M, m = lv % M;
        const double* cc = &c_[kx * P + 6 * m];
        const double j0 = (cc[3] - cc[0]) - 3.0 * (cc[2] - cc[1]);
        const double j1 = (cc[4] - cc[1]) - 3.0 * (cc[3] - cc[2]);
        const double j2 = (cc[5] - cc[2]) - 3.0 * (cc[4] - cc[3]);
        const double quad = 0.2 * (j0 * j0 + j2 * j2) + (2.0 / 15.0) * j1 * j1 + 0.2 * (j0 * j1 + j1 * j2) + (1.0 / 15.0) * j0 * j2;
        double pp = 0.5 * q2s * 3600.0 * quad;
        const double ok_ = org[kx];
        double corr = 0;
#pragma unroll
        for (int i = 0; i < 6; i++) {
            double r = 0;
#pragma unroll
            for (int ip = 0; ip < 6; ip++) r += cls.dQ[i * 6 + ip] * (cc[ip] + ok_);
            corr += r * (cc[i] + ok_);
        }
        pp += cls.w_c * corr;
        const double dgoal = cc[5] - goal[kx];
        pp += (m >= M - ts) ? cls.w_t * dgoal * dgoal : 0.0;
        part += pp;
    }
    const double obj = block_sum(part);
    for (int e = tid; e < NX; e += T) x_out[q * NX + e] = c_[e] + org[e / P];
    if (tid == 0) {
        obj_out[q] = obj;
        status_out[q] = LSCQP_STATUS_OPTIMAL;
        if (info_out) {
            info_out[q].iterations = steps;
            info_out[q].flags = LSCQP_INFO_ACTIVE_SET;
            info_out[q].res_primal = res_p;
            info_out[q].res_dual = res_d;
            info_out[q].gap = 0.0;  // complementarity is exact: a row is either in the set (slack 0) or carries no multiplier
        }
    }
    DAS_T(7);  // epilogue
}

}  // namespace lscqp_das

// ---- host side ------------------------------------------------------------------------------------------------------------------------

// The tables of a class: for ts = 1 .. M  [U1 | U2 | G1 | C], C = T (T'Hx T)^-1 T' in extended precision (cond(T'Hx T) ~ 2e5 .. 3e6).
// Returns the number of doubles written (M * table_stride(M)); out may be NULL to ask for the size.
extern "C" size_t lscqp_das_build_tables(int M, int es, double dt, double w_c, double w_t, double* out) {
    const int P = 6 * M, NZA = 3 * (M - 1) + (es ? 1 : 3);
    const size_t stride = lscqp_das::table_stride(M);
    if (!out) return stride * (size_t)M;
    typedef long double ld;
    const ld q2s = 2.0L * (ld)w_c / ((ld)dt * dt * dt * dt * dt);
    static const int kq[6][6] = {{720, -1800, 1200, 0, 0, -120},  {-1800, 4800, -3600, 0, 600, 0}, {1200, -3600, 3600, -1200, 0, 0},
                                 {0, 0, -1200, 3600, -3600, 1200}, {0, 600, 0, -3600, 4800, -1800}, {-120, 0, 0, 1200, -1800, 720}};
    static const int tbm[3][3] = {{0, 0, 1}, {0, -1, 2}, {1, -4, 4}};
    // T: P x NZA
    std::vector<ld> Tm((size_t)P * NZA, 0.0L);
    for (int m = 0; m < M; m++) {
        const bool last = es && m == M - 1;
        for (int j = 0; j < 3; j++) Tm[(size_t)(6 * m + 3 + j) * NZA + 3 * m + (last ? 0 : j)] = 1.0L;
        if (m >= 1)
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) Tm[(size_t)(6 * m + i) * NZA + 3 * (m - 1) + j] = (ld)tbm[i][j];
    }
    for (int ts = 1; ts <= M; ts++) {
        std::vector<ld> Hx((size_t)P * P, 0.0L);
        for (int m = 0; m < M; m++)
            for (int i = 0; i < 6; i++)
                for (int j = 0; j < 6; j++) Hx[(size_t)(6 * m + i) * P + 6 * m + j] += q2s * (ld)kq[i][j];
        for (int m = M - ts; m < M; m++) Hx[(size_t)(6 * m + 5) * P + 6 * m + 5] += 2.0L * (ld)w_t;
        // K0 = T'Hx T
        std::vector<ld> HT((size_t)P * NZA, 0.0L), K0((size_t)NZA * NZA, 0.0L);
        for (int i = 0; i < P; i++)
            for (int l = 0; l < P; l++) {
                const ld h = Hx[(size_t)i * P + l];
                if (h == 0.0L) continue;
                for (int j = 0; j < NZA; j++) HT[(size_t)i * NZA + j] += h * Tm[(size_t)l * NZA + j];
            }
        for (int i = 0; i < P; i++)
            for (int a = 0; a < NZA; a++) {
                const ld t = Tm[(size_t)i * NZA + a];
                if (t == 0.0L) continue;
                for (int b = 0; b < NZA; b++) K0[(size_t)a * NZA + b] += t * HT[(size_t)i * NZA + b];
            }
        // Cholesky K0 = R'R ... inverse through the factor
        std::vector<ld> Lc((size_t)NZA * NZA, 0.0L);
        for (int j = 0; j < NZA; j++) {
            ld d = K0[(size_t)j * NZA + j];
            for (int l = 0; l < j; l++) d -= Lc[(size_t)j * NZA + l] * Lc[(size_t)j * NZA + l];
            if (!(d > 0.0L)) return 0;  // (not SPD: the class has no active-set phase; cannot happen for w_c, w_t > 0)
            const ld dj = sqrtl(d);
            Lc[(size_t)j * NZA + j] = dj;
            for (int i = j + 1; i < NZA; i++) {
                ld s = K0[(size_t)i * NZA + j];
                for (int l = 0; l < j; l++) s -= Lc[(size_t)i * NZA + l] * Lc[(size_t)j * NZA + l];
                Lc[(size_t)i * NZA + j] = s / dj;
            }
        }
        // X = Lc^-1 T'  (NZA x P), C = X'X
        std::vector<ld> X((size_t)NZA * P, 0.0L);
        for (int col = 0; col < P; col++) {
            for (int i = 0; i < NZA; i++) {
                ld s = Tm[(size_t)col * NZA + i];
                for (int l = 0; l < i; l++) s -= Lc[(size_t)i * NZA + l] * X[(size_t)l * P + col];
                X[(size_t)i * P + col] = s / Lc[(size_t)i * NZA + i];
            }
        }
        double* const tb = out + (size_t)(ts - 1) * stride;
        std::vector<ld> Cm((size_t)P * P, 0.0L);
        for (int a = 0; a < P; a++)
            for (int b = 0; b <= a; b++) {
                ld s = 0.0L;
                for (int i = 0; i < NZA; i++) s += X[(size_t)i * P + a] * X[(size_t)i * P + b];
                Cm[(size_t)a * P + b] = Cm[(size_t)b * P + a] = s;
            }
        for (int e = 0; e < P; e++) {
            ld u1 = 0.0L, u2 = 0.0L, g1 = 0.0L;
            for (int i = 0; i < 6; i++) {
                u1 += Cm[(size_t)e * P + i] * q2s * (ld)kq[i][1];
                u2 += Cm[(size_t)e * P + i] * q2s * (ld)kq[i][2];
            }
            for (int m = M - ts; m < M; m++) g1 += Cm[(size_t)e * P + 6 * m + 5];
            tb[e] = (double)u1;
            tb[P + e] = (double)u2;
            tb[2 * P + e] = (double)g1;
        }
        for (size_t e = 0; e < (size_t)P * P; e++) tb[3 * P + e] = (double)Cm[e];
    }
    return stride * (size_t)M;
}

#ifdef LSCQP_DAS_TIMING
extern "C" int lscqp_das_cycles(unsigned long long* out, int reset) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(lscqp_das::das_cycles), sizeof(unsigned long long) * 16);
    if (reset) {
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(lscqp_das::das_cycles), z, sizeof z);
    }
    return 0;
}
#endif

extern "C" size_t lscqp_das_lds_bytes(int M, int dim, int kmax, int cacheC, int stage_rows) {
    return sizeof(double) * (size_t)lscqp_das::Layout::make(M, dim, kmax, cacheC, stage_rows).total;
}

// Launch of the phase over a batch.  threads: 64, 128 or 256 per QP; kmax <= 32 active rows; stage_rows: LSC rows per instance kept in LDS
// after the first pass (0: re-read from L2 in every pass; an instance with more rows than that re-reads them too); cap: the obstacle capacity of the kernel
// instance that runs behind the phase (an instance beyond it is left to that kernel's LSCQP_STATUS_CAPACITY).
extern "C" hipError_t lscqp_launch_das(const lscqp::DevClass* cls, int M, int dim, int es, int cap, int threads, int kmax, int max_steps, int cacheC,
                                       int stage_rows, const double* d_tab, int64_t n, const lscqp_header* hdr, const lscqp_row* rows, const uint64_t* row_offsets,
                                       const lscqp_box* sfc, const double* x_init, double* x_out, double* obj_out, int32_t* status_out,
                                       lscqp_info* info_out, hipStream_t stream) {
    if (kmax < 1 || kmax > lscqp_das::kMaxK || (threads != 64 && threads != 128 && threads != 256)) return hipErrorInvalidValue;
    const size_t lds = lscqp_das_lds_bytes(M, dim, kmax, cacheC, stage_rows);
    if (lds > lscqp::kMaxLdsBytes) return hipErrorInvalidValue;
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lscqp_das::das_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)lscqp::kMaxLdsBytes);
        if (e != hipSuccess) return e;
        attr_set[dev].store(true, std::memory_order_release);
    }
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(lscqp_das::das_kernel, dim3((unsigned)n), dim3((unsigned)threads), lds, stream, *cls, M, dim, es, cap, kmax, max_steps, cacheC, stage_rows, d_tab, n,
                       hdr, rows, row_offsets, sfc, x_init, x_out, obj_out, status_out, info_out);
    return hipGetLastError();
}
