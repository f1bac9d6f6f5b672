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
//     one step for row p      w_p = C a_p;   r = S^-1 A'w_p  (S = A'W over the active rows, carried as J = L^-1 of its Cholesky factor:
//                             S^-1 = J'J, two products per step, a row appended when one joins, rotations when one leaves);
//                             dc = w_p - W r;   t = min( min_{r_j > 0} u_j / r_j ,  -slack_p / a_p'dc );   c += t dc,  u -= t r,  u_p += t
//                             t = the second: p joins the active set;  t = the first: row j leaves it and the step is repeated.
//
// The work per QP is one pass over the rows per step (read from HBM the first time; afterwards from LDS in the small-batch form, from
// L2 otherwise) plus a handful of short vector operations: small batches are bound by the chain of memory round trips of their slowest
// instance, large ones by instruction issue at 0.27 of the HBM roof (NOTES.md sections 12-13).  What it returns is a KKT point of the reference's model: primal violation <= 1e-9 m on EVERY row
// (the last pass), multipliers >= 0, exact complementarity, and the reduced stationarity residual verified against the same scale the
// interior-point kernel uses (<= 1e-9) -- after a final "polish" that rebuilds the point from its multipliers and refines them once.
// An instance the phase does not finish (more active rows than its budget, more steps than its budget, a dependent active set, an
// infeasible row system, a failed verification) is LEFT to the interior-point kernel, which runs behind it over the same batch in
// "first pass after the active-set phase" mode (cls.repair == 3) and skips what is already OPTIMAL.  Nothing here is a CPU fallback and
// nothing is approximate: both methods return the optimum of the same strictly convex QP.
//
// Organisation: one workgroup (64 .. 256 threads) per QP, M / dim / end stop / n_obs are run-time values (one kernel for every class);
// row ids:  [LSC rows o*P + cp | interval lo/hi per (axis, cp) | velocity lo/hi | acceleration lo/hi | communication pairs lo/hi],
// selection = the most violated row (raw slack), lowest id on ties: results are reproducible bit for bit from run to run and across
// the kernel's forms (wavefronts per QP, row format, first look peeled or not; built with -ffp-contract=on for that).
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
__host__ __device__ inline int num_pairs(int M, int dim) { return dim * (6 * M + 5 * M + 4 * M + M * (M - 1) / 2); }

// LDS carve of one QP, in doubles.
struct Layout {
    int P, NX, kmax, NPAIR;
    int o_hdr, o_sfc, o_c, o_cu, o_lam, o_plo, o_phi, o_pix, o_W, o_L, o_u, o_r, o_arhs, o_acoef, o_aint, o_red, o_ctl, o_wb, o_dq, o_C, o_rows, o_tl, n_stage, total;
    __host__ __device__ static Layout make(int M, int dim, int kmax, int cacheC, int stage_rows = 0) {
        Layout s;
        s.P = 6 * M, s.NX = dim * s.P, s.kmax = kmax, s.NPAIR = num_pairs(M, dim);
        int o = 0;
        auto take = [&](int n) { const int at = o; o += (n + 1) & ~1; return at; };
        s.o_hdr = take(32);
        s.o_sfc = take(6 * M);
        s.o_c = take(3 * s.P), s.o_cu = take(s.NX), s.o_lam = take(s.NX);  // (c_: a third, zero axis in 2-D: row evaluation without a branch on dim)
        s.o_plo = take(s.NPAIR), s.o_phi = take(s.NPAIR), s.o_pix = take((s.NPAIR + 1) / 2);  // two-sided rows: bounds, packed stencil
        s.o_W = take((kmax + 1) * s.NX);      // w_j = C a_j of the active rows; slot k (the next free one) holds the candidate's
        s.o_L = take(kmax * (kmax + 1));
        s.o_u = take(kmax + 1), s.o_r = take(kmax + 4), s.o_arhs = take(kmax + 1);
        s.o_acoef = take(3 * (kmax + 1));
        s.o_aint = take(2 * (kmax + 1) + 2);  // ints: per active row {id, entry0, entry1, entry2} (+ the candidate); entry = axis << 16 | control point
        s.o_red = take(2 * 24);               // cross-wavefront reductions, double buffered
        s.o_ctl = take(8);
        s.o_wb = take(8);  // world box of the class (a kernel argument indexed with a run-time axis would be fetched through vector memory)
        s.o_dq = take(36);  // the objective's coefficient-rounding term (36 doubles as a kernel argument live in 72 scalar registers the kernel does not have)
        s.o_C = take(cacheC ? s.P * s.P : 0);
        s.n_stage = stage_rows;  // LSC rows of the instance kept in LDS after the first pass (SoA nx | ny | nz | b), 0: re-read from L2
        s.o_rows = take(4 * stage_rows);
#ifdef LSCQP_DAS_TIMING
        s.o_tl = take(16);
#endif
        s.total = o;
        return s;
    }
};

// Wave reductions on the DPP network (lscqp_kernel.hpp: ~150 cycles per fp64 value against ~600 for a ds_bpermute butterfly).
__device__ __forceinline__ double wave_max(double v) { return lscqp::wave_max(v); }
__device__ __forceinline__ double wave_min(double v) { return -lscqp::wave_max(-v); }
__device__ __forceinline__ double wave_sum(double v) { return lscqp::wave_sum(v); }
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_min_i32(int v) {
    return min(v, __builtin_amdgcn_update_dpp(2147483647, v, CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ int wave_min_i32(int v) {  // (the scan of lscqp::wave_reduce1 on one register per value instead of two)
    v = dpp_min_i32<0x111, 0xf>(v);
    v = dpp_min_i32<0x112, 0xf>(v);
    v = dpp_min_i32<0x114, 0xf>(v);
    v = dpp_min_i32<0x118, 0xf>(v);
    v = dpp_min_i32<0x142, 0xa>(v);
    v = dpp_min_i32<0x143, 0xc>(v);
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ void wave_argmin(double& v, int& id) {  // lexicographic (value, id), ids >= 0: every lane ends with the result
    const double vm = wave_min(v);
    id = wave_min_i32((v == vm) ? id : 2147483647);
    v = vm;
}
// a / b for small non-negative integers (a < 2^20, b <= 2^10) through one fp32 multiplication: exact, and a handful of instructions where an
// integer division by a run-time value costs ~40
__device__ __forceinline__ int fdiv(int a, float inv_b) { return (int)(((float)a + 0.5f) * inv_b); }

// LDS hand-overs.  Inside ONE wavefront: its LDS operations execute in order, the fences keep the compiler from moving them.  Across the
// workgroup: s_barrier behind a wait on the LDS counter only -- a __syncthreads() would also wait for every global load in flight, and the
// rows of the first pass are meant to stay in flight across the barriers of the prologue.
#ifndef LSCQP_DAS_FULL_SYNC
#define LSCQP_DAS_WAVE_SYNC()                                   \
    do {                                                        \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  \
        __builtin_amdgcn_wave_barrier();                        \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  \
    } while (0)
#define LSCQP_DAS_BARRIER()                                                           \
    do {                                                                              \
        if constexpr (NW > 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); \
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                       \
    } while (0)
#else
// The TWIN of the race test (tests/test_race_twin.py; lsc_dr_planner_amd/build.py builds liblscqp_sync.so from this file with
// -DLSCQP_DAS_FULL_SYNC): every hand-over waits for EVERYTHING in flight -- vector memory, LDS, scalar memory -- behind workgroup-scope fences,
// and the workgroup barrier is the compiler's own __syncthreads().  Slower, and by construction free of the one assumption the product's
// hand-overs make (LDS-counter waits only, global loads left in flight); the test demands bit-identical results from both.
#define LSCQP_DAS_WAVE_SYNC()                                           \
    do {                                                                \
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");          \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     \
        __builtin_amdgcn_wave_barrier();                                \
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");          \
    } while (0)
#define LSCQP_DAS_BARRIER()                                             \
    do {                                                                \
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");          \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     \
        if constexpr (NW > 1) __syncthreads();                          \
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");          \
    } while (0)
#endif

// Development aid: per-phase cycle totals, compiled in only with -DLSCQP_DAS_TIMING (tools/das_timing.py)
#ifdef LSCQP_DAS_TIMING
__device__ unsigned long long das_cycles[16];
// (thread 0 accumulates in LDS and adds to the global totals once, at the end: an atomic behind every probe would be waited for by the next
// wait on vector memory -- a round trip of microseconds booked on whatever phase comes next)
#define DAS_T_DECL()                                                                                                                  \
    unsigned long long* const das_tl_ = reinterpret_cast<unsigned long long*>(smem + Layout::make(M, dim, kmax, cacheC, stage_rows).o_tl); \
    if (threadIdx.x == 0)                                                                                                             \
        for (int i_ = 0; i_ < 16; i_++) das_tl_[i_] = 0;                                                                              \
    unsigned long long tprev_ = __builtin_readcyclecounter()
#define DAS_T(slot)                                                        \
    do {                                                                   \
        if ((LSCQP_DAS_TIMING >> (slot)) & 1) {                            \
            const unsigned long long now_ = __builtin_readcyclecounter();  \
            if (tid == 0) das_tl_[slot] += now_ - tprev_;                  \
            tprev_ = now_;                                                 \
        }                                                                  \
    } while (0)
#ifndef LSCQP_DAS_TIMING_MIN_STEPS
#define LSCQP_DAS_TIMING_MIN_STEPS 0  /* only instances with at least that many steps are booked */
#endif
#define DAS_T_FLUSH()                                                                  \
    do {                                                                               \
        if (tid == 0 && steps >= LSCQP_DAS_TIMING_MIN_STEPS)                           \
            for (int i_ = 0; i_ < 16; i_++) atomicAdd(&das_cycles[i_], das_tl_[i_]);   \
    } while (0)
#else
#define DAS_T_DECL() \
    do {             \
    } while (0)
#define DAS_T(slot) \
    do {            \
    } while (0)
#define DAS_T_FLUSH() \
    do {              \
    } while (0)
#endif

// One row of the model as (<= 3 entries, right-hand side): a'c >= h.  An entry names (axis, control point) as axis << 16 | cp.
struct Row {
    int ent[3];
    double coef[3];
    double rhs;
};
__device__ __forceinline__ int ent_axis(int e) { return e >> 16; }
__device__ __forceinline__ int ent_cp(int e) { return e & 0xffff; }

// C couples control points of the same axis only.
// (C a)[axis kx, control point cp]
__device__ __forceinline__ double ccol(const int* ea, const double* ca, int kx, int cp, const double* __restrict__ Cm, int P) {
    // (loads without a test -- an unused entry is (axis 0, control point 0) with coefficient 0 -- so that the three are in flight together)
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const double v = Cm[(size_t)ent_cp(ea[i]) * P + cp];
        s += ((ent_axis(ea[i]) == kx) ? ca[i] : 0.0) * v;
    }
    return s;
}
// a'C b for two rows
__device__ __forceinline__ double cdot(const int* ea, const double* ca, const int* eb, const double* cb, const double* __restrict__ Cm, int P) {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const double v = Cm[(size_t)ent_cp(ea[i]) * P + ent_cp(eb[j])];
            s += ((ent_axis(ea[i]) == ent_axis(eb[j])) ? ca[i] * cb[j] : 0.0) * v;
        }
    }
    return s;
}

#ifndef LSCQP_DAS_KU1
#define LSCQP_DAS_KU1 2
#endif
#ifndef LSCQP_DAS_WPE1
#define LSCQP_DAS_WPE1 3
#endif
// SCREEN: the lean form for batches that fill the chip -- unconstrained minimiser, ONE pass over the rows, verification; an instance with a
// violated row is left (LSCQP_STATUS_ITER_LIMIT) to the full form, which runs behind it over the same batch and skips what is OPTIMAL
// (`behind` != 0).  Without the step loop the kernel needs half the registers: twice the wavefronts per SIMD for the phase that streams
// the rows from HBM.
#ifndef LSCQP_DAS_WPES
#define LSCQP_DAS_WPES 4
#endif
template <int NW, bool F32, bool SCREEN = false, bool PEEL = false>
__global__ __launch_bounds__(64 * NW, (SCREEN ? LSCQP_DAS_WPES : NW == 1 ? LSCQP_DAS_WPE1 : 1)) void das_kernel(DevClass cls, int M, int dim, int es, int cap, int kmax, int max_steps, int cacheC, int stage_rows, int behind,
                                                      const double* __restrict__ tab, int64_t n, const lscqp_header* __restrict__ hdr,
                                                      const lscqp_row* __restrict__ rows, const uint64_t* __restrict__ row_offsets,
                                                      const lscqp_box* __restrict__ sfc, const double* __restrict__ x_init, double* __restrict__ x_out,
                                                      double* __restrict__ obj_out, int32_t* __restrict__ status_out, lscqp_info* __restrict__ info_out) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int T = 64 * NW;
    constexpr int kU = SCREEN ? 4 : (NW == 1) ? LSCQP_DAS_KU1 : 4;  // LSC rows in flight per thread (the one-wavefront full form trades them for a third wavefront per SIMD)
    const int64_t k0 = blockIdx.x;
    if (k0 >= n) return;
    const int64_t q = cls.order ? (int64_t)cls.order[k0] : k0;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    DAS_T_DECL();
    const Layout L = Layout::make(M, dim, kmax, cacheC, stage_rows);
    const int P = L.P, NX = L.NX, NPAIR = L.NPAIR;
    double* const H_ = smem + L.o_hdr;
    double* const sfc_ = smem + L.o_sfc;
    double* const c_ = smem + L.o_c;
    double* const cu_ = smem + L.o_cu;
    double* const lam_ = smem + L.o_lam;
    double* const plo_ = smem + L.o_plo;
    double* const phi_ = smem + L.o_phi;
    int* const pix_ = reinterpret_cast<int*>(smem + L.o_pix);  // packed stencil of a two-sided row: type << 24 | first entry index (in 0 .. NX-1) << 12 | second
    double* const W_ = smem + L.o_W;   // [kmax + 1][NX]
    double* const Jm_ = smem + L.o_L;  // [kmax][kmax + 1] J = L^-1, S = A'C A = L L' over the active rows
    double* const u_ = smem + L.o_u;
    double* const r_ = smem + L.o_r;
    double* const arhs_ = smem + L.o_arhs;    // [kmax + 1]: slot kmax = the candidate row
    double* const acoef_ = smem + L.o_acoef;  // [kmax + 1][3]
    int* const aint_ = reinterpret_cast<int*>(smem + L.o_aint);  // [kmax + 1][4]: id, entry0, entry1, entry2
    double* const red_ = smem + L.o_red;      // [2][16]
    double* const ctl_ = smem + L.o_ctl;      // step decision of wavefront 0: t, kind, leaving row, accumulated multiplier of the candidate
    double* const wb_ = smem + L.o_wb;
    double* const dq_ = smem + L.o_dq;        // world_min[3], world_max[3]
    double* const Cc_ = smem + L.o_C;
    double* const Sx_ = smem + L.o_rows;      // staged LSC rows: [nx | ny | nz | b] x stage_rows
    double* const Sy_ = Sx_ + stage_rows;
    double* const Sz_ = Sy_ + stage_rows;
    double* const Sb_ = Sz_ + stage_rows;
    const int LDL = kmax + 1;
    int par = 0;  // which half of red_ the next cross-wavefront reduction uses (double buffered: one barrier per reduction)

    if (!SCREEN && behind) {  // (uniform) behind the lean form: what it finished is skipped before anything is fetched
        if (status_out[q] == LSCQP_STATUS_OPTIMAL) return;
    }
    // ---- header, corridor boxes (and the instance's row offset: one memory round trip for all three) -------------------------------
    const uint64_t roff = row_offsets ? row_offsets[q] : 0;
    // the class's two-sided rows (lscqp_das_build_pairs, behind the tables and the 36 rounding terms): the first four of this thread are asked
    // for NOW -- they depend on nothing the header holds
    constexpr int kPB = 4;
    const int2* const pairs_g = reinterpret_cast<const int2*>(tab + (size_t)M * table_stride(M) + 36);
    int2 pwr[kPB];
#pragma unroll
    for (int u = 0; u < kPB; u++) pwr[u] = pairs_g[min(tid + u * T, NPAIR - 1)];
    {
        const double* hsrc = reinterpret_cast<const double*>(hdr + q);
        const double* ssrc = reinterpret_cast<const double*>(sfc) + q * 6 * M;
        for (int e = tid; e < 32 + (cls.use_sfc ? 6 * M : 0); e += T) (e < 32 ? H_[e] : sfc_[e - 32]) = e < 32 ? hsrc[e] : ssrc[e - 32];
        if (tid >= 64 - 36 && tid < 64) dq_[tid - (64 - 36)] = tab[(size_t)M * table_stride(M) + (tid - (64 - 36))];  // (behind the tables: lscqp_api.hip, das_refresh)
        if (tid < 6) {  // (selects, not an indexed kernel argument)
            const double w = tid == 0 ? cls.world_min[0] : tid == 1 ? cls.world_min[1] : tid == 2 ? cls.world_min[2] : tid == 3 ? cls.world_max[0] : tid == 4 ? cls.world_max[1] : cls.world_max[2];
            wb_[tid] = w;
        }
    }
    __syncthreads();
    DAS_T(0);  // header, boxes, row offset
    const lscqp_header* Hd = reinterpret_cast<const lscqp_header*>(H_);
    const lscqp_box* sfcl = reinterpret_cast<const lscqp_box*>(sfc_);
    const int n_obs = Hd->n_obs;
    // Handing an instance over: the interior-point kernel behind this phase solves whatever is not OPTIMAL (cls.repair == 3 there).
    // (why: LSCQP_DAS_WHY_* of include/lscqp.h, left in lscqp_info.res_dual of an instance nobody solves afterwards -- LSCQP_ACTIVE_SET_ONLY,
    // tests and tools/loaded_probe.py; the pass that solves the instance overwrites the record)
    auto hand_over = [&](int steps, int why) {
        for (int e = tid; e < NX; e += T) x_out[q * NX + e] = x_init ? x_init[q * NX + e] : Hd->p0[fdiv(e, 1.0f / (float)P)];
        if (tid == 0) {
            obj_out[q] = 0.0;
            status_out[q] = LSCQP_STATUS_ITER_LIMIT;
            if (info_out) {
                info_out[q].iterations = 0;
                info_out[q].flags = 0;
                info_out[q].res_primal = 0.0;
                info_out[q].res_dual = (double)why;
                info_out[q].gap = (double)steps;  // (overwritten by the pass that solves the instance)
            }
        }
    };
    // An instance whose row system has no point, PROVEN inside the phase (an empty interval; a violated row whose normal lies in the span of
    // the active rows' with no multiplier to give way: a Farkas certificate, taken only where the violation is beyond doubt -- the same
    // 1e-6 m the interior-point kernel judges by): LSCQP_STATUS_INFEASIBLE here and now, nothing left for the kernel behind.  The
    // reference's caller keeps initial_traj for any failure (src/traj_planner.cpp:767-797); x_out is the handed-over start, as above.
    auto infeasible_out = [&](int steps, double violation) {
        for (int e = tid; e < NX; e += T) x_out[q * NX + e] = x_init ? x_init[q * NX + e] : Hd->p0[fdiv(e, 1.0f / (float)P)];
        if (tid == 0) {
            obj_out[q] = 0.0;
            status_out[q] = LSCQP_STATUS_INFEASIBLE;
            if (info_out) {
                info_out[q].iterations = steps;
                info_out[q].flags = LSCQP_INFO_ACTIVE_SET;
                info_out[q].res_primal = violation;
                info_out[q].res_dual = 0.0;
                info_out[q].gap = 0.0;
            }
        }
    };
    if (n_obs > cap || n_obs < 0) {  // the kernel instance behind this phase refuses it (LSCQP_STATUS_CAPACITY): its verdict, not ours
        hand_over(0, LSCQP_DAS_WHY_CAPACITY);
        return;
    }
    const double dt = cls.dt;
    const double org0 = Hd->p0[0], org1 = Hd->p0[1], org2 = Hd->p0[2];
    const double* const org = Hd->p0;  // (LDS: indexed with a run-time axis; a register array would be materialised in scratch memory)
    int ts = Hd->terminal_segments;
    if (ts <= 0) {  // src/traj_optimizer.cpp:530-538 in fp64 (as lscqp_kernel.hpp)
        const double g0 = Hd->goal[0] - org0, g1 = Hd->goal[1] - org1, g2 = Hd->goal[2] - org2;
        ts = (int)((M * dt - sqrt(g0 * g0 + g1 * g1 + g2 * g2) / Hd->nominal_velocity + 1e-9) / dt);
        if (ts < 1) ts = 1;
    }
    if (ts > M) ts = M;
    const double q2s = cls.q2s, wt2 = 2.0 * cls.w_t;
    const double* const tb = tab + (size_t)(ts - 1) * table_stride(M);
    const double* const U1 = tb, * const U2 = tb + P, * const G1 = tb + 2 * P, * const Cg = tb + 3 * P;
    const bool comm_on = cls.comm_range > 0;
    const double rho_pair = 0.5 * cls.comm_range - Hd->radius;  // :484
    const double rho_wp = 0.5 * cls.comm_range - 1e-5;          // :495
    const int nL = n_obs * P;
    const float iP = 1.0f / (float)P;
    const bool staged = stage_rows > 0 && nL <= stage_rows;  // (uniform)
    bool rows_in_lds = false;                                // set by the first pass

    // ---- memory first: the table vectors of this thread's control points and its first LSC rows are requested before anything is computed --
    // (the row format is a template parameter and the index is clamped instead of guarded: a branch around a load makes the compiler wait
    // for every load right behind it -- the loads of a thread have to be in flight TOGETHER)
    auto fetch_row = [&](int j, double& x, double& y, double& z, double& w) {  // raw row j of this instance
        if constexpr (F32) {
            const float4 f = reinterpret_cast<const float4*>(rows)[roff + (uint64_t)j];
            x = f.x, y = f.y, z = f.z, w = f.w;
        } else {
            const double4 d = *reinterpret_cast<const double4*>(&rows[roff + (uint64_t)j]);
            x = d.x, y = d.y, z = d.z, w = d.w;
        }
    };
    constexpr int NE = 4;  // control-point entries per thread the prologue handles in registers (NX <= 4 T for every shape: 216 at M = 12 in 3-D)
    double tu1[NE], tu2[NE], tg1[NE];
#pragma unroll
    for (int i = 0; i < NE; i++) {
        const int e = tid + i * T;
        if (e < NX) {
            const int cp = e - P * fdiv(e, iP);
            tu1[i] = U1[cp], tu2[i] = U2[cp], tg1[i] = G1[cp];
        }
    }
    double px[kU], py[kU], pz[kU], pw[kU];  // the first rows of this thread, raw
#pragma unroll
    for (int u = 0; u < kU; u++) px[u] = py[u] = pz[u] = 0.0, pw[u] = -1.0;
    if (nL > 0) {
#pragma unroll
        for (int u = 0; u < kU; u++) {
            const int j = tid + u * T;
            fetch_row(j < nL ? j : 0, px[u], py[u], pz[u], pw[u]);
        }
    }

    // The small-batch form asks for the class's table NOW as well (an instance with a step would otherwise wait a memory round trip for it at
    // its first step -- and in a batch of 64 that one instance is the launch's time); a quiet instance never waits for these loads.
    constexpr int kCPre = (NW == 4 && !PEEL && !SCREEN) ? 6 : 0;  // table entries per thread held in registers (6 x 256 >= 36 M^2 up to M = 6)
    double cpre[kCPre > 0 ? kCPre : 1];
    const bool c_prefetched = kCPre > 0 && cacheC && P * P <= kCPre * T;  // (uniform)
    if constexpr (kCPre > 0) {
        if (c_prefetched) {
#pragma unroll
            for (int i = 0; i < kCPre; i++) {
                const int e = tid + i * T;
                cpre[i] = Cg[e < P * P ? e : 0];
            }
        }
    }

    // ---- the two-sided rows, one table for all four families (ids nL + 2 r + side; side 0: stencil - lo >= 0, side 1: hi - stencil >= 0) ----
    //   r in [0, NX)                    interval of one control point: world box, corridor, communication rows on c[m][5]  (:252-265, 372-397, 482-497)
    //   then dim * 5M velocity rows     c[i+1] - c[i],            |.| <= vmax dt / n          (:448-453)
    //   then dim * 4M acceleration rows c[i+2] - 2 c[i+1] + c[i], |.| <= amax dt^2 / (n (n-1)) (:462-471)
    //   then dim * NCP pairs (uu, up)   c[uu][5] - c[up+1][0],    |.| <= rho                   (:482-487 with mi = up + 1 >= 1)
    // type 0: no row; 1: interval; 2: velocity; 3: acceleration; 4: pair.  Entry indices are positions in c_ (axis * P + control point).
    bool empty = false;
    const double hv_c = dt * 0.2, ha_c = dt * dt * 0.05;
    for (int r0 = tid; r0 < NPAIR; r0 += kPB * T) {
        if (r0 != tid) {
#pragma unroll
            for (int u = 0; u < kPB; u++) pwr[u] = pairs_g[min(r0 + u * T, NPAIR - 1)];
        }
#pragma unroll
        for (int u = 0; u < kPB; u++) {
            if ((r0 - tid) + u * T >= NPAIR) break;  // (uniform: no thread of the workgroup has a row in this slot)
            const int r = r0 + u * T;
            const int w0 = pwr[u].x, w1 = pwr[u].y;
            const int fam = w1 & 3, k = (w1 >> 2) & 3, m = (w1 >> 4) & 15;
            const bool last = (w1 >> 8) & 1, rs = (w1 >> 9) & 1;
            // every load of every family, without a test (one branch per family made the compiler wait for each family's loads in turn)
            const double ok_ = org[k], wlo = wb_[k], whi = wb_[3 + k], wpk = Hd->next_waypoint[k] - ok_;
            const double hv = Hd->vmax[k] * hv_c, ha = Hd->amax[k] * ha_c;
            double lo = wlo - ok_, hi = whi - ok_;  // :252-253,260-265
            if (cls.rsfc && rs) {                  // :255-258
                lo = -100.0 - ok_;
                hi = 100.0 - ok_;
            }
            if (cls.use_sfc) {  // (uniform) :372-397
                lo = fmax(lo, sfcl[m].bmin[k] - ok_);
                hi = fmin(hi, sfcl[m].bmax[k] - ok_);
            }
            const double clo = fmax(lo, fmax(-rho_pair, wpk - rho_wp)), chi = fmin(hi, fmin(rho_pair, wpk + rho_wp));  // pairs (m, mi = 0) :482-487, waypoint rows :494-497
            lo = (comm_on && last) ? clo : lo;
            hi = (comm_on && last) ? chi : hi;
            const double hs = fam == 1 ? hv : fam == 2 ? ha : rho_pair;  // :448-453, :462-471, :482-487
            lo = fam == 0 ? lo : -hs;
            hi = fam == 0 ? hi : hs;
            if (r < NPAIR) {
                if (fam == 0 && (w0 >> 24) != 0 && lo > hi) empty = true;
                plo_[r] = lo, phi_[r] = hi;
                pix_[r] = w0;
            }
        }
    }
    // ---- unconstrained optimum: c_u[k] = cfix[k] - c1_k U1 - c2_k U2 + 2 w_t goal_k G1 ---------------------------------------------------
#pragma unroll
    for (int i = 0; i < NE; i++) {
        const int e = tid + i * T;
        if (e < NX) {
            const int k = fdiv(e, iP), cp = e - k * P;
            const double c1 = Hd->v0[k] * dt * 0.2;
            const double c2 = Hd->a0[k] * dt * dt * 0.05 + 2.0 * c1;
            const double fixv = (cp == 1) ? c1 : (cp == 2) ? c2 : 0.0;
            const double cv = fixv - c1 * tu1[i] - c2 * tu2[i] + wt2 * (Hd->goal[k] - org[k]) * tg1[i];
            c_[e] = cv;
            cu_[e] = cv;
        }
    }
    if (dim == 2)
        for (int e = tid; e < P; e += T) c_[2 * P + e] = 0.0;
    if (tid == 0) ctl_[4] = 0.0;
    LSCQP_DAS_BARRIER();
    if (empty) ctl_[4] = 1.0;  // (benign race: every writer stores 1)
    LSCQP_DAS_BARRIER();
    if (ctl_[4] != 0.0) {  // an empty interval (lo > hi on one control point: exact)
        infeasible_out(0, 1.0);
        return;
    }
    DAS_T(1);  // tables, two-sided rows, unconstrained optimum

    // ---- rows by id ---------------------------------------------------------------------------------------------------------------------
    // LSC row j, translated to the agent's position; false: the reference drops it (:404-406: first three control points, :409-411: zero normal)
    auto translate = [&](int j, double x, double y, double z, double w, double& nx, double& ny, double& nz, double& b) -> bool {
        nx = x, ny = y, nz = (dim == 3) ? z : 0.0;
        b = w - (x * org0 + y * org1 + (dim == 3 ? z * org2 : 0.0));
        return !(x * x + y * y + z * z < 1e-10) && (j - P * fdiv(j, iP)) >= 3;
    };
    auto load_row = [&](int j, double& nx, double& ny, double& nz, double& b) -> bool {  // (single rows: the candidate's)
        if (rows_in_lds) {  // (uniform) staged by the first pass: dropped rows hold (0, 0, 0 | -1)
            nx = Sx_[j], ny = Sy_[j], nz = Sz_[j], b = Sb_[j];
            return b != -1.0 || nx != 0.0 || ny != 0.0 || nz != 0.0;
        }
        double x, y, z, w;
        fetch_row(j, x, y, z, w);
        return translate(j, x, y, z, w, nx, ny, nz, b);
    };
    // stencil of a two-sided row on a vector in c_ layout
    // (every family as the three-point stencil v[i0] + a1 v[i1] + a2 v[e0] with loads that carry no test: a branch per family makes the
    // compiler wait for each family's loads in turn; the products with -1, -2, 1, 0 are exact, the value is the plain expression's)
    auto pair_val = [&](int pk, const double* vec) -> double {
        const int type = pk >> 24, e0 = (pk >> 12) & 0xfff, e1 = pk & 0xfff;
        const int i0 = (type == 4) ? e1 : e0 + max(type - 1, 0);  // 1: c[e0]   2: c[e0+1] - c[e0]   3: c[e0+2] - 2 c[e0+1] + c[e0]   4: c[e1] - c[e0]
        const int i1 = (type == 3) ? e0 + 1 : e0;
        const double a1 = (type <= 1) ? 0.0 : (type == 3) ? -2.0 : -1.0;
        const double a2 = (type == 3) ? 1.0 : 0.0;
        const double v0 = vec[i0], v1 = vec[i1], v2 = vec[e0];
        return (v0 + a1 * v1) + a2 * v2;
    };
    auto ent_of = [&](int e) -> int {  // position in c_ -> axis << 16 | control point
        const int k = fdiv(e, iP);
        return (k << 16) | (e - k * P);
    };
    // the row with id `rid` as entries (uniform over the workgroup)
    auto decode = [&](int rid, Row& R) {
        R.ent[0] = R.ent[1] = R.ent[2] = 0;
        R.coef[0] = R.coef[1] = R.coef[2] = 0.0;
        if (rid < nL) {
            const int cp = rid - P * fdiv(rid, iP);
            double nx, ny, nz, b;
            (void)load_row(rid, nx, ny, nz, b);
            R.ent[0] = cp, R.ent[1] = (1 << 16) | cp, R.ent[2] = (2 << 16) | cp;
            R.coef[0] = nx, R.coef[1] = ny, R.coef[2] = (dim == 3) ? nz : 0.0;
            R.rhs = b;
            return;
        }
        const int s = rid - nL, r = s >> 1, pk = pix_[r];
        const int type = pk >> 24, e0 = (pk >> 12) & 0xfff, e1 = pk & 0xfff;
        const double sg = (s & 1) ? -1.0 : 1.0;
        R.rhs = (s & 1) ? -phi_[r] : plo_[r];
        if (type == 1) {
            R.ent[0] = ent_of(e0), R.coef[0] = sg;
        } else if (type == 2) {
            R.ent[0] = ent_of(e0 + 1), R.ent[1] = ent_of(e0), R.coef[0] = sg, R.coef[1] = -sg;
        } else if (type == 3) {
            R.ent[0] = ent_of(e0 + 2), R.ent[1] = ent_of(e0 + 1), R.ent[2] = ent_of(e0), R.coef[0] = sg, R.coef[1] = -2.0 * sg, R.coef[2] = sg;
        } else {
            R.ent[0] = ent_of(e1), R.ent[1] = ent_of(e0), R.coef[0] = sg, R.coef[1] = -sg;
        }
    };
    auto row_dot = [&](const int* ent, const double* coef, const double* vec) -> double {  // a'vec for a vector in c_ layout
        return coef[0] * vec[ent_axis(ent[0]) * P + ent_cp(ent[0])] + coef[1] * vec[ent_axis(ent[1]) * P + ent_cp(ent[1])] +
               coef[2] * vec[ent_axis(ent[2]) * P + ent_cp(ent[2])];
    };

    // ---- one pass over every row: the most violated one (normalised slack, lowest id on ties) and the largest raw violation ---------
    // The FIRST pass consumes the rows requested in the prologue (HBM), asks for the rest four at a time and, in the staged form (small
    // batches: LDS to spare), leaves them translated in LDS; later passes read them from there, or from L2.
    bool first_pass = true;  // (uniform)
    // Rows are judged by their RAW slack (metres, the interior-point kernel's bar), which is also what picks the candidate.  Straight-line code:
    // a dropped row, or a slot behind the instance's last row, is the harmless row (0, 0, 0 | -1) -- slack +1 -- instead of a branch.
    auto pass_local = [&](double& bv, int& bi) {
        bv = 1e300;
        bi = 0x7fffffff;
        auto see = [&](double slack, int id) {
            // (a NaN slack -- NaN in a row, in the header, in the point -- must never read as "satisfied": it becomes the most violated row
            // there is; the step for it finds no length and the instance goes to the interior-point kernel, which answers NUMERIC)
            slack = (slack == slack) ? slack : -1e308;
            const bool lt = slack < bv;  // (ids ascend within a thread: the first minimum is the lowest id)
            bv = lt ? slack : bv;
            bi = lt ? id : bi;
        };
        auto eval = [&](int j0, const double* rx, const double* ry, const double* rz, const double* rb) {
#pragma unroll
            for (int u = 0; u < kU; u++) {
                const int j = j0 + u * T, jc = j < nL ? j : 0;
                const int cp = jc - P * fdiv(jc, iP);
                see(rx[u] * c_[cp] + ry[u] * c_[P + cp] + rz[u] * c_[2 * P + cp] - rb[u], j);
            }
        };
        // raw -> translated, dropped rows and slots past the end neutralised
        auto prep = [&](int j0, const double* x, const double* y, const double* z, const double* w, double* rx, double* ry, double* rz, double* rb) {
#pragma unroll
            for (int u = 0; u < kU; u++) {
                const int j = j0 + u * T, jc = j < nL ? j : 0;
                const bool ok = translate(jc, x[u], y[u], z[u], w[u], rx[u], ry[u], rz[u], rb[u]) && j < nL;
                rx[u] = ok ? rx[u] : 0.0, ry[u] = ok ? ry[u] : 0.0, rz[u] = ok ? rz[u] : 0.0, rb[u] = ok ? rb[u] : -1.0;
            }
        };
        auto stage = [&](int j0, const double* rx, const double* ry, const double* rz, const double* rb) {
#pragma unroll
            for (int u = 0; u < kU; u++) {
                const int j = j0 + u * T;
                if (j < nL) Sx_[j] = rx[u], Sy_[j] = ry[u], Sz_[j] = rz[u], Sb_[j] = rb[u];
            }
        };
        double rx[kU], ry[kU], rz[kU], rb[kU];
        if (first_pass) {
            // the prologue's rows, then the rest from memory
            prep(tid, px, py, pz, pw, rx, ry, rz, rb);
            if (staged) stage(tid, rx, ry, rz, rb);
            eval(tid, rx, ry, rz, rb);
            for (int j0 = tid + kU * T; j0 < nL; j0 += kU * T) {
                double x[kU], y[kU], z[kU], w[kU];
#pragma unroll
                for (int u = 0; u < kU; u++) fetch_row(j0 + u * T < nL ? j0 + u * T : 0, x[u], y[u], z[u], w[u]);
                prep(j0, x, y, z, w, rx, ry, rz, rb);
                if (staged) stage(j0, rx, ry, rz, rb);
                eval(j0, rx, ry, rz, rb);
            }
        } else if (rows_in_lds) {
            for (int j0 = tid; j0 < nL; j0 += kU * T) {
#pragma unroll
                for (int u = 0; u < kU; u++) {
                    const int j = j0 + u * T, jc = j < nL ? j : 0;
                    rx[u] = Sx_[jc], ry[u] = Sy_[jc], rz[u] = Sz_[jc], rb[u] = (j < nL) ? Sb_[jc] : 1e300;  // (a slot past the end: slack -> -inf guard below)
                    if (!(j < nL)) rx[u] = ry[u] = rz[u] = 0.0, rb[u] = -1.0;
                }
                eval(j0, rx, ry, rz, rb);
            }
        } else {
            for (int j0 = tid; j0 < nL; j0 += kU * T) {
                double x[kU], y[kU], z[kU], w[kU];
#pragma unroll
                for (int u = 0; u < kU; u++) fetch_row(j0 + u * T < nL ? j0 + u * T : 0, x[u], y[u], z[u], w[u]);
                prep(j0, x, y, z, w, rx, ry, rz, rb);
                eval(j0, rx, ry, rz, rb);
            }
        }
        first_pass = false;
        DAS_T(8);
        for (int r0 = tid; r0 < NPAIR; r0 += 2 * T) {  // two at a time: their LDS round trips overlap
            const int r1 = r0 + T, r1c = r1 < NPAIR ? r1 : r0;
            const int pk0 = pix_[r0], pk1 = pix_[r1c];
            const double lo0 = plo_[r0], hi0 = phi_[r0], lo1 = plo_[r1c], hi1 = phi_[r1c];
            const double d0 = pair_val(pk0, c_), d1 = pair_val(pk1, c_);
            const bool on0 = (pk0 >> 24) != 0, on1 = (pk1 >> 24) != 0 && r1 < NPAIR;
            see(on0 ? d0 - lo0 : 1.0, nL + 2 * r0);
            see(on0 ? hi0 - d0 : 1.0, nL + 2 * r0 + 1);
            see(on1 ? d1 - lo1 : 1.0, nL + 2 * r1);
            see(on1 ? hi1 - d1 : 1.0, nL + 2 * r1 + 1);
        }
        DAS_T(9);
    };
    // the workgroup's (slack, id) minimum out of the wavefronts' (buffer rb_: values at [0, 4), ids as ints at [4, 6))
    auto pass_combine = [&](const double* rb_, double& bv, int& bi) {
        bv = rb_[0], bi = reinterpret_cast<const int*>(rb_ + 4)[0];
#pragma unroll
        for (int w = 1; w < NW; w++) {
            const double ov = rb_[w];
            const int oi = reinterpret_cast<const int*>(rb_ + 4)[w];
            if (ov < bv || (ov == bv && oi < bi)) bv = ov, bi = oi;
        }
    };
    auto pass = [&](double& best, int& bid) {
        const bool was_first = first_pass;
        double bv;
        int bi;
        pass_local(bv, bi);
        wave_argmin(bv, bi);
        if constexpr (NW > 1) {
            double* const rb_ = red_ + 24 * par;
            par ^= 1;
            if (lane == 0) {
                rb_[wv] = bv;
                reinterpret_cast<int*>(rb_ + 4)[wv] = bi;
            }
            LSCQP_DAS_BARRIER();
            pass_combine(rb_, bv, bi);
        } else if (staged && was_first) {
            LSCQP_DAS_BARRIER();  // the staged rows are read by other lanes from now on
        }
        if (staged) rows_in_lds = true;
        best = bv, bid = bi;
    };

    // ---- the small factor: S = A'C A (k x k, SPD), S = Lm Lm', rows owned by the lanes of wavefront 0 -------------------------------------
    const double* Cm = Cg;  // column cp = Cm + cp * P (symmetric); the LDS copy once a step needs it
    // The small system S = A'C A of the active rows is carried as J = L^-1, the INVERSE of its Cholesky factor (lower triangular, zeros kept
    // above the diagonal): S^-1 = J'J, so r = S^-1 v is two matrix-vector products without a dependent chain (a substitution through L is
    // 2k dependent broadcast-multiply-subtract steps), a joining row appends the row (-r' , 1) / sqrt(a_p'w_p - v'r) -- r is this step's --
    // and a leaving row costs k - l rotations of row pairs.  Lane i of wavefront 0 = row i (products) / column i (rotations).
    //
    // Row l leaves: rotations of the rows (l, l + 1), (l + 1, l + 2), ... push column l's content into the last row, which is dropped with
    // the column (Goldfarb-Idnani's downdate on J).  In place: the lane of column c writes column c - [c > l] of row j after every lane
    // has read rows j and j + 1 (LDS operations of one wavefront execute in order).  false on a vanishing pivot.
    auto factor_remove = [&](int k_, int l_) -> bool {
        bool ok = true;
        if (wv == 0) {
            const int k = __builtin_amdgcn_readfirstlane(k_), l = __builtin_amdgcn_readfirstlane(l_);
            const bool mine = lane < k;
            const int c = mine ? lane : 0;
            const int cn = c - (c > l ? 1 : 0);
            double carry = Jm_[l * LDL + c];
            double a = Jm_[l * LDL + l];
            for (int j = l; j < k - 1; j++) {
                const double x = Jm_[(j + 1) * LDL + c];
                const double b = Jm_[(j + 1) * LDL + l];
                const double r2 = a * a + b * b;
                if (!(r2 > 1e-280)) ok = false;
                const double ir = rsqrt(fmax(r2, 1e-300));
                const double ca = a * ir, sb = b * ir;
                const double fin = ca * x - sb * carry;  // the new row j: nothing left in column l, a positive diagonal
                carry = ca * carry + sb * x;
                a = r2 * ir;
                if (mine && lane != l) Jm_[j * LDL + cn] = fin;
            }
            if (mine) Jm_[(k - 1) * LDL + lane] = 0.0, Jm_[lane * LDL + k - 1] = 0.0;  // (J is zero outside its k x k block: solve_factor reads windows)
            LSCQP_DAS_WAVE_SYNC();
        }
        return ok;
    };
    // r = S^-1 v = J'(J v) (wavefront 0; lane j holds v_j on entry and r_j on return, also left in r_; yy = v'S^-1 v = |J v|^2 if asked for).
    // The factor's entries are requested a window of eight columns ahead so that no LDS round trip sits between the multiply-adds; the
    // loads carry no condition (J is zero outside its block, the index is clamped into the array): a test around a load makes the compiler
    // wait for each one in turn.
    auto solve_factor = [&](int k_, double vi, double* yy) -> double {
        const int k = __builtin_amdgcn_readfirstlane(k_);
        const bool mine = lane < k;
        const int ll = min(lane, kmax - 1);
        vi = mine ? vi : 0.0;
        double yi = 0.0, ri = 0.0;
        for (int j0 = 0; j0 < k; j0 += 8) {  // y = J v (row ll of J)
            double Jr[8];
#pragma unroll
            for (int t_ = 0; t_ < 8; t_++) Jr[t_] = Jm_[ll * LDL + min(j0 + t_, kmax)];  // (column kmax: always zero)
#pragma unroll
            for (int t_ = 0; t_ < 8; t_++) yi += Jr[t_] * lscqp::bcast(vi, j0 + t_);  // (k <= 32: the lane index stays below 64; lanes >= k hold 0)
        }
        yi = mine ? yi : 0.0;
        if (yy) *yy = wave_sum(yi * yi);
        for (int j0 = 0; j0 < k; j0 += 8) {  // r = J'y (column ll of J)
            double Jc[8];
#pragma unroll
            for (int t_ = 0; t_ < 8; t_++) Jc[t_] = Jm_[min(j0 + t_, kmax - 1) * LDL + ll];  // (a clamped row meets y = 0)
#pragma unroll
            for (int t_ = 0; t_ < 8; t_++) ri += Jc[t_] * lscqp::bcast(yi, j0 + t_);
        }
        ri = mine ? ri : 0.0;
        if (lane < kmax + 4) r_[lane] = ri;  // (zeros behind the active rows: the update of c reads a window of four without a test)
        return ri;
    };
    // c_[e] = base[e] (or c_[e]) + sum_j wts[j] W_j[e] over the active rows
    auto add_columns = [&](int k, const double* wts, const double* base) {
        for (int e = tid; e < NX; e += T) {
            double a = base ? base[e] : c_[e];
            for (int j = 0; j < k; j++) a += wts[j] * W_[(size_t)j * NX + e];
            c_[e] = a;
        }
    };

    // ---- verification + objective, one reduction: reduced stationarity T'(Hx c + fx - A'u) scaled as lscqp_info.res_dual; objective exactly
    // as cplex.getObjValue() reports it (as lscqp_kernel.hpp).  Every z thread evaluates the <= 4 control-point rows of Hx it needs itself.
    const int NZA = 3 * (M - 1) + (es ? 1 : 3);
    auto finish_local = [&](int k, double& rd, double& gs, double& part) {
        if (k > 0) {  // A'u, per control point
            for (int e = tid; e < NX; e += T) lam_[e] = 0.0;
            LSCQP_DAS_BARRIER();
            if (tid == 0) {
                for (int j = 0; j < k; j++)
#pragma unroll
                    for (int t_ = 0; t_ < 3; t_++) {
                        const int en = aint_[4 * j + 1 + t_];
                        lam_[ent_axis(en) * P + ent_cp(en)] += u_[j] * acoef_[3 * j + t_];
                    }
            }
            LSCQP_DAS_BARRIER();
        }
        rd = 0.0, gs = 0.0;
        for (int zi = tid; zi < dim * NZA; zi += T) {
            const int kx = zi / NZA, a = zi - kx * NZA;
            const bool last = es && a == 3 * (M - 1);
            const int m = last ? M - 1 : a / 3, j = last ? 0 : a % 3;
            const double c1 = Hd->v0[kx] * dt * 0.2;
            const double c2 = Hd->a0[kx] * dt * dt * 0.05 + 2.0 * c1;
            const double gk = Hd->goal[kx] - org[kx];
            // the six rows of Hx of one segment on this axis: g = Hx c + fx, g0 = Hx cfix + fx, lm = A'u (constant indices: registers)
            auto seg = [&](int mm, double* g, double* g0, double* lm) {
                const double* cc = &c_[kx * P + 6 * mm];
                const double v0 = cc[0], v1 = cc[1], v2 = cc[2], v3 = cc[3], v4 = cc[4], v5 = cc[5];
                lscqp::static_for<0, 6>([&](auto Ic) {
                    constexpr int i = decltype(Ic)::value;
                    g[i] = q2s * (KQ(i, 0) * v0 + KQ(i, 1) * v1 + KQ(i, 2) * v2 + KQ(i, 3) * v3 + KQ(i, 4) * v4 + KQ(i, 5) * v5);
                    g0[i] = (mm == 0) ? q2s * (KQ(i, 1) * c1 + KQ(i, 2) * c2) : 0.0;
                    lm[i] = (k > 0) ? lam_[kx * P + 6 * mm + i] : 0.0;
                });
                if (mm >= M - ts) {
                    g[5] += wt2 * (v5 - gk);
                    g0[5] += -wt2 * gk;
                }
            };
            double g[6], g0[6], lm[6];
            seg(m, g, g0, lm);
            double cf, cg, c0;  // T' of: full residual, gradient, gradient at the fixed part
            if (last) {
                cf = (g[3] - lm[3]) + (g[4] - lm[4]) + (g[5] - lm[5]), cg = g[3] + g[4] + g[5], c0 = g0[3] + g0[4] + g0[5];
            } else {
                const double gs_ = j == 0 ? g[3] : j == 1 ? g[4] : g[5], ls_ = j == 0 ? lm[3] : j == 1 ? lm[4] : lm[5], g0s = j == 0 ? g0[3] : j == 1 ? g0[4] : g0[5];
                cf = gs_ - ls_, cg = gs_, c0 = g0s;
            }
            if (m + 1 < M) {  // (c0, c1, c2) of the next segment = TB (c3, c4, c5) of this one, TB = [[0,0,1],[0,-1,2],[1,-4,4]]
                seg(m + 1, g, g0, lm);
                const double w0 = (j == 2) ? 1.0 : 0.0, w1 = (j == 1) ? -1.0 : (j == 2) ? 2.0 : 0.0, w2 = (j == 0) ? 1.0 : (j == 1) ? -4.0 : 4.0;
                cf += w0 * (g[0] - lm[0]) + w1 * (g[1] - lm[1]) + w2 * (g[2] - lm[2]);
                cg += w0 * g[0] + w1 * g[1] + w2 * g[2];
                c0 += w0 * g0[0] + w1 * g0[1] + w2 * g0[2];
            }
            rd = fmax(rd, fabs(cf));
            gs = fmax(gs, fmax(fabs(cg), fabs(c0)));
        }
        part = 0.0;
        // (the objective's threads sit in the second wavefront when there is one: its arithmetic runs beside the stationarity rows' instead of behind them)
        for (int lv = tid - (NW > 1 ? 64 : 0); lv < dim * M; lv += T) {
            if (lv < 0) continue;
            const int kx = lv / M, m = lv - kx * M;
            const double* cc = &c_[kx * P + 6 * m];
            const double j0 = (cc[3] - cc[0]) - 3.0 * (cc[2] - cc[1]);
            const double j1 = (cc[4] - cc[1]) - 3.0 * (cc[3] - cc[2]);
            const double j2 = (cc[5] - cc[2]) - 3.0 * (cc[4] - cc[3]);
            const double quad = 0.2 * (j0 * j0 + j2 * j2) + (2.0 / 15.0) * j1 * j1 + 0.2 * (j0 * j1 + j1 * j2) + (1.0 / 15.0) * j0 * j2;
            double pp = 0.5 * q2s * 3600.0 * quad;
            const double ok_ = org[kx];
            double corr = 0;
            const double s0 = cc[0] + ok_, s1 = cc[1] + ok_, s2 = cc[2] + ok_, s3 = cc[3] + ok_, s4 = cc[4] + ok_, s5 = cc[5] + ok_;
#pragma unroll 1
            for (int i = 0; i < 6; i++) {  // (a row of the term at a time: unrolled, its 36 entries would be requested -- and held in registers -- at once)
                const double* dr = dq_ + 6 * i;
                double r = 0;
                r += dr[0] * s0, r += dr[1] * s1, r += dr[2] * s2, r += dr[3] * s3, r += dr[4] * s4, r += dr[5] * s5;
                corr += r * (cc[i] + ok_);
            }
            pp += cls.w_c * corr;
            const double dgoal = cc[5] - (Hd->goal[kx] - ok_);
            pp += (m >= M - ts) ? cls.w_t * dgoal * dgoal : 0.0;
            part += pp;
        }
    };
    // (buffer rb_: the wavefronts' stationarity maxima at [8, 12), gradient scales at [12, 16), objective parts at [16, 20))
    auto finish_combine = [&](const double* rb_, double& rd, double& gs, double& part) {
        rd = rb_[8], gs = rb_[12], part = rb_[16];
#pragma unroll
        for (int w = 1; w < NW; w++) rd = fmax(rd, rb_[8 + w]), gs = fmax(gs, rb_[12 + w]), part += rb_[16 + w];  // (fixed order: reproducible)
    };
    auto finish_verdict = [&](double rd, double gs, double part, double& res_d, double& obj) {
        res_d = (part == part && fabs(part) < 1e300) ? rd / fmax(1.0, gs) : 1e300;  // (a non-finite objective is not a pass)
        obj = part;
    };
    auto finish = [&](int k, double& res_d, double& obj) {
        double rd, gs, part;
        finish_local(k, rd, gs, part);
        lscqp::wave_reduce3<lscqp::OpMax, lscqp::OpMax, lscqp::OpSum>(rd, gs, part);
        if constexpr (NW > 1) {
            double* const rb_ = red_ + 24 * par;
            par ^= 1;
            if (lane == 0) rb_[8 + wv] = rd, rb_[12 + wv] = gs, rb_[16 + wv] = part;
            LSCQP_DAS_BARRIER();
            finish_combine(rb_, rd, gs, part);
        }
        finish_verdict(rd, gs, part, res_d, obj);
    };

    // ---- the loop ---------------------------------------------------------------------------------------------------------------------
    int k = 0, steps = 0, why = LSCQP_DAS_WHY_VERIFICATION;
    bool polished = false, solved = false, haveC = false, haveJ = false, proven = false;
    double res_p = 0.0, res_d = 0.0, obj = 0.0;
    // the result: control points in the world frame
    auto write_out = [&]() {
        for (int e = tid; e < NX; e += T) x_out[q * NX + e] = c_[e] + org[fdiv(e, iP)];
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
    };
    // PEEL (batches beyond two workgroups per CU; the lean form): the first look stands in front of the loop of steps and a quiet instance
    // leaves the kernel from it.  What the loop keeps invariant -- addresses, reciprocals, spilled scalars: some 360 instructions of code a
    // quiet instance never reaches -- the compiler prepares in front of the loop, and with the first look inside the loop in front of that
    // too: 4096 quiet instances run 4 % faster peeled, 1024 x M10 x 40 4.5 %.  Not for the small batches: there ONE instance with a step
    // sets the launch's time, and it runs 2.5 % faster when the pass and the verification it repeats are the code it has just run.
    double best;
    int bid;
    bool peeled = false;
    if constexpr (PEEL || SCREEN) {
        pass(best, bid);
        DAS_T(2);  // first pass
        if (!(best < -kTolP)) {
            res_p = fmax(0.0, -best);
            finish(0, res_d, obj);
            DAS_T(4);  // verification + objective
            if (res_d <= kTolD) write_out();
            else hand_over(0, LSCQP_DAS_WHY_VERIFICATION);  // (the tables' rounding, never seen; the interior-point kernel solves the instance)
            DAS_T(7);  // epilogue
            DAS_T_FLUSH();
            return;
        }
        peeled = true;
    }
    for (;;) {
        if (!((PEEL || SCREEN) && peeled)) {  // (the peeled forms come with their first look taken)
            pass(best, bid);
            DAS_T(steps == 0 ? 2 : 3);  // first pass / later passes
        }
        peeled = false;
        if (!(best < -kTolP)) {
            res_p = fmax(0.0, -best);
            finish(k, res_d, obj);
            DAS_T(4);  // verification + objective
            if (res_d <= kTolD) {
                solved = true;
                break;
            }
            if (polished || k == 0) break;  // (never seen on the bench's classes; the interior-point kernel then solves the instance)
            // POLISH (a stationarity residual above the bar: rounding accumulated over many steps): the point rebuilt from its
            // multipliers, c = c_u + sum u_j C a_j -- stationary up to the table's rounding -- and one refinement of the multipliers that
            // puts the active rows back at zero slack:  rho = h_A - A c,  du = S^-1 rho,  u += du,  c += sum du_j C a_j.  Then every row
            // is looked at again.
            LSCQP_DAS_BARRIER();
            add_columns(k, u_, cu_);
            LSCQP_DAS_BARRIER();
            if (wv == 0) {
                const double rho = (lane < k) ? arhs_[lane] - row_dot(&aint_[4 * lane + 1], &acoef_[3 * lane], c_) : 0.0;
                const double du = solve_factor(k, rho, nullptr);
                // multipliers >= 0 is the one KKT condition the passes do not look at again: a refined multiplier below zero by more than
                // rounding is not this phase's to return -- the interior-point kernel solves the instance; rounding-size negatives are zero
                const double un = (lane < k) ? u_[lane] + du : 0.0;
                const double umax = wave_max(fabs(un));
                const double neg = wave_max((lane < k && un < -1e-12 * umax) ? 1.0 : 0.0);
                if (lane < k) u_[lane] = fmax(un, 0.0);
                if (lane == 0) ctl_[7] = neg;
            }
            LSCQP_DAS_BARRIER();
            if (ctl_[7] != 0.0) {  // (uniform; not solved: handed over below)
                why = LSCQP_DAS_WHY_MULTIPLIER;
                break;
            }
            add_columns(k, r_, nullptr);
            LSCQP_DAS_BARRIER();
            polished = true;
            continue;
        }
        polished = false;
        if constexpr (SCREEN) break;  // (a violated row: the full form's)
        if (k >= kmax) {  // more active rows than this launch holds: the interior-point kernel's
            why = LSCQP_DAS_WHY_ROWS;
            break;
        }
        if (!haveJ) {  // (before the first step; by wavefront 0, the only one that touches J: in order with its own use)
            if (wv == 0)
                for (int e = lane; e < kmax * LDL; e += 64) Jm_[e] = 0.0;
            haveJ = true;
        }
        if (cacheC && !haveC) {  // the table of this instance's ts in LDS from the first step on (every step reads a few of its columns)
            bool copied = false;
            if constexpr (kCPre > 0) {
                if (c_prefetched) {
#pragma unroll
                    for (int i = 0; i < kCPre; i++) {
                        const int e = tid + i * T;
                        if (e < P * P) Cc_[e] = cpre[i];
                    }
                    copied = true;
                }
            }
            if (!copied)
                for (int e = tid; e < P * P; e += T) Cc_[e] = Cg[e];
            haveC = true;
            Cm = Cc_;
            LSCQP_DAS_BARRIER();  // (every thread reads columns other threads copied)
        }
        // ---- the candidate row p = bid, slot kmax of the descriptors ----
        Row Rp;
        decode(bid, Rp);
        if (tid == 0) {
            aint_[4 * kmax] = bid;
            for (int t = 0; t < 3; t++) aint_[4 * kmax + 1 + t] = Rp.ent[t], acoef_[3 * kmax + t] = Rp.coef[t];
            arhs_[kmax] = Rp.rhs;
            ctl_[3] = 0.0;  // the candidate's multiplier so far
        }
        // w_p = C a_p goes into slot k of W.  Wavefront 0's decision of the first partial step does not read it -- a_p'C a_p comes straight from
        // the table, v_j = a_j'C a_p = a_p'w_j from the columns the active rows already have -- so the OTHER wavefronts compute w_p while
        // wavefront 0 decides, and one barrier serves both (one wavefront per QP: first w_p, then the decision).
        auto compute_wp = [&](int first_thread, int n_threads) {
            for (int e = tid - first_thread; e < NX; e += n_threads) {
                if (e < 0) continue;
                const int kx = fdiv(e, iP), cp = e - kx * P;
                W_[(size_t)k * NX + e] = ccol(Rp.ent, Rp.coef, kx, cp, Cm, P);
            }
        };
        if constexpr (NW == 1) compute_wp(0, T);
        DAS_T(5);  // candidate: decode, table copy (one wavefront: w_p)
        double spp = 0.0;  // a_p'C a_p (wavefront 0)
        bool stop = false, first_step = true;
        for (;;) {  // partial steps until p has joined the set
            steps++;
            if (steps > max_steps) {
                stop = true;
                why = LSCQP_DAS_WHY_STEPS;
                break;
            }
            if (NW > 1 && first_step && wv != 0) compute_wp(64, T - 64);
            // Wavefront 0 decides the step: v = A'w_p, r = S^-1 v, curvature a_p'w_p - v'r, dual bound t1, primal length t2.
            if (wv == 0 && __builtin_amdgcn_readfirstlane(k) == 0) {
                // the FIRST active row (most stepping instances of a plan never hold a second): nothing to solve, no row can leave -- the general
                // decision below with k = 0, minus its two reductions and its solve; the same values to the bit
                if (first_step) spp = cdot(Rp.ent, Rp.coef, Rp.ent, Rp.coef, Cm, P);
                const double sp = row_dot(Rp.ent, Rp.coef, c_) - Rp.rhs;
                const double t = (spp > 1e-12 * spp) ? -sp / spp : 1e300;
                const int kind = (t < 1e299) ? 1 : 0;
                if (lane == 0) {
                    if (kind == 1) Jm_[0] = rsqrt(spp), u_[0] = ctl_[3] + t;
                    ctl_[0] = t;
                    ctl_[1] = (double)kind;
                    ctl_[2] = 0.0;
                    ctl_[3] += t;
                    ctl_[5] = (t < 1e299) ? 1.0 : 0.0;
                }
            } else if (wv == 0) {
                if (first_step) spp = cdot(Rp.ent, Rp.coef, Rp.ent, Rp.coef, Cm, P);
                const double vj = (lane < k) ? row_dot(Rp.ent, Rp.coef, W_ + (size_t)lane * NX) : 0.0;
                DAS_T(13);
                double yy;
                const double ri = solve_factor(k, vj, &yy);
                DAS_T(14);
                const double curv = spp - yy;  // a_p'w_p - v'S^-1 v
                const double sp = row_dot(Rp.ent, Rp.coef, c_) - Rp.rhs;
                const double t2 = (curv > 1e-12 * spp) ? -sp / curv : 1e300;
                double t1 = (lane < k && ri > 0.0) ? u_[lane] / ri : 1e300;
                int l = lane;
                wave_argmin(t1, l);
                const double t = fmin(t1, t2);
                int kind;  // 0: no step exists (hand over); 1: p joins; 2: row l leaves; 3: no step exists and the row is violated beyond doubt: no point satisfies the rows
                if (!(t < 1e299)) kind = (sp < -1e-6) ? 3 : 0;
                else if (t2 <= t1) kind = 1;
                else kind = 2;
                DAS_T(15);
                if (kind == 1 || kind == 2) {
                    if (lane < k) u_[lane] = fmax(0.0, u_[lane] - t * ri);
                    if (kind == 1) {  // one more row of J: (-r', 1) / sqrt(curv); the column above its diagonal entry is zero
                        const double idl = rsqrt(curv);
                        if (lane < k) Jm_[k * LDL + lane] = -ri * idl, Jm_[lane * LDL + k] = 0.0;
                        if (lane == 0) Jm_[k * LDL + k] = idl, u_[k] = ctl_[3] + t;
                    }
                }
                if (lane == 0) {
                    ctl_[0] = t;
                    ctl_[1] = (double)kind;
                    ctl_[2] = (double)l;
                    ctl_[3] += t;
                    ctl_[5] = (t2 < 1e299) ? 1.0 : 0.0;  // a primal step is taken
                    ctl_[7] = -sp;
                }
            }
            first_step = false;
            DAS_T(10);  // the step's decision (wavefront 0)
            LSCQP_DAS_BARRIER();
            const double t = ctl_[0];
            const int kind = __builtin_amdgcn_readfirstlane((int)ctl_[1]), l = __builtin_amdgcn_readfirstlane((int)ctl_[2]);  // (uniform: scalar loop bounds)
            if (kind == 0 || kind == 3) {
                stop = true;
                why = LSCQP_DAS_WHY_NO_STEP;
                proven = kind == 3;
                break;
            }
            // c += t (w_p - sum r_j w_j)   (r_ holds this step's r); a leaving row closes the gap in W on the way (the candidate moves down too)
            const int ks = __builtin_amdgcn_readfirstlane(k);  // (uniform by construction; said so: scalar loop bounds)
            const bool primal = __builtin_amdgcn_readfirstlane((int)ctl_[5]) != 0;
            for (int e = tid; e < NX; e += T) {
                if (primal) {
                    double a = W_[(size_t)ks * NX + e];
                    for (int j0 = 0; j0 < ks; j0 += 4) {
                        double wj[4], rj[4];
#pragma unroll
                        for (int t_ = 0; t_ < 4; t_++)  // (past the end: the candidate's column with r_'s zero)
                            wj[t_] = W_[(size_t)min(j0 + t_, ks) * NX + e], rj[t_] = r_[j0 + t_];
#pragma unroll
                        for (int t_ = 0; t_ < 4; t_++) a -= rj[t_] * wj[t_];
                    }
                    c_[e] += t * a;
                }
                if (kind == 2) {  // (every thread its own elements: the copies of one element are ordered, those of different elements independent)
                    double nxt = W_[(size_t)(l + 1) * NX + e];
                    for (int j = l; j < ks; j++) {
                        const double cur = nxt;
                        nxt = W_[(size_t)min(j + 2, ks) * NX + e];
                        W_[(size_t)j * NX + e] = cur;
                    }
                }
            }
            DAS_T(11);  // the step itself: c, W
            if (kind == 1) {
                if (tid == 0) {
                    for (int t_ = 0; t_ < 4; t_++) aint_[4 * k + t_] = aint_[4 * kmax + t_];
                    for (int t_ = 0; t_ < 3; t_++) acoef_[3 * k + t_] = acoef_[3 * kmax + t_];
                    arhs_[k] = arhs_[kmax];
                }
                k++;
                LSCQP_DAS_BARRIER();
                break;
            }
            // row l leaves: close the gap in descriptors and multipliers (lane j takes slot j + 1's: every lane reads before any lane writes),
            // downdate the factor (wavefront 0)
            if (wv == 0) {
                const bool mv = lane >= l && lane + 1 < k;
                const int from = mv ? lane + 1 : 0;
                int ai[4];
                double ac[3];
#pragma unroll
                for (int t_ = 0; t_ < 4; t_++) ai[t_] = aint_[4 * from + t_];
#pragma unroll
                for (int t_ = 0; t_ < 3; t_++) ac[t_] = acoef_[3 * from + t_];
                const double ah = arhs_[from], uu = u_[from];
                LSCQP_DAS_WAVE_SYNC();
                if (mv) {
#pragma unroll
                    for (int t_ = 0; t_ < 4; t_++) aint_[4 * lane + t_] = ai[t_];
#pragma unroll
                    for (int t_ = 0; t_ < 3; t_++) acoef_[3 * lane + t_] = ac[t_];
                    arhs_[lane] = ah, u_[lane] = uu;
                }
            }
            const bool okf = factor_remove(k, l);
            k--;
            if (wv == 0 && lane == 0) ctl_[6] = okf ? 0.0 : 1.0;
            LSCQP_DAS_BARRIER();
            DAS_T(12);  // a leaving row: descriptors, factor
            if (ctl_[6] != 0.0) {
                stop = true;
                why = LSCQP_DAS_WHY_PIVOT;
                break;
            }
        }
        DAS_T(6);  // the partial steps of the candidate
        if (stop) break;
    }
    if (!solved) {
        if (proven) infeasible_out(steps, ctl_[7]);
        else hand_over(steps, why);
        DAS_T_FLUSH();
        return;
    }

    write_out();
    DAS_T(7);  // epilogue
    DAS_T_FLUSH();
}

}  // namespace lscqp_das

// ---- host side ------------------------------------------------------------------------------------------------------------------------

// The tables of a class: for ts = 1 .. M  [U1 | U2 | G1 | C], C = T (T'Hx T)^-1 T' in extended precision (cond(T'Hx T) ~ 2e5 .. 3e6).
// Returns the number of doubles written (M * table_stride(M)); out may be NULL to ask for the size.
// The two-sided rows of a class, as far as they do not depend on the instance (the order and the ids of the kernel's table: intervals, velocity
// rows, acceleration rows, communication pairs): per row two ints -- the packed stencil the kernel keeps in LDS (type << 24 | e0 << 12 | e1)
// and what its bounds depend on (family | axis << 2 | segment << 4 | last control point of its segment << 8 | axis 2 of segment 0 << 9).
// The kernel reads them from behind the tables and the objective's rounding term instead of deriving them thread by thread with one
// branch per family.  Returns the number of rows (num_pairs); out == NULL: only that.
extern "C" size_t lscqp_das_build_pairs(int M, int dim, int comm_on, int32_t* out) {
    const int P = 6 * M, NX = dim * P, NCP = M * (M - 1) / 2;
    const int oV = NX, oA = oV + dim * 5 * M, oC = oA + dim * 4 * M, NPAIR = lscqp_das::num_pairs(M, dim);
    if (!out) return (size_t)NPAIR;
    for (int r = 0; r < NPAIR; r++) {
        int type = 0, e0 = 0, e1 = 0, fam = 0, k = 0, m = 0, last = 0;
        if (r < oV) {
            k = r / P;
            const int cp = r - k * P;
            m = cp / 6, last = cp % 6 == 5;
            type = cp >= 3 ? 1 : 0, e0 = r, fam = 0;
        } else if (r < oA) {
            const int s = r - oV;
            k = s / (5 * M);
            const int rr = s - k * 5 * M, i = rr % 5;
            m = rr / 5;
            type = (m == 0 && i < 2) ? 0 : 2, e0 = k * P + 6 * m + i, fam = 1;
        } else if (r < oC) {
            const int s = r - oA;
            k = s / (4 * M);
            const int rr = s - k * 4 * M, i = rr % 4;
            m = rr / 4;
            type = (m == 0 && i < 1) ? 0 : 3, e0 = k * P + 6 * m + i, fam = 2;
        } else {
            const int s = r - oC;
            k = s / NCP;
            const int ci = s - k * NCP;
            int uu = 1;
            while (uu * (uu + 1) / 2 <= ci) uu++;
            const int up = ci - uu * (uu - 1) / 2;
            type = comm_on ? 4 : 0, e0 = k * P + 6 * (up + 1), e1 = k * P + 6 * uu + 5, fam = 3, m = 0;
        }
        out[2 * r] = (type << 24) | (e0 << 12) | e1;
        out[2 * r + 1] = fam | (k << 2) | (m << 4) | (last << 8) | ((fam == 0 && k == 2 && m == 0) ? 1 << 9 : 0);
    }
    return (size_t)NPAIR;
}

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

// workgroups of the one-wavefront form a CU holds at once (registers and the LDS footprint of a launch with `kmax` active rows, no table
// copy, no staged rows), as the runtime computes it; 0 without a device
extern "C" int lscqp_das_blocks_per_cu(int M, int dim, int kmax, int rows_f32) {
    int nb = 0;
    const size_t lds = lscqp_das_lds_bytes(M, dim, kmax, 0, 0);
    const hipError_t e = rows_f32 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, lscqp_das::das_kernel<1, true, false, true>, 64, lds)
                                  : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, lscqp_das::das_kernel<1, false, false, true>, 64, lds);
    return e == hipSuccess ? nb : 0;
}

// Launch of the phase over a batch.  threads: 64, 128 or 256 per QP; kmax <= 32 active rows; stage_rows: LSC rows per instance kept in LDS
// after the first pass (0: re-read from L2 in every pass; an instance with more rows than that re-reads them too); cap: the obstacle
// capacity of the kernel instance that runs behind the phase (an instance beyond it is left to that kernel's LSCQP_STATUS_CAPACITY).
// screen bit 1: the first look inside the loop of steps (PEEL = false; four wavefronts only).  screen bit 0: the lean one-wavefront form first (unconstrained minimiser + one pass + verification at twice the occupancy), then the
// full form over what it left -- for batches that fill the chip, where most instances hold no active row at all.
extern "C" hipError_t lscqp_launch_das(const lscqp::DevClass* cls, int M, int dim, int es, int cap, int threads, int kmax, int max_steps, int cacheC,
                                       int stage_rows, int screen, const double* d_tab, int64_t n, const lscqp_header* hdr, const lscqp_row* rows,
                                       const uint64_t* row_offsets, const lscqp_box* sfc, const double* x_init, double* x_out, double* obj_out,
                                       int32_t* status_out, lscqp_info* info_out, hipStream_t stream) {
    if (kmax < 1 || kmax > lscqp_das::kMaxK || (threads != 64 && threads != 128 && threads != 256)) return hipErrorInvalidValue;
    const size_t lds = lscqp_das_lds_bytes(M, dim, kmax, cacheC, stage_rows);
    if (lds > lscqp::kMaxLdsBytes) return hipErrorInvalidValue;
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        for (const void* f : {reinterpret_cast<const void*>(lscqp_das::das_kernel<1, false, false, true>), reinterpret_cast<const void*>(lscqp_das::das_kernel<2, false, false, true>),
                              reinterpret_cast<const void*>(lscqp_das::das_kernel<4, false, false, true>), reinterpret_cast<const void*>(lscqp_das::das_kernel<1, true, false, true>),
                              reinterpret_cast<const void*>(lscqp_das::das_kernel<2, true, false, true>), reinterpret_cast<const void*>(lscqp_das::das_kernel<4, true, false, true>),
                              reinterpret_cast<const void*>(lscqp_das::das_kernel<4, false>), reinterpret_cast<const void*>(lscqp_das::das_kernel<4, true>),
                              reinterpret_cast<const void*>(lscqp_das::das_kernel<1, false, true>), reinterpret_cast<const void*>(lscqp_das::das_kernel<1, true, true>)}) {
            const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lscqp::kMaxLdsBytes);
            if (e != hipSuccess) return e;
        }
        attr_set[dev].store(true, std::memory_order_release);
    }
    if (n <= 0) return hipSuccess;
    const bool f32 = cls->rows_f32 != 0;
    int behind = 0;
    const bool loop_form = (screen & 2) != 0;  // (the first look inside the loop of steps: the four-wavefront form of small batches)
    if (screen & 1) {
        const size_t lds_s = lscqp_das_lds_bytes(M, dim, 1, 0, 0);  // (no active rows, no table copy, no staged rows)
#define LSCQP_DAS_SCREEN(F_)                                                                                                                                  \
    hipLaunchKernelGGL((lscqp_das::das_kernel<1, F_, true>), dim3((unsigned)n), dim3(64), lds_s, stream, *cls, M, dim, es, cap, 1, 0, 0, 0, 0, d_tab, n, hdr, rows, \
                       row_offsets, sfc, x_init, x_out, obj_out, status_out, info_out)
        if (f32) LSCQP_DAS_SCREEN(true); else LSCQP_DAS_SCREEN(false);
#undef LSCQP_DAS_SCREEN
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        behind = 1;
    }
#define LSCQP_DAS_LAUNCH(NW_, F_, PEEL_)                                                                                                                       \
    hipLaunchKernelGGL((lscqp_das::das_kernel<NW_, F_, false, PEEL_>), dim3((unsigned)n), dim3(64 * NW_), lds, stream, *cls, M, dim, es, cap, kmax, max_steps, cacheC, \
                       stage_rows, behind, d_tab, n, hdr, rows, row_offsets, sfc, x_init, x_out, obj_out, status_out, info_out)
    if (threads == 64) { if (f32) LSCQP_DAS_LAUNCH(1, true, true); else LSCQP_DAS_LAUNCH(1, false, true); }
    else if (threads == 128) { if (f32) LSCQP_DAS_LAUNCH(2, true, true); else LSCQP_DAS_LAUNCH(2, false, true); }
    else if (!loop_form) { if (f32) LSCQP_DAS_LAUNCH(4, true, true); else LSCQP_DAS_LAUNCH(4, false, true); }
    else { if (f32) LSCQP_DAS_LAUNCH(4, true, false); else LSCQP_DAS_LAUNCH(4, false, false); }
#undef LSCQP_DAS_LAUNCH
    return hipGetLastError();
}
