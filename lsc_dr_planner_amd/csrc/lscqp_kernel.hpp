// lscqp_kernel.hpp — batched primal-dual interior point for the LSC trajectory QP, gfx950 (CDNA4) only.
//
// Replaces the arithmetic of TrajOptimizer::solve (reference src/traj_optimizer.cpp:18-156: CPLEX) for the
// model TrajOptimizer::populatebyrow builds (src/traj_optimizer.cpp:216-514).  Not a translation of either:
//
//   * ONE WORKGROUP OF W WAVEFRONTS PER QP (W = 1: the throughput form; W = 2 / 4: row passes on several SIMDs for batches
//     that leave the chip idle); the kernel is issue/latency bound on fp64 VALU, so everything is organised to minimise the
//     instruction count of one Mehrotra iteration.  With W > 1: lane -> thread of the workgroup, wave reductions -> wave + LDS.
//     Reduced systems of up to 64 rows are factorised by every wavefront redundantly (uniform verdict on a failed pivot);
//     larger ones (M = 8, 9, 10 in 3-D: nz = 66 .. 90) and M = 10 in 2-D with W >= 2 use a NESTED DISSECTION over two wavefronts (Cfg::ND).
//     FT = float instantiates the mixed-precision form (float32 matrix / factor / substitutions, everything else fp64).
//   * The equality rows (:318-368, 502-511) are eliminated analytically: per axis the free variables are
//     z = (c3,c4,c5) of every segment (one scalar for the last segment under the LSC end stop);
//     (c0,c1,c2) of segment m+1 = TB (c3,c4,c5) of segment m, TB = [[0,0,1],[0,-1,2],[1,-4,4]].
//     nz = dim*(3M-2) <= 64: lane r owns row r of the reduced KKT matrix in registers.
//   * LSC rows (n.c >= b, one control point each): lane = (group g, control point cp); the lane's rows are the
//     obstacles o = g, g+G, ... of that control point.  Row constants (n, b) sit in LDS as SoA [obstacle][cp]
//     (conflict-free ds_read_b64), row STATE (s, lambda and the per-iteration residual / direction scalars) sits in
//     registers (NSLOT slots per lane), the per-control-point 3x3 blocks and x-space vectors are accumulated in
//     registers over the slots and flushed once per pass.
//   * The per-axis structured rows (merged interval bounds from world box / SFC / communication range, velocity and
//     acceleration differences, communication pairs) are two-sided rows in registers, one ROW TYPE per slot so a
//     wave instruction never mixes types.
//   * Reduced matrix: every lane builds its row from per-(axis,segment) 6x6 local blocks; LDL^T and the two
//     triangular solves run entirely in registers with v_readlane broadcasts of the pivot row.
//   * All arithmetic fp64 (FT = double), in coordinates translated to the agent's position.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/lscqp.h"

namespace lscqp {

struct DevClass {
    double dt, w_c, w_t, comm_range;
    double world_min[3], world_max[3];
    double q2s;     // 2 * w_c * pow(dt,-5): Q2[i][j] = q2s * KQ[i][j]  (closed form of src/traj_optimizer.cpp:163-178)
    double dQ[36];  // the reference's per-entry coefficient rounding, see lscqp_api.hip
    double tol;
    int max_iter;
    int use_sfc;
    int n_obs_max;  // number of obstacles the launch must accommodate (<= NSLOT * G of the instance)
    int rows_f32;   // lscqp_class_desc.row_format == LSCQP_ROWS_F32
    int rsfc;       // LSCQP_PLANNER_RSFC: z bounds of segment 0 are +-100, not the world box (src/traj_optimizer.cpp:255-258)
    int repair;     // second pass over a batch: only instances whose status_out is neither OPTIMAL nor CAPACITY are solved
                    // (3: the first interior-point pass behind the dual active-set phase: same skip, no REPAIRED flag)
    double warm_mu0, warm_s0;  // centring of an instance that comes with an initial trajectory (lscqp_class_desc.warm_start)
    double warm_net;           // > 0: first-step length below which such an instance returns to the (1e-3, 0.03) centring
    // Work distribution of a launch (round 4).  order: the k-th workgroup-slot of the launch solves instance order[k] (NULL: k) -- a
    // permutation of 0 .. n-1, longest expected work first (lscqp_order_by_work_device).  queue: a zeroed counter; when set, the launch
    // is PERSISTENT: gridDim.x workgroups (what the chip holds at once) take their first instance from blockIdx.x and every further
    // one from gridDim.x + atomicAdd(queue, 1) -- true list scheduling instead of the hardware's dispatch order (NULL: one instance
    // per workgroup, grid = n).
    const int32_t* order;
    int* queue;
    // round 6, the passes BEHIND the dual active-set phase (repair == 3, and the second pass, repair == 1) in the PERSIST instances: scan != 0 --
    // gridDim.x workgroups (at most what the chip holds) look through status_out themselves, 64 instances per workgroup and round trip (lane l:
    // instance base + l * gridDim.x), and solve the ones the phase marked LSCQP_STATUS_ITER_LIMIT (second pass: whatever is not finished).  On a batch the phase finished -- every BASELINE batch -- the launch is
    // gridDim.x workgroups that load one status each and leave, instead of n of them: 4096 x M5, 4.5 -> 1.7 us per call.  No list, no counter,
    // no atomics: which workgroup solves which instance is a function of the statuses alone.
    int scan;
};

// Q_base * dt^5 for n = 5, phi = 3 (integers)
__device__ __forceinline__ constexpr double KQ(int i, int j) {
    constexpr int q[6][6] = {{720, -1800, 1200, 0, 0, -120},  {-1800, 4800, -3600, 0, 600, 0}, {1200, -3600, 3600, -1200, 0, 0},
                             {0, 0, -1200, 3600, -3600, 1200}, {0, 600, 0, -3600, 4800, -1800}, {-120, 0, 0, 1200, -1800, 720}};
    return (double)q[i][j];
}
// TB rows: (c0,c1,c2) of the next segment in terms of (c3,c4,c5) of this one: [[0,0,1],[0,-1,2],[1,-4,4]].
// Written with selects, not a table: it is also evaluated with lane-dependent i, and a constexpr table indexed
// dynamically would be materialised in scratch memory.
__device__ __forceinline__ constexpr double TBc(int i, int j) {
    return i == 0 ? (j == 2 ? 1.0 : 0.0) : i == 1 ? (j == 0 ? 0.0 : j == 1 ? -1.0 : 2.0) : (j == 0 ? 1.0 : j == 1 ? -4.0 : 4.0);
}

// compile-time loop: guarantees constant register indices where '#pragma unroll' gives up on nested strided loops
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

__device__ __forceinline__ double bcast(double v, int lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float bcast(float v, int lane) {  // (mixed-precision instances: one v_readlane per value)
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
// Wave reductions with DPP moves instead of ds_bpermute butterflies (tools/ubench.hip: a 6-step bpermute butterfly
// costs ~460 cycles per value on a lone wavefront, the DPP form ~150): four row_shr steps build an inclusive scan
// inside each row of 16 lanes, row_bcast:15 / row_bcast:31 carry the row totals across, lane 63 ends up with the
// result and two v_readlane make it uniform.  Lanes without a source keep the identity in `old`.
struct OpSum {
    static __device__ __forceinline__ double id() { return 0.0; }
    static __device__ __forceinline__ double ap(double a, double b) { return a + b; }
};
struct OpMax {  // all reduced maxima are over non-negative quantities or use -1e300 as "nothing"
    static __device__ __forceinline__ double id() { return -1e300; }
    static __device__ __forceinline__ double ap(double a, double b) { return fmax(a, b); }
};
template <class Op, int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_step(double v) {
    const double idn = Op::id();
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(idn), __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(idn), __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return Op::ap(v, __hiloint2double(hi, lo));
}
__device__ __forceinline__ double bcast63(double v) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}
// three independent reductions advanced together so their DPP latencies overlap
template <class OpA, class OpB, class OpC>
__device__ __forceinline__ void wave_reduce3(double& a, double& b, double& c) {
#define LSCQP_DPP3(CTRL, MASK)        \
    a = dpp_step<OpA, CTRL, MASK>(a); \
    b = dpp_step<OpB, CTRL, MASK>(b); \
    c = dpp_step<OpC, CTRL, MASK>(c)
    LSCQP_DPP3(0x111, 0xf);  // row_shr:1
    LSCQP_DPP3(0x112, 0xf);  // row_shr:2
    LSCQP_DPP3(0x114, 0xf);  // row_shr:4
    LSCQP_DPP3(0x118, 0xf);  // row_shr:8
    LSCQP_DPP3(0x142, 0xa);  // row_bcast:15 into rows 1 and 3
    LSCQP_DPP3(0x143, 0xc);  // row_bcast:31 into rows 2 and 3
#undef LSCQP_DPP3
    a = bcast63(a);
    b = bcast63(b);
    c = bcast63(c);
}
template <class Op>
__device__ __forceinline__ double wave_reduce1(double v) {
    v = dpp_step<Op, 0x111, 0xf>(v);
    v = dpp_step<Op, 0x112, 0xf>(v);
    v = dpp_step<Op, 0x114, 0xf>(v);
    v = dpp_step<Op, 0x118, 0xf>(v);
    v = dpp_step<Op, 0x142, 0xa>(v);
    v = dpp_step<Op, 0x143, 0xc>(v);
    return bcast63(v);
}
__device__ __forceinline__ void wave_sum2_max1(double& a, double& b, double& c) { wave_reduce3<OpSum, OpSum, OpMax>(a, b, c); }
__device__ __forceinline__ void wave_max2(double& a, double& b) {
    double c = 0;
    wave_reduce3<OpMax, OpMax, OpMax>(a, b, c);
}
__device__ __forceinline__ void wave_max1_sum1(double& a, double& b) {
    double c = 0;
    wave_reduce3<OpMax, OpSum, OpSum>(a, b, c);
}
__device__ __forceinline__ double wave_sum(double v) { return wave_reduce1<OpSum>(v); }
__device__ __forceinline__ double wave_max(double v) { return wave_reduce1<OpMax>(v); }
// Reciprocals of the row passes (1/s, 1/lambda: weights and step ratios): v_rcp_f64 + ONE Newton step.  The weights only
// steer the direction (residuals and the objective are computed exactly), so the last bits do not matter: against two
// steps the solutions differ by 1e-12 m / 3e-15 relative in the objective, and ~90 reciprocals per lane and iteration get
// two instructions shorter (-2 % kernel time).  The pivots of the factorisation keep fast_rcp's two steps.
#ifndef LSCQP_ROW_RCP_STEPS
#define LSCQP_ROW_RCP_STEPS 1
#endif
__device__ __forceinline__ double row_rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    if (LSCQP_ROW_RCP_STEPS > 1) r = fma(fma(-d, r, 1.0), r, r);
    return r;
}
// A double pinned to two accumulation registers (AGPRs).  The row state (slack, multiplier of every row slot) lives
// across the whole iteration loop; left to the register allocator it is assigned to AGPRs anyway (the 256 VGPRs are
// needed by the matrix row and the pass temporaries) but shuttled in and out ~3x per use.  With the residence made
// explicit every access is exactly one v_accvgpr_read/write per dword, and the allocator never considers these values
// for VGPRs.
struct AD {
    int lo, hi;
    __device__ __forceinline__ double get() const {
        int l, h;
        asm("v_accvgpr_read_b32 %0, %1" : "=v"(l) : "a"(lo));
        asm("v_accvgpr_read_b32 %0, %1" : "=v"(h) : "a"(hi));
        return __hiloint2double(h, l);
    }
    __device__ __forceinline__ void set(double v) {
        asm("v_accvgpr_write_b32 %0, %1" : "=a"(lo) : "v"(__double2loint(v)));
        asm("v_accvgpr_write_b32 %0, %1" : "=a"(hi) : "v"(__double2hiint(v)));
    }
};
// 1/d to full fp64 precision: v_rcp_f64 + two Newton steps (5 instructions instead of an IEEE division sequence)
__device__ __forceinline__ double fast_rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;
}

__device__ __forceinline__ float fast_rcp(float d) {  // v_rcp_f32 (1 ulp) + one Newton step
    float r = __builtin_amdgcn_rcpf(d);
    return fmaf(fmaf(-d, r, 1.0f), r, r);
}
template <class FT>
__device__ __forceinline__ bool pivot_ok(FT d) {
    return std::is_same<FT, double>::value ? (d > (FT)1e-300) : (d > (FT)1e-30);
}

// Development aid: per-phase cycle totals (s_memtime), compiled in only with -DLSCQP_PHASE_TIMING.  The markers
// are scheduling barriers and drain the memory counters, so phases do not overlap in the instrumented build (it
// runs ~30 % slower than the product build, whose scheduler interleaves neighbouring phases).
// Cross-lane hand-off through LDS inside ONE wavefront: the LDS executes a wave's DS instructions in order, so no
// s_barrier is needed, but the compiler must neither reorder LDS accesses across the hand-off nor keep values in
// registers.  (__syncthreads() is not enough here: with a 64-thread workgroup hipcc elides it completely and at -O3
// then moves the uniform-address reads above the per-lane store.)
#define LSCQP_WAVE_LDS_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
// Keeps the instruction scheduler from hoisting the LDS loads of all row slots to the top of a pass (which
// multiplies the live registers by the slot count and forces scratch spills).
#define LSCQP_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

// Development aid: -DLSCQP_DEBUG_STOP=k leaves the iteration loop at stage k (bisecting device faults).
#ifndef LSCQP_FACT_HYBRID
#define LSCQP_FACT_HYBRID 5  // percent of a pivot row (beyond the first 4 entries) broadcast with v_readlane
#endif
#ifndef LSCQP_NEAR_CONFIRM
#define LSCQP_NEAR_CONFIRM 2  // consecutive iterations with the gap at its target and the stationarity below 1e-8 before a point is accepted
#endif
#ifndef LSCQP_CENTRALITY_GAMMA
#define LSCQP_CENTRALITY_GAMMA 1e-4
#endif
#ifndef LSCQP_SMU_FLOOR
#define LSCQP_SMU_FLOOR 0.2  // the corrector's centring target sigma mu is at least this fraction of the gap target's mu (see the corrector)
#endif
#ifndef LSCQP_DEBUG_STOP
#define LSCQP_DEBUG_STOP 0
#endif
#define LSCQP_STOP(k)                  \
    if (LSCQP_DEBUG_STOP == (k)) {     \
        status = 100 + (k);            \
        break;                         \
    }
// Position markers for the build's instruction count (lsc_dr_planner_amd/isa_work.py reads the work of one iteration off the
// machine code: SURVEY.md 8d's fp64-VALU figure): s_nop 13 = top of the iteration body, s_nop 12 = behind the convergence test
// (where the last, partial pass leaves), s_nop 14 = end of the body.  The compiler itself emits s_nop 0..7 only; ~14 idle cycles each.
// Nested-dissection instances: regions only SOME wavefronts of the workgroup execute are bracketed by s_nop 8 (wavefront 0 only) /
// s_nop 9 (wavefront 1 only) / s_nop 10 (wavefronts 0 and 1) ... s_nop 11 (end), so that the count knows who runs what.
// (-DLSCQP_NO_MARKERS builds the kernel without them -- an A/B library, LSCQP_AB=nomark; measured cost of carrying them: DESIGN.md section 4)
#ifdef LSCQP_NO_MARKERS
#define LSCQP_MARK(k) \
    do {              \
    } while (0)
#define LSCQP_MARKP(k) \
    do {               \
    } while (0)
#else
#define LSCQP_MARK(k) asm volatile("s_nop " #k)
// (the brackets of the wavefront-partial regions are fenced: without the fence the scheduler moved an end marker in FRONT of its
// region's register-only arithmetic, and the count took wavefront 0's separator solve for everybody's)
#define LSCQP_MARKP(k)                         \
    do {                                       \
        __builtin_amdgcn_sched_barrier(0);     \
        asm volatile("s_nop " #k);             \
        __builtin_amdgcn_sched_barrier(0);     \
    } while (0)
#endif
// Development aid: -DLSCQP_TRACE records, per iteration of the first LSCQP_TRACE_Q instances of a launch, the quantities the stopping
// tests look at (tools/floor_probe.py reads them back): [q][it][0..7] = max|r_p|, stationarity / scale, gap figure, mu, step length,
// sigma, exit code of the iteration (0 none, 1 optimal, 2 pivot breakdown, 3 stalled step, 4 infeasible), stationarity scale, then
// max w = lambda / s over the LSC rows, max lambda, max |dz| of the step, spare.
#ifdef LSCQP_TRACE
#ifndef LSCQP_TRACE_Q
#define LSCQP_TRACE_Q 1024
#endif
__device__ double lscqp_dbg_trace[LSCQP_TRACE_Q][64][12];
#define LSCQP_TR(slot, val)                                                                       \
    do {                                                                                          \
        if (lane == 0 && q < LSCQP_TRACE_Q && it < 64) lscqp_dbg_trace[q][it][slot] = (val);      \
    } while (0)
#else
#define LSCQP_TR(slot, val) \
    do {                    \
    } while (0)
#endif
#ifdef LSCQP_PHASE_TIMING
__device__ unsigned long long lscqp_dbg_cycles[16];
#define LSCQP_T(slot)                                                     \
    do {                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");       \
        const unsigned long long now_ = __builtin_readcyclecounter();    \
        if (lane == 0) atomicAdd(&lscqp_dbg_cycles[slot], now_ - tprev_); \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");       \
        tprev_ = __builtin_readcyclecounter();                            \
        __builtin_amdgcn_sched_barrier(0);                                \
    } while (0)
#else
#define LSCQP_T(slot) \
    do {              \
    } while (0)
#endif

// FB_ = bytes of the scalar the reduced matrix is held, factorised and substituted in: 8 (fp64), or 4 for the MIXED-PRECISION
// instances (BASELINE configs[4]: float32 LDL^T / substitutions steering an iteration whose residuals, multipliers, control
// points and stopping tests stay fp64; tools/proto_fp32.py).
#ifndef LSCQP_ND_MIN_M
#define LSCQP_ND_MIN_M 10  // nested dissection below 64 rows from this horizon on (measured: see Cfg::ND)
#endif
template <int M_, int DIM_, bool ES_, int NSLOT_, int W_ = 1, int FB_ = 8>
struct Cfg {
    static constexpr int M = M_, DIM = DIM_, NSLOT = NSLOT_, W = W_;
    static constexpr int T = 64 * W;  // threads per QP
    static constexpr bool ES = ES_;
    static constexpr int P = 6 * M;
    static constexpr int CP = P - 3;  // control points that carry LSC rows (all but the initial state, :404-406)
    static constexpr int NZA = 3 * (M - 1) + (ES ? 1 : 3);
    static constexpr int NZ = DIM * NZA;
    static constexpr int NX = DIM * P;  // x-space size
    static constexpr int G = (T / CP) > 0 ? (T / CP) : 1;  // lane groups of the LSC pass
    static constexpr int MAX_OBS = NSLOT * G;
    // two-sided per-axis rows, one type per register slot
    static constexpr int NV = DIM * 5 * M, NA = DIM * 4 * M, NCP = M * (M - 1) / 2, NC = DIM * NCP;
    static constexpr int SI = (NX + T - 1) / T, SV = (NV + T - 1) / T, SA = (NA + T - 1) / T, SC = (NC + T - 1) / T;
    static constexpr int NS2 = SI + SV + SA + SC;
    // weights of the two-sided rows in LDS: [interval NX][vel NV][acc NA][comm NC]
    static constexpr int OV = NX, OA = NX + NV, OC = NX + NV + NA, NOM = NX + NV + NA + NC;
    static constexpr int LDH = NZ | 1;
    // NESTED DISSECTION of the reduced system for classes with more than 64 rows (M = 8, 9, 10 in 3-D: nz = 66 .. 90).  The (c3, c4)
    // variables of a segment couple only to the neighbouring segments (through the C2 joins) and to the c5 variables; the c5
    // variables of one axis are all coupled (communication-pair rows, src/traj_optimizer.cpp:476-500).  Separator S = the (c3, c4)
    // of the middle segment MS + every c5; it cuts the rest into two independent banded blocks, L = (c3, c4) of segments
    // 0..MS-1 and R = (c3, c4) of segments MS+1..MLAST (M-2 under the end stop, else M-1), which two wavefronts eliminate CONCURRENTLY (each with its share of the
    // separator's Schur complement), then one wavefront factorises the separator.  Row length in registers: NB + NS instead of nz.
    static constexpr int MS = (M - 1) / 2;
    static constexpr int MLAST = ES ? M - 2 : M - 1;   // last segment that owns a (c3, c4) pair
    static constexpr int NB0 = 2 * DIM * MS;            // block L (wavefront 0): segments 0 .. MS-1, order (m, axis, j)
    static constexpr int NB1 = 2 * DIM * (MLAST - MS);  // block R (wavefront 1): segments MS+1 .. MLAST
    static constexpr int NS = 2 * DIM + DIM * M;        // separator: (c3, c4) of segment MS, then c5 of (segment, axis)
    // Wavefront 0 carries L and the whole separator (rows = columns = NB0 + NS).  Wavefront 1 carries R and, as accumulator rows /
    // columns, only the separator variables R can reach (NT1: (c3, c4) of segment MS and c5 of segments MS .. M-1), in the same
    // order with the unreachable ones left out -- that is what lets the end-stop-free 3-D classes (nz = 90, 81, 72) and unequal
    // blocks fit 64 lanes.
    // (only where R | S does not fit: M = 10 without the end stop, 30 + 36 lanes.  Everywhere else wavefront 1 keeps one accumulator per
    // separator variable -- compacting moved its columns to other registers and cost the round-2 instances 1.5 %)
    static constexpr bool ND_COMPACT = NB1 + NS > 64;
    static constexpr int NT1 = ND_COMPACT ? 2 * DIM + DIM * (M - MS) : NS;
    static constexpr int NR0 = NB0 + NS, NR1 = NB1 + NT1;
    static constexpr bool ND_FITS = NR0 <= 64 && NR1 <= 64 && MS >= 1 && MLAST > MS;
    // ... and wherever two wavefronts per QP are used and the two blocks come out equal (M = 6, 10): M = 10 in 2-D 56 dense pivots
    // -> 16 + 24 (-7 % on the forest10 replica), M = 6 in 3-D 48 -> 12 + 24
    static constexpr bool ND = (NZ > 64) || (W >= 2 && ES_ && FB_ == 8 && M >= LSCQP_ND_MIN_M && NB0 == NB1 && ND_FITS);
    static constexpr int BWB = 4 * DIM - 1;  // half bandwidth inside a block
    static constexpr int NAR = ND ? (NR0 > NR1 ? NR0 : NR1) : NZ;  // entries of a lane's matrix row
    static_assert(!ND || (ND_FITS && W >= 2 && FB_ == 8), "nested dissection: both wavefronts' rows within 64 lanes, two or more wavefronts, fp64");
    static constexpr int nd_nb(int w) { return w == 0 ? NB0 : NB1; }
    static constexpr int nd_nr(int w) { return w == 0 ? NR0 : NR1; }
    // separator variable <-> its position among wavefront 1's accumulators
    static constexpr int nd_unrank1(int r) { return (!ND_COMPACT || r < 2 * DIM) ? r : r + MS * DIM; }
    static constexpr int nd_rank1(int sc) { return (!ND_COMPACT || sc < 2 * DIM) ? sc : sc - MS * DIM; }
    // local column of separator variable sc in wavefront w's rows
    static constexpr int nd_col(int w, int sc) { return w == 0 ? NB0 + sc : NB1 + nd_rank1(sc); }
    // z index (axis-major: k * NZA + a) of variable (axis k, segment m, j): j = 0, 1, 2 <-> c3, c4, c5; the last segment under the
    // end stop has one variable
    static constexpr int zi_kmj(int k, int m, int j) { return k * NZA + ((ES && m == M - 1) ? 3 * (M - 1) : 3 * m + j); }
    // nested dissection: local column c of wavefront w (0: L | S, 1: R | reachable S) -> z index
    static constexpr int nd_zi(int w, int c) {
        if (c < nd_nb(w)) {
            const int seg = c / (2 * DIM), r = c % (2 * DIM);
            return zi_kmj(r / 2, (w == 0 ? 0 : MS + 1) + seg, r % 2);
        }
        int s = w == 0 ? c - NB0 : nd_unrank1(c - NB1);
        if (s < 2 * DIM) return zi_kmj(s / 2, MS, s % 2);
        s -= 2 * DIM;
        return zi_kmj(s % DIM, s / DIM, 2);
    }
    // Separator columns a block can reach (before and after fill-in): (c3, c4) of segment m couple to c5 of segments m-1, m, m+1
    // only, so block L (segments 0 .. MS-1) never touches c5 of segments > MS and block R (MS+1 .. MLAST) never c5 of segments
    // < MS: the block phase and the hand-over skip those columns (21 resp. 24 of 36 at M = 10 in 3-D).
    static constexpr bool nd_touched(int w, int sc) {
        return sc < 2 * DIM || (w == 0 ? sc < 2 * DIM + (MS + 1) * DIM : sc >= 2 * DIM + MS * DIM);
    }
    static_assert(NZ <= T, "lane-per-row kernel needs dim*(3M-2) <= 64*W");
    static_assert(CP <= T, "6M-3 <= 64*W");
    static_assert(W == 1 || W == 2 || W == 4, "1, 2 or 4 wavefronts per QP");
    static_assert(M >= 2, "the reference assumes M >= 2 (src/traj_optimizer.cpp:341-352)");
    static_assert(FB_ == 8 || (FB_ == 4 && W == 1), "mixed precision: one wavefront per QP (the throughput form)");
    // LDS carve (in doubles)
    static constexpr int o_c = 0;               // control points (translated)
    static constexpr int o_dca = o_c + NX;      // affine direction, x-space
    static constexpr int o_dc = o_dca + NX;     // final direction, x-space
    static constexpr int o_x0 = o_dc + NX;      // x-space accumulators: XL, XA, XB1, XB2
    static constexpr int o_S = o_x0 + 4 * NX;   // per-cp 3x3 sym blocks [P][6]
    static constexpr int o_om = o_S + 6 * P;    // two-sided row weights
    static constexpr int o_z = o_om + NOM;      // z, dz
    static constexpr int o_goal = o_z + 2 * T;  // goal - p0 (4 doubles), then the origin p0 (4 doubles)
    static constexpr int o_red = o_goal + 8;    // cross-wave reduction scratch (W > 1): 2 buffers x W waves x 4
    static constexpr int o_col = o_red + (W > 1 ? 8 * W : 0);  // pivot-column / solve broadcast buffer, 2 x T
    static_assert(2 * T + 2 >= 32 + 6 * M, "the prologue parks the header (32 doubles) and the corridor boxes (6 M) in the column buffer");
    static constexpr int o_zs = o_col + 2 * T + 2;  // the last point that met the acceptance tests (z), NZ
    static constexpr int o_H = ((o_zs + NZ + 1) / 2) * 2;  // per-lane scratch rows of the reduced matrix [NZ][LDH], scalars of FB_ bytes
    // (nested dissection also parks the factor of the NS accumulator lanes there: [NZ + NS + 1][LDP])
    static constexpr int LDP = NAR | 1;
    static constexpr int H_DOUBLES = ND ? (((NZ + 1) * LDH > (NZ + NS + 1) * LDP) ? (NZ + 1) * LDH : (NZ + NS + 1) * LDP)
                                        : ((NZ + 1) * LDH * FB_ + 7) / 8;
    static constexpr int o_rows = ((o_H + H_DOUBLES + 1) / 2) * 2;  // + one dummy row for non-z lanes  // LSC row constants SoA nx,ny,nz,b [MAX_OBS*CP]
    static constexpr int NROW = MAX_OBS * CP;  // + one dead row per array
    static constexpr size_t lds_bytes() { return sizeof(double) * ((size_t)o_rows + 4 * ((size_t)NROW + 1)); }
};

#ifndef LSCQP_KERNEL_ATTR
#define LSCQP_KERNEL_ATTR
#endif
// ONE instance, solved by the calling workgroup (the body of the kernel below).
template <int M, int DIM, bool ES, int NSLOT, int W = 1, class FT = double>
__device__ __forceinline__ void lscqp_pdip_one(const DevClass& cls, const int64_t q, double* const smem, const lscqp_header* __restrict__ hdr,
                                               const lscqp_row* __restrict__ rows, const uint64_t* __restrict__ row_offsets,
                                               const lscqp_box* __restrict__ sfc, const double* __restrict__ x_init, double* __restrict__ x_out,
                                               double* __restrict__ obj_out, int32_t* __restrict__ status_out, lscqp_info* __restrict__ info_out) {
    using C = Cfg<M, DIM, ES, NSLOT, W, (int)sizeof(FT)>;
    constexpr bool MIXED = !std::is_same<FT, double>::value;
    constexpr int P = C::P, CP = C::CP, NZA = C::NZA, NZ = C::NZ, NX = C::NX, G = C::G, LDH = C::LDH, T = C::T;
    double* const c_ = smem + C::o_c;
    double* const dca_ = smem + C::o_dca;
    double* const dc_ = smem + C::o_dc;
    double* const XL = smem + C::o_x0;            // G' lambda
    double* const XA = smem + C::o_x0 + NX;       // G' q_aff
    double* const XB1 = smem + C::o_x0 + 2 * NX;  // G' (1/s)
    double* const XB2 = smem + C::o_x0 + 3 * NX;  // G' (-ds_a dl_a/s - w rp)
    double* const S_ = smem + C::o_S;
    double* const om_ = smem + C::o_om;
    double* const z_ = smem + C::o_z;
    double* const dz_ = smem + C::o_z + T;
    double* const goal_ = smem + C::o_goal;
    double* const red_ = smem + C::o_red;
    double* const col_ = smem + C::o_col;
    (void)red_;
    double* const zs_ = smem + C::o_zs;
    FT* const Hs = reinterpret_cast<FT*>(smem + C::o_H);
    constexpr int NROW = C::NROW;
    double* const Rnx = smem + C::o_rows;
    double* const Rny = Rnx + (NROW + 1);
    double* const Rnz = Rny + (NROW + 1);
    double* const Rb = Rnz + (NROW + 1);

    const int lane = threadIdx.x;  // thread of the QP's workgroup, 0 .. 64*W-1 ("lane" throughout)
#ifdef LSCQP_PHASE_TIMING
    unsigned long long tprev_ = __builtin_readcyclecounter();
#endif
    // Hand-off through LDS between the lanes of the QP: see LSCQP_WAVE_LDS_SYNC for W = 1; a real barrier for W = 2.
    // Every branch around a BLOCK_SYNC is workgroup-uniform (conditions come out of block reductions / LDS broadcasts).
#define LSCQP_BLOCK_SYNC()                       \
    do {                                         \
        if constexpr (W == 1) {                  \
            LSCQP_WAVE_LDS_SYNC();               \
        } else {                                 \
            __syncthreads();                     \
        }                                        \
    } while (0)
    // cross-wave stage of the block reductions (W = 2): wave results meet in LDS, alternating between two buffers so
    // that one barrier per reduction suffices
    int red_flip = 0;
    auto cross3 = [&](double& a, double& b, double& c, auto opa, auto opb, auto opc) {
        if constexpr (W > 1) {
            double* const buf = red_ + red_flip * 4 * W;
            red_flip ^= 1;
            if ((lane & 63) == 0) {
                double* const mine = buf + 4 * (lane >> 6);
                mine[0] = a;
                mine[1] = b;
                mine[2] = c;
            }
            __syncthreads();
            a = buf[0];
            b = buf[1];
            c = buf[2];
#pragma unroll
            for (int w = 1; w < W; w++) {
                a = opa(a, buf[4 * w + 0]);
                b = opb(b, buf[4 * w + 1]);
                c = opc(c, buf[4 * w + 2]);
            }
        }
    };
    auto op_sum = [](double x, double y) { return x + y; };
    auto op_max = [](double x, double y) { return fmax(x, y); };
    auto block_sum2_max1 = [&](double& a, double& b, double& c) {
        wave_sum2_max1(a, b, c);
        cross3(a, b, c, op_sum, op_sum, op_max);
    };
    auto block_max2 = [&](double& a, double& b) {
        wave_max2(a, b);
        double c = 0;
        cross3(a, b, c, op_max, op_max, op_max);
    };
    auto block_max1_sum1 = [&](double& a, double& b) {
        wave_max1_sum1(a, b);
        double c = 0;
        cross3(a, b, c, op_max, op_sum, op_max);
    };
    auto block_sum = [&](double v) -> double {
        v = wave_sum(v);
        double b = 0, c = 0;
        cross3(v, b, c, op_sum, op_sum, op_sum);
        return v;
    };
    auto block_max = [&](double v) -> double {
        v = wave_max(v);
        double b = 0, c = 0;
        cross3(v, b, c, op_max, op_max, op_max);
        return v;
    };
    // The header (256 B) and the instance's corridor boxes (48 B per segment) are fetched ONCE, one 8-byte load per lane, and parked in
    // the pivot-column buffer (idle until the first factorisation); the prologue reads them from there.  Read field by field from
    // global memory they were a string of dependent scalar / vector loads: 9.6 k + 11.9 k cycles of a 4096-QP launch's prologue per QP
    // (header + control points, two-sided row set-up; tools/phase_timing.py) against 3.5 k + 5.2 k now.  `H` below points into LDS.
    double* const org_ = goal_ + 4;
    if (cls.repair) {  // (uniform) what an earlier pass -- or the dual active-set phase in front of this kernel -- finished is skipped before anything is fetched
        const int st_early = status_out[q];
        if (st_early == LSCQP_STATUS_OPTIMAL || st_early == LSCQP_STATUS_CAPACITY) return;
        // (round 6: an instance the dual active-set phase PROVED infeasible -- INFEASIBLE with LSCQP_INFO_ACTIVE_SET -- is finished: the pass right
        // behind the phase takes only what the phase marked ITER_LIMIT; a later pass recognises the verdict by the flag)
        if (cls.repair == 3 && st_early != LSCQP_STATUS_ITER_LIMIT) return;
        if (st_early == LSCQP_STATUS_INFEASIBLE && info_out && (info_out[q].flags & LSCQP_INFO_ACTIVE_SET)) return;
    }
    {
        const double* hsrc = reinterpret_cast<const double*>(hdr + q);
        const double* ssrc = reinterpret_cast<const double*>(sfc) + q * 6 * M;
        for (int e = lane; e < 32 + (cls.use_sfc ? 6 * M : 0); e += T) col_[e] = e < 32 ? hsrc[e] : ssrc[e - 32];
        __syncthreads();
    }
    const lscqp_header* H = reinterpret_cast<const lscqp_header*>(col_);
    const lscqp_box* const sfcl = reinterpret_cast<const lscqp_box*>(col_ + 32);  // boxes of THIS instance, [M]
    int flags = 0;           // lscqp_info.flags
    int it_before = 0;       // iterations of an earlier pass over this instance
    if (cls.repair) {        // uniform over the workgroup
        const int st0 = status_out[q];
        if (st0 == LSCQP_STATUS_OPTIMAL || st0 == LSCQP_STATUS_CAPACITY) return;
        if (cls.repair != 3) {  // (3: the FIRST interior-point pass, behind the dual active-set phase of lscqp_das.hip -- nothing was repaired)
            flags |= LSCQP_INFO_REPAIRED;
            if (info_out) it_before = info_out[q].iterations;
        }
    }
    // An instance with more obstacles than this kernel instance has row slots for is REFUSED, never truncated: dropping LSC
    // rows silently would void the collision-avoidance guarantee the rows exist for (reference: every obstacle gets its rows,
    // src/traj_optimizer.cpp:399-437).  The caller picks a larger instance through n_obs_max or splits the batch.
    if (H->n_obs > C::MAX_OBS) {
        for (int e = lane; e < NX; e += T) x_out[q * NX + e] = x_init ? x_init[q * NX + e] : H->p0[e / P];
        if (lane == 0) {
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
    const int n_obs = H->n_obs;
    const double dt = cls.dt;

    // ---- per-QP scalars (uniform) ------------------------------------------------------------------------
    // The origin is only needed by the prologue (and re-read from the header by the epilogue); the translated goal is
    // read per axis in every iteration and sits in LDS: as per-lane scalars these uniform values would be twelve
    // long-lived VGPRs, as a register array hipcc selects between the elements through scratch memory.
    const double org0 = H->p0[0], org1 = H->p0[1], org2 = H->p0[2];
    const double goal0 = H->goal[0] - org0, goal1 = H->goal[1] - org1, goal2 = H->goal[2] - org2;
    if (lane < 3) goal_[lane] = H->goal[lane] - H->p0[lane], org_[lane] = H->p0[lane];
    int ts = H->terminal_segments;
    if (ts <= 0) {  // src/traj_optimizer.cpp:530-538 in fp64
        const double d2 = goal0 * goal0 + goal1 * goal1 + goal2 * goal2;
        ts = (int)((M * dt - sqrt(d2) / H->nominal_velocity + 1e-9) / dt);
        if (ts < 1) ts = 1;
    }
    if (ts > M) ts = M;
    const double q2s = cls.q2s;
    const double wt2 = 2.0 * cls.w_t;

    // ---- lane roles ---------------------------------------------------------------------------------------
    // z lane: row r = lane of the reduced system, r = k*NZA + a
    // (which z variable a lane owns: lane r <-> z index r for nz <= 64; the nested-dissection layout of Cfg for nz > 64)
    auto lane_zi = [](int lv) -> int {
        if constexpr (!C::ND) {
            return lv < NZ ? lv : -1;
        } else {
            const int w = lv >> 6, t = lv & 63;
            const bool own = (w == 0 && t < C::NR0) || (w == 1 && t < C::NB1);
            const int tc = own ? t : 0;
            // C::nd_zi with run-time arguments (the accumulator lanes of wavefront 1 own no variable)
            int zi;
            if (tc < (w == 0 ? C::NB0 : C::NB1)) {
                const int seg = tc / (2 * DIM), r = tc % (2 * DIM);
                zi = C::zi_kmj(r / 2, (w == 0 ? 0 : C::MS + 1) + seg, r % 2);
            } else {
                int s_ = tc - C::NB0;
                const bool mid = s_ < 2 * DIM;
                const int s2 = mid ? s_ : s_ - 2 * DIM;
                zi = mid ? C::zi_kmj(s2 / 2, C::MS, s2 % 2) : C::zi_kmj(s2 % DIM, s2 / DIM, 2);
            }
            return own ? zi : -1;
        }
    };
    // scratch-matrix row of a lane (z lanes only; the others share the dummy row NZ)
    auto lane_slot = [](int lv) -> int {
        if constexpr (!C::ND) {
            return lv < NZ ? lv : NZ;
        } else {
            const int w = lv >> 6, t = lv & 63;
            return (w == 0 && t < C::NR0) ? t : (w == 1 && t < C::NB1) ? C::NR0 + t : NZ;
        }
    };
    // where a lane parks its factor row while pass 2 runs (the scratch matrix is free then); nested dissection: [NZ + NS + 1][LDP]
    auto park_ptr = [&](int lv) -> FT* {
        if constexpr (C::ND) {
            const int w = lv >> 6, t = lv & 63;
            return Hs + ((w < 2 && t < (w == 0 ? C::NR0 : C::NR1)) ? w * C::NR0 + t : NZ + C::NS) * C::LDP;
        } else {
            return &Hs[(lv < NZ ? lv : NZ) * LDH];
        }
    };
    const int zi0 = lane_zi(lane);
    const bool zl = zi0 >= 0;
    const int zk = zl ? zi0 / NZA : 0;
    const int za = zl ? zi0 % NZA : 0;
    const bool zlast = ES && (za == 3 * (M - 1));
    const int zm = zlast ? (M - 1) : za / 3;
    const int zj = zlast ? 0 : za % 3;
    const bool has_next = (zm + 1 < M);
    // LOOP-INVARIANT LANE ROLES ARE RECOMPUTED IN EVERY PHASE OF THE ITERATION from an opaque copy of the lane id
    // (LSCQP_PHASE_LANE).  Derived from `lane` directly, hipcc hoists every role, LDS address, selector weight and
    // lane-vs-column compare mask out of the iteration loop (~150 VGPRs + ~240 SGPRs worth), spills them to scratch
    // memory / VGPR lanes and reloads them in the middle of dependent LDS chains (~150 cycles per scratch reload,
    // tools/ubench.hip).  A handful of integer instructions per phase is far cheaper.
#define LSCQP_PHASE_LANE(name) \
    int name = lane;           \
    asm volatile("" : "+v"(name))
    struct ZRole {  // the z variable of lane `lv`: z index zi = k*NZA + a, scratch row `slot`
        bool zl, zlast, has_next;
        int zk, zm, zj, gbase, gnext, zi, slot;
    };
    auto zrole = [&](int lv) -> ZRole {
        ZRole R;
        const int zi_ = lane_zi(lv);
        R.zl = zi_ >= 0;
        R.zi = R.zl ? zi_ : 0;
        R.slot = lane_slot(lv);
        R.zk = R.zl ? zi_ / NZA : 0;
        const int za_ = R.zl ? zi_ % NZA : 0;
        R.zlast = ES && (za_ == 3 * (M - 1));
        R.zm = R.zlast ? (M - 1) : za_ / 3;
        R.zj = R.zlast ? 0 : za_ % 3;
        R.has_next = (R.zm + 1 < M);
        R.gbase = R.zk * P + 6 * R.zm;
        R.gnext = R.zk * P + 6 * (R.has_next ? R.zm + 1 : R.zm);
        return R;
    };
    struct LRole {  // LSC lane: (group lg, control point lcp in [0,CP))
        bool ll;
        int lg, lcp, lx;
    };
    auto lrole = [](int lv) -> LRole {
        LRole R;
        R.ll = lv < G * CP;
        R.lg = R.ll ? lv / CP : 0;
        R.lcp = R.ll ? lv % CP : 0;
        R.lx = R.lcp + 3;
        return R;
    };
    // Selector weights of the z variable (which of c3,c4,c5 it drives, and the TB column towards the next segment)
#define LSCQP_LANE_WEIGHTS(R)                                                                                        \
    const double e0 = (R.zlast || R.zj == 0) ? 1.0 : 0.0, e1 = (R.zlast || R.zj == 1) ? 1.0 : 0.0,                   \
                 e2 = (R.zlast || R.zj == 2) ? 1.0 : 0.0;                                                            \
    const double tb0 = R.has_next ? (R.zj == 2 ? 1.0 : 0.0) : 0.0;                                                   \
    const double tb1 = R.has_next ? (R.zj == 0 ? 0.0 : R.zj == 1 ? -1.0 : 2.0) : 0.0;                                \
    const double tb2 = R.has_next ? (R.zj == 0 ? 1.0 : R.zj == 1 ? -4.0 : 4.0) : 0.0
    // phase-local copies under the names the prologue uses (they shadow the prologue's, which stay loop invariant)
#define LSCQP_Z_ROLES()                                                                                  \
    LSCQP_PHASE_LANE(lvz_);                                                                              \
    const ZRole ZR = zrole(lvz_);                                                                        \
    const bool zl = ZR.zl, zlast = ZR.zlast, has_next = ZR.has_next;                                     \
    const int zk = ZR.zk, zm = ZR.zm, zj = ZR.zj, gbase = ZR.gbase, gnext = ZR.gnext, zi = ZR.zi;        \
    FT* const hrow = &Hs[ZR.slot * LDH];                                                                 \
    (void)zlast, (void)has_next, (void)zk, (void)zm, (void)zj, (void)gbase, (void)gnext, (void)hrow, (void)zi
#define LSCQP_L_ROLES()                                                                                  \
    LSCQP_PHASE_LANE(lvl_);                                                                              \
    const LRole LR = lrole(lvl_);                                                                        \
    const bool ll = LR.ll;                                                                               \
    const int lg = LR.lg, lcp = LR.lcp, lx = LR.lx;                                                      \
    (void)lcp
    auto zidx = [](int m, int j) -> int { return (ES && m == M - 1) ? 3 * (M - 1) : 3 * m + j; };
    // LSC lane: (group lg, control point lcp in [0,CP)); x-space index of the lane's control point per axis
    const bool ll = lane < G * CP;
    const int lg = ll ? lane / CP : 0;
    const int lcp = ll ? lane % CP : 0;
    const int lx = lcp + 3;  // index within one axis of c_

    // ---- control points: fixed part from (p0, v0, a0) (:321-338), free part = "stay at c2" ----------------
    for (int e = lane; e < NX; e += T) {
        const int k = e / P, cp = e % P;
        const double cf1 = H->v0[k] * dt * 0.2;                    // c1 - c0
        const double cf2 = H->a0[k] * dt * dt * 0.05 + 2.0 * cf1;  // c2 - c0
        c_[e] = (cp == 0) ? 0.0 : (cp == 1) ? cf1 : cf2;
        dca_[e] = 0.0;
        dc_[e] = 0.0;
    }
    if (zl) {
        const double cf1 = H->v0[zk] * dt * 0.2;
        z_[zi0] = H->a0[zk] * dt * dt * 0.05 + 2.0 * cf1;
        FT* hrow0 = &Hs[lane_slot(lane) * LDH];
#pragma unroll
        for (int cidx = 0; cidx < NZ; cidx++) hrow0[cidx] = (FT)0;  // entries outside the lane's pattern stay zero
        // Primal start from the caller's initial trajectory (TrajOptimizer::solve's `initial_traj`: the shifted previous
        // plan): the free control points c3..c5 of every segment (c5 alone under the end stop), translated.  The
        // equalities are re-imposed by c = c_fixed + T z below, so float32 rounding of the plan does no harm.
        if (x_init) {
            const double o_k = (zk == 0) ? org0 : (zk == 1) ? org1 : org2;
            z_[zi0] = x_init[q * NX + zk * P + 6 * zm + (zlast ? 5 : 3 + zj)] - o_k;
        }
    }
    if (x_init) {  // uniform over the grid
        __syncthreads();
        for (int e = lane; e < NX; e += T) {
            const int k = e / P, cp = e % P, m = cp / 6, i = cp % 6;
            const double vhi = z_[k * NZA + ((ES && m == M - 1) ? 3 * (M - 1) : 3 * m + (i >= 3 ? i - 3 : 0))];
            const double* zz = &z_[k * NZA + 3 * (m >= 1 ? m - 1 : 0)];
            const int ii = i < 3 ? i : 0;
            const double vlo = TBc(ii, 0) * zz[0] + TBc(ii, 1) * zz[1] + TBc(ii, 2) * zz[2];
            if (cp >= 3) c_[e] = (i >= 3) ? vhi : vlo;
        }
    }
    LSCQP_T(11);  // prologue a: header, control points, scratch rows
    // ---- stage LSC row constants: HBM (AoS 32 B, [oi][m][i]) -> LDS SoA [oi][cp], translated to the agent origin --
    {
        const uint64_t roff = row_offsets[q];
        const lscqp_row* R = rows + roff;
        const int nact = n_obs * CP;
        // The loads of a lane are issued LSCQP_STAGE_UNROLL at a time before any of them is consumed: with one load in flight per
        // pass through the loop the staging was a chain of HBM round trips (25.5 k cycles per QP when 4096 QPs stage at once, 3.6 k in a
        // 64-QP launch; tools/phase_timing.py).  Lanes past the end re-read row 0 of the instance (always present when nact > 0).
#ifndef LSCQP_STAGE_UNROLL
#define LSCQP_STAGE_UNROLL 4
#endif
        for (int e0 = lane; e0 < nact; e0 += T * LSCQP_STAGE_UNROLL) {
            double4 v[LSCQP_STAGE_UNROLL];
#pragma unroll
            for (int u = 0; u < LSCQP_STAGE_UNROLL; u++) {
                const int e = e0 + T * u;
                const int ec = e < nact ? e : 0;
                const int o = ec / CP, cp = ec % CP;
                if (cls.rows_f32) {  // LSCQP_ROWS_F32: 16-byte rows, widened here; everything after this line is the same arithmetic
                    const float4 f = reinterpret_cast<const float4*>(rows)[roff + (uint64_t)(o * P + cp + 3)];
                    v[u] = double4{(double)f.x, (double)f.y, (double)f.z, (double)f.w};
                } else {
                    v[u] = *reinterpret_cast<const double4*>(&R[o * P + cp + 3]);
                }
            }
#pragma unroll
            for (int u = 0; u < LSCQP_STAGE_UNROLL; u++) {
                const int e = e0 + T * u;
                double nx = v[u].x, ny = v[u].y, nz = (DIM == 3) ? v[u].z : 0.0;
                double b = v[u].w - (v[u].x * org0 + v[u].y * org1 + (DIM == 3 ? v[u].z * org2 : 0.0));
                if (sqrt(v[u].x * v[u].x + v[u].y * v[u].y + v[u].z * v[u].z) < 1e-5) {  // dropped like the reference does (:409-411)
                    nx = ny = nz = 0.0;
                    b = -1.0;
                }
                if (e < nact) {
                    Rnx[e] = nx;
                    Rny[e] = ny;
                    Rnz[e] = nz;
                    Rb[e] = b;
                }
            }
        }
        if (lane == 0) {  // the dead row
            Rnx[NROW] = Rny[NROW] = Rnz[NROW] = 0.0;
            Rb[NROW] = -1.0;
        }
    }
    __syncthreads();
    LSCQP_T(12);  // prologue b: row staging

    // ---- two-sided rows, one type per slot (registers) ----------------------------------------------------
    // slot layout: [0,SI) interval | [SI,SI+SV) velocity | [..,+SA) acceleration | [..,+SC) communication pair
    constexpr int NS2 = C::NS2;
    int t_ix[NS2];                                      // x-space index of the row's first control point (-1: no row)
    int t_i2[NS2];                                      // second index (comm pairs only)
#ifdef LSCQP_STATE_IN_VGPR
    struct VD {  // same interface as AD, plain register (measured alternative)
        double v;
        __device__ __forceinline__ double get() const { return v; }
        __device__ __forceinline__ void set(double x) { v = x; }
    };
    using SD = VD;
#else
    using SD = AD;
#endif
    SD t_lo[NS2], t_hi[NS2];                        // bounds
    SD t_sl[NS2], t_sh[NS2], t_ll[NS2], t_lh[NS2];  // slack / multiplier of the lo and hi side
    // Only (s, lambda) persist across the factorisation; residuals and directions of a row are recomputed in every
    // pass from the x-space vectors (a handful of FMAs) instead of being kept live next to the matrix row A[].
    {
        const double rho_pair = 0.5 * cls.comm_range - H->radius;  // :484
        const double rho_wp = 0.5 * cls.comm_range - 1e-5;         // :495
        const bool comm_on = cls.comm_range > 0;
#pragma unroll
        for (int u = 0; u < NS2; u++) {
            t_ix[u] = -1;
            t_i2[u] = 0;
            double lo_u = -1.0, hi_u = 1.0;
            if (u < C::SI) {  // interval on one control point
                const int e = lane + T * u;
                if (e < NX) {
                    const int k = e / P, cp = e % P, m = cp / 6;
                    if (cp >= 3) {
                        t_ix[u] = e;
                        // (org[] is only ever indexed with constants: a dynamic index would put it in scratch memory)
                        const double ok_ = (k == 0) ? org0 : (k == 1) ? org1 : org2;
                        double lo = cls.world_min[k] - ok_, hi = cls.world_max[k] - ok_;  // :252-253,260-265
                        if (cls.rsfc && k == 2 && m == 0) {  // :255-258
                            lo = -100.0 - ok_;
                            hi = 100.0 - ok_;
                        }
                        if (cls.use_sfc) {                                                // :372-397
                            lo = fmax(lo, sfcl[m].bmin[k] - ok_);
                            hi = fmin(hi, sfcl[m].bmax[k] - ok_);
                        }
                        if (comm_on && cp % 6 == 5) {  // pairs (m, mi=0) :482-487 and waypoint rows :494-497
                            const double wpk = H->next_waypoint[k] - ok_;
                            lo = fmax(lo, fmax(-rho_pair, wpk - rho_wp));
                            hi = fmin(hi, fmin(rho_pair, wpk + rho_wp));
                        }
                        lo_u = lo;
                        hi_u = hi;
                    }
                }
            } else if (u < C::SI + C::SV) {  // velocity (m,i): c[i+1]-c[i], |.| <= vmax dt/n   (:448-453)
                const int v = lane + T * (u - C::SI);
                if (v < C::NV) {
                    const int k = v / (5 * M), r = v % (5 * M), m = r / 5, i = r % 5;
                    if (!(m == 0 && i < 2)) {
                        t_ix[u] = k * P + 6 * m + i;
                        hi_u = H->vmax[k] * dt * 0.2;
                        lo_u = -hi_u;
                    }
                }
            } else if (u < C::SI + C::SV + C::SA) {  // acceleration (m,i): c[i+2]-2c[i+1]+c[i]   (:462-471)
                const int a = lane + T * (u - C::SI - C::SV);
                if (a < C::NA) {
                    const int k = a / (4 * M), r = a % (4 * M), m = r / 4, i = r % 4;
                    if (!(m == 0 && i < 1)) {
                        t_ix[u] = k * P + 6 * m + i;
                        hi_u = H->amax[k] * dt * dt * 0.05;
                        lo_u = -hi_u;
                    }
                }
            } else {  // pair (uu, up<uu): c[uu][5] - c[up+1][0]   (:482-487 with mi = up+1 >= 1)
                const int cc = lane + T * (u - C::SI - C::SV - C::SA);
                if (cc < C::NC && comm_on) {
                    const int k = cc / C::NCP, ci = cc % C::NCP;
                    int uu = 1;
                    while (uu * (uu + 1) / 2 <= ci) uu++;
                    const int up = ci - uu * (uu - 1) / 2;
                    t_ix[u] = k * P + 6 * (up + 1);
                    t_i2[u] = k * P + 6 * uu + 5;
                    hi_u = rho_pair;
                    lo_u = -rho_pair;
                }
            }
            t_lo[u].set(lo_u);
            t_hi[u].set(hi_u);
        }
    }
    // value of the row's stencil on an x-space vector (slot type is compile-time)
    auto row_val = [&](const double* v, int u) -> double {
        const int ix = t_ix[u] < 0 ? 0 : t_ix[u];
        if (u < C::SI) return v[ix];
        if (u < C::SI + C::SV) return v[ix + 1] - v[ix];
        if (u < C::SI + C::SV + C::SA) return v[ix + 2] - 2.0 * v[ix + 1] + v[ix];
        return v[t_i2[u]] - v[ix];
    };
    auto row_scatter = [&](double* X, int u, double val, bool pred) {
        const int ix = t_ix[u];
        if (u < C::SI) {
            if (pred) atomicAdd(&X[ix], val);
        } else if (u < C::SI + C::SV) {
            if (pred) atomicAdd(&X[ix + 1], val);
            if (pred) atomicAdd(&X[ix], -val);
        } else if (u < C::SI + C::SV + C::SA) {
            if (pred) atomicAdd(&X[ix + 2], val);
            if (pred) atomicAdd(&X[ix + 1], -2.0 * val);
            if (pred) atomicAdd(&X[ix], val);
        } else {
            if (pred) atomicAdd(&X[t_i2[u]], val);
            if (pred) atomicAdd(&X[ix], -val);
        }
    };
    // LDS accumulation in a fixed order: with two wavefronts per QP the ds_add_f64 of the waves would interleave
    // differently from run to run (floating-point sums are order dependent); the waves take turns instead, so results
    // stay bitwise reproducible.  f(turn) issues its atomics predicated on `turn`.
    auto wave_ordered = [&](auto&& f) {
        if constexpr (W == 1) {
            f(true);
        } else {
#pragma unroll
            for (int w_ = 0; w_ < W; w_++) {
                f((lane >> 6) == w_);
                if (w_ + 1 < W) __syncthreads();
            }
        }
    };
    auto om_index = [&](int u) -> int {  // where this slot's weight lives in om_
        if (u < C::SI) return lane + T * u;
        if (u < C::SI + C::SV) return C::OV + lane + T * (u - C::SI);
        if (u < C::SI + C::SV + C::SA) return C::OA + lane + T * (u - C::SI - C::SV);
        return C::OC + lane + T * (u - C::SI - C::SV - C::SA);
    };
    auto om_limit = [&](int u) -> int {
        if (u < C::SI) return C::OV;
        if (u < C::SI + C::SV) return C::OA;
        if (u < C::SI + C::SV + C::SA) return C::OC;
        return C::NOM;
    };

    LSCQP_T(13);  // prologue c: two-sided row setup
    // ---- initial slacks / multipliers ----------------------------------------------------------------------
    // Centred start: s = max(residual, 0.1), lambda = mu0 / s with mu0 = 3e-3, i.e. every complementarity product starts
    // at mu0.  Most rows are far from active at the optimum and get a small multiplier, rows close to their bound get a
    // large one.  Swept in tools/proto_pdip.py together with the step rule: uniform lambda0 = 1 needed ~7 iterations on
    // the forest workload, uniform 0.03 ~5 (4.1 with the adaptive step rule), the centred start 3.4 with the same or a
    // shorter tail (maximum over a 64-QP batch 5-6; smaller mu0 lowers the mean further but lengthens the tail).
    // With the caller's initial trajectory as primal start the residuals start smaller and a tighter start pays
    // (mu0 = 1e-3, slack floor 0.03: 3.3 -> 3.2 iterations; from hover the same setting lengthens the tail).
#ifndef LSCQP_WARM_MU0
#define LSCQP_WARM_MU0 1e-3
#endif
#ifndef LSCQP_WARM_S0
#define LSCQP_WARM_S0 0.03
#endif
#ifndef LSCQP_COLD_MU0
#define LSCQP_COLD_MU0 3e-3
#endif
    // mu0 was tuned where the terminal cost pulls with ts * w_t * |goal - p0| <= ~3 (a goal at most ~1.5 m away, weight 1: every
    // workload of BASELINE.json and the reference's own mission).  The multipliers at the optimum grow with that pull; with a far
    // goal or a heavy weight a start three decades below them costs 6-10 extra iterations, during which the primal residual creeps
    // (w_t = 50, goal 7 m away, every segment terminal: 22 iterations where the oracle needs 13, and at w_t = 200 the creep was
    // mistaken for infeasibility).  Scaling mu0 with the pull beyond that range restores 11-17 iterations (measured mu0 = 3e-3 ..
    // 3 on M = 5, 6, 10); inside it the factor is 1 and nothing changes.
    const double pull = (double)ts * cls.w_t * fmax(fabs(goal0), fmax(fabs(goal1), fabs(goal2)));
    const double mu_scale = fmin(1e3, fmax(1.0, 0.25 * pull));
    const double MU0 = (x_init ? cls.warm_mu0 : LSCQP_COLD_MU0) * mu_scale, S0MIN = x_init ? cls.warm_s0 : 0.1;
    int status = LSCQP_STATUS_ITER_LIMIT;
    double m_tot = 0;
    SD r_s[NSLOT], r_l[NSLOT];  // LSC row state: slack, multiplier (lambda == 0 marks a dead slot)
    // centred row state at the current control points c_: s = max(residual, s_c), lambda = mu_c / s.  Used for the start and,
    // at most once, to re-centre an iteration that has jammed against the boundary (see the loop).
    auto centre_rows = [&](double mu_c, double s_c, double& cnt, bool& bad) {
#pragma unroll
        for (int u = 0; u < NS2; u++) {
            double sl0 = 1.0, sh0 = 1.0, l0_ = 0.0;
            if (t_ix[u] >= 0) {
                const double y = row_val(c_, u);
                const double lo = t_lo[u].get(), hi = t_hi[u].get();
                if (lo > hi) bad = true;
                sl0 = fmax(y - lo, s_c);
                sh0 = fmax(hi - y, s_c);
                l0_ = 1.0;  // marks an existing row; the multipliers are set below
                cnt += 2.0;
            }
            t_sl[u].set(sl0);
            t_sh[u].set(sh0);
            t_ll[u].set(l0_ * mu_c * fast_rcp(sl0));  // (an IEEE fp64 division is ~30 instructions; 22 of them per lane here)
            t_lh[u].set(l0_ * mu_c * fast_rcp(sh0));
        }
        LSCQP_L_ROLES();  // phase-local lane role: the lambda also runs inside the loop
        const double cx = ll ? c_[lx] : 0.0, cy = ll ? c_[P + lx] : 0.0, cz = (ll && DIM == 3) ? c_[2 * P + lx] : 0.0;
#pragma unroll
        for (int u = 0; u < NSLOT; u++) {
            double s_init = 1.0, l_init = 0.0;
            const int o = lg + G * u;
            if (ll && o < n_obs) {
                const int e = o * CP + lcp;
                const double nx = Rnx[e], ny = Rny[e], nz = Rnz[e];
                if ((nx != 0.0) || (ny != 0.0) || (nz != 0.0)) {
                    s_init = fmax(nx * cx + ny * cy + nz * cz - Rb[e], s_c);
                    l_init = mu_c * fast_rcp(s_init);
                    cnt += 1.0;
                }
            }
            r_s[u].set(s_init);
            r_l[u].set(l_init);
        }
    };
    {
        double cnt = 0;
        bool bad = false;
        centre_rows(MU0, S0MIN, cnt, bad);
        m_tot = block_sum(cnt);
        if (block_max(bad ? 1.0 : 0.0) > 0.0) status = LSCQP_STATUS_INFEASIBLE;  // empty interval: lo > hi
    }
    const double inv_m = 1.0 / m_tot;

    // objective exactly as cplex.getObjValue() reports it (src/traj_optimizer.cpp:100): jerk cost
    //   x'(w_c Q)x == w_c * 3600 dt^-5 * sum_seg (D3 c)' MB (D3 c)   (third differences: no cancellation)
    // plus the terminal cost including its constant goal^2 term (:301-316).  Translation invariant; the optional
    // world-frame correction restores the reference's own coefficient rounding (O(1e-9 |x|^2)).  Branch-free.
    auto objective = [&](bool ref_rounding, int lv) -> double {
        const bool on = lv < DIM * M;
        const int k = on ? lv / M : 0, m = on ? lv % M : 0;
        const double* cc = &c_[k * P + 6 * m];
        const double j0 = (cc[3] - cc[0]) - 3.0 * (cc[2] - cc[1]);
        const double j1 = (cc[4] - cc[1]) - 3.0 * (cc[3] - cc[2]);
        const double j2 = (cc[5] - cc[2]) - 3.0 * (cc[4] - cc[3]);
        const double quad =
            0.2 * (j0 * j0 + j2 * j2) + (2.0 / 15.0) * j1 * j1 + 0.2 * (j0 * j1 + j1 * j2) + (1.0 / 15.0) * j0 * j2;
        double part = 0.5 * q2s * 3600.0 * quad;
        if (ref_rounding) {  // compile-time constant at both call sites
            const double ok_ = org_[k];  // (the header's copy in the column buffer is long gone by the epilogue)
            double corr = 0;
#pragma unroll
            for (int i = 0; i < 6; i++) {
                double r = 0;
#pragma unroll
                for (int ip = 0; ip < 6; ip++) r += cls.dQ[i * 6 + ip] * (cc[ip] + ok_);
                corr += r * (cc[i] + ok_);
            }
            part += cls.w_c * corr;
        }
        const double dgoal = cc[5] - goal_[k];
        part += (m >= M - ts) ? cls.w_t * dgoal * dgoal : 0.0;
        return block_sum(on ? part : 0.0);
    };
    // gather an x-space vector into the own z component: (T' v)_r.  Lanes without a next segment read the clamped
    // segment with zero weights; non-z lanes compute a value that is never used.
    auto gatherT = [&](const double* X, const ZRole& R) -> double {
        LSCQP_LANE_WEIGHTS(R);
        const double* xs = &X[R.gbase];
        const double* xn = &X[R.gnext];
        return e0 * xs[3] + e1 * xs[4] + e2 * xs[5] + tb0 * xn[0] + tb1 * xn[1] + tb2 * xn[2];
    };
    // x-space vector = T * (z-space vector in LDS), all NX entries, branch-free
    auto expandT = [&](const double* zsrc, double* out, bool keep_fixed) {
        LSCQP_PHASE_LANE(lve_);
#pragma unroll
        for (int t = 0; t < (NX + T - 1) / T; t++) {
            const int e0_ = lve_ + T * t;
            const bool on = e0_ < NX;
            const int e = on ? e0_ : 0;
            const int k = e / P, cp = e % P, m = cp / 6, i = cp % 6;
            const double vhi = zsrc[k * NZA + zidx(m, i >= 3 ? i - 3 : 0)];
            const double* zz = &zsrc[k * NZA + 3 * (m >= 1 ? m - 1 : 0)];
            const double vlo = TBc(i < 3 ? i : 0, 0) * zz[0] + TBc(i < 3 ? i : 0, 1) * zz[1] + TBc(i < 3 ? i : 0, 2) * zz[2];
            const double v = (i >= 3) ? vhi : ((m >= 1) ? vlo : 0.0);
            const bool wr = on && !(keep_fixed && cp < 3);  // the initial state (p0,v0,a0) is never touched
            if (wr) out[e] = v;
        }
    };

    FT A[C::NAR];  // the lane's row of the reduced KKT matrix (all nz columns, or L | S resp. R | S under nested dissection), then its LDL^T factors
    FT dinv_own = (FT)0;
    double res_p = 0, res_d = 0, res_gap = 0, obj_abs = 0;
    double snap_p = 0, snap_d = 0, snap_gap = 0;  // residuals of the last point that met the acceptance tests (kept in zs_)
    bool restore = false;                         // the result is that remembered point, not the current iterate
    int it = 0, near_cnt = 0, floor_cnt = 0;
    float rp_ref = 3.0e38f;  // primal residual four iterations ago (infeasibility test below)
    int stalled = 0;         // consecutive checks at which it had not shrunk by 30 %
    float alpha_first = 1.0f;  // step length of the first iteration (safety net of the tight warm start)
    bool net_done = false;
    float gap_mark = 3.0e38f;  // jam test: the gap when it last improved tenfold, iterations since, done once
    int jam_since = 0;
    bool recentred = false;
    int shift_level = 0;  // 0: no diagonal shift; 1, 2: the factorisation lost a pivot and the matrix carries 1e-14 / 1e-12 max|K| on its diagonal
    bool repeated = false;  // this pass of the loop repeats an iteration at the SAME point (a lost pivot, now with a diagonal shift): the point's
                            // tests and counters were taken the first time and are not taken again, and the iteration is not counted twice
    const double tol = cls.tol;
    const bool comm_on_k = cls.comm_range > 0;
    // (row of the scratch matrix a lane assembles into: non-z lanes share one dummy row, index NZ, that is never read)
    LSCQP_T(15);  // prologue d: start point, objective lambdas

    // THE LOOP BODY IS WRITTEN WITHOUT DIVERGENT REGIONS.  Every lane owns persistent state in registers (matrix row,
    // row slacks/multipliers); hipcc spills such state around large divergent regions with exec-masked stores, so the
    // lanes that are inactive inside the region (z lanes that own no LSC control point, LSC lanes that are no z lane,
    // ...) would get garbage back.  Inactive rows are therefore masked arithmetically (dead LDS row, zero weights,
    // selects) and only single LDS stores / atomics are predicated.
    if (status != LSCQP_STATUS_INFEASIBLE)
        for (it = 0; it < cls.max_iter; it++) {
            LSCQP_T(0);
            LSCQP_MARK(13);
            LSCQP_STOP(1)
            // ============ pass 1: residuals, weights, per-cp blocks =========================================
#pragma unroll
            for (int t = 0; t < (4 * NX + 6 * P + T - 1) / T; t++) {
                const int e = lane + T * t;
                if (e < 4 * NX + 6 * P) smem[C::o_x0 + e] = 0.0;  // XL,XA,XB1,XB2,S
            }
            LSCQP_BLOCK_SYNC();
            double sum_sl = 0, sum_pinf = 0, max_rp = 0;
#ifdef LSCQP_TRACE
            double tr_wmax = 0, tr_lmax = 0;
#endif
            double sc1[NS2], sc2[NS2];  // W > 1: scatter values of the two-sided rows, flushed in wave order below
#pragma unroll
            for (int u = 0; u < NS2; u++) {
                const bool on = t_ix[u] >= 0;  // rows that do not exist carry s = 1, lambda = 0, lo = -1, hi = 1
                const double tsl = t_sl[u].get(), tsh = t_sh[u].get(), tll = t_ll[u].get(), tlh = t_lh[u].get();
                const double tlo = t_lo[u].get(), thi = t_hi[u].get();
                const double y = row_val(c_, u);
                const double rpl = on ? (y - tlo) - tsl : 0.0, rph = on ? (thi - y) - tsh : 0.0;
                const double isl = row_rcp(tsl), ish = row_rcp(tsh);
                const double wl = tll * isl, wh = tlh * ish;
                sum_sl += tsl * tll + tsh * tlh;
                sum_pinf += tll * fabs(rpl) + tlh * fabs(rph);
                max_rp = fmax(max_rp, fmax(fabs(rpl), fabs(rph)));
                if constexpr (W == 1) {
                    row_scatter(XL, u, tll - tlh, on);
                    row_scatter(XA, u, wh * rph - wl * rpl, on);
                } else {
                    sc1[u] = tll - tlh;
                    sc2[u] = wh * rph - wl * rpl;
                }
                if (om_index(u) < om_limit(u)) om_[om_index(u)] = wl + wh;
            }
            {
                LSCQP_L_ROLES();
                double s00 = 0, s01 = 0, s02 = 0, s11 = 0, s12 = 0, s22 = 0, l0 = 0, l1 = 0, l2 = 0, a0 = 0, a1 = 0, a2 = 0;
                const double cx = c_[lx], cy = c_[P + lx], cz = (DIM == 3) ? c_[2 * P + lx] : 0.0;
#pragma unroll
                for (int u = 0; u < NSLOT; u++) {
                    const int o = lg + G * u;
                    // slots without a row read the dead row kept at index NROW (n = 0, b = -1; their s = 1, l = 0),
                    // which contributes exact zeros everywhere
                    const int e = (ll && o < n_obs) ? (o * CP + lcp) : NROW;
                    const double nx = Rnx[e], ny = Rny[e], nz = Rnz[e], s = r_s[u].get(), lam = r_l[u].get();
                    const double rp = (nx * cx + ny * cy + nz * cz - Rb[e]) - s;
                    const double is = row_rcp(s);
                    const double w = lam * is;
#ifdef LSCQP_TRACE
                    tr_wmax = fmax(tr_wmax, w);
                    tr_lmax = fmax(tr_lmax, lam);
#endif
                    sum_sl += s * lam;
                    sum_pinf += lam * fabs(rp);
                    max_rp = fmax(max_rp, fabs(rp));
                    const double wx = w * nx, wy = w * ny, wz = w * nz;
                    s00 += wx * nx; s01 += wx * ny; s11 += wy * ny;
                    l0 += lam * nx; l1 += lam * ny;
                    const double qa = -w * rp;
                    a0 += qa * nx; a1 += qa * ny;
                    if (DIM == 3) {
                        s02 += wx * nz; s12 += wy * nz; s22 += wz * nz;
                        l2 += lam * nz;
                        a2 += qa * nz;
                    }
                    }
                const int cp6 = lx * 6;
                wave_ordered([&](bool turn) {
                    if constexpr (W > 1) {
#pragma unroll
                        for (int u = 0; u < NS2; u++) {
                            row_scatter(XL, u, sc1[u], turn && t_ix[u] >= 0);
                            row_scatter(XA, u, sc2[u], turn && t_ix[u] >= 0);
                        }
                    }
                    if (ll && turn) {
                        atomicAdd(&S_[cp6 + 0], s00); atomicAdd(&S_[cp6 + 1], s01); atomicAdd(&S_[cp6 + 3], s11);
                        atomicAdd(&XL[lx], l0); atomicAdd(&XL[P + lx], l1);
                        atomicAdd(&XA[lx], a0); atomicAdd(&XA[P + lx], a1);
                        if (DIM == 3) {
                            atomicAdd(&S_[cp6 + 2], s02); atomicAdd(&S_[cp6 + 4], s12); atomicAdd(&S_[cp6 + 5], s22);
                            atomicAdd(&XL[2 * P + lx], l2);
                            atomicAdd(&XA[2 * P + lx], a2);
                        }
                    }
                });
            }
            block_sum2_max1(sum_sl, sum_pinf, max_rp);
            const double mu = sum_sl * inv_m;
            LSCQP_BLOCK_SYNC();
            LSCQP_T(1);
            LSCQP_STOP(2)

            // ============ cost gradient gathered into z-space, residual norms, convergence ====================
            // gx[k][cp] = sum_i' Q2[i][i'] c[m][i'] + 2 w_t (c[m][5] - goal) [terminal segments]  (:285-316)
            double gcost, gl, ga;
            {
                LSCQP_Z_ROLES();
                const double gk = goal_[zk];
                const double* cs = &c_[gbase];
                const double* cn = &c_[gnext];
                double g3 = 0, g4 = 0, g5 = 0, h0 = 0, h1 = 0, h2 = 0;
#pragma unroll
                for (int ip = 0; ip < 6; ip++) {
                    g3 += KQ(3, ip) * cs[ip]; g4 += KQ(4, ip) * cs[ip]; g5 += KQ(5, ip) * cs[ip];
                    h0 += KQ(0, ip) * cn[ip]; h1 += KQ(1, ip) * cn[ip]; h2 += KQ(2, ip) * cn[ip];
                }
                g5 = q2s * g5 + ((zm >= M - ts) ? wt2 * (cs[5] - gk) : 0.0);
                LSCQP_LANE_WEIGHTS(ZR);
                gcost = e0 * (q2s * g3) + e1 * (q2s * g4) + e2 * g5 + q2s * (tb0 * h0 + tb1 * h1 + tb2 * h2);
                gl = gatherT(XL, ZR);
                ga = gatherT(XA, ZR);
                gcost = zl ? gcost : 0.0;
                gl = zl ? gl : 0.0;
                ga = zl ? ga : 0.0;
            }
            double rdn = fabs(gcost - gl);
            double gls = fmax(fabs(gcost), fabs(gl));
            block_max2(rdn, gls);
            gls = fmax(1.0, gls);
            res_p = max_rp;
            res_d = rdn / gls;
#ifdef LSCQP_TRACE
            tr_wmax = block_max(tr_wmax);
            tr_lmax = block_max(tr_lmax);
            LSCQP_TR(7, gls);
            LSCQP_TR(8, tr_wmax);
            LSCQP_TR(9, tr_lmax);
#endif
            LSCQP_TR(0, max_rp);
            LSCQP_TR(1, res_d);
            LSCQP_TR(2, (sum_sl + sum_pinf));
            LSCQP_TR(3, mu);
            // Infeasible instances (opposing half-spaces, a waypoint out of communication range, limits the start state
            // violates ...) show a primal residual that stays above 1e-2 m and shrinks by less than 30 % over four
            // iterations, for ever; feasible ones are below 1e-4 m by iteration 10 in every class measured (M = 10 with 40
            // neighbours included).  Stop such an instance instead of running it to the iteration limit: the workgroup's
            // launch lasts as long as its slowest QP, and the caller falls back to the initial trajectory anyway
            // (reference src/traj_planner.cpp:767-797).  Tested every fourth iteration from the tenth on.
            // Marginally infeasible instances creep down to ~1e-3 m before they stall while their multipliers run away
            // (sum lambda |r_p| passes 1e6 within a few more iterations): the stall test therefore goes down to 1e-4 m, and a
            // runaway multiplier-weighted residual ends the instance as well.
            if (!repeated && (it & 3) == 2) {
                // (round 2: the stall has to show at two checks in a row -- a feasible instance started far below its multipliers'
                // scale crept for ten iterations and then converged; see mu_scale above)
                // (no shortcut for a flat residual either: a feasible M = 10, 40-neighbour instance sat at 5e-4 m for four iterations
                // around the tenth and converged afterwards -- tools/stress_parity.py, shape 5, seed 114)
                stalled = (it >= 10 && max_rp > 1e-4 && max_rp > 0.7 * (double)rp_ref) ? stalled + 1 : 0;
                if (stalled >= 2 || (it >= 10 && max_rp > 1e-5 && sum_pinf > 1e6)) {  // uniform over the QP's lanes
                    status = LSCQP_STATUS_INFEASIBLE;
                    break;
                }
                rp_ref = (float)max_rp;
            }
            // stop: primal residual (metres), scaled stationarity, and duality gap + multiplier-weighted primal
            // residual in objective units (the latter is what bounds the objective error to first order)
            // The stationarity residual has a rounding floor of ~eps * cond(Hred) * |grad| (cond up to 3e6 at M = 10),
            // which can sit between the strict target and 1e-8: a point that satisfies the primal and gap tests and
            // whose stationarity has been below 1e-8 (the stated KKT tolerance) for two iterations is accepted too,
            // and so is such a point when the next factorisation breaks down (W = lambda/s spans > 1e20 by then).
            // A point that meets the primal and gap tests while its stationarity is still above 1e-8 (cond(Hred) ~ 1e7 at
            // M = 10 in 3-D leaves a rounding floor of ~2e-8) is remembered as well (floor_cnt): if the iteration later
            // stalls or breaks down numerically, the result is accepted rather than reported as a failure.
            if (repeated) {
                // (nothing: see `repeated`)
            } else if (max_rp <= 1e-9 && rdn <= 1e-6 * gls) {  // uniform over the QP's lanes
                LSCQP_PHASE_LANE(lvo_);
                obj_abs = fabs(objective(false, lvo_));
                res_gap = (sum_sl + sum_pinf) / (1.0 + obj_abs);
                if (res_gap <= tol || (res_gap <= 10.0 * tol && rdn <= 1e-7 * gls)) {
                    // remember the point: the fallback exits return a REMEMBERED iterate (tested), not whatever the iteration
                    // moved on to afterwards -- and of the remembered ones the best: a point that meets the gap target beats one
                    // within a decade of it, and among equals the smaller stationarity residual wins.  (Round 4, from the
                    // per-iteration traces of the 11 flagged instances of BASELINE configs[3], tools/floor_probe.py: the
                    // stationarity of a degenerate instance -- |dz| ~ sqrt(mu) -- gets WORSE as mu falls, by c eps max(lambda/s) |dz|
                    // per step, so the latest point was often not the best one.)
                    const bool at_target = res_gap <= tol, had_target = snap_gap <= tol;
                    if (floor_cnt == 0 || (at_target && !had_target) || (at_target == had_target && res_d < snap_d)) {
                        LSCQP_PHASE_LANE(lvs_);
                        if (lvs_ < NZ) zs_[lvs_] = z_[lvs_];
                        snap_p = max_rp;
                        snap_d = res_d;
                        snap_gap = res_gap;
                    }
                }
                if (res_gap <= tol) {
                    floor_cnt++;
                    if (rdn <= 1e-8 * gls) {
                        near_cnt++;
                        if (rdn <= 10.0 * tol * gls || near_cnt >= LSCQP_NEAR_CONFIRM) {
                            LSCQP_TR(6, 1.0);
                            status = LSCQP_STATUS_OPTIMAL;
                            break;
                        }
                    }
                } else if (res_gap <= 10.0 * tol && rdn <= 1e-7 * gls) {
                    // (stationarity within a decade of the stated 1e-8: at nz = 84 the factorisation can break down with the gap
                    // at 1.3x its target and the stationarity at 1.5e-8 -- x within 3e-9 m of the oracle's, objective within 1e-11 --
                    // seen in the 1024-QP batch of BASELINE configs[3])
                    // duplicated active rows (generateCLSC puts one plane on all six control points of the last segment, and
                    // the end stop makes three of them the same variable) leave the gap hovering just above the target until
                    // the factorisation breaks down: such a point -- KKT residuals within the stated 1e-8, objective within
                    // 1e-9 -- is kept as a fallback as well, never as a reason to stop
                    floor_cnt++;
                }
                // A warm-started iteration can jam: residuals at machine precision, the gap stuck near 1e-6 because some rows sit
                // at the boundary with the wrong member of their (slack, multiplier) pair at zero -- a few instances in ten
                // thousand, which the default start solves in a handful of iterations (tests/golden/warm_start_jam.json).  If the
                // gap has not improved tenfold within six such iterations the row state is re-centred ONCE at the current control
                // points (every product back to mu0) and the iteration carries on from there.
                // (counted only for an iteration that is already close, gap <= 1e-4: before that many iterations are legitimately
                // spent bringing a large gap down)
#ifndef LSCQP_NO_RECENTRE  // (A/B switch; carrying the test costs ~1 % of a step, measured with tools/_bis builds)
                // (round 2: counted from the default start too -- 2 of 67 000 instances of a 100-seed sweep from hover sat at a gap
                // of 1e-6 for fifty iterations, 4e-6 m from the optimum; the tenfold-in-six test is cumulative, so the slow linear
                // phase of a hard instance, 6-8x per iteration, does not trip it)
                if (res_gap > tol && res_gap <= 1e-4) {
                    if ((float)res_gap <= 0.1f * gap_mark) {
                        gap_mark = (float)res_gap;
                        jam_since = 0;
                    } else {
                        jam_since++;
                    }
                }
#endif
            } else
                res_gap = sum_sl + sum_pinf;
            // (never once a point that meets the gap target has been seen: then the iteration is polishing its stationarity
            // residual, not jammed, and that point is the fallback the exits below rely on)
#ifndef LSCQP_NO_RECENTRE
            // Safety net of the TIGHT warm start (lscqp_class_desc.warm_start = LSCQP_WARM_TIGHT): it presumes that the caller's initial trajectory is
            // close to the optimum with the right rows near their bounds.  Where it is not, the first step is short (the iterate is
            // boxed in by slacks of 3 mm) and the iteration would crawl; such an instance goes back to the round-1 centring after that
            // one iteration.  (tools/proto_pdip.py `early_recentre`: on 60 instances dumped from the forest10 closed loop 10.6 -> 6.1
            // iterations with it, two instances +2; without it single instances take 18-23.)
            repeated = false;
            const bool net = cls.warm_net > 0 && x_init != nullptr && it == 1 && !net_done && (double)alpha_first < cls.warm_net;
            if (net || (!recentred && jam_since >= 6 && floor_cnt == 0)) {  // uniform over the QP's lanes
                double cnt_ = 0;
                bool bad_ = false;
                centre_rows(1e-3 * mu_scale, 0.03, cnt_, bad_);
                if (net) net_done = true; else recentred = true;
                rp_ref = 3.0e38f;
                near_cnt = 0;
                LSCQP_BLOCK_SYNC();
                continue;
            }
#endif
            LSCQP_MARK(12);
            LSCQP_T(2);
            LSCQP_STOP(3)

            // ============ assemble own row of Hred = T'(H + G'WG)T ==========================================
            // A non-positive pivot is rounding, not the matrix (Hred is SPD): once lambda / s of the active rows spans ~1e15 -- a row that
            // a long step left 1e3 below the average product does that at mu = 1e-12 already -- the factorisation's noise, eps max|K|,
            // reaches the smallest eigenvalues of Hred (~0.1: cond 3e6).  Round 3 ended such an instance (NUMERIC unless a point had
            // been remembered); now the iteration is REPEATED (see below the factorisation) with a diagonal shift of 1e-14, then
            // 1e-12, times the largest weight that went into the matrix, kept for the rest of the solve: residuals and stopping tests
            // are exact whatever the direction, so a shifted direction can cost an iteration, never accuracy.
            // (tools/floor_probe.py, seed 228: an M = 10 instance 17x from its gap target with a stationarity of 7e-11 lost a pivot.)
            double diag_shift = 0.0;
            if (shift_level > 0) {  // uniform over the QP's lanes; rare
                LSCQP_PHASE_LANE(lvr_);
                double big = 0.0;
                for (int e = lvr_; e < 6 * P; e += T) big = fmax(big, fabs(S_[e]));
                for (int e = lvr_; e < C::NOM; e += T) big = fmax(big, om_[e]);
                big = block_max(big);
                diag_shift = (shift_level == 1 ? 1e-14 : 1e-12) * fmax(big, 1.0);
                LSCQP_TR(11, (double)shift_level);
            }
            {
                LSCQP_Z_ROLES();
                double q2 = q2s;  // opaque as well: q2*KQ(i,j) would otherwise be hoisted as 21 VGPR pairs
                asm volatile("" : "+v"(q2));
                const int sd = (zk == 0) ? 0 : (zk == 1) ? 3 : 5;  // S diagonal entry of axis zk
                // local 6x6 block (upper triangle) of (axis zk, segment m): 2 w_c Q + terminal + interval/vel/acc
                // weights + the LSC same-axis diagonal
                auto local_block = [&](int m, double (&B)[6][6]) {
#pragma unroll
                    for (int i = 0; i < 6; i++)
#pragma unroll
                        for (int ip = i; ip < 6; ip++) B[i][ip] = q2 * KQ(i, ip);
                    B[5][5] += (m >= M - ts) ? wt2 : 0.0;
                    const double* omi = &om_[zk * P + 6 * m];
                    const double* omv = &om_[C::OV + zk * 5 * M + 5 * m];
                    const double* oma = &om_[C::OA + zk * 4 * M + 4 * m];
#pragma unroll
                    for (int i = 0; i < 6; i++) B[i][i] += omi[i] + S_[(6 * m + i) * 6 + sd];
#pragma unroll
                    for (int i = 0; i < 5; i++) {
                        const double w = omv[i];
                        B[i][i] += w; B[i + 1][i + 1] += w; B[i][i + 1] -= w;
                    }
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const double w = oma[i];
                        B[i][i] += w; B[i][i + 1] -= 2.0 * w; B[i][i + 2] += w;
                        B[i + 1][i + 1] += 4.0 * w; B[i + 1][i + 2] -= 2.0 * w; B[i + 2][i + 2] += w;
                    }
                };
                auto sym = [](const double (&B)[6][6], int i, int ip) -> double { return i <= ip ? B[i][ip] : B[ip][i]; };
                LSCQP_LANE_WEIGHTS(ZR);
                const double ej[3] = {e0, e1, e2};
                const double tbj[3] = {tb0, tb1, tb2};
                const int mn = has_next ? zm + 1 : zm;  // clamped: its weights tb* are zero when there is no next
                double own[3], prv[3], nxt[3];
                {
                    double Bm[6][6];
                    local_block(zm, Bm);
                    double bi[3];
#pragma unroll
                    for (int i = 0; i < 3; i++) bi[i] = ej[0] * sym(Bm, i, 3) + ej[1] * sym(Bm, i, 4) + ej[2] * sym(Bm, i, 5);
#pragma unroll
                    for (int jp = 0; jp < 3; jp++) {
                        own[jp] = ej[0] * sym(Bm, 3, 3 + jp) + ej[1] * sym(Bm, 4, 3 + jp) + ej[2] * sym(Bm, 5, 3 + jp);
                        prv[jp] = TBc(0, jp) * bi[0] + TBc(1, jp) * bi[1] + TBc(2, jp) * bi[2];
                    }
                }
                {
                    double Bn[6][6];
                    local_block(mn, Bn);
                    double ta[3];
#pragma unroll
                    for (int ip = 0; ip < 3; ip++) ta[ip] = tbj[0] * sym(Bn, 0, ip) + tbj[1] * sym(Bn, 1, ip) + tbj[2] * sym(Bn, 2, ip);
#pragma unroll
                    for (int jp = 0; jp < 3; jp++) {
                        own[jp] += ta[0] * TBc(0, jp) + ta[1] * TBc(1, jp) + ta[2] * TBc(2, jp);
                        nxt[jp] = tbj[0] * sym(Bn, 0, 3 + jp) + tbj[1] * sym(Bn, 1, 3 + jp) + tbj[2] * sym(Bn, 2, 3 + jp);
                    }
                }
                // stores: every iteration overwrites exactly the same pattern entries, the rest of the row stays 0.
                // Merged columns (the single end-stop variable) receive the sum.
                FT* hk = &hrow[zk * NZA];
                const bool nlast = ES && has_next && (zm + 1 == M - 1);  // the next block is the end-stop variable
                const double osum = own[0] + own[1] + own[2], nsum = nxt[0] + nxt[1] + nxt[2];
                const int ob = zlast ? 3 * (M - 1) : 3 * zm;
                hk[ob] = zlast ? osum : own[0];
                if (!zlast) hk[ob + 1] = own[1];
                if (!zlast) hk[ob + 2] = own[2];
                const int pb = 3 * (zm >= 1 ? zm - 1 : 0);
                if (zm >= 1) hk[pb + 0] = prv[0];
                if (zm >= 1) hk[pb + 1] = prv[1];
                if (zm >= 1) hk[pb + 2] = prv[2];
                const int nb = nlast ? 3 * (M - 1) : 3 * mn;
                if (has_next) hk[nb] = nlast ? nsum : nxt[0];
                if (has_next && !nlast) hk[nb + 1] = nxt[1];
                if (has_next && !nlast) hk[nb + 2] = nxt[2];
                // cross-axis blocks come only from the LSC rows, block-diagonal in the segment index
#pragma unroll
                for (int l = 0; l < DIM; l++) {
                    const int a_ = zk < l ? zk : l, b_ = zk < l ? l : zk;
                    const int so = (a_ == 0) ? (b_ == 0 ? 0 : b_) : (b_ == 1 ? 3 : 4);  // (0,1)->1 (0,2)->2 (1,2)->4
                    double cr[3];
#pragma unroll
                    for (int jp = 0; jp < 3; jp++)
                        cr[jp] = ej[jp] * S_[(6 * zm + 3 + jp) * 6 + so] + tbj[0] * TBc(0, jp) * S_[(6 * mn + 0) * 6 + so] +
                                 tbj[1] * TBc(1, jp) * S_[(6 * mn + 1) * 6 + so] + tbj[2] * TBc(2, jp) * S_[(6 * mn + 2) * 6 + so];
                    FT* hl = &hrow[l * NZA];
                    const bool wr = (l != zk);
                    if (wr) hl[ob] = zlast ? (cr[0] + cr[1] + cr[2]) : cr[0];
                    if (wr && !zlast) hl[ob + 1] = cr[1];
                    if (wr && !zlast) hl[ob + 2] = cr[2];
                }
                // communication pairs couple the c5 variables of one axis: adjacent ones land on band entries
                // (read-modify-write after the stores above, LDS is in order), far ones are plain stores
                {
                    const bool c5 = (zlast || zj == 2) && comm_on_k;
                    const double* omc = &om_[C::OC + zk * C::NCP];
                    double dsum = 0;
#pragma unroll
                    for (int up = 0; up < M; up++) {
                        const bool use = c5 && (up != zm);
                        const int hi_ = zm > up ? zm : up, lo_ = zm > up ? up : zm;
                        const double w = use ? omc[hi_ * (hi_ - 1) / 2 + (use ? lo_ : 0)] : 0.0;
                        dsum += w;
                        FT* t = &hk[zidx(up, 2)];
                        const bool near = (up == zm - 1) || (up == zm + 1);
                        const double old = (double)*t;
                        if (use) *t = (FT)(near ? (old - w) : -w);
                    }
                    hrow[zi] += (FT)(dsum + diag_shift);  // dsum == 0 for lanes that are not c5 variables (zi = 0 for lanes without a variable: the dummy row)
                }
                // W > 1 with nz <= 64: the system fits wavefront 0, whose lanes assembled it; EVERY wavefront loads the same rows
                // (lane & 63) and factorises the same matrix redundantly.  The verdict on a failed pivot is then identical in all
                // wavefronts by construction -- a break taken by wavefront 0 alone would leave the others iterating against
                // mismatched barriers until the iteration limit (seen: one QP of a 512-QP dense-maze batch held its launch for
                // 1.6 ms instead of 0.25 ms) -- at the price of one barrier, with no flag to carry across the solves (measured
                // alternative: wavefront 0 posts a flag that all read behind the predictor solve, +5 % on the 64-QP step).
                if constexpr (C::ND) {
                    // nested dissection: wavefront 0 holds rows L | S over columns L | S, wavefront 1 rows R | (accumulator of S)
                    // over columns R | S.  The accumulator lane of separator variable s starts from K[s][R] (assembled by the S
                    // lane of wavefront 0 in ITS scratch row) and an all-zero separator block.
                    __syncthreads();
                    const int wv_ = __builtin_amdgcn_readfirstlane(lvz_ >> 6);
                    const int t_ = lvz_ & 63;
                    const bool acc = (wv_ == 1) && t_ >= C::NB1 && t_ < C::NR1;  // accumulator lane
                    const bool rowl = (wv_ == 0 && t_ < C::NR0) || (wv_ == 1 && t_ < C::NB1);
                    // (the lane of separator variable s in wavefront 0 is NB0 + s, and that is its slot)
                    const int sacc = C::NB0 + (acc ? ((!C::ND_COMPACT || (t_ - C::NB1) < 2 * DIM) ? (t_ - C::NB1) : (t_ - C::NB1) + C::MS * DIM) : 0);
                    const FT* const src = &Hs[(acc ? sacc : (rowl ? ZR.slot : NZ)) * LDH];
                    if (wv_ == 0) {
                        static_for<0, C::NAR>([&](auto Cc) {
                            constexpr int c = decltype(Cc)::value;
                            if constexpr (c < C::NR0) {
                                const FT v = src[C::nd_zi(0, c)];
                                A[c] = rowl ? v : (FT)0;
                            } else {
                                A[c] = (FT)0;
                            }
                        });
                    } else {
                        static_for<0, C::NAR>([&](auto Cc) {
                            constexpr int c = decltype(Cc)::value;
                            if constexpr (c < C::NR1) {
                                const FT v = src[C::nd_zi(1, c)];
                                A[c] = (rowl || (acc && c < C::NB1)) ? v : (FT)0;
                            } else {
                                A[c] = (FT)0;
                            }
                        });
                    }
                    __syncthreads();  // the scratch matrix is reused for the hand-overs of the factorisation
                } else if constexpr (W > 1 && NZ <= 64) {
                    __syncthreads();
                    const int rsrc = lvz_ & 63;
                    const bool zrow = rsrc < NZ;
                    const FT* const src = &Hs[(zrow ? rsrc : NZ) * LDH];
#pragma unroll
                    for (int cidx = 0; cidx < NZ; cidx++) {
                        const FT v = src[cidx];
                        A[cidx] = zrow ? v : (FT)0;
                    }
                } else {
                    LSCQP_WAVE_LDS_SYNC();
#pragma unroll
                    for (int cidx = 0; cidx < NZ; cidx++) {
                        const FT v = hrow[cidx];
                        A[cidx] = zl ? v : (FT)0;
                    }
                }
            }
            LSCQP_T(3);

            // ============ LDL^T in registers: lane i holds row i ===============================================
            bool pivot_bad = false;
            if constexpr (!C::ND) {
#ifndef LSCQP_FACT_READLANE
            // Pivot-row broadcast over BOTH pipes.  A v_readlane costs ~8 cycles of VALU issue (16 per fp64 value, next
            // to 4.6 for the FMA it feeds); a uniform-address ds_read_b64 costs ~11 cycles of the LDS pipe and none of
            // the VALU's (tools/ubench.hip).  By symmetry of the trailing matrix the pivot row of step j equals the
            // pivot COLUMN, of which every lane holds one entry (its A[j]); each step publishes column j+1 to LDS as
            // soon as it is final, the first ~1/4 of the next row is broadcast with v_readlane (no LDS dependency, hides
            // the LDS latency) and the rest is read back from LDS while the VALU works.  The pivot reciprocal of step
            // j+1 is started inside step j as well.
            {
                int lf = lane & 63;  // (lane of the wavefront: see the identity rows of a second wavefront)
                asm volatile("" : "+v"(lf));
                FT* const colw = reinterpret_cast<FT*>(col_) + (W > 1 ? 2 * 64 * (lane >> 6) : 0);  // each wavefront its own column buffer
                colw[lf] = A[0];
                FT d = bcast(A[0], 0);
                FT invd = fast_rcp(d);
                static_for<0, NZ>([&](auto Jc) {
                    constexpr int j = decltype(Jc)::value;
                    constexpr int n = NZ - j - 1;  // trailing entries k = j+1 .. NZ-1
                    constexpr int r0 = (n * LSCQP_FACT_HYBRID + 50) / 100;
                    constexpr int nr = n <= 4 ? n : (r0 < 3 ? 4 : r0 + 1);  // via v_readlane (k = j+1 always)
                    constexpr int nl = n - (nr < n ? nr : n);              // via LDS
                    constexpr int NR = n - nl;
                    const FT* const cb = colw + (j & 1) * 64;
                    FT* const cbn = colw + ((j + 1) & 1) * 64;
                    pivot_bad = pivot_bad || !pivot_ok<FT>(d);
                    dinv_own = (lf == j) ? invd : dinv_own;
                    const FT li = (lf > j) ? A[j] * invd : (FT)0;
                    FT ul[nl > 0 ? nl : 1];
                    static_for<0, nl>([&](auto Tc) {
                        constexpr int t = decltype(Tc)::value;
                        ul[t] = cb[j + 1 + NR + t];
                    });
                    if constexpr (NR > 0) {
                        FT ur[NR];
                        static_for<0, NR>([&](auto Tc) {
                            constexpr int t = decltype(Tc)::value;
                            ur[t] = bcast(A[j + 1 + t], j);
                        });
                        A[j + 1] = fma(-li, ur[0], A[j + 1]);
                        cbn[lf] = A[j + 1];
                        d = bcast(A[j + 1], j + 1);
                        invd = fast_rcp(d);
                        static_for<1, NR>([&](auto Tc) {
                            constexpr int t = decltype(Tc)::value;
                            A[j + 1 + t] = fma(-li, ur[t], A[j + 1 + t]);
                        });
                    }
                    static_for<0, nl>([&](auto Tc) {
                        constexpr int t = decltype(Tc)::value;
                        A[j + 1 + NR + t] = fma(-li, ul[t], A[j + 1 + NR + t]);
                    });
                    A[j] = (lf > j) ? li : A[j];
                    asm volatile("" ::: "memory");  // LDS program order between the steps (no wait: the LDS runs in order)
                });
            }
#else
            // Measured alternative (-DLSCQP_FACT_READLANE, 0.153 vs 0.148 ms per 64-QP batch): the whole pivot row of
            // lane j broadcast with v_readlane (2 per fp64 value), issued in batches of BB into distinct
            // scalar registers so the v_readlane -> v_fma hazard slots are filled by other broadcasts, not s_nops
            // The lane-vs-column comparisons below are loop invariant; compared against `lane` itself hipcc hoists all
            // 3*nz of them out of the iteration loop, which needs 2 SGPRs each, spills those into VGPR lanes
            // (v_writelane) and pays a v_readlane + hazard nops per use.  An opaque copy of the lane id per phase keeps
            // them where they are used: one v_cmp each.
            int lf = lane & 63;
            asm volatile("" : "+v"(lf));
            static_for<0, NZ>([&](auto Jc) {
                constexpr int j = decltype(Jc)::value;
                const FT d = bcast(A[j], j);
                pivot_bad = pivot_bad || !pivot_ok<FT>(d);
                const FT invd = fast_rcp(d);
                dinv_own = (lf == j) ? invd : dinv_own;
                const FT li = (lf > j) ? A[j] * invd : (FT)0;
#ifdef LSCQP_FACT_BB
                constexpr int BB = LSCQP_FACT_BB;
#else
                constexpr int BB = 8;  // measured on MI355X: 1 -> 0.231 ms, 4 -> 0.214 ms, 8 -> 0.210 ms per 64-QP batch
#endif
                constexpr int NCH = (NZ - j - 1 + BB - 1) / BB;
                static_for<0, NCH>([&](auto Cc) {
                    constexpr int k0 = j + 1 + decltype(Cc)::value * BB;
                    FT ub[BB];
                    static_for<0, BB>([&](auto Tc) {
                        constexpr int t = decltype(Tc)::value;
                        if constexpr (k0 + t < NZ) ub[t] = bcast(A[k0 + t], j);
                    });
                    static_for<0, BB>([&](auto Tc) {
                        constexpr int t = decltype(Tc)::value;
                        if constexpr (k0 + t < NZ) A[k0 + t] = fma(-li, ub[t], A[k0 + t]);
                    });
                });
                A[j] = (lf > j) ? li : A[j];
            });
#endif
            } else {
                // nz > 64: NESTED DISSECTION over two wavefronts (layout: Cfg).
                //   1. both wavefronts eliminate their own banded block (NB pivots, concurrently, no barrier): per pivot the
                //      in-block band (<= BWB columns) and the NS separator columns; the separator rows of wavefront 0 and the
                //      accumulator rows of wavefront 1 collect the two shares of the Schur complement on the way;
                //   2. ONE hand-over through LDS: wavefront 1's share of the separator block is added to wavefront 0's;
                //   3. wavefront 0 factorises the NS x NS separator (dense, pivot row over both pipes as in the one-wavefront case).
                // Pivot rows are read as pivot COLUMNS (symmetry) from a per-wavefront LDS column buffer; the column of the next
                // pivot is published as soon as it is final.  The branches on the wavefront id are scalar.
                constexpr int NB0 = C::NB0, NB1 = C::NB1, NR0 = C::NR0, NR1 = C::NR1, NS = C::NS, BWB = C::BWB;
                const int wv = __builtin_amdgcn_readfirstlane(lane >> 6);
                int lf = lane & 63;
                asm volatile("" : "+v"(lf));
                double* const colw = col_ + 2 * 64 * (wv & 1);
                double* const Sx = reinterpret_cast<double*>(Hs);  // [NS][NS]: wavefront 1's share of the separator block
                double* const flag = Sx + NS * NS;                  // [2] pivot failures
                // (one instantiation per wavefront: the separator columns a block can reach differ, Cfg::nd_touched.  The separator
                // entries of the pivot row go over BOTH pipes: every second one is read from lane j's own registers with
                // v_readlane, the others from the published pivot column in LDS -- two wavefronts share the CU's one LDS pipe
                // here, and with all of them on it the phase was LDS bound: 34.7 k cycles for 24 pivots.)
                auto block_phase = [&](auto Wc) {
                    constexpr int w_ = decltype(Wc)::value;
                    constexpr int NB = C::nd_nb(w_);
                    colw[lf] = A[0];
                    double d = bcast(A[0], 0);
                    double invd = fast_rcp(d);
                    static_for<0, NB>([&](auto Jc) {
                        constexpr int j = decltype(Jc)::value;
                        constexpr int b1 = (j + BWB < NB - 1) ? j + BWB : NB - 1;  // last in-block column of the band
                        const double* const cb = colw + (j & 1) * 64;
                        double* const cbn = colw + ((j + 1) & 1) * 64;
                        pivot_bad = pivot_bad || !pivot_ok<double>(d);
                        dinv_own = (lf == j) ? invd : dinv_own;
                        const double li = (lf > j) ? A[j] * invd : 0.0;
                        if constexpr (j + 1 < NB) {  // the next pivot column first: its broadcast and reciprocal overlap the rest
                            A[j + 1] = fma(-li, bcast(A[j + 1], j), A[j + 1]);
                            cbn[lf] = A[j + 1];
                            d = bcast(A[j + 1], j + 1);
                            invd = fast_rcp(d);
                        }
                        static_for<j + 2, b1 + 1>([&](auto Kc) {
                            constexpr int k = decltype(Kc)::value;
                            A[k] = fma(-li, cb[k], A[k]);
                        });
                        static_for<0, NS>([&](auto Kc) {
                            constexpr int sc = decltype(Kc)::value;
                            if constexpr (C::nd_touched(w_, sc)) {
                                constexpr int k = C::nd_col(w_, sc);
                                const double u = (sc & 1) ? cb[k] : bcast(A[k], j);
                                A[k] = fma(-li, u, A[k]);
                            }
                        });
                        A[j] = (lf > j) ? li : A[j];
                        asm volatile("" ::: "memory");  // LDS program order between the steps
                    });
                };
                if (wv == 0) {
                    LSCQP_MARKP(8);
                    block_phase(std::integral_constant<int, 0>{});
                    LSCQP_MARKP(11);
                }
                if (wv == 1) {
                    LSCQP_MARKP(9);
                    block_phase(std::integral_constant<int, 1>{});
                    LSCQP_MARKP(11);
                }
                LSCQP_T(14);  // (development timing: the block phase of the nested dissection)
                {   // hand-over of wavefront 1's share (single predicated stores; everything else is masked arithmetically)
                    // Sx[s][c], s and c separator variables wavefront 1 reaches; the other rows / columns are never written
                    const bool give = (wv == 1) && lf >= NB1 && lf < NR1;
                    const int r1 = give ? lf - NB1 : 0;
                    const int srow1 = (!C::ND_COMPACT || r1 < 2 * DIM) ? r1 : r1 + C::MS * DIM;  // Cfg::nd_unrank1
                    static_for<0, NS>([&](auto Cc) {
                        constexpr int c = decltype(Cc)::value;
                        if constexpr (C::nd_touched(1, c)) {
                            if (give) Sx[srow1 * NS + c] = A[C::nd_col(1, c)];
                        }
                    });
                    if (lf == 0 && wv < 2) flag[wv] = pivot_bad ? 1.0 : 0.0;
                    __syncthreads();
                    const bool sep0 = (wv == 0) && lf >= NB0 && lf < NR0;
                    const int srow0 = sep0 ? lf - NB0 : 0;
                    if constexpr (C::ND_COMPACT) {  // (rows of separator variables wavefront 1 cannot reach were not written)
                        const bool take = sep0 && (srow0 < 2 * DIM || srow0 >= 2 * DIM + C::MS * DIM);  // Cfg::nd_touched(1, srow0)
                        const double* const sxr = Sx + (take ? srow0 : 0) * NS;
                        static_for<0, NS>([&](auto Cc) {
                            constexpr int c = decltype(Cc)::value;
                            if constexpr (C::nd_touched(1, c)) {
                                const double v = sxr[c];
                                A[NB0 + c] += take ? v : 0.0;
                            }
                        });
                    } else {  // (... hold zeros there)
                        const double take = sep0 ? 1.0 : 0.0;
                        static_for<0, NS>([&](auto Cc) {
                            constexpr int c = decltype(Cc)::value;
                            if constexpr (C::nd_touched(1, c)) A[NB0 + c] = fma(take, Sx[srow0 * NS + c], A[NB0 + c]);
                        });
                    }
                }
                if (wv == 0) {  // the separator: dense LDL^T on columns / lanes NB0 .. NR0-1
                    LSCQP_MARKP(8);
                    constexpr int NB = NB0, NA = NR0;
                    colw[lf] = A[NB];
                    double d = bcast(A[NB], NB);
                    double invd = fast_rcp(d);
                    static_for<NB, NA>([&](auto Jc) {
                        constexpr int j = decltype(Jc)::value;
                        constexpr int n = NA - j - 1;
                        constexpr int r0 = (n * LSCQP_FACT_HYBRID + 50) / 100;
                        constexpr int nr = n <= 4 ? n : (r0 < 3 ? 4 : r0 + 1);
                        constexpr int nl = n - (nr < n ? nr : n);
                        constexpr int NR = n - nl;
                        const double* const cb = colw + (j & 1) * 64;
                        double* const cbn = colw + ((j + 1) & 1) * 64;
                        pivot_bad = pivot_bad || !pivot_ok<double>(d);
                        dinv_own = (lf == j) ? invd : dinv_own;
                        const double li = (lf > j) ? A[j] * invd : 0.0;
                        double ul[nl > 0 ? nl : 1];
                        static_for<0, nl>([&](auto Tc) {
                            constexpr int t = decltype(Tc)::value;
                            ul[t] = cb[j + 1 + NR + t];
                        });
                        if constexpr (NR > 0) {
                            double ur[NR];
                            static_for<0, NR>([&](auto Tc) {
                                constexpr int t = decltype(Tc)::value;
                                ur[t] = bcast(A[j + 1 + t], j);
                            });
                            A[j + 1] = fma(-li, ur[0], A[j + 1]);
                            cbn[lf] = A[j + 1];
                            d = bcast(A[j + 1], j + 1);
                            invd = fast_rcp(d);
                            static_for<1, NR>([&](auto Tc) {
                                constexpr int t = decltype(Tc)::value;
                                A[j + 1 + t] = fma(-li, ur[t], A[j + 1 + t]);
                            });
                        }
                        static_for<0, nl>([&](auto Tc) {
                            constexpr int t = decltype(Tc)::value;
                            A[j + 1 + NR + t] = fma(-li, ul[t], A[j + 1 + NR + t]);
                        });
                        A[j] = (lf > j) ? li : A[j];
                        asm volatile("" ::: "memory");
                    });
                    if (lf == 0) flag[0] = (pivot_bad || flag[0] != 0.0) ? 1.0 : 0.0;
                    LSCQP_MARKP(11);
                }
                __syncthreads();
                pivot_bad = (flag[0] != 0.0) || (flag[1] != 0.0);
                __syncthreads();  // the flags sit in the scratch matrix, which the next phase overwrites
            }
            LSCQP_STOP(4)
            // (not in the mixed-precision instances: a float32 factorisation that loses a pivot hands the instance to the fp64 second
            // pass, which is what that pass exists for -- retrying in float32 first cost configs[4]-mixed 0.71 -> 0.89 ms)
            if (pivot_bad && shift_level < 2 && !MIXED) {  // uniform over the QP's lanes: repeat this iteration with a (larger) diagonal shift
                shift_level++;
                flags |= LSCQP_INFO_SHIFTED;
                LSCQP_TR(6, 2.0);
                // the scratch rows must be all-zero outside the assembly pattern again (nested dissection reused them for its hand-overs)
                LSCQP_PHASE_LANE(lvr_);
                FT* const hz = &Hs[lane_slot(lvr_) * LDH];
#pragma unroll
                for (int cidx = 0; cidx < NZ; cidx++) hz[cidx] = (FT)0;
                LSCQP_BLOCK_SYNC();
                repeated = true;
                it--;  // (the loop's increment makes it this iteration again)
                continue;
            }
            auto numeric_exit = [&]() {
                status = (near_cnt > 0 || floor_cnt > 0) ? LSCQP_STATUS_OPTIMAL : LSCQP_STATUS_NUMERIC;
                restore = status == LSCQP_STATUS_OPTIMAL;
            };
            if (pivot_bad) {  // uniform over the QP's lanes (nz <= 64 with several wavefronts: every wavefront factorises the same matrix)
                LSCQP_TR(6, 2.0);
                numeric_exit();
                break;
            }
            // broadcast of lane j's value to all lanes of the QP: v_readlane (W = 1) or an LDS slot + barrier (W = 2; one
            // slot per column and direction, so no slot is rewritten while a slower wavefront may still read it)
            auto bcast_q = [&](FT v, int j, int ls, double* slots) -> FT {
                if constexpr (!C::ND) {
                    return bcast(v, j);
                } else {
                    if (ls == j) slots[j] = v;
                    __syncthreads();
                    return (FT)slots[j];
                }
            };
#ifdef LSCQP_SOLVE_FROM_LDS
            // the factor is parked in the lane's scratch-matrix row right after the factorisation and both solves read
            // it from there (lane-private row, conflict-free ds_read_b64, independent of the solve's dependency chain)
            {
                LSCQP_PHASE_LANE(lvp_);
                FT* const hrow = &Hs[(lvp_ < NZ ? lvp_ : NZ) * LDH];
#pragma unroll
                for (int cidx = 0; cidx < NZ; cidx++) hrow[cidx] = A[cidx];
                LSCQP_WAVE_LDS_SYNC();
            }
#define LSCQP_FACTOR_ENTRY(j) hr[j]
#else
#define LSCQP_FACTOR_ENTRY(j) A[j]
#endif
            // nz > 64: the triangular solves of the nested-dissection factor.  Forward: both wavefronts substitute through their
            // own block concurrently (the separator / accumulator rows collect the two shares of the separator's right-hand side),
            // ONE hand-over, wavefront 0 runs forward and backward through the separator (its own block rows follow along in the
            // same broadcasts), ONE hand-over of the separator's solution, both wavefronts finish their blocks backward.
            // (generic lambda: instantiated only for the nz > 64 instances; called twice: must not become a real call)
            auto solve_blocked = [&](double b, auto) __attribute__((always_inline)) -> double {
                constexpr int NB0 = C::NB0, NB1 = C::NB1, NR0 = C::NR0, NR1 = C::NR1, NS = C::NS;
                const int wv = __builtin_amdgcn_readfirstlane(lane >> 6);
                int ll_ = lane & 63;
                asm volatile("" : "+v"(ll_));
                double* const hand = col_;  // [64] hand-over buffer, indexed by separator variable (the column buffers are idle during the solves)
                const bool sep0 = ll_ >= NB0 && ll_ < NR0;   // separator lanes of wavefront 0
                const int srow0 = sep0 ? ll_ - NB0 : 0;
                const bool acc1 = ll_ >= NB1 && ll_ < NR1;   // accumulator lanes of wavefront 1
                const int r1 = acc1 ? ll_ - NB1 : 0;
                const int srow1 = (!C::ND_COMPACT || r1 < 2 * DIM) ? r1 : r1 + C::MS * DIM;  // Cfg::nd_unrank1
                auto block_forward = [&](auto Nc) {
                    static_for<0, decltype(Nc)::value>([&](auto Jc) {
                        constexpr int j = decltype(Jc)::value;
                        const double wj = bcast(b, j);
                        b = fma(-((ll_ > j) ? A[j] : 0.0), wj, b);
                    });
                };
                auto block_backward = [&](auto Nc) {
                    constexpr int NB = decltype(Nc)::value;
                    asm volatile("" : "+v"(ll_));
                    static_for<0, NB>([&](auto Jc) {
                        constexpr int j = NB - 1 - decltype(Jc)::value;
                        const double xj = bcast(b * dinv_own, j);
                        b = fma(-((ll_ < j) ? A[j] : 0.0), xj, b);
                    });
                };
                // ---- forward through the blocks: L w = b (unit lower) ----  (equal blocks: one copy of the code for both wavefronts --
                // the loop bodies of the M = 10 instances are larger than the instruction cache as it is)
                if constexpr (NB0 == NB1) {
                    if (wv < 2) {
                        LSCQP_MARKP(10);
                        block_forward(std::integral_constant<int, NB0>{});
                        LSCQP_MARKP(11);
                    }
                } else {
                    if (wv == 0) {
                        LSCQP_MARKP(8);
                        block_forward(std::integral_constant<int, NB0>{});
                        LSCQP_MARKP(11);
                    }
                    if (wv == 1) {
                        LSCQP_MARKP(9);
                        block_forward(std::integral_constant<int, NB1>{});
                        LSCQP_MARKP(11);
                    }
                }
                if (wv == 1 && acc1) hand[srow1] = b;  // accumulator rows started from 0: their value IS wavefront 1's share
                __syncthreads();
                if (wv == 0) {
                    LSCQP_MARKP(8);
                    if constexpr (C::ND_COMPACT) {
                        const bool take = sep0 && (srow0 < 2 * DIM || srow0 >= 2 * DIM + C::MS * DIM);  // Cfg::nd_touched(1, srow0)
                        const double v = hand[take ? srow0 : 0];
                        b += take ? v : 0.0;
                    } else {
                        b += sep0 ? hand[srow0] : 0.0;
                    }
                    static_for<NB0, NR0>([&](auto Jc) {  // forward through the separator
                        constexpr int j = decltype(Jc)::value;
                        const double wj = bcast(b, j);
                        b = fma(-((ll_ > j) ? A[j] : 0.0), wj, b);
                    });
                    // ---- backward: (D L') x = w; row i of the upper factor is A[j > i] of lane i ----
                    asm volatile("" : "+v"(ll_));
                    static_for<NB0, NR0>([&](auto Jc) {
                        constexpr int j = NR0 - 1 - (decltype(Jc)::value - NB0);
                        const double xj = bcast(b * dinv_own, j);
                        b = fma(-((ll_ < j) ? A[j] : 0.0), xj, b);  // (the block rows of wavefront 0 take their separator part here)
                    });
                    LSCQP_MARKP(11);
                }
                // (same wavefront, LDS in program order: the shares were read above before the buffer is overwritten here)
                if (wv == 0 && sep0) hand[srow0] = b * dinv_own;  // the separator's solution
                __syncthreads();
                if (wv == 1) {  // block rows of wavefront 1: U[i][S] x_S
                    LSCQP_MARKP(9);
                    static_for<0, NS>([&](auto Cc) {
                        constexpr int c = decltype(Cc)::value;
                        if constexpr (C::nd_touched(1, c)) b = fma(-A[C::nd_col(1, c)], hand[c], b);  // (accumulator lanes compute a value nobody uses)
                    });
                    if constexpr (NB0 != NB1) block_backward(std::integral_constant<int, NB1>{});
                    LSCQP_MARKP(11);
                }
                if constexpr (NB0 == NB1) {
                    if (wv < 2) {
                        LSCQP_MARKP(10);
                        block_backward(std::integral_constant<int, NB0>{});
                        LSCQP_MARKP(11);
                    }
                } else if (wv == 0) {
                    LSCQP_MARKP(8);
                    block_backward(std::integral_constant<int, NB0>{});
                    LSCQP_MARKP(11);
                }
                return b * dinv_own;  // (final once the lane's own column has been broadcast; 0 for lanes without a row)
            };
            // (mixed precision: the right-hand side is rounded to float32, the direction comes back as fp64; no refinement --
            // the residuals the NEXT iteration computes are fp64, so an inexact direction costs iterations, not accuracy:
            // +0.4 iterations on the forest class, tools/proto_fp32.py)
            auto solve = [&](double b64) __attribute__((always_inline)) -> double {
                if constexpr (C::ND) return solve_blocked(b64, 0);
                FT b = (FT)b64;
                int ls = (NZ <= 64) ? (lane & 63) : lane;  // opaque per call, see the factorisation
                asm volatile("" : "+v"(ls));
                const FT* const hr = &Hs[(ls < NZ ? ls : NZ) * LDH];
                (void)hr;
#pragma unroll
                for (int j = 0; j < NZ; j++) {  // L w = b (unit lower)
                    const FT wj = bcast_q(b, j, ls, col_);
                    b = fma(-((ls > j) ? LSCQP_FACTOR_ENTRY(j) : (FT)0), wj, b);
                }
                asm volatile("" : "+v"(ls));
#pragma unroll
                for (int j = NZ - 1; j >= 0; j--) {  // (D L') x = w : row i of the upper factor is A[j>i] of lane i
                    const FT xj = bcast_q(b * dinv_own, j, ls, col_ + T);
                    b = fma(-((ls < j) ? LSCQP_FACTOR_ENTRY(j) : (FT)0), xj, b);
                }
                // lane i's b is final once column i has been broadcast (later columns j < i leave it alone), so its own
                // component needs no per-step select
                return (double)(b * dinv_own);
            };
            LSCQP_T(4);
            LSCQP_STOP(5)

            // ============ predictor ========================================================================
            const double dza = solve(-gcost + ga);
            if (zl) dz_[zi0] = dza;
            LSCQP_BLOCK_SYNC();
#ifndef LSCQP_SOLVE_FROM_LDS
            // park the factor in the lane's scratch-matrix row while pass 2 runs: A[] is then dead across the pass,
            // which removes most register spills of the pass.  (Behind the barrier: in the multi-wavefront instances other
            // wavefronts LOADED their copy of these rows after the assembly; every wavefront has finished its factorisation,
            // hence those loads, when it arrives here.)
            {
                LSCQP_PHASE_LANE(lvp_);
                FT* const prow = park_ptr(lvp_);
#pragma unroll
                for (int cidx = 0; cidx < C::NAR; cidx++) prow[cidx] = A[cidx];
            }
#endif
            expandT(dz_, dca_, false);
            LSCQP_BLOCK_SYNC();
            LSCQP_T(5);
            LSCQP_STOP(6)
            // ============ pass 2: affine step length, mu_aff, corrector right-hand side ======================
            // with w = lam/s, ds = G dc + rp, dl = -lam - w ds:   -ds/s = -t,  -dl/lam = 1 + t,  t = ds/s
            double rmax = 1.0, sB = 0;  // rmax = 1/alpha_aff (>= 1 caps alpha at 1)
#pragma unroll
            for (int u = 0; u < NS2; u++) {
                const bool on = t_ix[u] >= 0;
                const double tsl = t_sl[u].get(), tsh = t_sh[u].get(), tll = t_ll[u].get(), tlh = t_lh[u].get();
                const double tlo = t_lo[u].get(), thi = t_hi[u].get();
                const double y = row_val(c_, u), dy = row_val(dca_, u);
                const double rpl = (y - tlo) - tsl, rph = (thi - y) - tsh;
                const double isl = row_rcp(tsl), ish = row_rcp(tsh);
                const double tl = (dy + rpl) * isl, th = (rph - dy) * ish;
                const double rr = fmax(fmax(-tl, 1.0 + tl), fmax(-th, 1.0 + th));
                rmax = fmax(rmax, on ? rr : 1.0);
                const double pl = -(dy + rpl) * tll * (1.0 + tl), ph = -(rph - dy) * tlh * (1.0 + th);
                sB += on ? (pl + ph) : 0.0;
                const double v1 = isl - ish;
                const double v2 = (-pl - tll * rpl) * isl - (-ph - tlh * rph) * ish;
                if constexpr (W == 1) {
                    row_scatter(XB1, u, v1, on);
                    row_scatter(XB2, u, v2, on);
                } else {
                    sc1[u] = v1;
                    sc2[u] = v2;
                }
            }
            {
                LSCQP_L_ROLES();
                double b10 = 0, b11 = 0, b12 = 0, b20 = 0, b21 = 0, b22 = 0;
                const double cx = c_[lx], cy = c_[P + lx], cz = (DIM == 3) ? c_[2 * P + lx] : 0.0;
                const double dx = dca_[lx], dy = dca_[P + lx], dzz = (DIM == 3) ? dca_[2 * P + lx] : 0.0;
#pragma unroll
                for (int u = 0; u < NSLOT; u++) {
                    const int o = lg + G * u;
                    const int e = (ll && o < n_obs) ? (o * CP + lcp) : NROW;
                    const double nx = Rnx[e], ny = Rny[e], nz = Rnz[e], s = r_s[u].get(), l = r_l[u].get();
                    const double rp = (nx * cx + ny * cy + nz * cz - Rb[e]) - s;
                    const double is = row_rcp(s);
                    const double ds = (nx * dx + ny * dy + nz * dzz) + rp;
                    const double t = ds * is;
                    rmax = fmax(rmax, (l > 0.0) ? fmax(-t, 1.0 + t) : 1.0);
                    const double pa = -ds * l * (1.0 + t);  // ds_a * dl_a
                    sB += pa;
                    const double t2 = (-pa - l * rp) * is;  // -ds dl/s - w rp
                    b10 += is * nx; b11 += is * ny;
                    b20 += t2 * nx; b21 += t2 * ny;
                    if (DIM == 3) {
                        b12 += is * nz;
                        b22 += t2 * nz;
                    }
                    }
                wave_ordered([&](bool turn) {
                    if constexpr (W > 1) {
#pragma unroll
                        for (int u = 0; u < NS2; u++) {
                            row_scatter(XB1, u, sc1[u], turn && t_ix[u] >= 0);
                            row_scatter(XB2, u, sc2[u], turn && t_ix[u] >= 0);
                        }
                    }
                    if (ll && turn) {
                        atomicAdd(&XB1[lx], b10); atomicAdd(&XB1[P + lx], b11);
                        atomicAdd(&XB2[lx], b20); atomicAdd(&XB2[P + lx], b21);
                        if (DIM == 3) {
                            atomicAdd(&XB1[2 * P + lx], b12);
                            atomicAdd(&XB2[2 * P + lx], b22);
                        }
                    }
                });
            }
            block_max1_sum1(rmax, sB);
            const double a_aff = fast_rcp(rmax);
            // sum(s dl_a + lam ds_a) == -sum(s lam) by construction of the affine direction
            const double mu_aff = ((1.0 - a_aff) * sum_sl + a_aff * a_aff * sB) * inv_m;
            double sigma = fmax(mu_aff, 0.0) / mu;
            sigma = sigma * sigma * sigma;
            // The centring target is never set below a fraction of what the gap target needs, mu_goal = tol (1 + |obj|) / m.  Mehrotra's
            // sigma = (mu_aff / mu)^3 reaches 1e-6 .. 1e-12 in the last iterations; rows whose linearisation is exact then land AT
            // sigma mu, two or more decades below the average the second-order terms leave (mu_new ~ 0.05 mu), and their lambda / s
            // is what amplifies the rounding of the next direction: the step of a degenerate instance (|dz| ~ sqrt(mu)) injects
            // ~ c eps max(lambda / s) |dz| of stationarity residual, which GROWS as mu falls.  Round 4, per-iteration traces of
            // BASELINE configs[3] (tools/floor_probe.py): 11 of its 1024 instances went from a stationarity of 1e-10 at mu = 1e-11 to
            // 2e-8 .. 8e-8 by the time the gap met its target and then lost a pivot; with the floor none does (floor 0.1 .. 0.5:
            // zero flagged instances, mean iterations 4.609 -> 4.600 at configs[3], unchanged at configs[1] and the configs[4] shape
            // for <= 0.3; 0.5 costs configs[1] an iteration on one instance).  Aiming below the target buys nothing anyway.
            const double smu = fmax(sigma * mu, LSCQP_SMU_FLOOR * tol * (1.0 + obj_abs) * inv_m);
            LSCQP_BLOCK_SYNC();
            LSCQP_T(6);
            LSCQP_STOP(7)
            // ============ corrector solve ==================================================================
            double gb;
            {
                LSCQP_Z_ROLES();
                const double* x1 = &XB1[gbase];
                const double* x2 = &XB2[gbase];
                const double* y1 = &XB1[gnext];
                const double* y2 = &XB2[gnext];
                LSCQP_LANE_WEIGHTS(ZR);
                gb = e0 * (smu * x1[3] + x2[3]) + e1 * (smu * x1[4] + x2[4]) + e2 * (smu * x1[5] + x2[5]) +
                     tb0 * (smu * y1[0] + y2[0]) + tb1 * (smu * y1[1] + y2[1]) + tb2 * (smu * y1[2] + y2[2]);
                gb = zl ? gb : 0.0;
#ifndef LSCQP_SOLVE_FROM_LDS
                // (W = 1: lanes without a row parked zeros in the shared dummy row, nothing to mask; with more wavefronts the
                // dummy row holds whatever the last lane parked)
                {
                    const FT* const prow = park_ptr(lvz_);
                    const bool has_row = C::ND ? ((lvz_ >> 6) < 2 && (lvz_ & 63) < ((lvz_ >> 6) == 0 ? C::NR0 : C::NR1)) : (W == 1 || zl);
#pragma unroll
                    for (int cidx = 0; cidx < C::NAR; cidx++) {
                        const FT v = prow[cidx];
                        A[cidx] = has_row ? v : (FT)0;
                    }
                }
#endif
            }
            const double dzc = solve(-gcost + gb);
            {  // the scratch row must be all-zero outside the assembly pattern again
                LSCQP_PHASE_LANE(lvp_);
                FT* const hrow = &Hs[lane_slot(lvp_) * LDH];
#pragma unroll
                for (int cidx = 0; cidx < NZ; cidx++) hrow[cidx] = (FT)0;
            }
#ifdef LSCQP_TRACE
            {
                const double dzm = block_max(zl ? fabs(dzc) : 0.0);
                LSCQP_TR(10, dzm);
            }
#endif
            if (zl) dz_[zi0] = dzc;  // expandT(dca_) finished reading dz_ before
            LSCQP_BLOCK_SYNC();
            expandT(dz_, dc_, false);
            LSCQP_BLOCK_SYNC();
            LSCQP_T(7);
            LSCQP_STOP(8)
            // ============ pass 3: step length ==============================================================
            // ds = G dc + rp,  dl = (sigma mu - ds_a dl_a)/s - lam - w ds;  ratios -ds/s and -dl/lam
            rmax = 0.0;
            double sdl = 0, sdd = 0;  // sum(s dl + lam ds), sum(ds dl): mu along the step is a quadratic in alpha
            double t_ds[2 * NS2], t_dl[2 * NS2], r_ds[NSLOT], r_dl[NSLOT];  // live only until the update below
#pragma unroll
            for (int u = 0; u < NS2; u++) {
                const bool on = t_ix[u] >= 0;
                const double tsl = t_sl[u].get(), tsh = t_sh[u].get(), tll = t_ll[u].get(), tlh = t_lh[u].get();
                const double tlo = t_lo[u].get(), thi = t_hi[u].get();
                const double y = row_val(c_, u), dya = row_val(dca_, u), dyc = row_val(dc_, u);
                const double rpl = (y - tlo) - tsl, rph = (thi - y) - tsh;
                const double isl = row_rcp(tsl), ish = row_rcp(tsh);
                const double tl = (dya + rpl) * isl, th = (rph - dya) * ish;
                const double pl = -(dya + rpl) * tll * (1.0 + tl), ph = -(rph - dya) * tlh * (1.0 + th);
                const double dsl = dyc + rpl, dsh = rph - dyc;
                const double dll = (smu - pl) * isl - tll - tll * isl * dsl;
                const double dlh = (smu - ph) * ish - tlh - tlh * ish * dsh;
                // rows that do not exist have lambda == 0: give them a harmless divisor
                const double ill = row_rcp(on ? tll : 1.0), ilh = row_rcp(on ? tlh : 1.0);
                const double rr = fmax(fmax(-dsl * isl, -dsh * ish), fmax(-dll * ill, -dlh * ilh));
                rmax = fmax(rmax, on ? rr : 0.0);
                t_ds[2 * u] = on ? dsl : 0.0; t_ds[2 * u + 1] = on ? dsh : 0.0;
                t_dl[2 * u] = on ? dll : 0.0; t_dl[2 * u + 1] = on ? dlh : 0.0;
                sdl += tsl * t_dl[2 * u] + tll * t_ds[2 * u] + tsh * t_dl[2 * u + 1] + tlh * t_ds[2 * u + 1];
                sdd += t_ds[2 * u] * t_dl[2 * u] + t_ds[2 * u + 1] * t_dl[2 * u + 1];
            }
            {
                LSCQP_L_ROLES();
                const double cx = c_[lx], cy = c_[P + lx], cz = (DIM == 3) ? c_[2 * P + lx] : 0.0;
                const double ax = dca_[lx], ay = dca_[P + lx], az = (DIM == 3) ? dca_[2 * P + lx] : 0.0;
                const double dx = dc_[lx], dy = dc_[P + lx], dzz = (DIM == 3) ? dc_[2 * P + lx] : 0.0;
#pragma unroll
                for (int u = 0; u < NSLOT; u++) {
                    const int o = lg + G * u;
                    const int e = (ll && o < n_obs) ? (o * CP + lcp) : NROW;
                    const double nx = Rnx[e], ny = Rny[e], nz = Rnz[e], s = r_s[u].get(), l = r_l[u].get();
                    const double rp = (nx * cx + ny * cy + nz * cz - Rb[e]) - s;
                    const double is = row_rcp(s);
                    const double dsa = (nx * ax + ny * ay + nz * az) + rp;
                    const double pa = -dsa * l * (1.0 + dsa * is);
                    const double ds = (nx * dx + ny * dy + nz * dzz) + rp;
                    const bool act = l > 0.0;
                    const double dl = act ? ((smu - pa) * is - l - l * is * ds) : 0.0;
                    const double il = row_rcp(act ? l : 1.0);
                    rmax = fmax(rmax, fmax(-ds * is, -dl * il));
                    r_ds[u] = act ? ds : 0.0;
                    r_dl[u] = dl;
                    sdl += s * r_dl[u] + l * r_ds[u];
                    sdd += r_ds[u] * r_dl[u];
                    }
            }
            block_sum2_max1(sdl, sdd, rmax);
            // alpha = min(1, tau / rmax).  Fraction to the boundary tau = 0.9995, which caps the reduction of mu at 2000x
            // per iteration; once the predictor announces an almost full step (sigma < 1e-4) tau follows mu,
            // tau = max(0.9995, 1 - mu), and the final phase converges quadratically: 5.05 -> 4.14 mean iterations on the
            // forest workload (tools/proto_pdip.py).  Ungated it costs the ill-conditioned M = 10, 3-D class iterations.
            const double tau = (sigma < 1e-4) ? fmax(0.9995, 1.0 - mu) : 0.9995;
            const double irmax = fast_rcp(rmax);
            const double alpha_std = (rmax > 0.9995) ? 0.9995 * irmax : 1.0;
            double alpha = (rmax > tau) ? tau * irmax : 1.0;
            // Centrality safeguard: no complementarity product may fall below GAMMA * mu(alpha).  Without it Mehrotra's
            // heuristic occasionally drives single products to ~1e-4 mu; the next directions are then blocked at
            // alpha ~ 0.07 and mu cycles around 1e-9 forever (seen at M = 10, dim 3, 40 neighbours; tools/proto_pdip.py).
            // Costs one extra block reduction per iteration; the loop is uniform over the QP's lanes.
            // The step is applied to the row state inside the same loop (trial 0 writes s + alpha ds; a rejected trial is
            // corrected by the difference of the step lengths), so the state is read and written once per iteration
            // instead of once for the test and once more for the update.
            double applied = 0.0;
            for (int bt = 0;; bt++) {
                const double mu_a = (sum_sl + alpha * (sdl + alpha * sdd)) * inv_m;
                const double delta = alpha - applied;
                applied = alpha;
                double pmin = 1e300;
#pragma unroll
                for (int u = 0; u < NS2; u++) {
                    const bool on = t_ix[u] >= 0;
                    const double nsl = fma(delta, t_ds[2 * u], t_sl[u].get()), nll = fma(delta, t_dl[2 * u], t_ll[u].get());
                    const double nsh = fma(delta, t_ds[2 * u + 1], t_sh[u].get()), nlh = fma(delta, t_dl[2 * u + 1], t_lh[u].get());
                    t_sl[u].set(nsl);
                    t_ll[u].set(nll);
                    t_sh[u].set(nsh);
                    t_lh[u].set(nlh);
                    pmin = fmin(pmin, on ? fmin(nsl * nll, nsh * nlh) : 1e300);
                }
#pragma unroll
                for (int u = 0; u < NSLOT; u++) {
                    const double ns = fma(delta, r_ds[u], r_s[u].get()), nl = fma(delta, r_dl[u], r_l[u].get());
                    r_s[u].set(ns);
                    r_l[u].set(nl);
                    pmin = fmin(pmin, (nl > 0.0) ? ns * nl : 1e300);
                }
                pmin = -block_max(-pmin);
                if (pmin >= LSCQP_CENTRALITY_GAMMA * mu_a || bt == 9) break;
                // a blocked step taken to within mu of the boundary leaves the blocking product at ~mu^2: first retreat
                // to the standard fraction, then shorten
                alpha = (bt == 0 && alpha_std < alpha) ? alpha_std : 0.7 * alpha;
            }
            LSCQP_T(8);
            LSCQP_STOP(9)
            // ============ update of z and the control points =================================================
            if (it == 0) alpha_first = (float)alpha;
            LSCQP_TR(4, alpha);
            LSCQP_TR(5, sigma);
            if (zl) z_[zi0] += alpha * dzc;
            LSCQP_BLOCK_SYNC();
            // c = c_fixed + T z, recomputed from z so the eliminated equalities hold to rounding every iteration
            expandT(z_, c_, true);
            LSCQP_BLOCK_SYNC();
            if (!(alpha > 1e-12) || !(mu == mu)) {  // stalled or NaN (wave-uniform)
                LSCQP_TR(6, 3.0);
                status = (near_cnt > 0 || floor_cnt > 0) ? LSCQP_STATUS_OPTIMAL : LSCQP_STATUS_NUMERIC;
                restore = status == LSCQP_STATUS_OPTIMAL;
                break;
            }
            LSCQP_MARK(14);
            LSCQP_T(9);
            LSCQP_STOP(10)
        }
    LSCQP_T(10);
    if (status == LSCQP_STATUS_ITER_LIMIT && (near_cnt > 0 || floor_cnt > 0)) {
        status = LSCQP_STATUS_OPTIMAL;
        restore = true;
    }
    if (status == LSCQP_STATUS_ITER_LIMIT && res_p > 1e-6) status = LSCQP_STATUS_INFEASIBLE;
    if (status == LSCQP_STATUS_NUMERIC && res_p > 1e-6) status = LSCQP_STATUS_INFEASIBLE;
    // Fallback acceptance (breakdown, stall or iteration limit after a point had met the primal and gap tests with its
    // stationarity at the rounding floor, <= 1e-6 relative): the result IS that remembered point, and lscqp_info says so.
    if (restore) {  // uniform over the QP's lanes
        if (lane < NZ) z_[lane] = zs_[lane];
        LSCQP_BLOCK_SYNC();
        expandT(z_, c_, true);
        LSCQP_BLOCK_SYNC();
        res_p = snap_p;
        res_d = snap_d;
        res_gap = snap_gap;
        // The flag marks the stated deviation: a point returned with its scaled stationarity ABOVE 1e-8 (or its gap above the target).
        // A remembered point that meets the three strict tests (1e-9 m, 1e-8, tol) lacks only the second confirmation of
        // LSCQP_NEAR_CONFIRM -- whose purpose, letting the iteration polish on, is moot once the iteration has ended -- and is an
        // ordinary OPTIMAL result.
        flags |= LSCQP_INFO_REMEMBERED;
        if (!(snap_d <= 1e-8 && snap_gap <= tol && snap_p <= 1e-9)) {
            // (round 5: the mixed-precision instances no longer accept at the floor -- the fp64 second pass of the call exists for exactly
            // such instances and re-solves them; BASELINE configs[4]: 56 floor acceptances of 4096 -> 0)
            if constexpr (MIXED) status = LSCQP_STATUS_NUMERIC;
            else flags |= LSCQP_INFO_FLOOR_ACCEPTED;
        }
    }
    if (recentred || net_done) flags |= LSCQP_INFO_RECENTRED;

    // ---- epilogue: objective, control points back in the world frame ---------------------------------------
    const double obj = objective(true, lane);
    for (int e = lane; e < NX; e += T) {
        const int k = e / P;
        x_out[q * NX + e] = c_[e] + org_[k];
    }
    if (lane == 0) {
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

// The kernel: one workgroup of W wavefronts per instance AT A TIME.  grid = n and one instance per workgroup when the launch fits the
// chip at once; otherwise, in the PERSIST instances (cls.queue set), grid = the workgroups the chip holds and each of them keeps taking
// instances from the queue until it is empty -- list scheduling by construction.  Measured why (tools/lpt_probe.py, 1024 x M10 x 40 on
// one MI355X, one workgroup per CU): with grid = n the launch lasted 1.07 - 1.21 ms in the given and in random orders -- MORE than list
// scheduling's worst case, sum / 256 + longest = 0.57 + 0.39 ms: the hardware does not hand a free CU the next workgroup of the grid
// in general -- and 0.83 ms with the instances sorted longest first.
// PERSIST is a per-instance choice (lscqp_inst.hip): the loop keeps every kernel argument live across the whole body for the next
// instance (ten pointers and the class, most of them dead after the prologue of the one-instance form), which pushes the
// register-tight one-wavefront instances into scratch (<5,3,true,10,1>: 0 -> 176 B per lane; reading the arguments back from the kernarg
// segment per instance made it 632 B) -- those keep one instance per workgroup and take only the ORDER of the launch.
template <int M, int DIM, bool ES, int NSLOT, int W = 1, class FT = double, bool PERSIST = false>
__global__ __launch_bounds__(64 * W) LSCQP_KERNEL_ATTR void lscqp_pdip_kernel(DevClass cls, int64_t n, const lscqp_header* __restrict__ hdr,
                                                        const lscqp_row* __restrict__ rows,
                                                        const uint64_t* __restrict__ row_offsets,
                                                        const lscqp_box* __restrict__ sfc, const double* __restrict__ x_init,
                                                        double* __restrict__ x_out,
                                                        double* __restrict__ obj_out, int32_t* __restrict__ status_out,
                                                        lscqp_info* __restrict__ info_out) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    if constexpr (!PERSIST) {
        const int64_t k = blockIdx.x;
        if (k >= n) return;
        const int64_t q = cls.order ? (int64_t)cls.order[k] : k;
        lscqp_pdip_one<M, DIM, ES, NSLOT, W, FT>(cls, q, smem, hdr, rows, row_offsets, sfc, x_init, x_out, obj_out, status_out, info_out);
    } else {
        int64_t k = blockIdx.x;
        int64_t sbase = (int64_t)blockIdx.x - (int64_t)gridDim.x * 64;  // scan mode: first instance of the current group of 64
        unsigned long long smask = 0;                                    // ... and which of the group are still to be solved
#pragma unroll 1
        for (;;) {
            int64_t q;
            if (cls.scan) {  // (uniform: a kernel argument)
                while (smask == 0) {
                    sbase += (int64_t)gridDim.x * 64;
                    if (sbase >= n) return;
                    const int64_t qi = sbase + (int64_t)(threadIdx.x & 63) * gridDim.x;  // (every wavefront of the workgroup loads the same 64 statuses)
                    const int st = qi < n ? status_out[qi] : LSCQP_STATUS_OPTIMAL;
                    // (right behind the phase: what it marked; a later pass: whatever is neither finished nor refused -- lscqp_pdip_one looks again)
                    smask = __builtin_amdgcn_ballot_w64(cls.repair == 3 ? st == LSCQP_STATUS_ITER_LIMIT
                                                                        : (st != LSCQP_STATUS_OPTIMAL && st != LSCQP_STATUS_CAPACITY));
                }
                const int l = __builtin_ctzll(smask);
                smask &= smask - 1;
                q = sbase + (int64_t)l * gridDim.x;
            } else {
                if (k >= n) return;
                q = cls.order ? (int64_t)cls.order[k] : k;
            }
            lscqp_pdip_one<M, DIM, ES, NSLOT, W, FT>(cls, q, smem, hdr, rows, row_offsets, sfc, x_init, x_out, obj_out, status_out, info_out);
            if (cls.scan) {
                __syncthreads();  // (the next instance reuses the LDS of this one)
                continue;
            }
            if (!cls.queue) return;  // (uniform: a kernel argument)
            // next instance: one atomic per workgroup, handed to its lanes through LDS (the instance just finished no longer needs it)
            __syncthreads();
            if (threadIdx.x == 0) *reinterpret_cast<volatile long long*>(smem) = (long long)gridDim.x + (long long)atomicAdd(cls.queue, 1);
            if constexpr (W == 1) {
                asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)" ::: "memory");
            } else {
                __syncthreads();
            }
            k = *reinterpret_cast<volatile long long*>(smem);
            if constexpr (W == 1) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else {
                __syncthreads();
            }
        }
    }
}

}  // namespace lscqp
