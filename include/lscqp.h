/*
 * lscqp.h — C ABI of the MI355X-native batched trajectory-QP solver.
 *
 * This is the drop-in boundary for the ONE hot path of qwerty35/lsc_dr_planner:
 * the per-agent, per-replan trajectory QP that the reference builds in
 * TrajOptimizer::populatebyrow (src/traj_optimizer.cpp:216-514) and solves with
 * CPLEX in TrajOptimizer::solve (src/traj_optimizer.cpp:18-156).
 *
 * The reference has no FFI of its own (everything is one statically linked C++
 * executable), so the boundary is a C++ class surface.  The C++ shim in
 * lsc_dr_planner_amd/shim/ keeps that surface (TrajOptimizer /
 * CollisionConstraints / Trajectory, namespace DynamicPlanning) and calls the
 * entry points below; nothing but plain pointers and sizes crosses this ABI.
 *
 * Each entry point cites the reference interface it replaces.
 *
 * Conventions
 *   P  = M*(n+1) control points per axis, n = 5, phi = 3 (the only values the
 *        reference supports, src/traj_optimizer.cpp:184-201).
 *   nv = dim*P decision variables, index x[k*P + m*(n+1) + i]  (axis-major, then
 *        segment, then control point — src/traj_optimizer.cpp:55-57,220-221,241).
 *   All floating-point payloads are fp64.  Inputs that are float32 in the
 *   reference (octomap::point3d) are widened by the caller.
 */
#ifndef LSCQP_H
#define LSCQP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LSCQP_VERSION_MAJOR 0
#define LSCQP_VERSION_MINOR 8

/* ---- return codes of the API calls themselves (misuse / runtime errors) ---- */
enum {
    LSCQP_OK = 0,
    LSCQP_ERR_INVALID_ARGUMENT = 1, /* mirrors std::invalid_argument in the reference ctor
                                       (src/traj_optimizer.cpp:200) and populatebyrow (:249) */
    LSCQP_ERR_UNSUPPORTED = 2,      /* class shape has no compiled kernel instance */
    LSCQP_ERR_NO_DEVICE = 3,        /* no HIP device / HIP runtime error: the product path has
                                       no CPU fallback and fails loudly */
    LSCQP_ERR_HIP = 4
};

/* ---- per-instance solve status (status_out[]) ----
 * The reference signals every non-success as `throw PlanningReport::QPFAILED`
 * (src/traj_optimizer.cpp:143,152); the shim maps status != OPTIMAL to that throw. */
enum {
    LSCQP_STATUS_OPTIMAL = 0,
    LSCQP_STATUS_INFEASIBLE = 1, /* CPLEX Infeasible / InfeasibleOrUnbounded, :105-106 */
    LSCQP_STATUS_ITER_LIMIT = 2,
    LSCQP_STATUS_NUMERIC = 3,
    LSCQP_STATUS_CAPACITY = 4    /* hdr.n_obs exceeds the row capacity of the kernel instance the launch selected (n_obs_max too
                                    small, or no compiled instance that large): the instance is REFUSED -- LSC rows are never
                                    dropped silently, the reference gives every obstacle its rows (:399-437) */
};

/* PlannerMode values the QP reads (include/sp_const.hpp:19-26).  Only LSC adds the
 * end-of-horizon stop rows (src/traj_optimizer.cpp:502-511). */
enum {
    LSCQP_PLANNER_DLSC = 0,
    LSCQP_PLANNER_LSC = 1,
    LSCQP_PLANNER_BVC = 2,
    LSCQP_PLANNER_RSFC = 3 /* RECIPROCALRSFC: no end-stop rows, and the z variables of segment 0 are bounded by +-100 instead of the
                              world box (src/traj_optimizer.cpp:255-258).  That planner runs with slack_mode COLLISIONCONSTRAINT:
                              every LSC row then carries a slack in (-inf, 0] that appears in no cost term (:272-283, 423-425), i.e.
                              no LSC row can bind -- callers pass n_obs = 0 (the shim does) */
};

/* Problem class: everything TrajOptimizer caches from Param/Mission at construction
 * (src/traj_optimizer.cpp:4-16) plus the Param fields populatebyrow reads. */
/* Storage format of the packed LSC rows, a property of the class (row_format below).  The reference's LSC record holds its
 * normal and obstacle point as float32 and only d as double (include/collision_constraints.hpp:19-33); LSCQP_ROWS_F32 stores
 * the packed row (nx, ny, nz, b = d + n.p_obs) as four floats -- half the HBM bytes of the batch, SURVEY.md section 8d's byte model
 * for BASELINE configs[4] (10 832 B per QP at M = 5, 20 neighbours; the SFC boxes stay fp64) -- and is widened to fp64 when a kernel stages it: every
 * entry point that reads or writes rows (solve, goal LP, constraint generation) follows the handle's format, the arithmetic
 * stays fp64, and a solve on f32 rows equals bit for bit the solve on f64 rows holding the same float values. */
#define LSCQP_ROWS_F64 0
#define LSCQP_ROWS_F32 1

typedef struct lscqp_class_desc {
    int32_t M;            /* param.M   — number of segments, >= 2 */
    int32_t n;            /* param.n   — must be 5 */
    int32_t phi;          /* param.phi — must be 3 */
    int32_t phi_n;        /* param.phi_n — must be 1 */
    int32_t dim;          /* param.world_dimension, 2 or 3 */
    int32_t planner_mode; /* LSCQP_PLANNER_* */
    int32_t use_sfc;      /* param.world_use_octomap: SFC rows present (:372) */
    int32_t row_format;   /* LSCQP_ROWS_F64 (0, default): lscqp_row, 32 B; LSCQP_ROWS_F32: lscqp_row_f32, 16 B (see below) */
    double dt;                   /* param.dt */
    double control_input_weight; /* param.control_input_weight (:294) */
    double terminal_weight;      /* param.terminal_weight (:304) */
    double communication_range;  /* param.communication_range; <= 0 disables rows (:478) */
    double world_min[3];         /* mission.world_min — variable lower bounds (:252) */
    double world_max[3];         /* mission.world_max — variable upper bounds (:253) */
    /* solver controls (0 selects the default) */
    int32_t max_iter;  /* default 60 */
    int32_t precision; /* LSCQP_PRECISION_* below */
    double tol;        /* relative duality-gap tolerance, 0 = default 1e-10 */
    int32_t warm_start; /* LSCQP_WARM_* below: how an instance that comes with an initial trajectory is centred */
    int32_t active_set; /* LSCQP_ACTIVE_SET_* below: the dual active-set phase in front of the interior-point kernel (0 = default: on) */
} lscqp_class_desc;

/* The dual active-set phase (round 5).  The QP's Hessian is a constant of the class -- jerk cost and terminal pull never depend on the
 * neighbours -- and a plan's optimum holds few of its rows (none at all for 61 of the 64 QPs of BASELINE configs[1]): a first launch
 * over the batch starts every instance at its unconstrained minimiser (three vectors of a per-class table, no factorisation) and adds
 * the violated rows one at a time (Goldfarb-Idnani; csrc/lscqp_das.hip).  What it finishes is marked LSCQP_INFO_ACTIVE_SET and meets
 * the same bar as an interior-point result (1e-9 m on every row, 1e-9 scaled stationarity, multipliers >= 0, exact complementarity);
 * what it does not finish inside its budget (active rows, steps) -- or cannot judge: dependent active rows, capacity, row systems whose
 * emptiness it cannot prove -- is solved by the interior-point kernel behind it in the same call, on the same stream, exactly as without
 * the phase.  Round 6: a row system the phase PROVES empty -- an empty interval, or a violated row (by more than 1e-6 m, the interior-point
 * kernel's own bar) whose normal lies in the span of the active rows' with no multiplier to give way: a Farkas certificate -- is returned
 * LSCQP_STATUS_INFEASIBLE by the phase itself (LSCQP_INFO_ACTIVE_SET set, res_primal = the violation) and the kernel behind skips it.
 *   LSCQP_ACTIVE_SET_DEFAULT  on
 *   LSCQP_ACTIVE_SET_OFF      the interior-point kernel alone (rounds 1-4; also: LSCQP_ACTIVE_SET=0 in the environment when the handle is
 *                             CREATED -- the library reads its environment in lscqp_create and nowhere else)
 *   LSCQP_ACTIVE_SET_ONLY     the phase alone: instances it leaves are returned LSCQP_STATUS_ITER_LIMIT (development / tests) */
/* Why the phase left an instance (LSCQP_ACTIVE_SET_ONLY: lscqp_info.res_dual of an instance returned LSCQP_STATUS_ITER_LIMIT holds the code,
 * lscqp_info.gap the steps taken; with the interior-point kernel behind the phase the record is that kernel's). */
#define LSCQP_DAS_WHY_CAPACITY 1       /* more obstacles than the launch's kernels hold: LSCQP_STATUS_CAPACITY is the kernel's to give */
#define LSCQP_DAS_WHY_EMPTY_INTERVAL 2 /* lower bound above upper bound on one control point (an empty corridor / world box) */
#define LSCQP_DAS_WHY_ROWS 3           /* more active rows than the launch's budget */
#define LSCQP_DAS_WHY_STEPS 4          /* more steps than the launch's budget */
#define LSCQP_DAS_WHY_NO_STEP 5        /* no step exists for a violated row: its normal lies in the span of the active rows' and no multiplier
                                          blocks -- the row system has no point (a Farkas certificate up to rounding) */
#define LSCQP_DAS_WHY_PIVOT 6          /* a leaving row's rotation lost its pivot (dependent active rows) */
#define LSCQP_DAS_WHY_VERIFICATION 7   /* stationarity above the bar after the polish */
#define LSCQP_DAS_WHY_MULTIPLIER 8     /* a polished multiplier below zero by more than rounding */
#define LSCQP_ACTIVE_SET_DEFAULT 0
#define LSCQP_ACTIVE_SET_OFF 1
#define LSCQP_ACTIVE_SET_ONLY 2

/* Centring of warm-started instances (x_init given).  Every complementarity product starts at mu0 with the slacks floored at s0.
 *   LSCQP_WARM_DEFAULT  (mu0, s0) = (1e-3, 3 cm): safe whatever the quality of the initial trajectory; the shortest TAIL, which is
 *                       what bounds a small batch (a launch lasts as long as its slowest QP).
 *   LSCQP_WARM_TIGHT    (1e-7, 3 mm): presumes the initial trajectory is close to the optimum with the right rows near their bounds --
 *                       the shifted previous plan of a replanning loop usually is.  An instance whose first step comes out shorter
 *                       than 0.6 returns to the default centring after that iteration.  Fewer iterations on average where the start
 *                       is good (4096 x M5 x 20: 3.1 -> 2.2, +6..15 % QP/s; agents holding position: 7.1 -> 4.8) with a longer tail
 *                       everywhere -- 64 x M5: 0.106 -> 0.129 ms, the M = 10 shapes 10..50 % slower, single instances run to the
 *                       iteration limit and are re-solved by the second pass.  A throughput option, not a default. */
#define LSCQP_WARM_DEFAULT 0
#define LSCQP_WARM_TIGHT 1

/* Arithmetic of the interior-point iteration (lscqp_class_desc.precision).
 *   LSCQP_PRECISION_F64    everything fp64 (default; what the reference's CPLEX call computes in, src/traj_optimizer.cpp:66-70).
 *   LSCQP_PRECISION_MIXED  BASELINE configs[4], "fp32 PDIP with fp64 residual check": the reduced KKT matrix is rounded to
 *                          float32 and factorised / substituted in float32 (half the registers and broadcasts of the dominant
 *                          phase); control points, slacks, multipliers, residuals and every stopping test stay fp64, so an
 *                          accepted point meets exactly the same KKT bar as in fp64 mode -- the float32 directions cost
 *                          iterations (+0.4 on the forest class), never accuracy.  Instances on which the float32
 *                          factorisation breaks down (ill-conditioned final iterations; rare on the forest class, common in
 *                          dense mazes) are re-solved by the fp64 kernel in a second pass over the batch inside the same call
 *                          (no host round trip; lscqp_info.flags & LSCQP_INFO_REPAIRED).  The result format (fp64 control
 *                          points, float32 truncation by the shim as in :71-83) is unchanged.
 *                          Available for the throughput shapes (one wavefront per QP); lscqp_create returns
 *                          LSCQP_ERR_UNSUPPORTED for a shape without a compiled mixed instance. */
#define LSCQP_PRECISION_F64 0
#define LSCQP_PRECISION_MIXED 1

/* A packed row in the LSCQP_ROWS_F32 format; pointers declared `lscqp_row*` below then point at arrays of this type, and row
 * offsets stay in units of rows. */
typedef struct lscqp_row_f32 {
    float nx, ny, nz, b;
} lscqp_row_f32;

/* Per-QP header: the fields of Agent (include/sp_const.hpp:146-160) that populatebyrow reads.
 * Exactly 256 bytes. */
typedef struct lscqp_header {
    double p0[3];             /* agent.current_state.position      (:321) */
    double v0[3];             /* agent.current_state.velocity      (:332) */
    double a0[3];             /* agent.current_state.acceleration  (:338) */
    double goal[3];           /* agent.current_goal_point          (:305-313) */
    double next_waypoint[3];  /* agent.next_waypoint               (:494-497) */
    double vmax[3];           /* agent.max_vel[k]                  (:450) */
    double amax[3];           /* agent.max_acc[k]                  (:466) */
    double radius;            /* agent.radius                      (:484) */
    double nominal_velocity;  /* agent.nominal_velocity            (:533) */
    int32_t n_obs;            /* constraints.getObsSize()          (:219) */
    int32_t terminal_segments;/* getTerminalSegments_old(agent) (:530-538) computed by the caller with the
                                 reference's float32 semantics; <= 0: the solver computes it in fp64 */
    uint32_t reserved[2];
    double pad[7];
} lscqp_header;

/* One packed LSC half-space: the constraint  nx*cx + ny*cy + nz*cz >= b  on one control point.
 * From the reference LSC{obs_control_point p, normal_vector nrm, d}
 * (include/collision_constraints.hpp:19-33; row built at src/traj_optimizer.cpp:413-429):
 *     nrm.(c - p) - d >= 0   <=>   nrm.c >= d + nrm.p =: b.
 * Rows with ||nrm|| < 1e-5 are skipped as in the reference (:409-411).  32 bytes. */
typedef struct lscqp_row {
    double nx, ny, nz, b;
} lscqp_row;

/* One SFC box per segment (Box{box_min, box_max}, include/collision_constraints.hpp:39-46;
 * faces via Box::convertToLSCs, src/collision_constraints.cpp:37-59). 48 bytes. */
typedef struct lscqp_box {
    double bmin[3];
    double bmax[3];
} lscqp_box;

/* Optional per-instance solver diagnostics. 32 bytes. */
#define LSCQP_INFO_FLOOR_ACCEPTED 1 /* OPTIMAL by the fallback rule with the one stated deviation: the iteration broke down / stalled /
                                       hit the limit after a point had met the primal (1e-9 m) and gap tests with its stationarity
                                       residual at the rounding floor (<= 1e-6 relative instead of 1e-8); the returned point IS
                                       that point (LSCQP_INFO_REMEMBERED is set too).  Round 4: no instance of the BASELINE
                                       workloads carries it any more (tests/test_floor_audit.py) */
#define LSCQP_INFO_REPAIRED 2       /* solved by the second pass of the call (fp64 after a mixed-precision breakdown, or the
                                       default start after a warm-started failure); iterations counts both passes */
#define LSCQP_INFO_RECENTRED 4      /* a jammed warm start was re-centred once inside the kernel */
#define LSCQP_INFO_REMEMBERED 8     /* the iteration ended by a breakdown / stall / the limit and the result is the BEST point it had
                                       remembered; without LSCQP_INFO_FLOOR_ACCEPTED that point meets the strict tests (1e-9 m,
                                       1e-8, tol) -- it lacks only the second confirmation an iteration that goes on would give */
#define LSCQP_INFO_SHIFTED 16       /* a factorisation lost a pivot to rounding and was repeated with a diagonal shift of
                                       1e-14 (1e-12) max|K|; residuals and stopping tests are exact whatever the direction */
#define LSCQP_INFO_RESCUED 32       /* solved by the RESCUE pass: the instance had run into the iteration limit (a limit cycle of the
                                       predictor-corrector iteration: one cold DLSC instance in ~30 000 of the stress sweeps) or broken
                                       down numerically, and was re-solved on the run-time-shaped kernel with the corrector's
                                       second-order term weighted by the affine step length where that step is blocked (< 0.3) in
                                       the feasible phase.  Run by the host-pointer entries for batches that still hold such an
                                       instance after their other passes, and by retry == 2 of the device entries. */
#define LSCQP_INFO_ACTIVE_SET 64    /* solved by the DUAL ACTIVE SET phase (round 5, csrc/lscqp_das.hip) that runs in front of the interior-point
                                       kernel: Goldfarb-Idnani from the unconstrained minimiser on the class's tabulated inverse.
                                       `iterations` then counts its steps (rows added + rows dropped; 0 = the unconstrained minimiser
                                       violates no row), res_primal is the largest violation over EVERY row at the returned point
                                       (<= 1e-9 m), res_dual the reduced stationarity residual on the interior-point kernel's scale
                                       (verified <= 1e-9), gap is 0: complementarity is exact.  See lscqp_class_desc.active_set. */
typedef struct lscqp_info {
    int32_t iterations;
    int32_t flags;     /* LSCQP_INFO_* */
    double res_primal; /* max inequality violation, metres */
    double res_dual;   /* inf-norm of the reduced stationarity residual, scaled */
    double gap;        /* complementarity: mean s*lambda, scaled */
} lscqp_info;

typedef struct lscqp_solver* lscqp_handle;

/* Replaces TrajOptimizer::TrajOptimizer (src/traj_optimizer.cpp:4-16): validates (n,phi)==(5,3) like
 * buildAeqBase (:198-201), precomputes Q_base (:163-178) and the continuity structure (:180-214),
 * uploads class constants to the device. */
int lscqp_create(const lscqp_class_desc* desc, lscqp_handle* out);

/* Replaces TrajOptimizer::updateParam (src/traj_optimizer.cpp:158-160): re-derive class constants.  Launches already enqueued (or captured
 * in a graph) keep the constants they were enqueued with: the class travels by value in kernel arguments, and the active-set tables of a
 * device are immutable once a launch may have seen them -- an update that changes them (dt, weights, M, planner mode, communication range
 * on/off) gives every device that holds a copy a FRESH buffer and retires the old one; the reference's own updates (slack_mode flips,
 * src/traj_planner.cpp:155,160,187,214) change none of these and touch nothing on the device.  Not a device synchronisation point. */
int lscqp_update(lscqp_handle h, const lscqp_class_desc* desc);

/* The class's active-set tables on the CURRENT device, now.  lscqp_create loads them on the device that is current then; any other device's
 * first solve loads them lazily -- which a launch inside a stream capture cannot do (hipMalloc / hipMemcpy are not allowed there): such a
 * launch returns LSCQP_ERR_HIP and names this function, instead of silently running without the phase.  lscqp_comm_prepare does it for every
 * device of a communicator. */
int lscqp_prepare_device(lscqp_handle h);

int lscqp_destroy(lscqp_handle h);

/* nv = dim*M*(n+1): number of doubles per instance in x_out. */
int lscqp_num_variables(lscqp_handle h);
/* M, whether the class carries SFC rows, and the bytes of one packed row in the handle's row_format (32 or 16). */
int lscqp_num_segments(lscqp_handle h);
/* The largest number of obstacles per agent any compiled kernel instance of the handle's shape and precision holds: the n_obs_max
 * beyond which lscqp_solve_batch_device returns LSCQP_ERR_UNSUPPORTED.  A caller that must not drop neighbours (the reference
 * never does, src/multi_sync_simulator.cpp:318-333) sizes its row buffers from lscqp_select_neighbours_device's in-range counts,
 * up to this bound. */
int lscqp_max_obstacles(lscqp_handle h);
int lscqp_uses_sfc(lscqp_handle h);
int lscqp_row_bytes(lscqp_handle h);

/* Replaces TrajOptimizer::solve (src/traj_optimizer.cpp:18-156) for a batch of n independent agents
 * (the sequential loop at src/multi_sync_simulator.cpp:354-362 issues n == 1).
 * HOST pointers; synchronous; copies in, launches the HIP kernel, copies out.
 *   hdr          [n]
 *   rows         packed LSC rows; instance q owns rows[row_offsets[q] .. row_offsets[q] + hdr[q].n_obs*P),
 *                ordered [oi][m][i] exactly like CollisionConstraints::lscs (include/collision_constraints.hpp:173)
 *                (the m==0,i<3 entries are present and ignored, :404-406)
 *   row_offsets  [n+1] in units of rows (uint64)
 *   sfc          [n*M] boxes, or NULL when !use_sfc
 *   x_init       [n*nv] or NULL: the `initial_traj` argument of TrajOptimizer::solve (the shifted previous plan,
 *                src/traj_planner.cpp:399-411) as control points in the reference variable order.  CPLEX ignores it
 *                (src/traj_optimizer.cpp:516-528 is dead code); here it is the primal start of the interior-point
 *                iteration (same optimum, about one iteration fewer in steady state).  NULL: start from hover.
 *   x_out        [n*nv]  raw fp64 control points, reference variable order (no float32 truncation)
 *   obj_out      [n]     objective INCLUDING the constant terminal term, == cplex.getObjValue() (:100)
 *   status_out   [n]     LSCQP_STATUS_*
 *   info_out     [n] or NULL
 * An iteration started from the caller's trajectory can jam against the boundary (a few instances per ten thousand); the
 * kernel re-centres such an instance once.  With x_init given, instances that still end in ITER_LIMIT / NUMERIC are solved
 * once more from the default start before this call returns (lscqp_info.iterations then counts both attempts); the device
 * variant below cannot look at the statuses without synchronising and leaves that to its caller. */
int lscqp_solve_batch(lscqp_handle h, int64_t n, const lscqp_header* hdr, const lscqp_row* rows,
                      const uint64_t* row_offsets, const lscqp_box* sfc, const double* x_init, double* x_out,
                      double* obj_out, int32_t* status_out, lscqp_info* info_out);

/* lscqp_solve_batch with the H2D copy, the kernels and the D2H copy issued on the CALLER's stream (a hipStream_t passed as
 * void*; NULL = a private non-blocking stream of the call); returns after that stream has been synchronised.
 * THREAD SAFETY of the host-pointer entry points (this one, lscqp_solve_batch, lscqp_optimize_goal, lscqp_construct_sfc): each
 * call stages through its own (device buffer, pinned mirror, stream) slot taken from a pool inside the handle, so any number of
 * threads may call them concurrently on the same solver / map handle -- the reference builds a fresh IloEnv per call and is
 * re-entrant in the same sense (src/traj_optimizer.cpp:25-29).  lscqp_update / lscqp_destroy must not race with calls in flight. */
int lscqp_solve_batch_stream(lscqp_handle h, int64_t n, const lscqp_header* hdr, const lscqp_row* rows,
                             const uint64_t* row_offsets, const lscqp_box* sfc, const double* x_init, double* x_out,
                             double* obj_out, int32_t* status_out, lscqp_info* info_out, void* stream);

/* Same, DEVICE pointers, asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream).
 * Inputs must already be resident in HBM; nothing is copied and nothing is synchronised.
 * n_obs_max: upper bound of d_hdr[q].n_obs over the batch; it selects the kernel instance (row slots in registers, LDS
 * staging area).  An instance whose n_obs exceeds the selected kernel's capacity is refused with LSCQP_STATUS_CAPACITY --
 * rows are never dropped -- so pass the true maximum; LSCQP_ERR_UNSUPPORTED if no compiled instance holds n_obs_max.
 * retry != 0: a second pass over the batch on the same stream re-solves, from the default start, the instances that the
 * first pass did not bring to OPTIMAL (jammed / diverged warm starts; what lscqp_solve_batch does for its callers) --
 * no host round trip, ~3 us when there is nothing to repair.  retry == 2: the second pass runs on the compiled instance that
 * eliminates the reduced system in the OTHER order (natural vs nested dissection; M = 10 in 2-D has both), also for batches
 * without a start trajectory -- a factorisation that breaks down in one order usually survives in the other; that instance costs
 * ~35 us to launch even with nothing to repair, so it is not what retry == 1 does.  retry == 2 also ends with the RESCUE pass
 * (LSCQP_INFO_RESCUED: instances still at the iteration limit / a numerical breakdown, on the run-time-shaped kernel with a weighted
 * corrector).  retry == 3: the second pass from the default start AND the rescue pass, without the other-order pass -- two launches that
 * return at once for every instance already OPTIMAL (each tests the status before it fetches anything); what lscqp_plan's chain enqueues
 * in every replan (the reference tries a failed QP again on the spot, src/traj_planner.cpp:763-766).  Other values: LSCQP_ERR_INVALID_ARGUMENT.
 * The host-pointer entries run both by themselves, and only for batches that still hold such an instance after the first call.
 * In LSCQP_PRECISION_MIXED the fp64 second pass always runs. */
int lscqp_solve_batch_device_ex(lscqp_handle h, int64_t n, int32_t n_obs_max, const lscqp_header* d_hdr,
                                const lscqp_row* d_rows, const uint64_t* d_row_offsets, const lscqp_box* d_sfc,
                                const double* d_x_init, double* d_x_out, double* d_obj_out, int32_t* d_status_out,
                                lscqp_info* d_info_out, int32_t retry, void* stream);
/* lscqp_solve_batch_device_ex with retry = 0. */
int lscqp_solve_batch_device(lscqp_handle h, int64_t n, int32_t n_obs_max, const lscqp_header* d_hdr,
                             const lscqp_row* d_rows, const uint64_t* d_row_offsets, const lscqp_box* d_sfc,
                             const double* d_x_init, double* d_x_out, double* d_obj_out, int32_t* d_status_out,
                             lscqp_info* d_info_out, void* stream);

/* Work order of a launch (round 4).  A launch with more instances than the chip holds at once is run by PERSISTENT workgroups that take
 * instance after instance from a queue (list scheduling; nothing to ask for, lscqp_solve_batch_device[_ex] do it by themselves).  The
 * launch then lasts sum / slots + (what is still running when the queue is empty): starting the LONGEST instances first shortens that
 * tail -- 1024 x M10 x 40 on one MI355X: 1.14 ms in the given order with one instance per workgroup, see DESIGN.md section 4 for the
 * figures with the queue and with the order.  The iteration count of the previous replan's solve of the same agent is the hint a
 * planner has (lscqp_plan carries it from replan to replan by itself when its QP launch exceeds lscqp_launch_capacity):
 *   lscqp_order_by_work_device   d_order_out[n] := the instances sorted by d_info_prev[i].iterations, most first, ties in index order
 *   lscqp_solve_batch_device_ordered   lscqp_solve_batch_device_ex with the k-th slot of the launch solving instance d_order[k]
 *                                (a permutation of 0 .. n-1; NULL = identity).  Results land at the instance's own index, bit for bit
 *                                what any other order gives.  The permutation is trusted; a handle created with LSCQP_CHECK_ORDER=1
 *                                in the environment verifies it on the device in every call first (an allocation and a synchronisation:
 *                                a debugging aid) and returns LSCQP_ERR_INVALID_ARGUMENT for a stale or short order buffer. */
/* instances of a launch of n the device works on at once (CUs x workgroups per CU of the kernel instance the launch selects); a launch of
 * more runs in rounds and has a tail -- that is where the order pays.  -1 without a device. */
int64_t lscqp_launch_capacity(lscqp_handle h, int64_t n, int32_t n_obs_max);
/* instances ONE device works on at once in the kernel that carries a solve of this class: the dual active-set phase's resident workgroups
 * when the phase is on AND finishing what it is given; lscqp_launch_capacity (the interior-point instance's) when the phase is off or the
 * handle's last host-pointer call had to run the interior-point passes behind it.  -1 without a device. */
int64_t lscqp_device_fill(lscqp_handle h, int64_t n, int32_t n_obs_max);
int lscqp_order_by_work_device(int64_t n, const lscqp_info* d_info_prev, int32_t* d_order_out, void* stream);
int lscqp_solve_batch_device_ordered(lscqp_handle h, int64_t n, int32_t n_obs_max, const lscqp_header* d_hdr, const lscqp_row* d_rows,
                                     const uint64_t* d_row_offsets, const lscqp_box* d_sfc, const double* d_x_init, double* d_x_out,
                                     double* d_obj_out, int32_t* d_status_out, lscqp_info* d_info_out, int32_t retry,
                                     const int32_t* d_order, void* stream);

/* ---- multi-GPU: the agent batch over the GPUs of one node, from ONE host process ----------------
 *
 * The reference's host is a single process (one ROS node); within a replan step its N QPs are independent
 * (src/multi_sync_simulator.cpp:354-362) and the only exchange is broadcastMsgs (:305-352: every agent receives the others'
 * previous plans).  A communicator owns one private stream, one staging pool and one RCCL communicator per device
 * (ncclCommInitAll; RCCL is bound at run time, a single-GPU host needs none).  Partitioning: contiguous blocks of
 * ceil(N / G) agents in id order; NO collective on the solve path.
 *   lscqp_comm_create        n_devices <= 0: all visible devices; device_ids NULL: 0 .. n_devices-1
 *   lscqp_comm_backend       "rccl <version> (ncclCommInitAll, G devices)" or "none: single device, ..." -- what actually runs
 *   lscqp_comm_devices_for   how many devices a batch of n agents is spread over: clamp(n / min_agents_per_device, 1, G), default
 *                            256 agents per device -- one 64-QP block per device finishes no sooner than 512 QPs on one
 *                            device (0.108 vs 0.136 ms at M = 5), so spreading a small batch buys nothing and only adds the
 *                            exchange; north_star: "only when agent count justifies it".  lscqp_comm_set_min_agents_per_device
 *                            moves the threshold (1 = always use every device).
 *   lscqp_comm_devices_for_class   the rule lscqp_solve_batch_sharded applies (round 5), with the class in hand: clamp(n / fill, 1, G),
 *                            fill = lscqp_device_fill(h, n, n_obs_max) = the instances ONE device works on at once in the first
 *                            kernel of a solve (the dual active-set phase's resident workgroups, or lscqp_launch_capacity without
 *                            the phase, or while the phase keeps leaving work to the slower kernel): a second device is used only
 *                            where one would need a second round.  A threshold set with lscqp_comm_set_min_agents_per_device
 *                            overrides it.  THE rule of the library: lscqp_solve_batch_sharded applies it, and a caller of
 *                            lscqp_solve_batch_sharded_device (who cuts the blocks himself) asks it here; lscqp_comm_devices_for is
 *                            its class-free fallback for callers without a handle.
 *   lscqp_comm_prepare       the class's active-set tables on every device of the communicator, now (a device's first solve loads
 *                            them lazily otherwise -- which a launch inside a stream capture cannot do: it fails, loudly)
 *   lscqp_comm_shard         the block [first, first + count) of device g when n agents are spread over n_used devices
 *   lscqp_solve_batch_sharded        HOST pointers for the whole batch (same arguments as lscqp_solve_batch): every block is staged
 *                            to its device, solved there (with the retry pass) and fetched back, all devices concurrently, each
 *                            on its own stream; results land in the caller's arrays in agent order.  *n_devices_used reports the
 *                            spread.  This is what TrajOptimizer::solveBatch calls when it has a communicator.
 *   lscqp_solve_batch_sharded_device DEVICE-resident blocks: arrays of G per-device pointers, n[g] agents on device g, launched on
 *                            the communicator's streams, asynchronous (lscqp_comm_synchronize waits for all of them)
 *   lscqp_allgather          the device analogue of broadcastMsgs: device g contributes d_send[g][count] doubles (its block of
 *                            solved control points, x_out) and receives everybody's into d_recv[g][G * count]; one grouped
 *                            ncclAllGather over xGMI on the communicator's streams, asynchronous.  Equal counts per device:
 *                            pad the last block.  count * 8 B per device is 46 KB (64 agents, M = 5) to 369 KB (512 agents):
 *                            latency-bound on xGMI, hence the spreading rule above. */
typedef struct lscqp_comm_s* lscqp_comm;
int lscqp_comm_create(int32_t n_devices, const int32_t* device_ids, lscqp_comm* out);
void lscqp_comm_destroy(lscqp_comm c);
int32_t lscqp_comm_size(lscqp_comm c);
int32_t lscqp_comm_device(lscqp_comm c, int32_t g);
void* lscqp_comm_stream(lscqp_comm c, int32_t g); /* hipStream_t of device g */
const char* lscqp_comm_backend(lscqp_comm c);
int lscqp_comm_set_min_agents_per_device(lscqp_comm c, int64_t n);
int32_t lscqp_comm_devices_for(lscqp_comm c, int64_t n);
int32_t lscqp_comm_devices_for_class(lscqp_comm c, lscqp_handle h, int64_t n, int32_t n_obs_max);
int lscqp_comm_prepare(lscqp_comm c, lscqp_handle h);
int lscqp_comm_shard(lscqp_comm c, int64_t n, int32_t n_used, int32_t g, int64_t* first, int64_t* count);
/* the same partition rule without a communicator (no device needed): block g of n agents over n_used devices */
int lscqp_shard_range(int64_t n, int32_t n_used, int32_t g, int64_t* first, int64_t* count);
int lscqp_comm_synchronize(lscqp_comm c);
int lscqp_solve_batch_sharded(lscqp_handle h, lscqp_comm c, int64_t n, const lscqp_header* hdr, const lscqp_row* rows,
                              const uint64_t* row_offsets, const lscqp_box* sfc, const double* x_init, double* x_out, double* obj_out,
                              int32_t* status_out, lscqp_info* info_out, int32_t* n_devices_used);
int lscqp_solve_batch_sharded_device(lscqp_handle h, lscqp_comm c, const int64_t* n, int32_t n_obs_max, const lscqp_header* const* d_hdr,
                                     const lscqp_row* const* d_rows, const uint64_t* const* d_row_offsets, const lscqp_box* const* d_sfc,
                                     const double* const* d_x_init, double* const* d_x_out, double* const* d_obj_out,
                                     int32_t* const* d_status_out, lscqp_info* const* d_info_out, int32_t retry);
int lscqp_allgather(lscqp_comm c, const double* const* d_send, double* const* d_recv, int64_t count);
/* The exchange that follows a sharded replan, as a list of collective operations (no device needed: this is the bookkeeping
 * lscqp_plan_group_step executes over RCCL, exported so that it can be checked for any device count on a host without GPUs).
 * Device g owns agents [first[g], first[g] + count[g]) of n_total; the blocks must be consecutive in device order and cover the
 * mission (LSCQP_ERR_INVALID_ARGUMENT otherwise).  Every device holds a buffer of n_total * per doubles in agent order and has
 * just rewritten its own block.  Equal blocks: ONE in-place all-gather (kind LSCQP_XCHG_ALLGATHER: every device sends `count`
 * doubles from `offset_of_device[g] = first[g] * per`, block g lands at first[g] * per on every device).  A short or empty last
 * block: one broadcast per NON-EMPTY owner (kind LSCQP_XCHG_BROADCAST, root = the owner, `offset` / `count` in doubles), empty
 * owners are skipped by every device alike.  Returns the number of operations in *n_ops (<= n_devices).
 *
 * lscqp_exchange_schedule_padded: the same with `pad_agents` agents of room BEHIND the mission in every device's buffer.  The blocks
 * lscqp_shard_range cuts -- ceil(n_total / G) agents each, a shorter one, then empty ones -- start at multiples of the block size, so a
 * ragged mission is ONE in-place all-gather of the full block size as well (device g sends from g * B * per; what a short or empty
 * block sends beyond its agents lands behind the mission, in the padding) whenever G * B - n_total <= pad_agents; otherwise, and for
 * blocks of any other shape, the broadcasts above.  The three buffers lscqp_plan_group_step exchanges (LSCQP_PLAN_BUF_PLAN, _STATE,
 * _GOAL) are allocated with LSCQP_PLAN_EXCHANGE_PAD agents of such room (G * B - n_total < G <= 64 always): a group of plans cut by
 * lscqp_shard_range exchanges with one all-gather per buffer whatever the agent count.  lscqp_plan_buffer keeps reporting the
 * mission's own bytes. */
#define LSCQP_PLAN_EXCHANGE_PAD 64
#define LSCQP_XCHG_ALLGATHER 0
#define LSCQP_XCHG_BROADCAST 1
typedef struct lscqp_exchange_op {
    int32_t kind; /* LSCQP_XCHG_* */
    int32_t root; /* broadcast: the owner; all-gather: -1 */
    int64_t offset; /* doubles from the start of the buffer: broadcast: the owner's block; all-gather: 0 (block g at g * count) */
    int64_t count;  /* doubles: broadcast: the owner's block; all-gather: the (equal) block of every device */
} lscqp_exchange_op;
int lscqp_exchange_schedule(int64_t n_total, int32_t n_devices, const int64_t* first, const int64_t* count, int64_t per,
                            lscqp_exchange_op* ops, int32_t max_ops, int32_t* n_ops);
int lscqp_exchange_schedule_padded(int64_t n_total, int32_t n_devices, const int64_t* first, const int64_t* count, int64_t per, int64_t pad_agents,
                                   lscqp_exchange_op* ops, int32_t max_ops, int32_t* n_ops);

/* ---- next row of the path: the producer of the LSC rows --------------------------------
 *
 * Replaces TrajPlanner::generateLSC for agent-type obstacles (reference src/traj_planner.cpp:611-657), i.e.
 * normalVectorBetweenPolys (:1179-1205), the closest point of closestPointsBetweenPointAndConvexHull
 * (include/geometry.hpp:266-296; openGJK in the reference), downwashBetween (:1229-1240),
 * Trajectory::coordinateTransform (src/trajectory.cpp:207-219) and CollisionConstraints::setLSC
 * (src/collision_constraints.cpp:514-521) -- and writes the result directly in the packed row layout that
 * lscqp_solve_batch_device consumes (rows of local agent a at offset a * n_obs * M * (n+1), order [oi][m][i]).
 * DEVICE pointers, asynchronous on `stream`.
 *   d_traj        [n_total][M][n+1][3]  control points of ALL agents: the planning agent's initial trajectory and a
 *                 neighbour's predicted trajectory are both shifted previous plans (:273-310, 399-411); values as the
 *                 reference holds them (float32-representable; lscqp_shift_traj_device produces exactly this)
 *   d_neighbours  [n_agents][n_obs]     global agent ids of each local agent's obstacles; < 0 = none -> all-zero rows
 *                 (the solver drops them like any normal shorter than 1e-5, src/traj_optimizer.cpp:409-411)
 *   d_radius, d_downwash [n_total]      Agent::radius, Agent::downwash
 *   d_goal        [n_agents][3]         current_goal_point: fallback normal when the hull contains the origin (:624-633)
 *   first_agent   global id of local agent 0 (the local shard is [first_agent, first_agent + n_agents))
 *   d_rows_out    [n_agents * n_obs * M * (n+1)] rows
 * Dynamic (non-agent) obstacles and collision-predicted obstacles (the other branches of generateLSC) are not handled;
 * generateCLSC and generateBVC are served by lscqp_generate_constraints_device below. */
int lscqp_generate_lsc_device(lscqp_handle h, int64_t n_agents, int32_t n_obs, int64_t first_agent, const double* d_traj,
                              const int32_t* d_neighbours, const double* d_radius, const double* d_downwash,
                              const double* d_goal, lscqp_row* d_rows_out, void* stream);

/* The planner's other constraint generators for agent-type obstacles, same inputs / output layout / launch as
 * lscqp_generate_lsc_device, selected by `mode`:
 *   LSCQP_GEN_LSC   generateLSC  (src/traj_planner.cpp:611-657)  -- identical to lscqp_generate_lsc_device
 *   LSCQP_GEN_CLSC  generateCLSC (:659-706): what constructLSC() runs for the reference's default launch (mode/planner = lsc
 *                   with mode/goal = grid_based_planner, :551-553).  Segments m < M-1 as generateLSC without the fallback
 *                   normal (a hull around the origin leaves zero rows); the last segment separates the line segments
 *                   (last control point -> goal point) of the neighbour and of the agent with
 *                   closestPointsBetweenLineSegments (include/geometry.hpp:174-263) and carries one obstacle point and one
 *                   margin for all control points (CollisionConstraints::setLSC, src/collision_constraints.cpp:532-539)
 *   LSCQP_GEN_BVC   generateBVC  (:708-734): buffered Voronoi cell from the two start points
 *   d_goal_all [n_total][3]: current goal point of EVERY agent (the planning agent's own, agent.current_goal_point, and the
 *   neighbours', obstacles[oi].goal_point as broadcast), indexed by global id. */
#define LSCQP_GEN_LSC 0
#define LSCQP_GEN_CLSC 1
#define LSCQP_GEN_BVC 2
int lscqp_generate_constraints_device(lscqp_handle h, int32_t mode, int64_t n_agents, int32_t n_obs, int64_t first_agent,
                                      const double* d_traj, const int32_t* d_neighbours, const double* d_radius,
                                      const double* d_downwash, const double* d_goal_all, lscqp_row* d_rows_out, void* stream);

/* Who is whose obstacle: MultiSyncSimulator::broadcastMsgs (src/multi_sync_simulator.cpp:305-352) hands agent i every other
 * agent j with LInfinityDistance(p_i, p_j) <= communication_range (include/util.hpp:122-131; all of them if the range is <= 0),
 * in id order.  d_positions [n_total][3]: the agents' current positions (float32 values, all-gathered when sharded).
 * d_neighbours_out [n_agents][n_obs]: the global ids in ascending order, padded with -1 -- the list
 * lscqp_generate_*_device consumes.  The reference's list is unbounded; here the row buffers hold n_obs neighbours per
 * agent, so when more are in range the n_obs NEAREST are kept (L-infinity distance, ties to the smaller id; still listed in
 * id order).  d_count_out [n_agents]: how many were in range (> n_obs tells the caller that the list was cut). */
int lscqp_select_neighbours_device(lscqp_handle h, int64_t n_agents, int64_t first_agent, int64_t n_total, int32_t n_obs,
                                   double communication_range, const double* d_positions, int32_t* d_neighbours_out,
                                   int32_t* d_count_out, void* stream);

/* Replaces TrajPlanner::initialTrajPlanningPrevSol (src/traj_planner.cpp:399-411) on the solver's output: segment m of
 * the new initial trajectory := segment m+1 of the previous plan, the last segment := its last point; control points
 * truncated to float32 like TrajOptResult::desired_traj (src/traj_optimizer.cpp:71-83); dim == 2 -> z := z_2d.
 *   d_x_prev [n][dim*M*(n+1)] (x_out of lscqp_solve_batch_device)  ->  d_traj [n][M][n+1][3]
 *   shift_segments: 1 = the reference's shift; 0 = layout change and truncation only (replanning from the same state) */
int lscqp_shift_traj_device(lscqp_handle h, int64_t n, int32_t shift_segments, double z_2d, const double* d_x_prev,
                            double* d_traj, void* stream);

/* lscqp_generate_constraints_device writing into a row buffer that holds n_obs_total obstacle slots per agent: the n_obs
 * neighbour slots of this call go to slots slot0 .. slot0 + n_obs - 1 of each agent's block (the other slots belong to another
 * producer, e.g. lscqp_generate_lsc_obstacles_device).  n_obs_total = n_obs, slot0 = 0 is lscqp_generate_constraints_device. */
int lscqp_generate_constraints_device_ex(lscqp_handle h, int32_t mode, int64_t n_agents, int32_t n_obs, int64_t first_agent,
                                         const double* d_traj, const int32_t* d_neighbours, const double* d_radius,
                                         const double* d_downwash, const double* d_goal_all, lscqp_row* d_rows_out, int32_t n_obs_total,
                                         int32_t slot0, void* stream);

/* The NON-AGENT branches of TrajPlanner::generateLSC (reference src/traj_planner.cpp:611-657) with what feeds them:
 * constant-velocity prediction of a dynamic obstacle (obstaclePredictionWithPrevSol :283-285, Trajectory::planConstVelTraj,
 * src/trajectory.cpp:79-91), checkObstacleDisturbance (:312-319), obstacleSizePredictionWithConstAcc (:321-358: the obstacle's
 * radius grows with 1/2 max_acc t^2 over the uncertainty horizon, plus a velocity guard of the planning agent), downwashBetween
 * for a non-agent (:1235-1237), the z component of the relative hull dropped for obstacles taller than obs_downwash_threshold
 * (:1188-1191), margin d = predicted size + agent radius (:646-648), and the fallback normal (:624-633).
 * (normalVectorDynamicObs, :1207-1227, is dead code in the reference: col_pred_obs_indices is never filled.)
 *   d_obstacle_ids [n_agents][n_dyn]  index into d_obstacles of each local agent's obstacles, < 0 = none -> all-zero rows
 *   d_obstacles    [..]               the obstacle table (include/obstacle.hpp:13-27, the fields the planner reads)
 *   d_traj         [n_total][M][n+1][3]  initial trajectories (only the local agents' are read), d_radius [n_total]
 *   d_goal         [n_agents][3]      current goal points of the local agents;  d_hdr [n_agents]: v0 and amax[0] (velocity guard)
 *   rows go to slots slot0 .. slot0 + n_dyn - 1 of each agent's block of n_obs_total slots in d_rows_out. */
typedef struct lscqp_obstacle {
    double position[3];
    double velocity[3];
    double radius, downwash, max_acc;
    int32_t type; /* 0 = DYNAMICOBSTACLE, 1 = AGENT (include/sp_const.hpp:135-138) */
    int32_t reserved;
} lscqp_obstacle;
typedef struct lscqp_obstacle_param { /* src/param.cpp:63-67, 106-108 */
    double obs_uncertainty_horizon; /* obs/uncertainty_horizon (1) */
    double velocity_guard_ratio;    /* obs/velocity_guard_ratio (0.75) */
    double obs_downwash_threshold;  /* plan/obs_downwash_threshold (3.0) */
    double reset_threshold;         /* plan/reset_threshold (0.1; 0.5 in the launch files) */
    int32_t obs_size_prediction;    /* obs/size_prediction (true) */
    int32_t use_velocity_guard;     /* obs/use_velocity_guard (true) */
} lscqp_obstacle_param;
int lscqp_generate_lsc_obstacles_device(lscqp_handle h, const lscqp_obstacle_param* param, int64_t n_agents, int32_t n_dyn,
                                        int64_t first_agent, const double* d_traj, const int32_t* d_obstacle_ids,
                                        const lscqp_obstacle* d_obstacles, const double* d_radius, const double* d_goal,
                                        const lscqp_header* d_hdr, lscqp_row* d_rows_out, int32_t n_obs_total, int32_t slot0, void* stream);

/* initialTrajPlanningPrevSol / obstaclePredictionWithPrevSol when the simulation step is SHORTER than a segment
 * (multisim_time_step < dt, reference src/traj_planner.cpp:296-306, 413-421): segment 0 of the new initial trajectory :=
 * prev_traj[0].subSegment(fraction, 1) (Segment::subSegment, src/trajectory.cpp:15-49), the other segments are kept.
 * fraction = multisim_time_step / dt in (0, 1); layouts and float32 truncation as lscqp_shift_traj_device. */
int lscqp_shift_traj_partial_device(lscqp_handle h, int64_t n, double fraction, double z_2d, const double* d_x_prev, double* d_traj,
                                    void* stream);

/* Algorithmic HBM bytes of one lscqp_generate_lsc_device call: rows written + every agent's control points,
 * neighbour list, radius, downwash and goal read once. */
int64_t lscqp_generate_lsc_bytes(lscqp_handle h, int64_t n_agents, int32_t n_obs, int64_t n_total);

/* ---- next row of the path: the planner's other CPLEX call -------------------------------
 *
 * Replaces GoalOptimizer::solve (reference src/goal_optimizer.cpp:7-70; model :72-147): the one-variable LP
 *     min t in [0, 1 + 1e-5]   s.t.   n_r . ((g - w) t + w - p_r) - d_r >= 0
 * over the SFC faces of the last segment (if use_sfc) and the LSC rows (oi, M-1, n) of every obstacle, solved in
 * closed form on the device.  On entry hdr[q].goal = current_goal_point g and hdr[q].next_waypoint = w; on return
 * hdr[q].goal = (g - w) t* + w (:55), ready for lscqp_solve_batch*.  rows / row_offsets / sfc exactly as for the solve.
 * status_out[q] = LSCQP_STATUS_OPTIMAL, or LSCQP_STATUS_INFEASIBLE where the reference throws QPFAILED (goal unchanged).
 * |g - w| < 1e-5 returns w like the reference (:12-14). */
int lscqp_optimize_goal_device(lscqp_handle h, int64_t n, lscqp_header* d_hdr, const lscqp_row* d_rows,
                               const uint64_t* d_row_offsets, const lscqp_box* d_sfc, int32_t* d_status_out, void* stream);
/* Same, HOST pointers, synchronous (hdr is updated in place). */
int lscqp_optimize_goal(lscqp_handle h, int64_t n, lscqp_header* hdr, const lscqp_row* rows, const uint64_t* row_offsets,
                        const lscqp_box* sfc, int32_t* status_out);

/* ---- next row of the path: what the planner does with a solved trajectory ----------------
 *
 * Replaces, for a batch, TrajPlanner::isSolValid (reference src/traj_planner.cpp:990-1045: SFC containment of the control
 * points, velocity / acceleration within 1 % of the limits at the simulation step), Trajectory::getStateAt
 * (src/trajectory.cpp:156-170) and AgentManager::doStep (src/agent_manager.cpp:29-50: the agent's next state is the
 * trajectory's state at time_step).  Control points are first truncated to float32 like TrajOptResult::desired_traj
 * (src/traj_optimizer.cpp:71-83); dim == 2: z := z_2d.  DEVICE pointers, asynchronous on `stream`.
 *   d_x [n][dim*M*(n+1)]   x_out of the solve        d_hdr [n]  (max_vel / max_acc are read from it)      d_sfc [n*M] or NULL
 *   d_valid_out [n]  1 = isSolValid would return true        d_state_out [n][9]  position, velocity, acceleration */
int lscqp_validate_step_device(lscqp_handle h, int64_t n, double time_step, double z_2d, const double* d_x,
                               const lscqp_header* d_hdr, const lscqp_box* d_sfc, int32_t* d_valid_out, double* d_state_out,
                               void* stream);

/* Safety metrics of one planned step, the figures MultiSyncSimulator::update accumulates for the reference's summary CSV
 * (src/multi_sync_simulator.cpp:486-577; log/summary_*.csv columns safety_ratio_agent, vel/acc excess): for each local
 * agent and each sample time s * record_time_step, s < n_samples (the reference samples while future_time <
 * multisim_time_step - 1e-5, :489), the smallest ellipsoidal distance to any other agent over the sum of the radii
 * (ellipsoidalDistance, include/util.hpp:155-159, radius-weighted downwash :505-507) and the positive parts of
 * (v_k - max_vel_k) / max_vel_k, (a_k - max_acc_k) / max_acc_k (:560-572).  States are Trajectory::getStateAt of the
 * float32 control points.  The mission-wide figures are the min / max of these per-agent records (over ranks: one
 * MIN / MAX all-reduce).  The obstacle safety ratio (:527-557) is lscqp_safety_obstacles_device below.
 *   d_x_all [n_total][dim*M*(n+1)]  every agent's current plan (x_out of the solve, all-gathered)
 *   d_radius, d_downwash [n_total]   d_hdr [n_agents] (max_vel / max_acc)   d_out [n_agents] */
typedef struct lscqp_safety {
    double safety_ratio;        /* min over samples and other agents; +inf if there is no other agent */
    int32_t closest_agent;      /* global id attaining it (first sample, then smallest id, like the reference's strict <) */
    int32_t sample;             /* sample index attaining it */
    double vel_excess_ratio[3]; /* max over samples, 0 where the limit is kept */
    double acc_excess_ratio[3];
} lscqp_safety;
int lscqp_safety_metrics_device(lscqp_handle h, int64_t n_agents, int64_t first_agent, int64_t n_total, int32_t n_samples,
                                double record_time_step, double z_2d, const double* d_x_all, const double* d_radius,
                                const double* d_downwash, const lscqp_header* d_hdr, lscqp_safety* d_out, void* stream);
/* The OBSTACLE leg of the same loop (src/multi_sync_simulator.cpp:527-557): per local agent the minimum, over the samples of its plan
 * and over the obstacles of the mission, of ellipsoidalDistance(agent position, obstacle position, downwash) / (r_i + r_o) with the
 * mixed downwash (r_o dw_o + r_i dw_i) / (r_i + r_o) (:538-540, include/util.hpp:155-159).  The obstacle table is the one the
 * constraint generator takes (lscqp_obstacle: position, radius, downwash are read); positions are the obstacle generator's CURRENT
 * ones, the same for every sample (:534 -- only the agents move along future_time), narrowed to float32 like point3d.  Entries with
 * type == LSCQP_OBSTACLE_REAL are skipped (:531-532: "real" obstacles are tracked robots, not simulated ones).  safety_ratio_obs is
 * +inf and closest_obstacle -1 when nothing was compared (SP_INFINITY in the reference).  MIN over ranks gives the mission's figure.
 *   d_x_all [n_total][dim*M*(n+1)], d_radius / d_downwash [n_total] (only the local agents' entries are read), d_obstacles [n_obstacles],
 *   d_out [n_agents] */
#define LSCQP_OBSTACLE_REAL 2
typedef struct lscqp_safety_obs {
    double safety_ratio_obs;  /* min over samples and obstacles */
    int32_t closest_obstacle; /* index into d_obstacles attaining it (first sample, then smallest index: the reference's strict <) */
    int32_t sample;
} lscqp_safety_obs;
int lscqp_safety_obstacles_device(lscqp_handle h, int64_t n_agents, int64_t first_agent, int64_t n_total, int32_t n_samples,
                                  double record_time_step, double z_2d, const double* d_x_all, const double* d_radius,
                                  const double* d_downwash, int32_t n_obstacles, const lscqp_obstacle* d_obstacles,
                                  lscqp_safety_obs* d_out, void* stream);

/* ---- next row of the path: the producer of the SFC boxes -----------------------------------
 *
 * A build-owned voxel map replaces the reference's octomap + DynamicEDTOctomap (both third-party, absent here):
 *   lscqp_map_create           MapManager::updateOctreeFromCSV (src/map_manager.cpp:262-305): the rows of the world CSV --
 *                              boxes [n_boxes][6] = centre x,y,z, size x,y,z -- are rasterised exactly as the reference
 *                              does, over the bounding box world_min .. world_max (DynamicEDTOctomap's, :13-14,74-76),
 *                              and every voxel gets its nearest occupied cell within max_dist (the reference passes 1.0 m):
 *                              exact Euclidean distance between cell centres.  Ties between equally near cells, which
 *                              dynamicEDT3D resolves by propagation order, go to the smallest (dz, dy), then the nearer
 *                              side with -x first.  HOST pointers; the map lives in HBM.
 *   lscqp_map_create_from_csv  the same from a world CSV file (the reference's CSVRange loop)
 *   lscqp_map_info / lscqp_map_download   grid size, first octomap key, and the two fields (for inspection and tests):
 *                              occ [dims[2]][dims[1]][dims[0]] bytes; nearest, same shape, int32:
 *                              (dx+128) | (dy+128)<<8 | (dz+128)<<16 | 1<<24, or 0 = none within max_dist
 * and the corridor of each agent is updated on the device, one workgroup (4 wavefronts) per agent:
 *   lscqp_construct_sfc_device, mode
 *     LSCQP_SFC_INIT        CollisionConstraints::initializeSFC (src/collision_constraints.cpp:366-384): all M boxes :=
 *                           expandSFC of the grid cell around the position; status 0 where the reference throws
 *     LSCQP_SFC_FROM_HULL   constructSFCFromConvexHull (:414-436; the default launch, goal mode grid_based_planner): boxes
 *                           shift down by one segment, the last one := expandSFCFromConvexHull of {last point, goal point,
 *                           next waypoint} (:692-722), else of {last point, goal point} inside the previous box (:724-775),
 *                           else the previous box (status 0)
 *     LSCQP_SFC_FROM_POINT  constructSFCFromPoint (:396-412): the last box := expandSFCFromPoint(last point) with the
 *                           goal-ordered axis candidates (setAxisCand :1134-1170), else the previous box (status 0)
 *   with isObstacleInSFC (:777-808), isSFCInBoundary (:810-817) and expandSFC (:819-946) in the reference's float32 /
 *   double arithmetic -- including its quirk: where the distance-map query finds no cell within max_dist (or the sample lies
 *   outside the map) `closest_point` stays default-constructed and the sample is measured against a cell at the WORLD ORIGIN
 *   (:796-800), which cuts corridors short within margin + res/2 of the origin.  (Round 1 treated that case as "no obstacle".)
 *   d_points [n][3][3]  per agent: position (INIT) or last point of the initial trajectory, current goal point, next waypoint
 *   d_radius [n]        Agent::radius (the margin)       d_sfc [n][M] boxes, updated in place       d_status_out [n] */
typedef struct lscqp_map_s* lscqp_map;
#define LSCQP_SFC_INIT 0
#define LSCQP_SFC_FROM_HULL 1
#define LSCQP_SFC_FROM_POINT 2
int lscqp_map_create(const double* boxes, int64_t n_boxes, const double* world_min, const double* world_max, double resolution,
                     double max_dist, lscqp_map* out);
int lscqp_map_create_from_csv(const char* path, const double* world_min, const double* world_max, double resolution,
                              double max_dist, lscqp_map* out);
void lscqp_map_destroy(lscqp_map map);
/* Optional acceleration of the corridor construction, once per map (set-up time: not concurrently with corridor launches on the map):
 * a summed-area table over the cells that could make an isObstacleInSFC test fail for an agent of radius <= max_radius.  A box of
 * sample points whose cell range holds none of them passes its test without a single point being evaluated -- in open space that
 * is every test of expandSFC, the whole-box re-tests included; all other boxes are evaluated exactly as without the table, so the
 * corridors are the same bit for bit.  4 bytes per map cell.  lscqp_plan_create calls it with the largest radius of its agents. */
int lscqp_map_prepare(lscqp_map map, double max_radius);
int lscqp_map_info(lscqp_map map, int32_t* dims, int32_t* key0);
int lscqp_map_download(lscqp_map map, uint8_t* occ, int32_t* nearest);
int lscqp_construct_sfc_device(lscqp_handle h, lscqp_map map, int32_t mode, int64_t n, const double* d_points,
                               const double* d_radius, lscqp_box* d_sfc, int32_t* d_status_out, void* stream);
/* The same with a work order (round 4): workgroup k of the launch builds the corridor of agent d_order[k] (a permutation of 0 .. n-1;
 * NULL = identity) and every agent's cost -- the cycles its workgroup took, >> 4 -- is left in d_cost_out[agent] (NULL: not recorded).
 * lscqp_order_by_cost_device sorts by the costs of the PREVIOUS replan, most expensive first (stable; 16 levels scaled to the largest):
 * a corridor along a wall costs three times one in open space and a launch ends with its last workgroup -- 4096 agents 0.98 -> 0.83 ms.
 * Same boxes bit for bit in any order; lscqp_plan carries costs and order from replan to replan by itself where its launches exceed the chip. */
int lscqp_order_by_cost_device(int64_t n, const uint32_t* d_cost_prev, int32_t* d_order_out, void* stream);
int lscqp_construct_sfc_device_ordered(lscqp_handle h, lscqp_map map, int32_t mode, int64_t n, const double* d_points,
                                       const double* d_radius, lscqp_box* d_sfc, int32_t* d_status_out, const int32_t* d_order,
                                       uint32_t* d_cost_out, void* stream);
/* Same, HOST pointers, synchronous; M boxes per agent (sfc is read and updated in place). */
int lscqp_construct_sfc(lscqp_map map, int32_t mode, int32_t M, int64_t n, const double* points, const double* radius,
                        lscqp_box* sfc, int32_t* status_out);

/* ---- the caller of the path: one replan of a batch of agents as one chain of device work ----------
 *
 * Device analogue of TrajPlanner::plan / planImpl (reference src/traj_planner.cpp:33-60, 117-139) run for every local agent of
 * the mission at once, and of the loop around it in MultiSyncSimulator (src/multi_sync_simulator.cpp:305-400).  A plan object
 * owns the state the reference's planners carry from replan to replan -- previous plans (TrajPlanner::prev_traj), current goal
 * points, corridors -- plus the work buffers, all in HBM, and enqueues
 *     obstaclePrediction / initialTrajPlanning (:228-319, 360-423; all three modes,    lscqp_shift_traj(_partial)_device + the chain's
 *       checkObstacleDisturbance)                                                       prepare step
 *     broadcastMsgs' range filter (src/multi_sync_simulator.cpp:318-333)              lscqp_select_neighbours_device
 *     constructLSC (:552-569)                                                         lscqp_generate_constraints_device
 *     constructSFC / generateSFC (:571-579, 738-753; initializeSFC on the first replan) lscqp_construct_sfc_device
 *     goalPlanningWithGridBasedPlanner (:545-550)                                     lscqp_optimize_goal_device; the goal is then
 *                                                                                     held as point3d (float32) and
 *                                                                                     getTerminalSegments_old is evaluated with
 *                                                                                     octomap's float32 arithmetic
 *     trajOptimization (:755-803)                                                     lscqp_solve_batch_device_ex with the initial
 *                                                                                     trajectory as the start and the device-side
 *                                                                                     second pass; a QP that is not OPTIMAL leaves
 *                                                                                     desired_traj = initial_traj (failsafe :796-797)
 *     prev_traj = desired_traj (:51); AgentManager::doStep (src/agent_manager.cpp:29-50)  plans updated in place;
 *                                                                                     lscqp_validate_step_device (isSolValid is
 *                                                                                     reported, not acted on: the reference consults
 *                                                                                     it in DLSC mode only, :763-766)
 * on ONE stream, with no host synchronisation, allocation or copy in between.  lscqp_plan_step_graph captures that chain once in
 * a hipGraph and replays it: one graph launch per replan instead of ten kernel launches.
 * Host work that remains is what the survey leaves out of scope: the grid planner / MAPF layer writes each local agent's next
 * waypoint into LSCQP_PLAN_BUF_WAYPOINT before the step (and, unless closed_loop is set, the simulator writes the agents' states).
 * Dynamic (non-agent) obstacles are not part of the chain (lscqp_generate_lsc_obstacles_device is available separately).
 * Sharding (section 8e): rank r owns the agents [first_agent, first_agent + n_agents) of n_total; its plan needs every agent's
 * previous plan, state and goal point, so the owners' slices of LSCQP_PLAN_BUF_PLAN / _STATE / _GOAL are all-gathered between
 * steps (lscqp_allgather; the layout is [n_total][...] on every rank, so the gather is in place). */
typedef struct lscqp_plan_s* lscqp_plan;
typedef struct lscqp_agent_param { /* the Agent fields the chain reads (include/sp_const.hpp Agent; mission file values) */
    double radius, downwash;
    double max_vel[3], max_acc[3];
    double nominal_velocity;
} lscqp_agent_param;
typedef struct lscqp_plan_desc {
    int64_t n_agents;        /* local agents */
    int64_t n_total;         /* agents of the mission (each other's obstacles) */
    int64_t first_agent;     /* global id of local agent 0 */
    int32_t n_obs;           /* row slots per agent: in-range agents beyond it are cut to the nearest ones and reported in
                                LSCQP_PLAN_BUF_IN_RANGE (> n_obs), never silently */
    int32_t constraint_mode; /* LSCQP_GEN_LSC / _CLSC / _BVC (constructLSC's switch, :555-566) */
    int32_t sfc_mode;        /* LSCQP_SFC_FROM_HULL (goal mode grid_based_planner) or LSCQP_SFC_FROM_POINT; ignored without a map */
    int32_t optimize_goal;   /* != 0: GoalOptimizer moves the goal point towards the waypoint (goal mode grid_based_planner);
                                0: the goal points are left as the caller set them (static goal modes) */
    int32_t closed_loop;     /* != 0: the local agents' next states (doStep) become their current states for the next replan */
    int32_t safety_samples;  /* > 0: the step ends with the safety figures of MultiSyncSimulator::update over this many samples of the
                                new plans (lscqp_safety_metrics_device; needs n_agents == n_total: every agent's new plan), 0: off */
    double time_step;        /* multisim_time_step: == dt shifts the plans by one segment, < dt uses Segment::subSegment */
    double z_2d;             /* world_z_2d of 2-D missions */
    double record_time_step; /* spacing of the safety samples (multisim_save_time_step) */
    int32_t tight_warm_start; /* != 0: the plan's QP solves centre their warm starts in LSCQP_WARM_TIGHT mode whatever the handle's class
                                 says (a private clone of the class).  Pays where the plans barely change from replan to replan (agents
                                 holding position: forest10 chain 413 -> 320 us); with agents on the move it does not (10 agents: equal,
                                 64 / 256 agents: 20 % slower, the batch waits for its slowest QP) -- hence off by default */
    int32_t prediction_mode;  /* how the OTHER agents' trajectories are predicted (obstaclePrediction, src/traj_planner.cpp:228-253):
                                 LSCQP_TRAJ_FROM_PREVIOUS_SOLUTION (0; mode/planner lsc and dlsc, src/param.cpp:127-166: their shifted
                                 previous plans; constant velocity on the first replan, :276-279), _FROM_POSITION (mode/planner bvc:
                                 they stay where they are) or _FROM_VELOCITY (circle_test: Trajectory::planConstVelTraj from the
                                 current state) */
    int32_t initial_traj_mode; /* the planning agent's own initial trajectory (initialTrajPlanning, :360-423): same three values */
    int32_t reserved_;
    double reset_threshold;   /* checkObstacleDisturbance (:312-319, plan/reset_threshold, 0.1 in the launch files): an agent whose predicted
                                 trajectory starts further than this from its current position is predicted to stay where it is
                                 (a disturbed or externally moved robot; only with closed_loop == 0 can that happen).  <= 0: no check */
} lscqp_plan_desc;
#define LSCQP_TRAJ_FROM_PREVIOUS_SOLUTION 0
#define LSCQP_TRAJ_FROM_POSITION 1
#define LSCQP_TRAJ_FROM_VELOCITY 2
/* Buffers of a plan (device pointers through lscqp_plan_buffer; lscqp_plan_upload / _download copy synchronously).
 * "all": [n_total] entries indexed by global id; "local": [n_agents] entries. */
#define LSCQP_PLAN_BUF_STATE 0        /* in   all    double[9]: position, velocity, acceleration (float32 values, as State holds them) */
#define LSCQP_PLAN_BUF_WAYPOINT 1     /* in   local  double[3]: agent.next_waypoint */
#define LSCQP_PLAN_BUF_PLAN 2         /* i/o  all    double[dim*M*6]: previous plans in, the local agents' new plans out */
#define LSCQP_PLAN_BUF_GOAL 3         /* i/o  all    double[3]: current goal points; the local agents' are updated */
#define LSCQP_PLAN_BUF_HEADER 4       /* out  local  lscqp_header of the last replan */
#define LSCQP_PLAN_BUF_ROWS 5         /* out  local  lscqp_row[n_obs*M*6] of the last replan */
#define LSCQP_PLAN_BUF_SFC 6          /* i/o  local  lscqp_box[M]: corridors */
#define LSCQP_PLAN_BUF_STATUS 7       /* out  local  int32: LSCQP_STATUS_* of the QP */
#define LSCQP_PLAN_BUF_GOAL_STATUS 8  /* out  local  int32: LSCQP_STATUS_* of the goal LP */
#define LSCQP_PLAN_BUF_SFC_STATUS 9   /* out  local  int32: 1 = corridor updated, 0 = previous box kept / seed inside an obstacle */
#define LSCQP_PLAN_BUF_VALID 10       /* out  local  int32: isSolValid */
#define LSCQP_PLAN_BUF_IN_RANGE 11    /* out  local  int32: agents within communication range */
#define LSCQP_PLAN_BUF_NEXT_STATE 12  /* out  local  double[9]: state at time_step along the new plan (closed_loop: the local slice of
                                         LSCQP_PLAN_BUF_STATE itself) */
#define LSCQP_PLAN_BUF_OBJECTIVE 13   /* out  local  double */
#define LSCQP_PLAN_BUF_INFO 14        /* out  local  lscqp_info */
#define LSCQP_PLAN_BUF_SAFETY 15      /* out  local  lscqp_safety (safety_samples > 0) */
#define LSCQP_PLAN_BUF_COUNT 16
/* agents [n_total] (host).  map: required exactly when the class uses corridors.  The class's row_format must be LSCQP_ROWS_F64. */
int lscqp_plan_create(lscqp_handle h, lscqp_map map, const lscqp_plan_desc* desc, const lscqp_agent_param* agents, lscqp_plan* out);
void lscqp_plan_destroy(lscqp_plan plan);
/* Mission start (host pointers, synchronous): every agent hovers at its start position [n_total][3] (the plans the reference's
 * planners start from, planner_seq < 2), goal points := goal_points, or the start positions if NULL (AgentManager's constructor),
 * waypoints := the goal points; the next step is a FIRST replan (initializeSFC). */
int lscqp_plan_reset(lscqp_plan plan, const double* start_positions, const double* goal_points);
void* lscqp_plan_buffer(lscqp_plan plan, int32_t which, uint64_t* bytes_out);
int lscqp_plan_upload(lscqp_plan plan, int32_t which, const void* host, uint64_t offset, uint64_t bytes);
int lscqp_plan_download(lscqp_plan plan, int32_t which, void* host, uint64_t offset, uint64_t bytes);
/* One replan of all local agents, asynchronous on `stream` (hipStream_t as void*; NULL = default stream). */
int lscqp_plan_step(lscqp_plan plan, void* stream);
/* The same through a hipGraph captured at the first call that is not a first replan (that one runs eagerly: it differs, and it
 * warms the kernels' one-time attributes up).  Results are bit-for-bit those of lscqp_plan_step.  The captured launches carry the
 * solver class by value; the plan notices lscqp_update(h) / lscqp_map_prepare(map) by their generation counters, drops the graph and
 * re-captures at the next call (both step entry points).  An update that changes the SHAPE the plan was sized for -- segments,
 * dimension, corridor rows on/off, row format, a neighbour capacity below desc.n_obs, dt below desc.time_step -- is refused with
 * LSCQP_ERR_INVALID_ARGUMENT at the next step: such a plan has to be destroyed and created again. */
int lscqp_plan_step_graph(lscqp_plan plan, void* stream);
int64_t lscqp_plan_graph_nodes(lscqp_plan plan); /* nodes of the captured graph, 0 before the capture */
/* One replan of a mission spread over the devices of a communicator (section 8e): plans[g] was created with device g of `c` current
 * (its own handle and map live there too) and owns the block [first_agent, first_agent + n_agents) of the n_total agents, blocks
 * consecutive in device order and covering the mission.  Every plan's chain is enqueued on its device's stream
 * (lscqp_comm_stream), eagerly or through its graph, followed on the same streams by the one exchange the reference has --
 * MultiSyncSimulator::broadcastMsgs, src/multi_sync_simulator.cpp:305-352: every planner receives the others' previous plans --
 * as an in-place RCCL exchange of the owners' slices of LSCQP_PLAN_BUF_PLAN / _STATE / _GOAL (grouped ncclAllGather for equal
 * blocks, one ncclBroadcast per owner otherwise).  Asynchronous: lscqp_comm_synchronize waits for all devices.  Waypoints are
 * written per plan before the call, as for lscqp_plan_step. */
int lscqp_plan_group_step(lscqp_comm c, const lscqp_plan* plans, int32_t use_graph);

/* ---- work counters of a launch (the fp64-VALU figure next to the HBM one) -----------------------------
 *
 * The kernel is bound by fp64 vector issue, not by HBM; the figure that goes with that roof is
 * "iterations x flops per iteration".  The iterations are the kernel's own count (lscqp_info.iterations, per instance); the flops
 * of one pass through the iteration body are a property of the kernel instance's machine code and are read off it when the library
 * is built (lsc_dr_planner_amd/isa_work.py: fp64 vector instructions between the position markers of the loop body; FMA = 2 flops
 * per lane, any other fp64 arithmetic instruction 1; x 64 lanes x wavefronts per instance).  lscqp_instance_work reports them for
 * the instance a launch of n QPs with n_obs_max obstacles selects -- the same selection lscqp_solve_batch_device makes.
 *   flops of an instance that ran `it` iterations  =  flops_fixed + it * flops_per_iteration + flops_last_pass
 * (the pass that detects convergence runs the residual pass and the test only).  These are the lane-flops the SIMDs EXECUTE,
 * masked lanes and the in-line blocks most iterations skip included (about 15 % above the hardware's own SQ_INSTS_VALU_*_F64
 * count, profiles/), i.e. an upper bound of the useful work and the quantity the 78.6 TFLOP/s fp64 vector peak of MI355X is about.
 * Regions of the nested-dissection instances that only one or two of the workgroup's wavefronts execute are counted for those
 * wavefronts only (the source brackets them with position markers). */
typedef struct lscqp_work {
    double flops_fixed;              /* prologue + epilogue */
    double flops_per_iteration;
    double flops_last_pass;
    double f64_insts_per_iteration;  /* per wavefront (averaged over the workgroup's wavefronts where they run different code) */
    double valu_insts_per_iteration; /* per wavefront, all vector-ALU instructions */
    double lds_insts_per_iteration;  /* per wavefront */
    double valu_insts_fixed;         /* per wavefront: prologue + epilogue + last pass */
    int32_t wavefronts;              /* per QP */
    int32_t nslot;                   /* LSC row slots per lane */
    int32_t max_obstacles;           /* capacity of the instance */
    int32_t lds_bytes;               /* per workgroup */
    char kernel[96];                 /* "lscqp_pdip_kernel<M,DIM,ES,NSLOT,W,FT>" */
} lscqp_work;
int lscqp_instance_work(lscqp_handle h, int64_t n, int32_t n_obs_max, lscqp_work* out);

/* ---- failure diagnostics (the reference's answer to a failed QP) -----------------------------------------------------------
 *
 * When CPLEX fails the reference exports the model as an LP file and runs the conflict refiner to NAME the rows that cannot hold
 * together (src/traj_optimizer.cpp:103-137; :45-52 with param.log_solver), and its caller prints every SFC / LSC row that the
 * fallback trajectory `initial_traj` violates, with obstacle, segment, control point and margin (src/traj_planner.cpp:767-797).
 *   lscqp_diagnose[_device]  evaluates every row of the reference's model (populatebyrow, src/traj_optimizer.cpp:238-511) on the
 *                            trajectories x [n][nv] (reference variable order, world frame): per row family the largest violation
 *                            (<= 0: satisfied, then the smallest margin negated) and the number of rows violated by more than `tol`,
 *                            plus the most violated row by name.  On initial_traj it IS the caller's debug loop; on the x_out of a
 *                            non-OPTIMAL instance (the iterate the solver stopped at) it names the rows that instance cannot satisfy.
 *                            Units are the reference's: metres for bounds / SFC / LSC (normal-weighted) / communication rows,
 *                            m/s and m/s^2 for the dynamic limits, the row's own scaling for the equalities.
 *   lscqp_dump_instance      one instance as a CPLEX LP file, rows and variable names (x_m_i, y_m_i, z_m_i) as populatebyrow
 *                            creates them -- what cplex.exportModel writes to log/QPmodel_trajOpt.lp; HOST pointers, rows / sfc of
 *                            THIS instance ([n_obs*P] rows, [M] boxes); needs no device.
 *   lscqp_row_family_name    "SFC", "LSC", ... for messages. */
enum {
    LSCQP_ROW_BOUND = 0,         /* variable box = world box (:252-265) */
    LSCQP_ROW_SFC = 1,           /* corridor faces (:372-397) */
    LSCQP_ROW_LSC = 2,           /* LSC / BVC half-spaces (:399-437); obstacle = oi */
    LSCQP_ROW_VEL = 3,           /* :448-453 */
    LSCQP_ROW_ACC = 4,           /* :462-471 */
    LSCQP_ROW_COMM_PAIR = 5,     /* :480-489; obstacle = mi (the segment whose first point the row ties to) */
    LSCQP_ROW_COMM_WAYPOINT = 6, /* :492-498 */
    LSCQP_ROW_EQUALITY = 7,      /* initial state, joins, end stop (:318-368, 502-511) */
    LSCQP_ROW_FAMILIES = 8
};
typedef struct lscqp_diag { /* 128 bytes */
    double worst[LSCQP_ROW_FAMILIES];    /* largest violation per family (0 for a family without rows) */
    int32_t violated[LSCQP_ROW_FAMILIES]; /* rows violated by more than tol */
    double violation;                     /* of the most violated row overall ... */
    int32_t family, obstacle, segment, point, axis, reserved; /* ... and its name; -1 where a field does not apply */
} lscqp_diag;
int lscqp_diagnose_device(lscqp_handle h, int64_t n, const lscqp_header* d_hdr, const lscqp_row* d_rows, const uint64_t* d_row_offsets,
                          const lscqp_box* d_sfc, const double* d_x, double tol, lscqp_diag* d_out, void* stream);
int lscqp_diagnose(lscqp_handle h, int64_t n, const lscqp_header* hdr, const lscqp_row* rows, const uint64_t* row_offsets, const lscqp_box* sfc,
                   const double* x, double tol, lscqp_diag* out);
int lscqp_dump_instance(lscqp_handle h, const lscqp_header* hdr, const lscqp_row* rows, const lscqp_box* sfc, const char* path);
const char* lscqp_row_family_name(int32_t family);

/* Number of inequality rows populatebyrow adds for an agent with n_obs obstacles (SFC + LSC + velocity +
 * acceleration + communication, src/traj_optimizer.cpp:370-500), not counting rows dropped for tiny normals. */
int lscqp_num_inequalities(lscqp_handle h, int32_t n_obs);

/* Algorithmic HBM bytes of one instance: rows + boxes + header in, x/obj/status out. */
int64_t lscqp_algorithmic_bytes(lscqp_handle h, int32_t n_obs);

/* Human-readable description of the last error on this thread. */
const char* lscqp_last_error(void);

/* "lscqp <major>.<minor> (gfx950, ...)" */
const char* lscqp_version(void);

#ifdef __cplusplus
}
#endif
#endif /* LSCQP_H */
