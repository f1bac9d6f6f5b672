/*
 * lscpred_oracle.c -- TEST INFRASTRUCTURE (CPU oracle).  Restates, in plain C, the parts of the reference's per-replan
 * pipeline that sit between the previous plan and the constraint generators when an obstacle is NOT an agent or the
 * simulation step is shorter than a segment:
 *
 *   Segment<point3d>::subSegment                       src/trajectory.cpp:15-49
 *   Trajectory<T>::planConstVelTraj                     src/trajectory.cpp:79-91
 *   TrajPlanner::initialTrajPlanningPrevSol             src/traj_planner.cpp:399-423   (both time-step cases)
 *   TrajPlanner::obstaclePredictionWithPrevSol          src/traj_planner.cpp:273-310   (non-agent obstacles: constant velocity)
 *   TrajPlanner::checkObstacleDisturbance               src/traj_planner.cpp:312-319
 *   TrajPlanner::obstacleSizePredictionWithConstAcc     src/traj_planner.cpp:321-358
 *   TrajPlanner::generateLSC for a non-agent obstacle   src/traj_planner.cpp:611-657 with downwashBetween :1229-1240,
 *                                                       normalVectorBetweenPolys :1179-1205 (z dropped for tall obstacles)
 * (normalVectorDynamicObs, :1207-1227, is dead code: col_pred_obs_indices is never inserted into.)
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
 * Compiled WITHOUT FMA contraction (octomap::point3d is float32 arithmetic; the reference's x86-64 build has no FMA).
 */
#include <math.h>
#include <string.h>

#include "lscqp_oracle.h"



static int n_choose_k(int n, int k) { /* include/polynomial.hpp:9-20 */
    if (k > n) return 0;
    if (k * 2 > n) k = n - k;
    if (k == 0) return 1;
    int result = n;
    for (int i = 2; i <= k; i++) {
        result *= (n - i + 1);
        result /= i;
    }
    return result;
}

/* B (Bernstein -> monomial, include/polynomial.hpp:281-294) and its inverse for n = 5.  The reference inverts numerically
 * (Eigen); the inverse is known in closed form, t^i = sum_{j >= i} C(j,i) / C(n,i) b_{j,n}(t), exact to rounding. */
static void bernstein_matrices(double B[6][6], double Binv[6][6]) {
    const int n = 5;
    for (int i = 0; i <= n; i++)
        for (int j = 0; j <= n; j++) {
            B[i][j] = (j >= i) ? n_choose_k(n, i) * n_choose_k(n - i, n - j) * (((j - i) & 1) ? -1.0 : 1.0) : 0.0;
            Binv[i][j] = (j >= i) ? (double)n_choose_k(j, i) / (double)n_choose_k(n, i) : 0.0;
        }
}

/* src/trajectory.cpp:15-49: control points of the piece [t0, tf] (normalised) of one degree-5 segment.
 * cp, out: [6][3] doubles holding float32 values (point3d); the matrix product is evaluated left to right in double like
 * `c_tr * B * A * B_inv`, the result is rounded to float32 by the point3d constructor. */
void orc_sub_segment(const double* cp, double t0, double tf, double* out) {
    double B[6][6], Binv[6][6], A[6][6];
    bernstein_matrices(B, Binv);
    const double b = t0, a = tf - t0;
    memset(A, 0, sizeof A);
    for (int i = 0; i < 6; i++)
        for (int j = 0; j <= i; j++) A[i][j] = n_choose_k(i, j) * pow(a, j) * pow(b, i - j);
    for (int k = 0; k < 3; k++) {
        double c[6], t1[6], t2[6];
        for (int i = 0; i < 6; i++) c[i] = (double)(float)cp[3 * i + k];
        for (int j = 0; j < 6; j++) {
            t1[j] = 0;
            for (int l = 0; l < 6; l++) t1[j] += c[l] * B[l][j];
        }
        for (int j = 0; j < 6; j++) {
            t2[j] = 0;
            for (int l = 0; l < 6; l++) t2[j] += t1[l] * A[l][j];
        }
        for (int j = 0; j < 6; j++) {
            double v = 0;
            for (int l = 0; l < 6; l++) v += t2[l] * Binv[l][j];
            out[3 * j + k] = (double)(float)v;
        }
    }
}

/* src/traj_planner.cpp:399-423 on control points: prev [M][6][3] (float32 values) -> init [M][6][3].
 * fraction = multisim_time_step / dt: 1 -> shift by one segment, last segment := last point; < 1 -> segment 0 := subSegment(fraction, 1),
 * the others unchanged. */
void orc_shift_prev_plan(int M, double fraction, const double* prev, double* init) {
    if (fraction >= 1.0) {
        for (int m = 0; m < M; m++)
            for (int i = 0; i < 6; i++)
                for (int k = 0; k < 3; k++) init[(m * 6 + i) * 3 + k] = (m == M - 1) ? prev[((M - 1) * 6 + 5) * 3 + k] : prev[((m + 1) * 6 + i) * 3 + k];
    } else {
        memcpy(init, prev, sizeof(double) * M * 18);
        orc_sub_segment(prev, fraction, 1.0, init);
    }
}

/* src/trajectory.cpp:79-91 with point3d (float) arithmetic: p + v * (float)time, time accumulated in double */
void orc_const_vel_traj(int M, double dt, const double* pos, const double* vel, double* out) {
    double time = 0;
    for (int m = 0; m < M; m++)
        for (int i = 0; i < 6; i++) {
            for (int k = 0; k < 3; k++) {
                const float pv = (float)vel[k] * (float)time;
                out[(m * 6 + i) * 3 + k] = (double)((float)pos[k] + pv);
            }
            time += dt / 5;
        }
}

typedef struct orc_obstacle { /* the Obstacle fields the planner reads for a non-agent obstacle (include/obstacle.hpp:13-27) */
    double position[3], velocity[3];
    double radius, downwash, max_acc;
    int type; /* 0 = DYNAMICOBSTACLE, 1 = AGENT (include/sp_const.hpp:135-138) */
    int pad;
} orc_obstacle;

typedef struct orc_obs_param { /* Param fields of the obstacle branches (src/param.cpp:63-67, 106-108) */
    double dt;
    double obs_uncertainty_horizon, velocity_guard_ratio, obs_downwash_threshold, reset_threshold;
    int obs_size_prediction, use_velocity_guard;
} orc_obs_param;

/* src/traj_planner.cpp:321-358: size control points [M][6] of one obstacle as seen by an agent with velocity v_agent and
 * max_acc[0] = amax0.  (planner_mode != RECIPROCALRSFC.) */
void orc_obstacle_sizes(int M, const orc_obs_param* p, const orc_obstacle* o, const double* v_agent, double amax0, double* size) {
    const int Mu = (int)((p->obs_uncertainty_horizon + 1e-9) / p->dt); /* SP_EPSILON = 1e-9 */
    double guard = 0;
    if (p->use_velocity_guard) {
        const float vx = (float)v_agent[0], vy = (float)v_agent[1], vz = (float)v_agent[2];
        const double nsq = (double)(vx * vx + vy * vy + vz * vz); /* octomath norm_sq(): float expression */
        guard = p->velocity_guard_ratio * nsq / amax0;
    }
    if (p->obs_size_prediction && o->type == 0) {
        double B[6][6], Binv[6][6];
        bernstein_matrices(B, Binv);
        for (int m = 0; m < Mu && m < M; m++) {
            const double c0 = 0.5 * o->max_acc * pow(m * p->dt, 2), c1 = o->max_acc * m * p->dt * p->dt, c2 = 0.5 * o->max_acc * pow(p->dt, 2);
            for (int i = 0; i < 6; i++) size[m * 6 + i] = o->radius + guard + (c0 * Binv[0][i] + c1 * Binv[1][i] + c2 * Binv[2][i]);
        }
        for (int m = Mu; m < M; m++)
            for (int i = 0; i < 6; i++) size[m * 6 + i] = o->radius + guard + 0.5 * o->max_acc * pow(Mu * p->dt, 2);
    } else {
        for (int m = 0; m < M; m++)
            for (int i = 0; i < 6; i++) size[m * 6 + i] = o->radius; /* planConstVelTraj(radius, 0) on Trajectory<double> */
    }
}

/* One agent against one NON-AGENT obstacle, all segments: prediction (:283-285), disturbance reset (:312-319), sizes, generateLSC.
 *   own  [M][6][3] the agent's initial trajectory;  goal: agent.current_goal_point;  r_own: agent.radius
 *   out  [M][6] LSC records */
void orc_generate_lsc_obstacle(int M, int dim, const orc_obs_param* p, const double* own, const double* goal, double r_own,
                               const double* v_agent, double amax0, const orc_obstacle* o, orc_lsc* out) {
    double pred[6 * 3 * 32], size[6 * 32];
    if (M > 32) return;
    orc_const_vel_traj(M, p->dt, o->position, o->velocity, pred);
    { /* checkObstacleDisturbance: the prediction starts at the obstacle's position by construction; kept for the record */
        float d2 = 0;
        for (int k = 0; k < 3; k++) {
            const float dk = (float)pred[k] - (float)o->position[k];
            d2 += dk * dk;
        }
        if (sqrt((double)d2) > p->reset_threshold) {
            const double zero[3] = {0, 0, 0};
            orc_const_vel_traj(M, p->dt, o->position, zero, pred);
        }
    }
    orc_obstacle_sizes(M, p, o, v_agent, amax0, size);
    /* downwashBetween for a non-agent obstacle (:1235-1237); 2-D missions plan in the plane */
    const double downwash = (dim == 3) ? (r_own + o->downwash * o->radius) / (r_own + o->radius) : 1.0;
    const float dwf = (float)downwash;
    const int flat = (o->type == 0 && o->downwash > p->obs_downwash_threshold);
    for (int m = 0; m < M; m++) {
        double rel[18];
        float relf[18];
        for (int i = 0; i < 6; i++) {
            const double *a = &own[(m * 6 + i) * 3], *b = &pred[(m * 6 + i) * 3];
            float az = (float)a[2], bz = (float)b[2];
            if (dim == 3) {
                az = az / dwf;
                bz = bz / dwf;
            }
            relf[3 * i + 0] = (float)a[0] - (float)b[0];
            relf[3 * i + 1] = (float)a[1] - (float)b[1];
            relf[3 * i + 2] = (dim == 3 && !flat) ? az - bz : 0.0f; /* :1188-1191: z dropped for tall dynamic obstacles */
            for (int k = 0; k < 3; k++) rel[3 * i + k] = (double)relf[3 * i + k];
        }
        double cp[3];
        orc_hull_closest_point(rel, 6, cp);
        float nf[3] = {(float)cp[0], (float)cp[1], (float)cp[2]};
        float len = sqrtf(nf[0] * nf[0] + nf[1] * nf[1] + nf[2] * nf[2]);
        if (len < 1e-5f) { /* :624-633 */
            nf[0] = (float)(goal[0] - o->position[0]);
            nf[1] = (float)(goal[1] - o->position[1]);
            nf[2] = (dim == 3) ? (float)(goal[2] - o->position[2]) / dwf : 0.0f;
            len = sqrtf(nf[0] * nf[0] + nf[1] * nf[1] + nf[2] * nf[2]);
        }
        if (len > 0.0f)
            for (int k = 0; k < 3; k++) nf[k] /= len;
        for (int i = 0; i < 6; i++) {
            orc_lsc* l = &out[m * 6 + i];
            for (int k = 0; k < 3; k++) l->p[k] = (double)(float)pred[(m * 6 + i) * 3 + k];
            l->nrm[0] = (double)nf[0];
            l->nrm[1] = (double)nf[1];
            l->nrm[2] = (dim == 3) ? (double)(float)((double)nf[2] / downwash) : 0.0; /* :653 */
            l->d = size[m * 6 + i] + r_own;                                            /* :646-648 */
        }
    }
}
