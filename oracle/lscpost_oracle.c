/*
 * lscpost_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the planner's post-solve checks:
 *   TrajPlanner::isSolValid   reference src/traj_planner.cpp:990-1045 (SFC containment :992-1010 with Box::isPointInBox,
 *                             src/collision_constraints.cpp:81-88; dynamic limits at multisim_time_step :1029-1042; the LSC
 *                             check is commented out in the reference, :1013-1027)
 *   AgentManager::doStep      src/agent_manager.cpp:29-50 (next state = getFutureState(time_step); 2-D: z := world_z_2d)
 * on top of orc_state_at (oracle/lscqp_oracle.c: Trajectory::getStateAt, src/trajectory.cpp:111-199), which is pinned
 * against the reference's result log (tests/golden/kat_log.json: states at t = 0.1 s and 0.2 s).
 */
#include <math.h>
#include <stddef.h>

#include "lscqp_oracle.h"

/* x: raw fp64 solution [dim][M][6]; returns 1 if isSolValid would accept; state9 = {p, v, a} at time_step (float32 values) */
int orc_validate_step(const orc_class* c, const orc_agent* ag, const orc_box* sfc, const double* x, double time_step, double z_2d,
                      double* state9) {
    const int M = c->M, dim = c->dim, P = M * 6;
    double xf[3 * 6 * 16];
    for (int i = 0; i < dim * P; i++) xf[i] = (double)(float)x[i]; /* desired_traj is float32 (src/traj_optimizer.cpp:71-83) */
    int ok = 1;
    if (c->use_sfc) {
        for (int m = 0; m < M; m++)
            for (int i = (m == 0 ? 3 : 0); i < 6; i++)
                for (int k = 0; k < 3; k++) {
                    const double v = (k < dim) ? xf[k * P + 6 * m + i] : (double)(float)z_2d;
                    const double lo = (double)(float)sfc[m].bmin[k], hi = (double)(float)sfc[m].bmax[k];
                    if (!(v > lo - 1e-5 && v < hi + 1e-5)) ok = 0;
                }
    }
    double pos[3] = {0, 0, 0}, vel[3] = {0, 0, 0}, acc[3] = {0, 0, 0};
    orc_state_at(c, xf, time_step, pos, vel, acc);
    for (int k = 0; k < 3; k++) {
        state9[k] = (k < dim) ? (double)(float)pos[k] : (double)(float)z_2d;
        state9[3 + k] = (k < dim) ? (double)(float)vel[k] : 0.0;
        state9[6 + k] = (k < dim) ? (double)(float)acc[k] : 0.0;
        if (k < dim) {
            if (fabs(state9[3 + k]) > ag->vmax[k] * 1.01) ok = 0;
            if (fabs(state9[6 + k]) > ag->amax[k] * 1.01) ok = 0;
        }
    }
    return ok;
}

/*
 * MultiSyncSimulator::update's safety metrics (reference src/multi_sync_simulator.cpp:486-577) for agents
 * [first, first + n_agents) of n_total: per agent the minimum over samples t_s = s * step and over the other agents of
 * ellipsoidalDistance (include/util.hpp:155-159) / (r_i + r_j), the first (sample, j) attaining it, and the positive parts
 * of the signed velocity / acceleration excess ratios (:560-572).  out: [n_agents][9] = ratio, j, sample, vex[3], aex[3].
 */
void orc_safety_metrics(const orc_class* c, int n_agents, int first, int n_total, int n_samples, double step, double z_2d,
                        const double* x_all, const double* radius, const double* downwash, const orc_agent* ag, double* out) {
    const int nv = c->dim * c->M * 6;
    for (int a = 0; a < n_agents; a++) {
        const int gi = first + a;
        double best = INFINITY, bj = -1, bs = -1, vex[3] = {0, 0, 0}, aex[3] = {0, 0, 0};
        for (int s = 0; s < n_samples; s++) {
            const double t = s * step;
            double xf[3 * 6 * 16], pi[3] = {0, 0, 0}, vi[3] = {0, 0, 0}, ai[3] = {0, 0, 0};
            for (int i = 0; i < nv; i++) xf[i] = (double)(float)x_all[(size_t)gi * nv + i];
            orc_state_at(c, xf, t, pi, vi, ai);
            if (c->dim == 2) pi[2] = z_2d;
            for (int k = 0; k < c->dim; k++) {
                const double ve = ((double)(float)vi[k] - ag[a].vmax[k]) / ag[a].vmax[k];
                const double ae = ((double)(float)ai[k] - ag[a].amax[k]) / ag[a].amax[k];
                if (ve > 0 && ve > vex[k]) vex[k] = ve;
                if (ae > 0 && ae > aex[k]) aex[k] = ae;
            }
            for (int j = 0; j < n_total; j++) {
                if (j == gi) continue;
                double pj[3] = {0, 0, 0}, vj[3], aj[3];
                for (int i = 0; i < nv; i++) xf[i] = (double)(float)x_all[(size_t)j * nv + i];
                orc_state_at(c, xf, t, pj, vj, aj);
                if (c->dim == 2) pj[2] = z_2d;
                const double dwn = (downwash[gi] * radius[gi] + downwash[j] * radius[j]) / (radius[gi] + radius[j]);
                const float dx = (float)pi[0] - (float)pj[0], dy = (float)pi[1] - (float)pj[1];
                const float dz = (float)((double)((float)pi[2] - (float)pj[2]) / dwn);
                const float nsq = dx * dx + dy * dy + dz * dz;
                const double ratio = sqrt((double)nsq) / (radius[gi] + radius[j]);
                if (ratio < best) {
                    best = ratio;
                    bj = j;
                    bs = s;
                }
            }
        }
        double* o = &out[(size_t)a * 9];
        o[0] = best, o[1] = bj, o[2] = bs;
        for (int k = 0; k < 3; k++) o[3 + k] = vex[k], o[6 + k] = aex[k];
    }
}

/*
 * The OBSTACLE leg of the same loop (reference src/multi_sync_simulator.cpp:527-557): per agent the minimum over samples and
 * over the non-"real" obstacles of ellipsoidalDistance(agent position at t_s, obstacle position, mixed downwash) /
 * (r_i + r_o), with downwash = (r_o * dw_o + r_i * dw_i) / (r_i + r_o) (:538-540).  The obstacle positions are those of
 * obstacle_generator.getObstacle(oi) -- the generator's CURRENT state, the same for every sample of the step (:534; only the
 * agents move along future_time) -- held as point3d, i.e. float32.  obs: [n_obs][5] = x, y, z, radius, downwash; skip[o] != 0
 * marks a "real" obstacle (:531-532).  out: [n_agents][3] = ratio (+inf if nothing was compared), obstacle index, sample of the
 * first strict minimum in (sample, obstacle) order.
 */
void orc_safety_obstacles(const orc_class* c, int n_agents, int first, int n_samples, double step, double z_2d, const double* x_all,
                          const double* radius, const double* downwash, int n_obs, const double* obs, const int* skip, double* out) {
    const int nv = c->dim * c->M * 6;
    for (int a = 0; a < n_agents; a++) {
        const int gi = first + a;
        double best = INFINITY, bo = -1, bs = -1;
        double xf[3 * 6 * 16];
        for (int i = 0; i < nv; i++) xf[i] = (double)(float)x_all[(size_t)gi * nv + i];
        for (int s = 0; s < n_samples; s++) {
            const double t = s * step;
            double pi[3] = {0, 0, 0}, vi[3] = {0, 0, 0}, ai[3] = {0, 0, 0};
            orc_state_at(c, xf, t, pi, vi, ai);
            if (c->dim == 2) pi[2] = z_2d;
            for (int o = 0; o < n_obs; o++) {
                if (skip && skip[o]) continue;
                const double* ob = &obs[(size_t)o * 5];
                const double dwn = (ob[3] * ob[4] + radius[gi] * downwash[gi]) / (radius[gi] + ob[3]);
                const float dx = (float)pi[0] - (float)ob[0], dy = (float)pi[1] - (float)ob[1];
                const float dz = (float)((double)((float)pi[2] - (float)ob[2]) / dwn);
                const float nsq = dx * dx + dy * dy + dz * dz;
                const double ratio = sqrt((double)nsq) / (radius[gi] + ob[3]);
                if (ratio < best) {
                    best = ratio;
                    bo = o;
                    bs = s;
                }
            }
        }
        out[(size_t)a * 3] = best, out[(size_t)a * 3 + 1] = bo, out[(size_t)a * 3 + 2] = bs;
    }
}
