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
