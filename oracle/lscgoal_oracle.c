/*
 * lscgoal_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's goal LP, row for row as GoalOptimizer::populatebyrow builds it
 * (reference src/goal_optimizer.cpp:72-147), on the reference's own LSC / Box records (not on the packed rows of the
 * product):
 *   variable            t in [0, 1 + SP_EPSILON_FLOAT]                          :106-110
 *   objective           min t                                                   :112-115
 *   SFC rows            faces of constraints.getSFC(M-1), Box::convertToLSCs    :118-135, src/collision_constraints.cpp:37-59
 *   LSC rows            constraints.getLSC(oi, M-1, n), skipped if |n| < 1e-5    :138-155
 *   result              goal = (g - w) t + w                                    :55;  |g - w| < 1e-5 -> w  :12-14
 * The LP itself is solved by CPLEX in the reference (absent here).  Deliberately not the product's running-maximum
 * formula: every candidate optimum (0 and every row's breakpoint) is tested against ALL rows and the smallest feasible
 * one is returned.  PARITY PINNING: scipy.optimize.linprog (HiGHS) solutions of the same rows, tests/golden/goal_lp.json
 * (tools/make_golden_goal.py), and the reference log's agent-1 case (goal x = 2.55).
 */
#include <math.h>
#include <stdlib.h>

#include "lscqp_oracle.h"

#define GOAL_EPS_FLOAT 1e-5
#define GOAL_FEAS_TOL 1e-9

/* rows a_r t + c_r >= 0 of the model; returns the number of rows */
int orc_goal_rows(const orc_class* cls, const double* g, const double* w, int n_obs, const orc_lsc* lsc, const orc_box* sfc_last,
                  double* a, double* c) {
    int nr = 0;
    const int dim = cls->dim, M = cls->M;
    if (cls->use_sfc) {
        /* Box::convertToLSCs: for each axis the faces (+e_k, d = box_min_k) and (-e_k, d = -box_max_k), point 0 */
        for (int k = 0; k < dim; k++) {
            a[nr] = (g[k] - w[k]);
            c[nr] = w[k] - sfc_last->bmin[k];
            nr++;
            a[nr] = -(g[k] - w[k]);
            c[nr] = -w[k] + sfc_last->bmax[k];
            nr++;
        }
    }
    for (int oi = 0; oi < n_obs; oi++) {
        const orc_lsc* L = &lsc[((size_t)oi * M + (M - 1)) * 6 + 5];
        const double nn = sqrt(L->nrm[0] * L->nrm[0] + L->nrm[1] * L->nrm[1] + L->nrm[2] * L->nrm[2]);
        if (nn < GOAL_EPS_FLOAT) continue;
        double aa = 0, cc = 0;
        for (int k = 0; k < dim; k++) {
            aa += L->nrm[k] * (g[k] - w[k]);
            cc += L->nrm[k] * (w[k] - L->p[k]);
        }
        a[nr] = aa;
        c[nr] = cc - L->d;
        nr++;
    }
    return nr;
}

/* returns 0 (optimal, goal_out set) or 1 (infeasible) */
int orc_goal_opt(const orc_class* cls, const double* g, const double* w, int n_obs, const orc_lsc* lsc, const orc_box* sfc_last,
                 double* goal_out, double* t_out) {
    double d2 = 0;
    for (int k = 0; k < 3; k++) d2 += (g[k] - w[k]) * (g[k] - w[k]);
    if (sqrt(d2) < GOAL_EPS_FLOAT) {
        for (int k = 0; k < 3; k++) goal_out[k] = w[k];
        if (t_out) *t_out = 0;
        return 0;
    }
    const int cap = 2 * cls->dim + n_obs + 2;
    double* a = (double*)malloc(sizeof(double) * cap);
    double* c = (double*)malloc(sizeof(double) * cap);
    double* cand = (double*)malloc(sizeof(double) * (cap + 2));
    const int nr = orc_goal_rows(cls, g, w, n_obs, lsc, sfc_last, a, c);
    const double ub = 1.0 + GOAL_EPS_FLOAT;
    int nc = 0;
    cand[nc++] = 0.0;
    for (int r = 0; r < nr; r++)
        if (a[r] != 0) cand[nc++] = -c[r] / a[r];
    double best = INFINITY;
    for (int i = 0; i < nc; i++) {
        const double t = cand[i];
        if (t < -GOAL_FEAS_TOL || t > ub + GOAL_FEAS_TOL || t >= best) continue;
        int ok = 1;
        for (int r = 0; r < nr && ok; r++) {
            const double scale = fabs(a[r]) > 1 ? fabs(a[r]) : 1.0;
            if (a[r] * t + c[r] < -GOAL_FEAS_TOL * scale) ok = 0;
        }
        if (ok) best = t;
    }
    free(a);
    free(c);
    free(cand);
    if (!isfinite(best)) return 1;
    if (best < 0) best = 0;
    if (best > ub) best = ub;
    for (int k = 0; k < 3; k++) goal_out[k] = (g[k] - w[k]) * best + w[k];
    if (t_out) *t_out = best;
    return 0;
}
