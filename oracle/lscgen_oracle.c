/*
 * lscgen_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's LSC generation for agent-type obstacles, the producer of the
 * rows the trajectory QP consumes.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * What it follows (paths relative to the reference checkout):
 *   TrajPlanner::generateLSC                 src/traj_planner.cpp:611-657   (loop, safety margin d_i, z back-scaling)
 *   TrajPlanner::normalVectorBetweenPolys    src/traj_planner.cpp:1179-1205 (relative control points, unit normal)
 *   TrajPlanner::downwashBetween             src/traj_planner.cpp:1229-1240 (radius-weighted downwash)
 *   Trajectory::coordinateTransform          src/trajectory.cpp:207-219     (z /= (float)downwash)
 *   closestPointsBetweenPointAndConvexHull   include/geometry.hpp:266-296   (closest point of a 6-point hull to 0)
 *   CollisionConstraints::setLSC             src/collision_constraints.cpp:514-521 (one LSC per control point)
 *   LSC                                      include/collision_constraints.hpp:17-33
 *
 * The closest point itself is computed by the reference with openGJK (src/openGJK/openGJK.cpp, vendored in the
 * reference tree).  The closest point of a convex hull to the origin is unique, so any exact method must agree with
 * it; here it is found by enumeration of the hull's candidate faces (6 vertices, 15 edges, 20 triangles) plus a
 * supporting-plane test for a hull that contains the origin.
 * PARITY PINNING: tests/test_lscgen.py checks this routine against the REFERENCE's openGJK — compiled from the
 * reference's own source into oracle/_ref/libref_gjk.so (oracle/Makefile, target ref) when /root/reference is
 * present, and through the committed outputs of that library (tests/golden/gjk_hulls.json, tools/make_golden_gjk.py)
 * everywhere else.
 *
 * float32 semantics: the reference's control points, relative points, closest point and normal are octomap::point3d
 * (3 x float); the stages below round to float where the reference holds a point3d, GJK itself runs in double.
 */
#include <math.h>
#include <string.h>

#include "lscqp_oracle.h"

/* closest point to the origin on conv{p_0..p_{k-1}} (k <= 8), enumeration of vertices / edges / triangles */
double orc_hull_closest_point(const double* pts, int k, double* out) {
    double best = INFINITY, bp[3] = {0, 0, 0};
    for (int i = 0; i < k; i++) {
        const double* a = &pts[3 * i];
        double d = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
        if (d < best) {
            best = d;
            memcpy(bp, a, sizeof bp);
        }
    }
    for (int i = 0; i < k; i++)
        for (int j = i + 1; j < k; j++) {
            const double *a = &pts[3 * i], *b = &pts[3 * j];
            double ab[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
            double den = ab[0] * ab[0] + ab[1] * ab[1] + ab[2] * ab[2];
            if (!(den > 1e-18)) continue;
            double t = -(a[0] * ab[0] + a[1] * ab[1] + a[2] * ab[2]) / den;
            if (t < 0 || t > 1) continue;
            double p[3] = {a[0] + t * ab[0], a[1] + t * ab[1], a[2] + t * ab[2]};
            double d = sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
            if (d < best) {
                best = d;
                memcpy(bp, p, sizeof bp);
            }
        }
    for (int i = 0; i < k; i++)
        for (int j = i + 1; j < k; j++)
            for (int l = j + 1; l < k; l++) {
                const double *a = &pts[3 * i], *b = &pts[3 * j], *c = &pts[3 * l];
                double e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
                double g11 = e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2];
                double g12 = e1[0] * e2[0] + e1[1] * e2[1] + e1[2] * e2[2];
                double g22 = e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2];
                double r1 = -(a[0] * e1[0] + a[1] * e1[1] + a[2] * e1[2]), r2 = -(a[0] * e2[0] + a[1] * e2[1] + a[2] * e2[2]);
                double det = g11 * g22 - g12 * g12;
                if (!(det > 1e-14 * fmax(g11 * g22, 1e-300))) continue; /* degenerate triangle: its edges cover it */
                double u = (r1 * g22 - r2 * g12) / det, v = (r2 * g11 - r1 * g12) / det;
                if (u < 0 || v < 0 || u + v > 1) continue;
                double p[3] = {a[0] + u * e1[0] + v * e2[0], a[1] + u * e1[1] + v * e2[1], a[2] + u * e1[2] + v * e2[2]};
                double d = sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
                if (d < best) {
                    best = d;
                    memcpy(bp, p, sizeof bp);
                }
            }
    /* Origin inside the hull -> distance 0 (openGJK reports 0 once its simplex reaches 4 vertices).  If bp were the
     * closest point of a hull that does not contain the origin, every vertex would lie beyond the supporting plane
     * through bp (p_i . bp >= |bp|^2); a hull around the origin has vertices on the far side of any direction. */
    if (best > 0) {
        const double bb = bp[0] * bp[0] + bp[1] * bp[1] + bp[2] * bp[2];
        for (int i = 0; i < k; i++) {
            const double* a = &pts[3 * i];
            if (a[0] * bp[0] + a[1] * bp[1] + a[2] * bp[2] < 0.5 * bb) { /* outside: all >= bb; inside: some < 0 */
                best = 0;
                bp[0] = bp[1] = bp[2] = 0;
                break;
            }
        }
    }
    memcpy(out, bp, sizeof bp);
    return best;
}

/*
 * LSCs of one agent against one neighbour, all segments.
 *   own, obs : control points [M][6][3] (doubles holding float32 values), agent's initial trajectory and the
 *              neighbour's predicted trajectory (src/traj_planner.cpp:273-310, 399-411: both are shifted previous plans)
 *   fallback : agent.current_goal_point - obstacle.position, used when the hull contains the origin (:624-633)
 *   out      : [M][6] LSC records in the reference's order lscs[oi][m][i]
 */
void orc_generate_lsc_pair(int M, int dim, const double* own, const double* obs, double r_own, double r_obs, double dw_own,
                           double dw_obs, const double* fallback, orc_lsc* out) {
    /* :1229-1240 (both are agents) */
    const double downwash = (dim == 3) ? (dw_own * r_own + dw_obs * r_obs) / (r_own + r_obs) : 1.0;
    const float dwf = (float)downwash;
    for (int m = 0; m < M; m++) {
        double rel[18];
        float relf[18];
        for (int i = 0; i < 6; i++) {
            const double *a = &own[(m * 6 + i) * 3], *b = &obs[(m * 6 + i) * 3];
            /* coordinateTransform: z /= (float)downwash on float control points (src/trajectory.cpp:214), then the float
             * difference of :1186 */
            float az = (float)a[2], bz = (float)b[2];
            if (dim == 3) {
                az = az / dwf;
                bz = bz / dwf;
            }
            relf[3 * i + 0] = (float)a[0] - (float)b[0];
            relf[3 * i + 1] = (float)a[1] - (float)b[1];
            relf[3 * i + 2] = (dim == 3) ? az - bz : 0.0f;
            for (int k = 0; k < 3; k++) rel[3 * i + k] = (double)relf[3 * i + k]; /* point3DsToArray: float -> double */
        }
        double cp[3];
        orc_hull_closest_point(rel, 6, cp);
        /* closest_point2 is a point3d (float), normalized() in float (:1195) */
        float nf[3] = {(float)cp[0], (float)cp[1], (float)cp[2]};
        float len = sqrtf(nf[0] * nf[0] + nf[1] * nf[1] + nf[2] * nf[2]);
        if (len < 1e-5f) { /* SP_EPSILON_FLOAT, :624-633: vector from the obstacle to the agent's goal, transformed */
            nf[0] = (float)fallback[0];
            nf[1] = (float)fallback[1];
            nf[2] = (dim == 3) ? (float)fallback[2] / dwf : 0.0f;
            len = sqrtf(nf[0] * nf[0] + nf[1] * nf[1] + nf[2] * nf[2]);
        }
        if (len > 0) {
            nf[0] /= len;
            nf[1] /= len;
            nf[2] /= len;
        }
        const double collision_dist = r_obs + r_own; /* :641 */
        for (int i = 0; i < 6; i++) {
            /* :642-643: float dot product, double arithmetic around it */
            const float dot = relf[3 * i + 0] * nf[0] + relf[3 * i + 1] * nf[1] + relf[3 * i + 2] * nf[2];
            orc_lsc* L = &out[m * 6 + i];
            L->d = 0.5 * (collision_dist + (double)dot);
            L->nrm[0] = (double)nf[0];
            L->nrm[1] = (double)nf[1];
            L->nrm[2] = (dim == 3) ? (double)(float)((double)nf[2] / downwash) : 0.0; /* :653, float z / double downwash */
            for (int k = 0; k < 3; k++) L->p[k] = (double)(float)obs[(m * 6 + i) * 3 + k]; /* :654 obs control point */
        }
    }
}

/* all agents of a shard: traj [n_total][M][6][3], neighbours [n_agents][n_obs] (global ids, < 0: none -> zero rows) */
void orc_generate_lsc(int M, int dim, int n_agents, int n_obs, int first_agent, const double* traj, const int* neighbours,
                      const double* radius, const double* downwash, const double* goal, orc_lsc* out) {
#pragma omp parallel for schedule(static)
    for (int a = 0; a < n_agents; a++) {
        const int ga = first_agent + a;
        for (int o = 0; o < n_obs; o++) {
            orc_lsc* dst = &out[((size_t)a * n_obs + o) * M * 6];
            const int gb = neighbours[(size_t)a * n_obs + o];
            if (gb < 0) {
                memset(dst, 0, sizeof(orc_lsc) * M * 6);
                continue;
            }
            const double* own = &traj[(size_t)ga * M * 18];
            const double* obs = &traj[(size_t)gb * M * 18];
            double fb[3] = {goal[3 * a + 0] - obs[0], goal[3 * a + 1] - obs[1], goal[3 * a + 2] - obs[2]};
            orc_generate_lsc_pair(M, dim, own, obs, radius[ga], radius[gb], downwash[ga], downwash[gb], fb, dst);
        }
    }
}
