/*
 * lscmode_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's OTHER linear-constraint generators for agent-type obstacles — the producers of the
 * QP's rows in the planner modes that do not go through generateLSC (SURVEY.md §8f-1 names generateLSC/generateCLSC):
 *
 *   mode 1  TrajPlanner::generateCLSC   src/traj_planner.cpp:659-706  — what constructLSC() (:551-553) runs for the
 *           reference's DEFAULT launch (mode/planner = lsc, mode/goal = grid_based_planner, launch/simulation.launch:44-45):
 *           segments m < M-1 as generateLSC but WITHOUT the fallback normal (a hull around the origin leaves a zero
 *           normal, and TrajOptimizer drops such rows, src/traj_optimizer.cpp:409-411); the last segment separates the
 *           two line segments (last point -> goal point) of the neighbour and of the agent
 *           (closestPointsBetweenLineSegments, include/geometry.hpp:174-263, with closestPointsBetweenPointAndLineSegment
 *           :67-102 and closestPointsBetweenLines :129-172) and stores ONE obstacle point and ONE margin for all
 *           control points (CollisionConstraints::setLSC, src/collision_constraints.cpp:532-539).
 *   mode 2  TrajPlanner::generateBVC    src/traj_planner.cpp:708-734  — buffered Voronoi cell: one normal per obstacle
 *           from the two start points, the same margin for every control point of every segment.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this file.
 *
 * float32 semantics: the reference computes all of this on octomap::point3d (octomath::Vector3, 3 x float).  octomap
 * (a third-party dependency, find_package(octomap) in the reference's CMakeLists.txt, headers absent from the checkout)
 * defines, in octomap/math/Vector3.h: operator-, operator+, operator*(float), cross in float; dot(), norm_sq() as float
 * expressions returned as double; norm() = sqrt(norm_sq()); distance() = sqrt of the double sum of squared float
 * differences; normalize() divides by (float)norm() when the norm is > 0; operator== compares components exactly.
 * Those semantics are restated in the helpers below.  The 3x3 solve of closestPointsBetweenLines is Eigen's
 * Matrix3f::inverse() (cofactors / determinant, float) times the right-hand side.  The file is compiled without
 * floating-point contraction (the reference's x86-64 build has no FMA).
 * PARITY PINNING: octomap and Eigen are absent, so geometry.hpp cannot be compiled here; tests/test_lscmode.py pins the
 * segment-segment routine against an independent double-precision solution (exact clamped closest points, checked by
 * scipy's bounded minimiser; tests/golden/segseg.json) on the configurations where the reference's procedure is exact,
 * and bounds it (feasible pair on both segments, distance >= the true distance) everywhere else.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "lscqp_oracle.h"

typedef struct {
    float x, y, z;
} v3f;

static v3f v_sub(v3f a, v3f b) { return (v3f){a.x - b.x, a.y - b.y, a.z - b.z}; }
static v3f v_add(v3f a, v3f b) { return (v3f){a.x + b.x, a.y + b.y, a.z + b.z}; }
static v3f v_scale(v3f a, float s) { return (v3f){a.x * s, a.y * s, a.z * s}; }
static v3f v_neg(v3f a) { return (v3f){-a.x, -a.y, -a.z}; }
static v3f v_cross(v3f a, v3f b) { return (v3f){a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static double v_dot(v3f a, v3f b) { return (double)(a.x * b.x + a.y * b.y + a.z * b.z); }
static double v_norm(v3f a) { return sqrt((double)(a.x * a.x + a.y * a.y + a.z * a.z)); }
static double v_dist(v3f a, v3f b) {
    const double dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z; /* float differences widened */
    return sqrt(dx * dx + dy * dy + dz * dz);
}
static v3f v_normalized(v3f a) {
    const double len = v_norm(a);
    if (len > 0) {
        const float f = (float)len;
        a.x /= f;
        a.y /= f;
        a.z /= f;
    }
    return a;
}
static int v_eq(v3f a, v3f b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

/* include/geometry.hpp:67-102: cp1 = point, cp2 = closest point of the segment */
static double point_segment(v3f point, v3f s, v3f e, v3f* cp1, v3f* cp2) {
    const v3f a = v_sub(s, point), b = v_sub(e, point);
    double dist_min = v_norm(a);
    v3f rel = a;
    if (!v_eq(a, b)) {
        double dist = v_norm(b);
        if (dist_min > dist) {
            dist_min = dist;
            rel = b;
        }
        const v3f n_line = v_normalized(v_sub(b, a));
        const v3f c = v_sub(a, v_scale(n_line, (float)v_dot(a, n_line)));
        dist = v_norm(c);
        if (v_dot(v_sub(c, a), v_sub(c, b)) < 0 && dist_min > dist) {
            dist_min = dist;
            rel = c;
        }
    }
    *cp1 = point;
    *cp2 = v_add(rel, point);
    return dist_min;
}

/* Eigen Matrix3f::inverse() * b: cofactor matrix over the determinant, all float */
static void solve3f(const float A[3][3], const float b[3], float x[3]) {
    float c[3][3];
    c[0][0] = A[1][1] * A[2][2] - A[1][2] * A[2][1];
    c[0][1] = A[0][2] * A[2][1] - A[0][1] * A[2][2];
    c[0][2] = A[0][1] * A[1][2] - A[0][2] * A[1][1];
    c[1][0] = A[1][2] * A[2][0] - A[1][0] * A[2][2];
    c[1][1] = A[0][0] * A[2][2] - A[0][2] * A[2][0];
    c[1][2] = A[0][2] * A[1][0] - A[0][0] * A[1][2];
    c[2][0] = A[1][0] * A[2][1] - A[1][1] * A[2][0];
    c[2][1] = A[0][1] * A[2][0] - A[0][0] * A[2][1];
    c[2][2] = A[0][0] * A[1][1] - A[0][1] * A[1][0];
    const float det = A[0][0] * c[0][0] + A[1][0] * c[0][1] + A[2][0] * c[0][2];
    const float invdet = 1.0f / det;
    for (int i = 0; i < 3; i++) x[i] = (c[i][0] * invdet) * b[0] + (c[i][1] * invdet) * b[1] + (c[i][2] * invdet) * b[2];
}

/* include/geometry.hpp:129-172 (the callers guarantee non-degenerate lines) */
static double line_line(v3f s1, v3f e1, v3f s2, v3f e2, v3f* cp1, v3f* cp2) {
    const v3f n1 = v_normalized(v_sub(e1, s1)), n2 = v_normalized(v_sub(e2, s2));
    if (v_dist(n1, n2) < 1e-5 || v_dist(n1, v_neg(n2)) < 1e-5) {
        v3f delta = v_sub(s2, s1);
        delta = v_sub(delta, v_scale(n1, (float)v_dot(delta, n1)));
        *cp1 = s1;
        *cp2 = v_add(s1, delta);
        return v_norm(delta);
    }
    const v3f delta = v_sub(s2, s1);
    const v3f n3 = v_normalized(v_cross(n2, n1));
    const float A[3][3] = {{n1.x, -n2.x, n3.x}, {n1.y, -n2.y, n3.y}, {n1.z, -n2.z, n3.z}};
    const float b[3] = {delta.x, delta.y, delta.z};
    float al[3];
    solve3f(A, b, al);
    *cp1 = v_add(s1, v_scale(n1, al[0]));
    *cp2 = v_add(s2, v_scale(n2, al[1]));
    return fabs((double)al[2]);
}

/* include/geometry.hpp:174-263.  Inputs and outputs are float32 values held in doubles. */
double orc_segseg_closest(const double* l1s, const double* l1e, const double* l2s, const double* l2e, double* cp1_out,
                          double* cp2_out) {
    const v3f s1 = {(float)l1s[0], (float)l1s[1], (float)l1s[2]}, e1 = {(float)l1e[0], (float)l1e[1], (float)l1e[2]};
    const v3f s2 = {(float)l2s[0], (float)l2s[1], (float)l2s[2]}, e2 = {(float)l2e[0], (float)l2e[1], (float)l2e[2]};
    v3f cp1, cp2;
    double dist;
    if (v_dist(s1, e1) < 1e-5) {
        dist = point_segment(s1, s2, e2, &cp1, &cp2);
    } else if (v_dist(s2, e2) < 1e-5) {
        dist = point_segment(s2, s1, e1, &cp2, &cp1); /* :179-180: computed from line2's point, then swapped */
    } else {
        const v3f v1 = v_sub(e1, s1), v2 = v_sub(e2, s2);
        const double l1 = v_norm(v1), l2 = v_norm(v2);
        const v3f n1 = v_scale(v1, (float)(1 / l1)), n2 = v_scale(v2, (float)(1 / l2));
        if (v_norm(v_cross(n1, n2)) < 1e-5) { /* parallel segments, :192-219 */
            double bound_min = v_dot(v_sub(s2, s1), n1), bound_max = v_dot(v_sub(e2, s1), n1);
            v3f p2_min = s2, p2_max = e2;
            if (bound_max < bound_min) {
                const double t = bound_min;
                bound_min = bound_max;
                bound_max = t;
                const v3f tp = p2_min;
                p2_min = p2_max;
                p2_max = tp;
            }
            v3f delta = v_sub(s2, s1);
            delta = v_sub(delta, v_scale(n1, (float)v_dot(delta, n1)));
            if (l1 < bound_min) {
                cp1 = e1;
                cp2 = p2_min;
            } else if (bound_max < 0) {
                cp1 = s1;
                cp2 = p2_max;
            } else if (bound_min < 0) {
                cp1 = s1;
                cp2 = v_add(s1, delta);
            } else {
                cp1 = v_sub(p2_min, delta);
                cp2 = p2_min;
            }
            dist = v_dist(cp1, cp2);
        } else { /* :220-259 */
            line_line(s1, e1, s2, e2, &cp1, &cp2);
            const double alpha1 = v_dot(v_sub(cp1, s1), n1) / l1, alpha2 = v_dot(v_sub(cp2, s2), n2) / l2;
            if (alpha1 < 0)
                cp1 = s1;
            else if (alpha1 > 1)
                cp1 = e1;
            if (alpha2 < 0)
                cp2 = s2;
            else if (alpha2 > 1)
                cp2 = e2;
            if (alpha1 < 0 || alpha1 > 1) {
                double dot = v_dot(n2, v_sub(cp1, s2));
                if (dot < 0)
                    dot = 0;
                else if (dot > l2)
                    dot = l2;
                cp2 = v_add(s2, v_scale(n2, (float)dot));
            }
            if (alpha2 < 0 || alpha2 > 1) {
                double dot = v_dot(n1, v_sub(cp2, s1));
                if (dot < 0)
                    dot = 0;
                else if (dot > l1)
                    dot = l1;
                cp1 = v_add(s1, v_scale(n1, (float)dot));
            }
            dist = v_dist(cp1, cp2);
        }
    }
    cp1_out[0] = cp1.x, cp1_out[1] = cp1.y, cp1_out[2] = cp1.z;
    cp2_out[0] = cp2.x, cp2_out[1] = cp2.y, cp2_out[2] = cp2.z;
    return dist;
}

static v3f cp_of(const double* traj, int m, int i) {
    const double* p = &traj[(m * 6 + i) * 3];
    return (v3f){(float)p[0], (float)p[1], (float)p[2]};
}

/* Trajectory::coordinateTransform (src/trajectory.cpp:207-219): z /= (float)downwash.  generateCLSC skips it for 2-D
 * worlds (:666-672); generateLSC / generateBVC always apply it, with downwash = 1 ... unless the agents' downwash differs */
static v3f transf(v3f p, int apply, float dwf) {
    if (apply) p.z = p.z / dwf;
    return p;
}

/*
 * Constraints of one agent against one neighbour, all segments.
 *   mode 1 (CLSC) / 2 (BVC); own, obs: control points [M][6][3]; goal_own = agent.current_goal_point,
 *   goal_obs = obstacles[oi].goal_point (the neighbour's current goal point, as broadcast).
 */
void orc_generate_mode_pair(int mode, int M, int dim, const double* own, const double* obs, double r_own, double r_obs,
                            double dw_own, double dw_obs, const double* goal_own, const double* goal_obs, orc_lsc* out) {
    const double downwash = (dw_own * r_own + dw_obs * r_obs) / (r_own + r_obs); /* :1229-1240, both are agents */
    const float dwf = (float)downwash;
    const double collision_dist = r_obs + r_own;
    const int tr = (mode == 1) ? (dim != 2) : 1;
    if (mode == 2) { /* generateBVC :708-734 */
        const v3f diff = v_sub(transf(cp_of(own, 0, 0), tr, dwf), transf(cp_of(obs, 0, 0), tr, dwf));
        v3f nrm = v_normalized(diff);
        const double d = 0.5 * (collision_dist + v_dot(diff, nrm));
        nrm.z = (float)((double)nrm.z / downwash);
        for (int m = 0; m < M; m++)
            for (int i = 0; i < 6; i++) {
                orc_lsc* L = &out[m * 6 + i];
                const v3f p = cp_of(obs, m, i);
                L->d = d;
                L->nrm[0] = nrm.x, L->nrm[1] = nrm.y, L->nrm[2] = nrm.z;
                L->p[0] = p.x, L->p[1] = p.y, L->p[2] = p.z;
            }
        return;
    }
    for (int m = 0; m < M; m++) {
        if (m < M - 1) { /* :677-690 */
            v3f relf[6];
            double rel[18], cp[3];
            for (int i = 0; i < 6; i++) {
                relf[i] = v_sub(transf(cp_of(own, m, i), tr, dwf), transf(cp_of(obs, m, i), tr, dwf));
                rel[3 * i] = relf[i].x, rel[3 * i + 1] = relf[i].y, rel[3 * i + 2] = relf[i].z;
            }
            orc_hull_closest_point(rel, 6, cp);
            v3f nrm = v_normalized((v3f){(float)cp[0], (float)cp[1], (float)cp[2]}); /* no fallback in generateCLSC */
            const float nz_out = (float)((double)nrm.z / downwash);
            for (int i = 0; i < 6; i++) {
                orc_lsc* L = &out[m * 6 + i];
                const v3f p = cp_of(obs, m, i);
                L->d = 0.5 * (collision_dist + v_dot(relf[i], nrm));
                L->nrm[0] = nrm.x, L->nrm[1] = nrm.y, L->nrm[2] = nz_out;
                L->p[0] = p.x, L->p[1] = p.y, L->p[2] = p.z;
            }
        } else { /* :691-703: the goal points are NOT transformed by the reference */
            const v3f o_last = transf(cp_of(obs, M - 1, 5), tr, dwf), a_last = transf(cp_of(own, M - 1, 5), tr, dwf);
            const double l1s[3] = {o_last.x, o_last.y, o_last.z}, l2s[3] = {a_last.x, a_last.y, a_last.z};
            const double l1e[3] = {(float)goal_obs[0], (float)goal_obs[1], (float)goal_obs[2]};
            const double l2e[3] = {(float)goal_own[0], (float)goal_own[1], (float)goal_own[2]};
            double c1[3], c2[3];
            const double dist = orc_segseg_closest(l1s, l1e, l2s, l2e, c1, c2);
            const v3f p1 = {(float)c1[0], (float)c1[1], (float)c1[2]}, p2 = {(float)c2[0], (float)c2[1], (float)c2[2]};
            v3f nrm = v_normalized(v_sub(p2, p1));
            const double d = 0.5 * (collision_dist + dist);
            nrm.z = (float)((double)nrm.z / downwash);
            for (int i = 0; i < 6; i++) {
                orc_lsc* L = &out[m * 6 + i];
                L->d = d;
                L->nrm[0] = nrm.x, L->nrm[1] = nrm.y, L->nrm[2] = nrm.z;
                L->p[0] = p1.x, L->p[1] = p1.y, L->p[2] = p1.z;
            }
        }
    }
}

/* all agents of a shard: traj [n_total][M][6][3], goal_all [n_total][3], neighbours [n_agents][n_obs] (< 0: zero rows) */
void orc_generate_mode(int mode, int M, int dim, int n_agents, int n_obs, int first_agent, const double* traj,
                       const int* neighbours, const double* radius, const double* downwash, const double* goal_all, orc_lsc* out) {
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
            orc_generate_mode_pair(mode, M, dim, &traj[(size_t)ga * M * 18], &traj[(size_t)gb * M * 18], radius[ga], radius[gb],
                                   downwash[ga], downwash[gb], &goal_all[3 * (size_t)ga], &goal_all[3 * (size_t)gb], dst);
        }
    }
}

/*
 * MultiSyncSimulator::broadcastMsgs' range filter (reference src/multi_sync_simulator.cpp:318-333): the obstacles of agent i
 * are the other agents with LInfinityDistance (include/util.hpp:122-131, float differences widened) <= range (all if range <= 0),
 * in id order.  nbr [n_agents][n_obs] (-1 padded), count [n_agents] = how many were in range.  Capacity rule of the build (the
 * reference's list is unbounded): when more than n_obs are in range the n_obs nearest are kept, ties to the smaller id.
 */
void orc_select_neighbours(int n_agents, int first, int n_total, int n_obs, double range, const double* pos, int* nbr, int* count) {
    for (int a = 0; a < n_agents; a++) {
        const int gi = first + a;
        double* d = (double*)malloc(sizeof(double) * (size_t)n_total);
        int total = 0;
        for (int j = 0; j < n_total; j++) {
            double m = 0;
            for (int k = 0; k < 3; k++) {
                const float df = (float)pos[3 * gi + k] - (float)pos[3 * j + k];
                const double ad = fabs((double)df);
                if (m < ad) m = ad;
            }
            d[j] = (j == gi || (range > 0 && m > range)) ? -1.0 : m; /* -1: not a neighbour */
            if (d[j] >= 0) total++;
        }
        count[a] = total;
        if (total > n_obs) { /* drop the farthest (largest id among equals) until n_obs remain */
            for (int drop = total - n_obs; drop > 0; drop--) {
                int worst = -1;
                for (int j = 0; j < n_total; j++)
                    if (d[j] >= 0 && (worst < 0 || d[j] >= d[worst])) worst = j;
                d[worst] = -1.0;
            }
        }
        int w = 0;
        for (int j = 0; j < n_total && w < n_obs; j++)
            if (d[j] >= 0) nbr[(size_t)a * n_obs + w++] = j;
        for (; w < n_obs; w++) nbr[(size_t)a * n_obs + w] = -1;
        free(d);
    }
}
