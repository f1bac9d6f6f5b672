// lscgen.hip — LSC generation on the device, gfx950 only: the producer of the rows the trajectory QP
// consumes, plus the shift of the previous plan that feeds it.  C ABI in include/lscqp.h.
//
// Replaces, for agent-type obstacles, TrajPlanner::generateLSC (reference src/traj_planner.cpp:611-657) with
// normalVectorBetweenPolys (:1179-1205), the closest point of closestPointsBetweenPointAndConvexHull
// (include/geometry.hpp:266-296: openGJK in the reference), downwashBetween (:1229-1240),
// Trajectory::coordinateTransform (src/trajectory.cpp:207-219) and CollisionConstraints::setLSC
// (src/collision_constraints.cpp:514-521).  Not a translation: the reference runs GJK per (neighbour, segment) on the CPU
// and stores 56-byte LSC records in nested vectors; here one lane owns one (agent, neighbour, segment) unit, finds the
// closest point of the 6-point hull by enumerating its candidate faces (the closest point of a convex hull is unique,
// so this agrees with GJK; branch-light, no iteration count) and writes the six packed 32-byte rows
// (nx, ny, nz, b = d + n.p_obs) the QP kernel stages, straight into HBM.
//
// The kernel is HBM-side work: per unit 192 B written + 144 B (neighbour's control points) read, ~2 kflop fp64.
// Rows of one wavefront are staged through LDS so that every global store instruction writes 64 x 16 contiguous bytes.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/lscqp.h"

namespace lscgen {

constexpr int kThreads = 256;

struct P3 {
    double x, y, z;
};
__device__ __forceinline__ P3 sub(P3 a, P3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ double dot(P3 a, P3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ P3 cross(P3 a, P3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ P3 axpy(P3 a, double t, P3 b) { return {a.x + t * b.x, a.y + t * b.y, a.z + t * b.z}; }

// 1/d to full fp64 precision: v_rcp_f64 + two Newton steps (5 instructions instead of an IEEE division sequence)
__device__ __forceinline__ double fast_rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;
}

// Closest point to the origin on conv{p[0..5]}: 6 vertices, 15 edges, 20 triangles, then a supporting-plane test that
// decides whether the origin is inside (distance 0).  Everything is evaluated on the Gram matrix G_ij = p_i . p_j (21
// dot products, computed once): an edge or triangle candidate then costs a dozen scalar operations instead of vector
// arithmetic, and only the winner's barycentric weights are turned back into a point.
// `planar`: all points share z = 0 (2-D missions).  The closest point of a planar hull to an origin in its plane lies on an
// edge or a vertex unless the origin is inside, which the supporting-plane test at the end detects on its own: the 20
// triangle candidates are skipped (a wave-uniform branch), the result is the same.
__device__ __forceinline__ P3 hull_closest_point(const P3 (&p)[6], bool planar = false) {
    double G[6][6];
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = i; j < 6; j++) {
            G[i][j] = dot(p[i], p[j]);
            G[j][i] = G[i][j];
        }
    double best = 1e300;  // squared distance of the best candidate a + u (b - a) + v (c - a)
    int bi = 0, bj = 0, bl = 0;
    double bu = 0, bv = 0;
    auto consider = [&](double d2, bool ok, int i, int j, int l, double u, double v) {
        const bool take = ok && d2 < best;
        best = take ? d2 : best;
        bi = take ? i : bi;
        bj = take ? j : bj;
        bl = take ? l : bl;
        bu = take ? u : bu;
        bv = take ? v : bv;
    };
#pragma unroll
    for (int i = 0; i < 6; i++) consider(G[i][i], true, i, i, i, 0.0, 0.0);
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = i + 1; j < 6; j++) {
            const double den = G[i][i] - 2.0 * G[i][j] + G[j][j];  // |b - a|^2
            const double r1 = G[i][i] - G[i][j];                   // -a . (b - a)
            const bool ok = den > 1e-18;
            const double t = r1 * fast_rcp(ok ? den : 1.0);
            consider(G[i][i] - t * r1, ok && t >= 0.0 && t <= 1.0, i, j, i, t, 0.0);
        }
    if (!planar)
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = i + 1; j < 6; j++)
#pragma unroll
            for (int l = j + 1; l < 6; l++) {
                const double g11 = G[j][j] - 2.0 * G[i][j] + G[i][i], g22 = G[l][l] - 2.0 * G[i][l] + G[i][i];
                const double g12 = G[j][l] - G[i][j] - G[i][l] + G[i][i];
                const double r1 = G[i][i] - G[i][j], r2 = G[i][i] - G[i][l];
                const double det = g11 * g22 - g12 * g12;
                const bool ok = det > 1e-14 * fmax(g11 * g22, 1e-300);  // degenerate triangle: its edges cover it
                const double idet = fast_rcp(ok ? det : 1.0);
                const double u = (r1 * g22 - r2 * g12) * idet, v = (r2 * g11 - r1 * g12) * idet;
                consider(G[i][i] - u * r1 - v * r2, ok && u >= 0.0 && v >= 0.0 && u + v <= 1.0, i, j, l, u, v);
            }
    auto pick = [&](int k) -> P3 {
        P3 r = p[0];
#pragma unroll
        for (int i = 1; i < 6; i++) {
            r.x = (k == i) ? p[i].x : r.x;
            r.y = (k == i) ? p[i].y : r.y;
            r.z = (k == i) ? p[i].z : r.z;
        }
        return r;
    };
    const P3 a = pick(bi), b = pick(bj), c = pick(bl);
    P3 bp = axpy(axpy(a, bu, sub(b, a)), bv, sub(c, a));
    const double bb = dot(bp, bp);
    // Hull around the origin -> distance 0 (what openGJK reports once its simplex has 4 vertices).  If bp were the closest
    // point of a hull that does not contain the origin, every vertex would lie beyond the supporting plane through bp
    // (p_i . bp >= |bp|^2); a hull around the origin has a vertex with p_i . bp < 0.  The threshold sits halfway.
    bool inside = false;
#pragma unroll
    for (int i = 0; i < 6; i++) inside = inside || (dot(p[i], bp) < 0.5 * bb);
    if (inside) bp = {0, 0, 0};
    return bp;
}

// ---- octomap::point3d arithmetic (3 x float) of the reference's geometry helpers, used by the CLSC / BVC generators.
// Semantics of octomath::Vector3: component arithmetic and cross() in float, dot() / norm_sq() float expressions widened
// to double, norm() = sqrt of that, distance() = sqrt of the double sum of squared float differences, normalize()
// divides by (float)norm().  No FMA contraction in these routines: the reference's x86-64 build has none, and the CPU
// oracle restates them the same way, so both sides round identically.
struct F3 {
    float x, y, z;
};
__device__ __forceinline__ F3 fsub(F3 a, F3 b) {
#pragma clang fp contract(off)
    return {a.x - b.x, a.y - b.y, a.z - b.z};
}
__device__ __forceinline__ F3 fadd(F3 a, F3 b) {
#pragma clang fp contract(off)
    return {a.x + b.x, a.y + b.y, a.z + b.z};
}
__device__ __forceinline__ F3 fscale(F3 a, float s) {
#pragma clang fp contract(off)
    return {a.x * s, a.y * s, a.z * s};
}
__device__ __forceinline__ F3 fcross(F3 a, F3 b) {
#pragma clang fp contract(off)
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ double fdot(F3 a, F3 b) {
#pragma clang fp contract(off)
    return (double)(a.x * b.x + a.y * b.y + a.z * b.z);
}
__device__ __forceinline__ double fnorm(F3 a) {
#pragma clang fp contract(off)
    return sqrt((double)(a.x * a.x + a.y * a.y + a.z * a.z));
}
__device__ __forceinline__ double fdist(F3 a, F3 b) {
#pragma clang fp contract(off)
    const double dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    return sqrt(dx * dx + dy * dy + dz * dz);
}
__device__ __forceinline__ F3 fnormalized(F3 a) {
#pragma clang fp contract(off)
    const double len = fnorm(a);
    if (len > 0) {
        const float f = (float)len;
        a.x /= f;
        a.y /= f;
        a.z /= f;
    }
    return a;
}

// closestPointsBetweenPointAndLineSegment (reference include/geometry.hpp:67-102): cp2 = closest point of [s, e] to `point`
__device__ __forceinline__ double point_segment(F3 point, F3 s, F3 e, F3& cp2) {
#pragma clang fp contract(off)
    const F3 a = fsub(s, point), b = fsub(e, point);
    double dist_min = fnorm(a);
    F3 rel = a;
    if (!(a.x == b.x && a.y == b.y && a.z == b.z)) {
        double dist = fnorm(b);
        if (dist_min > dist) {
            dist_min = dist;
            rel = b;
        }
        const F3 n_line = fnormalized(fsub(b, a));
        const F3 c = fsub(a, fscale(n_line, (float)fdot(a, n_line)));
        dist = fnorm(c);
        if (fdot(fsub(c, a), fsub(c, b)) < 0 && dist_min > dist) {
            dist_min = dist;
            rel = c;
        }
    }
    cp2 = fadd(rel, point);
    return dist_min;
}

// closestPointsBetweenLineSegments (reference include/geometry.hpp:174-263, with closestPointsBetweenLines :129-172 and
// its Eigen Matrix3f inverse written out as cofactors / determinant)
__device__ __forceinline__ double segseg_closest(F3 s1, F3 e1, F3 s2, F3 e2, F3& cp1, F3& cp2) {
#pragma clang fp contract(off)
    if (fdist(s1, e1) < 1e-5) {
        cp1 = s1;
        return point_segment(s1, s2, e2, cp2);
    }
    if (fdist(s2, e2) < 1e-5) {
        cp2 = s2;
        return point_segment(s2, s1, e1, cp1);
    }
    const F3 v1 = fsub(e1, s1), v2 = fsub(e2, s2);
    const double l1 = fnorm(v1), l2 = fnorm(v2);
    const F3 n1 = fscale(v1, (float)(1 / l1)), n2 = fscale(v2, (float)(1 / l2));
    if (fnorm(fcross(n1, n2)) < 1e-5) {  // parallel segments, :192-219
        double bound_min = fdot(fsub(s2, s1), n1), bound_max = fdot(fsub(e2, s1), n1);
        F3 p2_min = s2, p2_max = e2;
        if (bound_max < bound_min) {
            const double t = bound_min;
            bound_min = bound_max;
            bound_max = t;
            const F3 tp = p2_min;
            p2_min = p2_max;
            p2_max = tp;
        }
        F3 delta = fsub(s2, s1);
        delta = fsub(delta, fscale(n1, (float)fdot(delta, n1)));
        if (l1 < bound_min) {
            cp1 = e1;
            cp2 = p2_min;
        } else if (bound_max < 0) {
            cp1 = s1;
            cp2 = p2_max;
        } else if (bound_min < 0) {
            cp1 = s1;
            cp2 = fadd(s1, delta);
        } else {
            cp1 = fsub(p2_min, delta);
            cp2 = p2_min;
        }
        return fdist(cp1, cp2);
    }
    {  // closestPointsBetweenLines :129-172
        const F3 m1 = fnormalized(v1), m2 = fnormalized(v2);
        const F3 delta = fsub(s2, s1);
        const F3 neg2 = {-m2.x, -m2.y, -m2.z};
        if (fdist(m1, m2) < 1e-5 || fdist(m1, neg2) < 1e-5) {
            const F3 dl = fsub(delta, fscale(m1, (float)fdot(delta, m1)));
            cp1 = s1;
            cp2 = fadd(s1, dl);
        } else {
            const F3 m3 = fnormalized(fcross(m2, m1));
            // A = [m1 | -m2 | m3] (columns); x = inverse(A) * delta, inverse = cofactors^T / det, float
            const float a00 = m1.x, a01 = -m2.x, a02 = m3.x, a10 = m1.y, a11 = -m2.y, a12 = m3.y, a20 = m1.z, a21 = -m2.z, a22 = m3.z;
            const float c00 = a11 * a22 - a12 * a21, c01 = a02 * a21 - a01 * a22, c02 = a01 * a12 - a02 * a11;
            const float c10 = a12 * a20 - a10 * a22, c11 = a00 * a22 - a02 * a20, c12 = a02 * a10 - a00 * a12;
            const float det = a00 * c00 + a10 * c01 + a20 * c02;
            const float invdet = 1.0f / det;
            const float al0 = (c00 * invdet) * delta.x + (c01 * invdet) * delta.y + (c02 * invdet) * delta.z;
            const float al1 = (c10 * invdet) * delta.x + (c11 * invdet) * delta.y + (c12 * invdet) * delta.z;
            cp1 = fadd(s1, fscale(m1, al0));
            cp2 = fadd(s2, fscale(m2, al1));
        }
    }
    const double alpha1 = fdot(fsub(cp1, s1), n1) / l1, alpha2 = fdot(fsub(cp2, s2), n2) / l2;
    if (alpha1 < 0)
        cp1 = s1;
    else if (alpha1 > 1)
        cp1 = e1;
    if (alpha2 < 0)
        cp2 = s2;
    else if (alpha2 > 1)
        cp2 = e2;
    if (alpha1 < 0 || alpha1 > 1) {
        double dt = fdot(n2, fsub(cp1, s2));
        dt = dt < 0 ? 0 : (dt > l2 ? l2 : dt);
        cp2 = fadd(s2, fscale(n2, (float)dt));
    }
    if (alpha2 < 0 || alpha2 > 1) {
        double dt = fdot(n1, fsub(cp2, s1));
        dt = dt < 0 ? 0 : (dt > l1 ? l1 : dt);
        cp1 = fadd(s1, fscale(n1, (float)dt));
    }
    return fdist(cp1, cp2);
}

// one lane per (agent a, obstacle slot o, segment m); unit index t = (a*n_obs + o)*M + m, rows of unit t = out[6t .. 6t+6)
// MODE: LSCQP_GEN_LSC = generateLSC (:611-657), LSCQP_GEN_CLSC = generateCLSC (:659-706), LSCQP_GEN_BVC = generateBVC (:708-734)
// goal: current goal point of local agent a at goal[3a]; goal_all: of global agent g at goal_all[3g] (CLSC only)
template <int MODE>
__global__ __launch_bounds__(kThreads) void generate_lsc_kernel(int M, int dim, int64_t n_units, int32_t n_obs, int64_t first_agent,
                                                                const double* __restrict__ traj,
                                                                const double* __restrict__ own_traj,
                                                                const int32_t* __restrict__ neighbours,
                                                                const double* __restrict__ radius,
                                                                const double* __restrict__ downwash,
                                                                const double* __restrict__ goal,
                                                                const double* __restrict__ goal_all, int rows_f32,
                                                                int32_t n_obs_total, int32_t slot0, lscqp_row* __restrict__ out) {
    __shared__ double4 stage[kThreads * 6];  // the block's rows in output order: [lane][row], 32 B each (48 KiB)
    const int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const bool live = t < n_units;
    const int64_t tt = live ? t : 0;
    const int m = (int)(tt % M);
    const int64_t ao = tt / M;
    const int o = (int)(ao % n_obs);
    const int64_t a = ao / n_obs;
    const int64_t ga = first_agent + a;
    const int32_t gb_raw = neighbours[a * n_obs + o];
    const bool has = live && gb_raw >= 0;
    const int64_t gb = has ? gb_raw : ga;

    const double r_own = radius[ga], r_obs = radius[gb];
    double4* mine = stage + (size_t)threadIdx.x * 6;
    if constexpr (MODE != LSCQP_GEN_LSC) {
        // generateCLSC / generateBVC: float point3d arithmetic throughout, downwash never gated by the dimension (a 2-D
        // mission has equal z everywhere, so the transform cancels); generateCLSC skips the transform in 2-D (:666-672)
        const double dw = (downwash[ga] * r_own + downwash[gb] * r_obs) / (r_own + r_obs);
        const float dwf = (float)dw;
        const bool tr = (MODE == LSCQP_GEN_BVC) || dim != 2;
        const double collision_dist = r_obs + r_own;
        auto cpt = [&](int64_t g, int mm, int i) -> F3 {
            // (own_traj: the planning agent's initial trajectory kept apart from the predicted trajectories, [local agent][M][6][3])
            const double* p = (own_traj && g == ga) ? own_traj + ((a * M + mm) * 6 + i) * 3 : traj + ((g * M + mm) * 6 + i) * 3;
            return F3{(float)p[0], (float)p[1], (float)p[2]};
        };
        auto trf = [&](F3 p) -> F3 {
            if (tr) p.z = p.z / dwf;
            return p;
        };
        F3 nrm = {0, 0, 0};
        double d[6];
        F3 pob[6];
        if (MODE == LSCQP_GEN_BVC) {
            const F3 diff = fsub(trf(cpt(ga, 0, 0)), trf(cpt(gb, 0, 0)));
            nrm = fnormalized(diff);
            const double dd = 0.5 * (collision_dist + fdot(diff, nrm));
#pragma unroll
            for (int i = 0; i < 6; i++) {
                d[i] = dd;
                pob[i] = cpt(gb, m, i);
            }
        } else if (m < M - 1) {
            F3 relf[6];
            P3 rel[6];
#pragma unroll
            for (int i = 0; i < 6; i++) {
                pob[i] = cpt(gb, m, i);
                relf[i] = fsub(trf(cpt(ga, m, i)), trf(pob[i]));
                rel[i] = {(double)relf[i].x, (double)relf[i].y, (double)relf[i].z};
            }
            bool flat = dim == 2;  // generateCLSC does not transform in 2-D (:666-672): planar only if all z are equal
#pragma unroll
            for (int i = 0; i < 6; i++) flat = flat && relf[i].z == 0.0f;
            const P3 cp = hull_closest_point(rel, flat);
            nrm = fnormalized(F3{(float)cp.x, (float)cp.y, (float)cp.z});  // no fallback normal in generateCLSC: zero rows drop out
#pragma unroll
            for (int i = 0; i < 6; i++) d[i] = 0.5 * (collision_dist + fdot(relf[i], nrm));
        } else {
            // :691-703: separate the segments (last point -> goal point) of the neighbour and of the agent; the goal
            // points are used untransformed, and one obstacle point / one margin serve all control points (:532-539)
            const double* go = goal_all + 3 * gb;
            const double* gw = goal + 3 * a;
            F3 cp1, cp2;
            const double dist = segseg_closest(trf(cpt(gb, M - 1, 5)), F3{(float)go[0], (float)go[1], (float)go[2]},
                                               trf(cpt(ga, M - 1, 5)), F3{(float)gw[0], (float)gw[1], (float)gw[2]}, cp1, cp2);
            nrm = fnormalized(fsub(cp2, cp1));
#pragma unroll
            for (int i = 0; i < 6; i++) {
                d[i] = 0.5 * (collision_dist + dist);
                pob[i] = cp1;
            }
        }
        const double onx = (double)nrm.x, ony = (double)nrm.y, onz = (double)(float)((double)nrm.z / dw);
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const double b = d[i] + (onx * (double)pob[i].x + ony * (double)pob[i].y + onz * (double)pob[i].z);
            mine[i] = has ? double4{onx, ony, onz, b} : double4{0, 0, 0, 0};
        }
    } else {
    // downwashBetween, both agents (:1229-1240); 2-D missions plan in the plane
    const double dw = (dim == 3) ? (downwash[ga] * r_own + downwash[gb] * r_obs) / (r_own + r_obs) : 1.0;
    const float dwf = (float)dw;
    const double* own = own_traj ? own_traj + (a * M + m) * 18 : traj + (ga * M + m) * 18;
    const double* obs = traj + (gb * M + m) * 18;
    P3 pobs[6];
    float relf[6][3];
    P3 rel[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const double ax = own[3 * i], ay = own[3 * i + 1], az = own[3 * i + 2];
        const double bx = obs[3 * i], by = obs[3 * i + 1], bz = obs[3 * i + 2];
        pobs[i] = {(double)(float)bx, (double)(float)by, (double)(float)bz};
        // point3d arithmetic of the reference: float control points, z /= (float)downwash (src/trajectory.cpp:214),
        // float difference (:1186)
        float azf = (float)az, bzf = (float)bz;
        if (dim == 3) {
            azf = azf / dwf;
            bzf = bzf / dwf;
        }
        relf[i][0] = (float)ax - (float)bx;
        relf[i][1] = (float)ay - (float)by;
        relf[i][2] = (dim == 3) ? azf - bzf : 0.0f;
        rel[i] = {(double)relf[i][0], (double)relf[i][1], (double)relf[i][2]};
    }
    const P3 cp = hull_closest_point(rel, dim == 2);
    // closest_point2 is a point3d (float), normalized() in float (:1195); hull around the origin -> fallback (:624-633)
    float nx = (float)cp.x, ny = (float)cp.y, nz = (float)cp.z;
    float len = sqrtf(nx * nx + ny * ny + nz * nz);
    if (len < 1e-5f) {
        // agent.current_goal_point - obstacles[oi].position: the obstacle's CURRENT position for every segment (:629-631), i.e. the
        // first control point of its shifted plan, not the first control point of segment m
        const double* g = goal + 3 * a;
        const double* opos = traj + (gb * M) * 18;
        nx = (float)(g[0] - opos[0]);
        ny = (float)(g[1] - opos[1]);
        nz = (dim == 3) ? (float)(g[2] - opos[2]) / dwf : 0.0f;
        len = sqrtf(nx * nx + ny * ny + nz * nz);
    }
    if (len > 0.0f) {
        nx /= len;
        ny /= len;
        nz /= len;
    }
    const double collision_dist = r_obs + r_own;  // :641
    const double onx = (double)nx, ony = (double)ny;
    const double onz = (dim == 3) ? (double)(float)((double)nz / dw) : 0.0;  // :653
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const float dotf = relf[i][0] * nx + relf[i][1] * ny + relf[i][2] * nz;  // float dot of :642-643
        const double d = 0.5 * (collision_dist + (double)dotf);
        const double b = d + (onx * pobs[i].x + ony * pobs[i].y + onz * pobs[i].z);  // packed row: n.c >= b = d + n.p_obs
        // rows of a missing neighbour are all-zero: the solver drops normals shorter than 1e-5 (:409-411)
        mine[i] = has ? double4{onx, ony, onz, b} : double4{0, 0, 0, 0};
    }
    }  // MODE == LSCQP_GEN_LSC
    __syncthreads();
    // cooperative store: the block's rows are contiguous in `out` in exactly the staging order, so every global store
    // instruction of a wavefront writes 64 x 32 contiguous bytes
    const int64_t base_row = (int64_t)blockIdx.x * kThreads * 6;
    const int64_t n_rows = n_units * 6;
    double4* o4 = reinterpret_cast<double4*>(out);
    float4* o4f = reinterpret_cast<float4*>(out);  // LSCQP_ROWS_F32: the same rows rounded to float32, 16 B each
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const int rix = k * kThreads + threadIdx.x;  // row index within the block
        if (base_row + rix < n_rows) {
            const double4 v = stage[rix];
            // destination: the unit's rows inside an agent's block of n_obs_total obstacle slots, starting at slot0 (the
            // identity when this generator fills all slots)
            int64_t dst = base_row + rix;
            if (n_obs_total != n_obs || slot0 != 0) {
                const int64_t u = dst / 6, kk = dst % 6, mm = u % M, ao_ = u / M;
                dst = (((ao_ / n_obs) * n_obs_total + slot0 + (ao_ % n_obs)) * M + mm) * 6 + kk;
            }
            if (rows_f32)
                o4f[dst] = float4{(float)v.x, (float)v.y, (float)v.z, (float)v.w};
            else
                o4[dst] = v;
        }
    }
}

// initialTrajPlanningPrevSol (reference src/traj_planner.cpp:399-411) on the solver's output layout:
// x_prev [n][dim][M][6] fp64 (reference variable order) -> traj [n][M][6][3], segment m := previous segment m+1, last
// segment := the previous plan's last point; values rounded to float32 like desired_traj (src/traj_optimizer.cpp:71-83);
// dim == 2: z := z_2d (world_z_2d)
__global__ __launch_bounds__(kThreads) void shift_traj_kernel(int M, int dim, int64_t n, int shift, double z_2d,
                                                              const double* __restrict__ x_prev, double* __restrict__ traj) {
    const int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (t >= n * M * 6) return;
    const int i = (int)(t % 6);
    const int m = (int)((t / 6) % M);
    const int64_t a = t / (6 * M);
    // shift == 0: layout change + float32 truncation only (replanning from the same state)
    const int ms = (m + shift < M) ? m + shift : M - 1, is = (m + shift < M) ? i : 5;
    const double* x = x_prev + a * dim * M * 6;
    double* o = traj + ((a * M + m) * 6 + i) * 3;
    o[0] = (double)(float)x[(0 * M + ms) * 6 + is];
    o[1] = (double)(float)x[(1 * M + ms) * 6 + is];
    o[2] = (dim == 3) ? (double)(float)x[(2 * M + ms) * 6 + is] : (double)(float)z_2d;
}

// initialTrajPlanningPrevSol / obstaclePredictionWithPrevSol when the simulation step is SHORTER than a segment
// (multisim_time_step < dt, reference src/traj_planner.cpp:296-306, 413-421): segment 0 := prev_traj[0].subSegment(fraction, 1)
// (Segment::subSegment, src/trajectory.cpp:15-49: control points x (B A B^-1), W = B A B^-1 computed by the host in double), the other
// segments are kept.  Same layouts and float32 truncation as shift_traj_kernel.
struct SubSegW {
    double w[36];
};
__global__ __launch_bounds__(kThreads) void shift_traj_partial_kernel(int M, int dim, int64_t n, SubSegW W, double z_2d,
                                                                      const double* __restrict__ x_prev, double* __restrict__ traj) {
#pragma clang fp contract(off)
    const int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (t >= n * M * 6) return;
    const int i = (int)(t % 6);
    const int m = (int)((t / 6) % M);
    const int64_t a = t / (6 * M);
    const double* x = x_prev + a * dim * M * 6;
    double* o = traj + ((a * M + m) * 6 + i) * 3;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        double v;
        if (k == 2 && dim != 3) {
            v = (double)(float)z_2d;  // every control point of the previous plan sits at z_2d: the sub-segment too
        } else if (m == 0) {
            v = 0;
#pragma unroll
            for (int l = 0; l < 6; l++) v += (double)(float)x[(k * M + 0) * 6 + l] * W.w[l * 6 + i];
            v = (double)(float)v;
        } else {
            v = (double)(float)x[(k * M + m) * 6 + i];
        }
        o[k] = v;
    }
}

// generateLSC for NON-AGENT obstacles (reference src/traj_planner.cpp:611-657): constant-velocity prediction
// (obstaclePredictionWithPrevSol :283-285 with Trajectory::planConstVelTraj, src/trajectory.cpp:79-91), disturbance reset (:312-319),
// size prediction under bounded acceleration (:321-358), downwashBetween for a non-agent (:1235-1237), the z component of the
// relative hull dropped for tall dynamic obstacles (:1188-1191), margin d = predicted size + agent radius (:646-648), fallback
// normal (:624-633).  One lane per (agent a, obstacle slot o, segment m); rows go to slots slot0 .. slot0+n_dyn-1 of the agent's
// block of n_obs_total obstacle slots.  Compiled without FMA contraction like the rest of this file's float32 restatements.
__global__ __launch_bounds__(kThreads) void generate_lsc_obstacle_kernel(int M, int dim, double dt, lscqp_obstacle_param P, int64_t n_units,
                                                                         int32_t n_dyn, int64_t first_agent, const double* __restrict__ traj,
                                                                         const int32_t* __restrict__ ids, const lscqp_obstacle* __restrict__ table,
                                                                         const double* __restrict__ radius, const double* __restrict__ goal,
                                                                         const lscqp_header* __restrict__ hdr, int rows_f32, int32_t n_obs_total,
                                                                         int32_t slot0, const double* __restrict__ binv3, lscqp_row* __restrict__ out) {
#pragma clang fp contract(off)
    const int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (t >= n_units) return;
    const int m = (int)(t % M);
    const int64_t ao = t / M;
    const int o = (int)(ao % n_dyn);
    const int64_t a = ao / n_dyn;
    const int64_t ga = first_agent + a;
    const int32_t id = ids[a * n_dyn + o];
    const int64_t row0 = (((a * n_obs_total) + slot0 + o) * M + m) * 6;
    double4* o4 = reinterpret_cast<double4*>(out);
    float4* o4f = reinterpret_cast<float4*>(out);
    auto put = [&](int i, double4 v) {
        if (rows_f32)
            o4f[row0 + i] = float4{(float)v.x, (float)v.y, (float)v.z, (float)v.w};
        else
            o4[row0 + i] = v;
    };
    if (id < 0) {  // no obstacle in this slot: all-zero rows, which the solver drops (:409-411)
        for (int i = 0; i < 6; i++) put(i, double4{0, 0, 0, 0});
        return;
    }
    const lscqp_obstacle ob = table[id];
    const double r_own = radius[ga];
    // prediction of segment m: p + v * (float)time, time accumulated in double from 0 in steps of dt / n (planConstVelTraj).
    // checkObstacleDisturbance compares its start point with the obstacle's position: identical by construction.
    double time = 0;
    for (int s_ = 0; s_ < m * 6; s_++) time += dt / 5;
    P3 pobs[6];
    const double* own = traj + (ga * M + m) * 18;
    const double dw = (dim == 3) ? (r_own + ob.downwash * ob.radius) / (r_own + ob.radius) : 1.0;
    const float dwf = (float)dw;
    const bool flat = ob.type == 0 && ob.downwash > P.obs_downwash_threshold;
    float relf[6][3];
    P3 rel[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const float tf = (float)time;
        const float bx = (float)ob.position[0] + (float)ob.velocity[0] * tf, by = (float)ob.position[1] + (float)ob.velocity[1] * tf,
                    bz = (float)ob.position[2] + (float)ob.velocity[2] * tf;
        time += dt / 5;
        pobs[i] = {(double)bx, (double)by, (double)bz};
        float azf = (float)own[3 * i + 2], bzf = bz;
        if (dim == 3) {
            azf = azf / dwf;
            bzf = bzf / dwf;
        }
        relf[i][0] = (float)own[3 * i] - bx;
        relf[i][1] = (float)own[3 * i + 1] - by;
        relf[i][2] = (dim == 3 && !flat) ? azf - bzf : 0.0f;
        rel[i] = {(double)relf[i][0], (double)relf[i][1], (double)relf[i][2]};
    }
    const P3 cp = hull_closest_point(rel, dim == 2 || flat);
    float nx = (float)cp.x, ny = (float)cp.y, nz = (float)cp.z;
    float len = sqrtf(nx * nx + ny * ny + nz * nz);
    if (len < 1e-5f) {
        const double* g = goal + 3 * a;
        nx = (float)(g[0] - ob.position[0]);
        ny = (float)(g[1] - ob.position[1]);
        nz = (dim == 3) ? (float)(g[2] - ob.position[2]) / dwf : 0.0f;
        len = sqrtf(nx * nx + ny * ny + nz * nz);
    }
    if (len > 0.0f) {
        nx /= len;
        ny /= len;
        nz /= len;
    }
    // predicted size of segment m (obstacleSizePredictionWithConstAcc); binv3: rows 0..2 of B^-1 (host, closed form)
    double guard = 0;
    if (P.use_velocity_guard) {
        const float vx = (float)hdr[a].v0[0], vy = (float)hdr[a].v0[1], vz = (float)hdr[a].v0[2];
        guard = P.velocity_guard_ratio * (double)(vx * vx + vy * vy + vz * vz) / hdr[a].amax[0];
    }
    const int Mu = (int)((P.obs_uncertainty_horizon + 1e-9) / dt);
    const bool grow = P.obs_size_prediction && ob.type == 0;
    const double onx = (double)nx, ony = (double)ny;
    const double onz = (dim == 3) ? (double)(float)((double)nz / dw) : 0.0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        double size = ob.radius;
        if (grow) {
            if (m < Mu) {
                const double c0 = 0.5 * ob.max_acc * pow(m * dt, 2), c1 = ob.max_acc * m * dt * dt, c2 = 0.5 * ob.max_acc * pow(dt, 2);
                size = ob.radius + guard + (c0 * binv3[i] + c1 * binv3[6 + i] + c2 * binv3[12 + i]);
            } else {
                size = ob.radius + guard + 0.5 * ob.max_acc * pow(Mu * dt, 2);
            }
        }
        const double d = size + r_own;
        put(i, double4{onx, ony, onz, d + (onx * pobs[i].x + ony * pobs[i].y + onz * pobs[i].z)});
    }
}

// MultiSyncSimulator::broadcastMsgs (reference src/multi_sync_simulator.cpp:305-352): agent i receives every other agent j
// with LInfinityDistance(p_i, p_j) <= communication_range (all of them when the range is <= 0), in id order.  One
// wavefront per local agent: the lanes test 64 candidates at a time, and a ballot + prefix population count compacts the
// accepted ids in order.  The row buffers have room for n_obs neighbours per agent; when more are in range the n_obs
// nearest are kept (L-infinity distance, then id; still listed in id order) and count_out tells the caller.
__global__ __launch_bounds__(64) void select_neighbours_kernel(int64_t n_agents, int64_t first_agent, int64_t n_total, int32_t n_obs,
                                                               double range, const double* __restrict__ pos,
                                                               int32_t* __restrict__ nbr, int32_t* __restrict__ count) {
    const int64_t a = blockIdx.x;
    if (a >= n_agents) return;
    const int lane = threadIdx.x;
    const int64_t gi = first_agent + a;
    // positions are point3d (float) in the reference; LInfinityDistance widens the float differences (include/util.hpp:122-131)
    const float px = (float)pos[3 * gi], py = (float)pos[3 * gi + 1], pz = (float)pos[3 * gi + 2];
    auto dist_of = [&](int64_t j) -> double {
        const float dx = px - (float)pos[3 * j], dy = py - (float)pos[3 * j + 1], dz = pz - (float)pos[3 * j + 2];
        return fmax(fmax(fabs((double)dx), fabs((double)dy)), fabs((double)dz));
    };
    auto in_range = [&](int64_t j, double limit) -> bool {
        return j < n_total && j != gi && !(range > 0 && dist_of(j) > limit);
    };
    // pass 1: count the agents in range and keep them (distance, id; id order) in LDS while they fit
    constexpr int kCap = 1024;
    __shared__ double cand_d[kCap];
    __shared__ int32_t cand_j[kCap];
    int total = 0;
    for (int64_t base = 0; base < n_total; base += 64) {
        const int64_t j = base + lane;
        const bool valid = j < n_total && j != gi;
        const double d = valid ? dist_of(j) : 0.0;
        const bool in = valid && !(range > 0 && d > range);
        const unsigned long long m = __ballot(in);
        const int p = total + __popcll(m & ((1ull << lane) - 1ull));
        if (in && p < kCap) {
            cand_d[p] = d;
            cand_j[p] = (int32_t)j;
        }
        total += __popcll(m);
    }
    __syncthreads();
    if (total <= kCap) {
        // everything in range is in LDS.  No overflow: copy the first `total`.  Overflow: an entry stays if fewer than n_obs
        // entries precede it in (distance, id) order -- a rank count over at most kCap candidates -- and the survivors are
        // compacted in id order (the list is in id order already)
        const bool capped_l = total > n_obs && n_obs > 0;
        int written = 0;
        for (int base = 0; base < total && written < n_obs; base += 64) {
            const int e = base + lane;
            bool keep = e < total;
            if (keep && capped_l) {
                const double de = cand_d[e];
                const int32_t je = cand_j[e];
                int rank = 0;
                for (int f = 0; f < total; f++) rank += (cand_d[f] < de || (cand_d[f] == de && cand_j[f] < je)) ? 1 : 0;
                keep = rank < n_obs;
            }
            const unsigned long long m = __ballot(keep);
            const int p = written + __popcll(m & ((1ull << lane) - 1ull));
            if (keep && p < n_obs) nbr[a * n_obs + p] = cand_j[e];
            written += __popcll(m);
        }
        written = written < n_obs ? written : n_obs;
        for (int p = written + lane; p < n_obs; p += 64) nbr[a * n_obs + p] = -1;
        if (lane == 0) count[a] = total;
        return;
    }
    // more than kCap agents in range: threshold distance by bisection over all candidates (below)
    double limit = range;
    int n_strict = 0;  // overflow: keep everything nearer than `limit`, then the smallest ids at exactly `limit`
    bool capped = false;
    if (total > n_obs && n_obs > 0) {
        // the n_obs-th smallest distance, by bisection on the float-valued distances (they are exact in double)
        double lo = 0, hi = range > 0 ? range : 3.0e38;
        if (!(range > 0)) {
            double mx = 0;
            for (int64_t base = 0; base < n_total; base += 64) {
                const int64_t j = base + lane;
                double d = (j < n_total && j != gi) ? dist_of(j) : 0.0;
                for (int o = 32; o > 0; o >>= 1) d = fmax(d, __shfl_xor(d, o));
                mx = fmax(mx, d);
            }
            hi = mx;
        }
        for (int it = 0; it < 64 && lo < hi; it++) {
            const double mid = 0.5 * (lo + hi);
            if (mid <= lo || mid >= hi) break;
            int c = 0;
            for (int64_t base = 0; base < n_total; base += 64) {
                const int64_t j = base + lane;
                c += __popcll(__ballot(j < n_total && j != gi && dist_of(j) <= mid));
            }
            if (c >= n_obs)
                hi = mid;
            else
                lo = mid;
        }
        limit = hi;  // smallest distance value with at least n_obs agents at or below it
        for (int64_t base = 0; base < n_total; base += 64) {
            const int64_t j = base + lane;
            n_strict += __popcll(__ballot(j < n_total && j != gi && dist_of(j) < limit));
        }
        capped = true;
    }
    int written = 0, ties_left = capped ? n_obs - n_strict : 0;
    for (int64_t base = 0; base < n_total && written < n_obs; base += 64) {
        const int64_t j = base + lane;
        bool take;
        if (!capped) {
            take = in_range(j, range);
        } else {
            const bool valid = j < n_total && j != gi;
            const double d = valid ? dist_of(j) : 0.0;
            const bool tie = valid && d == limit;
            const unsigned long long tm = __ballot(tie);
            const int tie_rank = __popcll(tm & ((1ull << lane) - 1ull));
            take = valid && (d < limit || (tie && tie_rank < ties_left));
            const int used = __popcll(tm);
            ties_left -= used < ties_left ? used : ties_left;
        }
        const unsigned long long m = __ballot(take);
        const int p = written + __popcll(m & ((1ull << lane) - 1ull));
        if (take && p < n_obs) nbr[a * n_obs + p] = (int32_t)j;
        written += __popcll(m);
    }
    written = written < n_obs ? written : n_obs;
    for (int p = written + lane; p < n_obs; p += 64) nbr[a * n_obs + p] = -1;
    if (lane == 0) count[a] = total;
}

}  // namespace lscgen

extern "C" int lscqp_set_error_(int code, const char* msg);  // lscqp_api.hip

extern "C" int lscqp_select_neighbours_raw_(int64_t n_agents, int64_t first_agent, int64_t n_total, int32_t n_obs, double range,
                                            const double* d_pos, int32_t* d_nbr, int32_t* d_count, void* stream) {
    if (n_agents == 0) return LSCQP_OK;
    hipLaunchKernelGGL(lscgen::select_neighbours_kernel, dim3((unsigned)n_agents), dim3(64), 0, (hipStream_t)stream, n_agents, first_agent,
                       n_total, n_obs, range, d_pos, d_nbr, d_count);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return lscqp_set_error_(LSCQP_ERR_HIP, (std::string("HIP launch failed: ") + hipGetErrorString(e)).c_str());
    return LSCQP_OK;
}

extern "C" int lscqp_generate_lsc_raw_(int mode, int M, int dim, int64_t n_agents, int32_t n_obs, int64_t first_agent,
                                       const double* d_traj, const double* d_own_traj, const int32_t* d_neighbours, const double* d_radius,
                                       const double* d_downwash, const double* d_goal, const double* d_goal_all, int rows_f32,
                                       int32_t n_obs_total, int32_t slot0, lscqp_row* d_rows_out, void* stream) {
    const int64_t n_units = n_agents * (int64_t)n_obs * M;
    if (n_units == 0) return LSCQP_OK;
    const unsigned blocks = (unsigned)((n_units + lscgen::kThreads - 1) / lscgen::kThreads);
    auto* kern = mode == LSCQP_GEN_CLSC  ? lscgen::generate_lsc_kernel<LSCQP_GEN_CLSC>
                 : mode == LSCQP_GEN_BVC ? lscgen::generate_lsc_kernel<LSCQP_GEN_BVC>
                                         : lscgen::generate_lsc_kernel<LSCQP_GEN_LSC>;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(lscgen::kThreads), 0, (hipStream_t)stream, M, dim, n_units, n_obs, first_agent, d_traj,
                       d_own_traj, d_neighbours, d_radius, d_downwash, d_goal, d_goal_all, rows_f32, n_obs_total, slot0, d_rows_out);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return lscqp_set_error_(LSCQP_ERR_HIP, (std::string("HIP launch failed: ") + hipGetErrorString(e)).c_str());
    return LSCQP_OK;
}

extern "C" int lscqp_shift_traj_partial_raw_(int M, int dim, int64_t n, const double* w36, double z_2d, const double* d_x_prev, double* d_traj,
                                             void* stream) {
    const int64_t total = n * M * 6;
    if (total == 0) return LSCQP_OK;
    lscgen::SubSegW W;
    for (int i = 0; i < 36; i++) W.w[i] = w36[i];
    const unsigned blocks = (unsigned)((total + lscgen::kThreads - 1) / lscgen::kThreads);
    hipLaunchKernelGGL(lscgen::shift_traj_partial_kernel, dim3(blocks), dim3(lscgen::kThreads), 0, (hipStream_t)stream, M, dim, n, W, z_2d,
                       d_x_prev, d_traj);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return lscqp_set_error_(LSCQP_ERR_HIP, (std::string("HIP launch failed: ") + hipGetErrorString(e)).c_str());
    return LSCQP_OK;
}

extern "C" int lscqp_generate_lsc_obstacles_raw_(int M, int dim, double dt, const lscqp_obstacle_param* p, int64_t n_agents, int32_t n_dyn,
                                                 int64_t first_agent, const double* d_traj, const int32_t* d_ids, const lscqp_obstacle* d_table,
                                                 const double* d_radius, const double* d_goal, const lscqp_header* d_hdr, int rows_f32,
                                                 int32_t n_obs_total, int32_t slot0, const double* d_binv3, lscqp_row* d_rows_out, void* stream) {
    const int64_t n_units = n_agents * (int64_t)n_dyn * M;
    if (n_units == 0) return LSCQP_OK;
    const unsigned blocks = (unsigned)((n_units + lscgen::kThreads - 1) / lscgen::kThreads);
    hipLaunchKernelGGL(lscgen::generate_lsc_obstacle_kernel, dim3(blocks), dim3(lscgen::kThreads), 0, (hipStream_t)stream, M, dim, dt, *p, n_units,
                       n_dyn, first_agent, d_traj, d_ids, d_table, d_radius, d_goal, d_hdr, rows_f32, n_obs_total, slot0, d_binv3, d_rows_out);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return lscqp_set_error_(LSCQP_ERR_HIP, (std::string("HIP launch failed: ") + hipGetErrorString(e)).c_str());
    return LSCQP_OK;
}

extern "C" int lscqp_shift_traj_raw_(int M, int dim, int64_t n, int shift, double z_2d, const double* d_x_prev, double* d_traj,
                                     void* stream) {
    const int64_t total = n * M * 6;
    if (total == 0) return LSCQP_OK;
    const unsigned blocks = (unsigned)((total + lscgen::kThreads - 1) / lscgen::kThreads);
    hipLaunchKernelGGL(lscgen::shift_traj_kernel, dim3(blocks), dim3(lscgen::kThreads), 0, (hipStream_t)stream, M, dim, n, shift, z_2d,
                       d_x_prev, d_traj);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return lscqp_set_error_(LSCQP_ERR_HIP, (std::string("HIP launch failed: ") + hipGetErrorString(e)).c_str());
    return LSCQP_OK;
}
