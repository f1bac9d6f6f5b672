// TEST INFRASTRUCTURE.  extern "C" door into the REFERENCE's own openGJK, which oracle/Makefile (target `ref`) compiles
// from /root/reference/src/openGJK/openGJK.cpp where it lies; nothing of the reference is copied into this repository.
// Mirrors the call made by closestPointsBetweenPointAndConvexHull (reference include/geometry.hpp:266-296):
// body 1 = the hull points, body 2 = the single query point, v = closest point of the hull relative to the query point.
#include <array>
#include <vector>

#include "openGJK/openGJK.hpp"

extern "C" double ref_gjk_point_hull(const double* hull, int n, const double* point, double* v_out) {
    struct bd bd1, bd2;
    struct simplex s;
    bd1.numpoints = n;
    bd1.coord.resize(n);
    for (int i = 0; i < n; i++) bd1.coord[i] = {hull[3 * i], hull[3 * i + 1], hull[3 * i + 2]};
    bd2.numpoints = 1;
    bd2.coord.resize(1);
    bd2.coord[0] = {point[0], point[1], point[2]};
    double v[3];
    const double dd = gjk(bd1, bd2, &s, v);
    v_out[0] = v[0];
    v_out[1] = v[1];
    v_out[2] = v[2];
    return dd;
}
