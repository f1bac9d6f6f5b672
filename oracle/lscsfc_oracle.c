/*
 * lscsfc_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's safe-flight-corridor construction, the producer of the boxes the
 * trajectory QP and the goal LP consume.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * What it follows (paths relative to the reference checkout):
 *   MapManager::updateOctreeFromCSV              src/map_manager.cpp:262-305   (world CSV boxes -> occupied cells)
 *   CollisionConstraints::isObstacleInSFC        src/collision_constraints.cpp:777-808
 *   CollisionConstraints::isSFCInBoundary        :810-817
 *   CollisionConstraints::expandSFC              :819-881 (fixed axis order), :883-946 (goal-ordered, setAxisCand :1134-1170)
 *   expandSFCFromPoint / expandSFCFromConvexHull :666-690, :692-722, :724-775
 *   initializeSFC / constructSFCFromPoint / constructSFCFromConvexHull   :366-384, :396-412, :414-436
 *   Box::isPointInBox / include / intersection / closestPoint / isSuperSetOfConvexHull   :81-88, :176-178, :190-197, :199-210, :135-150
 *
 * The distance map.  The reference asks DynamicEDTOctomap (dynamicEDT3D, a third-party library that ships with octomap;
 * neither is in the checkout or in this image) for the occupied cell nearest to the voxel that contains a sample point:
 * an exact Euclidean distance transform between cell centres, cut at maxdist = 1.0 m (src/map_manager.cpp:13-14,74-76).
 * It is restated here from its published definition by brute force: for every voxel the occupied cell with the smallest
 * squared centre distance (<= (maxdist/res)^2); ties, which dynamicEDT3D resolves by its propagation order, are resolved
 * by the first candidate in (dz, dy, |dx| with negative first) order.  PARITY UNPINNED at such ties; the construction as a
 * whole is pinned through the reference's own run: the corridor of forest10's agent 1 must put its -x face at 2.55, the
 * value that reproduces the reference's result log to its printed digits (tests/golden/kat_log.json).
 * Kept from the reference: when no occupied cell lies within maxdist (or the sample is outside the map) its query leaves
 * `closest_point` at its default (0,0,0) and the sample is measured against a phantom cell at the world origin (:796-800).
 *
 * float32 semantics: boxes and points are octomap::point3d (3 x float); the statements below keep float where the
 * reference stores a point3d component and double where it computes in double.  Compiled without FMA contraction.
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#include "lscqp_oracle.h"

/* ---- map ------------------------------------------------------------------------------------------------------ */
static int key_of(double coord, double res) { return (int)floor((1.0 / res) * coord); } /* OcTreeBaseImpl::coordToKey */

orc_map* orc_map_create(const double* boxes, int n_boxes, const double* world_min, const double* world_max, double res,
                        double max_dist) {
    orc_map* mp = (orc_map*)calloc(1, sizeof(orc_map));
    mp->res = res;
    for (int k = 0; k < 3; k++) {
        mp->world_min[k] = (float)world_min[k];
        mp->world_max[k] = (float)world_max[k];
        mp->key0[k] = key_of((double)mp->world_min[k], res); /* DynamicEDTOctomap bounding box: keys of bbxMin .. bbxMax */
        mp->dims[k] = key_of((double)mp->world_max[k], res) - mp->key0[k] + 1;
    }
    const size_t nvox = (size_t)mp->dims[0] * mp->dims[1] * mp->dims[2];
    mp->occ = (unsigned char*)calloc(nvox, 1);
    mp->nearest = (int*)malloc(nvox * sizeof(int));
    /* updateOctreeFromCSV: cells i in [round((c - s/2)/res), round((c + s/2)/res)), point (i + 0.5) res.  The CSV values
     * pass through point3d (float) before the arithmetic (:276-283). */
    for (int b = 0; b < n_boxes; b++) {
        int lo[3], hi[3];
        for (int k = 0; k < 3; k++) {
            const float com = (float)boxes[6 * b + k], size = (float)boxes[6 * b + 3 + k];
            lo[k] = (int)round((com - 0.5 * size) / res);
            hi[k] = (int)round((com + 0.5 * size) / res);
        }
        for (int i = lo[0]; i < hi[0]; i++)
            for (int j = lo[1]; j < hi[1]; j++)
                for (int k = lo[2]; k < hi[2]; k++) {
                    /* the inserted point is (i + 0.5) res as float; its key is i except where float rounding of the
                     * product crosses a cell border, which (x + 0.5) * 0.1 never does for |x| < 1e5 */
                    const int x = i - mp->key0[0], y = j - mp->key0[1], z = k - mp->key0[2];
                    if (x < 0 || y < 0 || z < 0 || x >= mp->dims[0] || y >= mp->dims[1] || z >= mp->dims[2]) continue;
                    mp->occ[((size_t)z * mp->dims[1] + y) * mp->dims[0] + x] = 1;
                }
    }
    const int R = (int)floor(max_dist / res + 1e-9);
    const int R2 = R * R;
    mp->radius_cells = R;
#pragma omp parallel for schedule(dynamic, 4)
    for (int z = 0; z < mp->dims[2]; z++)
        for (int y = 0; y < mp->dims[1]; y++)
            for (int x = 0; x < mp->dims[0]; x++) {
                int best = R2 + 1, bx = 0, by = 0, bz = 0, found = 0;
                for (int dz = -R; dz <= R; dz++) {
                    const int zz = z + dz;
                    if (zz < 0 || zz >= mp->dims[2]) continue;
                    for (int dy = -R; dy <= R; dy++) {
                        const int yy = y + dy;
                        if (yy < 0 || yy >= mp->dims[1] || dz * dz + dy * dy >= best) continue;
                        const unsigned char* row = &mp->occ[((size_t)zz * mp->dims[1] + yy) * mp->dims[0]];
                        for (int ad = 0; ad <= R; ad++) {
                            const int d2 = dz * dz + dy * dy + ad * ad;
                            if (d2 >= best) break;
                            int hit = 0, dx = 0;
                            if (x - ad >= 0 && row[x - ad]) {
                                hit = 1;
                                dx = -ad;
                            } else if (ad > 0 && x + ad < mp->dims[0] && row[x + ad]) {
                                hit = 1;
                                dx = ad;
                            }
                            if (hit) {
                                best = d2;
                                bx = dx, by = dy, bz = dz;
                                found = 1;
                                break;
                            }
                        }
                    }
                }
                mp->nearest[((size_t)z * mp->dims[1] + y) * mp->dims[0] + x] =
                    found ? (((bx + 128) & 255) | (((by + 128) & 255) << 8) | (((bz + 128) & 255) << 16) | (1 << 24)) : 0;
            }
    return mp;
}

void orc_map_destroy(orc_map* mp) {
    if (!mp) return;
    free(mp->occ);
    free(mp->nearest);
    free(mp);
}

void orc_map_info(const orc_map* mp, int* dims, int* key0) {
    for (int k = 0; k < 3; k++) dims[k] = mp->dims[k], key0[k] = mp->key0[k];
}
const unsigned char* orc_map_occ(const orc_map* mp) { return mp->occ; }
const int* orc_map_nearest(const orc_map* mp) { return mp->nearest; }

/* ---- isObstacleInSFC (:777-808) -------------------------------------------------------------------------------- */
typedef struct {
    float lo[3], hi[3];
} boxf;

static int obstacle_in(const orc_map* mp, const boxf* b, double margin) {
    const double res = mp->res;
    const float delta = (float)(0.5 * res);
    int n[3];
    for (int k = 0; k < 3; k++) n[k] = (int)floor(((double)(b->hi[k] - b->lo[k]) + 1e-5) / res) + 1;
    for (int i0 = 0; i0 < n[0]; i0++)
        for (int i1 = 0; i1 < n[1]; i1++)
            for (int i2 = 0; i2 < n[2]; i2++) {
                const int it[3] = {i0, i1, i2};
                float p[3];
                int v[3], inside = 1;
                for (int k = 0; k < 3; k++) {
                    p[k] = (float)((double)b->lo[k] + (double)it[k] * res); /* search_point(i) = box_min(i) + iter * res */
                    v[k] = key_of((double)p[k], res) - mp->key0[k];          /* worldToMap */
                    if (v[k] < 0 || v[k] >= mp->dims[k]) inside = 0;
                }
                /* getDistanceAndClosestObstacle leaves `closest_point` untouched when the sample lies outside the distance map or no
                 * occupied cell is within max_dist; the caller's point3d is default-constructed, so the reference then measures
                 * against a cell at the WORLD ORIGIN (:796-800).  Reproduced: near the origin corridors are cut short by it. */
                const int code = inside ? mp->nearest[((size_t)v[2] * mp->dims[1] + v[1]) * mp->dims[0] + v[0]] : 0;
                const int have = (code >> 24) != 0;
                const int off[3] = {(code & 255) - 128, ((code >> 8) & 255) - 128, ((code >> 16) & 255) - 128};
                double dist = 0;
                for (int k = 0; k < 3; k++) {
                    /* mapToWorld / keyToCoord: cell centre (key + 0.5) res as float; cell box = centre -+ delta (floats) */
                    const float c = have ? (float)(((double)(v[k] + off[k] + mp->key0[k]) + 0.5) * res) : 0.0f;
                    const float cmin = c - delta, cmax = c + delta;
                    const float q = p[k] < cmin ? cmin : (p[k] > cmax ? cmax : p[k]); /* Box::closestPoint */
                    const double dk = fabs((double)(q - p[k]));                        /* LInfinityDistance, float difference */
                    if (dist < dk) dist = dk;
                }
                if (dist < margin + 1e-5) return 1;
            }
    return 0;
}

static int in_boundary(const orc_map* mp, const boxf* b, double margin) { /* :810-817 */
    for (int k = 0; k < 3; k++) {
        if (!((double)b->lo[k] > (double)mp->world_min[k] + margin - 1e-5)) return 0;
        if (!((double)b->hi[k] < (double)mp->world_max[k] - margin + 1e-5)) return 0;
    }
    return 1;
}

/* setAxisCand (:1134-1170) */
static void axis_order(const boxf* b, const float* goal, int* cand) {
    float delta[3];
    int offsets[3];
    double values[3];
    for (int k = 0; k < 3; k++) {
        const float mid = (b->lo[k] + b->hi[k]) * 0.5f;
        delta[k] = goal[k] - mid;
        offsets[k] = delta[k] > 0 ? 3 : 0;
        values[k] = fabs((double)delta[k]);
    }
    int order[3], cnt = 0;
    double max_value = -1, min_value = 1e9; /* SP_INFINITY */
    for (int i = 0; i < 3; i++) {
        int pos;
        if (values[i] > max_value) {
            pos = 0;
            max_value = values[i];
        } else if (values[i] < min_value) {
            pos = cnt;
            min_value = values[i];
        } else {
            pos = 1;
        }
        for (int j = cnt; j > pos; j--) order[j] = order[j - 1];
        order[pos] = i;
        cnt++;
    }
    for (int i = 0; i < 3; i++) {
        cand[i] = order[i] + offsets[order[i]];
        cand[5 - i] = order[i] + (3 - offsets[order[i]]);
    }
}

/* expandSFC (:819-881 with goal == NULL, :883-946 otherwise) */
static int expand_sfc(const orc_map* mp, const boxf* initial, const float* goal, double margin, boxf* out) {
    if (obstacle_in(mp, initial, margin)) return 0;
    int cand[6] = {0, 1, 2, 3, 4, 5}, ncand = 6;
    if (goal) axis_order(initial, goal, cand);
    const double res = mp->res;
    boxf sfc = *initial, sfc_cand, sfc_update;
    int i = -1, axis;
    while (ncand > 0) {
        sfc_cand = sfc;
        sfc_update = sfc;
        while (in_boundary(mp, &sfc_update, 0) && !obstacle_in(mp, &sfc_update, margin)) {
            i++;
            if (i >= ncand) i = 0;
            axis = cand[i];
            sfc = sfc_cand;
            sfc_update = sfc_cand;
            if (axis < 3) {
                sfc_update.hi[axis] = sfc_cand.lo[axis];
                sfc_cand.lo[axis] = (float)((double)sfc_cand.lo[axis] - res);
                sfc_update.lo[axis] = sfc_cand.lo[axis];
            } else {
                sfc_update.lo[axis - 3] = sfc_cand.hi[axis - 3];
                sfc_cand.hi[axis - 3] = (float)((double)sfc_cand.hi[axis - 3] + res);
                sfc_update.hi[axis - 3] = sfc_cand.hi[axis - 3];
            }
        }
        /* the first pass can leave the loop before any axis was drawn (i == -1): vector::erase(begin() - 1) is undefined
         * in the reference; only reachable when the initial box is already outside the boundary -> treated as failure */
        if (i < 0) return 0;
        for (int j = i; j + 1 < ncand; j++) cand[j] = cand[j + 1];
        ncand--;
        if (i > 0)
            i--;
        else
            i = ncand - 1;
    }
    const double delta = margin - ((int)(margin / res) * res); /* SFC margin compensation, :868-877 */
    for (int k = 0; k < 3; k++) {
        if ((double)sfc.lo[k] > (double)mp->world_min[k] + 1e-5) sfc.lo[k] = (float)((double)sfc.lo[k] - delta);
        if ((double)sfc.hi[k] < (double)mp->world_max[k] - 1e-5) sfc.hi[k] = (float)((double)sfc.hi[k] + delta);
    }
    *out = sfc;
    return 1;
}

static int point_in(const boxf* b, const float* p) { /* Box::isPointInBox :81-88 */
    for (int k = 0; k < 3; k++)
        if (!((double)p[k] > (double)b->lo[k] - 1e-5 && (double)p[k] < (double)b->hi[k] + 1e-5)) return 0;
    return 1;
}

static int superset_of(const boxf* b, const float (*pts)[3], int n) { /* :135-150 */
    for (int k = 0; k < 3; k++) {
        float mn = pts[0][k], mx = pts[0][k];
        for (int i = 1; i < n; i++) {
            if (pts[i][k] < mn) mn = pts[i][k];
            if (pts[i][k] > mx) mx = pts[i][k];
        }
        if ((double)mn < (double)b->lo[k] - 1e-5 || (double)mx > (double)b->hi[k] + 1e-5) return 0;
    }
    return 1;
}

static void clip_to_prev(const boxf* prev, boxf* ini, double res) { /* :677-685, :762-770 */
    if (!(point_in(prev, ini->lo) && point_in(prev, ini->hi))) { /* not prev_sfc.include(initial_sfc) */
        for (int k = 0; k < 3; k++) {
            const float lo = prev->lo[k] > ini->lo[k] ? prev->lo[k] : ini->lo[k];
            const float hi = prev->hi[k] < ini->hi[k] ? prev->hi[k] : ini->hi[k];
            ini->lo[k] = (float)(ceil(((double)lo - 1e-5) / res) * res);
            ini->hi[k] = (float)(floor(((double)hi + 1e-5) / res) * res);
        }
    }
}

/*
 * One agent's corridor update.  sfc: [M] boxes (float32 values in doubles), updated in place.
 *   mode 0  initializeSFC(position = pts[0])                                       :366-384   (status 0 = the reference throws)
 *   mode 1  constructSFCFromConvexHull({pts[0] = last point, pts[1] = goal}, next_waypoint = pts[2])   :414-436
 *   mode 2  constructSFCFromPoint(point = pts[0], goal_point = pts[1])             :396-412
 * Returns 1 = a new box was found, 0 = the previous box was kept (modes 1, 2) / failure (mode 0).
 */
int orc_construct_sfc(const orc_map* mp, int mode, int M, const double* pts, double radius, orc_box* sfc) {
    const double res = mp->res;
    float P[3][3];
    for (int i = 0; i < 3; i++)
        for (int k = 0; k < 3; k++) P[i][k] = (float)pts[3 * i + k];
    boxf ini, out, prev;
    int ok;
    if (mode == 0) {
        for (int k = 0; k < 3; k++) {
            ini.lo[k] = (float)(floor((double)P[0][k] / res) * res);
            ini.hi[k] = (float)(ceil((double)P[0][k] / res) * res);
        }
        ok = expand_sfc(mp, &ini, NULL, radius, &out);
        if (!ok) return 0;
        for (int m = 0; m < M; m++)
            for (int k = 0; k < 3; k++) sfc[m].bmin[k] = out.lo[k], sfc[m].bmax[k] = out.hi[k];
        return 1;
    }
    for (int m = 0; m < M - 1; m++) sfc[m] = sfc[m + 1];
    for (int k = 0; k < 3; k++) prev.lo[k] = (float)sfc[M - 1].bmin[k], prev.hi[k] = (float)sfc[M - 1].bmax[k];
    if (mode == 1) {
        /* greedy hull with the next waypoint, aligned by round() (:692-722) */
        for (int k = 0; k < 3; k++) {
            float mn = P[0][k], mx = P[0][k];
            for (int i = 1; i < 3; i++) {
                if (P[i][k] < mn) mn = P[i][k];
                if (P[i][k] > mx) mx = P[i][k];
            }
            ini.lo[k] = (float)(round((double)mn / res) * res);
            ini.hi[k] = (float)(round((double)mx / res) * res);
        }
        ok = expand_sfc(mp, &ini, NULL, radius, &out);
        if (ok && !superset_of(&out, (const float(*)[3])P, 3)) ok = 0;
        if (!ok) { /* the hull alone inside the previous box, aligned outwards (:724-775) */
            for (int k = 0; k < 3; k++) {
                const float mn = P[0][k] < P[1][k] ? P[0][k] : P[1][k], mx = P[0][k] > P[1][k] ? P[0][k] : P[1][k];
                ini.lo[k] = (float)(floor((double)mn / res) * res);
                ini.hi[k] = (float)(ceil((double)mx / res) * res);
            }
            clip_to_prev(&prev, &ini, res);
            ok = expand_sfc(mp, &ini, NULL, radius, &out);
        }
    } else { /* expandSFCFromPoint :666-690 */
        for (int k = 0; k < 3; k++) {
            ini.lo[k] = (float)(floor((double)P[0][k] / res) * res);
            ini.hi[k] = (float)(ceil((double)P[0][k] / res) * res);
        }
        clip_to_prev(&prev, &ini, res);
        ok = expand_sfc(mp, &ini, P[1], radius, &out);
    }
    if (ok)
        for (int k = 0; k < 3; k++) sfc[M - 1].bmin[k] = out.lo[k], sfc[M - 1].bmax[k] = out.hi[k];
    return ok;
}

void orc_construct_sfc_batch(const orc_map* mp, int mode, int M, int n, const double* pts, const double* radius, orc_box* sfc,
                             int* status) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int a = 0; a < n; a++) status[a] = orc_construct_sfc(mp, mode, M, &pts[9 * (size_t)a], radius[a], &sfc[(size_t)a * M]);
}
