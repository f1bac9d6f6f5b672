// lscsfc.hip — safe-flight-corridor construction on the device, gfx950 only: the producer of the boxes
// (lscqp_box) the trajectory QP and the goal LP consume.  C ABI in include/lscqp.h.
//
// Replaces, on a build-owned voxel map (no octomap, no dynamicEDT3D),
//   MapManager::updateOctreeFromCSV                reference src/map_manager.cpp:262-305 (world CSV boxes -> occupied cells)
//   DynamicEDTOctomap (maxdist 1.0 m, :13-14,74-76)  the nearest occupied cell of every voxel
//   CollisionConstraints::isObstacleInSFC / isSFCInBoundary / expandSFC (both orders) / setAxisCand
//                                                  src/collision_constraints.cpp:777-946, 1134-1170
//   expandSFCFromPoint / expandSFCFromConvexHull   :666-775
//   initializeSFC / constructSFCFromPoint / constructSFCFromConvexHull   :366-436
// Not a translation: the reference grows one box per agent on the CPU, asking an octree-backed distance map point by
// point.  Here the map is a dense voxel grid in HBM (1 B occupancy + 4 B nearest-cell code per voxel; 288 GB holds
// kilometre-scale worlds at 0.1 m), its nearest-cell field is built by three separable passes (exact Euclidean, 3 x (2R+1)
// reads per voxel instead of (2R+1)^3), and one workgroup (16 wavefronts) owns one agent's corridor: the box state is uniform
// across the group; the next 12 tests of the expansion loop are known in advance and are evaluated together, a lane per column of
// sample points (obstacle_in_batch below), and the first test that fails decides how far the box has grown.
//
// Arithmetic: boxes and points are octomap::point3d (float) in the reference; every statement below keeps float where
// the reference stores a point3d component and double where it computes in double, and the file is compiled without
// FMA contraction, so box coordinates are reproducible bit for bit on any IEEE machine.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <type_traits>
#include <vector>

#include "../../include/lscqp.h"
#include "lscqp_staging.hpp"

#pragma clang fp contract(off)

extern "C" int lscqp_set_error_(int code, const char* msg);  // lscqp_api.hip

struct lscqp_map_s {
    double res;
    float world_min[3], world_max[3];
    int key0[3], dims[3];
    int radius_cells;
    uint8_t* d_occ;
    int32_t* d_nearest;
    int32_t* d_sat = nullptr;  // lscqp_map_prepare: summed-area table of the cells that are NOT provably free for agents up to sat_margin
    double sat_margin = 0;
    // staging of the host-pointer corridor call: a pinned buffer + device mirror + private stream per concurrent call (the map
    // handle is shared by all agents' CollisionConstraints: two threads must never share a staging buffer)
    lscqp::StagePool* pool = nullptr;
    int device = 0;           // the device the grids live on: a plan on another device must not be handed this map
    uint64_t generation = 0;  // bumped whenever the free-space table is rebuilt: captured graphs hold its margin by value
};
#ifndef LSCSFC_VARIANT_ONLY
extern "C" int lscqp_map_device_(lscqp_map mp) { return mp->device; }
extern "C" uint64_t lscqp_map_generation_(lscqp_map mp) { return mp->generation; }
#endif

namespace lscsfc {

// The map as the corridor kernel sees it.  Scalar members with selecting accessors, NOT arrays: one dynamically indexed array
// member makes the compiler keep the whole by-value kernel argument in scratch memory, and every `world_min[k]` of the expansion
// loop becomes a memory round trip (measured: 2 500 cycles per expansion step, half of the kernel).
struct MapView {
    double res;
    float wmin0, wmin1, wmin2, wmax0, wmax1, wmax2;
    int key00, key01, key02, dims0, dims1, dims2;
    const int32_t* nearest;
    const int32_t* sat;  // (NULL: lscqp_map_prepare was not called)
    double sat_margin;
    __host__ __device__ __forceinline__ float world_min(int k) const { return k == 0 ? wmin0 : (k == 1 ? wmin1 : wmin2); }
    __host__ __device__ __forceinline__ float world_max(int k) const { return k == 0 ? wmax0 : (k == 1 ? wmax1 : wmax2); }
    __host__ __device__ __forceinline__ int key0(int k) const { return k == 0 ? key00 : (k == 1 ? key01 : key02); }
    __host__ __device__ __forceinline__ int dims(int k) const { return k == 0 ? dims0 : (k == 1 ? dims1 : dims2); }
};

__host__ __device__ inline int key_of(double coord, double res) { return (int)floor((1.0 / res) * coord); }  // coordToKey

// ---- map construction ----------------------------------------------------------------------------------------------
// one block per world box: its cells [round((c - s/2)/res), round((c + s/2)/res)) per axis (updateOctreeFromCSV)
__global__ void rasterise_kernel(const double* __restrict__ boxes, double res, int kx0, int ky0, int kz0, int nx, int ny, int nz,
                                 uint8_t* __restrict__ occ) {
    const double* b = boxes + 6 * (int64_t)blockIdx.x;
    int lo[3], hi[3];
    for (int k = 0; k < 3; k++) {
        const float com = (float)b[k], size = (float)b[3 + k];
        lo[k] = (int)round((com - 0.5 * size) / res);
        hi[k] = (int)round((com + 0.5 * size) / res);
    }
    const int ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
    if (ex <= 0 || ey <= 0 || ez <= 0) return;
    const int64_t total = (int64_t)ex * ey * ez;
    for (int64_t t = threadIdx.x; t < total; t += blockDim.x) {
        const int x = lo[0] + (int)(t % ex) - kx0, y = lo[1] + (int)((t / ex) % ey) - ky0, z = lo[2] + (int)(t / ((int64_t)ex * ey)) - kz0;
        if (x < 0 || y < 0 || z < 0 || x >= nx || y >= ny || z >= nz) continue;
        occ[((int64_t)z * ny + y) * nx + x] = 1;
    }
}

// ---- "provably free" summary of the map (lscqp_map_prepare) ------------------------------------------------------------------
// A test of the expansion (isObstacleInSFC, reference src/collision_constraints.cpp:779-808) asks whether ANY sample point of a box
// lies within margin of the nearest occupied cell of the cell it falls into.  For a cell v, every sample p that maps to v lies in
// v's closed interval per axis (p -> key is a floor), so its distance to the box of the nearest occupied cell v + off is at least
// (max_k |off_k| - 1) res; without an occupied cell within max_dist the reference measures against a cell at the world origin
// (:796-800), at least the gap between v's interval and [-res/2, res/2] away.  A cell whose bound exceeds margin + 1e-5 by more than
// 1e-4 + 4 ulp of the world's size (float rounding of the sample, the centre and the half cell) can not make any test fail.  The table holds
// the 3-D inclusive prefix sums of the OTHER cells: a box whose samples all fall into the map and whose cell range sums to zero
// passes without a single sample being evaluated -- in open space that is every test, whole-box re-tests of a million points
// included; every other box goes through the exact evaluation as before, so the boxes stay bit for bit the reference's.
__global__ void classify_free_kernel(int nx, int ny, int nz, int kx0, int ky0, int kz0, double res, double margin, double slop,
                                     const int32_t* __restrict__ nearest, int32_t* __restrict__ notfree) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nvox = (int64_t)nx * ny * nz;
    if (v >= nvox) return;
    const int x = (int)(v % nx), y = (int)((v / nx) % ny), z = (int)(v / ((int64_t)nx * ny));
    const int code = nearest[v];
    const double need = margin + 1e-5 + slop;
    double bound;
    if ((code >> 24) != 0) {
        int mo = 0;
        for (int k = 0; k < 3; k++) {
            const int off = ((code >> (8 * k)) & 255) - 128;
            const int ao = off < 0 ? -off : off;
            mo = ao > mo ? ao : mo;
        }
        bound = (double)(mo - 1) * res;
    } else {
        const int key[3] = {x + kx0, y + ky0, z + kz0};
        bound = 0;
        for (int k = 0; k < 3; k++) {
            const double a = (double)key[k] * res, b = (double)(key[k] + 1) * res;  // the cell's interval
            const double gap = a - 0.5 * res > 0 ? a - 0.5 * res : (-0.5 * res - b > 0 ? -0.5 * res - b : 0.0);
            bound = gap > bound ? gap : bound;
        }
    }
    notfree[v] = bound >= need ? 0 : 1;
}

// inclusive prefix sums along one axis: a thread per line (stride: elements between neighbours of the line)
__global__ void prefix_axis_kernel(int64_t n_lines, int len, int64_t stride, int64_t line_a, int64_t mul_a, int64_t mul_b, int32_t* __restrict__ t) {
    const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= n_lines) return;
    int32_t* p = t + (l % line_a) * mul_a + (l / line_a) * mul_b;
    int32_t run = 0;
    for (int i = 0; i < len; i++) {
        run += p[(int64_t)i * stride];
        p[(int64_t)i * stride] = run;
    }
}

constexpr int kNone = 127;  // "no occupied cell within R in this row / plane"

// pass X: nearest occupied cell of the voxel's own row, |dx| <= R, the negative side first on ties
__global__ void nearest_x_kernel(int nx, int64_t nvox, int R, const uint8_t* __restrict__ occ, int8_t* __restrict__ dxo) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nvox) return;
    const int x = (int)(v % nx);
    const uint8_t* row = occ + (v - x);
    int best = kNone;
    for (int ad = 0; ad <= R; ad++) {
        if (x - ad >= 0 && row[x - ad]) {
            best = -ad;
            break;
        }
        if (ad > 0 && x + ad < nx && row[x + ad]) {
            best = ad;
            break;
        }
    }
    dxo[v] = (int8_t)best;
}

// pass Y: over the rows y + dy of the voxel's plane, ascending dy, strict improvement
__global__ void nearest_y_kernel(int nx, int ny, int64_t nvox, int R, const int8_t* __restrict__ dxi, int8_t* __restrict__ dxo,
                                 int8_t* __restrict__ dyo) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nvox) return;
    const int y = (int)((v / nx) % ny);
    int best = R * R + 1, bx = kNone, by = 0;
    for (int dy = -R; dy <= R; dy++) {
        const int yy = y + dy;
        if (yy < 0 || yy >= ny) continue;
        const int dx = dxi[v + (int64_t)dy * nx];
        if (dx == kNone) continue;
        const int d2 = dx * dx + dy * dy;
        if (d2 < best) {
            best = d2;
            bx = dx;
            by = dy;
        }
    }
    dxo[v] = (int8_t)bx;
    dyo[v] = (int8_t)by;
}

// pass Z: over the planes z + dz, ascending dz, strict improvement; writes the packed code
__global__ void nearest_z_kernel(int nx, int ny, int nz, int64_t nvox, int R, const int8_t* __restrict__ dxi,
                                 const int8_t* __restrict__ dyi, int32_t* __restrict__ out) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nvox) return;
    const int64_t plane = (int64_t)nx * ny;
    const int z = (int)(v / plane);
    int best = R * R + 1, bx = 0, by = 0, bz = 0, found = 0;
    for (int dz = -R; dz <= R; dz++) {
        const int zz = z + dz;
        if (zz < 0 || zz >= nz) continue;
        const int dx = dxi[v + dz * plane];
        if (dx == kNone) continue;
        const int dy = dyi[v + dz * plane];
        const int d2 = dx * dx + dy * dy + dz * dz;
        if (d2 < best) {
            best = d2;
            bx = dx, by = dy, bz = dz;
            found = 1;
        }
    }
    out[v] = found ? (((bx + 128) & 255) | (((by + 128) & 255) << 8) | (((bz + 128) & 255) << 16) | (1 << 24)) : 0;
}

// ---- corridor construction: one workgroup per agent -------------------------------------------------------------------
#ifndef LSCSFC_THREADS
#define LSCSFC_THREADS 1024
#endif
constexpr int kSfcThreads = LSCSFC_THREADS;  // one workgroup (16 wavefronts) per agent.  The batch test is a chain of LDS and map round trips per
                                             // wavefront, so more wavefronts per CU is what pays: 256 / 512 / 1024 threads: 335 / 204 / 171 us per launch
                                             // (before the look-ahead boxes were assembled by one wavefront: 131 us now, with 12 tests per batch and 4 loads per group)
                                             // (forest10, 10 agents; the 1024-thread build spills 172 B per lane and still wins)

struct BoxF {
    float lo[3], hi[3];
};

// true: none of the map cells [x0, x1] x [y0, y1] x [z0, z1] (inside the map) can make a test fail (lscqp_map_prepare's table)
__device__ __forceinline__ bool cells_free(const MapView& mp, int x0, int x1, int y0, int y1, int z0, int z1) {
    const int64_t sy = mp.dims(0), sz = (int64_t)mp.dims(0) * mp.dims(1);
    auto at = [&](int x, int y, int z) -> int64_t {  // prefix sum up to (x, y, z) inclusive; -1 on any axis: empty
        return (x < 0 || y < 0 || z < 0) ? 0 : (int64_t)mp.sat[x + y * sy + z * sz];
    };
    const int xa = x0 - 1, ya = y0 - 1, za = z0 - 1;
    const int64_t cnt = at(x1, y1, z1) - at(xa, y1, z1) - at(x1, ya, z1) - at(x1, y1, za) + at(xa, ya, z1) + at(xa, y1, za) + at(x1, ya, za) - at(xa, ya, za);
    return cnt == 0;
}

// a coordinate so far from the world origin that a sample measured against the origin's phantom cell can not be within margin of it
__device__ __forceinline__ bool far_from_origin(const MapView& mp, float p) { return (double)fabsf(p) - 0.5 * mp.res >= mp.sat_margin + 1e-3; }
// ... a whole range of samples [first, last] on one axis
__device__ __forceinline__ bool range_far_from_origin(const MapView& mp, float first, float last) {
    const double need = mp.sat_margin + 1e-3 + 0.5 * mp.res;
    return (double)first >= need || -(double)last >= need;
}
// vtab entry of such a sample beyond the map (the exact path sees "negative: outside", the table-driven paths "outside but harmless")
constexpr int kBeyondFree = -2;

// the cell range of n table entries from `tab` along one axis for the table-driven tests: 0 = not provable (a sample outside the map
// that could be near the origin), 1 = [v0, v1], 2 = every entry is beyond the map and harmless (nothing to ask)
__device__ __forceinline__ int cell_range(const int* vtab, int tab, int n, int& v0, int& v1) {
    v0 = vtab[tab];
    v1 = vtab[tab + n - 1];
    if (v1 == kBeyondFree) {  // (a face on the world's far boundary: key = dims)
        if (n == 1) return 2;
        n--;
        v1 = vtab[tab + n - 1];
    }
    if (v0 == kBeyondFree) {  // (a face on the near boundary whose float coordinate has drifted a hair below it: key = -1)
        if (n == 1) return 2;
        v0 = vtab[tab + 1];
    }
    return (v0 >= 0 && v1 >= v0) ? 1 : 0;
}

// true: no sample point of the box (n[k] points from lo[k] in steps of res) can be within margin of an obstacle -- see lscqp_map_prepare
__device__ __forceinline__ bool surely_free(const MapView& mp, double margin, float lo0, float lo1, float lo2, int n0, int n1, int n2) {
    if (mp.sat == nullptr || !(margin <= mp.sat_margin)) return false;
    const double res = mp.res;
    int a[3], b[3];
    const float lo[3] = {lo0, lo1, lo2};
    const int n[3] = {n0, n1, n2};
    bool inside = true;
    // a sample outside the distance map is measured against the phantom cell at the world origin: harmless if ONE of its coordinates is
    // far from 0 -- its own (far_from_origin), or any axis on which the whole box is (e.g. a box on the floor z = 0, away from x = y = 0)
    bool any_far = false;
#pragma unroll
    for (int k = 0; k < 3; k++) any_far = any_far || range_far_from_origin(mp, lo[k], (float)((double)lo[k] + (double)(n[k] - 1) * res));
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float pf = lo[k], pl = (float)((double)lo[k] + (double)(n[k] - 1) * res);  // first and last sample, as the tests form them
        a[k] = key_of((double)pf, res) - mp.key0(k);
        b[k] = key_of((double)pl, res) - mp.key0(k);
        if (b[k] >= mp.dims(k)) {
            // the last sample lies beyond the distance map (a box face on the world's far boundary: key = dims): such a sample is
            // measured against the cell at the world origin whatever its other coordinates are, and this coordinate alone keeps it
            // out of reach if it is far enough from 0
            if (!(any_far || far_from_origin(mp, pl))) return false;
            if (n[k] == 1) return true;  // (every sample of the box is one of those)
            b[k] = key_of((double)(float)((double)lo[k] + (double)(n[k] - 2) * res), res) - mp.key0(k);
        }
        if (a[k] < 0) {  // ... or the first one, on the near boundary (a float face a hair below it)
            if (!(any_far || far_from_origin(mp, pf))) return false;
            if (n[k] == 1) return true;
            a[k] = key_of((double)(float)((double)lo[k] + res), res) - mp.key0(k);
        }
        inside = inside && a[k] >= 0 && b[k] < mp.dims(k) && a[k] <= b[k];
    }
    if (!inside) return false;  // (a sample outside the distance map is measured against the origin cell: the exact path handles it)
    return cells_free(mp, a[0], b[0], a[1], b[1], a[2], b[2]);
}

// isObstacleInSFC (:777-808): the block's lanes take the sample points of the box in turn and vote
__device__ bool obstacle_in(const MapView& mp, const BoxF& b, double margin) {
    const double res = mp.res;
    const float delta = (float)(0.5 * res);
    int n[3];
    for (int k = 0; k < 3; k++) n[k] = (int)floor(((double)(b.hi[k] - b.lo[k]) + 1e-5) / res) + 1;
    // an inverted box (a hull clipped to a previous box it does not touch) has no sample points
    const int64_t total = (n[0] <= 0 || n[1] <= 0 || n[2] <= 0) ? 0 : (int64_t)n[0] * n[1] * n[2];
    if (total > 0 && surely_free(mp, margin, b.lo[0], b.lo[1], b.lo[2], n[0], n[1], n[2])) return false;  // (block-uniform)
    const int lane = threadIdx.x;
    // kU chunks of 64 points per vote: the kU nearest-cell loads of a lane are independent and in flight together, which is
    // what bounds a large slab (the loop is a chain of dependent HBM / L2 reads otherwise)
    constexpr int kU = 2;
    // 32-bit index arithmetic (a slab of a kilometre-scale world still has < 2^31 sample points; beyond that: 64-bit)
    const bool small = total <= 0x7fffffffLL;
    const uint32_t n2u = (uint32_t)n[2], n1u = (uint32_t)n[1];
    const int64_t n12 = (int64_t)n[1] * n[2];
    for (int64_t base = 0; base < total; base += kSfcThreads * kU) {
        float p[kU][3];
        int v[kU][3], code[kU];
        bool live[kU];
#pragma unroll
        for (int u = 0; u < kU; u++) {
            const int64_t idx = base + u * kSfcThreads + lane;
            bool inside = idx < total;
            live[u] = inside;
            const int64_t ii = inside ? idx : 0;
            int it[3];
            if (small) {
                const uint32_t iu = (uint32_t)ii, r = iu / n2u;
                it[2] = (int)(iu - r * n2u);
                it[0] = (int)(r / n1u);
                it[1] = (int)(r - (uint32_t)it[0] * n1u);
            } else {
                it[0] = (int)(ii / n12), it[1] = (int)((ii / n[2]) % n[1]), it[2] = (int)(ii % n[2]);
            }
            for (int k = 0; k < 3; k++) {
                p[u][k] = (float)((double)b.lo[k] + (double)it[k] * res);  // search_point(i) = box_min(i) + iter * res
                v[u][k] = key_of((double)p[u][k], res) - mp.key0(k);       // worldToMap
                inside = inside && v[u][k] >= 0 && v[u][k] < mp.dims(k);
            }
            code[u] = inside ? mp.nearest[((int64_t)v[u][2] * mp.dims(1) + v[u][1]) * mp.dims(0) + v[u][0]] : 0;
        }
        bool hit = false;
#pragma unroll
        for (int u = 0; u < kU; u++) {
            if (live[u]) {
                // no occupied cell within max_dist, or a sample outside the distance map: the reference's closest_point stays
                // default-constructed and it measures against a cell at the WORLD ORIGIN (:796-800) -- reproduced
                const bool have = (code[u] >> 24) != 0;
                const int off[3] = {(code[u] & 255) - 128, ((code[u] >> 8) & 255) - 128, ((code[u] >> 16) & 255) - 128};
                double dist = 0;
                for (int k = 0; k < 3; k++) {
                    // keyToCoord: cell centre (key + 0.5) res as float; closest point of the cell box; L-infinity distance
                    const float c = have ? (float)(((double)(v[u][k] + off[k] + mp.key0(k)) + 0.5) * res) : 0.0f;
                    const float cmin = c - delta, cmax = c + delta;
                    const float q = p[u][k] < cmin ? cmin : (p[u][k] > cmax ? cmax : p[u][k]);
                    const double dk = fabs((double)(q - p[u][k]));
                    dist = dist < dk ? dk : dist;
                }
                hit = hit || (dist < margin + 1e-5);
            }
        }
        if (__syncthreads_or(hit)) return true;  // block-uniform: every wavefront holds the same box state
    }
    return false;
}

__device__ bool in_boundary(const MapView& mp, const BoxF& b, double margin) {  // :810-817
    bool ok = true;
    for (int k = 0; k < 3; k++) {
        ok = ok && ((double)b.lo[k] > (double)mp.world_min(k) + margin - 1e-5);
        ok = ok && ((double)b.hi[k] < (double)mp.world_max(k) - margin + 1e-5);
    }
    return ok;
}

__device__ void axis_order(const BoxF& b, const float* goal, int* cand) {  // setAxisCand :1134-1170
    int offsets[3], order[3], cnt = 0;
    double values[3];
    for (int k = 0; k < 3; k++) {
        const float mid = (b.lo[k] + b.hi[k]) * 0.5f;
        const float d = goal[k] - mid;
        offsets[k] = d > 0 ? 3 : 0;
        values[k] = fabs((double)d);
    }
    double max_value = -1, min_value = 1e9;
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

// floor(x / res) without the fp64 division where that is provably the same number: q = x * (1 / res) differs from x / res by a few
// ulp of q, so unless q lies within 1e-9 of an integer both have the same floor; otherwise the division is carried out.
__device__ __forceinline__ double floor_div(double x, double res, double rinv) {
    const double q = x * rinv, f = floor(q);
    if (q - f < 1e-9 || f + 1.0 - q < 1e-9 || !(fabs(q) < 1e6)) return floor(x / res);
    return f;
}

// One step of the reference's inner loop body (:838-857 / :903-922): accept the candidate, then grow it by one cell along
// cand[i] and make the new layer the box to test.  Indexed by a loop so the boxes stay in registers.
__device__ __forceinline__ void grow(BoxF& sfc, BoxF& sfc_cand, BoxF& sfc_update, int axis, double res) {
    sfc = sfc_cand;
    sfc_update = sfc_cand;
    for (int k = 0; k < 3; k++) {
        if (axis == k) {
            sfc_update.hi[k] = sfc_cand.lo[k];
            sfc_cand.lo[k] = (float)((double)sfc_cand.lo[k] - res);
            sfc_update.lo[k] = sfc_cand.lo[k];
        }
        if (axis == k + 3) {
            sfc_update.lo[k] = sfc_cand.hi[k];
            sfc_cand.hi[k] = (float)((double)sfc_cand.hi[k] + res);
            sfc_update.hi[k] = sfc_cand.hi[k];
        }
    }
}

// The boxes of the next kAhead tests of the expansion loop, assuming each passes, are known in advance (pure arithmetic on the
// box state): they are tested TOGETHER, and the first test that fails decides how far the state advances.  Within a batch
//   * a sample point's coordinate, its map key and the centre of a map cell depend on ONE axis each, so they are tabulated per
//     (box, axis) / per axis in LDS by the reference's own expressions (a few hundred evaluations per batch instead of three
//     fp64 chains per sample point);
//   * a lane owns a COLUMN of a box -- the sample points along the box's thinnest axis -- so that the index arithmetic is paid
//     once per column and the column's nearest-cell loads are independent and in flight together; the columns of all boxes form
//     one index space, padded per box to whole wavefronts (a wavefront works on one box, its parameters are wave-uniform), x
//     running fastest where x is not the thin axis (consecutive lanes read consecutive words of the map);
//   * one barrier per batch instead of one per 512 points.
// Every sample point is classified by exactly the comparisons the sequential loop makes (the distance is a maximum of float
// differences, compared in double against margin + 1e-5), so the outcome is bit for bit the reference's; only the order of
// evaluation differs.  Forest10 world, 10 agents per launch (about 100 tests and 200-450 k sample points per corridor): 315 us as a
// chain of dependent 512-point rounds with the fp64 chains per point -> 131 us; 64 agents in a synthetic forest 552 -> 208 us.
// Measured and dropped: the corridor's part of the map staged in LDS (the rounds are bound by the CU's LDS pipe and by
// instruction issue, not by the map reads: slower), integer quick verdicts before the exact comparison (slower).
#ifdef LSCSFC_DEBUG
__device__ unsigned long long sfc_dbg[16];
#define SFC_DBG(i, v) do { if (threadIdx.x == 0) atomicAdd(&sfc_dbg[i], (unsigned long long)(v)); } while (0)
#else
#define SFC_DBG(i, v) do { } while (0)
#endif
#ifndef LSCSFC_GROUP
#define LSCSFC_GROUP 4
#endif
#ifndef LSCSFC_AHEAD
#define LSCSFC_AHEAD 63  // boxes of a batch (<= 63: lane j of wavefront 0 assembles box j)
#endif
#ifndef LSCSFC_SAMPLED
#define LSCSFC_SAMPLED 12  // ... of which at most this many have to be sampled (boxes the free-space table passes cost nothing)
#endif
constexpr int kAhead = LSCSFC_AHEAD;
#ifndef LSCSFC_TAB
#define LSCSFC_TAB 3072
#endif
#ifndef LSCSFC_CELL
#define LSCSFC_CELL 1024
#endif
#ifndef LSCSFC_TODO
#define LSCSFC_TODO 8192
#endif
constexpr int kTab = LSCSFC_TAB;    // entries of the per-(box, axis) tables of a batch
constexpr int kCell = LSCSFC_CELL;  // largest map extent (cells per axis) with the cell-centre table in LDS
constexpr int kTodo = LSCSFC_TODO;  // chunks of a batch the filter pass can list
struct Ahead {
    float lo[kAhead][3];    // minimum corner of box j
    int n[kAhead][3];
    int tab[kAhead][3];     // offset of the (box, axis) table
    float F[6][kAhead + 1]; // face d of the candidate box after c growths of direction d (d < 3: lo[d], else hi[d - 3])
    int verdict[4];         // of the scan over the records: tests in the batch, first boundary failure, "test box 0 alone", column chunks
    int axes[kAhead];       // thin axis | fast column axis << 2 | slow column axis << 4
    int cols[kAhead];       // columns of box j
    int first[kAhead + 1];  // prefix sums of the boxes' column counts, in wavefronts (64 columns)
    int fail;               // first failing test of the batch (kAhead: none)
    int ntodo;              // chunks the free-space table could not clear (the filter pass of obstacle_in_batch)
    int todo[kTodo];
    float ptab[kTab + 4];   // search_point(k) = box_min(k) + iter * res                      (:786-790)
    int vtab[kTab + 4];     // its map index, -1 outside the distance map                      (worldToMap)
    float ctab[3][kCell];   // centre of map cell v along axis k: keyToCoord, (key + 0.5) res as float
};

__device__ void fill_cell_table(const MapView& mp, Ahead& A) {
    for (int k = 0; k < 3; k++)
        for (int v = threadIdx.x; v < mp.dims(k) && v < kCell; v += kSfcThreads)
            A.ctab[k][v] = (float)(((double)(v + mp.key0(k)) + 0.5) * mp.res);
    __syncthreads();
}

// |closest point of the cell box - p| along axis k (Box::closestPoint, float arithmetic)
__device__ __forceinline__ float axis_dist(const Ahead& A, int k, bool have, int v, int code, float p, float delta) {
    const int off = ((code >> (8 * k)) & 255) - 128;
    const float c = have ? A.ctab[k][v + off] : 0.0f;
    const float cmin = c - delta, cmax = c + delta;
    const float q = p < cmin ? cmin : (p > cmax ? cmax : p);
    return fabsf(q - p);
}

// tests the J recorded boxes; returns the index of the first one holding an obstacle, kAhead if none
__device__ int obstacle_in_batch(const MapView& mp, Ahead& A, int J, double margin) {
    const double res = mp.res;
    const float delta = (float)(0.5 * res);
    const int lane = threadIdx.x;
    const long long tt0_ = clock64();
    for (int t = lane >> 6; t < 3 * J; t += kSfcThreads / 64) {  // per-(box, axis) tables, one per wavefront at a time
        const int j = t / 3, k = t - 3 * j;
        const int n = A.n[j][k], o = A.tab[j][k];
        const int key0 = mp.key0(k), dimk = mp.dims(k);
        const double lo = (double)A.lo[j][k];
        bool any_far = false;  // (the whole box far from the origin on some axis: its samples outside the map are harmless, see surely_free)
        if (mp.sat != nullptr)
            for (int m = 0; m < 3; m++)
                any_far = any_far || (A.n[j][m] > 0 && range_far_from_origin(mp, A.lo[j][m], (float)((double)A.lo[j][m] + (double)(A.n[j][m] - 1) * res)));
        for (int it = lane & 63; it < n; it += 64) {
            const float p = (float)(lo + (double)it * res);
            const int v = key_of((double)p, res) - key0;
            A.ptab[o + it] = p;
            A.vtab[o + it] = (v >= 0 && v < dimk) ? v : ((mp.sat != nullptr && (any_far || far_from_origin(mp, p))) ? kBeyondFree : -1);
        }
    }
    __syncthreads();
    SFC_DBG(7, clock64() - tt0_);
    const long long tt1_ = clock64();
    const int total = A.first[J] * 64;
    const double thr = margin + 1e-5;
    // (selected, not indexed: an indexed array of kernel arguments is copied to scratch memory)
    auto stride = [&](int k) -> int64_t { return k == 0 ? 1 : (k == 1 ? (int64_t)mp.dims(0) : (int64_t)mp.dims(0) * mp.dims(1)); };
    // Filter pass (with the free-space table): every 64-column chunk of the batch is asked ONCE, by one lane, whether the cells its
    // columns run through are provably free -- eight reads of the table instead of 64 x nl samples; what is left over (the chunks next
    // to obstacles: a few per cent of a layer of a large box) is listed and only that is evaluated.  The list's order does not matter:
    // the verdict is a minimum over the failing boxes.
    const int all_chunks = A.first[J];
    bool listed = false;
    int n_rounds = all_chunks;
    if (mp.sat != nullptr && margin <= mp.sat_margin && all_chunks >= 64 && all_chunks <= kTodo) {
        if (lane == 0) A.ntodo = 0;
        __syncthreads();
        for (int c = lane; c < all_chunks; c += kSfcThreads) {
            int j = 0;
            for (int t = 1; t < J; t++) j += (c >= A.first[t]) ? 1 : 0;
            const int ax = A.axes[j];
            const int la = ax & 3, ca = (ax >> 2) & 3, cb = (ax >> 4) & 3;
            const int na = A.n[j][ca], nl = A.n[j][la];
            const int c0 = (c - A.first[j]) * 64, cl = A.cols[j] - 1;
            const int c1 = c0 + 63 < cl ? c0 + 63 : cl;
            const int ib0 = c0 / na, ib1 = c1 / na;
            const int ia0 = ib0 == ib1 ? c0 - ib0 * na : 0, ia1 = ib0 == ib1 ? c1 - ib1 * na : na - 1;
            const int ta = A.tab[j][ca], tb = A.tab[j][cb], tl = A.tab[j][la];
            int va0, va1, vb0, vb1, vl0, vl1;
            const int ra = cell_range(A.vtab, ta + ia0, ia1 - ia0 + 1, va0, va1), rb = cell_range(A.vtab, tb + ib0, ib1 - ib0 + 1, vb0, vb1);
            const int rl = cell_range(A.vtab, tl, nl, vl0, vl1);
            bool free_ = ra == 2 || rb == 2 || rl == 2;  // (all of the chunk's samples lie beyond the map, far from the origin)
            if (!free_ && ra == 1 && rb == 1 && rl == 1) {  // (0: a sample outside the map that the exact path has to look at)
                const int x0 = ca == 0 ? va0 : (cb == 0 ? vb0 : vl0), x1 = ca == 0 ? va1 : (cb == 0 ? vb1 : vl1);
                const int y0 = ca == 1 ? va0 : (cb == 1 ? vb0 : vl0), y1 = ca == 1 ? va1 : (cb == 1 ? vb1 : vl1);
                const int z0 = ca == 2 ? va0 : (cb == 2 ? vb0 : vl0), z1 = ca == 2 ? va1 : (cb == 2 ? vb1 : vl1);
                free_ = cells_free(mp, x0, x1, y0, y1, z0, z1);
            }
            if (!free_) A.todo[atomicAdd(&A.ntodo, 1)] = c;
        }
        __syncthreads();
        n_rounds = A.ntodo;
        listed = true;
        SFC_DBG(12, all_chunks); SFC_DBG(13, n_rounds); SFC_DBG(14, 1);
    } else {
        SFC_DBG(15, all_chunks);
    }
    for (int r = lane >> 6; r < n_rounds; r += kSfcThreads / 64) {
        const int stop = *(volatile int*)&A.fail;  // tests behind a failure already found need not be finished
        const int chunk = __builtin_amdgcn_readfirstlane(listed ? A.todo[r] : r);  // wave-uniform: boxes are padded to whole wavefronts
        const int idx = chunk * 64 + (lane & 63);
        // the box this chunk belongs to: lane t asks "does box t + 1 start at or before it" (one LDS read per lane instead of a scan)
        const int lt = lane & 63;
        const int j = __builtin_popcountll(__builtin_amdgcn_ballot_w64(lt + 1 < J && chunk >= A.first[lt + 1 < kAhead ? lt + 1 : kAhead]));
        if (j > stop) continue;
        const int ax = __builtin_amdgcn_readfirstlane(A.axes[j]);
        const int la = ax & 3, ca = (ax >> 2) & 3, cb = (ax >> 4) & 3;
        const int col = idx - A.first[j] * 64;
        const bool live = col < A.cols[j];
        // column -> (fast, slow) iterations.  col < 2^22 (the batch is cut there): a float quotient is off by at most one
        const int na = __builtin_amdgcn_readfirstlane(A.n[j][ca]), nl = __builtin_amdgcn_readfirstlane(A.n[j][la]);
        const int cc = live ? col : 0;
        int ib = (int)((float)cc * (1.0f / (float)na));
        int ia = cc - ib * na;
        if (ia < 0) ib--, ia += na;
        if (ia >= na) ib++, ia -= na;
        const int el = __builtin_amdgcn_readfirstlane(A.tab[j][la]);
        if (!listed && nl >= 8 && mp.sat != nullptr && margin <= mp.sat_margin) {
            // the 64 columns of this chunk with a long thin axis (a slab of a large 3-D box: 64 x nl samples; for the one-sample columns
            // of a layer the table's eight reads cost more than the sample's one -- measured): if the cells they run
            // through are provably free there is nothing to evaluate.  Wave-uniform: the chunk's columns span the rows ib0 .. ib1 of
            // the slow axis -- one row: the fast range it covers, several: the whole fast range (conservative).
            const int c0 = (chunk - A.first[j]) * 64, cl = A.cols[j] - 1;
            const int c1 = c0 + 63 < cl ? c0 + 63 : cl;
            const int ib0 = c0 / na, ib1 = c1 / na;
            const int ia0 = ib0 == ib1 ? c0 - ib0 * na : 0, ia1 = ib0 == ib1 ? c1 - ib1 * na : na - 1;
            const int ta = A.tab[j][ca], tb = A.tab[j][cb];
            int va0, va1, vb0, vb1, vl0, vl1;
            const int ra = cell_range(A.vtab, ta + ia0, ia1 - ia0 + 1, va0, va1), rb = cell_range(A.vtab, tb + ib0, ib1 - ib0 + 1, vb0, vb1);
            const int rl = cell_range(A.vtab, el, nl, vl0, vl1);
            if (ra == 2 || rb == 2 || rl == 2) continue;
            if (ra == 1 && rb == 1 && rl == 1) {
                // (ca, cb, la) is a permutation of (x, y, z)
                const int x0 = ca == 0 ? va0 : (cb == 0 ? vb0 : vl0), x1 = ca == 0 ? va1 : (cb == 0 ? vb1 : vl1);
                const int y0 = ca == 1 ? va0 : (cb == 1 ? vb0 : vl0), y1 = ca == 1 ? va1 : (cb == 1 ? vb1 : vl1);
                const int z0 = ca == 2 ? va0 : (cb == 2 ? vb0 : vl0), z1 = ca == 2 ? va1 : (cb == 2 ? vb1 : vl1);
                if (__builtin_amdgcn_readfirstlane((int)cells_free(mp, x0, x1, y0, y1, z0, z1))) continue;
            }
        }
        const int ea = A.tab[j][ca] + ia, eb = A.tab[j][cb] + ib;
        const float pa = A.ptab[ea], pb = A.ptab[eb];
        const int va = A.vtab[ea], vb = A.vtab[eb];
        const bool inab = live && va >= 0 && vb >= 0;
        const int64_t base_ab = (int64_t)va * stride(ca) + (int64_t)vb * stride(cb);
        const int64_t sl = stride(la);
        bool hit = false;
        // all nearest-cell loads of a group are issued before the first is used: the map is read from beyond the L2 (it is cold
        // at every kernel start), ~2 000 cycles per dependent group -- so the whole column is ONE group whenever it fits
        auto group = [&](auto KZ, int z0) {
            constexpr int kZ = decltype(KZ)::value;
            int code[kZ], vl[kZ];
            float pl[kZ];
#pragma unroll
            for (int z = 0; z < kZ; z++) {
                const int zi = (z0 + z < nl) ? z0 + z : nl - 1;  // (the tail repeats the last point: same verdict)
                vl[z] = A.vtab[el + zi];
                pl[z] = A.ptab[el + zi];
                code[z] = (inab && vl[z] >= 0) ? mp.nearest[base_ab + (int64_t)vl[z] * sl] : 0;
            }
#pragma unroll
            for (int z = 0; z < kZ; z++) {
                // no occupied cell within max_dist, or a sample outside the distance map: the reference's closest_point stays
                // default-constructed and it measures against a cell at the WORLD ORIGIN (:796-800) -- reproduced
                const bool have = (code[z] >> 24) != 0;
                float dist = axis_dist(A, ca, have, va, code[z], pa, delta);  // LInfinityDistance of float differences:
                const float db = axis_dist(A, cb, have, vb, code[z], pb, delta);   // their maximum is a float, widened exactly
                const float dl = axis_dist(A, la, have, vl[z], code[z], pl[z], delta);
                dist = dist < db ? db : dist;
                dist = dist < dl ? dl : dist;
                hit = hit || ((double)dist < thr);
            }
        };
        if (nl <= 2) {
            group(std::integral_constant<int, 2>{}, 0);
        } else if (nl <= 8) {
            group(std::integral_constant<int, 8>{}, 0);
        } else {
            for (int z0 = 0; z0 < nl; z0 += LSCSFC_GROUP) group(std::integral_constant<int, LSCSFC_GROUP>{}, z0);
        }
        if (live && hit) atomicMin(&A.fail, j);
    }
    __syncthreads();
    SFC_DBG(8, clock64() - tt1_);
    SFC_DBG(11, (n_rounds * 64 + kSfcThreads - 1) / kSfcThreads);
    const int f = A.fail;
    __syncthreads();  // (the next batch resets A.fail)
    return f;
}

// expandSFC (:819-881 without goal, :883-946 with it)
__device__ __forceinline__ bool expand_sfc(const MapView& mp, Ahead& A, bool tables, const BoxF& initial, bool use_goal, const float* goal, double margin,
                                           BoxF& out) {
    if (obstacle_in(mp, initial, margin)) return false;
    // the candidate directions, one per nibble (an indexed register array would live in scratch memory: one memory round trip per
    // expansion step)
    unsigned cands = 0x543210u;
    int ncand = 6;
    if (use_goal) {
        int cand[6];
        axis_order(initial, goal, cand);
        cands = 0;
        for (int t = 0; t < 6; t++) cands |= (unsigned)cand[t] << (4 * t);
    }
    auto cand_at = [&](int t) -> int { return (int)((cands >> (4 * t)) & 15u); };
    // growths per direction (one byte each) after m growths of the loop whose next candidate index is `base`: growth u takes candidate
    // (base + u) mod ncand -- in closed form, the look-ahead asks for it at up to 62 growths
    auto growths = [&](int base, int m) -> unsigned long long {
        unsigned long long cp = 0;
        for (int sidx = 0; sidx < ncand; sidx++) {
            int r = (sidx - base) % ncand;  // the first growth that takes candidate sidx
            if (r < 0) r += ncand;
            const int cnt = m > r ? (m - 1 - r) / ncand + 1 : 0;
            cp += (unsigned long long)cnt << (8 * cand_at(sidx));
        }
        return cp;
    };
    auto face = [](const BoxF& b, int d) -> float {
        return d == 0 ? b.lo[0] : (d == 1 ? b.lo[1] : (d == 2 ? b.lo[2] : (d == 3 ? b.hi[0] : (d == 4 ? b.hi[1] : b.hi[2]))));
    };
    const double res = mp.res, rinv = 1.0 / res;
    BoxF sfc = initial, sfc_cand, sfc_update;
    int i = -1;
    while (ncand > 0) {
        sfc_cand = sfc;
        sfc_update = sfc;
        bool failed = false;
        while (!failed) {
            // The tests ahead: box j is what the loop condition sees after j passes.  A face of the candidate box only moves when
            // its own direction is grown, by the reference's float step each time, so the faces after c growths are six short
            // tables (one lane each); lane j then assembles box j from them and sizes it, and a uniform scan over the <= kAhead
            // records decides how many tests the batch holds.  (Growing the boxes one after the other in every lane -- dependent
            // fp64 chains -- cost 1 600 cycles per test, as much as testing them.)
            const long long tg0_ = clock64();
            __syncthreads();  // (the previous batch's readers of A are done)
            if (threadIdx.x < 6) {
                const int d = threadIdx.x;
                float x = face(sfc_cand, d);
                A.F[d][0] = x;
                for (int c = 1; c <= kAhead; c++) {
                    x = (float)((double)x + (d < 3 ? -res : res));
                    A.F[d][c] = x;
                }
            }
            __syncthreads();
            // (wavefront 0 alone: lane j assembles and sizes box j, a shuffle scan over the lanes places the boxes in the batch's index
            // space and a ballot finds where the batch ends; the other wavefronts wait at the barrier instead of competing for the SIMDs
            // and the LDS pipe with sixteen copies of the same scalar work)
            if (threadIdx.x < 64) {
                const int j = threadIdx.x;
                const bool mine = j < kAhead && tables;
                BoxF u = sfc_update;
                if (mine && j > 0) {
                    const unsigned long long cp = growths(i + 1, j - 1);  // what lies before box j: j - 1 growths
                    const int d = cand_at((i + j) % ncand);             // ... and the growth that makes it
                    for (int k = 0; k < 3; k++) {
                        const int cl = (int)((cp >> (8 * k)) & 255ull), ch = (int)((cp >> (8 * (k + 3))) & 255ull);
                        u.lo[k] = A.F[k][cl];
                        u.hi[k] = A.F[k + 3][ch];
                        if (d == k) u.hi[k] = u.lo[k], u.lo[k] = A.F[k][cl + 1];
                        if (d == k + 3) u.lo[k] = u.hi[k], u.hi[k] = A.F[k + 3][ch + 1];
                    }
                }
                const bool inb = in_boundary(mp, u, 0);
                int n[3];
                for (int k = 0; k < 3; k++) n[k] = (int)floor_div((double)(u.hi[k] - u.lo[k]) + 1e-5, res, rinv) + 1;
                // an inverted box (a hull clipped to a previous box it does not touch) has no sample points
                bool empty = n[0] <= 0 || n[1] <= 0 || n[2] <= 0;
                // a box the map's summary proves free passes like an empty one: no table entries, no columns (lscqp_map_prepare)
                empty = empty || (mine && inb && surely_free(mp, margin, u.lo[0], u.lo[1], u.lo[2], n[0], n[1], n[2]));
                // thin axis: the columns run along it (ties: the later axis, so that x stays a column axis)
                const int la = (n[2] <= n[1] && n[2] <= n[0]) ? 2 : (n[1] <= n[0] ? 1 : 0);
                const int ca = la == 0 ? 1 : 0, cb = la == 2 ? 1 : 2;
                const int64_t ncol64 = empty ? 0 : (la == 2 ? (int64_t)n[0] * n[1] : (la == 1 ? (int64_t)n[0] * n[2] : (int64_t)n[1] * n[2]));
                const bool over = !empty && (n[0] > kSfcThreads || n[1] > kSfcThreads || n[2] > kSfcThreads || ncol64 > (1 << 22));
                const int ncol = (mine && !over) ? (int)ncol64 : 0;
                const int need = (mine && !empty) ? n[0] + n[1] + n[2] : 0;
                const int mychunks = (ncol + 63) >> 6;
                int pc = mychunks, pu = need;  // inclusive prefix sums over the lanes
                for (int d = 1; d < kAhead; d <<= 1) {
                    const int tc = __shfl_up(pc, d), tu = __shfl_up(pu, d);
                    if (j >= d) pc += tc, pu += tu;
                }
                const int ec = pc - mychunks, eu = pu - need;  // what lies before box j
                // the batch ends at the first box outside the world boundary (that test fails without looking at the map: jstop) or
                // beyond the batch's limits (jbrk; box 0 beyond them is tested on its own, the sequential way)
                const unsigned long long mstop = __builtin_amdgcn_ballot_w64(mine && !inb);
                const unsigned long long mbrk = __builtin_amdgcn_ballot_w64(mine && (over || (int64_t)ec * 64 + ncol > (1 << 22) || eu + need > kTab));
                const int jstop_ = mstop ? __builtin_ctzll(mstop) : kAhead, jbrk = mbrk ? __builtin_ctzll(mbrk) : kAhead;
                // ... or behind the LSCSFC_SAMPLED-th box that has to be sampled: tests behind the first failure are wasted work, boxes
                // the free-space table passes are not
                const unsigned long long msamp = __builtin_amdgcn_ballot_w64(mine && ncol > 0);
                const int before = __builtin_popcountll(msamp & ((1ull << j) - 1ull));
                const unsigned long long mcap = __builtin_amdgcn_ballot_w64(mine && before >= LSCSFC_SAMPLED);
                const int jcap = mcap ? __builtin_ctzll(mcap) : kAhead;
                const int jlim = jbrk < jcap ? jbrk : jcap;
                const int J_ = tables ? (jstop_ < jlim ? jstop_ : jlim) : 0;
                const bool alone_ = !tables || (jbrk == 0 && jstop_ > 0);
                if (mine && j < J_) {
                    for (int k = 0; k < 3; k++) {
                        A.lo[j][k] = u.lo[k];
                        A.n[j][k] = empty ? 0 : n[k];  // (no table entries, no columns)
                    }
                    A.tab[j][0] = eu, A.tab[j][1] = eu + (empty ? 0 : n[0]), A.tab[j][2] = eu + (empty ? 0 : n[0] + n[1]);
                    A.axes[j] = la | (ca << 2) | (cb << 4);
                    A.cols[j] = ncol;
                    A.first[j + 1] = pc;
                }
                if (j == 0) {
                    A.first[0] = 0, A.fail = kAhead;
                    A.verdict[0] = J_, A.verdict[1] = tables ? jstop_ : kAhead, A.verdict[2] = alone_ ? 1 : 0;
                }
                // chunks of the whole batch: the inclusive sum at its last box
                const int total_chunks = __shfl(pc, J_ > 0 ? J_ - 1 : 0);
                if (j == 0) A.verdict[3] = J_ > 0 ? total_chunks : 0;
            }
            __syncthreads();
            int J = A.verdict[0], jstop = A.verdict[1];
            const bool alone = A.verdict[2] != 0;
            const int chunks = A.verdict[3];
            SFC_DBG(9, clock64() - tg0_);
            SFC_DBG(0, 1); SFC_DBG(1, J); SFC_DBG(2, chunks); SFC_DBG(3, alone ? 1 : 0);
            const long long t0_ = clock64();
            int jfail;
            if (alone) {
                if (!in_boundary(mp, sfc_update, 0)) {
                    jfail = 0;
                } else {
                    jfail = obstacle_in(mp, sfc_update, margin) ? 0 : kAhead;
                }
                J = 1;
                jstop = kAhead;
            } else {
                jfail = J > 0 ? obstacle_in_batch(mp, A, J, margin) : kAhead;
            }
            SFC_DBG(4, clock64() - t0_);
            const int jf = jfail < jstop ? jfail : jstop;  // first failing test of this batch, kAhead if none was seen
            const int passes = jf < J ? jf : J;
            const long long tr0_ = clock64();
            if (alone) {
                for (int j = 0; j < passes; j++) {
                    i++;
                    if (i >= ncand) i = 0;
                    grow(sfc, sfc_cand, sfc_update, cand_at(i), res);
                }
            } else if (passes > 0) {  // the state after `passes` growths, from the face tables
                const unsigned long long cp = growths(i + 1, passes - 1);
                i = (i + passes) % ncand;
                const int d = cand_at(i);
                for (int k = 0; k < 3; k++) {
                    const int cl = (int)((cp >> (8 * k)) & 255ull), ch = (int)((cp >> (8 * (k + 3))) & 255ull);
                    sfc.lo[k] = A.F[k][cl];
                    sfc.hi[k] = A.F[k + 3][ch];
                    sfc_cand.lo[k] = sfc_update.lo[k] = sfc.lo[k];
                    sfc_cand.hi[k] = sfc_update.hi[k] = sfc.hi[k];
                    if (d == k) sfc_update.hi[k] = sfc.lo[k], sfc_cand.lo[k] = sfc_update.lo[k] = A.F[k][cl + 1];
                    if (d == k + 3) sfc_update.lo[k] = sfc.hi[k], sfc_cand.hi[k] = sfc_update.hi[k] = A.F[k + 3][ch + 1];
                }
            }
            SFC_DBG(10, clock64() - tr0_);
            failed = jf <= J && jf < kAhead;
        }
        if (i < 0) return false;  // initial box outside the world: the reference erases begin() - 1 here (undefined)
        {  // erase direction i
            const unsigned below = cands & ((1u << (4 * i)) - 1u);
            cands = below | ((cands >> (4 * (i + 1))) << (4 * i));
        }
        ncand--;
        i = (i > 0) ? i - 1 : ncand - 1;
    }
    const double delta = margin - ((int)(margin / res) * res);  // margin compensation, :868-877
    for (int k = 0; k < 3; k++) {
        if ((double)sfc.lo[k] > (double)mp.world_min(k) + 1e-5) sfc.lo[k] = (float)((double)sfc.lo[k] - delta);
        if ((double)sfc.hi[k] < (double)mp.world_max(k) - 1e-5) sfc.hi[k] = (float)((double)sfc.hi[k] + delta);
    }
    out = sfc;
    return true;
}

__device__ bool point_in(const BoxF& b, const float* p) {  // Box::isPointInBox :81-88
    bool ok = true;
    for (int k = 0; k < 3; k++) ok = ok && ((double)p[k] > (double)b.lo[k] - 1e-5) && ((double)p[k] < (double)b.hi[k] + 1e-5);
    return ok;
}

__device__ void clip_to_prev(const BoxF& prev, BoxF& ini, double res) {  // :677-685, :762-770
    if (!(point_in(prev, ini.lo) && point_in(prev, ini.hi))) {
        for (int k = 0; k < 3; k++) {
            const float lo = prev.lo[k] > ini.lo[k] ? prev.lo[k] : ini.lo[k];
            const float hi = prev.hi[k] < ini.hi[k] ? prev.hi[k] : ini.hi[k];
            ini.lo[k] = (float)(ceil(((double)lo - 1e-5) / res) * res);
            ini.hi[k] = (float)(floor(((double)hi + 1e-5) / res) * res);
        }
    }
}

// (A/B knob of the development builds: LSCSFC_WAVES_PER_EU caps the registers so that several smaller workgroups share a CU)
#ifdef LSCSFC_WAVES_PER_EU
#define LSCSFC_KERNEL_ATTR __attribute__((amdgpu_waves_per_eu(LSCSFC_WAVES_PER_EU, LSCSFC_WAVES_PER_EU)))
#else
#define LSCSFC_KERNEL_ATTR
#endif
__global__ __launch_bounds__(kSfcThreads) LSCSFC_KERNEL_ATTR void construct_sfc_kernel(MapView mp, int mode, int M, int64_t n, const double* __restrict__ pts,
                                                           const double* __restrict__ radius, lscqp_box* __restrict__ sfc,
                                                           int32_t* __restrict__ status) {
    const int64_t a = blockIdx.x;
    if (a >= n) return;
    const double res = mp.res;
    const double margin = radius[a];
    float P[3][3];
    for (int i = 0; i < 3; i++)
        for (int k = 0; k < 3; k++) P[i][k] = (float)pts[9 * a + 3 * i + k];
    lscqp_box* S = sfc + a * M;
    BoxF ini, out, prev;
    bool ok;
    __shared__ Ahead A;
    const bool tables = mp.dims(0) <= kCell && mp.dims(1) <= kCell && mp.dims(2) <= kCell;  // (larger worlds: sequential tests)
    if (tables) fill_cell_table(mp, A);
#ifdef LSCSFC_DEBUG
    const long long tk0_ = clock64();
    struct Fin { long long t; __device__ ~Fin() { SFC_DBG(5, clock64() - t); SFC_DBG(6, 1); } } fin_{tk0_};
#endif
    // ONE call site of expand_sfc (so that it is inlined and the map stays in registers): the modes differ in the initial box,
    // in the goal-ordered directions, and constructSFCFromConvexHull may make a second attempt
    if (mode != LSCQP_SFC_INIT)
        for (int k = 0; k < 3; k++) prev.lo[k] = (float)S[M - 1].bmin[k], prev.hi[k] = (float)S[M - 1].bmax[k];
    float hull_mn[3], hull_mx[3];  // hull + next waypoint
    for (int k = 0; k < 3; k++) {
        float mn = P[0][k], mx = P[0][k];
        for (int i = 1; i < 3; i++) {
            mn = P[i][k] < mn ? P[i][k] : mn;
            mx = P[i][k] > mx ? P[i][k] : mx;
        }
        hull_mn[k] = mn, hull_mx[k] = mx;
    }
    ok = false;
    const int attempts = mode == LSCQP_SFC_FROM_HULL ? 2 : 1;
    for (int att = 0; att < attempts && !ok; att++) {
        if (mode == LSCQP_SFC_FROM_HULL && att == 0) {  // constructSFCFromConvexHull :414-436: hull + next waypoint, aligned by round() (:692-722)
            for (int k = 0; k < 3; k++) {
                ini.lo[k] = (float)(round((double)hull_mn[k] / res) * res);
                ini.hi[k] = (float)(round((double)hull_mx[k] / res) * res);
            }
        } else if (mode == LSCQP_SFC_FROM_HULL) {  // the hull alone, inside the previous box, aligned outwards (:724-775)
            for (int k = 0; k < 3; k++) {
                const float mn = P[0][k] < P[1][k] ? P[0][k] : P[1][k], mx = P[0][k] > P[1][k] ? P[0][k] : P[1][k];
                ini.lo[k] = (float)(floor((double)mn / res) * res);
                ini.hi[k] = (float)(ceil((double)mx / res) * res);
            }
            clip_to_prev(prev, ini, res);
        } else {  // initializeSFC :366-384; constructSFCFromPoint :396-412, expandSFCFromPoint :666-690
            for (int k = 0; k < 3; k++) {
                ini.lo[k] = (float)(floor((double)P[0][k] / res) * res);
                ini.hi[k] = (float)(ceil((double)P[0][k] / res) * res);
            }
            if (mode == LSCQP_SFC_FROM_POINT) clip_to_prev(prev, ini, res);
        }
        ok = expand_sfc(mp, A, tables, ini, mode == LSCQP_SFC_FROM_POINT, P[1], margin, out);
        if (ok && mode == LSCQP_SFC_FROM_HULL && att == 0)  // isSuperSetOfConvexHull :135-150
            for (int k = 0; k < 3; k++)
                ok = ok && !((double)hull_mn[k] < (double)out.lo[k] - 1e-5 || (double)hull_mx[k] > (double)out.hi[k] + 1e-5);
    }
    if (mode == LSCQP_SFC_INIT) {
        if (ok)
            for (int t = threadIdx.x; t < M * 6; t += kSfcThreads) {
                const int m = t / 6, c = t % 6;
                double val = 0;
                for (int k = 0; k < 3; k++) {
                    val = (c == k) ? (double)out.lo[k] : val;
                    val = (c == k + 3) ? (double)out.hi[k] : val;
                }
                (c < 3 ? S[m].bmin[c] : S[m].bmax[c - 3]) = val;
            }
        if (threadIdx.x == 0) status[a] = ok ? 1 : 0;
        return;
    }
    // sfcs[m] = sfcs[m + 1] for m < M - 1, then the new (or the kept) last box.  Every lane moves whole elements; the
    // values were all read before any is overwritten only if the shift is staged, so stage it in registers.
    double stage = 0;  // (M - 1) * 6 <= kSfcThreads: one element per lane
    const int ts = threadIdx.x;
    if (ts < (M - 1) * 6) stage = (ts % 6) < 3 ? S[ts / 6 + 1].bmin[ts % 6] : S[ts / 6 + 1].bmax[ts % 6 - 3];
    __syncthreads();
    if (ts < (M - 1) * 6) ((ts % 6) < 3 ? S[ts / 6].bmin[ts % 6] : S[ts / 6].bmax[ts % 6 - 3]) = stage;
    if (ok && threadIdx.x < 6) {
        const int c = threadIdx.x;
        double val = 0;
        for (int k = 0; k < 3; k++) {
            val = (c == k) ? (double)out.lo[k] : val;
            val = (c == k + 3) ? (double)out.hi[k] : val;
        }
        (c < 3 ? S[M - 1].bmin[c] : S[M - 1].bmax[c - 3]) = val;
    }
    if (threadIdx.x == 0) status[a] = ok ? 1 : 0;
}

}  // namespace lscsfc

#define LSCSFC_HIP(call)                                                                                              \
    do {                                                                                                              \
        const hipError_t e_ = (call);                                                                                 \
        if (e_ != hipSuccess) return lscqp_set_error_(LSCQP_ERR_HIP, (std::string(#call ": ") + hipGetErrorString(e_)).c_str()); \
    } while (0)

#ifdef LSCSFC_VARIANT_ONLY
// This translation unit is the THROUGHPUT build of the corridor kernel (lscsfc_tp.hip): the same source with 512 threads per agent and
// the registers capped so that two workgroups share a CU; only its launcher is exported, the map and host entry points live in lscsfc.hip.
extern "C" hipError_t lscsfc_launch_throughput_(const void* view, int mode, int M, int64_t n, const double* d_points, const double* d_radius,
                                                lscqp_box* d_sfc, int32_t* d_status_out, void* stream) {
    hipLaunchKernelGGL(lscsfc::construct_sfc_kernel, dim3((unsigned)n), dim3(lscsfc::kSfcThreads), 0, (hipStream_t)stream,
                       *reinterpret_cast<const lscsfc::MapView*>(view), mode, M, n, d_points, d_radius, d_sfc, d_status_out);
    return hipGetLastError();
}
#else
extern "C" hipError_t lscsfc_launch_throughput_(const void* view, int mode, int M, int64_t n, const double* d_points, const double* d_radius,
                                                lscqp_box* d_sfc, int32_t* d_status_out, void* stream);  // lscsfc_tp.hip
extern "C" int lscsfc_throughput_max_cells_(void);

extern "C" {

#ifdef LSCSFC_DEBUG
int lscsfc_dbg_read(unsigned long long* out, int reset) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(lscsfc::sfc_dbg), sizeof(unsigned long long) * 16);
    if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(lscsfc::sfc_dbg), z, sizeof z); }
    return 0;
}
#endif

int lscqp_map_create(const double* boxes, int64_t n_boxes, const double* world_min, const double* world_max, double resolution,
                     double max_dist, lscqp_map* out) {
    if (!out || !world_min || !world_max || (n_boxes > 0 && !boxes) || n_boxes < 0)
        return lscqp_set_error_(LSCQP_ERR_INVALID_ARGUMENT, "null argument");
    if (!(resolution > 0) || !(max_dist >= 0)) return lscqp_set_error_(LSCQP_ERR_INVALID_ARGUMENT, "resolution must be > 0, max_dist >= 0");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return lscqp_set_error_(LSCQP_ERR_NO_DEVICE, "no HIP device: lscqp has no CPU fallback");
    lscqp_map_s* mp = new lscqp_map_s();
    mp->pool = new lscqp::StagePool();
    (void)hipGetDevice(&mp->device);
    mp->res = resolution;
    int64_t nvox = 1;
    for (int k = 0; k < 3; k++) {
        mp->world_min[k] = (float)world_min[k];
        mp->world_max[k] = (float)world_max[k];
        mp->key0[k] = lscsfc::key_of((double)mp->world_min[k], resolution);  // DynamicEDTOctomap bounding box keys
        mp->dims[k] = lscsfc::key_of((double)mp->world_max[k], resolution) - mp->key0[k] + 1;
        if (mp->dims[k] <= 0) {
            delete mp;
            return lscqp_set_error_(LSCQP_ERR_INVALID_ARGUMENT, "world_max must not be below world_min");
        }
        nvox *= mp->dims[k];
    }
    const int R = (int)floor(max_dist / resolution + 1e-9);
    if (R > 120) {
        delete mp;
        return lscqp_set_error_(LSCQP_ERR_UNSUPPORTED, "max_dist / resolution must be <= 120 cells");
    }
    mp->radius_cells = R;
    mp->d_occ = nullptr;
    mp->d_nearest = nullptr;
    double* d_boxes = nullptr;
    int8_t *d_a = nullptr, *d_b = nullptr, *d_c = nullptr;
    auto cleanup = [&]() {
        if (d_boxes) (void)hipFree(d_boxes);
        if (d_a) (void)hipFree(d_a);
        if (d_b) (void)hipFree(d_b);
        if (d_c) (void)hipFree(d_c);
    };
    auto fail = [&](hipError_t e, const char* what) {
        cleanup();
        if (mp->d_occ) (void)hipFree(mp->d_occ);
        if (mp->d_nearest) (void)hipFree(mp->d_nearest);
        delete mp;
        return lscqp_set_error_(LSCQP_ERR_HIP, (std::string(what) + ": " + hipGetErrorString(e)).c_str());
    };
    hipError_t e;
    if ((e = hipMalloc(&mp->d_occ, nvox)) != hipSuccess) return fail(e, "hipMalloc(occupancy)");
    if ((e = hipMalloc(&mp->d_nearest, nvox * sizeof(int32_t))) != hipSuccess) return fail(e, "hipMalloc(nearest)");
    if ((e = hipMalloc(&d_a, nvox)) != hipSuccess || (e = hipMalloc(&d_b, nvox)) != hipSuccess || (e = hipMalloc(&d_c, nvox)) != hipSuccess)
        return fail(e, "hipMalloc(scratch)");
    if ((e = hipMemset(mp->d_occ, 0, nvox)) != hipSuccess) return fail(e, "hipMemset");
    if (n_boxes > 0) {
        if ((e = hipMalloc(&d_boxes, n_boxes * 6 * sizeof(double))) != hipSuccess) return fail(e, "hipMalloc(boxes)");
        if ((e = hipMemcpy(d_boxes, boxes, n_boxes * 6 * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "hipMemcpy(boxes)");
        hipLaunchKernelGGL(lscsfc::rasterise_kernel, dim3((unsigned)n_boxes), dim3(256), 0, 0, d_boxes, resolution, mp->key0[0], mp->key0[1],
                           mp->key0[2], mp->dims[0], mp->dims[1], mp->dims[2], mp->d_occ);
    }
    const unsigned blocks = (unsigned)((nvox + 255) / 256);
    hipLaunchKernelGGL(lscsfc::nearest_x_kernel, dim3(blocks), dim3(256), 0, 0, mp->dims[0], nvox, R, mp->d_occ, d_a);
    hipLaunchKernelGGL(lscsfc::nearest_y_kernel, dim3(blocks), dim3(256), 0, 0, mp->dims[0], mp->dims[1], nvox, R, d_a, d_b, d_c);
    hipLaunchKernelGGL(lscsfc::nearest_z_kernel, dim3(blocks), dim3(256), 0, 0, mp->dims[0], mp->dims[1], mp->dims[2], nvox, R, d_b, d_c,
                       mp->d_nearest);
    if ((e = hipGetLastError()) != hipSuccess) return fail(e, "map kernels");
    if ((e = hipDeviceSynchronize()) != hipSuccess) return fail(e, "map kernels");
    cleanup();
    *out = mp;
    return LSCQP_OK;
}

int lscqp_map_create_from_csv(const char* path, const double* world_min, const double* world_max, double resolution, double max_dist,
                              lscqp_map* out) {
    if (!path) return lscqp_set_error_(LSCQP_ERR_INVALID_ARGUMENT, "null path");
    FILE* f = fopen(path, "r");
    if (!f) return lscqp_set_error_(LSCQP_ERR_INVALID_ARGUMENT, (std::string("cannot open world file ") + path).c_str());
    std::vector<double> boxes;
    char line[4096];
    while (fgets(line, sizeof line, f)) {  // rows: centre x,y,z, size x,y,z; a row with fewer than two fields ends the list (:267-269)
        double v[6];
        int cnt = 0;
        char* s = line;
        while (cnt < 6) {
            char* end = nullptr;
            const double x = strtod(s, &end);
            if (end == s) break;
            v[cnt++] = x;
            s = end;
            while (*s == ',' || *s == ' ' || *s == '\t') s++;
        }
        if (cnt < 2) break;
        if (cnt < 6) {
            fclose(f);
            return lscqp_set_error_(LSCQP_ERR_INVALID_ARGUMENT, "world CSV row with fewer than 6 fields");
        }
        boxes.insert(boxes.end(), v, v + 6);
    }
    fclose(f);
    return lscqp_map_create(boxes.data(), (int64_t)(boxes.size() / 6), world_min, world_max, resolution, max_dist, out);
}

void lscqp_map_destroy(lscqp_map mp) {
    if (!mp) return;
    if (mp->d_occ) (void)hipFree(mp->d_occ);
    if (mp->d_nearest) (void)hipFree(mp->d_nearest);
    if (mp->d_sat) (void)hipFree(mp->d_sat);
    delete mp->pool;
    delete mp;
}

int lscqp_map_prepare(lscqp_map mp, double max_radius) {
    if (!mp) return lscqp_set_error_(LSCQP_ERR_INVALID_ARGUMENT, "null map");
    if (!(max_radius > 0)) return lscqp_set_error_(LSCQP_ERR_INVALID_ARGUMENT, "max_radius > 0 required");
    if (mp->d_sat && mp->sat_margin >= max_radius) return LSCQP_OK;  // (a table built for a larger margin serves the smaller ones)
    const int nx = mp->dims[0], ny = mp->dims[1], nz = mp->dims[2];
    const int64_t nvox = (int64_t)nx * ny * nz;
    if (nvox > 0x7fffffffLL) return lscqp_set_error_(LSCQP_ERR_UNSUPPORTED, "the free-space table counts cells in 32 bits: map too large (the corridors work without it)");
    hipError_t e = hipDeviceSynchronize();  // (no corridor launch may be reading the old table)
    if (e != hipSuccess) return lscqp_set_error_(LSCQP_ERR_HIP, (std::string("hipDeviceSynchronize: ") + hipGetErrorString(e)).c_str());
    if (!mp->d_sat && (e = hipMalloc(&mp->d_sat, nvox * sizeof(int32_t))) != hipSuccess) {
        mp->d_sat = nullptr;
        return lscqp_set_error_(LSCQP_ERR_HIP, (std::string("hipMalloc(free-space table): ") + hipGetErrorString(e)).c_str());
    }
    mp->sat_margin = 0;
    mp->generation++;
    // what the float arithmetic of the exact test can lose against the cell bound: the sample, the cell centre and centre -+ half a cell
    // are floats of the world's magnitude (half an ulp each; 7.6e-6 per ulp at 100 m)
    double wabs = 0;
    for (int k = 0; k < 3; k++) wabs = fmax(wabs, fmax(fabs((double)mp->world_min[k]), fabs((double)mp->world_max[k])) + 2.0 * mp->res);
    const double slop = 1e-4 + 4.0 * wabs * 1.1920929e-7;
    hipLaunchKernelGGL(lscsfc::classify_free_kernel, dim3((unsigned)((nvox + 255) / 256)), dim3(256), 0, 0, nx, ny, nz, mp->key0[0], mp->key0[1], mp->key0[2],
                       mp->res, max_radius, slop, mp->d_nearest, mp->d_sat);
    // x lines: one per (y, z); y lines: one per (x, z); z lines: one per (x, y)
    const int64_t lx = (int64_t)ny * nz, ly = (int64_t)nx * nz, lz = (int64_t)nx * ny;
    hipLaunchKernelGGL(lscsfc::prefix_axis_kernel, dim3((unsigned)((lx + 255) / 256)), dim3(256), 0, 0, lx, nx, (int64_t)1, lx, (int64_t)nx, (int64_t)0, mp->d_sat);
    hipLaunchKernelGGL(lscsfc::prefix_axis_kernel, dim3((unsigned)((ly + 255) / 256)), dim3(256), 0, 0, ly, ny, (int64_t)nx, (int64_t)nx, (int64_t)1, (int64_t)nx * ny, mp->d_sat);
    hipLaunchKernelGGL(lscsfc::prefix_axis_kernel, dim3((unsigned)((lz + 255) / 256)), dim3(256), 0, 0, lz, nz, (int64_t)nx * ny, lz, (int64_t)1, (int64_t)0, mp->d_sat);
    if ((e = hipGetLastError()) != hipSuccess || (e = hipDeviceSynchronize()) != hipSuccess) {
        // the table stays allocated (a captured plan graph may still hold the pointer) but serves nobody: margin 0, and the bumped
        // generation makes every plan drop its graph before the next replan
        return lscqp_set_error_(LSCQP_ERR_HIP, (std::string("free-space table: ") + hipGetErrorString(e)).c_str());
    }
    mp->sat_margin = max_radius;
    return LSCQP_OK;
}

int lscqp_map_info(lscqp_map mp, int32_t* dims, int32_t* key0) {
    if (!mp || !dims || !key0) return lscqp_set_error_(LSCQP_ERR_INVALID_ARGUMENT, "null argument");
    for (int k = 0; k < 3; k++) dims[k] = mp->dims[k], key0[k] = mp->key0[k];
    return LSCQP_OK;
}

int lscqp_map_download(lscqp_map mp, uint8_t* occ, int32_t* nearest) {
    if (!mp) return lscqp_set_error_(LSCQP_ERR_INVALID_ARGUMENT, "null map");
    const int64_t nvox = (int64_t)mp->dims[0] * mp->dims[1] * mp->dims[2];
    if (occ) LSCSFC_HIP(hipMemcpy(occ, mp->d_occ, nvox, hipMemcpyDeviceToHost));
    if (nearest) LSCSFC_HIP(hipMemcpy(nearest, mp->d_nearest, nvox * sizeof(int32_t), hipMemcpyDeviceToHost));
    return LSCQP_OK;
}

int lscqp_construct_sfc_raw_(lscqp_map mp, int mode, int M, int64_t n, const double* d_points, const double* d_radius, lscqp_box* d_sfc,
                             int32_t* d_status_out, void* stream) {
    lscsfc::MapView v;
    v.res = mp->res;
    v.wmin0 = mp->world_min[0], v.wmin1 = mp->world_min[1], v.wmin2 = mp->world_min[2];
    v.wmax0 = mp->world_max[0], v.wmax1 = mp->world_max[1], v.wmax2 = mp->world_max[2];
    v.key00 = mp->key0[0], v.key01 = mp->key0[1], v.key02 = mp->key0[2];
    v.dims0 = mp->dims[0], v.dims1 = mp->dims[1], v.dims2 = mp->dims[2];
    v.nearest = mp->d_nearest;
    v.sat = mp->d_sat;
    v.sat_margin = mp->sat_margin;
    // Two builds of the one kernel source.  LATENCY (this file): 1024 threads per agent, one workgroup per CU -- the sixteen wavefronts
    // shorten the chain of dependent tests of ONE corridor (171 / 204 / 335 us per corridor with 1024 / 512 / 256 threads), right while
    // agents <= CUs.  THROUGHPUT (lscsfc_tp.hip): 256 threads per agent, registers capped at 128, batch tables cut to 35 KB of LDS: four
    // workgroups per CU, the chains of four agents overlap -- 4096 agents 2.28 -> 1.06 ms with the free-space table, 2.89 -> 2.01 ms
    // without.  Same statements, same boxes.
    int n_cu = 256;
    {
        int dev = 0;
        hipDeviceProp_t prop;
        static int cached[64] = {};
        if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
            if (!cached[dev] && hipGetDeviceProperties(&prop, dev) == hipSuccess) cached[dev] = prop.multiProcessorCount;
            if (cached[dev]) n_cu = cached[dev];
        }
    }
    const char* force = getenv("LSCSFC_VARIANT");  // testing knob: "latency" | "throughput"
    const int tp_cells = lscsfc_throughput_max_cells_();  // (its cell-centre table is shorter: larger maps stay with the latency build)
    const bool fits = mp->dims[0] <= tp_cells && mp->dims[1] <= tp_cells && mp->dims[2] <= tp_cells;
    const bool tp = fits && (force ? (force[0] == 't') : (n > (int64_t)n_cu));
    hipError_t e;
    if (tp) {
        e = lscsfc_launch_throughput_(&v, mode, M, n, d_points, d_radius, d_sfc, d_status_out, stream);
    } else {
        hipLaunchKernelGGL(lscsfc::construct_sfc_kernel, dim3((unsigned)n), dim3(lscsfc::kSfcThreads), 0, (hipStream_t)stream, v, mode, M, n, d_points,
                           d_radius, d_sfc, d_status_out);
        e = hipGetLastError();
    }
    if (e != hipSuccess) return lscqp_set_error_(LSCQP_ERR_HIP, (std::string("HIP launch failed: ") + hipGetErrorString(e)).c_str());
    return LSCQP_OK;
}

// HOST pointers, synchronous: the batch-of-1 form the unchanged planner loop uses (TrajPlanner::generateSFC, one agent at a time)
int lscqp_construct_sfc(lscqp_map mp, int32_t mode, int32_t M, int64_t n, const double* points, const double* radius, lscqp_box* sfc,
                        int32_t* status_out) {
    if (!mp) return lscqp_set_error_(LSCQP_ERR_INVALID_ARGUMENT, "null map");
    if (mode != LSCQP_SFC_INIT && mode != LSCQP_SFC_FROM_HULL && mode != LSCQP_SFC_FROM_POINT)
        return lscqp_set_error_(LSCQP_ERR_INVALID_ARGUMENT, "mode must be LSCQP_SFC_INIT, LSCQP_SFC_FROM_HULL or LSCQP_SFC_FROM_POINT");
    if (n < 0 || M < 1 || M > 21) return lscqp_set_error_(LSCQP_ERR_INVALID_ARGUMENT, "n >= 0 and 1 <= M <= 21 required");
    if (n == 0) return LSCQP_OK;
    if (!points || !radius || !sfc || !status_out) return lscqp_set_error_(LSCQP_ERR_INVALID_ARGUMENT, "null buffer");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return lscqp_set_error_(LSCQP_ERR_NO_DEVICE, "no HIP device: lscqp has no CPU fallback");
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t b_p = al(n * 9 * sizeof(double)), b_r = al(n * sizeof(double)), b_s = al(n * M * sizeof(lscqp_box)),
                 b_st = al(n * sizeof(int32_t));
    const size_t total = b_p + b_r + b_s + b_st;  // [points | radius | boxes (in/out) | status (out)]
    lscqp::SlotGuard sg{*mp->pool, mp->pool->acquire(total)};
    if (!sg.slot) return lscqp_set_error_(LSCQP_ERR_HIP, "staging allocation failed");
    hipStream_t st = sg.slot->stream;
    char* const hb = (char*)sg.slot->h;
    char* const db = (char*)sg.slot->d;
    memcpy(hb, points, n * 9 * sizeof(double));
    memcpy(hb + b_p, radius, n * sizeof(double));
    memcpy(hb + b_p + b_r, sfc, n * M * sizeof(lscqp_box));
    LSCSFC_HIP(hipMemcpyAsync(db, hb, b_p + b_r + b_s, hipMemcpyHostToDevice, st));
    const int rc = lscqp_construct_sfc_raw_(mp, mode, M, n, (const double*)db, (const double*)(db + b_p), (lscqp_box*)(db + b_p + b_r),
                                            (int32_t*)(db + b_p + b_r + b_s), st);
    if (rc != LSCQP_OK) return rc;
    LSCSFC_HIP(hipMemcpyAsync(hb + b_p + b_r, db + b_p + b_r, b_s + b_st, hipMemcpyDeviceToHost, st));
    LSCSFC_HIP(hipStreamSynchronize(st));
    memcpy(sfc, hb + b_p + b_r, n * M * sizeof(lscqp_box));
    memcpy(status_out, hb + b_p + b_r + b_s, n * sizeof(int32_t));
    return LSCQP_OK;
}

}  // extern "C"
#endif  // LSCSFC_VARIANT_ONLY
