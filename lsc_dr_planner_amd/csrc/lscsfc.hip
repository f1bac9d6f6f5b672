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
// across the group; the next tests of the expansion loop (up to 126 -- 62 in the throughput build --, through the failures the world boundary will cause) are known in
// advance, the up to 10 of them that the map's free-space table cannot pass are evaluated together, a lane per column of sample
// points (obstacle_in_batch below), and the first test that fails decides how far the box has grown.
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
    // work distribution of a launch (set per launch; lscqp_construct_sfc_device_ordered): the k-th workgroup builds the corridor of agent
    // order[k] (NULL: k), and every workgroup leaves the cycles it took (>> 4) in cost[agent] (NULL: nothing) -- the hint for the order
    // of the NEXT replan's launch: corridors along a wall cost three times the ones in open space, and a launch ends with its last one
    const int32_t* order;
    uint32_t* cost;
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
    // (32-bit index arithmetic: lscqp_map_prepare refuses maps of more than 2^31 - 1 cells)
    const uint32_t sy = (uint32_t)mp.dims(0), sz = (uint32_t)mp.dims(0) * (uint32_t)mp.dims(1);
    auto at = [&](int x, int y, int z) -> int64_t {  // prefix sum up to (x, y, z) inclusive; -1 on any axis: empty
        return (x < 0 || y < 0 || z < 0) ? 0 : (int64_t)mp.sat[(uint32_t)x + (uint32_t)y * sy + (uint32_t)z * sz];
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
__device__ unsigned long long sfc_dbg[32];
#define SFC_DBG(i, v) do { if (threadIdx.x == 0) atomicAdd(&sfc_dbg[i], (unsigned long long)(v)); } while (0)
#else
#define SFC_DBG(i, v) do { } while (0)
#endif
#ifndef LSCSFC_GROUP
#define LSCSFC_GROUP 4
#endif
#ifndef LSCSFC_AHEAD
#define LSCSFC_AHEAD 126  // tests of a batch's look-ahead (lane j of the first ceil(AHEAD / 64) wavefronts assembles box j; growth counts are bytes: <= 254).
                          // Measured on the 3-D chain of bench.py (64 agents, a room with 24 boxes): 63 / 126 / 190 / 254 -> 432 / 400 / 404 / 413 us per replan
#endif
#ifndef LSCSFC_SAMPLED
#define LSCSFC_SAMPLED 10  // ... of which at most this many have to be sampled (boxes the free-space table passes cost nothing); <= 63
                           // (8 / 10 / 12 / 16 / 24: 3-D chain 396 / 396 / 400 / 439 / 530 us per replan; 4096 agents 0.94 / 0.92 / 0.97 ms with 8 / 10 / 12)
#endif
constexpr int kAhead = LSCSFC_AHEAD;
constexpr int kSamp = LSCSFC_SAMPLED;           // record slots of a batch: the boxes that have to be sampled
constexpr int kGenWaves = (kAhead + 63) / 64;   // wavefronts that assemble the look-ahead
constexpr int kBoxes = kGenWaves * 64;
static_assert(kAhead >= 2 && kAhead <= 254 && kSamp >= 1 && kSamp <= 63, "look-ahead: growth counts are bytes, the sampled boxes are looked up with one ballot");
static_assert(kBoxes <= kSfcThreads && kSfcThreads % 64 == 0, "a lane per test of the look-ahead");
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
static_assert(LSCSFC_TODO <= 65536, "a listed item packs the chunk into 16 bits");
struct Ahead {
    // the boxes of the batch that have to be sampled, in test order (slot k):
    float lo[kSamp][3];     // minimum corner
    int n[kSamp][3];        // sample points per axis
    int tab[kSamp][3];      // offset of the (box, axis) table
    int axes[kSamp];        // thin axis | fast column axis << 2 | slow column axis << 4
    int cols[kSamp];        // columns of the box
    int first[kSamp + 1];   // prefix sums of the boxes' column counts, in wavefronts (64 columns)
    int sbox[kSamp];        // which test of the look-ahead the slot is
    int nsamp;              // slots in use
    // the look-ahead:
    float F[6][kAhead + 2]; // face d of the accepted box after c growths of direction d (d < 3: lo[d], else hi[d - 3]); kept across batches
    int flen[6];            // valid entries of F[d]
    int gmax[6];            // growths of direction d the world boundary admits (counted from the accepted box)
    int need[kBoxes];       // per test j: table entries it needs,
    int ncol[kBoxes];       // its columns (0: nothing to sample),
    int flag[kBoxes];       // 1: the batch must end in front of it, 4: (j = 0) outside the world boundary; later: its slot, -1 none
    int seg_start[8];       // segment s of the look-ahead (one inner loop of the reference): its first test,
    unsigned long long seg_cp[8];  // growths per direction before it,
    unsigned seg_cands[8];  // its candidate directions,
    int seg_ncand[8];       // their number,
    int seg_i[8];           // the index of the last one taken
    int seg_t0[8];          // and the inner loop's test index of its first test (1: the batch starts behind a pending growth)
    int verdict[8];         // tests in the batch, "test 0 fails the boundary", "test 0 alone", column chunks, tests of the look-ahead, "it reaches the expansion's end", segments
    int fail;               // first failing test of the batch (kAhead: none)
    int ntodo, ntodo2;      // chunks / column segments the free-space table could not clear (the filter pass of obstacle_in_batch)
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

// tests the recorded boxes (slots 0 .. A.nsamp-1); returns the test index of the first one holding an obstacle, kAhead if none
__device__ int obstacle_in_batch(const MapView& mp, Ahead& A, double margin) {
    const int J = A.nsamp;
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
    // the verdict is a minimum over the failing boxes.  Columns longer than kSeg samples (the whole-box re-test behind every failure:
    // 40 - 60 samples along its thin axis) are asked per SEGMENT of kSeg samples: a chunk next to an obstacle is next to it over a few
    // of its segments only, and a listed segment is ONE group of nearest-cell loads instead of a chain of nl / 4 dependent ones
    // (measured in a room with a few boxes: 40 of a wavefront's 47 memory round trips per corridor were such chains).
    constexpr int kSeg = 8;
    const int all_chunks = A.first[J];
    bool listed = false, from_back = false;
    int n_rounds = all_chunks;
    if (mp.sat != nullptr && margin <= mp.sat_margin && all_chunks >= 64 && all_chunks <= kTodo) {
        if (lane == 0) A.ntodo = 0, A.ntodo2 = 0;
        int zsh = 0;  // segments per column, rounded up to a power of two (uniform: J <= kSamp slots)
        for (int t = 0; t < J; t++) {
            const int nl = A.n[t][A.axes[t] & 3];
            while (nl > kSeg && (kSeg << zsh) < nl) zsh++;
        }
        // is item (chunk c, segment zs; zs < 0: the whole columns) provably free?  `skip`: the columns have no such segment
        auto item_free = [&](int c, int zs, bool& skip, bool& segmented) -> bool {
            const long long ti0_ = clock64();
            int j = 0;
            for (int t = 1; t < J; t++) j += (c >= A.first[t]) ? 1 : 0;  // (J <= kSamp slots)
            const int ax = A.axes[j];
            const int la = ax & 3, ca = (ax >> 2) & 3, cb = (ax >> 4) & 3;
            const int na = A.n[j][ca], nl = A.n[j][la];
            const int c0 = (c - A.first[j]) * 64, cl = A.cols[j] - 1;
            const int c1 = c0 + 63 < cl ? c0 + 63 : cl;
            // (c0, c1 < 2^22: a float quotient is off by at most one)
            const float rna = 1.0f / (float)na;
            int ib0 = (int)((float)c0 * rna), ib1 = (int)((float)c1 * rna);
            ib0 += (c0 - ib0 * na < 0) ? -1 : ((c0 - ib0 * na >= na) ? 1 : 0);
            ib1 += (c1 - ib1 * na < 0) ? -1 : ((c1 - ib1 * na >= na) ? 1 : 0);
            const int ia0 = ib0 == ib1 ? c0 - ib0 * na : 0, ia1 = ib0 == ib1 ? c1 - ib1 * na : na - 1;
            segmented = nl > kSeg;
            const bool whole = zs < 0 || !segmented;
            const int zfirst = whole ? 0 : zs * kSeg;
            skip = zs >= 0 && (segmented ? zfirst >= nl : zs > 0);
            if (skip) return true;
            const int zn = whole ? nl : (nl - zfirst < kSeg ? nl - zfirst : kSeg);
            const int ta = A.tab[j][ca], tb = A.tab[j][cb], tl = A.tab[j][la];
            int va0, va1, vb0, vb1, vl0, vl1;
            const int ra = cell_range(A.vtab, ta + ia0, ia1 - ia0 + 1, va0, va1), rb = cell_range(A.vtab, tb + ib0, ib1 - ib0 + 1, vb0, vb1);
            const int rl = cell_range(A.vtab, tl + zfirst, zn, vl0, vl1);
            bool free_ = ra == 2 || rb == 2 || rl == 2;  // (all of the item's samples lie beyond the map, far from the origin)
            if (!free_ && ra == 1 && rb == 1 && rl == 1) {  // (0: a sample outside the map that the exact path has to look at)
                const int x0 = ca == 0 ? va0 : (cb == 0 ? vb0 : vl0), x1 = ca == 0 ? va1 : (cb == 0 ? vb1 : vl1);
                const int y0 = ca == 1 ? va0 : (cb == 1 ? vb0 : vl0), y1 = ca == 1 ? va1 : (cb == 1 ? vb1 : vl1);
                const int z0 = ca == 2 ? va0 : (cb == 2 ? vb0 : vl0), z1 = ca == 2 ? va1 : (cb == 2 ? vb1 : vl1);
                const long long ti1_ = clock64();
                free_ = cells_free(mp, x0, x1, y0, y1, z0, z1);
                SFC_DBG(27, ti1_ - ti0_); SFC_DBG(28, clock64() - ti1_ + (free_ ? 0 : 0)); SFC_DBG(29, 1);
            }
            return free_;
        };
        __syncthreads();
        SFC_DBG(30, clock64() - tt1_);
        for (int c = lane; c < all_chunks; c += kSfcThreads) {  // first the chunks, whole columns: listed from the front of A.todo
            bool skip, segmented;
            if (!item_free(c, -1, skip, segmented)) A.todo[atomicAdd(&A.ntodo, 1)] = c;  // (all_chunks <= kTodo; chunks < 2^16)
        }
        __syncthreads();
        SFC_DBG(26, clock64() - tt1_);
        n_rounds = A.ntodo;
        listed = true;
        if (zsh > 0 && n_rounds > 0) {  // ... then the segments of the chunks that are left: listed from its back, as long as both lists fit
            const int n1 = n_rounds;
            for (int it = lane; it < (n1 << zsh); it += kSfcThreads) {
                const int c = A.todo[it >> zsh], zs = it & ((1 << zsh) - 1);
                bool skip, segmented;
                const bool free_ = item_free(c, zs, skip, segmented);
                if (!skip && !free_) {
                    const int slot = atomicAdd(&A.ntodo2, 1);
                    if (n1 + slot < kTodo) A.todo[kTodo - 1 - slot] = c | ((segmented ? zs + 1 : 0) << 16);
                }
            }
            __syncthreads();
            if (n1 + A.ntodo2 <= kTodo) n_rounds = A.ntodo2, from_back = true;  // (else: the chunks as listed, whole columns)
        }
        SFC_DBG(31, clock64() - tt1_);
        SFC_DBG(12, all_chunks); SFC_DBG(13, n_rounds); SFC_DBG(14, 1);
    } else {
        SFC_DBG(15, all_chunks);
    }
    for (int r = lane >> 6; r < n_rounds; r += kSfcThreads / 64) {
        const int stop = *(volatile int*)&A.fail;  // tests behind a failure already found need not be finished
        const int item = __builtin_amdgcn_readfirstlane(listed ? A.todo[from_back ? kTodo - 1 - r : r] : r);  // wave-uniform: boxes are padded to whole wavefronts
        // (a LISTED item packs the chunk into 16 bits -- the list is only built when all_chunks <= kTodo <= 65536 -- with the segment
        // selector above it; an unlisted round IS the chunk, whatever its size: nothing to unpack, nothing to truncate)
        const int chunk = listed ? (item & 0xffff) : item, zsel = listed ? (item >> 16) : 0;  // (zsel > 0: segment zsel - 1 of the columns only)
        const int idx = chunk * 64 + (lane & 63);
        // the box this chunk belongs to: lane t asks "does box t + 1 start at or before it" (one LDS read per lane instead of a scan)
        const int lt = lane & 63;
        const int j = __builtin_popcountll(__builtin_amdgcn_ballot_w64(lt + 1 < J && chunk >= A.first[lt + 1 < kSamp ? lt + 1 : kSamp]));
        const int jbox = __builtin_amdgcn_readfirstlane(A.sbox[j]);
        if (jbox > stop) continue;
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
        if (zsel > 0) {
            group(std::integral_constant<int, kSeg>{}, (zsel - 1) * kSeg);
        } else if (nl <= 2) {
            group(std::integral_constant<int, 2>{}, 0);
        } else if (nl <= 8) {
            group(std::integral_constant<int, 8>{}, 0);
        } else {
            for (int z0 = 0; z0 < nl; z0 += LSCSFC_GROUP) group(std::integral_constant<int, LSCSFC_GROUP>{}, z0);
        }
        if (live && hit) atomicMin(&A.fail, jbox);
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
    // x / n for 0 <= x < 512 and 1 <= n <= 6 as (x * ceil(1024 / n)) >> 10 (exact there; an integer division by a run-time value is ~40 instructions)
    auto kdiv = [](int n) -> int { return n == 1 ? 1024 : (n == 2 ? 512 : (n == 3 ? 342 : (n == 4 ? 256 : (n == 5 ? 205 : 171)))); };
    auto mod_of = [](int x, int n, int K) -> int { return x - n * ((x * K) >> 10); };
    auto cand_of = [](unsigned cs, int t) -> int { return (int)((cs >> (4 * t)) & 15u); };
    auto erase_of = [](unsigned cs, int t) -> unsigned { return (cs & ((1u << (4 * t)) - 1u)) | ((cs >> (4 * (t + 1))) << (4 * t)); };
    // growths per direction (one byte each) after m growths of an inner loop with candidates cs[0 .. nc) whose next candidate index is
    // `base` (in [0, nc]): growth u takes candidate (base + u) mod nc -- in closed form, the look-ahead asks for it at up to 62 growths
    auto growths_of = [&](unsigned cs, int nc, int K, int base, int m) -> unsigned long long {
        const int b0 = base >= nc ? base - nc : base;
        unsigned long long cp = 0;
#pragma unroll
        for (int sidx = 0; sidx < 6; sidx++) {
            int r = sidx - b0;  // the first growth that takes candidate sidx
            r += r < 0 ? nc : 0;
            const int cnt = (sidx < nc && m > r) ? (((m - 1 - r) * K) >> 10) + 1 : 0;
            cp += (unsigned long long)cnt << (8 * cand_of(cs, sidx));
        }
        return cp;
    };
    auto face = [](const BoxF& b, int d) -> float {
        return d == 0 ? b.lo[0] : (d == 1 ? b.lo[1] : (d == 2 ? b.lo[2] : (d == 3 ? b.hi[0] : (d == 4 ? b.hi[1] : b.hi[2]))));
    };
    // the box whose faces have been grown cp[d] times since the batch's face tables were made
    auto box_of = [&](unsigned long long cp, BoxF& bx) {
        for (int k = 0; k < 3; k++) {
            bx.lo[k] = A.F[k][(int)((cp >> (8 * k)) & 255ull)];
            bx.hi[k] = A.F[k + 3][(int)((cp >> (8 * (k + 3))) & 255ull)];
        }
    };
    // ... the same box as the accepted one, its candidate grown once more along d and the new layer (what `grow` leaves behind)
    auto with_pending = [&](unsigned long long cp, int d, BoxF& acc, BoxF& cand, BoxF& upd) {
        for (int k = 0; k < 3; k++) {
            const int cl = (int)((cp >> (8 * k)) & 255ull), ch = (int)((cp >> (8 * (k + 3))) & 255ull);
            acc.lo[k] = A.F[k][cl];
            acc.hi[k] = A.F[k + 3][ch];
            cand.lo[k] = upd.lo[k] = acc.lo[k];
            cand.hi[k] = upd.hi[k] = acc.hi[k];
            if (d == k) upd.hi[k] = acc.lo[k], cand.lo[k] = upd.lo[k] = A.F[k][cl + 1];
            if (d == k + 3) upd.lo[k] = acc.hi[k], cand.hi[k] = upd.hi[k] = A.F[k + 3][ch + 1];
        }
    };
    const double res = mp.res, rinv = 1.0 / res;
    // The reference's two nested loops (while candidates are left: re-test the whole box, then grow it round robin until a test fails,
    // erase the direction that failed) as ONE loop over batches of tests.  State between batches: the accepted box `sfc`, whether a
    // growth is PENDING (the candidate box is sfc grown once along cand[i] and its new layer is the next box to test; else the next
    // test is the whole box, the first one of an inner loop), the candidates and the index i of the last one taken.
    BoxF sfc = initial;
    bool pend = false;
    int i = -1;
    unsigned long long shift = 0;  // growths accepted since the face tables were based
    bool first = true;
    const int wave = threadIdx.x >> 6, wlane = threadIdx.x & 63;
    while (ncand > 0) {
        // The tests ahead: test j is what the loop condition sees after j tests have passed.  A face of the box only moves when its
        // own direction is grown, by the reference's float step each time, so face d after c growths is a table F[d][c] (a chain of
        // dependent float steps: kept across batches, shifted by what was accepted and extended at its end).  A test that fails on the
        // WORLD BOUNDARY is known in advance (pure arithmetic on the face: isSFCInBoundary, :810-817), so the sequence continues behind
        // it as the reference would: the pending growth is dropped, its direction erased, the whole box re-tested, the remaining
        // directions grown round robin -- the look-ahead is a list of SEGMENTS (one inner loop each), and only a test that finds an
        // OBSTACLE (or the batch's limits) ends the batch.  Lane j of the first wavefronts assembles box j from the tables and its
        // segment's parameters and sizes it; one wavefront then scans the <= kAhead tests, decides how many the batch holds and hands
        // out the record slots of those that have to be sampled.
        // (Growing the boxes one after the other in every lane -- dependent fp64 chains -- cost 1 600 cycles per test, as much as
        // testing them; batches of at most 63 tests that ended at every boundary failure: 8 batches per corridor in a room with a few
        // boxes, 3 of them ended by the boundary and 2 by the 63.)
        const long long tg0_ = clock64();
        const int KA = kAhead;
        __syncthreads();  // (the previous batch's readers of A are done)
        for (int d = wave; d < 6; d += kSfcThreads / 64) {  // one wavefront per direction: shift, extend, count
            int L = 1;
            if (first) {
                if (wlane == 0) A.F[d][0] = face(sfc, d);
            } else {
                L = A.flen[d];
                const int dl = (int)((shift >> (8 * d)) & 255ull);
                if (dl > 0) {
                    float mv[(kAhead + 2 + 63) / 64];
#pragma unroll
                    for (int q = 0; q < (kAhead + 2 + 63) / 64; q++) {
                        const int c = wlane + 64 * q;
                        mv[q] = (c + dl < L) ? A.F[d][c + dl] : 0.0f;
                    }
                    __builtin_amdgcn_wave_barrier();  // (one wavefront, LDS in program order: every read above precedes every write below)
#pragma unroll
                    for (int q = 0; q < (kAhead + 2 + 63) / 64; q++) {
                        const int c = wlane + 64 * q;
                        if (c + dl < L) A.F[d][c] = mv[q];
                    }
                    L -= dl;
                }
            }
            __builtin_amdgcn_wave_barrier();
            // Extension of the chain x(c + 1) = (float)((double)x(c) + step) to KA + 2 entries, in parallel: while the values stay in
            // one binade every step adds the same multiple of its ulp, so x(s + m) = x(s) + m (x(s + 1) - x(s)) is a GUESS that each
            // lane then checks with the reference's own statement -- entry c is accepted iff it equals the step from the guessed entry
            // c - 1; by induction everything in front of the first mismatch is the chain's value, and the next round starts there (a
            // handful of rounds: binade crossings near 0 and at 1, 2, 4 ... m).  As a dependent chain: ~85 cycles per entry on one lane.
            {
                const double stp = d < 3 ? -res : res;
                int sl = L - 1;  // last valid entry
                while (sl < KA + 1) {
                    const float xs = A.F[d][sl];
                    const float x1 = (float)((double)xs + stp);
                    const double dlt = (double)x1 - (double)xs;
                    int bad = KA + 2;  // first entry that is not confirmed
#pragma unroll
                    for (int q = 0; q < (kAhead + 2 + 63) / 64; q++) {
                        const int m = 1 + wlane + 64 * q, c = sl + m;
                        const float guess = (float)((double)xs + (double)m * dlt), before = (float)((double)xs + (double)(m - 1) * dlt);
                        const bool okc = c > KA + 1 || (float)((double)before + stp) == guess;
                        const unsigned long long mb = __builtin_amdgcn_ballot_w64(!okc);
                        const int fb = mb ? sl + 1 + 64 * q + (int)__builtin_ctzll(mb) : KA + 2;
                        bad = fb < bad ? fb : bad;
                        if (c <= KA + 1 && c < bad) A.F[d][c] = guess;  // (bad only shrinks: an entry behind an earlier mismatch is never written)
                    }
                    sl = bad - 1;  // (entry sl + 1 = x1 is always confirmed: progress)
                    __builtin_amdgcn_wave_barrier();
                }
            }
            L = L < KA + 2 ? KA + 2 : L;
            __builtin_amdgcn_wave_barrier();
            // growths of direction d the world boundary admits (the faces move monotonically)
            const double wb = d < 3 ? (double)mp.world_min(d) + 0.0 - 1e-5 : (double)mp.world_max(d - 3) - 0.0 + 1e-5;
            int g = 0;
            for (int q = 0; q < (kAhead + 1 + 63) / 64; q++) {
                const int c = 1 + wlane + 64 * q;
                const float x = A.F[d][c < KA + 2 ? c : 0];
                const bool okc = c <= KA + 1 && (d < 3 ? (double)x > wb : (double)x < wb);
                g += __builtin_popcountll(__builtin_amdgcn_ballot_w64(okc));
            }
            if (wlane == 0) A.gmax[d] = g < 255 ? g : 255, A.flen[d] = L;
        }
        if (threadIdx.x == 0) A.nsamp = 0, A.verdict[3] = 0;
        shift = 0;
        first = false;
        __syncthreads();
        SFC_DBG(21, clock64() - tg0_);
        // ---- phase A: lane j assembles and sizes box j (the segment list is uniform scalar work, repeated per wavefront)
        BoxF u = sfc;
        int n[3] = {0, 0, 0}, la = 0, ca = 1, cb = 2, ncol = 0;
        if (threadIdx.x < kBoxes) {
            const int j = threadIdx.x;
            unsigned long long gm = 0;
            for (int d = 0; d < 6; d++) gm |= (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(A.gmax[d]) << (8 * d);
            // the segments: where the world boundary fails a test, and the state behind it
            unsigned cs = (unsigned)__builtin_amdgcn_readfirstlane((int)cands);
            int nc = __builtin_amdgcn_readfirstlane(ncand), is = __builtin_amdgcn_readfirstlane(pend ? i - 1 : i), j0 = 0, sN = 0;
            int t0 = __builtin_amdgcn_readfirstlane(pend ? 1 : 0);
            unsigned long long cps = 0;
            bool done = false;
            unsigned my_cs = cs;  // lane j: the parameters of the segment test j belongs to
            int my_nc = nc, my_is = is, my_j0 = 0, my_t0 = t0;
            unsigned long long my_cp = 0;
            bool my_end = false;  // test j is the one the boundary fails
            for (int sg = 0; sg < 7; sg++) {
                if (j == 0) A.seg_start[sg] = j0, A.seg_cp[sg] = cps, A.seg_cands[sg] = cs, A.seg_ncand[sg] = nc, A.seg_i[sg] = is, A.seg_t0[sg] = t0;
                const int K = kdiv(nc);
                int b0 = is + 1;
                b0 = b0 >= nc ? b0 - nc : b0;
                // (a lane per candidate slot, then six v_readlane: as uniform scalar code the loop over the slots was a chain of ~200
                // dependent instructions per segment, 6 k cycles per batch where the z directions leave a flat world at once)
                const int sl = j & 63;
                const int dsl = cand_of(cs, sl < 6 ? sl : 0);
                int rsl = sl - b0;  // the first growth that takes candidate sl
                rsl += rsl < 0 ? nc : 0;
                const int gsl = (int)((gm >> (8 * dsl)) & 255ull) - (int)((cps >> (8 * dsl)) & 255ull);  // growths its direction has left
                const int ufl = sl < nc ? rsl + gsl * nc : (1 << 20);
                int ubest = 1 << 20;  // the first growth of this inner loop whose face leaves the world
#pragma unroll
                for (int sidx = 0; sidx < 6; sidx++) {
                    const int uf = __builtin_amdgcn_readlane(ufl, sidx);
                    ubest = uf < ubest ? uf : ubest;
                }
                const int jend = j0 + ubest + 1 - t0;  // the test that fails: growth ubest makes the inner loop's test ubest + 1
                const bool in_seg = j >= j0;            // (later segments overwrite)
                my_cs = in_seg ? cs : my_cs, my_nc = in_seg ? nc : my_nc, my_is = in_seg ? is : my_is, my_j0 = in_seg ? j0 : my_j0;
                my_t0 = in_seg ? t0 : my_t0;
                my_cp = in_seg ? cps : my_cp;
                my_end = in_seg ? (j == jend) : my_end;
                if (jend >= KA) break;  // no failure on the boundary within the look-ahead: the segment runs to its end
                {   // ubest growths accepted (the lane's slot: how many of them), the pending one dropped,
                    const int cnt = (sl < nc && ubest > rsl) ? (((ubest - 1 - rsl) * K) >> 10) + 1 : 0;
                    const unsigned long long part = (unsigned long long)cnt << (8 * dsl);
                    const int plo = (int)(unsigned)part, phi = (int)(unsigned)(part >> 32);
#pragma unroll
                    for (int sidx = 0; sidx < 6; sidx++)
                        cps += (unsigned long long)(unsigned)__builtin_amdgcn_readlane(plo, sidx) | ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(phi, sidx) << 32);
                }
                const int ifail = mod_of(is + ubest + 1, nc, K);
                cs = erase_of(cs, ifail);                 // its direction erased
                nc--;
                is = ifail > 0 ? ifail - 1 : nc - 1;
                j0 = jend + 1;
                t0 = 0;
                sN = sg + 1;
                if (nc == 0) {
                    done = true;  // ... and with the last direction gone the expansion ends behind test jend
                    break;
                }
            }
            const int Jtot = done ? j0 : KA;  // tests the look-ahead holds
            if (j == 0 && done) A.seg_start[sN] = j0, A.seg_cp[sN] = cps, A.seg_cands[sN] = cs, A.seg_ncand[sN] = 0, A.seg_i[sN] = is, A.seg_t0[sN] = 0;
            if (j == 0) A.verdict[4] = Jtot, A.verdict[5] = done ? 1 : 0, A.verdict[6] = done ? sN : sN + 1;  // (segments that hold tests)
            SFC_DBG(22, clock64() - tg0_);
            // box j: test t of its inner loop -- t = 0 the whole box, else the layer of growth t, behind t - 1 accepted ones
            const bool mine = j < Jtot && tables;
            const int t = j - my_j0 + my_t0;
            if (mine) {
                const int K = kdiv(my_nc);
                const unsigned long long cp = my_cp + (t >= 1 ? growths_of(my_cs, my_nc, K, my_is + 1, t - 1) : 0ull);
                const int d = t >= 1 ? cand_of(my_cs, mod_of(my_is + t, my_nc, K)) : -1;
                for (int k = 0; k < 3; k++) {
                    const int cl = (int)((cp >> (8 * k)) & 255ull), ch = (int)((cp >> (8 * (k + 3))) & 255ull);
                    u.lo[k] = A.F[k][cl];
                    u.hi[k] = A.F[k + 3][ch];
                    if (d == k) u.hi[k] = u.lo[k], u.lo[k] = A.F[k][cl + 1];
                    if (d == k + 3) u.lo[k] = u.hi[k], u.hi[k] = A.F[k + 3][ch + 1];
                }
            }
            SFC_DBG(23, clock64() - tg0_);
            const bool inb = in_boundary(mp, u, 0);
            // (the boundary test of every box must agree with the segment list -- if one ever did not, the batch ends in front of it and
            // it is test 0 of the next batch, which is decided on its own)
            // (it does happen: a corridor seeded OUTSIDE the world -- an inverted box whose upper face lies below the world's lower bound --
            // fails the boundary on the layer's inner face, which the list does not look at; tools/sweep_corridors.py seed 3, agent 197)
#ifdef LSCSFC_TEST_MISM  // testing knob: treat test number LSCSFC_TEST_MISM of every batch as such a disagreement -- the boxes must not change
            const bool mism = mine && j > 0 && ((my_end != !inb) || j == LSCSFC_TEST_MISM);
#else
            const bool mism = mine && j > 0 && (my_end != !inb);
#endif
            for (int k = 0; k < 3; k++) n[k] = (int)floor_div((double)(u.hi[k] - u.lo[k]) + 1e-5, res, rinv) + 1;
            // an inverted box (a hull clipped to a previous box it does not touch) has no sample points
            bool empty = n[0] <= 0 || n[1] <= 0 || n[2] <= 0;
            // a box the boundary fails is not looked at (the reference's test is `obstacle || !boundary`); a box the map's summary
            // proves free passes like an empty one: no table entries, no columns (lscqp_map_prepare)
            empty = empty || !inb || (mine && surely_free(mp, margin, u.lo[0], u.lo[1], u.lo[2], n[0], n[1], n[2]));
            SFC_DBG(24, clock64() - tg0_);
            // thin axis: the columns run along it (ties: the later axis, so that x stays a column axis)
            la = (n[2] <= n[1] && n[2] <= n[0]) ? 2 : (n[1] <= n[0] ? 1 : 0);
            ca = la == 0 ? 1 : 0, cb = la == 2 ? 1 : 2;
            const int64_t ncol64 = empty ? 0 : (la == 2 ? (int64_t)n[0] * n[1] : (la == 1 ? (int64_t)n[0] * n[2] : (int64_t)n[1] * n[2]));
            const bool over = !empty && (n[0] > kSfcThreads || n[1] > kSfcThreads || n[2] > kSfcThreads || ncol64 > (1 << 22));
            ncol = (mine && !over) ? (int)ncol64 : 0;
            A.need[j] = (mine && !empty) ? n[0] + n[1] + n[2] : 0;
            A.ncol[j] = ncol;
            A.flag[j] = (mine ? 0 : 1) | ((mine && (mism || over)) ? 1 : 0) | ((mine && j == 0 && !inb) ? 4 : 0);
        }
        __syncthreads();
        SFC_DBG(25, clock64() - tg0_);
        // ---- phase B (wavefront 0): lane l scans tests kGenWaves * l ..; the batch ends at once if test 0 is outside the world boundary,
        // in front of the first box beyond the batch's limits (test 0 beyond them is decided on its own, the sequential way), and behind
        // the kSamp-th box that has to be sampled: tests behind the first failure are wasted work, boxes the free-space table passes are not
        if (threadIdx.x < 64) {
            constexpr int PB = kGenWaves;
            const int l = threadIdx.x;
            int ch_[PB], nd_[PB], fl_[PB], nc_[PB];
            int sc = 0, su = 0, ss = 0;
#pragma unroll
            for (int q = 0; q < PB; q++) {
                const int j = PB * l + q;
                nc_[q] = A.ncol[j], nd_[q] = A.need[j], fl_[q] = A.flag[j];
                ch_[q] = (nc_[q] + 63) >> 6;
                sc += ch_[q], su += nd_[q], ss += nc_[q] > 0 ? 1 : 0;
            }
            // inclusive prefix sums over the lanes: DPP row shifts inside the rows of 16 lanes, then the row broadcasts (12 VALU
            // instructions per sum; as six ds_bpermute steps each: ~600 cycles per sum on the batch's critical path)
            auto wave_scan = [](int v) -> int {
                v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);  // row_shr:1
                v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);  // row_shr:2
                v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);  // row_shr:4
                v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);  // row_shr:8
                v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1 and 3
                v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2 and 3
                return v;
            };
            const int pc = wave_scan(sc), pu = wave_scan(su), ps = wave_scan(ss);
            int ec = pc - sc, eu = pu - su, es = ps - ss;  // what lies before the lane's first test
            int jmine = 1 << 20;
            bool brk0 = false;
            int ec_[PB], eu_[PB], es_[PB];
#pragma unroll
            for (int q = 0; q < PB; q++) {
                const int j = PB * l + q;
                ec_[q] = ec, eu_[q] = eu, es_[q] = es;
                const bool brk = (fl_[q] & 1) != 0 || (int64_t)ec * 64 + nc_[q] > (1 << 22) || eu + nd_[q] > kTab;
                const bool stopj = brk || es >= kSamp;
                jmine = (stopj && j < jmine) ? j : jmine;
                brk0 = brk0 || (j == 0 && brk);
                ec += ch_[q], eu += nd_[q], es += nc_[q] > 0 ? 1 : 0;
            }
            const unsigned long long mlim = __builtin_amdgcn_ballot_w64(jmine < (1 << 20));
            const int jlim = mlim ? __builtin_amdgcn_readlane(jmine, __builtin_ctzll(mlim)) : kBoxes;  // (tests >= Jtot carry flag 1)
            const bool stop0 = (__builtin_amdgcn_readfirstlane(fl_[0]) & 4) != 0;
            const bool brk_at_0 = __builtin_amdgcn_readfirstlane(brk0 ? 1 : 0) != 0;
            const int J_ = tables ? (stop0 ? 0 : jlim) : 0;
            const bool alone_ = !tables || (brk_at_0 && !stop0);
            int mys = 0, myc = 0;
#pragma unroll
            for (int q = 0; q < PB; q++) {
                const int j = PB * l + q;
                const bool rec = j < J_ && nc_[q] > 0;
                if (rec) {
                    const int k = es_[q];
                    A.sbox[k] = j, A.first[k] = ec_[q], A.first[k + 1] = ec_[q] + ch_[q], A.tab[k][0] = eu_[q];
                    mys++, myc += ch_[q];
                }
                A.flag[j] = rec ? es_[q] : -1;
            }
            if (mys > 0) atomicAdd(&A.nsamp, mys), atomicAdd(&A.verdict[3], myc);
            if (l == 0) {
                A.fail = kAhead;
                A.verdict[0] = J_, A.verdict[1] = (tables && stop0) ? 0 : kAhead, A.verdict[2] = alone_ ? 1 : 0;
            }
        }
        __syncthreads();
        // ---- phase C: the lanes of the boxes that have to be sampled fill their record slots
        if (threadIdx.x < kBoxes) {
            const int k = A.flag[threadIdx.x];
            if (k >= 0) {
                for (int m = 0; m < 3; m++) A.lo[k][m] = u.lo[m], A.n[k][m] = n[m];
                const int eu = A.tab[k][0];
                A.tab[k][1] = eu + n[0], A.tab[k][2] = eu + n[0] + n[1];
                A.axes[k] = la | (ca << 2) | (cb << 4);
                A.cols[k] = ncol;
            }
        }
        __syncthreads();
        int J = A.verdict[0], jstop = A.verdict[1];
        const bool alone = A.verdict[2] != 0;
        const int chunks = A.verdict[3], Jtot = A.verdict[4], nseg = A.verdict[6];
        const bool done = A.verdict[5] != 0;
        SFC_DBG(9, clock64() - tg0_);
        SFC_DBG(0, 1); SFC_DBG(1, J); SFC_DBG(2, chunks); SFC_DBG(3, alone ? 1 : 0);
        const long long t0_ = clock64();
        int jfail;
        if (alone) {  // test 0 on its own, the sequential way
            BoxF acc, cand, upd;
            with_pending(0ull, pend ? cand_of(cands, i) : -1, acc, cand, upd);
            if (!in_boundary(mp, upd, 0)) {
                jfail = 0;
            } else {
                jfail = obstacle_in(mp, upd, margin) ? 0 : kAhead;
            }
            J = 1;
            jstop = kAhead;
        } else {
            jfail = (J > 0 && A.nsamp > 0) ? obstacle_in_batch(mp, A, margin) : kAhead;
        }
        SFC_DBG(4, clock64() - t0_);
        const int jf = jfail < jstop ? jfail : jstop;  // the test of this batch that ends it by failing, kAhead if none does
        const bool failed = jf < kAhead && (jf < J || (jf == 0 && J == 0));
        // (how batches end: 16 test 0 on the world boundary, 17 obstacle, 18 cut by the batch's limits, 19 tests passed or skipped, 20 the expansion's end)
        SFC_DBG(16, (failed && jstop <= jfail) ? 1 : 0); SFC_DBG(17, (failed && jfail < jstop) ? 1 : 0);
        SFC_DBG(18, failed ? 0 : 1); SFC_DBG(19, failed ? jf : J); SFC_DBG(20, (!failed && !alone && done && J == Jtot) ? 1 : 0);
        const long long tr0_ = clock64();
        if (!failed && !alone && done && J == Jtot) {  // every test up to the expansion's end was in the batch
            box_of(A.seg_cp[nseg], sfc);
            ncand = 0;
            continue;
        }
        // where the batch ends: test jq -- the failing one, or the first one not in the batch (a test that passed on its own: the next)
        const int jq = failed ? jf : J;
        int sq = 0;
        for (int q = 1; q < 7; q++) sq = (q < nseg && jq >= A.seg_start[q]) ? q : sq;
        const int tq = jq - A.seg_start[sq] + A.seg_t0[sq];  // its test index in its inner loop
        cands = A.seg_cands[sq], ncand = A.seg_ncand[sq];
        const int isq = A.seg_i[sq], Kq = kdiv(ncand);
        // tq - 1 growths of the inner loop accepted before it (tq = 0: it is the whole-box test)
        const unsigned long long cpq = A.seg_cp[sq] + (tq >= 1 ? growths_of(cands, ncand, Kq, isq + 1, tq - 1) : 0ull);
        box_of(cpq, sfc);
        shift = cpq;
        i = tq >= 1 ? mod_of(isq + tq, ncand, Kq) : isq;
        pend = tq >= 1;
        if (failed) {  // the growth under test dropped (tq = 0: the whole-box test itself failed), its direction erased
            if (i < 0) return false;  // initial box outside the world: the reference erases begin() - 1 here (undefined)
            cands = erase_of(cands, i);
            ncand--;
            i = (i > 0) ? i - 1 : ncand - 1;
            pend = false;
        }
        SFC_DBG(10, clock64() - tr0_);
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
    if ((int64_t)blockIdx.x >= n) return;
    const int64_t a = mp.order ? (int64_t)mp.order[blockIdx.x] : (int64_t)blockIdx.x;
    const long long t_begin_ = clock64();
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
        if (threadIdx.x == 0 && mp.cost) {
            const unsigned long long dtc = (unsigned long long)(clock64() - t_begin_) >> 4;
            mp.cost[a] = dtc > 0xffffffffull ? 0xffffffffu : (uint32_t)dtc;
        }
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
    if (threadIdx.x == 0 && mp.cost) {
        const unsigned long long dtc = (unsigned long long)(clock64() - t_begin_) >> 4;
        mp.cost[a] = dtc > 0xffffffffull ? 0xffffffffu : (uint32_t)dtc;
    }
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
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(lscsfc::sfc_dbg), sizeof(unsigned long long) * 32);
    if (reset) { unsigned long long z[32] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(lscsfc::sfc_dbg), z, sizeof z); }
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

int lscqp_construct_sfc_raw_ex_(lscqp_map mp, int mode, int M, int64_t n, const double* d_points, const double* d_radius, lscqp_box* d_sfc,
                                int32_t* d_status_out, const int32_t* d_order, uint32_t* d_cost_out, void* stream);
int lscqp_construct_sfc_raw_(lscqp_map mp, int mode, int M, int64_t n, const double* d_points, const double* d_radius, lscqp_box* d_sfc,
                             int32_t* d_status_out, void* stream) {
    return lscqp_construct_sfc_raw_ex_(mp, mode, M, n, d_points, d_radius, d_sfc, d_status_out, nullptr, nullptr, stream);
}
int lscqp_construct_sfc_raw_ex_(lscqp_map mp, int mode, int M, int64_t n, const double* d_points, const double* d_radius, lscqp_box* d_sfc,
                                int32_t* d_status_out, const int32_t* d_order, uint32_t* d_cost_out, void* stream) {
    lscsfc::MapView v;
    v.order = d_order;
    v.cost = d_cost_out;
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
