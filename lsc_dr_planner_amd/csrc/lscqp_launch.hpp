// Launch table shared by the API translation unit and the per-instance translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lscqp.h"

namespace lscqp {
struct DevClass;
constexpr size_t kMaxLdsBytes = 160 * 1024;  // LDS per CU on gfx950

using launch_fn = hipError_t (*)(const DevClass*, int64_t, const lscqp_header*, const lscqp_row*, const uint64_t*,
                                 const lscqp_box*, const double*, double*, double*, int32_t*, lscqp_info*, hipStream_t);
}  // namespace lscqp

// The compiled kernel instances, X(M, DIM, ES, NSLOT, W, MIXED):
//   W = wavefronts per QP; dim*(3M-2) <= 64*W is required by the lane-per-row factorisation (W = 2: M = 10 in 3-D);
//   W = 2 instances of shapes with nz <= 64 are the low-latency variants for small batches (lscqp_api.hip policy);
//   the nz = 84 class (one workgroup per CU by LDS anyway) also has a W = 4 instance: 2.6 -> 1.6 ms per 128-QP batch;
//   NSLOT = LSC row slots per lane (registers); an instance serves up to NSLOT * floor(64*W/(6M-3)) obstacles per agent;
//   round 3: the end-stop-free (DLSC / BVC / RSFC: ES = 0) classes of every horizon whose reduced system fits one wavefront (3 dim M <= 64),
//   M = 6, 8, 9 in 2-D, M = 8 in 3-D (nz = 66: nested dissection, two equal blocks of 18), and by nested dissection over UNEQUAL blocks
//   M = 9 in 3-D (nz = 75) and the end-stop-free 3-D classes at M = 8, 9, 10 (nz = 72, 81, 90); what is still missing -- M = 11, 12,
//   neighbour counts beyond the slots -- runs on the run-time-shaped kernel (lscqp_generic.hip); every (M <= 10, dim, ES) has an instance;
//   MIXED = 1: float32 factorisation / substitutions (LSCQP_PRECISION_MIXED, BASELINE configs[4]), one wavefront per QP.
#define LSCQP_INSTANCES(X)                                                                                              \
    X(5, 3, 1, 10, 1, 0) X(5, 3, 1, 24, 1, 0) X(6, 3, 1, 20, 1, 0) X(7, 3, 1, 12, 1, 0) X(4, 3, 1, 12, 1, 0) X(3, 3, 1, 8, 1, 0) X(2, 3, 1, 8, 1, 0) \
    X(10, 2, 1, 10, 1, 0) X(10, 2, 1, 24, 1, 0) X(8, 2, 1, 12, 1, 0) X(5, 2, 1, 12, 1, 0)                                           \
    X(5, 3, 0, 10, 1, 0) X(5, 2, 0, 12, 1, 0) X(10, 2, 0, 10, 1, 0)                                                               \
    X(2, 3, 0, 8, 1, 0) X(3, 3, 0, 8, 1, 0) X(4, 3, 0, 12, 1, 0) X(6, 3, 0, 7, 2, 0) X(7, 3, 0, 12, 1, 0)                           \
    X(8, 2, 0, 12, 1, 0) X(9, 2, 1, 12, 1, 0) X(9, 2, 0, 12, 1, 0) X(6, 2, 1, 12, 1, 0) X(6, 2, 0, 12, 1, 0) X(8, 3, 1, 12, 2, 0)   \
    X(2, 2, 1, 8, 1, 0) X(2, 2, 0, 8, 1, 0) X(3, 2, 1, 8, 1, 0) X(3, 2, 0, 8, 1, 0) X(4, 2, 1, 12, 1, 0) X(4, 2, 0, 12, 1, 0) X(7, 2, 1, 12, 1, 0) X(7, 2, 0, 12, 1, 0) \
    X(5, 3, 1, 8, 4, 0) X(5, 3, 1, 12, 4, 0) X(5, 3, 0, 12, 2, 0) X(5, 3, 0, 8, 4, 0) \
    X(10, 2, 0, 10, 4, 0) X(10, 2, 0, 5, 2, 0) X(6, 3, 1, 8, 4, 0) X(6, 3, 0, 8, 4, 0) \
    X(10, 3, 0, 7, 2, 0) X(10, 3, 0, 9, 4, 0) X(9, 3, 0, 7, 2, 0) X(9, 3, 0, 8, 4, 0) X(9, 3, 1, 7, 2, 0) X(9, 3, 1, 8, 4, 0) X(8, 3, 0, 7, 2, 0) X(8, 3, 0, 8, 4, 0) \
    X(10, 3, 1, 20, 2, 0) X(10, 2, 1, 20, 2, 0) X(5, 3, 1, 5, 2, 0) X(5, 3, 1, 12, 2, 0) X(6, 3, 1, 7, 2, 0) X(10, 2, 1, 5, 2, 0) X(10, 3, 1, 10, 4, 0) X(10, 2, 1, 10, 4, 0) \
    X(5, 3, 1, 10, 1, 1) X(5, 3, 1, 24, 1, 1) X(6, 3, 1, 20, 1, 1) X(10, 2, 1, 10, 1, 1)
