// Launch table shared by the API translation unit and the per-instance translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lscqp.h"

namespace lscqp {
struct DevClass;
constexpr size_t kMaxLdsBytes = 160 * 1024;  // LDS per CU on gfx950

using launch_fn = hipError_t (*)(const DevClass*, int64_t, const lscqp_header*, const lscqp_row*, const uint64_t*,
                                 const lscqp_box*, double*, double*, int32_t*, lscqp_info*, hipStream_t);
}  // namespace lscqp

// The list of compiled (M, dim, end_stop) kernel instances.  X(M, DIM, ES).
// dim*(3M-2) <= 64 is required by the lane-per-row factorisation.
#define LSCQP_INSTANCES(X) \
    X(2, 3, 1) X(3, 3, 1) X(4, 3, 1) X(5, 3, 1) X(6, 3, 1) X(7, 3, 1) \
    X(5, 2, 1) X(8, 2, 1) X(10, 2, 1) \
    X(5, 3, 0) X(5, 2, 0)
