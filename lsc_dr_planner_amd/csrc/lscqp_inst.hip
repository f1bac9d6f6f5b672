// One kernel instance per translation unit so the instances compile in parallel.
// Built with -DLSCQP_M=<M> -DLSCQP_DIM=<dim> -DLSCQP_ES=<0|1> -DLSCQP_NSLOT=<slots> -DLSCQP_W=<wavefronts per QP>
// -DLSCQP_MIXED=<0|1>; exports lscqp_launch_<M>_<dim>_<ES>_<NSLOT>_<W>_<MIXED>.
#include <atomic>

#include "lscqp_kernel.hpp"
#include "lscqp_launch.hpp"

#ifndef LSCQP_W
#define LSCQP_W 1
#endif
#ifndef LSCQP_MIXED
#define LSCQP_MIXED 0
#endif
#define LSCQP_CAT_(a, b, c, d, e, f, g) a##b##_##c##_##d##_##e##_##f##_##g
#define LSCQP_CAT(a, b, c, d, e, f, g) LSCQP_CAT_(a, b, c, d, e, f, g)
#define LSCQP_FN LSCQP_CAT(lscqp_launch_, LSCQP_M, LSCQP_DIM, LSCQP_ES, LSCQP_NSLOT, LSCQP_W, LSCQP_MIXED)
#if LSCQP_MIXED
typedef float lscqp_factor_t;
#else
typedef double lscqp_factor_t;
#endif

extern "C" hipError_t LSCQP_FN(const lscqp::DevClass* cls, int64_t n, const lscqp_header* hdr, const lscqp_row* rows,
                               const uint64_t* row_offsets, const lscqp_box* sfc, const double* x_init, double* x_out,
                               double* obj_out, int32_t* status_out, lscqp_info* info_out, hipStream_t stream) {
    using C = lscqp::Cfg<LSCQP_M, LSCQP_DIM, (LSCQP_ES != 0), LSCQP_NSLOT, LSCQP_W, (int)sizeof(lscqp_factor_t)>;
    auto kern = lscqp::lscqp_pdip_kernel<LSCQP_M, LSCQP_DIM, (LSCQP_ES != 0), LSCQP_NSLOT, LSCQP_W, lscqp_factor_t>;
    constexpr size_t lds = C::lds_bytes();
    static_assert(lds <= lscqp::kMaxLdsBytes, "instance does not fit the LDS of one CU");
    if (cls->n_obs_max > C::MAX_OBS) return hipErrorInvalidValue;
    // raise the dynamic-LDS cap (160 KiB per CU on gfx950) once PER DEVICE: the attribute belongs to the device's copy of the
    // kernel, and one process may drive several GPUs (lscqp_comm_*)
    static std::atomic<bool> attr_set[64];  // (zero-initialised; two host threads may launch for the first time at once: setting it twice is harmless)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set[dev].store(true, std::memory_order_release);
    }
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(kern, dim3((unsigned)n), dim3(C::T), lds, stream, *cls, n, hdr, rows, row_offsets, sfc, x_init, x_out,
                       obj_out, status_out, info_out);
    return hipGetLastError();
}
