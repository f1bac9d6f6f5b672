// One kernel instance per translation unit so the instances compile in parallel.
// Built with -DLSCQP_M=<M> -DLSCQP_DIM=<dim> -DLSCQP_ES=<0|1>; exports lscqp_launch_<M>_<dim>_<ES>.
#include "lscqp_kernel.hpp"
#include "lscqp_launch.hpp"

#define LSCQP_CAT_(a, b, c, d) a##b##_##c##_##d
#define LSCQP_CAT(a, b, c, d) LSCQP_CAT_(a, b, c, d)
#define LSCQP_FN LSCQP_CAT(lscqp_launch_, LSCQP_M, LSCQP_DIM, LSCQP_ES)

extern "C" hipError_t LSCQP_FN(const lscqp::DevClass* cls, int64_t n, const lscqp_header* hdr, const lscqp_row* rows,
                               const uint64_t* row_offsets, const lscqp_box* sfc, double* x_out, double* obj_out,
                               int32_t* status_out, lscqp_info* info_out, hipStream_t stream) {
    using C = lscqp::Cfg<LSCQP_M, LSCQP_DIM, (LSCQP_ES != 0)>;
    auto kern = lscqp::lscqp_pdip_kernel<LSCQP_M, LSCQP_DIM, (LSCQP_ES != 0)>;
    const size_t lds = C::lds_bytes(cls->n_obs_max);
    if (lds > lscqp::kMaxLdsBytes) return hipErrorInvalidValue;
    static size_t lds_set = 0;  // raise the dynamic-LDS cap once per size (160 KiB per CU on gfx950)
    if (lds > lds_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_set = lds;
    }
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(kern, dim3((unsigned)n), dim3(64), lds, stream, *cls, n, hdr, rows, row_offsets, sfc, x_out, obj_out,
                       status_out, info_out);
    return hipGetLastError();
}
