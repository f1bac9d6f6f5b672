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
    // PERSIST (lscqp_kernel.hpp): the instances whose register budget takes the work-queue loop without spilling -- the multi-wavefront
    // forms of the end-stop classes (every BASELINE shape that runs more than one round of workgroups: configs[0], [2], [3]) -- are
    // compiled in BOTH forms: the loop costs the one-instance-per-workgroup launches 1 - 3 % (arguments live across the body; A/B
    // against -DLSCQP_PERSIST=0: the 64-QP headline 0.1069 -> 0.1094 ms), so only launches that exceed the chip take it.  The others
    // keep one instance per workgroup (tools/instance_resources.py after a build shows who would spill).
#ifndef LSCQP_PERSIST
#define LSCQP_PERSIST (LSCQP_W >= 2 && LSCQP_NSLOT <= 10 && LSCQP_ES != 0 && LSCQP_M != 9 && !LSCQP_MIXED)
#endif
    constexpr bool kPersist = LSCQP_PERSIST;
    auto kern = lscqp::lscqp_pdip_kernel<LSCQP_M, LSCQP_DIM, (LSCQP_ES != 0), LSCQP_NSLOT, LSCQP_W, lscqp_factor_t, false>;
    auto kernp = lscqp::lscqp_pdip_kernel<LSCQP_M, LSCQP_DIM, (LSCQP_ES != 0), LSCQP_NSLOT, LSCQP_W, lscqp_factor_t, kPersist>;
    constexpr size_t lds = C::lds_bytes();
    static_assert(lds <= lscqp::kMaxLdsBytes, "instance does not fit the LDS of one CU");
    if (cls->n_obs_max > C::MAX_OBS) return hipErrorInvalidValue;
    // raise the dynamic-LDS cap (160 KiB per CU on gfx950) once PER DEVICE: the attribute belongs to the device's copy of the
    // kernel, and one process may drive several GPUs (lscqp_comm_*)
    static std::atomic<bool> attr_set[64];  // (zero-initialised; two host threads may launch for the first time at once: setting it twice is harmless)
    static std::atomic<int> resident[64];   // workgroups of the persistent form the device holds at once
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        int cap0 = 0;
        if (kPersist) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernp), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            // (registers and LDS decide: 1 - 4 workgroups per CU; asked of the runtime once per device)
            int per_cu = 0, cus = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kernp), C::T, lds) != hipSuccess || per_cu < 1) per_cu = 1;
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 1;
            cap0 = per_cu * cus;
        }
        resident[dev].store(cap0, std::memory_order_release);
        attr_set[dev].store(true, std::memory_order_release);
    }
    if (n <= 0) return hipSuccess;
    // A launch with more instances than the device holds at once runs PERSISTENT over the caller's work queue (cls->queue, zeroed by the
    // API on this stream): exactly that many workgroups, each taking instance after instance -- see lscqp_pdip_kernel.  Smaller launches
    // (and the instances without a persistent form) keep one instance per workgroup and no queue.
    const int cap = resident[dev].load(std::memory_order_acquire);
    lscqp::DevClass c = *cls;
    if (kPersist && c.scan && cap > 0) {
        // behind the dual active-set phase (lscqp_kernel.hpp: DevClass::scan): at most `cap` workgroups scan the statuses for what the phase left
        c.queue = nullptr;
        c.order = nullptr;
        const int64_t g = n < (int64_t)cap ? n : (int64_t)cap;
        hipLaunchKernelGGL(kernp, dim3((unsigned)g), dim3(C::T), lds, stream, c, n, hdr, rows, row_offsets, sfc, x_init, x_out, obj_out, status_out,
                           info_out);
    } else if (kPersist && c.queue != nullptr && cap > 0 && n > (int64_t)cap) {
        c.scan = 0;
        hipLaunchKernelGGL(kernp, dim3((unsigned)cap), dim3(C::T), lds, stream, c, n, hdr, rows, row_offsets, sfc, x_init, x_out, obj_out, status_out,
                           info_out);
    } else {
        c.queue = nullptr;
        c.scan = 0;
        hipLaunchKernelGGL(kern, dim3((unsigned)n), dim3(C::T), lds, stream, c, n, hdr, rows, row_offsets, sfc, x_init, x_out, obj_out, status_out,
                           info_out);
    }
    return hipGetLastError();
}
