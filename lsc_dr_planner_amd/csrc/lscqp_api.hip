// lscqp_api.hip — C ABI of the batched trajectory-QP solver (include/lscqp.h).  Host side, HIP runtime only.
// There is NO CPU fallback: without a HIP device every solve call fails loudly with LSCQP_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "lscqp_kernel.hpp"
#include "lscqp_launch.hpp"
#include "lscqp_staging.hpp"

#define LSCQP_DECL(M, D, E, S, W, X)                                                                                      \
    extern "C" hipError_t lscqp_launch_##M##_##D##_##E##_##S##_##W##_##X(const lscqp::DevClass*, int64_t, const lscqp_header*, \
                                                                  const lscqp_row*, const uint64_t*, const lscqp_box*,  \
                                                                  const double*, double*, double*, int32_t*, lscqp_info*, \
                                                                  hipStream_t);
LSCQP_INSTANCES(LSCQP_DECL)
#undef LSCQP_DECL

// the run-time-shaped instance (lscqp_generic.hip): every (M, dim, planner mode) / neighbour count the compiled table does not serve
extern "C" int lscqp_generic_supports(int M, int dim, int es);
extern "C" int lscqp_generic_max_obstacles(int M, int dim, int es);
extern "C" size_t lscqp_generic_lds_bytes(int M, int dim, int es, int n_obs_max);
extern "C" hipError_t lscqp_launch_generic(const lscqp::DevClass* cls, int M, int dim, int es, int64_t n, const lscqp_header* hdr, const lscqp_row* rows,
                                           const uint64_t* row_offsets, const lscqp_box* sfc, const double* x_init, double* x_out, double* obj_out,
                                           int32_t* status_out, lscqp_info* info_out, hipStream_t stream);

// the dual active-set phase (lscqp_das.hip)
extern "C" size_t lscqp_das_build_tables(int M, int es, double dt, double w_c, double w_t, double* out);
extern "C" size_t lscqp_das_build_pairs(int M, int dim, int comm_on, int32_t* out);
extern "C" size_t lscqp_das_lds_bytes(int M, int dim, int kmax, int cacheC, int stage_rows);
extern "C" hipError_t lscqp_launch_das(const lscqp::DevClass* cls, int M, int dim, int es, int cap, int threads, int kmax, int max_steps, int cacheC,
                                       int stage_rows, int screen, const double* d_tab, int64_t n, const lscqp_header* hdr, const lscqp_row* rows, const uint64_t* row_offsets,
                                       const lscqp_box* sfc, const double* x_init, double* x_out, double* obj_out, int32_t* status_out,
                                       lscqp_info* info_out, hipStream_t stream);
extern "C" int lscqp_generate_lsc_raw_(int mode, int M, int dim, int64_t n_agents, int32_t n_obs, int64_t first_agent,
                                       const double* d_traj, const double* d_own_traj, const int32_t* d_neighbours, const double* d_radius,
                                       const double* d_downwash, const double* d_goal, const double* d_goal_all, int rows_f32,
                                       int32_t n_obs_total, int32_t slot0, lscqp_row* d_rows_out, void* stream);
extern "C" int lscqp_shift_traj_partial_raw_(int M, int dim, int64_t n, const double* w36, double z_2d, const double* d_x_prev, double* d_traj,
                                             void* stream);
extern "C" int lscqp_generate_lsc_obstacles_raw_(int M, int dim, double dt, const lscqp_obstacle_param* p, int64_t n_agents, int32_t n_dyn,
                                                 int64_t first_agent, const double* d_traj, const int32_t* d_ids, const lscqp_obstacle* d_table,
                                                 const double* d_radius, const double* d_goal, const lscqp_header* d_hdr, int rows_f32,
                                                 int32_t n_obs_total, int32_t slot0, const double* d_binv3, lscqp_row* d_rows_out, void* stream);
extern "C" int lscqp_goal_fin_raw_(int M, int dim, int use_sfc, int rows_f32, double fin_dt, int64_t n, lscqp_header* d_hdr, const lscqp_row* d_rows,
                                   const uint64_t* d_row_offsets, const lscqp_box* d_sfc, int32_t* d_status, void* stream);
extern "C" int lscqp_goal_raw_(int M, int dim, int use_sfc, int rows_f32, int64_t n, lscqp_header* d_hdr, const lscqp_row* d_rows,
                               const uint64_t* d_row_offsets, const lscqp_box* d_sfc, int32_t* d_status, void* stream);
extern "C" int lscqp_safety_metrics_raw_(int M, int dim, double dt, int64_t n_agents, int64_t first_agent, int64_t n_total, int n_samples,
                                         double record_time_step, double z_2d, const double* d_x_all, const double* d_radius,
                                         const double* d_downwash, const lscqp_header* d_hdr, lscqp_safety* d_out, void* stream);
extern "C" int lscqp_construct_sfc_raw_(lscqp_map mp, int mode, int M, int64_t n, const double* d_points, const double* d_radius,
                                        lscqp_box* d_sfc, int32_t* d_status_out, void* stream);
extern "C" int lscqp_select_neighbours_raw_(int64_t n_agents, int64_t first_agent, int64_t n_total, int32_t n_obs, double range,
                                            const double* d_pos, int32_t* d_nbr, int32_t* d_count, void* stream);
extern "C" int lscqp_validate_step_raw_(int M, int dim, int use_sfc, double dt, int64_t n, double time_step, double z_2d, const double* d_x,
                                        const lscqp_header* d_hdr, const lscqp_box* d_sfc, int32_t* d_valid, double* d_state,
                                        void* stream);
extern "C" int lscqp_shift_traj_raw_(int M, int dim, int64_t n, int shift, double z_2d, const double* d_x_prev, double* d_traj,
                                     void* stream);

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

}  // namespace
extern "C" int lscqp_set_error_(int code, const char* msg) { return fail(code, msg); }
namespace {

// Development / test switches of a handle.  They are read from the environment ONCE, when the handle is created (load_knobs), and live in
// the handle from then on: nothing a solve call can reach calls getenv (a deployed process does not change algorithm because its
// environment changed under it).  The test suite flips them on a live handle through the library-internal lscqp_debug_reload_knobs_ /
// lscqp_debug_set_knob_ (api.py); the launch-shape overrides of the dual active-set phase (das_*) exist only through that setter.
struct Knobs {
    int force_generic = 0;   // LSCQP_FORCE_GENERIC=1: every fp64 launch on the run-time-shaped kernel (lscqp_generic.hip)
    int pin_waves = 0;       // LSCQP_WAVES=1|2|4: pins the wavefront count of the interior-point instance (every compiled instance is reachable by the tests)
    int active_set_off = 0;  // LSCQP_ACTIVE_SET=0 / LSCQP_ACTIVE_SET_NOW=0: rounds 1-4's solver (not for LSCQP_ACTIVE_SET_ONLY handles)
    int check_order = 0;     // LSCQP_CHECK_ORDER=1: d_order is verified to be a permutation (allocates and synchronises)
    int no_queue = 0;        // LSCQP_NO_QUEUE: no persistent workgroups (tools/lpt_probe.py)
    int behind_scan = 0;     // the interior-point pass behind the phase in the persistent kernels' scan form (DevClass::scan; measured: slower)
    // host-pointer entries: a call whose buffers fit in this many bytes runs on the pinned mirror itself (LSCQP_ZERO_COPY_BYTES; 0: always copy).
    // Measured (tools/_dbg/host_latency.py, MI355X, p50 per call, copies -> mapped): 1 QP 43.9 -> 37.3 us, 4 QPs 46.4 -> 39.0, 16 QPs 58.7 ->
    // 50.6, 64 QPs (1.4 MB) 110 -> 85; beyond a few MB the DMA engines' bandwidth wins back what their fixed cost loses.
    int zero_copy_bytes = 4 * 1024 * 1024;
    int defer_behind = 1;    // host-pointer entries: the interior-point pass behind the phase only when the phase left something (0: always enqueued)
    int das_threads = -1, das_kmax = -1, das_steps = -1, das_cache = -1, das_stage = -1, das_screen = -1, das_loop = -1;  // -1: the launch policy's value
};
void load_knobs(Knobs& k) {
    auto on = [](const char* name) { const char* v = getenv(name); return v && v[0] == '1'; };
    k.force_generic = on("LSCQP_FORCE_GENERIC");
    const char* pin = getenv("LSCQP_WAVES");
    k.pin_waves = (pin && (pin[0] == '1' || pin[0] == '2' || pin[0] == '4') && pin[1] == 0) ? pin[0] - '0' : 0;
    const char* as = getenv("LSCQP_ACTIVE_SET");
    const char* now = getenv("LSCQP_ACTIVE_SET_NOW");
    k.active_set_off = now ? now[0] == '0' : (as && as[0] == '0');
    k.check_order = on("LSCQP_CHECK_ORDER");
    k.no_queue = getenv("LSCQP_NO_QUEUE") != nullptr;
    const char* db = getenv("LSCQP_DEFER_BEHIND");
    k.defer_behind = db ? db[0] != '0' : 1;
    const char* zc = getenv("LSCQP_ZERO_COPY_BYTES");
    if (zc) k.zero_copy_bytes = atoi(zc);
}

struct Inst {
    int M, dim, es, max_obs, waves, mixed;
    int nd;     // nested-dissection elimination order (lscqp_kernel.hpp Cfg::ND); 0: the natural order
    int persist;  // the instance has a persistent form (lscqp_inst.hip: LSCQP_PERSIST): only its launches can use a work-queue counter
    int heavy;  // an instance whose state no longer fits the register file: more than 12 LSC slots per lane (376 - 1432 B/lane of
                // scratch measured for <6,3,.,20,1>, <5,3,.,24,1>, <10,2,.,24,1>: 2.3x slower per QP than the two-wavefront
                // instance of the shape at every batch size, M = 6, 512 .. 2048 QPs) or a long matrix row (M >= 7: 108 - 192 B/lane;
                // M = 10 in 2-D: 1.4x slower than the two-wavefront nested-dissection instance at 512 .. 4096 QPs)
    size_t lds;  // bytes of LDS per workgroup
    lscqp::launch_fn fn;
};
constexpr int max_obs_of(int M, int nslot, int w) { return nslot * ((64 * w / (6 * M - 3)) > 0 ? (64 * w / (6 * M - 3)) : 1); }
const Inst kInst[] = {
#define LSCQP_ROW(M, D, E, S, W, X) \
    {M, D, E, max_obs_of(M, S, W), W, X, lscqp::Cfg<M, D, (E != 0), S, W, (X ? 4 : 8)>::ND ? 1 : 0, (W >= 2 && S <= 10 && E != 0 && M != 9 && !X) ? 1 : 0, (S > 12 || (W == 1 && M >= 7)) ? 1 : 0, lscqp::Cfg<M, D, (E != 0), S, W, (X ? 4 : 8)>::lds_bytes(), lscqp_launch_##M##_##D##_##E##_##S##_##W##_##X},
    LSCQP_INSTANCES(LSCQP_ROW)
#undef LSCQP_ROW
};

// Instance of the shape that accommodates n_obs obstacles per agent; nullptr if none.
// Launch policy: a batch small enough to leave SIMDs idle (n <= 2 x CUs: at most two QPs per CU) takes the instance
// with the most wavefronts per QP the shape has -- the row passes of a QP then run on several SIMDs; larger batches take
// the fewest wavefronts per QP, which is what fills the chip (4 QPs per CU) -- unless the instance's LDS footprint admits
// only one workgroup per CU anyway (the nz = 84 class): then more wavefronts are free.  Within a wave count: smallest
// capacity.
const Inst* find_instance(const Knobs& kn, int M, int dim, int es, int mixed, int n_obs, int64_t n, int n_cu) {
    const Inst* best = nullptr;
    // (test switches of the handle: force_generic sends every fp64 launch to the run-time-shaped kernel, also for shapes that have compiled
    // instances -- so that kernel is tested on the shapes every fixture exists for; pin_waves pins the wavefront count)
    if (kn.force_generic && !mixed) return nullptr;
    const bool small = n <= 2 * (int64_t)n_cu;
    const int pin_w = kn.pin_waves;
    for (const Inst& i : kInst) {
        if (!(i.M == M && i.dim == dim && i.es == es && i.mixed == mixed && i.max_obs >= n_obs)) continue;
        if (pin_w && i.waves != pin_w && !mixed) continue;
        bool better = !best;
        if (best && i.heavy != best->heavy) {
            better = i.heavy < best->heavy;  // a spilling instance only when nothing else holds the obstacles
        } else if (best && i.max_obs != best->max_obs && (i.max_obs >= 2 * best->max_obs || best->max_obs >= 2 * i.max_obs)) {
            // slots beyond n_obs are processed as dead rows: an instance with twice the capacity needed is the slower one
            // (forest10 replica, 9 neighbours: <10,2,.,5,2> 0.230 ms, the 40-obstacle <10,2,.,10,4> 0.273 ms)
            better = i.max_obs < best->max_obs;
        } else if (best) {
            const bool one_wg_per_cu = i.lds > lscqp::kMaxLdsBytes / 2 && best->lds > lscqp::kMaxLdsBytes / 2;
            if (i.waves != best->waves)
                better = (small || one_wg_per_cu) ? (i.waves > best->waves) : (i.waves < best->waves);
            else
                better = i.max_obs < best->max_obs;
        }
        if (better) best = &i;
    }
    return best;
}
// An fp64 instance of the same shape that eliminates in the OTHER order (natural vs nested dissection), smallest capacity that holds
// n_obs; nullptr if the shape has none.  A pivot that cancels to <= 0 in one order late in the iteration usually survives in the other
// (tools/sweep_class_params.py: M = 10 in 2-D, 1 of ~570 feasible instances far outside the reference's parameters), so the second
// pass of a `retry` call runs on it.
const Inst* other_order_instance(const Inst* first, int n_obs) {
    const Inst* best = nullptr;
    for (const Inst& i : kInst) {
        if (!(i.M == first->M && i.dim == first->dim && i.es == first->es && i.mixed == 0 && i.max_obs >= n_obs && i.nd != first->nd)) continue;
        if (!best || i.max_obs < best->max_obs) best = &i;
    }
    return best;
}
int cu_count() {  // of the CURRENT device (one process may drive several: lscqp_comm_*)
    static int n_cu[64];
    static bool known[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (!known[dev]) {  // (a failed query is not remembered: the launch policies would then treat every batch as one that fills the chip, for the whole process)
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 0;
        n_cu[dev] = v;
        known[dev] = true;
    }
    return n_cu[dev];
}
// Work-queue counters of the persistent launches (lscqp_kernel.hpp: lscqp_pdip_kernel): zeroable ints per device, one per launch, cleared
// on the launch's own stream right before it.  Two launches must never share a counter while both run (instances would be skipped), so:
//   * an EAGER launch takes the next of kQueueEager rotating slots -- it would have to be overtaken by that many later launches, still
//     running, to meet one of them;
//   * a launch being CAPTURED into a graph (hipStreamIsCapturing) gets a slot of its own for good: the graph replays for as long as its
//     owner likes, concurrently with anything, and the memset node travels with it.  kQueueCaptured such slots per device and process;
//     when they are used up -- or when the ring does not exist yet: hipMalloc is not allowed inside a capture; lscqp_plan runs its first
//     replan eagerly -- the launch simply keeps one instance per workgroup (NULL).
constexpr unsigned kQueueEager = 4096, kQueueCaptured = 4096;
int* next_queue_counter(hipStream_t stream) {
    static int* ring[64] = {};
    static std::mutex mu;
    static std::atomic<unsigned> next_eager{0};
    static std::atomic<unsigned> next_captured[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = stream != nullptr && hipStreamIsCapturing(stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
    int* base = nullptr;
    {
        std::lock_guard<std::mutex> lk(mu);
        if (!ring[dev]) {
            if (capturing) return nullptr;
            if (hipMalloc(&ring[dev], sizeof(int) * (kQueueEager + kQueueCaptured)) != hipSuccess) {
                ring[dev] = nullptr;
                return nullptr;  // (the launch then runs one instance per workgroup, as before)
            }
        }
        base = ring[dev];
    }
    int* slot;
    if (capturing) {
        const unsigned k = next_captured[dev].fetch_add(1, std::memory_order_relaxed);
        if (k >= kQueueCaptured) return nullptr;
        slot = base + kQueueEager + k;
    } else {
        slot = base + (next_eager.fetch_add(1, std::memory_order_relaxed) % kQueueEager);
    }
    if (hipMemsetAsync(slot, 0, sizeof(int), stream) != hipSuccess) return nullptr;
    return slot;
}

// lscqp_order_by_work_device / lscqp_order_by_cost_device: a STABLE counting sort of the instances into kOrdK bins, largest key first, ties
// in index order, so the result is the same from run to run (the order only decides WHEN an instance is processed, never its result).
// One workgroup of kOrdW wavefronts; wavefront w owns the contiguous range [w c, (w + 1) c) and walks it in tiles of 64 (lane = element).
// Inside a tile the rank of an element among its own key is a ballot: the distinct keys of the tile are peeled off one at a time
// (a handful per tile).  Pass 1 counts per (wavefront, key), one prefix gives every (wavefront, key) its first slot, pass 2 scatters.
// (Round 4, first version: one thread per contiguous range and a serial prefix over the threads -- 20 us for 4096 instances, which is
// 5 % of the launch it sorts; this one: 4 us.)
constexpr int kOrdW = 16, kOrdT = 64 * kOrdW, kOrdK = 64;
template <class KeyOf>
__device__ __forceinline__ void order_by_key(int64_t n, KeyOf key_of, int32_t* __restrict__ order) {
    __shared__ int cnt[kOrdW][kOrdK];  // pass 1: elements of (wavefront, bin); then: the next free slot of (wavefront, bin)
    __shared__ int tot[kOrdK];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int e = threadIdx.x; e < kOrdW * kOrdK; e += kOrdT) (&cnt[0][0])[e] = 0;
    __syncthreads();
    const int64_t c = ((n + kOrdW - 1) / kOrdW + 63) / 64 * 64;  // per wavefront, whole tiles
    const int64_t lo = (int64_t)w * c, hi = lo + c < n ? lo + c : n;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int pass = 0; pass < 2; pass++) {
        for (int64_t base = lo; base < hi; base += 64) {
            const int64_t i = base + lane;
            const bool live = i < hi;
            const int bin = live ? key_of(i) : -1;  // 0 = processed first
            unsigned long long rest = __builtin_amdgcn_ballot_w64(live);
            while (rest) {
                const int k = __builtin_amdgcn_readlane(bin, __builtin_ctzll(rest));
                const unsigned long long m = __builtin_amdgcn_ballot_w64(bin == k);
                if (pass == 0) {
                    if (lane == 0) cnt[w][k] += __builtin_popcountll(m);
                } else {
                    const int first = cnt[w][k];
                    if (bin == k) order[first + __builtin_popcountll(m & lt)] = (int32_t)i;
                    __builtin_amdgcn_wave_barrier();
                    if (lane == 0) cnt[w][k] = first + __builtin_popcountll(m);
                }
                __builtin_amdgcn_wave_barrier();
                rest &= ~m;
            }
        }
        if (pass == 0) {
            __syncthreads();
            if (threadIdx.x < kOrdK) {
                int t = 0;
                for (int u = 0; u < kOrdW; u++) t += cnt[u][threadIdx.x];
                tot[threadIdx.x] = t;
            }
            __syncthreads();
            if (threadIdx.x < kOrdK * kOrdW) {  // first slot of (wavefront u, bin k): all smaller bins, then the earlier wavefronts' share of k
                const int k = threadIdx.x % kOrdK, u = threadIdx.x / kOrdK;
                int first = 0;
                for (int kk = 0; kk < k; kk++) first += tot[kk];
                for (int uu = 0; uu < u; uu++) first += cnt[uu][k];
                __syncthreads();
                cnt[u][k] = first;
            } else {
                __syncthreads();
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(kOrdT) void order_by_work_kernel(int64_t n, const lscqp_info* __restrict__ info, int32_t* __restrict__ order) {
    order_by_key(n, [&](int64_t i) -> int {
        const int it = info[i].iterations;
        return kOrdK - 1 - (it < 0 ? 0 : (it >= kOrdK ? kOrdK - 1 : it));
    }, order);
}

// 32-bit costs, scaled to the bins by the largest of them
__global__ __launch_bounds__(kOrdT) void order_by_cost_kernel(int64_t n, const uint32_t* __restrict__ cost, int32_t* __restrict__ order) {
    __shared__ unsigned int cmax[kOrdT];
    unsigned int m = 0;
    for (int64_t i = threadIdx.x; i < n; i += kOrdT) m = cost[i] > m ? cost[i] : m;
    cmax[threadIdx.x] = m;
    __syncthreads();
    for (int sft = kOrdT / 2; sft > 0; sft >>= 1) {
        if ((int)threadIdx.x < sft) cmax[threadIdx.x] = cmax[threadIdx.x] > cmax[threadIdx.x + sft] ? cmax[threadIdx.x] : cmax[threadIdx.x + sft];
        __syncthreads();
    }
    const unsigned long long top = cmax[0] > 0 ? cmax[0] : 1;
    // (16 levels: a tile of 64 costs then has at most 16 distinct keys to peel -- with 64 levels the sort took 30 us for 4096 agents, and
    // the order only has to put the expensive quarter first)
    order_by_key(n, [&](int64_t i) -> int { return 15 - (int)(((unsigned long long)cost[i] * 15) / top); }, order);
}

// Debug aid (environment LSCQP_CHECK_ORDER=1): is d_order a permutation of 0 .. n-1?  A stale or short order buffer would make two workgroups
// write the same instance and leave another untouched; the check costs an allocation and a synchronisation, hence the knob.
__global__ void check_order_kernel(int64_t n, const int32_t* __restrict__ order, int* __restrict__ seen, int* __restrict__ bad) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int32_t q = order[k];
    if (q < 0 || q >= n) atomicAdd(bad, 1);
    else if (atomicAdd(&seen[q], 1) != 0) atomicAdd(bad, 1);
}
int order_is_permutation(int64_t n, const int32_t* d_order, hipStream_t stream) {  // 1 yes, 0 no, -1 could not check
    int* buf = nullptr;
    if (hipMalloc(&buf, sizeof(int) * (size_t)(n + 1)) != hipSuccess) return -1;
    int bad = -1;
    if (hipMemsetAsync(buf, 0, sizeof(int) * (size_t)(n + 1), stream) == hipSuccess) {
        hipLaunchKernelGGL(check_order_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, n, d_order, buf, buf + n);
        if (hipMemcpyAsync(&bad, buf + n, sizeof(int), hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) bad = -1;
    }
    (void)hipFree(buf);
    return bad < 0 ? -1 : (bad == 0 ? 1 : 0);
}

bool shape_exists(int M, int dim, int es, int mixed) {
    for (const Inst& i : kInst)
        if (i.M == M && i.dim == dim && i.es == es && i.mixed == mixed) return true;
    return false;
}

// Closed form of Q_base for n = 5, phi = 3, phi_n = 1 (reference src/traj_optimizer.cpp:163-178:
// B Z B^T is this integer matrix, scaled by dt^(-5)).
const double kQInt[36] = {720, -1800, 1200, 0,     0,     -120, -1800, 4800, -3600, 0,     600,   0,
                          1200, -3600, 3600, -1200, 0,     0,    0,     0,    -1200, 3600,  -3600, 1200,
                          0,    600,   0,    -3600, 4800,  -1800, -120, 0,    0,     1200,  -1800, 720};

}  // namespace

// Tables of the dual active-set phase: one host copy per class generation, one device copy per device that has solved with the handle
// (a communicator drives several devices through one handle).  Shared by the copies lscqp_update makes of the handle.
// A device buffer is IMMUTABLE once a launch may have seen it: an update that changes the tables gives every device a FRESH buffer and
// retires the old one (a kernel of an earlier asynchronous solve, or a captured graph replaying on a non-blocking stream, may still be
// reading it -- until round 5 the copy was made in place).  An update that leaves the tables as they are (the planner's mode flips:
// reference src/traj_planner.cpp:155,160,187,214 change slack_mode only) touches nothing on the device.  Retired buffers are freed
// with the handle; should more than kMaxRetired accumulate (a caller that keeps changing dt or the weights), the devices are
// synchronised once and the list is emptied.
struct DasTables {
    std::mutex mu;
    std::vector<double> host;
    uint64_t gen = 0;
    double* dev[64] = {};
    size_t dev_n[64] = {};
    uint64_t dev_gen[64] = {};
    struct Retired {
        void* p;
        int dev;
    };
    std::vector<Retired> garbage;
    static constexpr size_t kMaxRetired = 256;
};

struct lscqp_solver {
    lscqp_class_desc desc;
    lscqp::DevClass dev;
    int nv, P, es;
    DasTables* das = nullptr;
    // staging of the host-pointer entry points: device buffer + pinned mirror + private stream per concurrent call
    // (one H2D and one D2H per host-pointer solve instead of eight small copies; lscqp_staging.hpp)
    lscqp::StagePool* pool = nullptr;
    Knobs knobs;              // read from the environment at lscqp_create, never afterwards (lscqp_debug_reload_knobs_ for the tests)
    std::atomic<int>* behind_needed = nullptr;  // host-pointer entries: did the last call's phase leave work for the interior-point kernel?
    uint64_t generation = 0;  // bumped by lscqp_update: holders of state derived from the class (a captured plan graph) re-derive it
};

static int derive(lscqp_solver* s, const lscqp_class_desc* d) {
    // same argument checks as the reference: (n, phi) must be (5, 3) (src/traj_optimizer.cpp:198-201),
    // dim <= 3 (:249); M >= 2 is assumed by the continuity rows (:341-352)
    if (d->n != 5 || d->phi != 3) return fail(LSCQP_ERR_INVALID_ARGUMENT, "[TrajOptimizer] Currently, only n=5, phi=3 is available");
    if (d->row_format != LSCQP_ROWS_F64 && d->row_format != LSCQP_ROWS_F32)
        return fail(LSCQP_ERR_INVALID_ARGUMENT, "row_format must be LSCQP_ROWS_F64 or LSCQP_ROWS_F32");
    if (d->phi_n != 1) return fail(LSCQP_ERR_INVALID_ARGUMENT, "phi_n must be 1");
    if (d->dim < 2 || d->dim > 3) return fail(LSCQP_ERR_INVALID_ARGUMENT, "[TrajOptimizer] Invalid output dimension, output_dim > 3");
    if (d->M < 2) return fail(LSCQP_ERR_INVALID_ARGUMENT, "M must be >= 2");
    if (!(d->dt > 0)) return fail(LSCQP_ERR_INVALID_ARGUMENT, "dt must be positive");
    const int es = (d->planner_mode == LSCQP_PLANNER_LSC) ? 1 : 0;
    if (d->precision != LSCQP_PRECISION_F64 && d->precision != LSCQP_PRECISION_MIXED)
        return fail(LSCQP_ERR_INVALID_ARGUMENT, "precision must be LSCQP_PRECISION_F64 or LSCQP_PRECISION_MIXED");
    // a shape without a compiled instance (csrc/lscqp_launch.hpp) is served by the run-time-shaped kernel (lscqp_generic.hip): the
    // reference builds its QP for whatever param.M is (src/traj_optimizer.cpp:4-16)
    if (!shape_exists(d->M, d->dim, es, 0) && !lscqp_generic_supports(d->M, d->dim, es)) {
        char buf[200];
        snprintf(buf, sizeof buf, "no kernel for M=%d dim=%d end_stop=%d: neither a compiled instance (csrc/lscqp_launch.hpp) nor the run-time-shaped kernel (M <= 12)", d->M, d->dim, es);
        return fail(LSCQP_ERR_UNSUPPORTED, buf);
    }
    if (d->precision == LSCQP_PRECISION_MIXED && !shape_exists(d->M, d->dim, es, 1)) {
        char buf[200];
        snprintf(buf, sizeof buf, "no compiled MIXED-precision kernel instance for M=%d dim=%d end_stop=%d (csrc/lscqp_launch.hpp lists them)", d->M, d->dim, es);
        return fail(LSCQP_ERR_UNSUPPORTED, buf);
    }
    s->desc = *d;
    s->es = es;
    s->P = d->M * 6;
    s->nv = d->dim * s->P;
    lscqp::DevClass& c = s->dev;
    memset(&c, 0, sizeof c);
    c.dt = d->dt;
    c.w_c = d->control_input_weight;
    c.w_t = d->terminal_weight;
    c.comm_range = d->communication_range;
    const double sc = std::pow(d->dt, -5.0);
    const long double scl = 1.0L / ((long double)d->dt * d->dt * d->dt * d->dt * d->dt);
    c.q2s = 2.0 * d->control_input_weight * sc;
    for (int i = 0; i < 36; i++) {
        // the reference's objective coefficient is fl(w_c * fl(int * pow(dt,-5))) (src/traj_optimizer.cpp:174-176,294);
        // dQ holds its deviation from the exact product, per unit w_c, so that obj = exact + w_c * c' dQ c
        const double q_ref = kQInt[i] * sc;
        const double p_ref = d->control_input_weight * q_ref;
        c.dQ[i] = d->control_input_weight != 0
                      ? (double)(((long double)p_ref - (long double)d->control_input_weight * kQInt[i] * scl) /
                                 (long double)d->control_input_weight)
                      : 0.0;
    }
    for (int k = 0; k < 3; k++) {
        c.world_min[k] = d->world_min[k];
        c.world_max[k] = d->world_max[k];
    }
    c.tol = d->tol > 0 ? d->tol : 1e-10;  // relative duality gap (1e-9 would save one of ~5 iterations but leaves up to 1.6e-6 m in x at M >= 7)
    c.max_iter = d->max_iter > 0 ? d->max_iter : 60;
    c.use_sfc = d->use_sfc;
    c.n_obs_max = 0;
    c.rows_f32 = d->row_format == LSCQP_ROWS_F32;
    c.repair = 0;
    c.rsfc = d->planner_mode == LSCQP_PLANNER_RSFC;
    if (d->active_set != LSCQP_ACTIVE_SET_DEFAULT && d->active_set != LSCQP_ACTIVE_SET_OFF && d->active_set != LSCQP_ACTIVE_SET_ONLY)
        return fail(LSCQP_ERR_INVALID_ARGUMENT, "active_set must be LSCQP_ACTIVE_SET_DEFAULT, _OFF or _ONLY");
    if (d->warm_start != LSCQP_WARM_DEFAULT && d->warm_start != LSCQP_WARM_TIGHT)
        return fail(LSCQP_ERR_INVALID_ARGUMENT, "warm_start must be LSCQP_WARM_DEFAULT or LSCQP_WARM_TIGHT");
    // (the build-time knobs LSCQP_WARM_MU0 / LSCQP_WARM_S0 of the kernel header are the default mode's values)
    c.warm_mu0 = d->warm_start == LSCQP_WARM_TIGHT ? 1e-7 : LSCQP_WARM_MU0;
    c.warm_s0 = d->warm_start == LSCQP_WARM_TIGHT ? 0.003 : LSCQP_WARM_S0;
    c.warm_net = d->warm_start == LSCQP_WARM_TIGHT ? 0.6 : 0.0;
    return LSCQP_OK;
}

extern "C" const lscqp_class_desc* lscqp_class_desc_of_(lscqp_handle h) { return &h->desc; }  // for lscplan.hip
extern "C" uint64_t lscqp_handle_generation_(lscqp_handle h) { return h->generation; }
// (library-internal, lscqp_comm.hip) does a batch of this shape have a second chance on the instance with the other elimination order?
extern "C" int lscqp_has_other_order_(lscqp_handle h, int64_t n, int32_t n_obs_max);

// one device's copy of the current tables, in a buffer no launch has seen yet (mu held; the caller restores the current device).
// false: allocation or copy failed -- the device then has no tables and its launches run without the phase (or fail loudly, das_device_table)
static bool das_upload(DasTables& D, int d) {
    if (D.dev[d]) {
        D.garbage.push_back({D.dev[d], d});
        D.dev[d] = nullptr;
    }
    if (D.host.empty()) return false;
    if (hipSetDevice(d) != hipSuccess) return false;
    double* p = nullptr;
    if (hipMalloc(&p, sizeof(double) * D.host.size()) != hipSuccess) return false;
    // (a blocking copy into memory nothing else knows about: no launch can race with it)
    if (hipMemcpy(p, D.host.data(), sizeof(double) * D.host.size(), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(p);
        return false;
    }
    D.dev[d] = p;
    D.dev_n[d] = D.host.size();
    D.dev_gen[d] = D.gen;
    return true;
}
// (re)build the host copy of the active-set tables after derive() and give every device that holds a copy the new generation
static void das_refresh(lscqp_solver* s) {
    if (!s->das) return;
    std::lock_guard<std::mutex> lk(s->das->mu);
    DasTables& D = *s->das;
    const size_t nd = lscqp_das_build_tables(s->desc.M, s->es, s->desc.dt, s->desc.control_input_weight, s->desc.terminal_weight, nullptr);
    // (the tables, then the class's 36 coefficient-rounding terms of the objective, then its two-sided rows as far as they are the class's: the
    // kernel reads all three from here)
    const size_t npair = lscqp_das_build_pairs(s->desc.M, s->desc.dim, s->desc.communication_range > 0, nullptr);
    std::vector<double> fresh(nd + 36 + npair, 0.0);
    if (s->desc.control_input_weight > 0 && s->desc.terminal_weight >= 0 &&
        lscqp_das_build_tables(s->desc.M, s->es, s->desc.dt, s->desc.control_input_weight, s->desc.terminal_weight, fresh.data()) == nd) {
        for (int i = 0; i < 36; i++) fresh[nd + i] = s->dev.dQ[i];
        static_assert(sizeof(double) == 2 * sizeof(int32_t), "two ints per table slot");
        lscqp_das_build_pairs(s->desc.M, s->desc.dim, s->desc.communication_range > 0, reinterpret_cast<int32_t*>(fresh.data() + nd + 36));
    } else {
        fresh.clear();  // (a class whose reduced Hessian is not positive definite has no active-set phase)
    }
    if (D.gen != 0 && fresh.size() == D.host.size() && (fresh.empty() || memcmp(fresh.data(), D.host.data(), sizeof(double) * fresh.size()) == 0))
        return;  // the update left the tables as they are: the device copies stay, bit for bit, and nothing in flight is disturbed
    D.host.swap(fresh);
    D.gen++;
    // devices that already hold a copy get the new generation NOW (create / update are host synchronisation points; a later launch may sit
    // inside a stream capture, where nothing can be allocated or copied), each in a fresh buffer
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    for (int d = 0; d < 64; d++)
        if (D.dev[d]) (void)das_upload(D, d);
    if (D.garbage.size() > DasTables::kMaxRetired) {  // (see the struct: bounded, at the price of one synchronisation per kMaxRetired changes)
        for (const DasTables::Retired& g : D.garbage) {
            if (hipSetDevice(g.dev) == hipSuccess) {
                (void)hipDeviceSynchronize();
                (void)hipFree(g.p);
            }
        }
        D.garbage.clear();
    }
    if (have_cur) (void)hipSetDevice(cur);
}
// the tables on the CURRENT device.  *err (when given) says why there are none: 0 the class has no tables (no phase, by construction),
// 1 allocation / copy failed, 2 the first use on this device falls inside a stream capture, where hipMalloc / hipMemcpy are not allowed --
// the caller decides (lscqp_prepare_device before capturing avoids it)
static const double* das_device_table(lscqp_solver* s, hipStream_t stream, int* err = nullptr) {
    if (err) *err = 0;
    if (!s->das) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lk(s->das->mu);
    DasTables& D = *s->das;
    if (D.host.empty()) return nullptr;
    if (D.dev[dev] && D.dev_gen[dev] == D.gen) return D.dev[dev];
    // (this stream capturing -- or, for the NULL stream, another stream of the process capturing in global mode, which the query reports as an
    // error: hipMalloc / a blocking hipMemcpy would invalidate that capture)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const hipError_t qe = hipStreamIsCapturing(stream, &cap);
    if (qe != hipSuccess) (void)hipGetLastError();
    const bool capturing = qe != hipSuccess || cap != hipStreamCaptureStatusNone;
    if (capturing) {
        if (err) *err = 2;
        return nullptr;
    }
    if (!das_upload(D, dev)) {
        if (err) *err = 1;
        return nullptr;
    }
    return D.dev[dev];
}

static inline size_t row_bytes(lscqp_handle h) { return h->dev.rows_f32 ? sizeof(lscqp_row_f32) : sizeof(lscqp_row); }

extern "C" {

int lscqp_create(const lscqp_class_desc* desc, lscqp_handle* out) {
    if (!desc || !out) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null argument");
    lscqp_solver* s = new lscqp_solver();
    int rc = derive(s, desc);
    if (rc != LSCQP_OK) {
        delete s;
        return rc;
    }
    s->pool = new lscqp::StagePool();
    s->das = new DasTables();
    s->behind_needed = new std::atomic<int>(0);
    load_knobs(s->knobs);  // the ONLY place the product reads its environment
    das_refresh(s);
    {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0) (void)das_device_table(s, nullptr);  // (a host without a device still creates handles: lscqp_dump_instance)
    }
    *out = s;
    return LSCQP_OK;
}

// The class's active-set tables on the CURRENT device, now: what the first solve on a device would otherwise do lazily -- and cannot do
// inside a stream capture.  lscqp_create does it for the device that is current then, lscqp_comm_create for every device of the communicator.
int lscqp_prepare_device(lscqp_handle h) {
    if (!h) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null handle");
    int ndev = 0;
    const hipError_t de = hipGetDeviceCount(&ndev);
    if (de != hipSuccess || ndev == 0) return fail(LSCQP_ERR_NO_DEVICE, "no HIP device: lscqp has no CPU fallback");
    int why = 0;
    if (!das_device_table(h, nullptr, &why) && why != 0)
        return fail(LSCQP_ERR_HIP, why == 2 ? "lscqp_prepare_device inside a stream capture" : "active-set tables: device allocation or copy failed");
    return LSCQP_OK;
}

// (library-internal, tests and development tools) re-read the handle's switches from the environment / set one by name; -1 restores a
// launch-shape override to the policy's value
int lscqp_debug_reload_knobs_(lscqp_handle h) {
    if (!h) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null handle");
    Knobs k = h->knobs;
    load_knobs(k);
    h->knobs = k;
    return LSCQP_OK;
}
int lscqp_debug_set_knob_(lscqp_handle h, const char* name, int value) {
    if (!h || !name) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null argument");
    const std::string n(name);
    Knobs& k = h->knobs;
    int* slot = n == "force_generic" ? &k.force_generic : n == "pin_waves" ? &k.pin_waves : n == "active_set_off" ? &k.active_set_off
              : n == "check_order" ? &k.check_order : n == "no_queue" ? &k.no_queue : n == "defer_behind" ? &k.defer_behind : n == "zero_copy_bytes" ? &k.zero_copy_bytes : n == "behind_scan" ? &k.behind_scan
              : n == "das_threads" ? &k.das_threads : n == "das_kmax" ? &k.das_kmax : n == "das_steps" ? &k.das_steps
              : n == "das_cache" ? &k.das_cache : n == "das_stage" ? &k.das_stage : n == "das_screen" ? &k.das_screen
              : n == "das_loop" ? &k.das_loop : nullptr;
    if (!slot) return fail(LSCQP_ERR_INVALID_ARGUMENT, "unknown knob: " + n);
    *slot = value;
    return LSCQP_OK;
}

int lscqp_update(lscqp_handle h, const lscqp_class_desc* desc) {
    if (!h || !desc) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null argument");
    lscqp_solver tmp = *h;
    int rc = derive(&tmp, desc);
    if (rc != LSCQP_OK) return rc;
    tmp.generation = h->generation + 1;
    *h = tmp;  // (the staging pool pointer and the active-set tables travel with the copy)
    das_refresh(h);
    {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0) (void)das_device_table(h, nullptr);
    }
    return LSCQP_OK;
}

int lscqp_destroy(lscqp_handle h) {
    if (!h) return LSCQP_OK;
    delete h->pool;
    delete h->behind_needed;
    if (h->das) {
        for (int d = 0; d < 64; d++)
            if (h->das->dev[d]) (void)hipFree(h->das->dev[d]);
        for (const DasTables::Retired& g : h->das->garbage) (void)hipFree(g.p);
        delete h->das;
    }
    delete h;
    return LSCQP_OK;
}

int lscqp_num_variables(lscqp_handle h) { return h ? h->nv : -1; }
int lscqp_num_segments(lscqp_handle h) { return h ? h->desc.M : -1; }
int lscqp_max_obstacles(lscqp_handle h) {
    if (!h) return -1;
    const int mixed = h->desc.precision == LSCQP_PRECISION_MIXED ? 1 : 0;
    int best = 0;
    for (const Inst& i : kInst)
        if (i.M == h->desc.M && i.dim == h->desc.dim && i.es == h->es && i.mixed == mixed && i.max_obs > best) best = i.max_obs;
    if (!mixed) {  // beyond the compiled instances' register slots the run-time-shaped kernel takes over
        const int g = lscqp_generic_max_obstacles(h->desc.M, h->desc.dim, h->es);
        if (g > best) best = g;
    }
    return best;
}
int lscqp_uses_sfc(lscqp_handle h) { return h ? (h->desc.use_sfc ? 1 : 0) : -1; }
// How many instances of a launch of n the device works on at once: every instance keeps one wavefront per SIMD (more than 256 registers
// per lane), so a CU holds 4 / W workgroups of W wavefronts, or fewer if their LDS does not fit.  A launch of more than this many
// instances runs in rounds and has a tail: that is where the work order pays (lscqp_order_by_work_device).
int64_t lscqp_launch_capacity(lscqp_handle h, int64_t n, int32_t n_obs_max) {
    if (!h || n < 0 || n_obs_max < 0) return -1;
    const int n_cu = cu_count();
    if (n_cu <= 0) return -1;
    const int mixed = h->desc.precision == LSCQP_PRECISION_MIXED ? 1 : 0;
    const Inst* inst = find_instance(h->knobs, h->desc.M, h->desc.dim, h->es, mixed, n_obs_max, n, n_cu);
    if (!inst) return (int64_t)n_cu;  // (the run-time-shaped kernel: one 256-thread workgroup per CU at the LDS sizes it runs with)
    const int by_lds = (int)(lscqp::kMaxLdsBytes / inst->lds), by_simd = 4 / inst->waves;
    const int per_cu = by_lds < by_simd ? by_lds : by_simd;
    return (int64_t)n_cu * (per_cu < 1 ? 1 : per_cu);
}

// Instances ONE device works on at once in the first kernel of a solve of this class (include/lscqp.h): with the dual active-set phase on,
// the phase's one-wavefront form at the occupancy the runtime reports for its LDS footprint; otherwise lscqp_launch_capacity.
extern "C" int lscqp_das_blocks_per_cu(int M, int dim, int kmax, int rows_f32);
int64_t lscqp_device_fill(lscqp_handle h, int64_t n, int32_t n_obs_max) {
    if (!h || n < 0 || n_obs_max < 0) return -1;
    const int n_cu = cu_count();
    if (n_cu <= 0) return -1;
    // (the phase's fill only while the phase finishes what it is given: when the handle's last host-pointer call had to run the interior-point
    // passes behind it -- a hard stretch of the mission, mixed precision on a loaded swarm -- the slower kernel is the one that fills the device)
    const bool ip_busy = h->behind_needed && h->behind_needed->load(std::memory_order_relaxed) != 0;
    if (!ip_busy && h->desc.active_set != LSCQP_ACTIVE_SET_OFF && !(h->knobs.active_set_off && h->desc.active_set != LSCQP_ACTIVE_SET_ONLY) && h->das && !h->das->host.empty()) {
        const int per_cu = lscqp_das_blocks_per_cu(h->desc.M, h->desc.dim, 8, h->dev.rows_f32);
        if (per_cu > 0) return (int64_t)n_cu * per_cu;
    }
    return lscqp_launch_capacity(h, n, n_obs_max);
}

// Work counters of the kernel instance a launch would select (include/lscqp.h): the per-wavefront instruction counts come from the
// table the build reads off each instance's machine code (lsc_dr_planner_amd/isa_work.py -> lscqp_work_table_, generated TU).
extern "C" int lscqp_work_table_(int M, int D, int E, int S, int W, int X, double* out24);
int lscqp_instance_work(lscqp_handle h, int64_t n, int32_t n_obs_max, lscqp_work* out) {
    if (!h || !out) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null argument");
    if (n < 0 || n_obs_max < 0) return fail(LSCQP_ERR_INVALID_ARGUMENT, "negative size");
    const int mixed = h->desc.precision == LSCQP_PRECISION_MIXED ? 1 : 0;
    int n_cu = cu_count();
    if (n_cu <= 0) n_cu = 256;  // (no device in this process: MI355X's CU count decides the small-batch policy)
    const Inst* inst = find_instance(h->knobs, h->desc.M, h->desc.dim, h->es, mixed, n_obs_max, n, n_cu);
    if (!inst) return fail(LSCQP_ERR_UNSUPPORTED, "no compiled kernel instance for this launch (the run-time-shaped kernel carries no instruction counts)");
    const int G = 64 * inst->waves / (6 * inst->M - 3) > 0 ? 64 * inst->waves / (6 * inst->M - 3) : 1;
    const int nslot = inst->max_obs / G;
    double t[24];
    if (lscqp_work_table_(inst->M, inst->dim, inst->es, nslot, inst->waves, inst->mixed, t) != 0)
        return fail(LSCQP_ERR_UNSUPPORTED, "the build holds no instruction counts for this kernel instance");
    memset(out, 0, sizeof *out);
    const double W = inst->waves;
    // per section (iteration body, last pass, fixed part) 8 numbers: {fma, other fp64, valu, lds} that EVERY wavefront of the
    // workgroup runs, then the same four for instructions only some wavefronts run, already summed over those wavefronts
    auto flops = [&](const double* s) { return 64.0 * (W * (2.0 * s[0] + s[1]) + (2.0 * s[4] + s[5])); };
    out->flops_per_iteration = flops(t);
    out->flops_last_pass = flops(t + 8);
    out->flops_fixed = flops(t + 16);
    out->f64_insts_per_iteration = t[0] + t[1] + (t[4] + t[5]) / W;  // averaged over the workgroup's wavefronts
    out->valu_insts_per_iteration = t[2] + t[6] / W;
    out->lds_insts_per_iteration = t[3] + t[7] / W;
    out->valu_insts_fixed = t[18] + t[10] + (t[22] + t[14]) / W;
    out->wavefronts = inst->waves;
    out->nslot = nslot;
    out->max_obstacles = inst->max_obs;
    out->lds_bytes = (int32_t)inst->lds;
    snprintf(out->kernel, sizeof out->kernel, "lscqp_pdip_kernel<%d,%d,%s,%d,%d,%s>", inst->M, inst->dim, inst->es ? "true" : "false", nslot,
             inst->waves, inst->mixed ? "float" : "double");
    return LSCQP_OK;
}
int lscqp_row_bytes(lscqp_handle h) { return h ? (int)row_bytes(h) : -1; }

int lscqp_generate_lsc_device(lscqp_handle h, int64_t n_agents, int32_t n_obs, int64_t first_agent, const double* d_traj,
                              const int32_t* d_neighbours, const double* d_radius, const double* d_downwash,
                              const double* d_goal, lscqp_row* d_rows_out, void* stream) {
    if (!h) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null handle");
    if (n_agents < 0 || n_obs < 0 || first_agent < 0) return fail(LSCQP_ERR_INVALID_ARGUMENT, "negative size");
    if (n_agents == 0 || n_obs == 0) return LSCQP_OK;
    if (!d_traj || !d_neighbours || !d_radius || !d_downwash || !d_goal || !d_rows_out)
        return fail(LSCQP_ERR_INVALID_ARGUMENT, "null buffer");
    int ndev = 0;
    const hipError_t de = hipGetDeviceCount(&ndev);
    if (de != hipSuccess || ndev == 0) return fail(LSCQP_ERR_NO_DEVICE, "no HIP device: lscqp has no CPU fallback");
    return lscqp_generate_lsc_raw_(LSCQP_GEN_LSC, h->desc.M, h->desc.dim, n_agents, n_obs, first_agent, d_traj, nullptr, d_neighbours, d_radius,
                                   d_downwash, d_goal, nullptr, h->dev.rows_f32, n_obs, 0, d_rows_out, stream);
}

int lscqp_generate_constraints_device(lscqp_handle h, int32_t mode, int64_t n_agents, int32_t n_obs, int64_t first_agent,
                                      const double* d_traj, const int32_t* d_neighbours, const double* d_radius,
                                      const double* d_downwash, const double* d_goal_all, lscqp_row* d_rows_out, void* stream) {
    return lscqp_generate_constraints_device_ex(h, mode, n_agents, n_obs, first_agent, d_traj, d_neighbours, d_radius, d_downwash, d_goal_all,
                                                d_rows_out, n_obs, 0, stream);
}

// rows 0..2 of B^-1 for n = 5 (monomial -> Bernstein, closed form C(j,i) / C(n,i)): the size polynomial of
// obstacleSizePredictionWithConstAcc has monomial coefficients (c0, c1, c2, 0, 0, 0)
static const double* device_binv3() {
    static double* d[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!d[dev]) {
        auto C5 = [](int n_, int k_) { double r = 1; for (int i = 1; i <= k_; i++) r = r * (n_ - k_ + i) / i; return r; };
        double hbuf[18];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 6; j++) hbuf[i * 6 + j] = (j >= i) ? C5(j, i) / C5(5, i) : 0.0;
        if (hipMalloc(&d[dev], sizeof hbuf) != hipSuccess) return nullptr;
        if (hipMemcpy(d[dev], hbuf, sizeof hbuf, hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    }
    return d[dev];
}

int lscqp_generate_lsc_obstacles_device(lscqp_handle h, const lscqp_obstacle_param* param, int64_t n_agents, int32_t n_dyn,
                                        int64_t first_agent, const double* d_traj, const int32_t* d_obstacle_ids,
                                        const lscqp_obstacle* d_obstacles, const double* d_radius, const double* d_goal,
                                        const lscqp_header* d_hdr, lscqp_row* d_rows_out, int32_t n_obs_total, int32_t slot0, void* stream) {
    if (!h || !param) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null handle");
    if (n_agents < 0 || n_dyn < 0 || first_agent < 0 || slot0 < 0 || n_obs_total < slot0 + n_dyn)
        return fail(LSCQP_ERR_INVALID_ARGUMENT, "inconsistent sizes (n_obs_total >= slot0 + n_dyn required)");
    if (h->desc.M > 32) return fail(LSCQP_ERR_UNSUPPORTED, "obstacle prediction supports M <= 32");
    if (n_agents == 0 || n_dyn == 0) return LSCQP_OK;
    if (!d_traj || !d_obstacle_ids || !d_obstacles || !d_radius || !d_goal || !d_hdr || !d_rows_out)
        return fail(LSCQP_ERR_INVALID_ARGUMENT, "null buffer");
    int ndev = 0;
    const hipError_t de = hipGetDeviceCount(&ndev);
    if (de != hipSuccess || ndev == 0) return fail(LSCQP_ERR_NO_DEVICE, "no HIP device: lscqp has no CPU fallback");
    const double* binv3 = device_binv3();
    if (!binv3) return fail(LSCQP_ERR_HIP, "constant upload failed");
    return lscqp_generate_lsc_obstacles_raw_(h->desc.M, h->desc.dim, h->desc.dt, param, n_agents, n_dyn, first_agent, d_traj, d_obstacle_ids,
                                             d_obstacles, d_radius, d_goal, d_hdr, h->dev.rows_f32, n_obs_total, slot0, binv3, d_rows_out, stream);
}

int lscqp_shift_traj_partial_device(lscqp_handle h, int64_t n, double fraction, double z_2d, const double* d_x_prev, double* d_traj,
                                    void* stream) {
    if (!h) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null handle");
    if (n < 0) return fail(LSCQP_ERR_INVALID_ARGUMENT, "negative size");
    if (!(fraction > 0.0 && fraction < 1.0)) return fail(LSCQP_ERR_INVALID_ARGUMENT, "fraction = multisim_time_step / dt must lie in (0, 1)");
    if (n == 0) return LSCQP_OK;
    if (!d_x_prev || !d_traj) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null buffer");
    int ndev = 0;
    const hipError_t de = hipGetDeviceCount(&ndev);
    if (de != hipSuccess || ndev == 0) return fail(LSCQP_ERR_NO_DEVICE, "no HIP device: lscqp has no CPU fallback");
    // W = B A B^-1 (src/trajectory.cpp:24-38): B Bernstein -> monomial (include/polynomial.hpp:281-294), A(i,j) = C(i,j) a^j b^(i-j) for
    // t -> a t + b with b = fraction, a = 1 - fraction, B^-1 in closed form
    auto Cn = [](int n_, int k_) { double r = 1; for (int i = 1; i <= k_; i++) r = r * (n_ - k_ + i) / i; return k_ > n_ ? 0.0 : r; };
    double B[6][6], Bi[6][6], A[6][6], BA[6][6], W[36];
    const double b = fraction, a = 1.0 - fraction;
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) {
            B[i][j] = (j >= i) ? Cn(5, i) * Cn(5 - i, 5 - j) * (((j - i) & 1) ? -1.0 : 1.0) : 0.0;
            Bi[i][j] = (j >= i) ? Cn(j, i) / Cn(5, i) : 0.0;
            A[i][j] = (j <= i) ? Cn(i, j) * std::pow(a, j) * std::pow(b, i - j) : 0.0;
        }
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) {
            BA[i][j] = 0;
            for (int l = 0; l < 6; l++) BA[i][j] += B[i][l] * A[l][j];
        }
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) {
            double v = 0;
            for (int l = 0; l < 6; l++) v += BA[i][l] * Bi[l][j];
            W[i * 6 + j] = v;
        }
    return lscqp_shift_traj_partial_raw_(h->desc.M, h->desc.dim, n, W, z_2d, d_x_prev, d_traj, stream);
}

// (library-internal, lscplan.hip) lscqp_generate_constraints_device_ex with the planning agents' initial trajectories kept apart from
// the predicted trajectories of the agents as obstacles: d_own_traj [n_agents][M][6][3], NULL = rows d_traj[first_agent + a]
extern "C" int lscqp_generate_constraints_own_(lscqp_handle h, int32_t mode, int64_t n_agents, int32_t n_obs, int64_t first_agent,
                                               const double* d_traj, const double* d_own_traj, const int32_t* d_neighbours, const double* d_radius,
                                               const double* d_downwash, const double* d_goal_all, lscqp_row* d_rows_out, int32_t n_obs_total,
                                               int32_t slot0, void* stream);

int lscqp_generate_constraints_device_ex(lscqp_handle h, int32_t mode, int64_t n_agents, int32_t n_obs, int64_t first_agent,
                                         const double* d_traj, const int32_t* d_neighbours, const double* d_radius,
                                         const double* d_downwash, const double* d_goal_all, lscqp_row* d_rows_out, int32_t n_obs_total,
                                         int32_t slot0, void* stream) {
    return lscqp_generate_constraints_own_(h, mode, n_agents, n_obs, first_agent, d_traj, nullptr, d_neighbours, d_radius, d_downwash, d_goal_all,
                                           d_rows_out, n_obs_total, slot0, stream);
}

int lscqp_generate_constraints_own_(lscqp_handle h, int32_t mode, int64_t n_agents, int32_t n_obs, int64_t first_agent,
                                    const double* d_traj, const double* d_own_traj, const int32_t* d_neighbours, const double* d_radius,
                                    const double* d_downwash, const double* d_goal_all, lscqp_row* d_rows_out, int32_t n_obs_total,
                                    int32_t slot0, void* stream) {
    if (!h) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null handle");
    if (slot0 < 0 || n_obs_total < slot0 + n_obs) return fail(LSCQP_ERR_INVALID_ARGUMENT, "n_obs_total >= slot0 + n_obs required");
    if (mode != LSCQP_GEN_LSC && mode != LSCQP_GEN_CLSC && mode != LSCQP_GEN_BVC)
        return fail(LSCQP_ERR_INVALID_ARGUMENT, "mode must be LSCQP_GEN_LSC, LSCQP_GEN_CLSC or LSCQP_GEN_BVC");
    if (n_agents < 0 || n_obs < 0 || first_agent < 0) return fail(LSCQP_ERR_INVALID_ARGUMENT, "negative size");
    if (n_agents == 0 || n_obs == 0) return LSCQP_OK;
    if (!d_traj || !d_neighbours || !d_radius || !d_downwash || !d_goal_all || !d_rows_out)
        return fail(LSCQP_ERR_INVALID_ARGUMENT, "null buffer");
    int ndev = 0;
    const hipError_t de = hipGetDeviceCount(&ndev);
    if (de != hipSuccess || ndev == 0) return fail(LSCQP_ERR_NO_DEVICE, "no HIP device: lscqp has no CPU fallback");
    return lscqp_generate_lsc_raw_(mode, h->desc.M, h->desc.dim, n_agents, n_obs, first_agent, d_traj, d_own_traj, d_neighbours, d_radius,
                                   d_downwash, d_goal_all + 3 * first_agent, d_goal_all, h->dev.rows_f32, n_obs_total, slot0, d_rows_out, stream);
}

int lscqp_shift_traj_device(lscqp_handle h, int64_t n, int32_t shift_segments, double z_2d, const double* d_x_prev, double* d_traj,
                            void* stream) {
    if (!h) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null handle");
    if (n < 0) return fail(LSCQP_ERR_INVALID_ARGUMENT, "negative size");
    if (shift_segments < 0 || shift_segments > 1) return fail(LSCQP_ERR_INVALID_ARGUMENT, "shift_segments must be 0 or 1");
    if (n == 0) return LSCQP_OK;
    if (!d_x_prev || !d_traj) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null buffer");
    int ndev = 0;
    const hipError_t de = hipGetDeviceCount(&ndev);
    if (de != hipSuccess || ndev == 0) return fail(LSCQP_ERR_NO_DEVICE, "no HIP device: lscqp has no CPU fallback");
    return lscqp_shift_traj_raw_(h->desc.M, h->desc.dim, n, shift_segments, z_2d, d_x_prev, d_traj, stream);
}

int64_t lscqp_generate_lsc_bytes(lscqp_handle h, int64_t n_agents, int32_t n_obs, int64_t n_total) {
    if (!h) return -1;
    const int64_t P = h->P;
    return n_agents * (int64_t)n_obs * P * (h->dev.rows_f32 ? 16 : 32) /* rows written */
           + n_total * (P * 24 + 16)               /* control points, radius, downwash */
           + n_agents * ((int64_t)n_obs * 4 + 24); /* neighbour ids, goal */
}

// (library-internal, lscplan.hip) the goal LP that also finishes the headers of the chain: goal as a point3d, terminal_segments (fin_dt = the class's dt)
int lscqp_optimize_goal_fin_device_(lscqp_handle h, int64_t n, lscqp_header* d_hdr, const lscqp_row* d_rows, const uint64_t* d_row_offsets,
                                    const lscqp_box* d_sfc, int32_t* d_status_out, double fin_dt, void* stream);
int lscqp_optimize_goal_device(lscqp_handle h, int64_t n, lscqp_header* d_hdr, const lscqp_row* d_rows,
                               const uint64_t* d_row_offsets, const lscqp_box* d_sfc, int32_t* d_status_out, void* stream) {
    return lscqp_optimize_goal_fin_device_(h, n, d_hdr, d_rows, d_row_offsets, d_sfc, d_status_out, 0.0, stream);
}
int lscqp_optimize_goal_fin_device_(lscqp_handle h, int64_t n, lscqp_header* d_hdr, const lscqp_row* d_rows, const uint64_t* d_row_offsets,
                                    const lscqp_box* d_sfc, int32_t* d_status_out, double fin_dt, void* stream) {
    if (!h) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null handle");
    if (n < 0) return fail(LSCQP_ERR_INVALID_ARGUMENT, "negative size");
    if (n == 0) return LSCQP_OK;
    if (!d_hdr || !d_status_out || (h->desc.use_sfc && !d_sfc)) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null buffer");
    int ndev = 0;
    const hipError_t de = hipGetDeviceCount(&ndev);
    if (de != hipSuccess || ndev == 0) return fail(LSCQP_ERR_NO_DEVICE, "no HIP device: lscqp has no CPU fallback");
    return lscqp_goal_fin_raw_(h->desc.M, h->desc.dim, h->desc.use_sfc, h->dev.rows_f32, fin_dt, n, d_hdr, d_rows, d_row_offsets, d_sfc, d_status_out, stream);
}

int lscqp_optimize_goal(lscqp_handle h, int64_t n, lscqp_header* hdr, const lscqp_row* rows, const uint64_t* row_offsets,
                        const lscqp_box* sfc, int32_t* status_out) {
    if (!h) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null handle");
    if (n < 0) return fail(LSCQP_ERR_INVALID_ARGUMENT, "negative size");
    if (n == 0) return LSCQP_OK;
    if (!hdr || !status_out) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null buffer");
    int ndev = 0;
    const hipError_t de = hipGetDeviceCount(&ndev);
    if (de != hipSuccess || ndev == 0) return fail(LSCQP_ERR_NO_DEVICE, "no HIP device: lscqp has no CPU fallback");
    int n_obs_max = 0;
    for (int64_t q = 0; q < n; q++) {
        if (hdr[q].n_obs < 0) return fail(LSCQP_ERR_INVALID_ARGUMENT, "negative n_obs");
        if (hdr[q].n_obs > n_obs_max) n_obs_max = hdr[q].n_obs;
    }
    if (n_obs_max > 0 && (!rows || !row_offsets)) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null row buffer");
    if (h->desc.use_sfc && !sfc) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null sfc buffer");
    const size_t n_rows = n_obs_max > 0 ? (size_t)row_offsets[n] : 0;
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t b_hdr = al(sizeof(lscqp_header) * n), b_rows = al(row_bytes(h) * n_rows), b_off = al(sizeof(uint64_t) * (n + 1)),
                 b_sfc = al(sizeof(lscqp_box) * n * h->desc.M), b_st = al(sizeof(int32_t) * n);
    // layout [rows | offsets | sfc | hdr | status]: one H2D of rows .. hdr, one D2H of hdr .. status (pinned staging)
    const size_t total = b_rows + b_off + b_sfc + b_hdr + b_st;
    lscqp::SlotGuard sg{*h->pool, h->pool->acquire(total)};
    if (!sg.slot) return fail(LSCQP_ERR_HIP, "staging allocation failed (hipMalloc / hipHostMalloc / stream)");
    hipStream_t st = sg.slot->stream;
    char* const db = (char*)sg.slot->d;
    char* const hb = (char*)sg.slot->h;
    const size_t o_rows = 0, o_off = b_rows, o_sfc = o_off + b_off, o_hdr = o_sfc + b_sfc, o_st = o_hdr + b_hdr;
#define LSCQP_CK(call)                                                                           \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) return fail(LSCQP_ERR_HIP, std::string(#call ": ") + hipGetErrorString(e_)); \
    } while (0)
    if (n_rows) memcpy(hb + o_rows, rows, row_bytes(h) * n_rows);
    if (n_obs_max > 0) memcpy(hb + o_off, row_offsets, sizeof(uint64_t) * (n + 1));
    else memset(hb + o_off, 0, sizeof(uint64_t) * (n + 1));
    if (h->desc.use_sfc) memcpy(hb + o_sfc, sfc, sizeof(lscqp_box) * n * h->desc.M);
    memcpy(hb + o_hdr, hdr, sizeof(lscqp_header) * n);
    LSCQP_CK(hipMemcpyAsync(db, hb, o_st, hipMemcpyHostToDevice, st));
    int rc = lscqp_optimize_goal_device(h, n, (lscqp_header*)(db + o_hdr), (const lscqp_row*)(db + o_rows), (const uint64_t*)(db + o_off),
                                        (const lscqp_box*)(db + o_sfc), (int32_t*)(db + o_st), st);
    if (rc != LSCQP_OK) return rc;
    LSCQP_CK(hipMemcpyAsync(hb + o_hdr, db + o_hdr, b_hdr + b_st, hipMemcpyDeviceToHost, st));
    LSCQP_CK(hipStreamSynchronize(st));
    memcpy(hdr, hb + o_hdr, sizeof(lscqp_header) * n);
    memcpy(status_out, hb + o_st, sizeof(int32_t) * n);
#undef LSCQP_CK
    return LSCQP_OK;
}

int lscqp_safety_metrics_device(lscqp_handle h, int64_t n_agents, int64_t first_agent, int64_t n_total, int32_t n_samples,
                                double record_time_step, double z_2d, const double* d_x_all, const double* d_radius,
                                const double* d_downwash, const lscqp_header* d_hdr, lscqp_safety* d_out, void* stream) {
    if (!h) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null handle");
    if (n_agents < 0 || first_agent < 0 || n_samples < 0 || n_total < first_agent + n_agents)
        return fail(LSCQP_ERR_INVALID_ARGUMENT, "inconsistent sizes");
    if (n_agents == 0) return LSCQP_OK;
    if (!d_x_all || !d_radius || !d_downwash || !d_hdr || !d_out) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null buffer");
    int ndev = 0;
    const hipError_t de = hipGetDeviceCount(&ndev);
    if (de != hipSuccess || ndev == 0) return fail(LSCQP_ERR_NO_DEVICE, "no HIP device: lscqp has no CPU fallback");
    return lscqp_safety_metrics_raw_(h->desc.M, h->desc.dim, h->desc.dt, n_agents, first_agent, n_total, n_samples, record_time_step, z_2d,
                                     d_x_all, d_radius, d_downwash, d_hdr, d_out, stream);
}

extern "C" int lscqp_safety_obstacles_raw_(int M, int dim, double dt, int64_t n_agents, int64_t first_agent, int n_samples, double record_time_step,
                                           double z_2d, const double* d_x_all, const double* d_radius, const double* d_downwash, int n_obstacles,
                                           const lscqp_obstacle* d_obstacles, lscqp_safety_obs* d_out, void* stream);

int lscqp_safety_obstacles_device(lscqp_handle h, int64_t n_agents, int64_t first_agent, int64_t n_total, int32_t n_samples,
                                  double record_time_step, double z_2d, const double* d_x_all, const double* d_radius,
                                  const double* d_downwash, int32_t n_obstacles, const lscqp_obstacle* d_obstacles,
                                  lscqp_safety_obs* d_out, void* stream) {
    if (!h) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null handle");
    if (n_agents < 0 || first_agent < 0 || n_samples < 0 || n_obstacles < 0 || n_total < first_agent + n_agents)
        return fail(LSCQP_ERR_INVALID_ARGUMENT, "inconsistent sizes");
    if (n_agents == 0) return LSCQP_OK;
    if (!d_x_all || !d_radius || !d_downwash || !d_out || (n_obstacles > 0 && !d_obstacles)) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null buffer");
    int ndev = 0;
    const hipError_t de = hipGetDeviceCount(&ndev);
    if (de != hipSuccess || ndev == 0) return fail(LSCQP_ERR_NO_DEVICE, "no HIP device: lscqp has no CPU fallback");
    return lscqp_safety_obstacles_raw_(h->desc.M, h->desc.dim, h->desc.dt, n_agents, first_agent, n_samples, record_time_step, z_2d, d_x_all,
                                       d_radius, d_downwash, n_obstacles, d_obstacles, d_out, stream);
}

extern "C" int lscqp_construct_sfc_raw_ex_(lscqp_map mp, int mode, int M, int64_t n, const double* d_points, const double* d_radius, lscqp_box* d_sfc,
                                           int32_t* d_status_out, const int32_t* d_order, uint32_t* d_cost_out, void* stream);

int lscqp_construct_sfc_device(lscqp_handle h, lscqp_map mp, int32_t mode, int64_t n, const double* d_points, const double* d_radius,
                               lscqp_box* d_sfc, int32_t* d_status_out, void* stream) {
    return lscqp_construct_sfc_device_ordered(h, mp, mode, n, d_points, d_radius, d_sfc, d_status_out, nullptr, nullptr, stream);
}

int lscqp_order_by_cost_device(int64_t n, const uint32_t* d_cost_prev, int32_t* d_order_out, void* stream) {
    if (n < 0 || n > 0x7fffffff) return fail(LSCQP_ERR_INVALID_ARGUMENT, "0 <= n < 2^31 required");
    if (n == 0) return LSCQP_OK;
    if (!d_cost_prev || !d_order_out) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null buffer");
    int ndev = 0;
    const hipError_t de = hipGetDeviceCount(&ndev);
    if (de != hipSuccess || ndev == 0) return fail(LSCQP_ERR_NO_DEVICE, "no HIP device: lscqp has no CPU fallback");
    hipLaunchKernelGGL(order_by_cost_kernel, dim3(1), dim3(kOrdT), 0, (hipStream_t)stream, n, d_cost_prev, d_order_out);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(LSCQP_ERR_HIP, std::string("HIP launch failed (order_by_cost): ") + hipGetErrorString(e));
    return LSCQP_OK;
}

int lscqp_construct_sfc_device_ordered(lscqp_handle h, lscqp_map mp, int32_t mode, int64_t n, const double* d_points, const double* d_radius,
                                       lscqp_box* d_sfc, int32_t* d_status_out, const int32_t* d_order, uint32_t* d_cost_out, void* stream) {
    if (!h || !mp) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null handle");
    if (mode != LSCQP_SFC_INIT && mode != LSCQP_SFC_FROM_HULL && mode != LSCQP_SFC_FROM_POINT)
        return fail(LSCQP_ERR_INVALID_ARGUMENT, "mode must be LSCQP_SFC_INIT, LSCQP_SFC_FROM_HULL or LSCQP_SFC_FROM_POINT");
    if (n < 0) return fail(LSCQP_ERR_INVALID_ARGUMENT, "negative size");
    if (n == 0) return LSCQP_OK;
    if (!d_points || !d_radius || !d_sfc || !d_status_out) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null buffer");
    if (h->desc.M > 21) return fail(LSCQP_ERR_UNSUPPORTED, "corridor shift supports M <= 21");
    int ndev = 0;
    const hipError_t de = hipGetDeviceCount(&ndev);
    if (de != hipSuccess || ndev == 0) return fail(LSCQP_ERR_NO_DEVICE, "no HIP device: lscqp has no CPU fallback");
    return lscqp_construct_sfc_raw_ex_(mp, mode, h->desc.M, n, d_points, d_radius, d_sfc, d_status_out, d_order, d_cost_out, stream);
}

int lscqp_select_neighbours_device(lscqp_handle h, int64_t n_agents, int64_t first_agent, int64_t n_total, int32_t n_obs,
                                   double communication_range, const double* d_positions, int32_t* d_neighbours_out,
                                   int32_t* d_count_out, void* stream) {
    if (!h) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null handle");
    if (n_agents < 0 || first_agent < 0 || n_obs < 0 || n_total < first_agent + n_agents)
        return fail(LSCQP_ERR_INVALID_ARGUMENT, "inconsistent sizes");
    if (n_agents == 0) return LSCQP_OK;
    if (!d_positions || !d_count_out || (n_obs > 0 && !d_neighbours_out)) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null buffer");
    int ndev = 0;
    const hipError_t de = hipGetDeviceCount(&ndev);
    if (de != hipSuccess || ndev == 0) return fail(LSCQP_ERR_NO_DEVICE, "no HIP device: lscqp has no CPU fallback");
    return lscqp_select_neighbours_raw_(n_agents, first_agent, n_total, n_obs, communication_range, d_positions, d_neighbours_out,
                                        d_count_out, stream);
}

int lscqp_validate_step_device(lscqp_handle h, int64_t n, double time_step, double z_2d, const double* d_x,
                               const lscqp_header* d_hdr, const lscqp_box* d_sfc, int32_t* d_valid_out, double* d_state_out,
                               void* stream) {
    if (!h) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null handle");
    if (n < 0 || !(time_step >= 0)) return fail(LSCQP_ERR_INVALID_ARGUMENT, "negative size or time");
    if (n == 0) return LSCQP_OK;
    if (!d_x || !d_hdr || !d_valid_out || !d_state_out || (h->desc.use_sfc && !d_sfc)) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null buffer");
    int ndev = 0;
    const hipError_t de = hipGetDeviceCount(&ndev);
    if (de != hipSuccess || ndev == 0) return fail(LSCQP_ERR_NO_DEVICE, "no HIP device: lscqp has no CPU fallback");
    return lscqp_validate_step_raw_(h->desc.M, h->desc.dim, h->desc.use_sfc, h->desc.dt, n, time_step, z_2d, d_x, d_hdr, d_sfc, d_valid_out,
                                    d_state_out, stream);
}

int lscqp_num_inequalities(lscqp_handle h, int32_t n_obs) {
    if (!h) return -1;
    const int M = h->desc.M, dim = h->desc.dim, P = h->P;
    int m = n_obs * (P - 3) + 2 * dim * (5 * M - 2) + 2 * dim * (4 * M - 1);
    if (h->desc.use_sfc) m += 2 * dim * (P - 3);
    if (h->desc.communication_range > 0) m += 2 * dim * (M * (M + 1) / 2 + M);
    return m;
}

int64_t lscqp_algorithmic_bytes(lscqp_handle h, int32_t n_obs) {
    if (!h) return -1;
    // rows (32 B each, all n_obs*P of them) + SFC boxes (48 B/segment) + header (256 B) in;
    // control points (8 B each) + objective + status (16 B) out.
    return (int64_t)(h->dev.rows_f32 ? 16 : 32) * n_obs * h->P + 48 * h->desc.M + 256 + 8 * h->nv + 16;
}

int lscqp_order_by_work_device(int64_t n, const lscqp_info* d_info_prev, int32_t* d_order_out, void* stream) {
    if (n < 0 || n > 0x7fffffff) return fail(LSCQP_ERR_INVALID_ARGUMENT, "0 <= n < 2^31 required");
    if (n == 0) return LSCQP_OK;
    if (!d_info_prev || !d_order_out) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null buffer");
    int ndev = 0;
    const hipError_t de = hipGetDeviceCount(&ndev);
    if (de != hipSuccess || ndev == 0) return fail(LSCQP_ERR_NO_DEVICE, "no HIP device: lscqp has no CPU fallback");
    hipLaunchKernelGGL(order_by_work_kernel, dim3(1), dim3(kOrdT), 0, (hipStream_t)stream, n, d_info_prev, d_order_out);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(LSCQP_ERR_HIP, std::string("HIP launch failed (order_by_work): ") + hipGetErrorString(e));
    return LSCQP_OK;
}

int lscqp_solve_batch_device_internal_(lscqp_handle h, int64_t n, int32_t n_obs_max, const lscqp_header* d_hdr, const lscqp_row* d_rows,
                                       const uint64_t* d_row_offsets, const lscqp_box* d_sfc, const double* d_x_init, double* d_x_out, double* d_obj_out,
                                       int32_t* d_status_out, lscqp_info* d_info_out, int32_t retry, const int32_t* d_order, void* stream, int* deferred);

int lscqp_solve_batch_device_ex(lscqp_handle h, int64_t n, int32_t n_obs_max, const lscqp_header* d_hdr,
                                const lscqp_row* d_rows, const uint64_t* d_row_offsets, const lscqp_box* d_sfc,
                                const double* d_x_init, double* d_x_out, double* d_obj_out, int32_t* d_status_out,
                                lscqp_info* d_info_out, int32_t retry, void* stream) {
    return lscqp_solve_batch_device_ordered(h, n, n_obs_max, d_hdr, d_rows, d_row_offsets, d_sfc, d_x_init, d_x_out, d_obj_out, d_status_out,
                                            d_info_out, retry, nullptr, stream);
}

int lscqp_solve_batch_device_ordered(lscqp_handle h, int64_t n, int32_t n_obs_max, const lscqp_header* d_hdr,
                                     const lscqp_row* d_rows, const uint64_t* d_row_offsets, const lscqp_box* d_sfc,
                                     const double* d_x_init, double* d_x_out, double* d_obj_out, int32_t* d_status_out,
                                     lscqp_info* d_info_out, int32_t retry, const int32_t* d_order, void* stream) {
    if (retry < 0 || retry > 3) return fail(LSCQP_ERR_INVALID_ARGUMENT, "retry must be 0, 1, 2 or 3");
    return lscqp_solve_batch_device_internal_(h, n, n_obs_max, d_hdr, d_rows, d_row_offsets, d_sfc, d_x_init, d_x_out, d_obj_out, d_status_out, d_info_out,
                                              retry, d_order, stream, nullptr);
}

// The worker behind the public device entries.  retry also takes the library's own pass codes: -2 = only the repair pass on the instance of
// the other elimination order, -3 = only the rescue pass (the host-pointer entries and lscqp_comm.hip run them after looking at the statuses),
// -10 - r = the interior-point passes of a call with retry = r whose dual active-set phase has ALREADY run (see `deferred`).
// deferred != NULL (the host-pointer entries): when the dual active-set phase runs, the call returns right behind it with *deferred = 1 and
// the interior-point passes are NOT enqueued -- the caller reads the statuses with the results and enqueues them (pass code -10 - retry)
// only if the phase left an instance: on the bench's batches it never does, and a near-empty launch costs 2 - 4 us of every call.
int lscqp_solve_batch_device_internal_(lscqp_handle h, int64_t n, int32_t n_obs_max, const lscqp_header* d_hdr,
                                       const lscqp_row* d_rows, const uint64_t* d_row_offsets, const lscqp_box* d_sfc,
                                       const double* d_x_init, double* d_x_out, double* d_obj_out, int32_t* d_status_out,
                                       lscqp_info* d_info_out, int32_t retry, const int32_t* d_order, void* stream, int* deferred) {
    if (!h) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null handle");
    if (deferred) *deferred = 0;
    bool behind_only = false;  // the phase of this call ran in an earlier invocation
    if (retry <= -10 && retry >= -13) {
        behind_only = true;
        retry = -10 - retry;
    }
    if (retry < -3 || retry > 3 || retry == -1) return fail(LSCQP_ERR_INVALID_ARGUMENT, "invalid pass code");
    if (n < 0 || n_obs_max < 0) return fail(LSCQP_ERR_INVALID_ARGUMENT, "negative size");
    if (n == 0) return LSCQP_OK;
    if (!d_hdr || !d_x_out || !d_obj_out || !d_status_out || (n_obs_max > 0 && (!d_rows || !d_row_offsets)) ||
        (h->desc.use_sfc && !d_sfc))
        return fail(LSCQP_ERR_INVALID_ARGUMENT, "null buffer");
    int ndev = 0;
    {
        const hipError_t de = hipGetDeviceCount(&ndev);
        if (de != hipSuccess || ndev == 0)
            return fail(LSCQP_ERR_NO_DEVICE, std::string("no HIP device: lscqp has no CPU fallback (hipGetDeviceCount: ") +
                                                 hipGetErrorString(de) + ", " + std::to_string(ndev) + " devices)");
    }
    const Knobs& kn = h->knobs;
    if (d_order && kn.check_order && order_is_permutation(n, d_order, (hipStream_t)stream) == 0)
        return fail(LSCQP_ERR_INVALID_ARGUMENT, "d_order is not a permutation of 0 .. n-1 (LSCQP_CHECK_ORDER)");
    const int mixed = h->desc.precision == LSCQP_PRECISION_MIXED ? 1 : 0;
    const Inst* inst = find_instance(h->knobs, h->desc.M, h->desc.dim, h->es, mixed, n_obs_max, n, cu_count());
    const Inst* inst64 = mixed ? find_instance(h->knobs, h->desc.M, h->desc.dim, h->es, 0, n_obs_max, n, cu_count()) : inst;
    lscqp::DevClass cls = h->dev;
    cls.n_obs_max = n_obs_max;
    cls.order = d_order;
    // a launch of more instances than the device has CUs MAY exceed what the chip holds at once: it gets a zeroed work-queue counter and
    // the instance's launcher decides (lscqp_inst.hip: persistent workgroups over the queue, or one instance per workgroup)
    const bool queued = !kn.no_queue && n > (int64_t)cu_count();  // (no_queue: tools/lpt_probe.py tells the queue and the order apart)
    // (a counter -- a memset on the stream, a slot of the ring -- only for a launch that can use it: the instance has a persistent form, and
    // the pass is not the near-empty one behind the dual active-set phase, where almost every workgroup returns at once)
    bool das_in_front = false;
    auto with_queue = [&](lscqp::DevClass& c, const Inst* i) {
        c.queue = (queued && i && i->persist && !das_in_front) ? next_queue_counter((hipStream_t)stream) : nullptr;
    };
    hipError_t e = hipSuccess;
    if (retry < 0 && h->desc.active_set == LSCQP_ACTIVE_SET_ONLY) return LSCQP_OK;  // (the host-pointer entries' extra passes are interior-point passes)
    // ---- the DUAL ACTIVE SET phase (lscqp_das.hip) in front of the first interior-point pass ----------------------------------------
    // One launch over the batch; what it finishes is OPTIMAL (LSCQP_INFO_ACTIVE_SET), everything else is marked for the interior-point
    // kernel, whose first pass then runs with cls.repair = 3 (skip what is OPTIMAL, nothing was "repaired").  Launch shape: a batch that
    // leaves the chip idle gets four wavefronts per QP, the whole budget of active rows and the class's table in LDS (latency); a batch
    // that fills it gets one wavefront per QP and a small LDS footprint (occupancy is what hides the row reads), and the few instances
    // with more active rows than that fall to the interior-point kernel.
    bool das_ran = behind_only;
    if (retry >= 0 && !behind_only && h->desc.active_set != LSCQP_ACTIVE_SET_OFF) {
        const bool off = kn.active_set_off && h->desc.active_set != LSCQP_ACTIVE_SET_ONLY;
        int why = 0;
        const double* d_tab = off ? nullptr : das_device_table(h, (hipStream_t)stream, &why);
        // (no tables on this device and none can be made right now: running without the phase would differ from an eager run in the last
        // bits -- silently, for the whole lifetime of a captured graph.  Say so instead.)
        if (!off && !d_tab && why == 2)
            return fail(LSCQP_ERR_HIP, "the class's active-set tables are not on this device and the launch sits inside a stream capture: call "
                                       "lscqp_prepare_device(handle) on this device before capturing");
        if (!off && !d_tab && why == 1) return fail(LSCQP_ERR_HIP, "active-set tables: device allocation or copy failed");
        int cap = 0;
        if (!inst || !inst64) cap = (mixed || n_obs_max > lscqp_generic_max_obstacles(h->desc.M, h->desc.dim, h->es)) ? -1 : n_obs_max;
        else cap = std::min(inst->max_obs, inst64->max_obs);
        if (d_tab && cap >= 0) {
            // Launch shape (measured, profiles/r05_das_launch_shapes.txt): up to two QPs per CU the launch is about latency -- four
            // wavefronts per QP, the whole budget of active rows, the class's table and the instance's rows in LDS; up to eight per CU
            // the launch still lasts as long as its slowest QP (1024 x M10 x 40: 0.28 ms with one wavefront per QP, 0.19 ms with four) but
            // LDS is what limits the resident workgroups -- four wavefronts, a small footprint; beyond that one wavefront per QP.
            const int64_t ncu = cu_count() > 0 ? cu_count() : 256;
            const bool small = n <= 2 * ncu, medium = n <= 8 * ncu;
            auto knob = [](int v, int dflt) { return v >= 0 ? v : dflt; };  // (overrides: lscqp_debug_set_knob_, tests and sweeps only)
            int threads = knob(kn.das_threads, medium ? 256 : 64);
            // (20 active rows, not the 32 the kernel could hold: the footprint decides how many workgroups a CU holds at once and whether the
            // instance's rows fit in LDS beside the rest -- 512 x M6: 43.0 -> 33.9 us, 128 x M10 x 40: 92.2 -> 84.3, 64 x M5: 12.9 -> 12.5; 24 would
            // already cost the M = 10 class its staged rows.  No feasible instance of a 6 000-instance sweep of the harder swarms needs more than
            // 12; ONE of the ~50 000 of the stress sweep needs 17-20, and at 16 it went to the interior-point kernel, which accepted it at its
            // rounding floor (stationarity 1.9e-7): profiles/r05_kmax_sweep.txt, NOTES.md section 13)
            // round 6 (tools/loaded_probe.py, swarms 8 - 30 replans into their exchange): a batch that leaves most CUs idle (n <= CUs) gets every
            // active row the kernel can hold -- the forest10 class mid-exchange holds > 20 rows at one agent's optimum for several replans, and
            // handing that ONE instance over cost the call 0.16 ms of phase + 0.34 ms of interior point against 0.34 ms without the phase; the
            // step budget of the other small batches is halved: a feasible instance of the loaded sweeps needs <= 50 steps (<= 33 beyond 64 agents),
            // an instance that keeps adding and dropping beyond that is, on those sweeps, one whose rows admit no point -- the kernel behind
            // says so in 14 iterations, and every step spent here before that is added to the call
            const bool tiny = n <= ncu;
            int kmax = knob(kn.das_kmax, tiny ? 32 : small ? 20 : 8);
            int steps = knob(kn.das_steps, tiny ? 96 : small ? 48 : 24);
            int cacheC = knob(kn.das_cache, small ? 1 : 0);
            int stage = knob(kn.das_stage, small ? 1 : 0) ? n_obs_max * 6 * h->desc.M : 0;
            // form: bit 0 the lean form in front (built and measured, no gain: the phase is bound by instruction issue, not occupancy); bit 1 the
            // first look inside the loop of steps (one copy of that code: batches of at most two workgroups per CU; lscqp_das.hip, PEEL)
            const int screen = (knob(kn.das_screen, 0) ? 1 : 0) | (knob(kn.das_loop, small ? 1 : 0) ? 2 : 0);
            const int Mx = h->desc.M, dx = h->desc.dim;
            // what does not fit the CU's LDS is given up in this order: staged rows, the table copy, active rows
            if (lscqp_das_lds_bytes(Mx, dx, kmax, cacheC, stage) > lscqp::kMaxLdsBytes) stage = 0;
            if (lscqp_das_lds_bytes(Mx, dx, kmax, cacheC, stage) > lscqp::kMaxLdsBytes) cacheC = 0;
            if (tiny && kn.das_kmax < 0 && kmax > 20 && lscqp_das_lds_bytes(Mx, dx, kmax, knob(kn.das_cache, 1), knob(kn.das_stage, 1) ? n_obs_max * 6 * Mx : 0) > lscqp::kMaxLdsBytes) {
                // (the larger budget never at the price of the staged rows or the table copy: 128 x M10 x 40 runs 8 % slower without them)
                kmax = 20;
                cacheC = knob(kn.das_cache, 1);
                stage = knob(kn.das_stage, 1) ? n_obs_max * 6 * Mx : 0;
                if (lscqp_das_lds_bytes(Mx, dx, kmax, cacheC, stage) > lscqp::kMaxLdsBytes) stage = 0;
                if (lscqp_das_lds_bytes(Mx, dx, kmax, cacheC, stage) > lscqp::kMaxLdsBytes) cacheC = 0;
            }
            while (kmax > 4 && lscqp_das_lds_bytes(Mx, dx, kmax, cacheC, stage) > lscqp::kMaxLdsBytes) kmax -= 4;
            if (lscqp_das_lds_bytes(Mx, dx, kmax, cacheC, stage) <= lscqp::kMaxLdsBytes) {
                e = lscqp_launch_das(&cls, Mx, dx, h->es, cap, threads, kmax, steps, cacheC, stage, screen, d_tab, n, d_hdr, d_rows, d_row_offsets, d_sfc,
                                     d_x_init, d_x_out, d_obj_out, d_status_out, d_info_out, (hipStream_t)stream);
                if (e != hipSuccess) return fail(LSCQP_ERR_HIP, std::string("HIP launch failed (dual active-set phase): ") + hipGetErrorString(e));
                das_ran = true;
            }
        }
        if (h->desc.active_set == LSCQP_ACTIVE_SET_ONLY) {
            if (!das_ran) return fail(LSCQP_ERR_UNSUPPORTED, "LSCQP_ACTIVE_SET_ONLY: the active-set phase could not run (no tables on this device, or capacity)");
            return LSCQP_OK;
        }
        if (das_ran && deferred) {  // the caller looks at the statuses first
            *deferred = 1;
            return LSCQP_OK;
        }
    }
    const int first_repair = das_ran ? 3 : 0;
    if (!inst || !inst64) {
        // no compiled instance serves this launch (shape without one, or more obstacles than its register slots hold): the
        // run-time-shaped kernel, fp64.  Same statuses, same second pass.
        if (mixed || n_obs_max > lscqp_generic_max_obstacles(h->desc.M, h->desc.dim, h->es)) {
            char buf[240];
            snprintf(buf, sizeof buf, "no kernel of M=%d dim=%d%s holds %d obstacles per agent (compiled instances and the run-time-shaped kernel: %d)",
                     h->desc.M, h->desc.dim, mixed ? " (mixed precision)" : "", n_obs_max, lscqp_max_obstacles(h));
            return fail(LSCQP_ERR_UNSUPPORTED, buf);
        }
        if (retry == -2) return LSCQP_OK;  // (no other elimination order to try)
        if (retry == -3) {  // (internal) only the rescue pass
            cls.repair = 2;
            e = lscqp_launch_generic(&cls, h->desc.M, h->desc.dim, h->es, n, d_hdr, d_rows, d_row_offsets, d_sfc, nullptr, d_x_out, d_obj_out, d_status_out,
                                     d_info_out, (hipStream_t)stream);
            if (e != hipSuccess) return fail(LSCQP_ERR_HIP, std::string("HIP launch failed (rescue pass): ") + hipGetErrorString(e));
            return LSCQP_OK;
        }
        cls.repair = first_repair;
        e = lscqp_launch_generic(&cls, h->desc.M, h->desc.dim, h->es, n, d_hdr, d_rows, d_row_offsets, d_sfc, d_x_init, d_x_out, d_obj_out, d_status_out,
                                 d_info_out, (hipStream_t)stream);
        if (e != hipSuccess) return fail(LSCQP_ERR_HIP, std::string("HIP launch failed (run-time-shaped kernel): ") + hipGetErrorString(e));
        if (retry && d_x_init) {
            cls.repair = 1;
            e = lscqp_launch_generic(&cls, h->desc.M, h->desc.dim, h->es, n, d_hdr, d_rows, d_row_offsets, d_sfc, nullptr, d_x_out, d_obj_out, d_status_out,
                                     d_info_out, (hipStream_t)stream);
            if (e != hipSuccess) return fail(LSCQP_ERR_HIP, std::string("HIP launch failed (run-time-shaped kernel, second pass): ") + hipGetErrorString(e));
        }
        if (retry == 2 || retry == 3) {
            cls.repair = 2;
            e = lscqp_launch_generic(&cls, h->desc.M, h->desc.dim, h->es, n, d_hdr, d_rows, d_row_offsets, d_sfc, nullptr, d_x_out, d_obj_out, d_status_out,
                                     d_info_out, (hipStream_t)stream);
            if (e != hipSuccess) return fail(LSCQP_ERR_HIP, std::string("HIP launch failed (rescue pass): ") + hipGetErrorString(e));
        }
        return LSCQP_OK;
    }
    // RESCUE pass (retry == -3: only it; retry == 2: after the other passes): what is still at the iteration limit or broke down
    // numerically goes through the run-time-shaped kernel once more with cls.repair = 2 (lscqp_generic.hip: weighted corrector)
    auto rescue = [&]() -> int {
        if (n_obs_max > lscqp_generic_max_obstacles(h->desc.M, h->desc.dim, h->es)) return LSCQP_OK;  // (the kernel cannot hold the batch: nothing to try)
        lscqp::DevClass rc_ = cls;
        rc_.repair = 2;
        rc_.queue = nullptr;
        const hipError_t er = lscqp_launch_generic(&rc_, h->desc.M, h->desc.dim, h->es, n, d_hdr, d_rows, d_row_offsets, d_sfc, nullptr, d_x_out, d_obj_out,
                                                   d_status_out, d_info_out, (hipStream_t)stream);
        if (er != hipSuccess) return fail(LSCQP_ERR_HIP, std::string("HIP launch failed (rescue pass): ") + hipGetErrorString(er));
        return LSCQP_OK;
    };
    if (retry == -3) return rescue();
    if (retry == -2) {  // (internal) only the repair pass, on the other-order instance: the statuses of a first pass are in d_status_out
        const Inst* other = other_order_instance(inst64, n_obs_max);
        if (!other) return LSCQP_OK;
        cls.repair = 1;
        with_queue(cls, other);
        e = other->fn(&cls, n, d_hdr, d_rows, d_row_offsets, d_sfc, nullptr, d_x_out, d_obj_out, d_status_out, d_info_out, (hipStream_t)stream);
        if (e != hipSuccess) return fail(LSCQP_ERR_HIP, std::string("HIP launch failed (other-order pass): ") + hipGetErrorString(e));
        return LSCQP_OK;
    }
    cls.repair = first_repair;
    das_in_front = das_ran;
    // Behind the phase the pass usually finds nothing to do, and what it costs then is its launch: n workgroups that load one status each and
    // leave (4096 x M5: 4.5 us and 33 MB of fetches per call; a mixed-precision class: two such launches, float32 then fp64).  So behind the
    // phase the pass runs on an fp64 instance of the same capacity that has the PERSIST form, in its scan mode (lscqp_kernel.hpp:
    // DevClass::scan): at most as many workgroups as the chip holds, each looking through 64 statuses per round trip.  A mixed-precision class
    // is served by that fp64 instance directly -- the float32 factorisation has nothing to add behind a phase that finishes the easy
    // instances, and its own second pass would be a third launch.  Without such an instance: as before.
    const Inst* first = inst;
    if (das_ran && !kn.behind_scan) {
        // MEASURED (round 6, profiles/r06_behind_scan.txt): the scan form LOSES -- 64 x M5 14.8 -> 17.3 us per call, 4096 x M5 41.0 -> 42.3 -- the
        // persistent form of the kernel pays more before its first status load than n one-status workgroups cost.  Off by default (knob
        // behind_scan); what stays is the mixed-precision class going straight to its fp64 instance behind the phase (one launch instead of two).
        first = mixed ? inst64 : inst;
    } else if (das_ran) {
        const int want = std::min(inst->max_obs, inst64->max_obs);
        if (!(inst->persist && !inst->mixed)) {
            const Inst* b = nullptr;
            for (const Inst& i : kInst)
                if (i.M == h->desc.M && i.dim == h->desc.dim && i.es == h->es && !i.mixed && i.persist && i.max_obs == want && (!kn.pin_waves || i.waves == kn.pin_waves) &&
                    (!b || i.waves < b->waves))
                    b = &i;
            first = b ? b : (mixed ? inst64 : inst);
        }
        cls.scan = (first->persist && !first->mixed) ? 1 : 0;
    }
    with_queue(cls, first);
    das_in_front = false;
    e = first->fn(&cls, n, d_hdr, d_rows, d_row_offsets, d_sfc, d_x_init, d_x_out, d_obj_out, d_status_out, d_info_out, (hipStream_t)stream);
    cls.scan = 0;
    if (e != hipSuccess) return fail(LSCQP_ERR_HIP, std::string("HIP launch failed: ") + hipGetErrorString(e));
    // Second pass over the batch, same stream, no host round trip: a workgroup whose instance is already OPTIMAL (or was
    // refused for capacity) returns at once.  Mixed precision: the fp64 kernel re-solves what the float32 factorisation could
    // not finish (same start).  retry: the fp64 kernel re-solves from the DEFAULT start what a warm start did not bring to
    // OPTIMAL -- a jammed or diverged warm start (ITER_LIMIT / NUMERIC, or relabelled INFEASIBLE on its primal residual) says
    // nothing about the problem, a cold start proves infeasibility independently of x_init.
    // retry = 2: the second pass on the instance of the other elimination order (also for cold batches).  Not the default of a retry:
    // the natural-order instances of the shapes that have both spill to scratch, and a kernel with a private segment costs ~35 us to
    // launch even when every workgroup returns at once (measured: 38.7 vs 4.6 us per call on the forest10 replica).
    const Inst* alt = retry == 2 ? other_order_instance(inst64, n_obs_max) : nullptr;
    if ((mixed && first->mixed) || (retry && (d_x_init || alt))) {
        cls.repair = 1;
        // (behind the phase the second pass, too, finds nothing on most batches: same instance, same scan form as the first)
        const Inst* second = alt ? alt : ((das_ran && kn.behind_scan && first->persist && !first->mixed) ? first : inst64);
        cls.scan = (das_ran && kn.behind_scan && second->persist) ? 1 : 0;
        with_queue(cls, cls.scan ? nullptr : second);
        e = second->fn(&cls, n, d_hdr, d_rows, d_row_offsets, d_sfc, (retry ? nullptr : d_x_init), d_x_out, d_obj_out, d_status_out,
                       d_info_out, (hipStream_t)stream);
        cls.scan = 0;
        if (e != hipSuccess) return fail(LSCQP_ERR_HIP, std::string("HIP launch failed (second pass): ") + hipGetErrorString(e));
    }
    if (retry == 2 || retry == 3) return rescue();
    return LSCQP_OK;
}

int lscqp_solve_batch_device(lscqp_handle h, int64_t n, int32_t n_obs_max, const lscqp_header* d_hdr,
                             const lscqp_row* d_rows, const uint64_t* d_row_offsets, const lscqp_box* d_sfc,
                             const double* d_x_init, double* d_x_out, double* d_obj_out, int32_t* d_status_out,
                             lscqp_info* d_info_out, void* stream) {
    return lscqp_solve_batch_device_ex(h, n, n_obs_max, d_hdr, d_rows, d_row_offsets, d_sfc, d_x_init, d_x_out, d_obj_out, d_status_out,
                                       d_info_out, 0, stream);
}

int lscqp_solve_batch(lscqp_handle h, int64_t n, const lscqp_header* hdr, const lscqp_row* rows,
                      const uint64_t* row_offsets, const lscqp_box* sfc, const double* x_init, double* x_out,
                      double* obj_out, int32_t* status_out, lscqp_info* info_out) {
    return lscqp_solve_batch_stream(h, n, hdr, rows, row_offsets, sfc, x_init, x_out, obj_out, status_out, info_out, nullptr);
}

int lscqp_solve_batch_stream(lscqp_handle h, int64_t n, const lscqp_header* hdr, const lscqp_row* rows,
                             const uint64_t* row_offsets, const lscqp_box* sfc, const double* x_init, double* x_out,
                             double* obj_out, int32_t* status_out, lscqp_info* info_out, void* stream) {
    if (!h) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null handle");
    if (n < 0) return fail(LSCQP_ERR_INVALID_ARGUMENT, "negative size");
    if (n == 0) return LSCQP_OK;
    if (!hdr || !x_out || !obj_out || !status_out) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null buffer");
    int ndev = 0;
    {
        const hipError_t de = hipGetDeviceCount(&ndev);
        if (de != hipSuccess || ndev == 0)
            return fail(LSCQP_ERR_NO_DEVICE, std::string("no HIP device: lscqp has no CPU fallback (hipGetDeviceCount: ") +
                                                 hipGetErrorString(de) + ", " + std::to_string(ndev) + " devices)");
    }
    int n_obs_max = 0;
    for (int64_t q = 0; q < n; q++) {
        if (hdr[q].n_obs < 0) return fail(LSCQP_ERR_INVALID_ARGUMENT, "negative n_obs");
        if (hdr[q].n_obs > n_obs_max) n_obs_max = hdr[q].n_obs;
    }
    if (n_obs_max > 0 && (!rows || !row_offsets)) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null row buffer");
    if (h->desc.use_sfc && !sfc) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null sfc buffer");
    const size_t n_rows = n_obs_max > 0 ? (size_t)row_offsets[n] : 0;
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t b_hdr = al(sizeof(lscqp_header) * n), b_rows = al(row_bytes(h) * n_rows),
                 b_off = al(sizeof(uint64_t) * (n + 1)), b_sfc = al(sizeof(lscqp_box) * n * h->desc.M),
                 b_x = al(sizeof(double) * n * h->nv), b_obj = al(sizeof(double) * n), b_st = al(sizeof(int32_t) * n),
                 b_info = al(sizeof(lscqp_info) * n), b_xi = x_init ? b_x : 0;
    // layout: [hdr | rows | offsets | sfc | x_init] = one contiguous input region, [x | obj | status | info] = one output region
    const size_t b_in = b_hdr + b_rows + b_off + b_sfc + b_xi, b_out = b_x + b_obj + b_st + b_info;
    const size_t total = b_in + b_out;
    lscqp::SlotGuard sg{*h->pool, h->pool->acquire(total)};
    if (!sg.slot) return fail(LSCQP_ERR_HIP, "staging allocation failed (hipMalloc / hipHostMalloc / stream)");
    hipStream_t st = stream ? (hipStream_t)stream : sg.slot->stream;  // the caller's stream, or the slot's private one
    char* const hbase = (char*)sg.slot->h;
    // Round 6: a SMALL call -- the unchanged simulator's one TrajOptimizer::solve at a time (src/multi_sync_simulator.cpp:357-362): 21 KB in,
    // 0.8 KB out -- runs on the pinned mirror itself: the kernels read their rows from host memory over the link and write the plan back into
    // it, and the two DMA copies (each a fixed ~8 us of submission and completion around a 1 us transfer) disappear from the call.
    const bool zero_copy = sg.slot->hd != nullptr && h->knobs.zero_copy_bytes > 0 && total <= (size_t)h->knobs.zero_copy_bytes;
    char* const dbase = zero_copy ? (char*)sg.slot->hd : (char*)sg.slot->d;
    const size_t o_hdr = 0, o_rows = o_hdr + b_hdr, o_off = o_rows + b_rows, o_sfc = o_off + b_off, o_xi = o_sfc + b_sfc,
                 o_x = b_in, o_obj = o_x + b_x, o_st = o_obj + b_obj, o_info = o_st + b_st;
    lscqp_header* d_hdr = (lscqp_header*)(dbase + o_hdr);
    lscqp_row* d_rows = (lscqp_row*)(dbase + o_rows);
    uint64_t* d_off = (uint64_t*)(dbase + o_off);
    lscqp_box* d_sfc = (lscqp_box*)(dbase + o_sfc);
    double* d_xi = x_init ? (double*)(dbase + o_xi) : nullptr;
    double* d_x = (double*)(dbase + o_x);
    double* d_obj = (double*)(dbase + o_obj);
    int32_t* d_st = (int32_t*)(dbase + o_st);
    lscqp_info* d_info = (lscqp_info*)(dbase + o_info);
#define LSCQP_CK(call)                                                                           \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) return fail(LSCQP_ERR_HIP, std::string(#call ": ") + hipGetErrorString(e_)); \
    } while (0)
    memcpy(hbase + o_hdr, hdr, sizeof(lscqp_header) * n);
    if (n_rows) memcpy(hbase + o_rows, rows, row_bytes(h) * n_rows);
    if (n_obs_max > 0) memcpy(hbase + o_off, row_offsets, sizeof(uint64_t) * (n + 1));
    else memset(hbase + o_off, 0, sizeof(uint64_t) * (n + 1));
    if (h->desc.use_sfc) memcpy(hbase + o_sfc, sfc, sizeof(lscqp_box) * n * h->desc.M);
    if (d_xi) memcpy(hbase + o_xi, x_init, sizeof(double) * n * h->nv);
    if (!zero_copy) LSCQP_CK(hipMemcpyAsync(dbase, hbase, b_in, hipMemcpyHostToDevice, st));
    // (retry = 1: instances a warm start did not bring to OPTIMAL are solved once more from the default start by a second pass
    // on the device, before the results are copied back)
    // The interior-point passes behind the dual active-set phase are enqueued with the phase only when the handle's previous host-pointer call
    // needed them (a swarm in a hard stretch keeps needing them: no second round trip then); otherwise the statuses are read with the results
    // and the passes follow only if the phase left an instance -- on quiet batches it never does, and the launch of a kernel that finds nothing
    // to do is 2 - 4 us of a 20 us call.
    int deferred = 0;
    const bool speculate = h->knobs.defer_behind && h->behind_needed && h->behind_needed->load(std::memory_order_relaxed) == 0;
    int rc = lscqp_solve_batch_device_internal_(h, n, n_obs_max, d_hdr, d_rows, d_off, d_sfc, d_xi, d_x, d_obj, d_st, d_info, 1, nullptr, st,
                                                speculate ? &deferred : nullptr);
    if (rc != LSCQP_OK) return rc;
    if (!zero_copy) LSCQP_CK(hipMemcpyAsync(hbase + b_in, dbase + b_in, b_out, hipMemcpyDeviceToHost, st));
    LSCQP_CK(hipStreamSynchronize(st));
    {
        const int32_t* st_h = (const int32_t*)(hbase + o_st);
        bool left = false;  // did the phase hand an instance over?  (its mark: ITER_LIMIT)
        if (deferred) {
            for (int64_t q = 0; q < n && !left; q++) left = st_h[q] == LSCQP_STATUS_ITER_LIMIT;  // (OPTIMAL and a PROVEN INFEASIBLE are final)
            if (left) {
                rc = lscqp_solve_batch_device_internal_(h, n, n_obs_max, d_hdr, d_rows, d_off, d_sfc, d_xi, d_x, d_obj, d_st, d_info, -11, nullptr, st, nullptr);
                if (rc != LSCQP_OK) return rc;
                if (!zero_copy) LSCQP_CK(hipMemcpyAsync(hbase + b_in, dbase + b_in, b_out, hipMemcpyDeviceToHost, st));
                LSCQP_CK(hipStreamSynchronize(st));
            }
            if (h->behind_needed) h->behind_needed->store(left ? 1 : 0, std::memory_order_relaxed);
        } else if (h->behind_needed && h->knobs.defer_behind && d_info) {
            // (the passes ran with the phase: keep doing that while the phase keeps leaving work -- lscqp_info tells who finished an instance)
            const lscqp_info* in_h = (const lscqp_info*)(hbase + o_info);
            bool any_ip = false;
            for (int64_t q = 0; q < n && !any_ip; q++) any_ip = !(in_h[q].flags & LSCQP_INFO_ACTIVE_SET);
            h->behind_needed->store(any_ip ? 1 : 0, std::memory_order_relaxed);
        }
    }
    {   // instances that are still not OPTIMAL get one more pass where the shape has an instance with the other elimination order
        // (a factorisation that breaks down in one order usually survives in the other); only batches with such instances pay for it
        const int32_t* st_h = (const int32_t*)(hbase + o_st);
        bool any = false;
        for (int64_t q = 0; q < n && !any; q++) any = st_h[q] != LSCQP_STATUS_OPTIMAL && st_h[q] != LSCQP_STATUS_CAPACITY;
        const Inst* first = any ? find_instance(h->knobs, h->desc.M, h->desc.dim, h->es, 0, n_obs_max, n, cu_count()) : nullptr;
        if (first && other_order_instance(first, n_obs_max)) {
            rc = lscqp_solve_batch_device_internal_(h, n, n_obs_max, d_hdr, d_rows, d_off, d_sfc, nullptr, d_x, d_obj, d_st, d_info, -2, nullptr, st, nullptr);
            if (rc != LSCQP_OK) return rc;
            if (!zero_copy) LSCQP_CK(hipMemcpyAsync(hbase + b_in, dbase + b_in, b_out, hipMemcpyDeviceToHost, st));
            LSCQP_CK(hipStreamSynchronize(st));
        }
        // ... and what is STILL at the iteration limit or broke down numerically gets the rescue pass (run-time-shaped kernel, weighted
        // corrector: LSCQP_INFO_RESCUED); again only batches with such an instance pay for it
        bool lim = false;
        for (int64_t q = 0; q < n && !lim; q++) lim = st_h[q] == LSCQP_STATUS_ITER_LIMIT || st_h[q] == LSCQP_STATUS_NUMERIC;
        if (lim && n_obs_max <= lscqp_generic_max_obstacles(h->desc.M, h->desc.dim, h->es)) {
            rc = lscqp_solve_batch_device_internal_(h, n, n_obs_max, d_hdr, d_rows, d_off, d_sfc, nullptr, d_x, d_obj, d_st, d_info, -3, nullptr, st, nullptr);
            if (rc != LSCQP_OK) return rc;
            if (!zero_copy) LSCQP_CK(hipMemcpyAsync(hbase + b_in, dbase + b_in, b_out, hipMemcpyDeviceToHost, st));
            LSCQP_CK(hipStreamSynchronize(st));
        }
    }
    memcpy(x_out, hbase + o_x, sizeof(double) * n * h->nv);
    memcpy(obj_out, hbase + o_obj, sizeof(double) * n);
    memcpy(status_out, hbase + o_st, sizeof(int32_t) * n);
    if (info_out) memcpy(info_out, hbase + o_info, sizeof(lscqp_info) * n);
#undef LSCQP_CK
    return LSCQP_OK;
}

int lscqp_has_other_order_(lscqp_handle h, int64_t n, int32_t n_obs_max) {
    const Inst* first = find_instance(h->knobs, h->desc.M, h->desc.dim, h->es, 0, n_obs_max, n, cu_count());
    return (first && other_order_instance(first, n_obs_max)) ? 1 : 0;
}

const char* lscqp_last_error(void) { return g_err.c_str(); }

const char* lscqp_version(void) { return "lscqp 0.8 (gfx950, fp64 / mixed-precision PDIP, 1-4 wavefronts per QP; constraint generation LSC/CLSC/BVC, corridors, goal LP, post-solve step, safety metrics, whole-replan chain + hipGraph, sharded over a communicator, failure diagnostics)"; }

}  // extern "C"
