// lscqp_comm.hip — the agent batch over the GPUs of one node, behind the C ABI (include/lscqp.h, "multi-GPU").
//
// Device analogue of the one exchange the reference has: MultiSyncSimulator::broadcastMsgs (reference
// src/multi_sync_simulator.cpp:305-352) hands every agent the previous plans of the others, then plan() solves the N
// independent QPs of the step (:354-362).  Here ONE PROCESS drives G devices (the C++/ROS host of the reference is one
// process): the batch is cut into contiguous blocks of ceil(N / G) agents in id order, each block is
// staged to, solved on and fetched from its own device on that device's private stream, all devices concurrently, with NO
// collective on the solve path.  The exchange step is one RCCL all-gather of the solved control points over xGMI
// (ncclCommInitAll, one communicator per device, grouped ncclAllGather) for callers that keep the next step's constraint
// generation on the devices; the library states when a batch is worth spreading at all (lscqp_comm_devices_for).
//
// RCCL is bound at run time (dlopen of librccl.so.1): liblscqp.so itself does not link against it, a single-GPU host needs
// no RCCL at all, and inside a torch process the loader hands back the librccl torch already loaded.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/lscqp.h"
#include "lscqp_staging.hpp"

extern "C" int lscqp_set_error_(int code, const char* msg);
extern "C" const lscqp_plan_desc* lscqp_plan_desc_of_(lscqp_plan p);  // lscplan.hip
extern "C" int lscqp_plan_device_(lscqp_plan p);
extern "C" int lscqp_solve_batch_device_internal_(lscqp_handle h, int64_t n, int32_t n_obs_max, const lscqp_header* d_hdr, const lscqp_row* d_rows,
                                                  const uint64_t* d_row_offsets, const lscqp_box* d_sfc, const double* d_x_init, double* d_x_out, double* d_obj_out,
                                                  int32_t* d_status_out, lscqp_info* d_info_out, int32_t retry, const int32_t* d_order, void* stream, int* deferred);
extern "C" int lscqp_has_other_order_(lscqp_handle h, int64_t n, int32_t n_obs_max);

namespace {

struct Rccl {
    void* lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    bool load(std::string& why) {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) {
            why = std::string("dlopen(librccl.so.1): ") + (dlerror() ? dlerror() : "not found");
            return false;
        }
#define LSCQP_SYM(field, sym)                                          \
    field = reinterpret_cast<decltype(field)>(dlsym(lib, #sym));       \
    if (!field) {                                                      \
        why = "librccl: symbol " #sym " missing";                      \
        return false;                                                  \
    }
        LSCQP_SYM(CommInitAll, ncclCommInitAll)
        LSCQP_SYM(CommDestroy, ncclCommDestroy)
        LSCQP_SYM(AllGather, ncclAllGather)
        LSCQP_SYM(Broadcast, ncclBroadcast)
        LSCQP_SYM(GroupStart, ncclGroupStart)
        LSCQP_SYM(GroupEnd, ncclGroupEnd)
        LSCQP_SYM(GetErrorString, ncclGetErrorString)
        LSCQP_SYM(GetVersion, ncclGetVersion)
#undef LSCQP_SYM
        return true;
    }
};

int fail(int code, const std::string& m) { return lscqp_set_error_(code, m.c_str()); }

struct DeviceGuard {  // the caller's current device is restored on every return path
    int prev = 0;
    DeviceGuard() { (void)hipGetDevice(&prev); }
    ~DeviceGuard() { (void)hipSetDevice(prev); }
};

}  // namespace

struct lscqp_comm_s {
    int G = 0;
    std::vector<int> dev;
    std::vector<hipStream_t> stream;
    std::vector<lscqp::StagePool*> pool;  // per device
    Rccl rccl;
    bool rccl_ok = false;
    std::vector<ncclComm_t> comms;
    std::string backend;
    int64_t min_agents_per_device = 256;  // see lscqp_comm_devices_for
    bool min_agents_set = false;          // the caller named the threshold: it overrides the class's own (lscqp_comm_devices_for_class)
    std::mutex mu;                        // one sharded call at a time per communicator (RCCL groups must not interleave)
};

extern "C" {

int lscqp_comm_create(int32_t n_devices, const int32_t* device_ids, lscqp_comm* out) {
    if (!out) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(LSCQP_ERR_NO_DEVICE, "no HIP device: lscqp has no CPU fallback");
    if (n_devices <= 0) n_devices = ndev;
    if (n_devices > ndev && !device_ids) return fail(LSCQP_ERR_INVALID_ARGUMENT, "more devices requested than visible");
    DeviceGuard dg;
    lscqp_comm_s* c = new lscqp_comm_s();
    c->G = n_devices;
    for (int g = 0; g < n_devices; g++) {
        const int d = device_ids ? device_ids[g] : g;
        if (d < 0 || d >= ndev) {
            delete c;
            return fail(LSCQP_ERR_INVALID_ARGUMENT, "device id out of range");
        }
        c->dev.push_back(d);
    }
    for (int g = 0; g < c->G; g++) {
        hipStream_t s = nullptr;
        if (hipSetDevice(c->dev[g]) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
            for (hipStream_t t : c->stream) (void)hipStreamDestroy(t);
            for (auto* p : c->pool) delete p;
            delete c;
            return fail(LSCQP_ERR_HIP, "stream creation failed");
        }
        c->stream.push_back(s);
        c->pool.push_back(new lscqp::StagePool());
    }
    // RCCL: one communicator per device of this process (ncclCommInitAll).  A one-device communicator is initialised too, so
    // the collective path is the same code on a single-GPU box; without RCCL a single device degrades to a device copy and
    // several devices are refused.
    std::string why;
    const char* off = getenv("LSCQP_NO_RCCL");
    if (!(off && off[0] == '1') && c->rccl.load(why)) {
        c->comms.resize(c->G);
        const ncclResult_t r = c->rccl.CommInitAll(c->comms.data(), c->G, c->dev.data());
        if (r == ncclSuccess) {
            int ver = 0;
            (void)c->rccl.GetVersion(&ver);
            char buf[96];
            snprintf(buf, sizeof buf, "rccl %d (ncclCommInitAll, %d device%s)", ver, c->G, c->G == 1 ? "" : "s");
            c->backend = buf;
            c->rccl_ok = true;
        } else {
            why = std::string("ncclCommInitAll: ") + c->rccl.GetErrorString(r);
            c->comms.clear();
        }
    } else if (off && off[0] == '1') {
        why = "disabled by LSCQP_NO_RCCL=1";
    }
    if (!c->rccl_ok) {
        if (c->G > 1) {
            for (hipStream_t t : c->stream) (void)hipStreamDestroy(t);
            for (auto* p : c->pool) delete p;
            const std::string msg = "RCCL unavailable (" + why + "): a communicator over several devices needs it";
            delete c;
            return fail(LSCQP_ERR_HIP, msg);
        }
        c->backend = "none: single device, all-gather is a device copy (" + why + ")";
    }
    *out = c;
    return LSCQP_OK;
}

void lscqp_comm_destroy(lscqp_comm c) {
    if (!c) return;
    DeviceGuard dg;
    for (int g = 0; g < c->G; g++) {
        (void)hipSetDevice(c->dev[g]);
        (void)hipStreamSynchronize(c->stream[g]);
        if (c->rccl_ok) (void)c->rccl.CommDestroy(c->comms[g]);
        (void)hipStreamDestroy(c->stream[g]);
        delete c->pool[g];
    }
    delete c;
}

int32_t lscqp_comm_size(lscqp_comm c) { return c ? c->G : -1; }
int32_t lscqp_comm_device(lscqp_comm c, int32_t g) { return (c && g >= 0 && g < c->G) ? c->dev[g] : -1; }
void* lscqp_comm_stream(lscqp_comm c, int32_t g) { return (c && g >= 0 && g < c->G) ? (void*)c->stream[g] : nullptr; }
const char* lscqp_comm_backend(lscqp_comm c) { return c ? c->backend.c_str() : ""; }

int lscqp_comm_set_min_agents_per_device(lscqp_comm c, int64_t n) {
    if (!c || n < 1) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null communicator or n < 1");
    c->min_agents_per_device = n;
    c->min_agents_set = true;
    return LSCQP_OK;
}

int32_t lscqp_comm_devices_for(lscqp_comm c, int64_t n) {
    if (!c) return -1;
    int64_t g = n / c->min_agents_per_device;
    if (g < 1) g = 1;
    if (g > c->G) g = c->G;
    return (int32_t)g;
}

// The spreading rule with the class in hand (round 5): a device is given work only where ONE device would need more than one round of
// workgroups for the batch -- lscqp_device_fill(h) instances are in flight on a device at once (3072 with the dual active-set phase at
// M = 5, one launch of which lasts ~50 us for 4096 QPs: far below the cost of an exchange over xGMI) -- clamp(n / fill, 1, G).  A threshold
// the caller set with lscqp_comm_set_min_agents_per_device overrides it.
int32_t lscqp_comm_devices_for_class(lscqp_comm c, lscqp_handle h, int64_t n, int32_t n_obs_max) {
    if (!c) return -1;
    if (c->min_agents_set || !h) return lscqp_comm_devices_for(c, n);
    DeviceGuard dg;
    (void)hipSetDevice(c->dev[0]);
    const int64_t fill = lscqp_device_fill(h, n, n_obs_max);
    if (fill <= 0) return lscqp_comm_devices_for(c, n);
    int64_t g = n / fill;
    if (g < 1) g = 1;
    if (g > c->G) g = c->G;
    return (int32_t)g;
}

// The class's active-set tables on every device of the communicator, now (lscqp_prepare_device per device): a device's first solve would
// otherwise load them lazily -- which a launch inside a stream capture cannot do (it fails loudly instead of running without the phase).
int lscqp_comm_prepare(lscqp_comm c, lscqp_handle h) {
    if (!c || !h) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null argument");
    DeviceGuard dg;
    for (int g = 0; g < c->G; g++) {
        if (hipSetDevice(c->dev[g]) != hipSuccess) return fail(LSCQP_ERR_HIP, "hipSetDevice failed");
        const int rc = lscqp_prepare_device(h);
        if (rc != LSCQP_OK) return rc;
    }
    return LSCQP_OK;
}

int lscqp_shard_range(int64_t n, int32_t n_used, int32_t g, int64_t* first, int64_t* count) {
    if (!first || !count || n < 0 || n_used < 1 || g < 0) return fail(LSCQP_ERR_INVALID_ARGUMENT, "bad shard query");
    const int64_t blk = (n + n_used - 1) / n_used;  // ceil(N / G), SURVEY.md section 8e
    int64_t f = blk * g, e = f + blk;
    if (f > n) f = n;
    if (e > n) e = n;
    if (g >= n_used) f = e = n;
    *first = f;
    *count = e - f;
    return LSCQP_OK;
}

int lscqp_exchange_schedule(int64_t n_total, int32_t n_devices, const int64_t* first, const int64_t* count, int64_t per,
                            lscqp_exchange_op* ops, int32_t max_ops, int32_t* n_ops) {
    return lscqp_exchange_schedule_padded(n_total, n_devices, first, count, per, 0, ops, max_ops, n_ops);
}

int lscqp_exchange_schedule_padded(int64_t n_total, int32_t n_devices, const int64_t* first, const int64_t* count, int64_t per, int64_t pad_agents,
                                   lscqp_exchange_op* ops, int32_t max_ops, int32_t* n_ops) {
    if (!first || !count || !ops || !n_ops || n_devices < 1 || n_total < 0 || per < 0 || pad_agents < 0) return fail(LSCQP_ERR_INVALID_ARGUMENT, "bad exchange query");
    int64_t next = 0;
    bool equal = true;
    for (int g = 0; g < n_devices; g++) {
        if (count[g] < 0) return fail(LSCQP_ERR_INVALID_ARGUMENT, "negative block size");
        if (first[g] != next) return fail(LSCQP_ERR_INVALID_ARGUMENT, "the plans of a group own consecutive blocks of agents in device order");
        next += count[g];
        if (count[g] != count[0]) equal = false;
    }
    if (next != n_total) return fail(LSCQP_ERR_INVALID_ARGUMENT, "the blocks of the group do not cover the mission (sum of n_agents != n_total)");
    // the blocks of lscqp_shard_range -- B = ceil(n / G) agents at multiples of B, one shorter, the rest empty -- with room behind the mission
    // for what the short and the empty ones send beyond their agents
    bool ceil_split = !equal && count[0] > 0;
    if (ceil_split) {
        const int64_t B = count[0];
        for (int g = 0; g < n_devices && ceil_split; g++) {
            const int64_t at = std::min<int64_t>((int64_t)g * B, n_total), want = std::min<int64_t>(B, n_total - at);
            if (first[g] != at || count[g] != want) ceil_split = false;
        }
        if ((int64_t)n_devices * B - n_total > pad_agents) ceil_split = false;
    }
    int k = 0;
    if (equal || ceil_split) {
        if (max_ops < 1) return fail(LSCQP_ERR_INVALID_ARGUMENT, "ops array too small");
        ops[k++] = {LSCQP_XCHG_ALLGATHER, -1, 0, count[0] * per};
    } else {
        for (int o = 0; o < n_devices; o++) {
            if (count[o] == 0) continue;  // an empty block owns nothing (every device skips it alike)
            if (k >= max_ops) return fail(LSCQP_ERR_INVALID_ARGUMENT, "ops array too small");
            ops[k++] = {LSCQP_XCHG_BROADCAST, o, first[o] * per, count[o] * per};
        }
    }
    *n_ops = k;
    return LSCQP_OK;
}

int lscqp_comm_shard(lscqp_comm c, int64_t n, int32_t n_used, int32_t g, int64_t* first, int64_t* count) {
    if (!c || n_used > c->G || g >= c->G) return fail(LSCQP_ERR_INVALID_ARGUMENT, "bad shard query");
    return lscqp_shard_range(n, n_used, g, first, count);
}

int lscqp_comm_synchronize(lscqp_comm c) {
    if (!c) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null communicator");
    DeviceGuard dg;
    for (int g = 0; g < c->G; g++) {
        if (hipSetDevice(c->dev[g]) != hipSuccess) return fail(LSCQP_ERR_HIP, "hipSetDevice failed");
        const hipError_t e = hipStreamSynchronize(c->stream[g]);
        if (e != hipSuccess) return fail(LSCQP_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
    }
    return LSCQP_OK;
}

int lscqp_solve_batch_sharded_device(lscqp_handle h, lscqp_comm c, const int64_t* n, int32_t n_obs_max, const lscqp_header* const* d_hdr,
                                     const lscqp_row* const* d_rows, const uint64_t* const* d_row_offsets, const lscqp_box* const* d_sfc,
                                     const double* const* d_x_init, double* const* d_x_out, double* const* d_obj_out,
                                     int32_t* const* d_status_out, lscqp_info* const* d_info_out, int32_t retry) {
    if (!h || !c || !n || !d_hdr || !d_x_out || !d_obj_out || !d_status_out) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null argument");
    DeviceGuard dg;
    for (int g = 0; g < c->G; g++) {
        if (n[g] <= 0) continue;
        if (hipSetDevice(c->dev[g]) != hipSuccess) return fail(LSCQP_ERR_HIP, "hipSetDevice failed");
        const int rc = lscqp_solve_batch_device_ex(h, n[g], n_obs_max, d_hdr[g], d_rows ? d_rows[g] : nullptr,
                                                   d_row_offsets ? d_row_offsets[g] : nullptr, d_sfc ? d_sfc[g] : nullptr,
                                                   d_x_init ? d_x_init[g] : nullptr, d_x_out[g], d_obj_out[g], d_status_out[g],
                                                   d_info_out ? d_info_out[g] : nullptr, retry, c->stream[g]);
        if (rc != LSCQP_OK) return rc;
    }
    return LSCQP_OK;
}

int lscqp_allgather(lscqp_comm c, const double* const* d_send, double* const* d_recv, int64_t count) {
    if (!c || !d_send || !d_recv || count < 0) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null argument or negative count");
    if (count == 0) return LSCQP_OK;
    DeviceGuard dg;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->rccl_ok) {  // (single device only, see lscqp_comm_create)
        if (hipSetDevice(c->dev[0]) != hipSuccess) return fail(LSCQP_ERR_HIP, "hipSetDevice failed");
        const hipError_t e = hipMemcpyAsync(d_recv[0], d_send[0], sizeof(double) * count, hipMemcpyDeviceToDevice, c->stream[0]);
        if (e != hipSuccess) return fail(LSCQP_ERR_HIP, std::string("hipMemcpyAsync: ") + hipGetErrorString(e));
        return LSCQP_OK;
    }
    ncclResult_t r = c->rccl.GroupStart();
    bool dev_failed = false;
    for (int g = 0; g < c->G && r == ncclSuccess && !dev_failed; g++) {
        if (hipSetDevice(c->dev[g]) != hipSuccess) {
            dev_failed = true;  // (the group is closed below on every path: a thread left in group mode poisons every later RCCL call)
            break;
        }
        r = c->rccl.AllGather(d_send[g], d_recv[g], (size_t)count, ncclDouble, c->comms[g], c->stream[g]);
    }
    const ncclResult_t r2 = c->rccl.GroupEnd();
    if (dev_failed) return fail(LSCQP_ERR_HIP, "hipSetDevice failed");
    if (r == ncclSuccess) r = r2;
    if (r != ncclSuccess) return fail(LSCQP_ERR_HIP, std::string("ncclAllGather: ") + c->rccl.GetErrorString(r));
    return LSCQP_OK;
}

int lscqp_solve_batch_sharded(lscqp_handle h, lscqp_comm c, int64_t n, const lscqp_header* hdr, const lscqp_row* rows,
                              const uint64_t* row_offsets, const lscqp_box* sfc, const double* x_init, double* x_out, double* obj_out,
                              int32_t* status_out, lscqp_info* info_out, int32_t* n_devices_used) {
    if (!h || !c) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null handle");
    if (n < 0) return fail(LSCQP_ERR_INVALID_ARGUMENT, "negative size");
    if (n_devices_used) *n_devices_used = 0;
    if (n == 0) return LSCQP_OK;
    if (!hdr || !x_out || !obj_out || !status_out) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null buffer");
    const int nv = lscqp_num_variables(h);
    int n_obs_max = 0;
    for (int64_t q = 0; q < n; q++) {
        if (hdr[q].n_obs < 0) return fail(LSCQP_ERR_INVALID_ARGUMENT, "negative n_obs");
        if (hdr[q].n_obs > n_obs_max) n_obs_max = hdr[q].n_obs;
    }
    if (n_obs_max > 0 && (!rows || !row_offsets)) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null row buffer");
    const int segs = lscqp_num_segments(h);
    const int use_sfc = lscqp_uses_sfc(h);
    if (use_sfc && !sfc) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null sfc buffer");
    const size_t rb = (size_t)lscqp_row_bytes(h);
    const int G = lscqp_comm_devices_for_class(c, h, n, n_obs_max);
    if (n_devices_used) *n_devices_used = G;
    DeviceGuard dg;
    std::lock_guard<std::mutex> lk(c->mu);
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    struct Shard {
        int64_t first = 0, cnt = 0;
        lscqp::StageSlot* slot = nullptr;
        size_t b_in = 0, b_out = 0, o_x = 0, o_obj = 0, o_st = 0, o_info = 0;
    };
    std::vector<Shard> sh(G);
    int rc = LSCQP_OK;
    // phase 1: stage + launch on every device (asynchronous), phase 2: wait + copy out
    for (int g = 0; g < G && rc == LSCQP_OK; g++) {
        Shard& S = sh[g];
        lscqp_comm_shard(c, n, G, g, &S.first, &S.cnt);
        if (S.cnt == 0) continue;
        if (hipSetDevice(c->dev[g]) != hipSuccess) {
            rc = fail(LSCQP_ERR_HIP, "hipSetDevice failed");
            break;
        }
        const int64_t m = S.cnt;
        const uint64_t r0 = n_obs_max > 0 ? row_offsets[S.first] : 0, r1 = n_obs_max > 0 ? row_offsets[S.first + m] : 0;
        const size_t n_rows = (size_t)(r1 - r0);
        const size_t b_hdr = al(sizeof(lscqp_header) * m), b_rows = al(rb * n_rows), b_off = al(sizeof(uint64_t) * (m + 1)),
                     b_sfc = al(sizeof(lscqp_box) * m * segs), b_x = al(sizeof(double) * m * nv), b_obj = al(sizeof(double) * m),
                     b_st = al(sizeof(int32_t) * m), b_info = al(sizeof(lscqp_info) * m), b_xi = x_init ? b_x : 0;
        S.b_in = b_hdr + b_rows + b_off + b_sfc + b_xi;
        S.b_out = b_x + b_obj + b_st + b_info;
        S.slot = c->pool[g]->acquire(S.b_in + S.b_out);
        if (!S.slot) {
            rc = fail(LSCQP_ERR_HIP, "staging allocation failed");
            break;
        }
        char* const hb = (char*)S.slot->h;
        char* const db = (char*)S.slot->d;
        const size_t o_hdr = 0, o_rows = b_hdr, o_off = o_rows + b_rows, o_sfc = o_off + b_off, o_xi = o_sfc + b_sfc;
        S.o_x = S.b_in, S.o_obj = S.o_x + b_x, S.o_st = S.o_obj + b_obj, S.o_info = S.o_st + b_st;
        memcpy(hb + o_hdr, hdr + S.first, sizeof(lscqp_header) * m);
        if (n_rows) memcpy(hb + o_rows, reinterpret_cast<const char*>(rows) + rb * r0, rb * n_rows);
        uint64_t* off = reinterpret_cast<uint64_t*>(hb + o_off);
        for (int64_t i = 0; i <= m; i++) off[i] = n_obs_max > 0 ? row_offsets[S.first + i] - r0 : 0;  // rebased to the block
        if (use_sfc) memcpy(hb + o_sfc, sfc + S.first * segs, sizeof(lscqp_box) * m * segs);
        if (x_init) memcpy(hb + o_xi, x_init + S.first * nv, sizeof(double) * m * nv);
        hipStream_t st = c->stream[g];
        if (hipMemcpyAsync(db, hb, S.b_in, hipMemcpyHostToDevice, st) != hipSuccess) {
            rc = fail(LSCQP_ERR_HIP, "hipMemcpyAsync (H2D) failed");
            break;
        }
        rc = lscqp_solve_batch_device_ex(h, m, n_obs_max, (const lscqp_header*)(db + o_hdr), (const lscqp_row*)(db + o_rows),
                                         (const uint64_t*)(db + o_off), use_sfc ? (const lscqp_box*)(db + o_sfc) : nullptr,
                                         x_init ? (const double*)(db + o_xi) : nullptr, (double*)(db + S.o_x), (double*)(db + S.o_obj),
                                         (int32_t*)(db + S.o_st), (lscqp_info*)(db + S.o_info), 1, st);
        if (rc != LSCQP_OK) break;
        if (hipMemcpyAsync(hb + S.b_in, db + S.b_in, S.b_out, hipMemcpyDeviceToHost, st) != hipSuccess) rc = fail(LSCQP_ERR_HIP, "hipMemcpyAsync (D2H) failed");
    }
    for (int g = 0; g < G; g++) {
        Shard& S = sh[g];
        if (!S.slot) continue;
        (void)hipSetDevice(c->dev[g]);
        hipError_t e = hipStreamSynchronize(c->stream[g]);
        if (e != hipSuccess && rc == LSCQP_OK) rc = fail(LSCQP_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
        if (rc == LSCQP_OK) {
            // what lscqp_solve_batch does for a single device: instances that are still not OPTIMAL get one more pass on the instance
            // with the other elimination order, where the shape has one -- so a sharded batch returns what the one-device call returns
            const int32_t* st_h = (const int32_t*)((const char*)S.slot->h + S.o_st);
            bool any = false;
            for (int64_t q = 0; q < S.cnt && !any; q++) any = st_h[q] != LSCQP_STATUS_OPTIMAL && st_h[q] != LSCQP_STATUS_CAPACITY;
            if (any && lscqp_has_other_order_(h, S.cnt, n_obs_max)) {
                char* const db = (char*)S.slot->d;
                char* const hb2 = (char*)S.slot->h;
                const size_t b_hdr = al(sizeof(lscqp_header) * S.cnt);
                const uint64_t r0 = n_obs_max > 0 ? row_offsets[S.first] : 0, r1 = n_obs_max > 0 ? row_offsets[S.first + S.cnt] : 0;
                const size_t b_rows = al(rb * (size_t)(r1 - r0)), b_off = al(sizeof(uint64_t) * (S.cnt + 1));
                const size_t o_rows = b_hdr, o_off = o_rows + b_rows, o_sfc = o_off + b_off;
                rc = lscqp_solve_batch_device_internal_(h, S.cnt, n_obs_max, (const lscqp_header*)db, (const lscqp_row*)(db + o_rows),
                                                 (const uint64_t*)(db + o_off), use_sfc ? (const lscqp_box*)(db + o_sfc) : nullptr, nullptr,
                                                 (double*)(db + S.o_x), (double*)(db + S.o_obj), (int32_t*)(db + S.o_st),
                                                 (lscqp_info*)(db + S.o_info), -2, nullptr, c->stream[g], nullptr);
                if (rc == LSCQP_OK && hipMemcpyAsync(hb2 + S.b_in, db + S.b_in, S.b_out, hipMemcpyDeviceToHost, c->stream[g]) != hipSuccess)
                    rc = fail(LSCQP_ERR_HIP, "hipMemcpyAsync (D2H) failed");
                if (rc == LSCQP_OK && (e = hipStreamSynchronize(c->stream[g])) != hipSuccess)
                    rc = fail(LSCQP_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
            }
        }
        if (rc == LSCQP_OK) {
            const char* hb = (const char*)S.slot->h;
            memcpy(x_out + S.first * nv, hb + S.o_x, sizeof(double) * S.cnt * nv);
            memcpy(obj_out + S.first, hb + S.o_obj, sizeof(double) * S.cnt);
            memcpy(status_out + S.first, hb + S.o_st, sizeof(int32_t) * S.cnt);
            if (info_out) memcpy(info_out + S.first, hb + S.o_info, sizeof(lscqp_info) * S.cnt);
        }
        c->pool[g]->release(S.slot);
    }
    return rc;
}

// One replan of a mission whose agents are spread over the communicator's devices: plan g (created with device g of the communicator
// current, on its own handle and map) owns the agents [first_agent, first_agent + n_agents) of n_total.  Every plan's chain is
// enqueued on its device's stream, then the owners' slices of the previous plans, states and goal points -- what the next replan's
// obstacle prediction, range filter and constraint generation read of the OTHER agents (reference
// src/multi_sync_simulator.cpp:305-352, broadcastMsgs) -- are exchanged in place over RCCL on the same streams: one grouped
// ncclAllGather per buffer when the blocks are equal, grouped ncclBroadcasts (one per owner) when the last block is short.
// Asynchronous; lscqp_comm_synchronize waits.
int lscqp_plan_group_step(lscqp_comm c, const lscqp_plan* plans, int32_t use_graph) {
    if (!c || !plans) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null argument");
    const int G = c->G;
    int64_t n_total = -1;
    std::vector<int64_t> first(G), count(G);
    for (int g = 0; g < G; g++) {
        if (!plans[g]) return fail(LSCQP_ERR_INVALID_ARGUMENT, "null plan in the group (one plan per device of the communicator)");
        const lscqp_plan_desc* d = lscqp_plan_desc_of_(plans[g]);
        if (lscqp_plan_device_(plans[g]) != c->dev[g])
            return fail(LSCQP_ERR_INVALID_ARGUMENT, "plan g of the group must live on device g of the communicator");
        if (n_total < 0) n_total = d->n_total;
        if (d->n_total != n_total) return fail(LSCQP_ERR_INVALID_ARGUMENT, "the plans of a group describe the same mission: n_total differs");
        first[g] = d->first_agent;
        count[g] = d->n_agents;
    }
    // the exchange as a list of operations in AGENTS (per = 1; scaled by each buffer's doubles per agent below)
    std::vector<lscqp_exchange_op> ops(G);
    int32_t n_ops = 0;
    {
        const int rc = lscqp_exchange_schedule_padded(n_total, G, first.data(), count.data(), 1, LSCQP_PLAN_EXCHANGE_PAD, ops.data(), G, &n_ops);
        if (rc != LSCQP_OK) return rc;
    }
    DeviceGuard dg;
    std::lock_guard<std::mutex> lk(c->mu);
    for (int g = 0; g < G; g++) {
        int rc = hipSetDevice(c->dev[g]) == hipSuccess ? LSCQP_OK : fail(LSCQP_ERR_HIP, "hipSetDevice failed");
        if (rc == LSCQP_OK) rc = use_graph ? lscqp_plan_step_graph(plans[g], c->stream[g]) : lscqp_plan_step(plans[g], c->stream[g]);
        if (rc != LSCQP_OK) {
            // the plans before g have replanned and the exchange has not run: the devices now disagree about the mission.  Say so --
            // the caller has to lscqp_plan_reset the group (or re-upload the buffers) before it steps again.
            const std::string why = lscqp_last_error();
            char buf[160];
            snprintf(buf, sizeof buf, "lscqp_plan_group_step: plan %d of %d failed after plans 0..%d had advanced, no exchange ran (reset the group): ", g, G, g - 1);
            return fail(rc, buf + why);
        }
    }
    if (!c->rccl_ok) return LSCQP_OK;  // (single device without RCCL, see lscqp_comm_create: the one plan owns every agent)
    static const int kExchanged[3] = {LSCQP_PLAN_BUF_PLAN, LSCQP_PLAN_BUF_STATE, LSCQP_PLAN_BUF_GOAL};
    ncclResult_t r = c->rccl.GroupStart();
    for (int b = 0; b < 3 && r == ncclSuccess; b++) {
        for (int g = 0; g < G && r == ncclSuccess; g++) {
            if (hipSetDevice(c->dev[g]) != hipSuccess) {
                (void)c->rccl.GroupEnd();
                return fail(LSCQP_ERR_HIP, "hipSetDevice failed");
            }
            uint64_t bytes = 0;
            double* const base = (double*)lscqp_plan_buffer(plans[g], kExchanged[b], &bytes);
            const size_t per = (size_t)(bytes / sizeof(double) / (uint64_t)n_total);  // doubles per agent
            for (int k = 0; k < n_ops && r == ncclSuccess; k++) {
                const lscqp_exchange_op& op = ops[k];
                if (op.kind == LSCQP_XCHG_ALLGATHER) {  // in place: device g's own block is where the all-gather puts it -- g blocks of op.count
                    // agents into the buffer (= first[g] for every block that holds agents; an EMPTY block of a ragged mission sends from the padding)
                    r = c->rccl.AllGather(base + (size_t)g * (size_t)op.count * per, base, (size_t)op.count * per, ncclDouble, c->comms[g], c->stream[g]);
                } else {
                    double* const blk = base + (size_t)op.offset * per;
                    r = c->rccl.Broadcast(blk, blk, (size_t)op.count * per, ncclDouble, op.root, c->comms[g], c->stream[g]);
                }
            }
        }
    }
    const ncclResult_t r2 = c->rccl.GroupEnd();
    if (r == ncclSuccess) r = r2;
    if (r != ncclSuccess) return fail(LSCQP_ERR_HIP, std::string("RCCL exchange of the plans: ") + c->rccl.GetErrorString(r));
    return LSCQP_OK;
}

}  // extern "C"
