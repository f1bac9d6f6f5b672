// Staging of the HOST-pointer entry points: a small pool of (device buffer, pinned host mirror, stream) slots per handle.
// A call takes a free slot (or makes one), stages its inputs there, runs on the slot's OWN non-blocking stream (or the caller's)
// and gives the slot back -- so two threads planning different agents through the same solver / map handle never share a
// buffer or the legacy NULL stream, and steady-state calls allocate nothing ("thread-safe per handle +
// stream").  Slots belong to the device that was current when they were created.
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>
#include <vector>

namespace lscqp {

struct StageSlot {
    void* d = nullptr;
    void* h = nullptr;
    void* hd = nullptr;  // the pinned mirror as the DEVICE addresses it (hipHostGetDevicePointer): small calls run on it directly, without copies
    size_t cap = 0;
    int device = -1;
    hipStream_t stream = nullptr;
    bool busy = false;
};

class StagePool {
public:
    // a free slot of the current device holding at least `bytes`; nullptr on allocation failure (nothing leaks: a slot whose
    // second allocation failed keeps cap = 0 and is retried by the next call)
    StageSlot* acquire(size_t bytes) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return nullptr;
        StageSlot* s = nullptr;
        {
            std::lock_guard<std::mutex> g(mu_);
            for (StageSlot* c : slots_)
                if (!c->busy && c->device == dev) {
                    s = c;
                    break;
                }
            if (!s) {
                s = new StageSlot();
                s->device = dev;
                slots_.push_back(s);
            }
            s->busy = true;
        }
        if (!s->stream && hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) {
            s->stream = nullptr;
            release(s);
            return nullptr;
        }
        if (bytes > s->cap) {
            if (s->d) (void)hipFree(s->d);
            if (s->h) (void)hipHostFree(s->h);
            s->d = s->h = s->hd = nullptr;
            s->cap = 0;
            if (hipMalloc(&s->d, bytes) != hipSuccess) {
                s->d = nullptr;
                release(s);
                return nullptr;
            }
            if (hipHostMalloc(&s->h, bytes, hipHostMallocDefault) != hipSuccess) {
                (void)hipFree(s->d);
                s->d = s->h = nullptr;
                release(s);
                return nullptr;
            }
            if (hipHostGetDevicePointer(&s->hd, s->h, 0) != hipSuccess) {
                (void)hipGetLastError();
                s->hd = nullptr;  // (no mapped view: such a slot always copies)
            }
            s->cap = bytes;
        }
        return s;
    }
    void release(StageSlot* s) {
        std::lock_guard<std::mutex> g(mu_);
        s->busy = false;
    }
    ~StagePool() {
        for (StageSlot* s : slots_) {
            if (s->d) (void)hipFree(s->d);
            if (s->h) (void)hipHostFree(s->h);
            if (s->stream) (void)hipStreamDestroy(s->stream);
            delete s;
        }
    }

private:
    std::mutex mu_;
    std::vector<StageSlot*> slots_;
};

struct SlotGuard {  // gives the slot back on every return path
    StagePool& pool;
    StageSlot* slot;
    ~SlotGuard() {
        if (slot) pool.release(slot);
    }
};

}  // namespace lscqp
