// Development aid: single-wavefront latency / issue micro-benchmarks for the instruction patterns the PDIP kernel is
// made of (gfx950).  One wave per workgroup, one workgroup: what a wave sees when it is alone on its SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

#define N 2048

__device__ __forceinline__ double bcast(double v, int lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ unsigned long long now() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    unsigned long long t = __builtin_readcyclecounter();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    return t;
}
__device__ __forceinline__ double fast_rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;
}

__global__ __launch_bounds__(64) void k(double* out, unsigned long long* cyc, double seed, int reps) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    double a = seed + lane * 1e-3, b = 1.0 + 1e-9 * lane, c = 1e-7;
    unsigned long long t0, t1;
    int slot = 0;
    lds[lane] = a;
    for (int i = lane; i < 4096; i += 64) lds[i] = 1e-9 * i;
    __syncthreads();
    // 0: dependent fma chain
    t0 = now();
#pragma unroll 1
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int i = 0; i < 64; i++) a = fma(a, b, c);
    }
    t1 = now();
    if (lane == 0) cyc[slot] = t1 - t0;
    slot++;
    // 1: 8 independent fma chains
    double x[8];
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = a + i;
    t0 = now();
#pragma unroll 1
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int u = 0; u < 8; u++) x[u] = fma(x[u], b, c);
    }
    t1 = now();
    if (lane == 0) cyc[slot] = t1 - t0;
    slot++;
#pragma unroll
    for (int i = 0; i < 8; i++) a += x[i];
    // 2: solve step chain: bcast + fma (dependent)
    t0 = now();
#pragma unroll 1
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int i = 0; i < 64; i++) {
            const double w = bcast(a, i);
            a = fma(-b, w, a);
        }
    }
    t1 = now();
    if (lane == 0) cyc[slot] = t1 - t0;
    slot++;
    // 3: fast_rcp chain
    t0 = now();
#pragma unroll 1
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int i = 0; i < 64; i++) a = fast_rcp(a) + 1.5;
    }
    t1 = now();
    if (lane == 0) cyc[slot] = t1 - t0;
    slot++;
    // 4: independent bcast (readlane throughput): 64 broadcasts of distinct lanes, summed
    t0 = now();
#pragma unroll 1
    for (int r = 0; r < reps; r++) {
        double s = 0;
#pragma unroll
        for (int i = 0; i < 64; i++) s += bcast(b, i);
        a += s * 1e-30;
    }
    t1 = now();
    if (lane == 0) cyc[slot] = t1 - t0;
    slot++;
    // 5: LDS read latency: pointer chase through LDS (dependent ds_read_b32)
    {
        int* li = reinterpret_cast<int*>(lds + 2048);
        for (int i = lane; i < 1024; i += 64) li[i] = (i * 17 + 5) & 1023;
        __syncthreads();
        int p = lane;
        t0 = now();
#pragma unroll 1
        for (int r = 0; r < reps; r++) {
#pragma unroll
            for (int i = 0; i < 64; i++) p = li[p];
        }
        t1 = now();
        if (lane == 0) cyc[slot] = t1 - t0;
        slot++;
        a += p;
    }
    // 6: LDS read throughput: 64 independent ds_read_b64 (conflict-free), summed
    t0 = now();
#pragma unroll 1
    for (int r = 0; r < reps; r++) {
        double s = 0;
#pragma unroll
        for (int i = 0; i < 64; i++) s += lds[lane + 64 * (i & 31)];
        a += s * 1e-30;
        asm volatile("" ::: "memory");
    }
    t1 = now();
    if (lane == 0) cyc[slot] = t1 - t0;
    slot++;
    // 7: LDS uniform (broadcast) reads throughput
    t0 = now();
#pragma unroll 1
    for (int r = 0; r < reps; r++) {
        double s = 0;
#pragma unroll
        for (int i = 0; i < 64; i++) s += lds[i * 3];
        a += s * 1e-30;
        asm volatile("" ::: "memory");
    }
    t1 = now();
    if (lane == 0) cyc[slot] = t1 - t0;
    slot++;
    // 8: ds_add_f64 (no return) throughput, conflict-free
    t0 = now();
#pragma unroll 1
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int i = 0; i < 64; i++) atomicAdd(&lds[lane + 64 * (i & 31)], c);
    }
    t1 = now();
    if (lane == 0) cyc[slot] = t1 - t0;
    slot++;
    // 9: ds_add_f64 with 2-way same-address conflicts (lanes l and l+32 hit the same word)
    t0 = now();
#pragma unroll 1
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int i = 0; i < 64; i++) atomicAdd(&lds[(lane & 31) + 64 * (i & 31)], c);
    }
    t1 = now();
    if (lane == 0) cyc[slot] = t1 - t0;
    slot++;
    // 10: LDS write -> sync -> uniform read round trip (cross-lane hand-off), dependent
    t0 = now();
#pragma unroll 1
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int i = 0; i < 64; i++) {
            lds[lane] = a;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            a = fma(lds[i], 1e-30, a);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    t1 = now();
    if (lane == 0) cyc[slot] = t1 - t0;
    slot++;
    // 11: global (scratch-like) load latency: dependent pointer chase in a small L1/L2-resident buffer
    {
        int* gi = reinterpret_cast<int*>(out + 64);
        int p = lane;
        t0 = now();
#pragma unroll 1
        for (int r = 0; r < reps; r++) {
#pragma unroll
            for (int i = 0; i < 64; i++) p = gi[p];
        }
        t1 = now();
        if (lane == 0) cyc[slot] = t1 - t0;
        slot++;
        a += p;
    }
    // 12: wave butterfly reduction (6 x ds_bpermute/dpp + add), dependent
    t0 = now();
#pragma unroll 1
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
            a *= 1e-3;
        }
    }
    t1 = now();
    if (lane == 0) cyc[slot] = t1 - t0;
    slot++;
    // 13: v_cndmask select pairs + fma (the masked solve step without the broadcast)
    t0 = now();
#pragma unroll 1
    for (int r = 0; r < reps; r++) {
        int lv = lane;
        asm volatile("" : "+v"(lv));
#pragma unroll
        for (int i = 0; i < 64; i++) a = fma(-((lv > i) ? b : 0.0), c, a);
    }
    t1 = now();
    if (lane == 0) cyc[slot] = t1 - t0;
    slot++;
    out[lane] = a;
}

int main() {
    double* out;
    unsigned long long* cyc;
    hipMalloc(&out, 1 << 20);
    hipMalloc(&cyc, 64 * 8);
    std::vector<int> chase(1024);
    for (int i = 0; i < 1024; i++) chase[i] = (i * 17 + 5) & 1023;
    hipMemcpy(reinterpret_cast<char*>(out) + 64 * 8, chase.data(), 4096, hipMemcpyHostToDevice);
    const int reps = 200;
    for (int w = 0; w < 2; w++) k<<<1, 64, 40000>>>(out, cyc, 1.0, reps);
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    k<<<1, 64, 40000>>>(out, cyc, 1.0, reps);
    hipDeviceSynchronize();
    auto t1 = std::chrono::steady_clock::now();
    unsigned long long h[16];
    hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    const char* names[] = {"dependent v_fma_f64 chain",        "8 independent v_fma_f64 chains (per fma)",
                           "bcast(2 readlane)+fma chain",      "fast_rcp + add chain (7 ops)",
                           "independent bcast + add",          "LDS dependent read (pointer chase)",
                           "LDS independent ds_read_b64",      "LDS uniform-address ds_read_b64",
                           "ds_add_f64 conflict-free",         "ds_add_f64 2 lanes per address",
                           "LDS write->wait->uniform read",    "global dependent load (L1/L2 hit)",
                           "wave butterfly sum (6 steps) x8/64", "masked fma (v_cmp + 2 cndmask + fma)"};
    unsigned long long tot = 0;
    for (int i = 0; i < 14; i++) {
        printf("%-42s %8.1f ticks/op\n", names[i], (double)h[i] / (64.0 * reps));
        tot += h[i];
    }
    const double us = std::chrono::duration<double, std::micro>(t1 - t0).count();
    printf("total ticks %llu in %.1f us wall (incl. launch) -> s_memtime tick = %.2f ns (%.0f MHz)\n", tot, us, us * 1e3 / tot,
           tot / us);
    return 0;
}
