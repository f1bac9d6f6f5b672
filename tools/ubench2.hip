// Development aid (continuation of ubench.hip): LDS read widths / broadcast forms, v_readlane, AGPR moves, fp64 MFMA.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define T0() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); t0 = __builtin_readcyclecounter(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define T1() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); t1 = __builtin_readcyclecounter(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); if (lane == 0) cyc[slot] = t1 - t0; slot++
typedef double double4_ __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void k(double* out, unsigned long long* cyc, double seed, int reps) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x;
    unsigned long long t0, t1;
    int slot = 0;
    for (int i = lane; i < 4096; i += 64) lds[i] = 1e-9 * i + seed;
    __syncthreads();
    double acc[8];
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = i;
    // 0: conflict-free ds_read_b64, 8 accumulators
    T0();
#pragma unroll 1
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int i = 0; i < 64; i++) acc[i & 7] += lds[lane + 64 * (i & 31)];
        asm volatile("" ::: "memory");
    }
    T1();
    // 1: uniform ds_read_b64 (odd stride so the compiler cannot merge), 8 accumulators
    T0();
#pragma unroll 1
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int i = 0; i < 64; i++) acc[i & 7] += lds[i * 5 + 1];
        asm volatile("" ::: "memory");
    }
    T1();
    // 2: uniform 16-byte reads (two doubles per instruction), per DOUBLE
    T0();
#pragma unroll 1
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int i = 0; i < 32; i++) {
            const double2 v = *reinterpret_cast<const double2*>(&lds[i * 6]);
            acc[i & 7] += v.x;
            acc[(i + 4) & 7] += v.y;
        }
        asm volatile("" ::: "memory");
    }
    T1();
    // 3: conflict-free 16-byte reads per lane, per DOUBLE
    T0();
#pragma unroll 1
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int i = 0; i < 32; i++) {
            const double2 v = *reinterpret_cast<const double2*>(&lds[2 * lane + 128 * (i & 15)]);
            acc[i & 7] += v.x;
            acc[(i + 4) & 7] += v.y;
        }
        asm volatile("" ::: "memory");
    }
    T1();
    // 4: only the 64 uniform reads issued, results consumed once at the end (issue cost of ds_read)
    T0();
#pragma unroll 1
    for (int r = 0; r < reps; r++) {
        double v[16];
#pragma unroll
        for (int q = 0; q < 4; q++) {
#pragma unroll
            for (int i = 0; i < 16; i++) v[i] = lds[(q * 16 + i) * 5 + 1];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("" ::"v"(v[i]));
        }
    }
    T1();
    // 5: v_readlane_b32 alone (64 independent), results consumed by s_add
    T0();
#pragma unroll 1
    for (int r = 0; r < reps; r++) {
        int s = 0;
#pragma unroll
        for (int i = 0; i < 64; i++) s += __builtin_amdgcn_readlane(lane * 3 + r, i);
        asm volatile("" ::"s"(s));
    }
    T1();
    // 6: v_readlane_b32 x2 + v_fma with SGPR operand, 8 accumulators (the factorisation inner step)
    {
        double src = seed + lane;
        T0();
#pragma unroll 1
        for (int r = 0; r < reps; r++) {
#pragma unroll
            for (int i = 0; i < 64; i++) {
                int lo = __builtin_amdgcn_readlane(__double2loint(src), i), hi = __builtin_amdgcn_readlane(__double2hiint(src), i);
                acc[i & 7] = fma(acc[i & 7], 0.999, __hiloint2double(hi, lo));
            }
        }
        T1();
    }
    // 7: ds_bpermute_b32 x2 + fma (arbitrary lane gather of a double), 8 accumulators
    {
        double src = seed + lane;
        T0();
#pragma unroll 1
        for (int r = 0; r < reps; r++) {
#pragma unroll
            for (int i = 0; i < 64; i++) {
                int lo = __builtin_amdgcn_ds_bpermute(i * 4, __double2loint(src)), hi = __builtin_amdgcn_ds_bpermute(i * 4, __double2hiint(src));
                acc[i & 7] = fma(acc[i & 7], 0.999, __hiloint2double(hi, lo));
            }
        }
        T1();
    }
    // 8: v_accvgpr_write + v_accvgpr_read round trip (independent)
    {
        T0();
#pragma unroll 1
        for (int r = 0; r < reps; r++) {
#pragma unroll
            for (int i = 0; i < 64; i++) {
                int v = __double2loint(acc[i & 7]);
                int a;
                asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(a) : "v"(v));
                asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a));
                acc[i & 7] = __hiloint2double(__double2hiint(acc[i & 7]), v);
            }
        }
        T1();
    }
    // 9: v_mfma_f64_16x16x4_f64 dependent chain (accumulate into the same tile)
    {
        double4_ c = {0, 0, 0, 0};
        double a = seed + lane, b = 1.0 / (1 + lane);
        T0();
#pragma unroll 1
        for (int r = 0; r < reps; r++) {
#pragma unroll
            for (int i = 0; i < 64; i++) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
        }
        T1();
        acc[0] += c[0] + c[1] + c[2] + c[3];
    }
    // 10: 4 independent MFMA f64 tiles
    {
        double4_ c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        double a = seed + lane, b = 1.0 / (1 + lane);
        T0();
#pragma unroll 1
        for (int r = 0; r < reps; r++) {
#pragma unroll
            for (int i = 0; i < 16; i++)
#pragma unroll
                for (int u = 0; u < 4; u++) c[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[u], 0, 0, 0);
        }
        T1();
        for (int u = 0; u < 4; u++) acc[u] += c[u][0] + c[u][1] + c[u][2] + c[u][3];
    }
    // 11: v_mfma_f64_4x4x4 (4 blocks) dependent chain
    {
        double c = 0;
        double a = seed + lane, b = 1.0 / (1 + lane);
        T0();
#pragma unroll 1
        for (int r = 0; r < reps; r++) {
#pragma unroll
            for (int i = 0; i < 64; i++) c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
        }
        T1();
        acc[0] += c;
    }
    // 12: v_rcp_f64 alone, dependent chain
    {
        double a = seed + lane + 2;
        T0();
#pragma unroll 1
        for (int r = 0; r < reps; r++) {
#pragma unroll
            for (int i = 0; i < 64; i++) a = __builtin_amdgcn_rcp(a);
        }
        T1();
        acc[1] += a;
    }
    // 13: 8 independent v_rcp_f64
    {
        T0();
#pragma unroll 1
        for (int r = 0; r < reps; r++) {
#pragma unroll
            for (int i = 0; i < 64; i++) acc[i & 7] = __builtin_amdgcn_rcp(acc[i & 7]);
        }
        T1();
    }
    // 14: DPP row broadcast-like move: v_mov_b32 with quad_perm (cheap cross-lane), 2 per double + fma
    {
        T0();
#pragma unroll 1
        for (int r = 0; r < reps; r++) {
#pragma unroll
            for (int i = 0; i < 64; i++) {
                int lo = __builtin_amdgcn_update_dpp(0, __double2loint(acc[i & 7]), 0x00 /*quad_perm 0,0,0,0*/, 0xf, 0xf, false);
                int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(acc[i & 7]), 0x00, 0xf, 0xf, false);
                acc[i & 7] = fma(acc[i & 7], 0.999, __hiloint2double(hi, lo));
            }
        }
        T1();
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += acc[i];
    out[lane] = s;
}
int main() {
    double* out;
    unsigned long long* cyc;
    (void)hipMalloc(&out, 1 << 20);
    (void)hipMalloc(&cyc, 64 * 8);
    const int reps = 200;
    for (int w = 0; w < 3; w++) k<<<1, 64, 40000>>>(out, cyc, 1.0, reps);
    (void)hipDeviceSynchronize();
    unsigned long long h[16];
    (void)hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    const char* names[] = {"ds_read_b64 conflict-free (8 acc)", "ds_read_b64 uniform address (8 acc)", "uniform 16-B read, per double", "conflict-free 16-B read, per double",
                           "uniform ds_read_b64 issue only", "v_readlane_b32 independent", "2 readlane + fma (8 acc)", "2 ds_bpermute + fma (8 acc)", "accvgpr write+read pair",
                           "mfma_f64_16x16x4 dependent", "mfma_f64_16x16x4 4 independent", "mfma_f64_4x4x4 dependent", "v_rcp_f64 dependent", "v_rcp_f64 8 independent", "2 dpp mov + fma (8 acc)"};
    for (int i = 0; i < 15; i++) printf("%-42s %8.1f cycles/op\n", names[i], (double)h[i] / (64.0 * reps));
    return 0;
}
