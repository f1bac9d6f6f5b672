// Development aid (round 4): the register LDL^T of the PDIP kernel on its own, and the alternatives VERDICT r03 item 5 asks to MEASURE.
//
// One wavefront factorises a dense SPD matrix of NZ rows, lane i holding row i in registers -- the layout of lscqp_kernel.hpp.
// Variants, each timed with s_memtime over `reps` factorisations of the same matrix (reloaded from LDS every time, outside the clock):
//   0  PRODUCT    the kernel's hybrid pivot-row broadcast (first entries by v_readlane, the rest read back from the published pivot
//                 column in LDS), copied from lscqp_kernel.hpp
//   1  CHAIN      the same with every trailing update that is not on the dependency chain removed: per pivot only the multiplier, the
//                 update of the next pivot column, its publication, the pivot broadcast and the reciprocal remain.  This is the FLOOR of
//                 any scheme that still runs NZ sequential pivots in this layout -- i.e. what a blocked factorisation whose trailing
//                 updates cost NOTHING (MFMA or otherwise) would be left with.
//   2  PANEL2     panels of two pivots: both raw pivot columns are published once, every lane forms the second pivot row itself
//                 (u1 = raw1 - l10 u0, the arithmetic lane j+1 would do), one LDS round trip and one exposed reciprocal chain per TWO
//                 pivots, 3 FMAs per trailing entry instead of 2.  Verified against variant 0.
//   3  CHAIN+MFMA variant 1 plus, per panel of four pivots, the v_mfma_f64_16x16x4_f64 instructions a 16x16-tiled trailing update of
//                 a 48 x 48 matrix would issue (6 / 3 / 1 tiles by block column), each fed by and feeding the chain (no layout
//                 conversions, no panel solves): an OPTIMISTIC bound for the MFMA-blocked design.
// Output: cycles per factorisation for every variant and the max deviation of variant 2's factor from variant 0's.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#ifndef NZ
#define NZ 39
#endif
#ifndef HYB
#define HYB 5
#endif

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}
__device__ __forceinline__ double bcast(double v, int lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double fast_rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;
}
typedef double double4_ __attribute__((ext_vector_type(4)));

#define CLK(t)                                                         \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");        \
    t = __builtin_readcyclecounter();                                  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

template <int VAR>
__global__ __launch_bounds__(64) void ldlt_kernel(const double* __restrict__ Kin, double* __restrict__ Lout, unsigned long long* __restrict__ cyc, int reps) {
    __shared__ double Ks[NZ][NZ | 1];
    __shared__ double col_[2 * 64 + 8];
    __shared__ double col2_[4 * 64 + 8];
    const int lane = threadIdx.x;
    for (int e = lane; e < NZ * NZ; e += 64) Ks[e / NZ][e % NZ] = Kin[e];
    __syncthreads();
    unsigned long long total = 0;
    double A[NZ];
    double dinv_own = 0;
    double4_ acc[6];
#pragma unroll
    for (int t = 0; t < 6; t++) acc[t] = double4_{0, 0, 0, 0};
#pragma unroll 1
    for (int r = 0; r < reps; r++) {
        int lf = lane;
        asm volatile("" : "+v"(lf));
        const bool zl = lf < NZ;
#pragma unroll
        for (int c = 0; c < NZ; c++) {
            const double v = Ks[zl ? lf : 0][c];
            A[c] = zl ? v : 0.0;
        }
        unsigned long long t0, t1;
        CLK(t0);
        if constexpr (VAR == 0 || VAR == 1 || VAR == 3) {
            double* const colw = col_;
            colw[lf] = A[0];
            double d = bcast(A[0], 0);
            double invd = fast_rcp(d);
            static_for<0, NZ>([&](auto Jc) {
                constexpr int j = decltype(Jc)::value;
                constexpr int n = NZ - j - 1;
                constexpr int r0 = (n * HYB + 50) / 100;
                constexpr int nr = n <= 4 ? n : (r0 < 3 ? 4 : r0 + 1);
                constexpr int nl = n - (nr < n ? nr : n);
                constexpr int NR = n - nl;
                const double* const cb = colw + (j & 1) * 64;
                double* const cbn = colw + ((j + 1) & 1) * 64;
                dinv_own = (lf == j) ? invd : dinv_own;
                const double li = (lf > j) ? A[j] * invd : 0.0;
                if constexpr (VAR == 0) {
                    double ul[nl > 0 ? nl : 1];
                    static_for<0, nl>([&](auto Tc) {
                        constexpr int t = decltype(Tc)::value;
                        ul[t] = cb[j + 1 + NR + t];
                    });
                    if constexpr (NR > 0) {
                        double ur[NR];
                        static_for<0, NR>([&](auto Tc) {
                            constexpr int t = decltype(Tc)::value;
                            ur[t] = bcast(A[j + 1 + t], j);
                        });
                        A[j + 1] = fma(-li, ur[0], A[j + 1]);
                        cbn[lf] = A[j + 1];
                        d = bcast(A[j + 1], j + 1);
                        invd = fast_rcp(d);
                        static_for<1, NR>([&](auto Tc) {
                            constexpr int t = decltype(Tc)::value;
                            A[j + 1 + t] = fma(-li, ur[t], A[j + 1 + t]);
                        });
                    }
                    static_for<0, nl>([&](auto Tc) {
                        constexpr int t = decltype(Tc)::value;
                        A[j + 1 + NR + t] = fma(-li, ul[t], A[j + 1 + NR + t]);
                    });
                } else {  // the dependency chain alone
                    if constexpr (NR > 0) {
                        const double u0 = bcast(A[j + 1], j);
                        A[j + 1] = fma(-li, u0, A[j + 1]);
                        cbn[lf] = A[j + 1];
                        d = bcast(A[j + 1], j + 1);
                        d = fmax(fabs(d), 1.0);  // (the matrix is no longer being factorised: keep the pivots finite)
                        invd = fast_rcp(d);
                    }
                    if constexpr (VAR == 3 && (j % 4) == 3) {
                        // the MFMAs of one four-pivot panel of a 48 x 48 tiled trailing update: operands derived from the chain's
                        // latest multiplier, results folded back into the next pivot column (so they sit ON the chain, as they would)
                        constexpr int bc = j / 16;
                        constexpr int nt = bc == 0 ? 6 : (bc == 1 ? 3 : 1);
                        static_for<0, nt>([&](auto Tc) {
                            constexpr int t = decltype(Tc)::value;
                            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(li, invd, acc[t], 0, 0, 0);
                        });
                        if constexpr (j + 1 < NZ) A[j + 1] += 1e-300 * acc[0].x;
                    }
                }
                A[j] = (lf > j) ? li : A[j];
                asm volatile("" ::: "memory");
            });
        } else {  // VAR == 2: panels of two pivots
            // columns j, j+1 of every lane are final (w.r.t. all earlier pivots) when panel j starts; both are published RAW
            double* const cw = col2_;
            cw[lf] = A[0];
            cw[64 + lf] = A[1];
            static_for<0, (NZ + 1) / 2>([&](auto Pc) {
                constexpr int j = 2 * decltype(Pc)::value;
                constexpr bool two = j + 1 < NZ;
                constexpr int n = NZ - j - (two ? 2 : 1);  // trailing entries beyond the panel
                const double* const c0 = cw + ((j / 2) & 1) * 128;
                const double* const c1 = c0 + 64;
                double* const n0 = cw + (((j / 2) + 1) & 1) * 128;
                double* const n1 = n0 + 64;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                // the 2 x 2 pivot block, redundantly in every lane: a = K[j][j], b = K[j+1][j], c = K[j+1][j+1]
                const double a = c0[j];
                const double invd0 = fast_rcp(a);
                double l10 = 0, invd1 = 0, bb = 0;
                if constexpr (two) {
                    bb = c0[j + 1];
                    const double cc = c1[j + 1];
                    l10 = bb * invd0;
                    invd1 = fast_rcp(fma(-l10, bb, cc));
                }
                dinv_own = (lf == j) ? invd0 : dinv_own;
                if constexpr (two) dinv_own = (lf == j + 1) ? invd1 : dinv_own;
                const double li0 = (lf > j) ? A[j] * invd0 : 0.0;
                double li1 = 0.0;
                if constexpr (two) {
                    const double tt = fma(-li0, bb, A[j + 1]);  // the lane's entry of column j+1 after pivot j
                    li1 = (lf > j + 1) ? tt * invd1 : 0.0;
                    A[j + 1] = (lf > j + 1) ? li1 : ((lf == j + 1) ? tt : A[j + 1]);  // lane j+1 keeps its pivot, lanes <= j their upper row
                }
                // the next panel's two columns first (their publication overlaps the rest of the update)
                static_for<0, n>([&](auto Tc) {
                    constexpr int t = decltype(Tc)::value;
                    constexpr int k = j + (two ? 2 : 1) + t;
                    const double u0 = c0[k];
                    double v = fma(-li0, u0, A[k]);
                    if constexpr (two) {
                        const double u1 = fma(-l10, u0, c1[k]);
                        v = fma(-li1, u1, v);
                    }
                    A[k] = v;
                    if constexpr (t == 0) n0[lf] = v;
                    if constexpr (t == 1) n1[lf] = v;
                });
                A[j] = (lf > j) ? li0 : A[j];
                asm volatile("" ::: "memory");
            });
        }
        CLK(t1);
        total += t1 - t0;
    }
    if (lane == 0) cyc[blockIdx.x] = total;
    double keep = 0;
#pragma unroll
    for (int t = 0; t < 6; t++) keep += acc[t].x;
    if (blockIdx.x == 0 && lane < NZ) {
#pragma unroll
        for (int c = 0; c < NZ; c++) Lout[lane * NZ + c] = A[c];
        Lout[NZ * NZ + lane] = dinv_own + 0.0 * keep;
    }
}

template <int VAR>
static double run(const double* dK, double* dL, unsigned long long* dC, int blocks, int reps, std::vector<double>* fac) {
    hipLaunchKernelGGL(ldlt_kernel<VAR>, dim3(blocks), dim3(64), 0, 0, dK, dL, dC, 2);
    hipLaunchKernelGGL(ldlt_kernel<VAR>, dim3(blocks), dim3(64), 0, 0, dK, dL, dC, reps);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> c(blocks);
    (void)hipMemcpy(c.data(), dC, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : c) s += (double)v;
    if (fac) {
        fac->resize(NZ * NZ + NZ);
        (void)hipMemcpy(fac->data(), dL, sizeof(double) * (NZ * NZ + NZ), hipMemcpyDeviceToHost);
    }
    return s / blocks / reps;
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 64, reps = argc > 2 ? atoi(argv[2]) : 200;
    // SPD test matrix with the conditioning of a late interior-point iteration: B'B + diag(1 .. 1e6)
    std::vector<double> K(NZ * NZ, 0.0), B(NZ * NZ);
    srand(7);
    for (auto& v : B) v = (rand() / (double)RAND_MAX) - 0.5;
    for (int i = 0; i < NZ; i++)
        for (int j = 0; j < NZ; j++) {
            double s = 0;
            for (int k = 0; k < NZ; k++) s += B[k * NZ + i] * B[k * NZ + j];
            K[i * NZ + j] = s + (i == j ? pow(10.0, 6.0 * i / (NZ - 1)) : 0.0);
        }
    double *dK, *dL;
    unsigned long long* dC;
    (void)hipMalloc(&dK, sizeof(double) * NZ * NZ);
    (void)hipMalloc(&dL, sizeof(double) * (NZ * NZ + NZ));
    (void)hipMalloc(&dC, sizeof(unsigned long long) * blocks);
    (void)hipMemcpy(dK, K.data(), sizeof(double) * NZ * NZ, hipMemcpyHostToDevice);
    std::vector<double> f0, f2;
    const double c0 = run<0>(dK, dL, dC, blocks, reps, &f0);
    const double c1 = run<1>(dK, dL, dC, blocks, reps, nullptr);
    const double c2 = run<2>(dK, dL, dC, blocks, reps, &f2);
    const double c3 = run<3>(dK, dL, dC, blocks, reps, nullptr);
    // variant 2 against variant 0: strictly lower factor L (entries k < i of row i) and 1/d
    double dev = 0, ref = 0;
    for (int i = 0; i < NZ; i++) {
        for (int k = 0; k < i; k++) {
            dev = fmax(dev, fabs(f0[i * NZ + k] - f2[i * NZ + k]));
            ref = fmax(ref, fabs(f0[i * NZ + k]));
        }
        dev = fmax(dev, fabs(f0[NZ * NZ + i] - f2[NZ * NZ + i]) / fabs(f0[NZ * NZ + i]));
    }
    printf("{\"nz\": %d, \"workgroups\": %d, \"reps\": %d, \"cycles_per_factorisation\": {\"product_hybrid\": %.0f, \"dependency_chain_only\": %.0f, "
           "\"panel_of_two\": %.0f, \"chain_plus_mfma_trailing_bound\": %.0f}, \"per_pivot\": {\"product_hybrid\": %.1f, \"dependency_chain_only\": %.1f, "
           "\"panel_of_two\": %.1f, \"chain_plus_mfma_trailing_bound\": %.1f}, \"panel_of_two_vs_product_max_dev\": %.3e, \"max_abs_L\": %.3e}\n",
           NZ, blocks, reps, c0, c1, c2, c3, c0 / NZ, c1 / NZ, c2 / NZ, c3 / NZ, dev, ref);
    return 0;
}
