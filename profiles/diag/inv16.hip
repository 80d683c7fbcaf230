// Diagnostic microbenchmark: in-wave inversion of a 16x16 SPD tile held in the MFMA accumulator layout
// (lane l: col = l&15, rows (l>>4)+4r).  Variant A: pivot column through LDS; variant B: ds_bpermute/readlane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double rcp_newton(double d) {
    double x = __builtin_amdgcn_rcp(d);
    double e = fma(-d, x, 1.0); x = fma(x, e, x); e = fma(-d, x, 1.0); x = fma(x, e, x);
    return x;
}
__device__ __forceinline__ d4 invA(d4 t, double* cb, int lc, int lq) {
    for (int k = 0; k < 16; ++k) {
        if (lc == k) { for (int r = 0; r < 4; ++r) cb[lq + 4 * r] = t[r]; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double cr[4];
        for (int r = 0; r < 4; ++r) cr[r] = cb[lq + 4 * r];
        const double cc = cb[lc], d = cb[k];
        const double pinv = rcp_newton(d), tc = cc * pinv;
        for (int r = 0; r < 4; ++r) {
            const int row = lq + 4 * r;
            double v = fma(-cr[r], tc, t[r]);
            if (row == k) v = (lc == k) ? -pinv : tc; else if (lc == k) v = cr[r] * pinv;
            t[r] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
    }
    return t;
}
template <int K>
__device__ __forceinline__ void stepB(d4& t, int lane, int lc, int lq) {
    double cr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) cr[r] = __shfl(t[r], (lane & 48) | K);          // a[lq+4r][K]
    const double cc = __shfl(t[K >> 2], ((K & 3) << 4) | lc);                  // a[K][lc]
    const double d = __shfl(t[K >> 2], ((K & 3) << 4) | K);                    // a[K][K]
    const double pinv = rcp_newton(d), tc = cc * pinv;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = lq + 4 * r;
        double v = fma(-cr[r], tc, t[r]);
        if (row == K) v = (lc == K) ? -pinv : tc; else if (lc == K) v = cr[r] * pinv;
        t[r] = v;
    }
}
__device__ __forceinline__ d4 invB(d4 t, int lane, int lc, int lq) {
    stepB<0>(t, lane, lc, lq); stepB<1>(t, lane, lc, lq); stepB<2>(t, lane, lc, lq); stepB<3>(t, lane, lc, lq);
    stepB<4>(t, lane, lc, lq); stepB<5>(t, lane, lc, lq); stepB<6>(t, lane, lc, lq); stepB<7>(t, lane, lc, lq);
    stepB<8>(t, lane, lc, lq); stepB<9>(t, lane, lc, lq); stepB<10>(t, lane, lc, lq); stepB<11>(t, lane, lc, lq);
    stepB<12>(t, lane, lc, lq); stepB<13>(t, lane, lc, lq); stepB<14>(t, lane, lc, lq); stepB<15>(t, lane, lc, lq);
    return t;
}

template <int CTRL>
__device__ __forceinline__ double dpp64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rcp_newton3(double d) {
    double x = __builtin_amdgcn_rcp(d);
    double e = fma(-d, x, 1.0);
    double e2 = fma(e, e, e);
    return fma(x, e2, x);
}
template <int K, int NEWTON>
__device__ __forceinline__ void stepC(d4& t, int lane, int lc, int lq) {
    const double cc = __shfl(t[K >> 2], ((K & 3) << 4) | lc);                  // a[K][lc]
    double cr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) cr[r] = dpp64<0x150 + K>(t[r]);                // a[lq+4r][K]  (row_newbcast:K)
    const double d = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(t[K >> 2]), ((K & 3) << 4) | K),
                                      __builtin_amdgcn_readlane(__double2loint(t[K >> 2]), ((K & 3) << 4) | K));
    const double pinv = NEWTON == 3 ? rcp_newton3(d) : rcp_newton(d);
    const double tc = cc * pinv;
    const bool pc = lc == K;
    const double m = pc ? -pinv : tc;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = lq + 4 * r;
        const double base = pc ? 0.0 : t[r];
        double v = fma(-cr[r], m, base);
        t[r] = (row == K) ? m : v;
    }
}
template <int NEWTON>
__device__ __forceinline__ d4 invC(d4 t, int lane, int lc, int lq) {
#define S(K) stepC<K, NEWTON>(t, lane, lc, lq);
    S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)
#undef S
    return t;
}


template <int CTRL>
__device__ __forceinline__ double dpp64b(double v) {
    return __builtin_amdgcn_update_dpp(0.0, v, CTRL, 0xf, 0xf, true);
}
template <int K>
__device__ __forceinline__ void stepE(d4& t, int lane, int lc, int lq) {
    const double cc = __shfl(t[K >> 2], ((K & 3) << 4) | lc);                  // a[K][lc]
    double cr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) cr[r] = dpp64b<0x150 + K>(t[r]);               // a[lq+4r][K]  (row_newbcast:K)
    const double d = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(t[K >> 2]), ((K & 3) << 4) | K),
                                      __builtin_amdgcn_readlane(__double2loint(t[K >> 2]), ((K & 3) << 4) | K));
    const double pinv = rcp_newton3(d);
    const double tc = cc * pinv;
    const bool pc = lc == K;
    const double m = pc ? -pinv : tc;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = lq + 4 * r;
        const double base = pc ? 0.0 : t[r];
        double v = fma(-cr[r], m, base);
        t[r] = (row == K) ? m : v;
    }
}
__device__ __forceinline__ d4 invE(d4 t, int lane, int lc, int lq) {
#define S(K) stepE<K>(t, lane, lc, lq);
    S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)
#undef S
    return t;
}

// deferred column scaling: from its own pivot on, column K is kept divided by pinv_K (its lanes skip the update of
// pivot K, the diagonal entry becomes -1); every later sweep is linear in that factor, which is applied at the end.
template <int K>
__device__ __forceinline__ void stepF(d4& t, double& mypinv, int lane, int lc, int lq) {
    const double cc = __shfl(t[K >> 2], ((K & 3) << 4) | lc);                  // a[K][lc]
    double cr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) cr[r] = dpp64b<0x150 + K>(t[r]);               // a[lq+4r][K]  (row_newbcast:K)
    const double d = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(t[K >> 2]), ((K & 3) << 4) | K),
                                      __builtin_amdgcn_readlane(__double2loint(t[K >> 2]), ((K & 3) << 4) | K));
    const double pinv = rcp_newton3(d);
    const double tc = cc * pinv;
    const bool pc = lc == K;
    if (!pc) {
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] = fma(-cr[r], tc, t[r]);
    }
    if (lq == (K & 3)) t[K >> 2] = pc ? -1.0 : tc;
    mypinv = pc ? pinv : mypinv;
}
__device__ __forceinline__ d4 invF(d4 t, int lane, int lc, int lq) {
    double mypinv = 0.0;
#define S(K) stepF<K>(t, mypinv, lane, lc, lq);
    S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)
#undef S
#pragma unroll
    for (int r = 0; r < 4; ++r) t[r] *= mypinv;
    return t;
}

// as F, but the pivot row is spread over the 4 lane-rows by one MFMA (A = one-hot k-slot K&3, B = the register
// holding rows 4(K>>2)..+3): D[i][j] = a[K][j] for every i, exact (1.0 * x + zeros).
template <int K>
__device__ __forceinline__ void stepG(d4& t, double& mypinv, int lane, int lc, int lq) {
    const double onehot = (lq == (K & 3)) ? 1.0 : 0.0;
    const d4 z = {0.0, 0.0, 0.0, 0.0};
    const d4 bc = __builtin_amdgcn_mfma_f64_16x16x4f64(onehot, t[K >> 2], z, 0, 0, 0);
    const double cc = bc[0];
    double cr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) cr[r] = dpp64b<0x150 + K>(t[r]);               // a[lq+4r][K]  (row_newbcast:K)
    const double d = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(t[K >> 2]), ((K & 3) << 4) | K),
                                      __builtin_amdgcn_readlane(__double2loint(t[K >> 2]), ((K & 3) << 4) | K));
    const double pinv = rcp_newton3(d);
    const double tc = cc * pinv;
    const bool pc = lc == K;
    if (!pc) {
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] = fma(-cr[r], tc, t[r]);
    }
    if (lq == (K & 3)) t[K >> 2] = pc ? -1.0 : tc;
    mypinv = pc ? pinv : mypinv;
    asm volatile("" ::"v"(bc));  // keep the 8 result registers reserved: reusing them early costs MFMA-hazard nops
}
__device__ __forceinline__ d4 invG(d4 t, int lane, int lc, int lq) {
    double mypinv = 0.0;
#define S(K) stepG<K>(t, mypinv, lane, lc, lq);
    S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)
#undef S
#pragma unroll
    for (int r = 0; r < 4; ++r) t[r] *= mypinv;
    return t;
}
// H: as F with the NEXT pivot formed ahead of the tile update.  The pivot of step K + 1 is the diagonal entry (K+1, K+1) after step K,
// a[K+1][K+1] - a[K+1][K] (a[K][K+1] pinv_K): its three inputs are read from the tile BEFORE step K's update (v_readlane), so the chain
// pinv_K -> next pivot -> pinv_{K+1} no longer waits for the tile (and its row broadcast); bit-identical to F.  PERM: the pivot row to the
// four lane-rows by v_permlane16_swap + v_permlane32_swap instead of ds_bpermute.
__device__ __forceinline__ double rl64(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
template <int S>
__device__ __forceinline__ unsigned row_bcast32(unsigned v) {
    const auto p = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    const unsigned x = (S & 1) ? p[1] : p[0];
    const auto q = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    return (S & 2) ? q[1] : q[0];
}
template <int S>
__device__ __forceinline__ double row_bcast64(double v) {
    return __hiloint2double((int)row_bcast32<S>((unsigned)__double2hiint(v)), (int)row_bcast32<S>((unsigned)__double2loint(v)));
}
template <int K, bool PERM>
__device__ __forceinline__ void stepH(d4& t, double& mypinv, double& dcur, int lane, int lc, int lq) {
    double a_r = 0.0, a_c = 0.0, d_n = 1.0;
    if (K < 15) {
        a_r = rl64(t[(K + 1) >> 2], (((K + 1) & 3) << 4) | K);
        a_c = rl64(t[K >> 2], ((K & 3) << 4) | ((K + 1) & 15));
        d_n = rl64(t[(K + 1) >> 2], (((K + 1) & 3) << 4) | ((K + 1) & 15));
    }
    const double cc = PERM ? row_bcast64<(K & 3)>(t[K >> 2]) : __shfl(t[K >> 2], ((K & 3) << 4) | lc);
    double cr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) cr[r] = dpp64b<0x150 + K>(t[r]);
    const double pinv = rcp_newton3(dcur);
    const double tc = cc * pinv;
    const bool pc = lc == K;
    const double tce = pc ? 0.0 : tc;
#pragma unroll
    for (int r = 0; r < 4; ++r) t[r] = fma(-cr[r], tce, t[r]);
    if (lq == (K & 3)) t[K >> 2] = pc ? -1.0 : tc;
    mypinv = pc ? pinv : mypinv;
    dcur = fma(-a_r, a_c * pinv, d_n);
}
template <bool PERM>
__device__ __forceinline__ d4 invH(d4 t, int lane, int lc, int lq) {
    double mypinv = 0.0, dcur = rl64(t[0], 0);
#define S(K) stepH<K, PERM>(t, mypinv, dcur, lane, lc, lq);
    S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)
#undef S
#pragma unroll
    for (int r = 0; r < 4; ++r) t[r] *= mypinv;
    return t;
}

template <int V>
__global__ void k(const double* A, double* out, long long* cyc) {
    __shared__ double cb[16];
    const int lane = threadIdx.x, lc = lane & 15, lq = lane >> 4;
    d4 t;
    for (int r = 0; r < 4; ++r) t[r] = A[(lq + 4 * r) * 16 + lc];
    __syncthreads();
    long long t0 = clock64();
    for (int rep = 0; rep < 9; ++rep) {   // sweep(sweep(A)) = A: odd count leaves -inv(A)
        if (V == 0) t = invA(t, cb, lc, lq);
        if (V == 1) t = invB(t, lane, lc, lq);
        if (V == 2) t = invC<2>(t, lane, lc, lq);
        if (V == 3) t = invC<3>(t, lane, lc, lq);
        if (V == 4) t = invE(t, lane, lc, lq);
        if (V == 5) t = invF(t, lane, lc, lq);
        if (V == 6) t = invG(t, lane, lc, lq);
        if (V == 7) t = invH<false>(t, lane, lc, lq);
        if (V == 8) t = invH<true>(t, lane, lc, lq);
    }
    __syncthreads();
    long long t1 = clock64();
    for (int r = 0; r < 4; ++r) out[(lq + 4 * r) * 16 + lc] = -t[r];
    if (lane == 0) cyc[0] = t1 - t0;
}
template <int V>
void run(const char* name, const std::vector<double>& A, double* dA, double* o, long long* dc) {
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k<V>, dim3(1), dim3(64), 0, 0, dA, o, dc); (void)hipDeviceSynchronize(); }
    std::vector<double> a(256); long long c;
    (void)hipMemcpy(a.data(), o, 2048, hipMemcpyDeviceToHost); (void)hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    double ea = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double sa = 0; for (int q = 0; q < 16; ++q) sa += A[i * 16 + q] * a[q * 16 + j]; ea = fmax(ea, fabs(sa - (i == j))); }
    printf("%-28s %6.0f ticks per 16x16 inverse (%.0f per pivot), |A inv - I| = %.2e\n", name, c / 9.0, c / 144.0, ea);
}

// contention test: wave 0 runs the sweep chain; if load != 0 waves 4 (same SIMD as wave 0 under round-robin
// placement) or 1 (another SIMD) issue back-to-back fp64 MFMAs meanwhile.
__global__ void k_contend(const double* A, double* out, long long* cyc, int load_wave, int prio) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, lc = lane & 15, lq = lane >> 4;
    if (w == 0) {
        if (prio) __builtin_amdgcn_s_setprio(3);
        d4 t;
        for (int r = 0; r < 4; ++r) t[r] = A[(lq + 4 * r) * 16 + lc];
        long long t0 = clock64();
        for (int rep = 0; rep < 9; ++rep) t = invF(t, lane, lc, lq);
        long long t1 = clock64();
        for (int r = 0; r < 4; ++r) out[(lq + 4 * r) * 16 + lc] = -t[r];
        if (lane == 0) cyc[0] = t1 - t0;
    } else if (w == load_wave) {
        d4 acc = {0.0, 0.0, 0.0, 0.0};
        const double a = A[lane], b = A[64 + lane];
        for (int i = 0; i < 1200; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        if (acc[0] == 123.456) out[0] = acc[1];
    } else if (w == load_wave + 8) {   // second independent MFMA stream on the same SIMD
        d4 acc = {0.0, 0.0, 0.0, 0.0};
        const double a = A[lane], b = A[64 + lane];
        for (int i = 0; i < 1200; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        if (acc[0] == 123.456) out[0] = acc[1];
    }
}
int main() {
    std::vector<double> A(256);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) A[i * 16 + j] = (i == j ? 20.0 + i : 0.0) - 1.0 / (1 + abs(i - j));
    double *dA, *o; long long* dc;
    (void)hipMalloc(&dA, 2048); (void)hipMalloc(&o, 2048); (void)hipMalloc(&dc, 32);
    (void)hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice);
    run<0>("LDS column", A, dA, o, dc);
    run<1>("bpermute", A, dA, o, dc);
    run<2>("dpp bcast, 2 newton", A, dA, o, dc);
    run<3>("dpp bcast, cubic newton", A, dA, o, dc);
    run<4>("dpp64", A, dA, o, dc);
    run<5>("dpp64, deferred col scale", A, dA, o, dc);
    run<6>("same + MFMA row broadcast", A, dA, o, dc);
    run<7>("F + pivot look-ahead", A, dA, o, dc);
    run<8>("F + look-ahead + permlane", A, dA, o, dc);
    for (int cfg = 0; cfg < 6; ++cfg) {
        const int lw[6] = {99, 4, 4, 1, 1, 4}, pr[6] = {0, 0, 1, 0, 1, 1};
        const int nthreads = cfg == 5 ? 64 * 13 : 64 * 5;
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k_contend, dim3(1), dim3(nthreads), 0, 0, dA, o, dc, lw[cfg], pr[cfg]); (void)hipDeviceSynchronize(); }
        long long c; (void)hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
        printf("sweep with MFMA load on wave %2d%s, prio %d: %6.0f cycles per 16x16 inverse\n", lw[cfg], cfg == 5 ? "+12" : "", pr[cfg], c / 9.0);
    }
    return 0;
}
