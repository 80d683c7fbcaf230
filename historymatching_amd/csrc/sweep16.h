// sweep16.h -- in-wave inversion of one 16x16 SPD tile held in the fp64 MFMA accumulator layout (shared by the
// Ny = 128 pressure kernels press128m.hip / press128s.hip).
#pragma once
#include <hip/hip_runtime.h>

typedef double d4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------------------
// In-wave symmetric sweep of one 16x16 tile held in the accumulator layout (lane (lq, lc), reg r <-> entry
// (lq + 4r, lc)): 16 pivots, on return t = -inv(tile).  One wave is issue bound (~6 cycles per instruction), so the
// step is written for instruction count (profiles/diag/inv16.hip: 156 cycles per pivot against 296 with the pivot column
// staged through LDS):
//   * pivot column to the lanes of each row:   v_mov_b64_dpp row_newbcast:K                     (4 instructions)
//   * pivot row to the 4 lane-rows:            ds_bpermute (the only cross-row move; symmetric tile).  Tried instead: one
//     v_mfma_f64_16x16x4 with a row selector as A and the tile's register as B (accumulator layout is the B layout) -- no LDS
//     round trip, but the wave's next VALU instruction waits out the matrix instruction: +1.2 k cycles per tile (spdinv.hip)
//   * pivot value:                             v_readlane -> SGPR, reciprocal by rcp + cubic Newton step
//   * deferred column scaling: the textbook sweep multiplies column K by 1/d_K at pivot K, which costs a select per
//     entry per pivot.  Every later sweep is linear in that factor, so the lanes of column K simply skip pivot K
//     (the diagonal entry becomes -1) and the factor is applied once after the last pivot.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double rcp_newton3(double d) {
    const double x = __builtin_amdgcn_rcp(d);
    const double e = fma(-d, x, 1.0);
    const double e2 = fma(e, e, e);  // x (1 + e + e^2): third-order step, error e^3
    return fma(x, e2, x);
}

template <int K, typename GEO>
__device__ __forceinline__ void sweep16_step(d4& t, double& mypinv, const GEO& g, int& bad) {
    const double cc = __shfl(t[K >> 2], ((K & 3) << 4) | g.lc);  // a[K][lc]
    double cr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) cr[r] = __builtin_amdgcn_update_dpp(0.0, t[r], 0x150 + K, 0xf, 0xf, true);  // a[lq+4r][K]
    const int src = ((K & 3) << 4) | K;
    const double d = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(t[K >> 2]), src),
                                      __builtin_amdgcn_readlane(__double2loint(t[K >> 2]), src));
    if (!(d > 0.0)) bad = 1;
    const double pinv = rcp_newton3(d);
    const double tc = cc * pinv;
    const bool pc = g.lc == K;
    // the lanes of column K skip pivot K: they take part with a ZERO multiplier (fma(-cr, 0, t) == t) instead of keeping t under a
    // select -- 2 selects per step instead of 8 (round 3: the compiler turned `if (!pc)` into 8 v_cndmask per step)
    const double tce = pc ? 0.0 : tc;
#pragma unroll
    for (int r = 0; r < 4; ++r) t[r] = fma(-cr[r], tce, t[r]);
    if (g.lq == (K & 3)) t[K >> 2] = pc ? -1.0 : tc;
    mypinv = pc ? pinv : mypinv;
}

template <typename GEO>
__device__ __forceinline__ void sweep16_inwave(d4& t, const GEO& g, int& bad) {
    double mypinv = 0.0;
#define S(K) sweep16_step<K, GEO>(t, mypinv, g, bad);
    S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)
#undef S
#pragma unroll
    for (int r = 0; r < 4; ++r) t[r] *= mypinv;
}

