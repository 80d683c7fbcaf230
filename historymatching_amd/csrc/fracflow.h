// fracflow.h -- fractional flow fw(S) of the explicit saturation sweeps (fp64).
//   fw = mw / (mw + mo),   mw = S^2 / vw,  mo = (1 - S)^2 / vo,   S normalised by (swc, sor)      (oracle/ressim.py: RelPerm, frac_flow)
// Every operation is rounded separately, as NumPy does (the sources are compiled with -ffp-contract=off), and the quotient must be
// the correctly rounded IEEE one.  The compiler's IEEE division is 11 instructions: two v_div_scale (operand pre-scaling for
// quotients near the ends of the exponent range), v_rcp + two Newton steps, the quotient + one residual correction (v_div_fmas,
// which undoes the scaling) and v_div_fixup (NaN / infinity / zero operands): ~60 cycles of a SIMD per wave, a third of the sweep
// (profiles/diag/valu_rate.hip).  For the upstream fluid (vw = vo = 1, swc = sor = 0: n = S^2, d = S^2 + (1-S)^2) the three scaling / fix-up
// instructions never do anything, so `div_unscaled` runs the same sequence without them -- 8 instructions, same bits (7 since round 5, below):
//   * 2^-480 <= |S| < 2^500: d >= 0.5 is normal, n >= 2^-960 is far above v_div_scale's thresholds (numerator exponent <= 53,
//     denormal quotient), the quotient is normal: both v_div_scale return their operand, VCC = 0 makes v_div_fmas a v_fma,
//     v_div_fixup returns its first operand -- instruction for instruction the same values;
//   * |S| < 2^-480 (this is where the compiler's sequence does scale -- or, for n = 0, produces NaN and lets v_div_fixup return 0):
//     1 - S and S^2 + 1 round to exactly 1.0, so d = 1.0; the two Newton steps take any seed within 2^-14 of 1 to exactly 1.0
//     (1 - eps^4 rounds to 1), the quotient n * 1.0 is n, the residual fma(-1, n, n) is 0: the result is n, which is n / 1.0 --
//     exact also when n is denormal or zero.
// Ahead of the front the saturation decays doubly exponentially (each cell about the square of its upwind neighbour: 1e-47, 1e-95,
// 1e-191, 0), so values of the second kind sit along the whole front all the time: a form that branched to the compiler's division
// for them was slower than no change at all.  Non-finite or absurd |S| >= 2^500 is outside the claim (such members are flagged).
// The general fluid keeps the compiler's division (four quotients with arbitrary operand ranges).
// fp32 sweeps (dtype = 32 plans): single precision has 2^32 operands, so there the short form is CHECKED on every one of them (below).
#pragma once
#include "fwd.h"

template <bool FD>
__device__ __forceinline__ double frac_flow_ieee(const FwdParams& p, double s) {
    double mw, mo;
    if (FD) {
        mw = s * s;
        double o = 1.0 - s;
        mo = o * o;
    } else {
        double den = (1.0 - p.swc) - p.sor;
        double S = (s - p.swc) / den;
        mw = (S * S) / p.vw;
        double o = 1.0 - S;
        mo = (o * o) / p.vo;
    }
    return mw / (mw + mo);
}

// n / d by the compiler's own sequence minus operand scaling and fix-up (see the header comment for when that is exact)
// Round 5: the reciprocal is refined by ONE cubic step r (1 + e + e^2), e = 1 - d r (three FMAs) instead of the compiler's two quadratic
// ones (four).  What the quotient's final correction needs of r is |r d - 1| <= 2^-53 (1 + tiny): q0 = RN(n r) is then within an ulp of
// n / d, the residual n - d q0 is exact, and RN(q0 + (n - d q0) r) = RN(n / d + (n / d - q0) eps) with a perturbation of 2^-53 ulp at
// most -- the same bound for both refinements (the seed's relative error is below 2^-20, its cube far below 2^-53; the quadratic pair
// ends at 2^-104, which this step does not use).  Same bits on every division of the parity tests (adversarial operands, whole runs).
__device__ __forceinline__ double div_unscaled(double n, double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    e = __builtin_fma(e, e, e);
    r = __builtin_fma(r, e, r);
    const double q = n * r;
    e = __builtin_fma(-d, q, n);
    return __builtin_fma(e, r, q);
}

template <bool FD>
__device__ __forceinline__ double frac_flow(const FwdParams& p, double s) {
    if (!FD) return frac_flow_ieee<false>(p, s);
    const double mw = s * s;
    const double o = 1.0 - s;
    const double mo = o * o;
    return div_unscaled(mw, mw + mo);
}

// ---------------------------------------------------------------- single precision (dtype = 32 plans)
template <bool FD>
__device__ __forceinline__ float frac_flow_ieee(const FwdParams& p, float s) {
    float mw, mo;
    if (FD) {
        mw = s * s;
        const float o = 1.0f - s;
        mo = o * o;
    } else {
        const float den = (float)((1.0 - p.swc) - p.sor);
        const float S = (s - (float)p.swc) / den;
        mw = (S * S) / (float)p.vw;
        const float o = 1.0f - S;
        mo = (o * o) / (float)p.vo;
    }
    return mw / (mw + mo);
}

// Round 5: FOUR instructions.  fw is a function of one float, so a shorter sequence can be checked on every bit pattern of s: on gfx950
// (its v_rcp_f32 seed included) rcp, q = n r and ONE residual correction give the IEEE quotient of n = s^2 by d = s^2 + (1 - s)^2 for all
// 2^32 values of s except |s| >= 6.5e18 -- exactly the values on which the eight-instruction form it replaces (the compiler's sequence
// without scaling and fix-up: v_rcp_f32, a Newton step, two residual corrections) differs from it as well
// (profiles/diag/div32_exhaustive.hip, profiles/r05/div32_exhaustive.txt).  Only for THIS n and d: a quotient
// in [0, 1] with d >= 0.5; not a general division.
__device__ __forceinline__ float div_unscaled(float n, float d) {
    const float r = __builtin_amdgcn_rcpf(d);
    const float q = n * r;
    const float e = __builtin_fmaf(-d, q, n);
    return __builtin_fmaf(e, r, q);
}

template <bool FD>
__device__ __forceinline__ float frac_flow(const FwdParams& p, float s) {
    if (!FD) return frac_flow_ieee<false>(p, s);
    const float mw = s * s;
    const float o = 1.0f - s;
    const float mo = o * o;
    return div_unscaled(mw, mw + mo);
}
