// sat32.h -- the float32 saturation state of dtype = 32 plans: a compensated pair (base, dS), S = base + dS.
//
// The reference integrates the saturation in fp64 (ResSim.sim, HistoryMatch.py:362); dtype = 32 plans are this build's fast mode with a
// stated bar of 1e-3 on S over a whole run.  A plain float32 accumulator S <- S + increment misses that bar by orders of
// magnitude on the larger grids: the increments shrink with 1 / Nts (615 / 2458 / 9831 sub-steps per time step at 128^2 / 256^2 / 512^2)
// while the ulp of S does not, so every sub-step loses up to half an ulp of S and increments below that vanish altogether -- measured
// against the fp64 mode over 40 steps: 5e-4 / 2.4e-2 / 0.30 on S, a water-in-place deficit of up to 1.2e-3 of the pore volume
// (profiles/r05/fp32_drift_*_before.txt).  So the state is carried as two float32 words per cell:
//     * a sub-step forms s = base + dS (rounded once, feeds the fractional flow only), the five products and their sum in float32
//       exactly as before, and adds the increment to dS alone: dS <- dS + (acc + fi d).  The ulp of dS is that of the CHANGE since the
//       last fold, four to seven orders below that of S;
//     * every F32_FOLD sub-steps dS is folded into base by an exact two-sum: base + dS == base' + dS' in real arithmetic, base' =
//       fl(base + dS), so nothing is lost and |dS| stays below F32_FOLD increments;
//     * the state stored between time steps is fl(base + dS), one rounding per time step.
// One extra add per cell and sub-step, one extra register per cell.  Every fp32 sweep (generic, streaming, tiled, the register sweep of
// sat32s.hip) performs these operations in this order: they are bit-identical to each other and to the NumPy float32 specification
// oracle/ressim.py:saturation_step_stencil_f32c (tests/test_forward_gpu.py).
#pragma once

constexpr int F32_FOLD = 64;  // sub-steps between folds (a power of two); oracle/ressim.py: F32_FOLD

// (base, dS) <- (fl(base + dS), base + dS - fl(base + dS)): Knuth's two-sum, exact for any magnitudes.  Compiled with -ffp-contract=off
// and without fast-math: the compiler may not re-associate it away.
__device__ __forceinline__ void fold32(float& base, float& dS) {
    const float t = base + dS;
    const float bb = t - base;
    const float e = (base - (t - bb)) + (dS - bb);
    base = t;
    dS = e;
}

// The injector's source term fi d of a float32 plan, rounded JOINTLY with the cell's diagonal coefficient (round 6).  At an injector every
// face flux leaves the cell, c_C = -d (outflow) and outflow = fi up to the solver's residual, so at fw = 1 the cell's increment is
// c_C + fi d = 0: the cell fills up to S = 1 and stays.  Rounded to float32 INDEPENDENTLY the two may differ by an ulp of fi d (3e-8 at
// d = 1/3), the cell then gains that ulp per sub-step at fw = 1 -- 9 831 sub-steps x 40 steps -- until S sits far enough above 1 for fw
// to fall again: 1 + 1.7e-4 observed at 512 x 512 (profiles/diag/c5_smax.py), and 0 <= S <= 1 is an invariant of the scheme.
// So for cells with fi > 0 the float32 source is what is left of the fp64 SUM after the rounded coefficient is taken out,
//     fid32 = fl32((c_C + fi d) - cC32),
// i.e. cC32 + fid32 is the fp64 sum to one rounding (exactly 0 for a pure source cell, whose fp64 sum is the solver's 1e-13 residual):
// the increment at the injector is fid32 (1 - fw) + (inflow terms) >= 0 and vanishes at fw = 1 -- S <= 1 by construction.  The injected
// volume changes by |cC32 - c_C| <= half an ulp of fi d per sub-step, the same size as the rounding of fi d it replaces.  No instruction
// in the sub-step loop changes; every fp32 sweep and oracle/ressim.py:saturation_step_stencil_f32c form the term this way.
__device__ __forceinline__ float source32(double cC64, float cC32, double fi, double d) {
    return fi > 0.0 ? (float)((cC64 + fi * d) - (double)cC32) : (float)(fi * d);
}

// The diagonal coefficient of a float32 plan, settled so that a SATURATED neighbourhood does not gain water (round 6).  Where every fw is
// 1 -- the zone around the injector late in a run -- a cell's increment is the sum of its six float32 coefficients, in exact arithmetic
// d (q - div V) = 0.  Rounded to float32 one by one and summed by the kernels' float32 additions it is an ulp-sized residue instead, the
// same in every sub-step of a time step: positive, the cell creeps above 1 (measured at 512 x 512, 9 831 sub-steps a step: neighbours of
// the injector's cell at 1 + 8e-5 after 32 steps) -- and beyond 1 the fractional flow FALLS again, so nothing pulls it back.  So for the
// cells that enter the time step at S >= F32_HOT = 0.9997 (fw rounds to 1 in float32 from S = 1 - 1.7e-4 on; a cell below F32_HOT that gets there within the
// step creeps by at most 9 831 residues of 3.7e-9 = 3.6e-5 before the next step flags it) c_C is lowered by whole ulps (at most four; one does it) until
// that residue, evaluated exactly as the kernels evaluate it --
//     ((((cE + cN) + cC) + cS) + cW) + fid        in float32, this order
// -- is not positive: 0 <= S <= 1 then holds at the fixed point by construction.  Such a cell loses at most an ulp of
// c_C per sub-step (3e-8 of its throughput); applied to EVERY cell the rule would bias the water in place by 2.5e-7 of the pore volume
// over 20 steps at 128 x 128 (half of all cells have a positive residue), hence the restriction to the handful that need it.  Setup
// only: no instruction of the sub-step loop changes.
constexpr float F32_HOT = 0.9997f;
__device__ __forceinline__ float diag32(float cC, float cE, float cN, float cS, float cW, float fid, float s0) {
    if (!(s0 >= F32_HOT)) return cC;
#pragma unroll 1
    for (int it = 0; it < 4; ++it) {
        const float r = ((((cE + cN) + cC) + cS) + cW) + fid;
        if (!(r > 0.0f)) break;
        cC = nextafterf(cC, -INFINITY);
    }
    return cC;
}
