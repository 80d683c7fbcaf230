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
