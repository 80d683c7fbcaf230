// forward.hip -- ensemble forward model (batched TPFA pressure + explicit upwind saturation):
// plan management, the C-ABI entry points, and the GENERIC kernels (any Nx, Ny <= 128).
//
// Replaces, for all N members at once:  forward_model -> utils.apply(comp1) -> set_perm -> ResSim.sim
//   notebooks/HistoryMatch.py:383-387, tools/utils.py:155-242, HistoryMatch.py:358-364, 160-164, 362.
// The arithmetic follows the listings restated in SURVEY.md Appendix A (the simulator package itself,
// TPFA-ResSim@adc89536, is not vendored in the reference).
//
// One workgroup = one ensemble member.  The generic kernels keep the per-member work arrays in HBM/L2
// scratch and use LDS only for the dense Ny x Ny Schur-complement block; they are the correctness
// baseline the 128x128 specialisations (press128s.hip, sat128.hip) are validated against.
//
// This file is compiled with -ffp-contract=off: every product/sum is rounded separately exactly as
// NumPy does; FMAs appear only where written explicitly (inside the pressure solve, which is not a
// bit-exact path).
#include "fwd_dev.h"
#include "sat32.h"
#include <algorithm>
#include <type_traits>
#include <chrono>

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double block_min(double v, double* red, int tid, int nthreads) {
    red[tid] = v;
    __syncthreads();
    for (int s = 1; s < nthreads; s <<= 1) {
        // tree over arbitrary thread counts: pairwise with stride doubling
        int idx = 2 * s * tid;
        if (idx + s < nthreads) red[idx] = fmin(red[idx], red[idx + s]);
        __syncthreads();
    }
    double r = red[0];
    __syncthreads();
    return r;
}

// ------------------------------------------------------------------------------------------------
// K = 0.1 + exp(5 x)          perm_transf, HistoryMatch.py:137-138
// ------------------------------------------------------------------------------------------------
__global__ void k_perm_transform(const double* __restrict__ x, double* __restrict__ K, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) K[i] = 0.1 + exp(5.0 * x[i]);
}

// ------------------------------------------------------------------------------------------------
// GENERIC pressure step: assemble TPFA transmissibilities, solve the 5-point SPD system by block
// elimination along ix (blocks of Ny unknowns; Schur complements inverted by symmetric Gauss-Jordan
// sweeps in LDS), back-substitute, form face fluxes.      SURVEY.md A.3 (listings TPFA, Pres)
// blockDim.x = Ny * groups.
// ------------------------------------------------------------------------------------------------
template <typename TS>
__global__ void k_pressure_generic(FwdParams p, const TS* __restrict__ S_base, long long S_stride, int k) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int m = blockIdx.x;
    const int tid = threadIdx.x;
    const int T = blockDim.x;
    const int Nx = p.Nx, Ny = p.Ny, Nxy = p.Nxy;
    const int LD = Ny | 1;
    const int groups = T / Ny;
    const int g = tid / Ny, c = tid % Ny;

    double* A = smem;                 // Ny*LD
    double* colbuf = A + Ny * LD;     // Ny
    double* yprev = colbuf + Ny;      // Ny
    double* ycur = yprev + Ny;        // Ny
    double* red = ycur + Ny;          // T

    const TS* S = S_base + (long long)m * S_stride;
    const double* Km = p.K + (long long)m * Nxy;
    const double* Kym = p.Ky ? p.Ky + (long long)m * Nxy : Km;
    double* TX = p.TX + (long long)m * (Nx + 1) * Ny;
    double* TY = p.TY + (long long)m * Nx * (Ny + 1);
    double* G = p.G + (long long)m * Nx * Ny * Ny;
    double* yv = p.yv + (long long)m * Nxy;
    double* P = p.P + (long long)m * Nxy;
    double* Vx = p.Vx + (long long)m * (Nx + 1) * Ny;
    double* Vy = p.Vy + (long long)m * Nx * (Ny + 1);
    const double* q = p.q + (long long)m * p.q_mstride + (long long)(p.q_cols > 1 ? k : 0) * Nxy;
    double* L = P;  // temporarily holds L = 1/(Mt*K)

    // --- mobility-weighted inverse permeability, harmonic-mean face transmissibilities
    assemble_transmissibilities<TS>(p, S, Km, Kym, L, TX, TY, tid, T);

    int bad = 0;
    // --- forward block elimination
    for (int i = 0; i < Nx; ++i) {
        const double* e = TX + i * Ny;  // coupling to block i-1: E = -diag(e)
        if (i > 0) {
            // t = G_{i-1} y_{i-1}   (A holds G_{i-1}; symmetric, so read columns as rows)
            double part = 0.0;
            if (g < groups)
                for (int r = g; r < Ny; r += groups) part = fma(A[r * LD + c], yprev[r], part);
            red[tid] = part;
            __syncthreads();
            if (tid < Ny) {
                double t = 0.0;
                for (int gg = 0; gg < groups; ++gg) t += red[gg * Ny + tid];
                ycur[tid] = q[i * Ny + tid] + e[tid] * t;
            }
            // A <- -(e G e)
            if (g < groups) {
                double ec = e[c];
                for (int r = g; r < Ny; r += groups) A[r * LD + c] = -(e[r] * A[r * LD + c] * ec);
            }
        } else {
            if (g < groups)
                for (int r = g; r < Ny; r += groups) A[r * LD + c] = 0.0;
            if (tid < Ny) ycur[tid] = q[tid];
        }
        __syncthreads();
        // add the tridiagonal block D_i
        if (tid < Ny) {
            int j = tid;
            double y1 = TY[i * (Ny + 1) + j], y2 = TY[i * (Ny + 1) + j + 1];
            double x1 = TX[i * Ny + j], x2 = TX[(i + 1) * Ny + j];
            double dg = y1 + y2 + x1 + x2;
            if (i == 0 && j == 0) dg += Km[0] + Kym[0];  // SPD pin: A[0,0] += Kx[0,0]+Ky[0,0]
            A[j * LD + j] += dg;
            if (j + 1 < Ny) {
                A[j * LD + j + 1] -= y2;
                A[(j + 1) * LD + j] -= y2;
            }
        }
        // symmetric sweeps: A <- -inv(A)
        for (int kk = 0; kk < Ny; ++kk) {
            __syncthreads();
            if (tid < Ny) colbuf[tid] = A[kk * LD + tid];
            __syncthreads();
            double d = colbuf[kk];
            if (!(d > 0.0) || !isfinite(d)) bad = 1;
            double pinv = 1.0 / d;
            if (g < groups) {
                double tc = colbuf[c] * pinv;
                for (int r = g; r < Ny; r += groups) {
                    double v;
                    if (r == kk)
                        v = (c == kk) ? -pinv : tc;
                    else if (c == kk)
                        v = colbuf[r] * pinv;
                    else
                        v = fma(-colbuf[r], tc, A[r * LD + c]);
                    A[r * LD + c] = v;
                }
            }
        }
        __syncthreads();
        // A <- -A = G_i ; store G_i and y_i
        if (g < groups)
            for (int r = g; r < Ny; r += groups) {
                double v = -A[r * LD + c];
                A[r * LD + c] = v;
                G[((long long)i * Ny + r) * Ny + c] = v;
            }
        if (tid < Ny) {
            yv[i * Ny + tid] = ycur[tid];
            yprev[tid] = ycur[tid];
        }
        __syncthreads();
    }
    // --- back substitution: x_i = G_i (y_i + e_{i+1} * x_{i+1}),   e_{i+1} = TX[i+1]
    for (int i = Nx - 1; i >= 0; --i) {
        if (tid < Ny) {
            double v = yv[i * Ny + tid];
            if (i < Nx - 1) v += TX[(i + 1) * Ny + tid] * ycur[tid];  // ycur holds x_{i+1}
            yprev[tid] = v;
        }
        __syncthreads();
        double part = 0.0;
        if (g < groups) {
            if (i == Nx - 1) {
                for (int r = g; r < Ny; r += groups) part = fma(A[r * LD + c], yprev[r], part);
            } else {
                const double* Gi = G + (long long)i * Ny * Ny;
                for (int r = g; r < Ny; r += groups) part = fma(Gi[r * Ny + c], yprev[r], part);
            }
        }
        red[tid] = part;
        __syncthreads();
        if (tid < Ny) {
            double t = 0.0;
            for (int gg = 0; gg < groups; ++gg) t += red[gg * Ny + tid];
            ycur[tid] = t;
            P[i * Ny + tid] = t;
        }
        __syncthreads();
    }
    // --- face fluxes
    face_fluxes(p, P, TX, TY, Vx, Vy, tid, T);
    if (bad && tid == 0) atomicOr(&p.status[m], HM_MEMBER_BAD_PIVOT);
}

// ------------------------------------------------------------------------------------------------
// GENERIC saturation step: CFL sub-step count, upwind coefficients, Nts explicit sub-steps, producer
// gather.   SURVEY.md A.4 (listings GenA, Upstream), obs_model HistoryMatch.py:212-213,363.
// Summation order of the update = the CSR row order SciPy gives the reference's matrix form
// (E, N, C, S, W), see oracle/ressim.py saturation_step_stencil.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void k_saturation_generic(FwdParams p, const T* __restrict__ Sin_base, T* __restrict__ Sout_base,
                                     long long S_stride, T* __restrict__ prods, int k) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int m = blockIdx.x;
    const int tid = threadIdx.x;
    const int NT = blockDim.x;
    const int Nx = p.Nx, Ny = p.Ny, Nxy = p.Nxy;
    double* red = smem;  // NT doubles

    const T* Sin = Sin_base + (long long)m * S_stride;
    T* S = Sout_base + (long long)m * S_stride;
    const double* Vx = p.Vx + (long long)m * (Nx + 1) * Ny;
    const double* Vy = p.Vy + (long long)m * Nx * (Ny + 1);
    const double* q = p.q + (long long)m * p.q_mstride + (long long)(p.q_cols > 1 ? k : 0) * Nxy;
    T* cE = (T*)p.coef + (long long)m * 6 * Nxy;
    T *cN = cE + Nxy, *cC = cN + Nxy, *cS = cC + Nxy, *cW = cS + Nxy, *fid = cW + Nxy;
    T* fw = (T*)p.fw + (long long)m * Nxy;
    constexpr bool F32 = std::is_same<T, float>::value;  // dtype = 32 plans: S[] is `base`, dSa[] the running change (sat32.h)
    float* dSa = F32 ? p.comp + (long long)m * Nxy : nullptr;

    // --- CFL: pm = min(pv / (Vi + fi))
    double lmin = INFINITY;
    for (int j = tid; j < Nxy; j += NT) {
        int ix = j / Ny, iy = j % Ny;
        double xp = fmax(Vx[ix * Ny + iy], 0.0), yp = fmax(Vy[ix * (Ny + 1) + iy], 0.0);
        double xn = fmin(Vx[(ix + 1) * Ny + iy], 0.0), yn = fmin(Vy[ix * (Ny + 1) + iy + 1], 0.0);
        double Vi = xp + yp - xn - yn;
        double fi = fmax(q[j], 0.0);
        double pv = p.h2 * (p.por ? p.por[j] : 1.0);
        lmin = fmin(lmin, pv / (Vi + fi));
    }
    double pm = block_min(lmin, red, tid, NT);
    double sat = p.swc + p.sor;
    double cfl = ((1.0 - sat) / 3.0) * pm;
    double ntsd = ceil(p.dt / cfl);
    int bad = !(ntsd >= 1.0 && ntsd <= 1.0e7);
    int Nts = bad ? 0 : (int)ntsd;
    if (tid == 0) {
        p.nts[(long long)m * p.nTime + k] = Nts;
        if (bad) atomicOr(&p.status[m], HM_MEMBER_BAD_CFL);
    }
    // --- upwind coefficients, pre-scaled by dtx = (dt/Nts)/pv
    for (int j = tid; j < Nxy; j += NT) {
        int ix = j / Ny, iy = j % Ny;
        double pv = p.h2 * (p.por ? p.por[j] : 1.0);
        double d = bad ? 0.0 : (p.dt / (double)Nts) / pv;
        double vxw = Vx[ix * Ny + iy], vxe = Vx[(ix + 1) * Ny + iy];
        double vys = Vy[ix * (Ny + 1) + iy], vyn = Vy[ix * (Ny + 1) + iy + 1];
        double fp = fmin(q[j], 0.0), fi = fmax(q[j], 0.0);
        double x1 = fmin(vxw, 0.0), x2 = fmax(vxe, 0.0), y1 = fmin(vys, 0.0), y2 = fmax(vyn, 0.0);
        const double cC64 = d * (fp + x1 - x2 + y1 - y2);
        cC[j] = (T)cC64;
        cW[j] = (T)(d * fmax(vxw, 0.0));
        cE[j] = (T)(d * (-fmin(vxe, 0.0)));
        cS[j] = (T)(d * fmax(vys, 0.0));
        cN[j] = (T)(d * (-fmin(vyn, 0.0)));
        if constexpr (F32) {
            fid[j] = source32(cC64, cC[j], fi, d);  // (sat32.h: rounded jointly with c_C)
            cC[j] = diag32(cC[j], cE[j], cN[j], cS[j], cW[j], fid[j], Sin[j]);  // (sat32.h: a saturated neighbourhood gains nothing)
        } else fid[j] = (T)(fi * d);
        S[j] = Sin[j];
        if constexpr (F32) dSa[j] = 0.0f;
    }
    __syncthreads();
    // --- explicit sub-steps
    for (int it = 0; it < Nts; ++it) {
        for (int j = tid; j < Nxy; j += NT) {
            T mw, mo;
            T s = S[j];
            if constexpr (F32) s = s + dSa[j];
            rel_perm<T>(p, s, mw, mo);
            fw[j] = mw / (mw + mo);
        }
        __syncthreads();
        for (int j = tid; j < Nxy; j += NT) {
            int ix = j / Ny, iy = j % Ny;
            T acc = (ix + 1 < Nx) ? cE[j] * fw[j + Ny] : T(0);
            if (iy + 1 < Ny) acc = acc + cN[j] * fw[j + 1];
            acc = acc + cC[j] * fw[j];
            if (iy > 0) acc = acc + cS[j] * fw[j - 1];
            if (ix > 0) acc = acc + cW[j] * fw[j - Ny];
            if constexpr (F32) {
                float b = S[j], e = dSa[j] + (acc + fid[j]);
                if ((it & (F32_FOLD - 1)) == F32_FOLD - 1) {
                    fold32(b, e);
                    S[j] = b;
                }
                dSa[j] = e;
            } else {
                S[j] = S[j] + (acc + fid[j]);
            }
        }
        __syncthreads();
    }
    if constexpr (F32) {  // the stored state of the next time step: fl(base + dS)
        for (int j = tid; j < Nxy; j += NT) S[j] = S[j] + dSa[j];
        __syncthreads();
    }
    // --- checks + producer observations
    int nonfinite = 0;
    for (int j = tid; j < Nxy; j += NT)
        if (!isfinite((double)S[j])) nonfinite = 1;
    if (nonfinite) atomicOr(&p.status[m], HM_MEMBER_NONFINITE);
    if (tid < p.nPrd) prods[((long long)m * p.nTime + k) * p.nPrd + tid] = S[p.prd_ind[m * p.prd_mstride + tid]];
}

// ------------------------------------------------------------------------------------------------
// STREAMING saturation step for grids too large for on-chip residency (Ny > 128; sat_variant 2 at any size): the same
// arithmetic as k_saturation_generic with a third of its memory traffic.  Per sub-step and cell the generic kernel moves
// ~96 B (fw written and re-read, six coefficient arrays, S read and written); here the upwind coefficients are re-derived
// from the face fluxes (Vx, Vy: 2 arrays) and the fractional flow of the four neighbours is re-evaluated from their
// saturations (cached reads) instead of being stored, so a sub-step is ONE pass -- read S_old, Vx, Vy, q; write S_new --
// and ONE barrier, ping-ponging between the output row and a scratch image, and the six coefficient images are not
// allocated at all (1.5 GB at N = 125, 512^2).  Five fw evaluations per cell instead of one make it ALU bound (fp64
// divisions) at about the generic kernel's memory-bound speed (measured 0.54 vs 0.61 ms per sub-step at 512^2); sharing
// the fw evaluations through LDS tiles is the next step for this path.
// Every expression is the generic kernel's (same operands, same order): bit-identical results.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void k_saturation_stream(FwdParams p, const T* __restrict__ Sin_base, T* __restrict__ Sout_base, long long S_stride,
                                    T* __restrict__ prods, int k) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int m = blockIdx.x, tid = threadIdx.x, NT = blockDim.x;
    const int Nx = p.Nx, Ny = p.Ny, Nxy = p.Nxy;
    double* red = smem;
    const T* Sin = Sin_base + (long long)m * S_stride;
    T* Sout = Sout_base + (long long)m * S_stride;
    T* Sbuf = (T*)p.fw + (long long)m * Nxy;  // second image of the ping-pong
    const double* Vx = p.Vx + (long long)m * (Nx + 1) * Ny;
    const double* Vy = p.Vy + (long long)m * Nx * (Ny + 1);
    const double* q = p.q + (long long)m * p.q_mstride + (long long)(p.q_cols > 1 ? k : 0) * Nxy;

    // --- CFL: pm = min(pv / (Vi + fi))
    double lmin = INFINITY;
    for (int j = tid; j < Nxy; j += NT) {
        const int ix = j / Ny, iy = j - ix * Ny;
        const double xp = fmax(Vx[ix * Ny + iy], 0.0), yp = fmax(Vy[ix * (Ny + 1) + iy], 0.0);
        const double xn = fmin(Vx[(ix + 1) * Ny + iy], 0.0), yn = fmin(Vy[ix * (Ny + 1) + iy + 1], 0.0);
        const double Vi = xp + yp - xn - yn;
        const double fi = fmax(q[j], 0.0);
        const double pv = p.h2 * (p.por ? p.por[j] : 1.0);
        lmin = fmin(lmin, pv / (Vi + fi));
    }
    const double pm = block_min(lmin, red, tid, NT);
    const double cfl = ((1.0 - (p.swc + p.sor)) / 3.0) * pm;
    const double ntsd = ceil(p.dt / cfl);
    const int bad = !(ntsd >= 1.0 && ntsd <= 1.0e7);
    const int Nts = bad ? 0 : (int)ntsd;
    if (tid == 0) {
        p.nts[(long long)m * p.nTime + k] = Nts;
        if (bad) atomicOr(&p.status[m], HM_MEMBER_BAD_CFL);
    }
    // dtype = 32 plans (sat32.h): the images of the ping-pong hold s = fl(base + dS), what the neighbours' fractional flow is formed
    // from; base and dS themselves are private to their cell's thread and live in two more arrays
    constexpr bool F32 = std::is_same<T, float>::value;
    float* basea = F32 ? p.comp + (long long)m * Nxy : nullptr;
    float* dSa = F32 ? p.comp + ((long long)p.N + m) * Nxy : nullptr;
    if constexpr (F32)
        for (int j = tid; j < Nxy; j += NT) {
            basea[j] = Sin[j];
            dSa[j] = 0.0f;
        }
    if (Nts == 0) {
        for (int j = tid; j < Nxy; j += NT) Sout[j] = F32 ? Sin[j] + T(0) : Sin[j];
        __syncthreads();
    }
    auto fwf = [&](T s) {
        T mw, mo;
        rel_perm<T>(p, s, mw, mo);
        return mw / (mw + mo);
    };
    const double d_uniform = Nts ? (p.dt / (double)Nts) / (p.h2 * 1.0) : 0.0;
    // sub-step it reads `src`, writes `dst`; the last one must write Sout
    for (int it = 0; it < Nts; ++it) {
        const T* __restrict__ src = it == 0 ? Sin : (((Nts - it) & 1) ? Sbuf : Sout);  // = destination of sub-step it-1
        T* __restrict__ dst = ((Nts - 1 - it) & 1) ? Sbuf : Sout;
        for (int j = tid; j < Nxy; j += NT) {
            const int ix = j / Ny, iy = j - ix * Ny;
            const double d = p.por ? (p.dt / (double)Nts) / (p.h2 * p.por[j]) : d_uniform;  // same value, two divisions less
            const double vxw = Vx[ix * Ny + iy], vxe = Vx[(ix + 1) * Ny + iy];
            const double vys = Vy[ix * (Ny + 1) + iy], vyn = Vy[ix * (Ny + 1) + iy + 1];
            const double qj = q[j];
            const double fp = fmin(qj, 0.0), fi = fmax(qj, 0.0);
            const double x1 = fmin(vxw, 0.0), x2 = fmax(vxe, 0.0), y1 = fmin(vys, 0.0), y2 = fmax(vyn, 0.0);
            const double cC64 = d * (fp + x1 - x2 + y1 - y2);
            T cC = (T)cC64;
            const T cW = (T)(d * fmax(vxw, 0.0));
            const T cE = (T)(d * (-fmin(vxe, 0.0)));
            const T cS = (T)(d * fmax(vys, 0.0));
            const T cN = (T)(d * (-fmin(vyn, 0.0)));
            T fid;
            if constexpr (F32) {
                fid = source32(cC64, cC, fi, d);           // (sat32.h: rounded jointly with c_C)
                cC = diag32(cC, cE, cN, cS, cW, fid, Sin[j]);      // (sat32.h: a saturated neighbourhood gains nothing)
            } else fid = (T)(fi * d);
            const T sc = src[j];
            T acc = (ix + 1 < Nx) ? cE * fwf(src[j + Ny]) : T(0);
            if (iy + 1 < Ny) acc = acc + cN * fwf(src[j + 1]);
            acc = acc + cC * fwf(sc);
            if (iy > 0) acc = acc + cS * fwf(src[j - 1]);
            if (ix > 0) acc = acc + cW * fwf(src[j - Ny]);
            if constexpr (F32) {
                float b = basea[j], e = dSa[j] + (acc + fid);
                if ((it & (F32_FOLD - 1)) == F32_FOLD - 1) {
                    fold32(b, e);
                    basea[j] = b;
                }
                dSa[j] = e;
                dst[j] = b + e;
            } else {
                dst[j] = sc + (acc + fid);
            }
        }
        __syncthreads();
    }
    int nonfinite = 0;
    for (int j = tid; j < Nxy; j += NT)
        if (!isfinite((double)Sout[j])) nonfinite = 1;
    if (nonfinite) atomicOr(&p.status[m], HM_MEMBER_NONFINITE);
    if (tid < p.nPrd) prods[((long long)m * p.nTime + k) * p.nPrd + tid] = Sout[p.prd_ind[m * p.prd_mstride + tid]];
}

// ------------------------------------------------------------------------------------------------
// TILED saturation step (default beyond 128 x 128; sat_variant 3 at any size): k_saturation_stream with the fractional
// flow shared through LDS.  The grid is swept in tiles of 64 x 256 cells; per tile the workgroup evaluates fw once for the
// tile and its one-cell halo into LDS (136 KB in fp64), then updates the tile from it.  One fw division per cell (+3 %
// halo) instead of five, still one pass over HBM per sub-step (read S_old, Vx, Vy, q; write S_new).  Same expressions,
// same order as the generic kernel: bit-identical.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(1024) void k_saturation_tiled(FwdParams p, const T* __restrict__ Sin_base, T* __restrict__ Sout_base,
                                                           long long S_stride, T* __restrict__ prods, int k, int only_timed_out, int* retried,
                                                           unsigned char* wet_mark, int wet_T) {
    constexpr int TH = 64, TW = 256, LW = TW + 2;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int m = blockIdx.x, tid = threadIdx.x, NT = blockDim.x;
    if (only_timed_out) {
        // The retry of the workgroup-team sweeps, decided on the device: this launch follows every team launch and redoes the time step
        // of exactly those members whose team gave up waiting for a neighbour (HM_MEMBER_SYNC_TIMEOUT: its workgroups were not all
        // resident -- CUs held by someone else).  S_in is untouched by the team sweep (it writes the other time slot), the result is
        // bit-identical; a member without the flag costs this workgroup one load.  No host synchronisation (round 4 read the flags back).
        __shared__ int go;
        if (tid == 0) {
            go = (p.status[m] & HM_MEMBER_SYNC_TIMEOUT) != 0;
            if (go) {
                atomicAnd(&p.status[m], ~HM_MEMBER_SYNC_TIMEOUT);
                if (retried) atomicAdd(retried, 1);  // hm_fwd_team_retries
                // (sat32s.hip keeps a record of which slabs of a member hold water: what a team that gave up wrote there describes its
                // broken run, not the state this retry produces -- every slab of this member counts as wet for the next step)
                for (int t = 0; wet_mark && t < wet_T; ++t) wet_mark[(size_t)m * wet_T + t] = 1;
            }
        }
        __syncthreads();
        if (!go) return;
    }
    const int Nx = p.Nx, Ny = p.Ny, Nxy = p.Nxy;
    double* red = smem;                          // NT doubles
    T* fwt = reinterpret_cast<T*>(smem + 1024);  // (TH + 2) x LW
    const T* Sin = Sin_base + (long long)m * S_stride;
    T* Sout = Sout_base + (long long)m * S_stride;
    T* Sbuf = (T*)p.fw + (long long)m * Nxy;
    const double* Vx = p.Vx + (long long)m * (Nx + 1) * Ny;
    const double* Vy = p.Vy + (long long)m * Nx * (Ny + 1);
    const double* q = p.q + (long long)m * p.q_mstride + (long long)(p.q_cols > 1 ? k : 0) * Nxy;

    double lmin = INFINITY;
    for (int j = tid; j < Nxy; j += NT) {
        const int ix = j / Ny, iy = j - ix * Ny;
        const double xp = fmax(Vx[ix * Ny + iy], 0.0), yp = fmax(Vy[ix * (Ny + 1) + iy], 0.0);
        const double xn = fmin(Vx[(ix + 1) * Ny + iy], 0.0), yn = fmin(Vy[ix * (Ny + 1) + iy + 1], 0.0);
        const double Vi = xp + yp - xn - yn;
        const double fi = fmax(q[j], 0.0);
        const double pv = p.h2 * (p.por ? p.por[j] : 1.0);
        lmin = fmin(lmin, pv / (Vi + fi));
    }
    const double pm = block_min(lmin, red, tid, NT);
    const double cfl = ((1.0 - (p.swc + p.sor)) / 3.0) * pm;
    const double ntsd = ceil(p.dt / cfl);
    const int bad = !(ntsd >= 1.0 && ntsd <= 1.0e7);
    const int Nts = bad ? 0 : (int)ntsd;
    if (tid == 0) {
        p.nts[(long long)m * p.nTime + k] = Nts;
        if (bad) atomicOr(&p.status[m], HM_MEMBER_BAD_CFL);
    }
    constexpr bool F32 = std::is_same<T, float>::value;  // dtype = 32 plans: as in k_saturation_stream
    float* basea = F32 ? p.comp + (long long)m * Nxy : nullptr;
    float* dSa = F32 ? p.comp + ((long long)p.N + m) * Nxy : nullptr;
    if constexpr (F32)
        for (int j = tid; j < Nxy; j += NT) {
            basea[j] = Sin[j];
            dSa[j] = 0.0f;
        }
    if (Nts == 0) {
        for (int j = tid; j < Nxy; j += NT) Sout[j] = F32 ? Sin[j] + T(0) : Sin[j];
        __syncthreads();
    }
    auto fwf = [&](T s) {
        T mw, mo;
        rel_perm<T>(p, s, mw, mo);
        return mw / (mw + mo);
    };
    const double d_uniform = Nts ? (p.dt / (double)Nts) / (p.h2 * 1.0) : 0.0;
    const int tx = tid >> 6, ty = tid & 63;  // 16 x 64 threads over a tile: rows stride 16, columns stride 64
    for (int it = 0; it < Nts; ++it) {
        const T* __restrict__ src = it == 0 ? Sin : (((Nts - it) & 1) ? Sbuf : Sout);
        T* __restrict__ dst = ((Nts - 1 - it) & 1) ? Sbuf : Sout;
        for (int x0 = 0; x0 < Nx; x0 += TH)
            for (int y0 = 0; y0 < Ny; y0 += TW) {
                const int th = min(TH, Nx - x0), tw = min(TW, Ny - y0);
                // fw of the tile and its halo (cells outside the grid: never used, their coefficients are skipped)
                for (int li = tx; li < th + 2; li += 16) {
                    const int ix = x0 + li - 1;
                    const bool rin = ix >= 0 && ix < Nx;
                    const int ixc = min(max(ix, 0), Nx - 1);
                    if (tw == TW) {  // full-width tile: 4 loads in flight, then the 4 divisions
                        T sv[4];
                        bool in[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int iy = y0 + ty + 64 * u - 1;
                            in[u] = rin && iy >= 0 && iy < Ny;
                            sv[u] = src[ixc * Ny + min(max(iy, 0), Ny - 1)];
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) fwt[li * LW + ty + 64 * u] = in[u] ? fwf(sv[u]) : T(0);
                        if (ty < 2) {
                            const int iy = y0 + ty + 255;
                            fwt[li * LW + ty + 256] = (rin && iy < Ny) ? fwf(src[ixc * Ny + min(iy, Ny - 1)]) : T(0);
                        }
                    } else {
                        for (int lj = ty; lj < tw + 2; lj += 64) {
                            const int iy = y0 + lj - 1;
                            const bool in = rin && iy >= 0 && iy < Ny;
                            fwt[li * LW + lj] = in ? fwf(src[ixc * Ny + min(max(iy, 0), Ny - 1)]) : T(0);
                        }
                    }
                }
                __syncthreads();
                auto update_cell = [&](int li, int lj, double vxw, double vxe, double vys, double vyn, double qj, T sc) {
                    const int ix = x0 + li, iy = y0 + lj, j = ix * Ny + iy;
                    const double d = p.por ? (p.dt / (double)Nts) / (p.h2 * p.por[j]) : d_uniform;
                    const double fp = fmin(qj, 0.0), fi = fmax(qj, 0.0);
                    const double x1 = fmin(vxw, 0.0), x2 = fmax(vxe, 0.0), y1 = fmin(vys, 0.0), y2 = fmax(vyn, 0.0);
                    const double cC64 = d * (fp + x1 - x2 + y1 - y2);
                    T cC = (T)cC64;
                    const T cW = (T)(d * fmax(vxw, 0.0));
                    const T cE = (T)(d * (-fmin(vxe, 0.0)));
                    const T cS = (T)(d * fmax(vys, 0.0));
                    const T cN = (T)(d * (-fmin(vyn, 0.0)));
                    T fid;
                    if constexpr (F32) {
                        fid = source32(cC64, cC, fi, d);           // (sat32.h: rounded jointly with c_C)
                        cC = diag32(cC, cE, cN, cS, cW, fid, Sin[j]);      // (sat32.h: a saturated neighbourhood gains nothing)
                    } else fid = (T)(fi * d);
                    const T* f = fwt + (li + 1) * LW + lj + 1;
                    T acc = (ix + 1 < Nx) ? cE * f[LW] : T(0);
                    if (iy + 1 < Ny) acc = acc + cN * f[1];
                    acc = acc + cC * f[0];
                    if (iy > 0) acc = acc + cS * f[-1];
                    if (ix > 0) acc = acc + cW * f[-LW];
                    if constexpr (F32) {
                        float b = basea[j], e = dSa[j] + (acc + fid);
                        if ((it & (F32_FOLD - 1)) == F32_FOLD - 1) {
                            fold32(b, e);
                            basea[j] = b;
                        }
                        dSa[j] = e;
                        dst[j] = b + e;
                    } else {
                        dst[j] = sc + (acc + fid);
                    }
                };
                for (int li = tx; li < th; li += 16) {
                    const int ix = x0 + li;
                    if (tw == TW) {  // full-width tile: the loads of the thread's 4 cells of this row are issued together
                        double a[4], b[4], c2[4], e[4], qq[4];
                        T sc[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int iy = y0 + ty + 64 * u, j = ix * Ny + iy;
                            a[u] = Vx[ix * Ny + iy]; b[u] = Vx[(ix + 1) * Ny + iy];
                            c2[u] = Vy[ix * (Ny + 1) + iy]; e[u] = Vy[ix * (Ny + 1) + iy + 1];
                            qq[u] = q[j]; sc[u] = src[j];
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) update_cell(li, ty + 64 * u, a[u], b[u], c2[u], e[u], qq[u], sc[u]);
                    } else {
                        for (int lj = ty; lj < tw; lj += 64) {
                            const int iy = y0 + lj, j = ix * Ny + iy;
                            update_cell(li, lj, Vx[ix * Ny + iy], Vx[(ix + 1) * Ny + iy], Vy[ix * (Ny + 1) + iy], Vy[ix * (Ny + 1) + iy + 1], q[j], src[j]);
                        }
                    }
                }
                __syncthreads();
            }
    }
    int nonfinite = 0;
    for (int j = tid; j < Nxy; j += NT)
        if (!isfinite((double)Sout[j])) nonfinite = 1;
    if (nonfinite) atomicOr(&p.status[m], HM_MEMBER_NONFINITE);
    if (tid < p.nPrd) prods[((long long)m * p.nTime + k) * p.nPrd + tid] = Sout[p.prd_ind[m * p.prd_mstride + tid]];
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
__global__ void k_or_status(int* status, int n, int bits) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicOr(&status[i], bits);
}

template <typename T>
__global__ void k_fill(T* p, T v, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}

template <typename TO>
__global__ void k_copy_rows(const TO* __restrict__ src, long long src_stride, TO* __restrict__ dst,
                            long long dst_stride, int rows, long long n) {
    // dst[r*dst_stride + j] = src[r*src_stride + j]
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < (long long)rows * n; i += stride) {
        long long r = i / n, j = i % n;
        dst[r * dst_stride + j] = src[r * src_stride + j];
    }
}

static int build_q(hm_fwd* f, int nInj, const int* inj_ind, const double* inj_rates, int inj_cols, int nPrd,
                   const int* prd_ind, const double* prd_rates, int prd_cols) {
    const FwdParams& p = f->p;
    int cols = (inj_cols > 1 || prd_cols > 1) ? p.nTime : 1;
    HM_REQUIRE(inj_cols == 1 || inj_cols == p.nTime, "inj_rates must have 1 or nTime columns (got %d)", inj_cols);
    HM_REQUIRE(prd_cols == 1 || prd_cols == p.nTime, "prd_rates must have 1 or nTime columns (got %d)", prd_cols);
    f->q_host.assign((size_t)cols * p.Nxy, 0.0);
    for (int k = 0; k < cols; ++k) {
        double* q = f->q_host.data() + (size_t)k * p.Nxy;
        double si = 0, sp = 0;
        for (int w = 0; w < nInj; ++w) {
            HM_REQUIRE(inj_ind[w] >= 0 && inj_ind[w] < p.Nxy, "injector %d outside the grid", w);
            double r = inj_rates[(size_t)w * inj_cols + (inj_cols > 1 ? k : 0)];
            q[inj_ind[w]] += r;
            si += r;
        }
        for (int w = 0; w < nPrd; ++w) {
            HM_REQUIRE(prd_ind[w] >= 0 && prd_ind[w] < p.Nxy, "producer %d outside the grid", w);
            double r = prd_rates[(size_t)w * prd_cols + (prd_cols > 1 ? k : 0)];
            q[prd_ind[w]] -= r;
            sp += r;
        }
        // HistoryMatch.py:182-184: total of the sources must equal that of the sinks
        HM_REQUIRE(fabs(si - sp) <= 1e-8 + 1e-5 * fabs(sp),
                   "sum of injection rates (%g) must equal sum of production rates (%g) at step %d", si, sp, k);
    }
    f->p.q_cols = cols;
    // runs of equal consecutive columns of the schedule: q_epoch[k] = first step of the run step k belongs to
    f->q_epoch.assign(cols, 0);
    for (int k = 1; k < cols; ++k) {
        const double *a = f->q_host.data() + (size_t)(k - 1) * p.Nxy, *b = a + p.Nxy;
        f->q_epoch[k] = std::equal(a, a + p.Nxy, b) ? f->q_epoch[k - 1] : k;
    }
    f->well_cells_host.assign(inj_ind, inj_ind + nInj);
    f->well_cells_host.insert(f->well_cells_host.end(), prd_ind, prd_ind + nPrd);
    return 0;
}

extern "C" int hm_fwd_create(hm_ctx* ctx, int N, int Nx, int Ny, double Lx, double Ly, int nInj, const int* inj_ind,
                             const double* inj_rates, int inj_rate_cols, int nPrd, const int* prd_ind,
                             const double* prd_rates, int prd_rate_cols, double dt, int nTime, double vw, double vo,
                             double swc, double sor, const double* porosity, int dtype, int keep_history,
                             hm_fwd** out) {
    HM_REQUIRE(ctx && out, "hm_fwd_create: NULL argument");
    HM_REQUIRE(N >= 1 && Nx >= 2 && Ny >= 2, "hm_fwd_create: need N>=1, Nx>=2, Ny>=2 (got %d,%d,%d)", N, Nx, Ny);
    HM_REQUIRE(Ny <= 4096 && Nx <= 4096, "hm_fwd_create: grid %dx%d too large (max 4096 per axis)", Nx, Ny);
    HM_REQUIRE(dtype == 64 || dtype == 32, "hm_fwd_create: dtype must be 64 or 32");
    HM_REQUIRE(nTime >= 1 && dt > 0, "hm_fwd_create: need nTime>=1, dt>0");
    HM_REQUIRE(nInj >= 1 && nPrd >= 1, "hm_fwd_create: need at least one injector and one producer");
    HM_REQUIRE(vw > 0 && vo > 0 && swc >= 0 && sor >= 0 && swc + sor < 1, "hm_fwd_create: bad fluid parameters");
    HM_HIP(hipSetDevice(ctx->device));
    hm_fwd* f = new hm_fwd();
    f->ctx = ctx;
    f->dtype = dtype;
    f->esz = dtype == 64 ? 8 : 4;
    f->keep_history = keep_history;
    FwdParams& p = f->p;
    p.N = N; p.Nx = Nx; p.Ny = Ny; p.Nxy = Nx * Ny;
    p.nInj = nInj; p.nPrd = nPrd; p.nTime = nTime;
    p.hx = Lx / Nx; p.hy = Ly / Ny; p.h2 = p.hx * p.hy;
    p.cx = 2 * p.hy / p.hx; p.cy = 2 * p.hx / p.hy;
    p.vw = vw; p.vo = vo; p.swc = swc; p.sor = sor;
    p.fluid_default = (vw == 1.0 && vo == 1.0 && swc == 0.0 && sor == 0.0);
    p.dt = dt;
    int rc = build_q(f, nInj, inj_ind, inj_rates, inj_rate_cols, nPrd, prd_ind, prd_rates, prd_rate_cols);
    if (rc) { delete f; return rc; }
    f->Lx = Lx; f->Ly = Ly;
    f->inj_ind_host.assign(inj_ind, inj_ind + nInj);
    f->prd_ind_host.assign(prd_ind, prd_ind + nPrd);
    f->inj_rates_host.assign(inj_rates, inj_rates + (size_t)nInj * inj_rate_cols);
    f->prd_rates_host.assign(prd_rates, prd_rates + (size_t)nPrd * prd_rate_cols);
    f->inj_cols = inj_rate_cols; f->prd_cols = prd_rate_cols;
    const size_t Nxy = p.Nxy, n = N;
#define ALLOC(buf, bytes) do { rc = hm_dev_alloc(f->buf, (bytes)); if (rc) { hm_fwd_destroy(f); return rc; } } while (0)
    ALLOC(K, n * Nxy * 8);
    ALLOC(perm_in, n * Nxy * 8);
    ALLOC(q, f->q_host.size() * 8);
    ALLOC(prd_ind, (size_t)nPrd * 4);
    ALLOC(well_cells, (size_t)(nInj + nPrd) * 4);
    ALLOC(TX, n * (Nx + 1) * Ny * 8);
    ALLOC(TY, n * Nx * (Ny + 1) * 8);
    const bool direct = Ny <= 128;  // block elimination with explicit inverse Schur complements; else CG
    // (the inverse Schur complements of the block elimination -- n * Nxy * Ny * 8 bytes, 16.8 GB at N_e = 1000, 128 x 128 -- are
    // allocated on the first launch of a kernel that needs them: launch_pressure)
    ALLOC(cg_r, direct ? 8 : n * Nxy * 8);
    ALLOC(cg_p, direct ? 8 : n * Nxy * 8);
    ALLOC(n_cg, n * nTime * 4);
    ALLOC(yv, n * Nxy * 8);
    ALLOC(P, n * Nxy * 8);
    ALLOC(Vx, n * (Nx + 1) * Ny * 8);
    ALLOC(Vy, n * Nx * (Ny + 1) * 8);
    ALLOC(status, n * 4);
    ALLOC(nts, n * nTime * 4);
    ALLOC(S, (keep_history ? n * (nTime + 1) : 2 * n) * Nxy * f->esz);
    ALLOC(prods, n * nTime * nPrd * f->esz);
    if (porosity) ALLOC(por, Nxy * 8);
#undef ALLOC
    p.K = (double*)f->K.p; p.q = (double*)f->q.p; p.prd_ind = (int*)f->prd_ind.p;
    p.well_cells = (int*)f->well_cells.p;
    p.TX = (double*)f->TX.p; p.TY = (double*)f->TY.p; p.G = (double*)f->G.p; p.yv = (double*)f->yv.p;
    p.P = (double*)f->P.p; p.Vx = (double*)f->Vx.p; p.Vy = (double*)f->Vy.p;
    p.status = (int*)f->status.p; p.nts = (int*)f->nts.p;
    p.cg_r = (double*)f->cg_r.p; p.cg_p = (double*)f->cg_p.p; p.n_cg = (int*)f->n_cg.p;
    p.cg_rtol = 1e-12; p.cg_max_iter = 40 * (Nx > Ny ? Nx : Ny) + 1000;
    f->cg_lazy = direct;
    p.por = porosity ? (double*)f->por.p : nullptr;
    p.coef = nullptr; p.fw = nullptr; p.comp = nullptr;
    hipStream_t s = ctx->stream;
    HM_HIP(hipMemcpyAsync(f->q.p, f->q_host.data(), f->q_host.size() * 8, hipMemcpyHostToDevice, s));
    HM_HIP(hipMemcpyAsync(f->prd_ind.p, prd_ind, (size_t)nPrd * 4, hipMemcpyHostToDevice, s));
    HM_HIP(hipMemcpyAsync(f->well_cells.p, f->well_cells_host.data(), f->well_cells_host.size() * 4, hipMemcpyHostToDevice, s));
    if (porosity) HM_HIP(hipMemcpyAsync(f->por.p, porosity, Nxy * 8, hipMemcpyHostToDevice, s));
    HM_HIP(hipMemsetAsync(f->status.p, 0, n * 4, s));
    HM_HIP(hipMemsetAsync(f->nts.p, 0, n * nTime * 4, s));
    HM_HIP(hipMemsetAsync(f->n_cg.p, 0, n * nTime * 4, s));
    HM_HIP(hipMemsetAsync(f->P.p, 0, f->P.bytes, s));  // CG warm start of the first step
    HM_HIP(hipMemsetAsync(f->Vx.p, 0, f->Vx.bytes, s));
    HM_HIP(hipMemsetAsync(f->Vy.p, 0, f->Vy.bytes, s));
    HM_HIP(hipStreamSynchronize(s));
    *out = f;
    return 0;
}

extern "C" void hm_fwd_destroy(hm_fwd* f) {
    if (!f) return;
    (void)hipSetDevice(f->ctx->device);
    (void)hipStreamSynchronize(f->ctx->stream);
    if (f->inner) hm_fwd_destroy(f->inner);
    if (f->is_inner) f->status = f->nts = f->n_cg = f->prods = DevBuf{};  // the outer plan's
    DevBuf* bufs[] = {&f->K, &f->por, &f->q, &f->prd_ind, &f->TX, &f->TY, &f->G, &f->yv, &f->P, &f->Vx,
                      &f->Vy, &f->coef, &f->fw, &f->status, &f->nts, &f->perm_in, &f->S, &f->prods, &f->well_cells, &f->cg_r, &f->cg_p, &f->n_cg,
                      &f->tl_TXc, &f->tl_TYc, &f->tl_pin, &f->tl_rc, &f->tl_yc, &f->tl_yv, &f->tl_G, &f->tl_cgs, &f->tl_done, &f->tl_ndone, &f->tl_z1, &f->tl_dinv, &f->tl_parts, &f->team_mem, &f->Ky, &f->comp, &f->slab_wet, &f->retried};
    for (DevBuf* b : bufs) hm_dev_free(*b);
    hm_nd_free(f->nd);
    f->t_total.destroy(); f->t_press.destroy(); f->t_sat.destroy();
    delete f;
}

extern "C" int hm_fwd_set_solver(hm_fwd* f, double rtol, int max_iter) {
    HM_REQUIRE(f, "hm_fwd_set_solver: NULL plan");
    HM_REQUIRE(rtol > 0 && rtol < 1 && max_iter >= 1, "hm_fwd_set_solver: need 0 < rtol < 1, max_iter >= 1");
    f->p.cg_rtol = rtol;
    f->p.cg_max_iter = max_iter;
    return 0;
}

extern "C" int hm_fwd_set_member_wells(hm_fwd* f, const double* q_all, int q_cols, const int* prd_ind_all) {
    HM_REQUIRE(f && q_all && prd_ind_all, "hm_fwd_set_member_wells: NULL argument");
    ++f->inputs_gen;
    FwdParams& p = f->p;
    HM_REQUIRE(q_cols == 1 || q_cols == p.nTime, "hm_fwd_set_member_wells: q_cols must be 1 or nTime (%d), got %d", p.nTime, q_cols);
    for (long long i = 0; i < (long long)p.N * p.nPrd; ++i)
        HM_REQUIRE(prd_ind_all[i] >= 0 && prd_ind_all[i] < p.Nxy, "hm_fwd_set_member_wells: producer cell %d outside the grid", prd_ind_all[i]);
    HM_HIP(hipSetDevice(f->ctx->device));
    hipStream_t s = f->ctx->stream;
    HM_HIP(hipStreamSynchronize(s));
    const size_t qb = (size_t)p.N * q_cols * p.Nxy * 8, pb = (size_t)p.N * p.nPrd * 4;
    int rc;
    if (f->q.bytes < qb) { hm_dev_free(f->q); if ((rc = hm_dev_alloc(f->q, qb))) return rc; }
    if (f->prd_ind.bytes < pb) { hm_dev_free(f->prd_ind); if ((rc = hm_dev_alloc(f->prd_ind, pb))) return rc; }
    HM_HIP(hipMemcpy(f->q.p, q_all, qb, hipMemcpyHostToDevice));
    HM_HIP(hipMemcpy(f->prd_ind.p, prd_ind_all, pb, hipMemcpyHostToDevice));
    p.q = (double*)f->q.p;
    p.prd_ind = (int*)f->prd_ind.p;
    p.q_cols = q_cols;
    p.q_mstride = (long long)q_cols * p.Nxy;
    p.prd_mstride = p.nPrd;
    return 0;
}

extern "C" int hm_fwd_set_variant(hm_fwd* f, int pressure_variant, int saturation_variant) {
    HM_REQUIRE(f, "hm_fwd_set_variant: NULL plan");
    ++f->inputs_gen;
    f->press_variant = pressure_variant;
    f->sat_variant = saturation_variant;
    return 0;
}

extern "C" int hm_fwd_set_inputs(hm_fwd* f, const void* perm, int perm_is_transformed, const void* wsat0) {
    HM_REQUIRE(f && perm, "hm_fwd_set_inputs: NULL argument");
    ++f->inputs_gen;
    HM_HIP(hipSetDevice(f->ctx->device));
    hipStream_t s = f->ctx->stream;
    const FwdParams& p = f->p;
    size_t n = (size_t)p.N * p.Nxy;
    if (perm_is_transformed) {
        int rc = hm_h2d_large(f->ctx, f->K.p, perm, n * 8);
        if (rc) return rc;
    } else {
        int rc = hm_h2d_large(f->ctx, f->perm_in.p, perm, n * 8);
        if (rc) return rc;
        hipLaunchKernelGGL(k_perm_transform, dim3(2048), dim3(256), 0, s, (const double*)f->perm_in.p, (double*)f->K.p, (long long)n);
        HM_HIP(hipGetLastError());
    }
    long long stride;
    void* S0 = fwd_S_ptr(f, 0, &stride);
    if (wsat0) {
        if (f->keep_history) {
            HM_HIP(hipMemcpy2DAsync(S0, (size_t)stride * f->esz, wsat0, (size_t)p.Nxy * f->esz, (size_t)p.Nxy * f->esz, p.N, hipMemcpyHostToDevice, s));
        } else {
            HM_HIP(hipMemcpyAsync(S0, wsat0, n * f->esz, hipMemcpyHostToDevice, s));
        }
    } else {
        if (f->keep_history) {
            HM_HIP(hipMemset2DAsync(S0, (size_t)stride * f->esz, 0, (size_t)p.Nxy * f->esz, p.N, s));
        } else {
            HM_HIP(hipMemsetAsync(S0, 0, n * f->esz, s));
        }
    }
    HM_HIP(hipMemsetAsync(f->status.p, 0, (size_t)p.N * 4, s));
    f->cur = 0;
    f->inner_S_step = -1;
    HM_HIP(hipStreamSynchronize(s));
    return 0;
}

// Anisotropic permeability: the y-component of K per member (the x-component is the `perm` of hm_fwd_set_inputs*).  The reference's
// set_perm stacks [p, p] (HistoryMatch.py:160-164), so its own runs never need this; the simulator's K is (2, Nx, Ny) all the same
// (TX from K[0], TY from K[1], the SPD pin K[0,0,0] + K[1,0,0]).  NULL returns the plan to Kx = Ky.
extern "C" int hm_fwd_set_perm_y(hm_fwd* f, const void* perm_y, int perm_is_transformed) {
    HM_REQUIRE(f, "hm_fwd_set_perm_y: NULL plan");
    ++f->inputs_gen;
    HM_HIP(hipSetDevice(f->ctx->device));
    hipStream_t s = f->ctx->stream;
    if (!perm_y) {
        f->p.Ky = nullptr;
        return 0;
    }
    const size_t n = (size_t)f->p.N * f->p.Nxy;
    if (!f->Ky.p) {
        int rc = hm_dev_alloc(f->Ky, n * 8);
        if (rc) return rc;
    }
    if (perm_is_transformed) {
        HM_HIP(hipMemcpyAsync(f->Ky.p, perm_y, n * 8, hipMemcpyHostToDevice, s));
    } else {
        HM_HIP(hipMemcpyAsync(f->perm_in.p, perm_y, n * 8, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_perm_transform, dim3(2048), dim3(256), 0, s, (const double*)f->perm_in.p, (double*)f->Ky.p, (long long)n);
        HM_HIP(hipGetLastError());
    }
    HM_HIP(hipStreamSynchronize(s));
    f->p.Ky = (const double*)f->Ky.p;
    return 0;
}

template <typename TI, typename TO>
__global__ void k_convert(const TI* __restrict__ in, TO* __restrict__ out, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = (TO)in[i];
}

// Device-resident chaining: the permeability input comes from a DEVICE buffer (e.g. the updated
// ensemble of an hm_upd plan, hm_upd_device_ptr(u, "E_out")), converted to fp64 if needed; initial saturation zero.
// Same stream as the producer when both plans share the context: no synchronisation, no PCIe.
extern "C" int hm_fwd_set_inputs_device(hm_fwd* f, const void* perm_dev, int perm_dtype, int perm_is_transformed) {
    HM_REQUIRE(f && perm_dev, "hm_fwd_set_inputs_device: NULL argument");
    ++f->inputs_gen;
    HM_REQUIRE(perm_dtype == 64 || perm_dtype == 32, "hm_fwd_set_inputs_device: perm_dtype must be 64 or 32");
    HM_HIP(hipSetDevice(f->ctx->device));
    hipStream_t s = f->ctx->stream;
    const FwdParams& p = f->p;
    const long long n = (long long)p.N * p.Nxy;
    double* dst = (double*)(perm_is_transformed ? f->K.p : f->perm_in.p);
    if (perm_dtype == 64) HM_HIP(hipMemcpyAsync(dst, perm_dev, (size_t)n * 8, hipMemcpyDeviceToDevice, s));
    else hipLaunchKernelGGL((k_convert<float, double>), dim3(2048), dim3(256), 0, s, (const float*)perm_dev, dst, n);
    if (!perm_is_transformed)
        hipLaunchKernelGGL(k_perm_transform, dim3(2048), dim3(256), 0, s, (const double*)f->perm_in.p, (double*)f->K.p, n);
    HM_HIP(hipGetLastError());
    long long stride;
    void* S0 = fwd_S_ptr(f, 0, &stride);
    if (f->keep_history) HM_HIP(hipMemset2DAsync(S0, (size_t)stride * f->esz, 0, (size_t)p.Nxy * f->esz, p.N, s));
    else HM_HIP(hipMemsetAsync(S0, 0, (size_t)n * f->esz, s));
    HM_HIP(hipMemsetAsync(f->status.p, 0, (size_t)p.N * 4, s));
    f->cur = 0;
    f->inner_S_step = -1;
    return 0;
}

static int ensure_generic_sat_scratch(hm_fwd* f, bool need_coef) {
    size_t n = (size_t)f->p.N * f->p.Nxy;
    int rc = 0;
    if (need_coef && !f->coef.p) {
        if ((rc = hm_dev_alloc(f->coef, 6 * n * f->esz))) return rc;
        f->p.coef = f->coef.p;
    }
    if (!f->fw.p) {
        if ((rc = hm_dev_alloc(f->fw, n * f->esz))) return rc;
        f->p.fw = f->fw.p;
    }
    if (f->dtype == 32 && !f->comp.p) {  // the compensated pair of dtype = 32 plans (sat32.h): base and dS images
        if ((rc = hm_dev_alloc(f->comp, 2 * n * 4))) return rc;
        f->p.comp = (float*)f->comp.p;
    }
    return 0;
}

static int generic_threads(int Ny) {
    int maxT = Ny > 32 ? 1024 : 256;
    return Ny * (maxT / Ny);
}

// ------------------------------------------------------------------------------------------------
// EMBEDDED GRIDS.  The fast kernels are written for 128 x 128, 256 x 256 and 512 x 512 cells (press_nd.hip, sat128r.hip, sat256s.hip,
// sat32s.hip); every other grid used to take the generic pair -- at 100 x 100 and 1000 members 68 + 29 ms a time step against the
// 5.6 + 8.3 ms of the LARGER 128 x 128 grid.  Such a grid now runs inside a plan of the next of those squares, with the same cell size:
// its cells in the corner at the origin, the others with permeability ZERO.  A face next to such a cell has 1 / (mobility K) = inf on one side, so its harmonic-mean
// transmissibility is c / inf = 0 exactly -- the no-flow boundary the grid has there anyway; the padding's own equations are 1 p = 0
// (press_nd.hip gives an all-zero row a unit diagonal), its fluxes 0, its saturation stays 0 (the sweeps skip dry bands), its CFL
// term pv / 0 = inf never is the minimum.  What the sweeps compute for the grid's own cells is the same arithmetic on the same
// operands as on the grid alone (a zero coefficient times a zero fractional flow adds +0), so they stay bit-identical to the oracle
// for given fluxes; the pressure solve is a different elimination order of the same system, as on every grid.
// The OUTER plan (the one the caller holds) keeps every buffer in the caller's layout; permeability and the current saturation are
// copied in when they changed, the new saturation is copied out after every sweep, P / Vx / Vy / TX / TY when somebody asks for them.
// Applies with the default kernel variants only: hm_fwd_set_variant(1, 1) (or hm_fwd_set_debug "embed" 0) is the generic pair on the
// grid as given, the in-library cross-check.
// ------------------------------------------------------------------------------------------------

template <typename T>
__global__ void k_embed2d(const T* __restrict__ src, long long s_ms, int s_cols, T* __restrict__ dst, long long d_ms, int d_rows, int d_cols,
                          int rows, int cols, T fill, int N) {
    const long long per = (long long)d_rows * d_cols, n = per * N, stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const long long m = i / per, rc = i - m * per;
        const int r = (int)(rc / d_cols), c = (int)(rc - (long long)r * d_cols);
        dst[m * d_ms + rc] = (r < rows && c < cols) ? src[m * s_ms + (long long)r * s_cols + c] : fill;
    }
}
template <typename T>
__global__ void k_extract2d(const T* __restrict__ src, long long s_ms, int s_cols, T* __restrict__ dst, long long d_ms, int rows, int cols, int N) {
    const long long per = (long long)rows * cols, n = per * N, stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const long long m = i / per, rc = i - m * per;
        const int r = (int)(rc / cols), c = (int)(rc - (long long)r * cols);
        dst[m * d_ms + rc] = src[m * s_ms + (long long)r * s_cols + c];
    }
}

static int launch_pressure(hm_fwd* f, int k);
static int launch_saturation(hm_fwd* f, int k);

// The inner 128 x 128 plan of `f` if this launch is to run embedded (created on first use), else NULL.
static hm_fwd* embedded_inner(hm_fwd* f, int* rc_out) {
    *rc_out = 0;
    const FwdParams& p = f->p;
    if (f->is_inner || !f->dbg_embed) return nullptr;
    const int big = p.Nx > p.Ny ? p.Nx : p.Ny, EMB = big <= 128 ? 128 : big <= 256 ? 256 : 512;  // the square the fast kernels exist for
    if (big > 512 || (p.Nx == EMB && p.Ny == EMB)) return nullptr;
    if (small_forward_applies(f)) return nullptr;                                      // the one-launch kernel of small grids takes it (small.hip: the same predicate)
    if (EMB > 128 && (p.Ny == 128 || (p.Nx % 128 == 0 && p.Ny % 128 == 0)) && f->dbg_embed != 2) return nullptr;  // grids of 128-wide blocks / tiles keep their own kernels ("embed" 2: they too)
    if (!(f->press_variant == 0 || f->press_variant == 12 || f->press_variant == 14) || f->sat_variant != 0) return nullptr;
    if (p.q_mstride != 0 || p.por != nullptr || p.Ky != nullptr) return nullptr;       // per-member wells, porosity field, anisotropy: generic
    if (f->raw_q_exposed) return nullptr;  // the caller holds a pointer to the OUTER source field: writes through it would never reach the inner plan's
    if (!f->inner) {
        auto remap = [&](int cell) { return (cell / p.Ny) * EMB + cell % p.Ny; };
        std::vector<int> inj(f->inj_ind_host), prd(f->prd_ind_host);
        for (int& c : inj) c = remap(c);
        for (int& c : prd) c = remap(c);
        hm_fwd* in = nullptr;
        int rc = hm_fwd_create(f->ctx, p.N, EMB, EMB, p.hx * EMB, p.hy * EMB, p.nInj, inj.data(), f->inj_rates_host.data(), f->inj_cols, p.nPrd,
                               prd.data(), f->prd_rates_host.data(), f->prd_cols, p.dt, p.nTime, p.vw, p.vo, p.swc, p.sor, nullptr, f->dtype, 0, &in);
        if (rc) { *rc_out = rc; return nullptr; }
        // per-member outputs are the outer plan's: the inner kernels write status, sub-step counts and producer series straight there
        hm_dev_free(in->status); hm_dev_free(in->nts); hm_dev_free(in->n_cg); hm_dev_free(in->prods);
        in->status = f->status; in->nts = f->nts; in->n_cg = f->n_cg; in->prods = f->prods;
        in->p.status = (int*)f->status.p; in->p.nts = (int*)f->nts.p; in->p.n_cg = (int*)f->n_cg.p;
        in->is_inner = true;
        // exactly the cell size of the outer grid (Lx / Nx * 128 / 128 may round differently)
        in->p.hx = p.hx; in->p.hy = p.hy; in->p.h2 = p.h2; in->p.cx = p.cx; in->p.cy = p.cy;
        f->inner = in;
        f->emb = EMB;
        f->inner_K_gen = -1;
        f->inner_S_step = -1;
    }
    // everything of the outer plan that steers the inner kernels: solver choice and tolerances, and what the caller may have written
    // behind the library's back (hm_fwd_device_ptr) -- the inner plan then keeps nothing from step to step either (sat32s.hip's wet-slab
    // record, press_nd.hip's dry fronts)
    f->inner->press_variant = f->press_variant;
    f->inner->p.cg_rtol = p.cg_rtol;
    f->inner->p.cg_max_iter = p.cg_max_iter;
    f->inner->raw_state_exposed = f->raw_state_exposed;
    f->inner->raw_field_exposed = f->raw_field_exposed;
    f->inner->dbg_slab_margin = f->dbg_slab_margin;
    f->inner->dbg_lazy_flux = f->dbg_lazy_flux;
    return f->inner;
}

#define EMB_GRID dim3(2048), dim3(256), 0, s
// permeability (when the outer plan's inputs changed) and the saturation of time index k (unless the inner plan's last sweep left it there)
static int embed_inputs(hm_fwd* f, hm_fwd* in, int k) {
    const FwdParams& p = f->p;
    const int EMB = f->emb;
    hipStream_t s = f->ctx->stream;
    if (f->inner_K_gen != f->inputs_gen || f->raw_field_exposed) {  // (a raw pointer to K is out: it may have been written since the last step)
        hipLaunchKernelGGL(k_embed2d<double>, EMB_GRID, (const double*)f->K.p, (long long)p.Nxy, p.Ny, (double*)in->K.p, (long long)EMB * EMB, EMB, EMB,
                           p.Nx, p.Ny, 0.0, p.N);
        f->inner_K_gen = f->inputs_gen;
        ++in->inputs_gen;
    }
    if (f->inner_S_step != k || f->raw_state_exposed) {
        long long so, si;
        const void* So = fwd_S_ptr(f, k, &so);
        void* Si = fwd_S_ptr(in, k, &si);
        if (f->dtype == 64) hipLaunchKernelGGL(k_embed2d<double>, EMB_GRID, (const double*)So, so, p.Ny, (double*)Si, si, EMB, EMB, p.Nx, p.Ny, 0.0, p.N);
        else hipLaunchKernelGGL(k_embed2d<float>, EMB_GRID, (const float*)So, so, p.Ny, (float*)Si, si, EMB, EMB, p.Nx, p.Ny, 0.0f, p.N);
        f->inner_S_step = k;
        in->cur = k;
    }
    HM_HIP(hipGetLastError());
    return 0;
}
// P, Vx, Vy, TX, TY of the outer plan from the inner plan's (when they are older)
static int extract_fields(hm_fwd* f, bool fluxes = true) {
    if (fluxes) {  // (lazy face fluxes, fwd.h: whoever is about to read or hand out Vx / Vy comes through here; the sweep dispatch decides for itself)
        if (int rc = nd128_materialize_fluxes(f)) return rc;
        if (f->inner)
            if (int rc = nd128_materialize_fluxes(f->inner)) return rc;
    } else if (f->inner && f->fields_stale) {
        if (int rc = nd128_materialize_fluxes(f->inner)) return rc;  // (the copy-out below reads the inner plan's fluxes)
    }
    if (!f->inner || !f->fields_stale) return 0;
    const FwdParams& p = f->p;
    hm_fwd* in = f->inner;
    hipStream_t s = f->ctx->stream;
    const int EMB = f->emb;
    const long long e2 = (long long)EMB * EMB, ex = (long long)(EMB + 1) * EMB;
    hipLaunchKernelGGL(k_extract2d<double>, EMB_GRID, (const double*)in->P.p, e2, EMB, (double*)f->P.p, (long long)p.Nxy, p.Nx, p.Ny, p.N);
    hipLaunchKernelGGL(k_extract2d<double>, EMB_GRID, (const double*)in->Vx.p, ex, EMB, (double*)f->Vx.p, (long long)(p.Nx + 1) * p.Ny, p.Nx + 1, p.Ny, p.N);
    hipLaunchKernelGGL(k_extract2d<double>, EMB_GRID, (const double*)in->TX.p, ex, EMB, (double*)f->TX.p, (long long)(p.Nx + 1) * p.Ny, p.Nx + 1, p.Ny, p.N);
    hipLaunchKernelGGL(k_extract2d<double>, EMB_GRID, (const double*)in->Vy.p, ex, EMB + 1, (double*)f->Vy.p, (long long)p.Nx * (p.Ny + 1), p.Nx, p.Ny + 1, p.N);
    hipLaunchKernelGGL(k_extract2d<double>, EMB_GRID, (const double*)in->TY.p, ex, EMB + 1, (double*)f->TY.p, (long long)p.Nx * (p.Ny + 1), p.Nx, p.Ny + 1, p.N);
    HM_HIP(hipGetLastError());
    f->fields_stale = false;
    return 0;
}
static int embedded_pressure(hm_fwd* f, hm_fwd* in, int k) {
    int rc = embed_inputs(f, in, k);
    if (!rc) rc = launch_pressure(in, k);
    f->fields_stale = true;
    f->inner_V_dirty = false;
    return rc;
}
static int embedded_saturation(hm_fwd* f, hm_fwd* in, int k) {
    const FwdParams& p = f->p;
    const int EMB = f->emb;
    hipStream_t s = f->ctx->stream;
    int rc = embed_inputs(f, in, k);
    if (rc) return rc;
    if (f->inner_V_dirty) {  // fluxes given by the caller (hm_fwd_set_field): zero on every face of the padding
        const long long ex = (long long)(EMB + 1) * EMB;
        hipLaunchKernelGGL(k_embed2d<double>, EMB_GRID, (const double*)f->Vx.p, (long long)(p.Nx + 1) * p.Ny, p.Ny, (double*)in->Vx.p, ex, EMB + 1, EMB, p.Nx + 1, p.Ny, 0.0, p.N);
        hipLaunchKernelGGL(k_embed2d<double>, EMB_GRID, (const double*)f->Vy.p, (long long)p.Nx * (p.Ny + 1), p.Ny + 1, (double*)in->Vy.p, ex, EMB, EMB + 1, p.Nx, p.Ny + 1, 0.0, p.N);
        f->inner_V_dirty = false;
        in->flux_pending = false;  // (the caller's fluxes, not the ones of the inner plan's last pressure step)
    }
    if ((rc = launch_saturation(in, k))) return rc;
    long long so, si;
    void* So = fwd_S_ptr(f, k + 1, &so);
    const void* Si = fwd_S_ptr(in, k + 1, &si);
    if (f->dtype == 64) hipLaunchKernelGGL(k_extract2d<double>, EMB_GRID, (const double*)Si, si, EMB, (double*)So, so, p.Nx, p.Ny, p.N);
    else hipLaunchKernelGGL(k_extract2d<float>, EMB_GRID, (const float*)Si, si, EMB, (float*)So, so, p.Nx, p.Ny, p.N);
    HM_HIP(hipGetLastError());
    f->inner_S_step = k + 1;
    in->cur = k + 1;
    return 0;
}
#undef EMB_GRID

static int launch_pressure(hm_fwd* f, int k) {
    const FwdParams& p = f->p;
    hipStream_t s = f->ctx->stream;
    long long stride;
    void* S = fwd_S_ptr(f, k, &stride);
    int rc = 0;
    hm_fwd* in = embedded_inner(f, &rc);
    if (rc) return rc;
    if ((rc = f->t_press.begin(s))) return rc;
    if (in) {
        if ((rc = embedded_pressure(f, in, k))) return rc;
        rc = f->t_press.end(s);
        f->n_press++;
        return rc;
    }
    f->fields_stale = false;  // (this plan's own pressure step: its fields are the current ones)
    f->flux_pending = false;  // (set again by the 128 x 128 nested dissection, which leaves the fluxes to their readers)
    int done = -1;
    // press_variant: 0 the default (128 x 128: nested dissection, press_nd.hip; other grids with Ny = 128: press128s), 1 generic (the
    // in-library cross-check), 9 Jacobi-CG, 7 the 16-wave form of press128s, 12 nested dissection (14: every front eliminated every step), 13 press128s (block elimination,
    // symmetric tiles, 8 waves) also at 128 x 128.  (The first three generations of the 128-wide solver -- rank-1 VALU sweeps, full-tile
    // rank-4 and rank-16 matrix-core panels -- were removed in round 2; their timings are in profiles/README.md.)
    // 256 x 256 and 512 x 512: nested dissection as well (press_nd256.o / press_nd512.o: the big-front kernels) for variants 0 / 12 / 14;
    // 15 (and 9, 11) keep the conjugate-gradient solvers there.  On these grids 0 (and 14) check every solve a posteriori and hand a member
    // the elimination cannot solve to the CG (one stream synchronisation per time step); 12 does not: hm_fwd_run stays asynchronous.
    const int pv = f->press_variant;
    const bool nd_variant = pv == 0 || pv == 12 || pv == 14;
    if (nd_variant && pressure_nd_applies256(p)) done = launch_pressure_nd256(f, S, stride, k);
    else if (nd_variant && pressure_nd_applies512(p)) done = launch_pressure_nd512(f, S, stride, k);
    else if (pv == 9 || p.Ny > 128) {
        if (f->cg_lazy) {  // CG requested on a small grid: the work vectors were not allocated at creation
            hm_dev_free(f->cg_r); hm_dev_free(f->cg_p);
            int rc2 = hm_dev_alloc(f->cg_r, (size_t)p.N * p.Nxy * 8);
            if (!rc2) rc2 = hm_dev_alloc(f->cg_p, (size_t)p.N * p.Nxy * 8);
            if (rc2) return rc2;
            f->p.cg_r = (double*)f->cg_r.p; f->p.cg_p = (double*)f->cg_p.p;
            f->cg_lazy = false;
        }
        if (pv != 9 && f->cg_precond == 0 && pressure_two_level_applies(p)) done = launch_pressure_two_level(f, S, stride, k);
        else done = launch_pressure_pcg(f, S, stride, k);
    } else if (nd_variant && pressure_nd_applies(p)) done = launch_pressure_nd(f, S, stride, k);  // 128 x 128: nested dissection (14: without reuse across time steps)
    if (done < 0 && p.Ny <= 128 && pv != 9) {
        if (!f->G.p) {  // every other direct solver keeps its inverse Schur complements
            int rc2 = hm_dev_alloc(f->G, (size_t)p.N * p.Nxy * p.Ny * 8);
            if (rc2) return rc2;
            f->p.G = (double*)f->G.p;
        }
        if (pv != 1) done = launch_pressure_128s(f, S, stride, k);  // 13: the block elimination also where nested dissection applies
    }
    if (done > 0) return done;
    if (done < 0) {
        int T = generic_threads(p.Ny);
        size_t lds = ((size_t)p.Ny * (p.Ny | 1) + 3 * p.Ny + T) * 8;
        if (f->dtype == 64) {
            HM_HIP(hipFuncSetAttribute((const void*)k_pressure_generic<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(k_pressure_generic<double>, dim3(p.N), dim3(T), lds, s, p, (const double*)S, stride, k);
        } else {
            HM_HIP(hipFuncSetAttribute((const void*)k_pressure_generic<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(k_pressure_generic<float>, dim3(p.N), dim3(T), lds, s, p, (const float*)S, stride, k);
        }
        HM_HIP(hipGetLastError());
    }
    rc = f->t_press.end(s);
    f->n_press++;
    return rc;
}

static int launch_saturation(hm_fwd* f, int k) {
    const FwdParams& p = f->p;
    hipStream_t s = f->ctx->stream;
    long long stride;
    void* Sin = fwd_S_ptr(f, k, &stride);
    void* Sout = fwd_S_ptr(f, k + 1, &stride);
    int rc = 0;
    hm_fwd* in = embedded_inner(f, &rc);
    if (rc) return rc;
    if ((rc = f->t_sat.begin(s))) return rc;
    if (in) {
        if ((rc = embedded_saturation(f, in, k))) return rc;
        rc = f->t_sat.end(s);
        f->n_sat++;
        return rc;
    }
    if ((rc = extract_fields(f, false))) return rc;  // (a sweep of this plan's own kernels behind an embedded pressure step: the fluxes it reads)
    f->inner_S_step = -1;
    int done = -1;
    bool teams128 = false;
    if (f->sat_variant != 1 && f->sat_variant != 2 && f->sat_variant != 3) {
        // fp64, register/LDS resident: fw in registers, scaled fluxes (sat128r.hip); sat_variant 5: fw image in LDS (sat128.hip)
        // small member shards at 128 x 128 (members x slabs <= CUs: one rank's share of a strong-scaled ensemble): a member as a team of
        // two or four workgroups on CUs of their own (sat128s.hip) -- sat128r's one workgroup per member would leave the other CUs idle
        if (f->sat_variant == 0 || f->sat_variant == 4) {
            if (sat128s_applies(f, k) && (rc = nd128_materialize_fluxes(f))) return rc;
            done = launch_saturation_128s(f, Sin, Sout, stride, k);
        }
        teams128 = done == 0;
        if (done < 0 && f->sat_variant != 5) done = launch_saturation_128r(f, Sin, Sout, stride, k);  // (forms pending fluxes itself: fwd.h)
        if (done < 0 && (rc = nd128_materialize_fluxes(f))) return rc;  // every other sweep reads Vx, Vy
        if (done < 0) done = launch_saturation_128(f, Sin, Sout, stride, k);
        if (done < 0) done = launch_saturation_32s(f, Sin, Sout, stride, k);  // dtype = 32 plans, grids 128 / 256 / 512 wide: slabs (workgroup teams), fw in registers
        if (done < 0 && f->sat_variant != 5) done = launch_saturation_256s(f, Sin, Sout, stride, k);  // fp64, grids 256 wide: slabs of 64 rows (workgroup teams), fw in registers
        if (done < 0) done = launch_saturation_128t(f, Sin, Sout, stride, k);  // fp64, grids of 128 x 128 tiles (workgroup teams)
    }
    if (done > 0) return done;
    if (done == 0 && (p.Nxy > 128 * 128 || teams128) && f->sat_variant != 1 && f->sat_variant != 2 && f->sat_variant != 3) {
        // The team sweeps spin on their neighbours' rows, which needs every workgroup of a team resident at once; the launch is sized for
        // an otherwise idle GPU (one workgroup per CU).  If something else held CUs (another process, a masked device), a team can be
        // partly resident: its workgroups give up after a bounded spin and flag the member HM_MEMBER_SYNC_TIMEOUT.  The single-workgroup
        // tiled sweep needs no co-residency and is bit-identical: it follows every team launch as a GATED launch -- a workgroup per
        // member that returns at once unless its member carries the flag (k_saturation_tiled, only_timed_out).  sat_variant 4 (a test hook)
        // flags every member first, so the retry path runs on all of them every step.
        rc = ensure_generic_sat_scratch(f, false);
        if (rc) return rc;
        if (!f->retried.p) {
            if ((rc = hm_dev_alloc(f->retried, 8))) return rc;
            HM_HIP(hipMemsetAsync(f->retried.p, 0, 8, s));
        }
        if (f->sat_variant == 4) {
            hipLaunchKernelGGL(k_or_status, dim3((p.N + 255) / 256), dim3(256), 0, s, (int*)f->status.p, p.N, (int)HM_MEMBER_SYNC_TIMEOUT);
            f->team_retries++;
        }
        const size_t lds = (size_t)1024 * 8 + (size_t)66 * 258 * f->esz;
        // the float32 slab sweep's record of wet slabs (image written by this step's launch): marked all-wet for a member that is retried
        const int wet_T = (f->dtype == 32 && f->slab_wet.p && (p.Ny == 128 || p.Ny == 256 || p.Ny == 512)) ? p.Nx / (16384 / p.Ny) : 0;
        unsigned char* wet_mark = wet_T > 0 ? (unsigned char*)f->slab_wet.p + (size_t)((k + 1) & 1) * p.N * wet_T : nullptr;
        if (f->dtype == 64) {
            HM_HIP(hipFuncSetAttribute((const void*)k_saturation_tiled<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(k_saturation_tiled<double>, dim3(p.N), dim3(1024), lds, s, f->p, (const double*)Sin, (double*)Sout, stride, (double*)f->prods.p, k, 1, (int*)f->retried.p, (unsigned char*)nullptr, 0);
        } else {
            HM_HIP(hipFuncSetAttribute((const void*)k_saturation_tiled<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(k_saturation_tiled<float>, dim3(p.N), dim3(1024), lds, s, f->p, (const float*)Sin, (float*)Sout, stride, (float*)f->prods.p, k, 1, (int*)f->retried.p, wet_mark, wet_T);
        }
        HM_HIP(hipGetLastError());
    }
    if (done < 0) {
        if ((rc = nd128_materialize_fluxes(f))) return rc;  // (these sweeps read Vx, Vy: fwd.h, lazy face fluxes)
        // sat_variant 1: generic (coefficient arrays + fw image); 2: streaming; 3: tiled; otherwise (no 128 x 128 specialisation
        // applies: other sizes, porosity field, two wells in one patch) tiled from 64 x 64 cells up (67.9 vs 124 ms per launch at
        // 128 x 128, N = 1000), generic below
        const bool tiled = f->sat_variant == 3 || (f->sat_variant != 1 && f->sat_variant != 2 && p.Nxy >= 64 * 64);
        const bool stream = f->sat_variant == 2;
        rc = ensure_generic_sat_scratch(f, !(stream || tiled));
        if (rc) return rc;
        int T = tiled ? 1024 : (p.Nxy >= 4096 ? 1024 : 256);
        size_t lds = tiled ? (size_t)1024 * 8 + (size_t)66 * 258 * f->esz : (size_t)T * 8;
#define SAT(KERN, TT, ...) hipLaunchKernelGGL(KERN<TT>, dim3(p.N), dim3(T), lds, s, f->p, (const TT*)Sin, (TT*)Sout, stride, (TT*)f->prods.p, k, ##__VA_ARGS__)
        if (tiled) {
            if (f->dtype == 64) { HM_HIP(hipFuncSetAttribute((const void*)k_saturation_tiled<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); SAT(k_saturation_tiled, double, 0, (int*)nullptr, (unsigned char*)nullptr, 0); }
            else { HM_HIP(hipFuncSetAttribute((const void*)k_saturation_tiled<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); SAT(k_saturation_tiled, float, 0, (int*)nullptr, (unsigned char*)nullptr, 0); }
        } else if (stream) { if (f->dtype == 64) SAT(k_saturation_stream, double); else SAT(k_saturation_stream, float); }
        else { if (f->dtype == 64) SAT(k_saturation_generic, double); else SAT(k_saturation_generic, float); }
#undef SAT
        HM_HIP(hipGetLastError());
    }
    rc = f->t_sat.end(s);
    f->n_sat++;
    return rc;
}

extern "C" int hm_fwd_run(hm_fwd* f, int first_step, int n_steps) {
    HM_REQUIRE(f, "hm_fwd_run: NULL plan");
    HM_REQUIRE(first_step >= 0 && n_steps >= 0 && first_step + n_steps <= f->p.nTime,
               "hm_fwd_run: steps [%d,%d) outside [0,%d)", first_step, first_step + n_steps, f->p.nTime);
    // the saturation of time index first_step must exist: a history plan holds rows 0..cur, a ping-pong plan
    // (keep_history = 0) only row cur -- its slot of index 0 is overwritten by step 1, so a re-run needs fresh inputs
    if (f->keep_history)
        HM_REQUIRE(first_step <= f->cur, "hm_fwd_run: first_step %d is beyond the last computed time index %d", first_step, f->cur);
    else
        HM_REQUIRE(first_step == f->cur, "hm_fwd_run: a plan without history continues at time index %d (got first_step %d); "
                   "call hm_fwd_set_inputs* to restart", f->cur, first_step);
    HM_HIP(hipSetDevice(f->ctx->device));
    // a run from time index 0 is a new forward_model call (HistoryMatch.py:383-387): it does the work a fresh plan does -- nothing the
    // pressure solve kept from an earlier run's time steps is reused, even where the inputs are the same
    if (first_step == 0) ++f->inputs_gen;
    int rc = 0;
    {   // one-time set-up of the nested-dissection solver (tables, factor / update buffers): before the run's clock starts
        const int pv = f->press_variant;
        if (pv == 0 || pv == 12 || pv == 14)
            if ((rc = prepare_pressure_nd(f)) || (rc = prepare_pressure_nd256(f)) || (rc = prepare_pressure_nd512(f))) return rc;
        hm_fwd* in = embedded_inner(f, &rc);
        if (rc) return rc;
        if (in && ((rc = prepare_pressure_nd(in)) || (rc = prepare_pressure_nd256(in)) || (rc = prepare_pressure_nd512(in)))) return rc;
    }
    if ((rc = f->t_total.begin(f->ctx->stream))) return rc;
    // small grids (the reference's default 20 x 20): the whole run as ONE launch, a wave per member (small.hip; bit-identical to the
    // generic kernels it restates) -- the per-step launches below cost such a grid 0.28 ms a step, nearly all of it the generic pressure
    // kernel's barriers
    rc = n_steps > 0 ? launch_small_forward(f, first_step, n_steps) : -1;
    if (rc > 0) return rc;
    if (rc == 0) {
        f->cur = first_step + n_steps;
        f->n_press += n_steps;  // (one launch: the per-kernel timers stay at zero, the launch counts say how many steps ran)
        f->n_sat += n_steps;
        return f->t_total.end(f->ctx->stream);
    }
    for (int k = first_step; k < first_step + n_steps; ++k) {
        if ((rc = launch_pressure(f, k))) return rc;
        if ((rc = launch_saturation(f, k))) return rc;
        f->cur = k + 1;
    }
    return f->t_total.end(f->ctx->stream);
}

extern "C" int hm_fwd_pressure_only(hm_fwd* f, int k) {
    HM_REQUIRE(f && k >= 0 && k < f->p.nTime, "hm_fwd_pressure_only: bad arguments");
    HM_HIP(hipSetDevice(f->ctx->device));
    return launch_pressure(f, k);
}

extern "C" int hm_fwd_saturation_only(hm_fwd* f, int k) {
    HM_REQUIRE(f && k >= 0 && k < f->p.nTime, "hm_fwd_saturation_only: bad arguments");
    HM_HIP(hipSetDevice(f->ctx->device));
    int rc = launch_saturation(f, k);
    if (!rc) f->cur = k + 1;
    return rc;
}

extern "C" int hm_fwd_sync(hm_fwd* f, hm_stats* st) {
    HM_REQUIRE(f, "hm_fwd_sync: NULL plan");
    HM_HIP(hipSetDevice(f->ctx->device));
    HM_HIP(hipStreamSynchronize(f->ctx->stream));
    if (f->retried.p && !f->dbg_team_rounds) {
        // a team of the one-launch slab sweep gave up waiting (sat32s.hip: its workgroups were not started in grid order, or CUs were held by
        // someone else): from now on this plan launches its teams in rounds of co-resident ones, which rely on neither
        int n = 0;
        HM_HIP(hipMemcpy(&n, f->retried.p, 4, hipMemcpyDeviceToHost));
        if (n > 0) f->dbg_team_rounds = 1;
    }
    if (st) {
        memset(st, 0, sizeof(*st));
        st->ms_total = f->t_total.total_ms();
        st->ms_pressure = f->t_press.total_ms();
        st->ms_saturation = f->t_sat.total_ms();
        st->n_pressure_launches = f->n_press;
        st->n_saturation_launches = f->n_sat;
        st->member_steps = (long long)f->p.N * f->n_sat;
        std::vector<int> nts((size_t)f->p.N * f->p.nTime);
        HM_HIP(hipMemcpy(nts.data(), f->nts.p, nts.size() * 4, hipMemcpyDeviceToHost));
        double sum = 0; long long cnt = 0;
        for (int v : nts) if (v > 0) { sum += v; ++cnt; }
        st->mean_nts = cnt ? sum / cnt : 0.0;
        HM_HIP(hipMemcpy(nts.data(), f->n_cg.p, nts.size() * 4, hipMemcpyDeviceToHost));
        sum = 0; cnt = 0;
        for (int v : nts) if (v > 0) { sum += v; ++cnt; }
        st->mean_n_cg = cnt ? sum / cnt : 0.0;
    }
    f->t_total.reset(); f->t_press.reset(); f->t_sat.reset();
    f->n_press = f->n_sat = 0;
    if (f->inner) {
        f->inner->t_total.reset(); f->inner->t_press.reset(); f->inner->t_sat.reset();
        f->inner->n_press = f->inner->n_sat = 0;
    }
    return 0;
}

extern "C" int hm_fwd_get_outputs(hm_fwd* f, void* wsats_out, void* prods_out, int* status) {
    HM_REQUIRE(f, "hm_fwd_get_outputs: NULL plan");
    HM_HIP(hipSetDevice(f->ctx->device));
    HM_HIP(hipStreamSynchronize(f->ctx->stream));
    const FwdParams& p = f->p;
    if (wsats_out) {
        if (f->keep_history) {
            int rc = hm_d2h_large(f->ctx, wsats_out, f->S.p, (size_t)p.N * (p.nTime + 1) * p.Nxy * f->esz);
            if (rc) return rc;
        } else {
            long long stride;
            void* S = fwd_S_ptr(f, f->cur, &stride);
            int rc = hm_d2h_large(f->ctx, wsats_out, S, (size_t)p.N * p.Nxy * f->esz);
            if (rc) return rc;
        }
    }
    if (prods_out) HM_HIP(hipMemcpy(prods_out, f->prods.p, (size_t)p.N * p.nTime * p.nPrd * f->esz, hipMemcpyDeviceToHost));
    if (status) HM_HIP(hipMemcpy(status, f->status.p, (size_t)p.N * 4, hipMemcpyDeviceToHost));
    return 0;
}

struct FieldRef { void* p; size_t bytes; bool strided; };

static int field_ref(hm_fwd* f, const char* name, FieldRef& r) {
    const FwdParams& p = f->p;
    size_t N = p.N;
    std::string s(name);
    r.strided = false;
    if (s == "P") r = {f->P.p, N * p.Nxy * 8, false};
    else if (s == "Vx") r = {f->Vx.p, N * (p.Nx + 1) * p.Ny * 8, false};
    else if (s == "Vy") r = {f->Vy.p, N * p.Nx * (p.Ny + 1) * 8, false};
    else if (s == "TX") r = {f->TX.p, N * (p.Nx + 1) * p.Ny * 8, false};
    else if (s == "TY") r = {f->TY.p, N * p.Nx * (p.Ny + 1) * 8, false};
    else if (s == "K") r = {f->K.p, N * p.Nxy * 8, false};
    else if (s == "nts") r = {f->nts.p, N * p.nTime * 4, false};
    else if (s == "G") r = {f->G.p, f->G.bytes, false};
    else if (s == "S") { r = {nullptr, N * p.Nxy * f->esz, true}; }
    else { hm_set_error("unknown field '%s'", name); return 2; }
    return 0;
}

extern "C" int hm_fwd_get_field(hm_fwd* f, const char* name, void* out) {
    HM_REQUIRE(f && name && out, "hm_fwd_get_field: NULL argument");
    HM_HIP(hipSetDevice(f->ctx->device));
    int rc = extract_fields(f);
    if (rc) return rc;
    HM_HIP(hipStreamSynchronize(f->ctx->stream));
    FieldRef r;
    rc = field_ref(f, name, r);
    if (rc) return rc;
    if (r.strided) {
        long long stride;
        void* S = fwd_S_ptr(f, f->cur, &stride);
        size_t row = (size_t)f->p.Nxy * f->esz;
        HM_HIP(hipMemcpy2D(out, row, S, (size_t)stride * f->esz, row, f->p.N, hipMemcpyDeviceToHost));
    } else {
        HM_HIP(hipMemcpy(out, r.p, r.bytes, hipMemcpyDeviceToHost));
    }
    return 0;
}

extern "C" int hm_fwd_set_field(hm_fwd* f, const char* name, const void* in) {
    HM_REQUIRE(f && name && in, "hm_fwd_set_field: NULL argument");
    ++f->inputs_gen;
    HM_HIP(hipSetDevice(f->ctx->device));
    int rc = extract_fields(f);  // (what the caller does not set keeps the values of the last step)
    if (rc) return rc;
    HM_HIP(hipStreamSynchronize(f->ctx->stream));
    f->inner_S_step = -1;
    f->inner_V_dirty = true;
    FieldRef r;
    rc = field_ref(f, name, r);
    if (rc) return rc;
    if (r.strided) {
        long long stride;
        void* S = fwd_S_ptr(f, f->cur, &stride);
        size_t row = (size_t)f->p.Nxy * f->esz;
        HM_HIP(hipMemcpy2D(S, (size_t)stride * f->esz, in, row, row, f->p.N, hipMemcpyHostToDevice));
    } else {
        HM_HIP(hipMemcpy(r.p, in, r.bytes, hipMemcpyHostToDevice));
    }
    return 0;
}

extern "C" long long hm_fwd_nd_fallbacks(hm_fwd* f) { return f ? f->nd_fallbacks : 0; }

static long long fwd_counter(hm_fwd* f, int which) {
    if (!f || !f->retried.p) return 0;
    int n[2] = {0, 0};
    if (hipSetDevice(f->ctx->device) != hipSuccess || hipStreamSynchronize(f->ctx->stream) != hipSuccess ||
        hipMemcpy(n, f->retried.p, 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return n[which];
}
extern "C" long long hm_fwd_team_retries(hm_fwd* f) { return fwd_counter(f, 0); }
extern "C" long long hm_fwd_slab_redos(hm_fwd* f) { return fwd_counter(f, 1); }

extern "C" int hm_fwd_set_debug(hm_fwd* f, const char* key, long long value) {
    HM_REQUIRE(f && key, "hm_fwd_set_debug: NULL argument");
    const std::string k(key);
    if (k == "nd_force_fallback") f->dbg_nd_force_fallback = (int)value;
    else if (k == "team_rounds") f->dbg_team_rounds = (int)value;
    else if (k == "embed") f->dbg_embed = (int)value;
    else if (k == "sat_teams") f->dbg_sat_teams = (int)value;
    else if (k == "top_per_level") f->dbg_top_per_level = (int)value;
    else if (k == "top_deal") f->dbg_top_deal = (int)value;
    else if (k == "small_wv") f->dbg_small_wv = (int)value;
    else if (k == "lazy_flux") {
        f->dbg_lazy_flux = (int)value;
        if (f->inner) f->inner->dbg_lazy_flux = (int)value;
    }
    else if (k == "slab_margin") {
        f->dbg_slab_margin = (int)value;
        if (f->inner) f->inner->dbg_slab_margin = (int)value;
    }
    else if (k == "nd_cap") {
        HM_REQUIRE(!f->nd, "hm_fwd_set_debug: \"nd_cap\" must be set before the plan's first run");
        f->dbg_nd_cap = (int)value;
    } else HM_REQUIRE(false, "hm_fwd_set_debug: unknown key \"%s\"", key);
    return 0;
}

extern "C" void* hm_fwd_device_ptr(hm_fwd* f, const char* name) {
    if (!f || !name) return nullptr;
    if (extract_fields(f)) return nullptr;  // (an embedded plan: a snapshot of the inner plan's fields as of the last step)
    std::string s(name);
    // (the saturation may be written through these pointers without the library seeing it: the float32 slab sweep then keeps no record of
    // which slabs are dry from one step to the next -- sat32s.hip)
    if (s == "S") { long long st; f->raw_state_exposed = true; return fwd_S_ptr(f, f->cur, &st); }
    if (s == "prods") return f->prods.p;
    if (s == "S_all") { f->raw_state_exposed = true; return f->S.p; }
    FieldRef r;
    if (field_ref(f, name, r)) return nullptr;
    if (s == "Vx" || s == "Vy") {  // a raw pointer to the fluxes may be read at any later time: from now on every pressure step writes them (fwd.h: lazy fluxes)
        f->dbg_lazy_flux = 0;
        if (f->inner) f->inner->dbg_lazy_flux = 0;
    }
    // the caller may write an INPUT of the pressure step (K, the transmissibilities it may overwrite, the source field) through this
    // pointer at any later time without the library seeing it: results cached across time steps (press_nd.hip) are not kept for this
    // plan any more.  Pure outputs (P, Vx, Vy, nts, status ...) are rewritten by every step before anything reads them: handing
    // those out costs nothing.
    if (s == "K" || s == "Ky" || s == "TX" || s == "TY" || s == "q" || s == "por") {
        ++f->inputs_gen;
        f->raw_field_exposed = true;
        if (s == "q") f->raw_q_exposed = true;
    }
    return r.p;
}

extern "C" int hm_fwd_run_to_host(hm_fwd* f, void* wsats_out, void* prods_out, int* status_per_member, hm_stats* stats) {
    HM_REQUIRE(f, "hm_fwd_run_to_host: NULL plan");
    const FwdParams& p = f->p;
    // A large saturation history leaves the device as it is produced: time index k of every member is copied out on the copy
    // stream while step k (which only reads it) runs on the launch stream.  At config 2 the 5.4 GB of history cost 0.13 s after
    // the last step; 131 MB per step hide under the step's 25 ms.
    const size_t row = (size_t)p.Nxy * f->esz;
    const bool stream_out = f->keep_history && wsats_out && (size_t)p.N * (p.nTime + 1) * row >= ((size_t)256 << 20) && row <= ((size_t)64 << 20);
    const bool trace = getenv("HM_TRACE_RUN_TO_HOST") != nullptr;  // phase times of this call on stderr
    auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_prev = now();
    auto lap = [&](const char* what) {
        if (!trace) return;
        const double t = now();
        fprintf(stderr, "hm_fwd_run_to_host: %-28s %8.2f ms\n", what, t - t_prev);
        t_prev = t;
    };
    int rc = 0;
    if (stream_out) {
        for (int k = 0; k <= p.nTime && !rc; ++k) {
            long long stride;
            const void* Sk = fwd_S_ptr(f, k, &stride);
            rc = hm_copy_mark(f->ctx);  // time index k is complete here; step k, enqueued next, runs beside its copy
            if (!rc && k < p.nTime) rc = hm_fwd_run(f, k, 1);
            if (!rc) rc = hm_d2h_rows(f->ctx, (char*)wsats_out + (size_t)k * row, (size_t)(p.nTime + 1) * row, Sk, (size_t)stride * f->esz, row, (size_t)p.N, true);
            if (trace && (k < 2 || k + 2 > p.nTime)) lap(k < p.nTime ? "step enqueued + a row out" : "last row out");
        }
        lap("steps + rows");
    } else {
        rc = hm_fwd_run(f, 0, p.nTime);
    }
    if (!rc) rc = hm_fwd_sync(f, stats);
    lap("sync + statistics");
    if (!rc) rc = hm_fwd_get_outputs(f, stream_out ? nullptr : wsats_out, prods_out, status_per_member);
    lap("outputs");
    return rc;
}

extern "C" int hm_forward_batched(hm_ctx* ctx, int N, int Nx, int Ny, double Lx, double Ly, const void* perm,
                                  int perm_is_transformed, const void* wsat0, int nInj, const int* inj_ind,
                                  const double* inj_rates, int inj_rate_cols, int nPrd, const int* prd_ind,
                                  const double* prd_rates, int prd_rate_cols, double dt, int nTime, double vw,
                                  double vo, double swc, double sor, const double* porosity, int dtype,
                                  int return_history, void* wsats_out, void* prods_out, int* status_per_member,
                                  hm_stats* stats) {
    hm_fwd* f = nullptr;
    int rc = hm_fwd_create(ctx, N, Nx, Ny, Lx, Ly, nInj, inj_ind, inj_rates, inj_rate_cols, nPrd, prd_ind, prd_rates,
                           prd_rate_cols, dt, nTime, vw, vo, swc, sor, porosity, dtype, return_history, &f);
    if (rc) return rc;
    rc = hm_fwd_set_inputs(f, perm, perm_is_transformed, wsat0);
    if (!rc) rc = hm_fwd_run_to_host(f, wsats_out, prods_out, status_per_member, stats);
    hm_fwd_destroy(f);
    return rc;
}
