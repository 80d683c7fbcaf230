// press_pcg.hip -- pressure step for grids too large for the direct block solver (Ny > 128): the same TPFA system
// (SURVEY.md A.3), solved per member by Jacobi-preconditioned conjugate gradients in fp64, warm-started from the
// previous time step's pressure.
//
// One workgroup (1024 threads) per member; the vectors x (= P), r, p, Ap live in HBM/L2 (512^2: 2 MB each), every
// pass is a coalesced sweep over the member's cells, reductions are fixed-order LDS trees (bit-reproducible).  Per
// iteration: SpMV + p.Ap | x, r update + r.z, r.r | p update = ~18 cell-vector passes (SURVEY.md 8d counts 13 for a
// minimal fused iteration; the diagonal is re-derived from the face transmissibilities instead of being stored).
// The matrix is never formed: row j is  dg_j p_j - x1 p_{j-Ny} - x2 p_{j+Ny} - y1 p_{j-1} - y2 p_{j+1}  with
// x1, x2, y1, y2 the face transmissibilities (zero on the boundary) and dg = y1 + y2 + x1 + x2 (+ the SPD pin on cell 0),
// exactly the entries the reference assembles.  Stops at ||r|| <= rtol ||q||; non-convergence within max_iter is
// reported per member (HM_MEMBER_NO_CONVERGENCE).  Any fp64 solver of this system differs from SuperLU by solver noise
// (tolerance model): the tests compare within that noise, not bitwise.
#include "fwd_dev.h"

namespace {

constexpr int PT = 1024;

// sum of two values over the workgroup, fixed order
__device__ __forceinline__ void block_sum2(double& a, double& b, double* red, int tid) {
    red[tid] = a;
    red[PT + tid] = b;
    __syncthreads();
    for (int s = PT / 2; s > 0; s >>= 1) {
        if (tid < s) {
            red[tid] += red[tid + s];
            red[PT + tid] += red[PT + tid + s];
        }
        __syncthreads();
    }
    a = red[0];
    b = red[PT];
    __syncthreads();
}

template <typename TS>
__global__ __launch_bounds__(PT) void k_pressure_pcg(FwdParams p, const TS* __restrict__ S_base, long long S_stride, int k) {
    __shared__ double red[2 * PT];
    const int m = blockIdx.x, tid = threadIdx.x;
    const int Nx = p.Nx, Ny = p.Ny, Nxy = p.Nxy;
    const TS* S = S_base + (long long)m * S_stride;
    const double* Km = p.K + (long long)m * Nxy;
    const double* Kym = p.Ky ? p.Ky + (long long)m * Nxy : Km;
    double* TX = p.TX + (long long)m * (Nx + 1) * Ny;
    double* TY = p.TY + (long long)m * Nx * (Ny + 1);
    double* x = p.P + (long long)m * Nxy;
    double* r = p.cg_r + (long long)m * Nxy;
    double* pv = p.cg_p + (long long)m * Nxy;
    double* Ap = p.yv + (long long)m * Nxy;  // also the scratch of the assembly
    double* Vx = p.Vx + (long long)m * (Nx + 1) * Ny;
    double* Vy = p.Vy + (long long)m * Nx * (Ny + 1);
    const double* q = p.q + (long long)m * p.q_mstride + (long long)(p.q_cols > 1 ? k : 0) * Nxy;
    const double pin = Km[0] + Kym[0];  // SPD pin: A[0,0] += Kx[0,0]+Ky[0,0]

    assemble_transmissibilities<TS>(p, S, Km, Kym, Ap /* scratch for L */, TX, TY, tid, PT);
    __syncthreads();

    auto diag = [&](int j, int ix, int iy) {
        const double y1 = TY[ix * (Ny + 1) + iy], y2 = TY[ix * (Ny + 1) + iy + 1];
        const double x1 = TX[ix * Ny + iy], x2 = TX[(ix + 1) * Ny + iy];
        double dg = y1 + y2 + x1 + x2;
        if (j == 0) dg += pin;
        return dg == 0.0 ? 1.0 : dg;  // (a cell of zero permeability -- the padding of an embedded grid, forward.hip: residual 0, correction 0)
    };
    // y = A v at cell j
    auto row = [&](const double* __restrict__ v, int j, int ix, int iy) {
        const double y1 = TY[ix * (Ny + 1) + iy], y2 = TY[ix * (Ny + 1) + iy + 1];
        const double x1 = TX[ix * Ny + iy], x2 = TX[(ix + 1) * Ny + iy];
        double dg = y1 + y2 + x1 + x2;
        if (j == 0) dg += pin;
        double s = dg * v[j];
        if (ix > 0) s -= x1 * v[j - Ny];
        if (ix + 1 < Nx) s -= x2 * v[j + Ny];
        if (iy > 0) s -= y1 * v[j - 1];
        if (iy + 1 < Ny) s -= y2 * v[j + 1];
        return s;
    };

    // r = q - A x0 (x0 = previous pressure, zero before the first step); z = r / dg; p = z
    double rz = 0.0, bb = 0.0;
    for (int j = tid; j < Nxy; j += PT) {
        const int ix = j / Ny, iy = j - ix * Ny;
        const double b = q[j];
        const double rj = b - row(x, j, ix, iy);
        const double zj = rj / diag(j, ix, iy);
        r[j] = rj;
        pv[j] = zj;
        rz += rj * zj;
        bb += b * b;
    }
    block_sum2(rz, bb, red, tid);
    const double stop2 = p.cg_rtol * p.cg_rtol * bb;
    int it = 0, converged = 0;
    {
        double rr = 0.0, dummy = 0.0;
        for (int j = tid; j < Nxy; j += PT) rr += r[j] * r[j];
        block_sum2(rr, dummy, red, tid);
        converged = rr <= stop2;
    }
    while (!converged && it < p.cg_max_iter) {
        double pAp = 0.0, dummy = 0.0;
        for (int j = tid; j < Nxy; j += PT) {
            const int ix = j / Ny, iy = j - ix * Ny;
            const double a = row(pv, j, ix, iy);
            Ap[j] = a;
            pAp += pv[j] * a;
        }
        block_sum2(pAp, dummy, red, tid);
        if (!(pAp > 0.0)) break;  // not SPD (K <= 0, NaN): reported below
        const double alpha = rz / pAp;
        double rz_new = 0.0, rr = 0.0;
        for (int j = tid; j < Nxy; j += PT) {
            const int ix = j / Ny, iy = j - ix * Ny;
            x[j] += alpha * pv[j];
            const double rj = r[j] - alpha * Ap[j];
            r[j] = rj;
            rz_new += rj * (rj / diag(j, ix, iy));
            rr += rj * rj;
        }
        block_sum2(rz_new, rr, red, tid);
        ++it;
        if (rr <= stop2) {
            converged = 1;
            break;
        }
        const double beta = rz_new / rz;
        rz = rz_new;
        for (int j = tid; j < Nxy; j += PT) {
            const int ix = j / Ny, iy = j - ix * Ny;
            pv[j] = r[j] / diag(j, ix, iy) + beta * pv[j];
        }
        __syncthreads();
    }
    __syncthreads();
    face_fluxes(p, x, TX, TY, Vx, Vy, tid, PT);
    if (tid == 0) {
        p.n_cg[(long long)m * p.nTime + k] = it;
        if (!converged) atomicOr(&p.status[m], isfinite(rz) ? HM_MEMBER_NO_CONVERGENCE : HM_MEMBER_BAD_PIVOT);
    }
}


// ------------------------------------------------------------------------------------------------------------
// Two-level preconditioned CG (grids with Ny = 128 c, Nx = c Nx_c), P = piecewise constant prolongation over c x c
// aggregates.  The preconditioner is the SYMMETRIC MULTIPLICATIVE two-grid cycle with one damped-Jacobi sweep on either
// side of the coarse correction (omega = 0.8 < 2 / lambda_max(D^-1 A), so the cycle is symmetric positive definite):
//     z1 = w D^-1 r;   z2 = z1 + P A_c^-1 P^T (r - A z1);   z = z2 + w D^-1 (r - A z2)
// (press_variant 11 keeps the first version, the additive M^-1 = D^-1 + P A_c^-1 P^T: 59 / 98 iterations at 256^2 / 512^2
// against 27 / 50 for the cycle on the same systems, NumPy/SciPy prototype and device alike; the cycle costs two more
// fine-grid operator passes per iteration but halves the coarse solves, which are the larger part of an iteration.)  Aggregating a TPFA system gives a TPFA system: the coarse face
// transmissibility is the sum of the fine ones across the shared aggregate boundary, interior faces cancel, the SPD pin
// stays on (coarse) cell 0.  With c = Ny / 128 the coarse system has Ny_c = 128 and is factored ONCE per time step by the
// direct block solver (press128s.hip, FACTOR mode); every CG iteration then costs one coarse solve (two substitution
// passes over the stored factor).  Measured on the same systems (NumPy/SciPy prototype): 59 iterations at 256^2, 98 at
// 512^2, against 2 000 / 4 500 with the Jacobi preconditioner.
// One iteration of the cycle = k_tg_spmv, k_tg_update, k_tg_restrict -> k_coarse_solve -> k_tg_correct, k_tg_postsmooth,
// k_tg_direction (below: G workgroups per member); of the additive form = k_tl_iter -> k_coarse_solve -> k_tl_dir; per-member
// scalars and convergence flags live in device memory, converged members drop out of every kernel.
// cgs[m]: [0] r.z  [1] stop^2  [2] iterations  [3] ||q||^2
// ------------------------------------------------------------------------------------------------------------
struct TlArgs {
    double* TXc;   // N x (Nxc+1) x 128
    double* TYc;   // N x Nxc x 129
    double* pin;   // N
    double* rc;    // N x Nxc x 128   restricted residual
    double* yc;    // N x Nxc x 128   coarse correction
    double* cgs;   // N x 4
    int* done;     // N: 0 running, 1 converged, 2 breakdown
    int* ndone;    // 1: members no longer running
    int c, Nxc;
    double* z1;    // N x Nxy   pre-smoothed iterate of the cycle (then z2, in place)
    double* dinv;  // N x Nxy   1 / diagonal
    double omega;  // Jacobi damping of the cycle; 0 = additive preconditioner (no smoothing passes)
};

__device__ __forceinline__ double block_sum1(double a, double* red, int tid) {
    double dummy = 0.0;
    block_sum2(a, dummy, red, tid);
    return a;
}

// fine operator pieces shared by the kernels
struct FineOp {
    const double *TX, *TY;
    int Nx, Ny;
    double pin;
    __device__ __forceinline__ double diag(int j, int ix, int iy) const {
        const double y1 = TY[ix * (Ny + 1) + iy], y2 = TY[ix * (Ny + 1) + iy + 1];
        const double x1 = TX[ix * Ny + iy], x2 = TX[(ix + 1) * Ny + iy];
        double dg = y1 + y2 + x1 + x2;
        if (j == 0) dg += pin;
        return dg;
    }
    __device__ __forceinline__ double row(const double* __restrict__ v, int j, int ix, int iy) const {
        const double y1 = TY[ix * (Ny + 1) + iy], y2 = TY[ix * (Ny + 1) + iy + 1];
        const double x1 = TX[ix * Ny + iy], x2 = TX[(ix + 1) * Ny + iy];
        double dg = y1 + y2 + x1 + x2;
        if (j == 0) dg += pin;
        double s = dg * v[j];
        if (ix > 0) s -= x1 * v[j - Ny];
        if (ix + 1 < Nx) s -= x2 * v[j + Ny];
        if (iy > 0) s -= y1 * v[j - 1];
        if (iy + 1 < Ny) s -= y2 * v[j + 1];
        return s;
    }
};

// rc = P^T r: sum of r over the c x c fine cells of every aggregate
__device__ __forceinline__ void restrict_residual(const double* __restrict__ r, double* __restrict__ rc, int Ny, int c, int Nxc, int tid) {
    for (int J = tid; J < Nxc * 128; J += PT) {
        const int I = J >> 7, Jy = J & 127;
        double s = 0.0;
        for (int a = 0; a < c; ++a)
            for (int b = 0; b < c; ++b) s += r[(I * c + a) * Ny + Jy * c + b];
        rc[J] = s;
    }
}

// Pre-smoothing half of the cycle: z1 = w D^-1 r (stored), then rc = P^T (r - A z1).  Ends with the block's writes visible.
__device__ __forceinline__ void presmooth_and_restrict(const FineOp& A, const double* __restrict__ r, const double* __restrict__ dinv,
                                                       double* __restrict__ z1, double* __restrict__ rc, double omega, int c, int Nxc, int tid) {
    const int Ny = A.Ny, Nxy = A.Nx * A.Ny;
    for (int j = tid; j < Nxy; j += PT) z1[j] = omega * (r[j] * dinv[j]);
    __syncthreads();
    for (int J = tid; J < Nxc * 128; J += PT) {
        const int I = J >> 7, Jy = J & 127;
        double s = 0.0;
        for (int a = 0; a < c; ++a)
            for (int b = 0; b < c; ++b) {
                const int ix = I * c + a, iy = Jy * c + b, j = ix * Ny + iy;
                s += r[j] - A.row(z1, j, ix, iy);
            }
        rc[J] = s;
    }
}

template <typename TS>
__global__ __launch_bounds__(PT) void k_tl_setup(FwdParams p, TlArgs t, const TS* __restrict__ S_base, long long S_stride, int k) {
    __shared__ double red[2 * PT];
    const int m = blockIdx.x, tid = threadIdx.x;
    const int Nx = p.Nx, Ny = p.Ny, Nxy = p.Nxy, c = t.c, Nxc = t.Nxc;
    const TS* S = S_base + (long long)m * S_stride;
    const double* Km = p.K + (long long)m * Nxy;
    const double* Kym = p.Ky ? p.Ky + (long long)m * Nxy : Km;
    double* TX = p.TX + (long long)m * (Nx + 1) * Ny;
    double* TY = p.TY + (long long)m * Nx * (Ny + 1);
    double* x = p.P + (long long)m * Nxy;
    double* r = p.cg_r + (long long)m * Nxy;
    double* Ap = p.yv + (long long)m * Nxy;
    const double* q = p.q + (long long)m * p.q_mstride + (long long)(p.q_cols > 1 ? k : 0) * Nxy;
    assemble_transmissibilities<TS>(p, S, Km, Kym, Ap /* scratch for L */, TX, TY, tid, PT);
    __syncthreads();
    // coarse transmissibilities = sums of the fine ones across the aggregate boundaries
    double* TXc = t.TXc + (long long)m * (Nxc + 1) * 128;
    double* TYc = t.TYc + (long long)m * Nxc * 129;
    for (int f = tid; f < (Nxc + 1) * 128; f += PT) {
        const int I = f >> 7, J = f & 127;
        double s = 0.0;
        if (I > 0 && I < Nxc)
            for (int b = 0; b < c; ++b) s += TX[(I * c) * Ny + J * c + b];
        TXc[f] = s;
    }
    for (int f = tid; f < Nxc * 129; f += PT) {
        const int I = f / 129, J = f - I * 129;
        double s = 0.0;
        if (J > 0 && J < 128)
            for (int a = 0; a < c; ++a) s += TY[(I * c + a) * (Ny + 1) + J * c];
        TYc[f] = s;
    }
    const FineOp A{TX, TY, Nx, Ny, Km[0] + Kym[0]};
    if (tid == 0) t.pin[m] = A.pin;
    double* dinv = t.dinv + (long long)m * Nxy;
    double rr = 0.0, bb = 0.0;
    for (int j = tid; j < Nxy; j += PT) {
        const int ix = j / Ny, iy = j - ix * Ny;
        const double b = q[j];
        const double rj = b - A.row(x, j, ix, iy);
        r[j] = rj;
        if (t.omega != 0.0) dinv[j] = 1.0 / A.diag(j, ix, iy);
        rr += rj * rj;
        bb += b * b;
    }
    block_sum2(rr, bb, red, tid);  // (its barriers also publish r)
    if (t.omega != 0.0) presmooth_and_restrict(A, r, dinv, t.z1 + (long long)m * Nxy, t.rc + (long long)m * Nxc * 128, t.omega, c, Nxc, tid);
    else restrict_residual(r, t.rc + (long long)m * Nxc * 128, Ny, c, Nxc, tid);
    if (tid == 0) {
        const double stop2 = p.cg_rtol * p.cg_rtol * bb;
        double* cg = t.cgs + 4 * m;
        cg[0] = 0.0; cg[1] = stop2; cg[2] = 0.0; cg[3] = bb;
        const int dn = rr <= stop2 ? 1 : 0;
        t.done[m] = dn;
        if (dn) atomicAdd(t.ndone, 1);
    }
}

// additive preconditioner (press_variant 11), one workgroup per member:  z = D^-1 r + P yc;
// first: p = z, rz = r.z;  else beta = (r.z)_new / (r.z)_old, p = z + beta p
__global__ __launch_bounds__(PT) void k_tl_dir(FwdParams p, TlArgs t, int first) {
    __shared__ double red[2 * PT];
    const int m = blockIdx.x, tid = threadIdx.x;
    if (t.done[m]) return;
    const int Nx = p.Nx, Ny = p.Ny, Nxy = p.Nxy, c = t.c, Nxc = t.Nxc;
    const double* TX = p.TX + (long long)m * (Nx + 1) * Ny;
    const double* TY = p.TY + (long long)m * Nx * (Ny + 1);
    const double* r = p.cg_r + (long long)m * Nxy;
    double* pv = p.cg_p + (long long)m * Nxy;
    double* z = p.yv + (long long)m * Nxy;
    const double* yc = t.yc + (long long)m * Nxc * 128;
    const FineOp A{TX, TY, Nx, Ny, t.pin[m]};
    double rz = 0.0;
    for (int j = tid; j < Nxy; j += PT) {
        const int ix = j / Ny, iy = j - ix * Ny;
        const double zj = r[j] / A.diag(j, ix, iy) + yc[(ix / c) * 128 + iy / c];
        z[j] = zj;
        rz += r[j] * zj;
    }
    rz = block_sum1(rz, red, tid);
    double* cg = t.cgs + 4 * m;
    const double beta = first ? 0.0 : rz / cg[0];
    for (int j = tid; j < Nxy; j += PT) pv[j] = first ? z[j] : z[j] + beta * pv[j];
    __syncthreads();
    if (tid == 0) cg[0] = rz;
}

// Ap = A p, alpha = rz / p.Ap, x += alpha p, r -= alpha Ap, convergence test, rc = P^T r
__global__ __launch_bounds__(PT) void k_tl_iter(FwdParams p, TlArgs t) {
    __shared__ double red[2 * PT];
    const int m = blockIdx.x, tid = threadIdx.x;
    if (t.done[m]) return;
    const int Nx = p.Nx, Ny = p.Ny, Nxy = p.Nxy;
    const double* TX = p.TX + (long long)m * (Nx + 1) * Ny;
    const double* TY = p.TY + (long long)m * Nx * (Ny + 1);
    double* x = p.P + (long long)m * Nxy;
    double* r = p.cg_r + (long long)m * Nxy;
    const double* pv = p.cg_p + (long long)m * Nxy;
    double* Ap = p.yv + (long long)m * Nxy;
    const FineOp A{TX, TY, Nx, Ny, t.pin[m]};
    double pAp = 0.0;
    for (int j = tid; j < Nxy; j += PT) {
        const int ix = j / Ny, iy = j - ix * Ny;
        const double a = A.row(pv, j, ix, iy);
        Ap[j] = a;
        pAp += pv[j] * a;
    }
    pAp = block_sum1(pAp, red, tid);
    double* cg = t.cgs + 4 * m;
    if (!(pAp > 0.0)) {  // not SPD (K <= 0, NaN)
        if (tid == 0) { t.done[m] = 2; atomicAdd(t.ndone, 1); }
        return;
    }
    const double alpha = cg[0] / pAp;
    double rr = 0.0;
    for (int j = tid; j < Nxy; j += PT) {
        x[j] += alpha * pv[j];
        const double rj = r[j] - alpha * Ap[j];
        r[j] = rj;
        rr += rj * rj;
    }
    rr = block_sum1(rr, red, tid);
    if (tid == 0) cg[2] += 1.0;
    if (rr <= cg[1]) {
        if (tid == 0) { t.done[m] = 1; atomicAdd(t.ndone, 1); }
        return;
    }
    restrict_residual(r, t.rc + (long long)m * t.Nxc * 128, Ny, t.c, t.Nxc, tid);
}

// ---- the same iteration with G workgroups per member (blockIdx.y = g: contiguous cell / aggregate ranges) ------------------
// With fewer members than CUs (config 5's shard: 125 members of 512 x 512 cells) one workgroup per member leaves half the
// chip idle and every pass latency-bound on a single CU.  Here a member's passes are split over G workgroups; every point
// where a pass needs ALL of the previous one (a dot product, or neighbour values across the range boundary) is a kernel
// boundary, dot products are two-stage with a fixed order (per-workgroup LDS tree, then the G partial sums in order g = 0..G-1
// by every consumer) -> bit-reproducible.  parts[kind][parity][m][g]: 0 = p.Ap, 1 = r.r, 2 = r.z; the parity alternates per
// iteration so r.z of the previous iteration stays readable while the new one is written.
struct TgArgs {
    double* parts;
    int G, par;
};
__device__ __forceinline__ double* part_slot(const TgArgs& a, int N, int kind, int par, int m, int g) {
    return a.parts + (((size_t)(kind * 2 + par) * N + m) * a.G + g);
}
__device__ __forceinline__ double sum_parts(const TgArgs& a, int N, int kind, int par, int m) {
    const double* q = part_slot(a, N, kind, par, m, 0);
    double s = 0.0;
    for (int g = 0; g < a.G; ++g) s += q[g];
    return s;
}
#define TG_PROLOGUE                                                                                   \
    const int m = blockIdx.x, g = blockIdx.y, tid = threadIdx.x;                                       \
    if (t.done[m]) return;                                                                             \
    const int Nx = p.Nx, Ny = p.Ny, Nxy = p.Nxy;                                                       \
    const int j0 = (int)((long long)Nxy * g / a.G), j1 = (int)((long long)Nxy * (g + 1) / a.G);        \
    const double* TX = p.TX + (long long)m * (Nx + 1) * Ny;                                            \
    const double* TY = p.TY + (long long)m * Nx * (Ny + 1);                                            \
    const FineOp A{TX, TY, Nx, Ny, t.pin[m]};                                                          \
    (void)j0; (void)j1; (void)A; (void)tid

// a: Ap = A p, partial p.Ap
__global__ __launch_bounds__(PT) void k_tg_spmv(FwdParams p, TlArgs t, TgArgs a) {
    __shared__ double red[2 * PT];
    TG_PROLOGUE;
    const double* pv = p.cg_p + (long long)m * Nxy;
    double* Ap = p.yv + (long long)m * Nxy;
    double pAp = 0.0;
    for (int j = j0 + tid; j < j1; j += PT) {
        const int ix = j / Ny, iy = j - ix * Ny;
        const double v = A.row(pv, j, ix, iy);
        Ap[j] = v;
        pAp += pv[j] * v;
    }
    pAp = block_sum1(pAp, red, tid);
    if (tid == 0) *part_slot(a, p.N, 0, a.par, m, g) = pAp;
}

// b: alpha = (r.z)_old / p.Ap, x += alpha p, r -= alpha Ap, partial r.r, z1 = w D^-1 r
__global__ __launch_bounds__(PT) void k_tg_update(FwdParams p, TlArgs t, TgArgs a) {
    __shared__ double red[2 * PT];
    TG_PROLOGUE;
    const double pAp = sum_parts(a, p.N, 0, a.par, m);
    if (!(pAp > 0.0)) {  // not SPD (K <= 0, NaN): every workgroup of the member sees the same value
        if (g == 0 && tid == 0) { t.done[m] = 2; atomicAdd(t.ndone, 1); }
        return;
    }
    const double alpha = sum_parts(a, p.N, 2, a.par ^ 1, m) / pAp;
    double* x = p.P + (long long)m * Nxy;
    double* r = p.cg_r + (long long)m * Nxy;
    const double* pv = p.cg_p + (long long)m * Nxy;
    const double* Ap = p.yv + (long long)m * Nxy;
    const double* dinv = t.dinv + (long long)m * Nxy;
    double* z1 = t.z1 + (long long)m * Nxy;
    double rr = 0.0;
    for (int j = j0 + tid; j < j1; j += PT) {
        x[j] += alpha * pv[j];
        const double rj = r[j] - alpha * Ap[j];
        r[j] = rj;
        rr += rj * rj;
        z1[j] = t.omega * (rj * dinv[j]);
    }
    rr = block_sum1(rr, red, tid);
    if (tid == 0) *part_slot(a, p.N, 1, a.par, m, g) = rr;
}

// c: convergence test, rc = P^T (r - A z1)
__global__ __launch_bounds__(PT) void k_tg_restrict(FwdParams p, TlArgs t, TgArgs a) {
    __shared__ double red[PT];
    TG_PROLOGUE;
    const double rr = sum_parts(a, p.N, 1, a.par, m);
    double* cg = t.cgs + 4 * m;
    if (g == 0 && tid == 0) cg[2] += 1.0;
    if (rr <= cg[1]) {
        if (g == 0 && tid == 0) { t.done[m] = 1; atomicAdd(t.ndone, 1); }
        return;
    }
    const double* r = p.cg_r + (long long)m * Nxy;
    const double* z1 = t.z1 + (long long)m * Nxy;
    double* rc = t.rc + (long long)m * t.Nxc * 128;
    const int c = t.c;
    const int I0 = (int)((long long)t.Nxc * g / a.G), I1 = (int)((long long)t.Nxc * (g + 1) / a.G);
    if (Ny <= PT) {
        // cell-parallel (coalesced): a thread sums the c cells of its column iy inside coarse row I, the c columns of an
        // aggregate are then added through LDS in a fixed order
        const int rpp = PT / Ny;                       // coarse rows per pass
        const int lr = tid / Ny, iy = tid - lr * Ny;
        for (int Ib = I0; Ib < I1; Ib += rpp) {
            const int I = Ib + lr;
            const bool valid = lr < rpp && I < I1;
            double sacc = 0.0;
            if (valid)
                for (int u = 0; u < c; ++u) {
                    const int ix = I * c + u, j = ix * Ny + iy;
                    sacc += r[j] - A.row(z1, j, ix, iy);
                }
            red[tid] = sacc;
            __syncthreads();
            if (valid && iy % c == 0) {
                double tot = 0.0;
                for (int v = 0; v < c; ++v) tot += red[tid + v];
                rc[I * 128 + iy / c] = tot;
            }
            __syncthreads();
        }
    } else {
        for (int J = I0 * 128 + tid; J < I1 * 128; J += PT) {
            const int I = J >> 7, Jy = J & 127;
            double sacc = 0.0;
            for (int u = 0; u < c; ++u)
                for (int v = 0; v < c; ++v) {
                    const int ix = I * c + u, iy = Jy * c + v, j = ix * Ny + iy;
                    sacc += r[j] - A.row(z1, j, ix, iy);
                }
            rc[J] = sacc;
        }
    }
}

// d: z2 = z1 + P yc (in place)
__global__ __launch_bounds__(PT) void k_tg_correct(FwdParams p, TlArgs t, TgArgs a) {
    TG_PROLOGUE;
    double* z2 = t.z1 + (long long)m * Nxy;
    const double* yc = t.yc + (long long)m * t.Nxc * 128;
    const int c = t.c;
    for (int j = j0 + tid; j < j1; j += PT) {
        const int ix = j / Ny, iy = j - ix * Ny;
        z2[j] += yc[(ix / c) * 128 + iy / c];
    }
}

// e: z = z2 + w D^-1 (r - A z2), partial r.z
__global__ __launch_bounds__(PT) void k_tg_postsmooth(FwdParams p, TlArgs t, TgArgs a) {
    __shared__ double red[2 * PT];
    TG_PROLOGUE;
    const double* r = p.cg_r + (long long)m * Nxy;
    const double* z2 = t.z1 + (long long)m * Nxy;
    const double* dinv = t.dinv + (long long)m * Nxy;
    double* z = p.yv + (long long)m * Nxy;
    double rz = 0.0;
    for (int j = j0 + tid; j < j1; j += PT) {
        const int ix = j / Ny, iy = j - ix * Ny;
        const double zj = z2[j] + t.omega * ((r[j] - A.row(z2, j, ix, iy)) * dinv[j]);
        z[j] = zj;
        rz += r[j] * zj;
    }
    rz = block_sum1(rz, red, tid);
    if (tid == 0) *part_slot(a, p.N, 2, a.par, m, g) = rz;
}

// f: beta = (r.z)_new / (r.z)_old, p = z + beta p   (first: p = z)
__global__ __launch_bounds__(PT) void k_tg_direction(FwdParams p, TlArgs t, TgArgs a, int first) {
    TG_PROLOGUE;
    const double beta = first ? 0.0 : sum_parts(a, p.N, 2, a.par, m) / sum_parts(a, p.N, 2, a.par ^ 1, m);
    double* pv = p.cg_p + (long long)m * Nxy;
    const double* z = p.yv + (long long)m * Nxy;
    for (int j = j0 + tid; j < j1; j += PT) pv[j] = first ? z[j] : z[j] + beta * pv[j];
}
#undef TG_PROLOGUE

__global__ __launch_bounds__(PT) void k_tl_final(FwdParams p, TlArgs t, int k) {
    const int m = blockIdx.x, tid = threadIdx.x;
    const int Nx = p.Nx, Ny = p.Ny, Nxy = p.Nxy;
    face_fluxes(p, p.P + (long long)m * Nxy, p.TX + (long long)m * (Nx + 1) * Ny, p.TY + (long long)m * Nx * (Ny + 1),
                p.Vx + (long long)m * (Nx + 1) * Ny, p.Vy + (long long)m * Nx * (Ny + 1), tid, PT);
    if (tid == 0) {
        p.n_cg[(long long)m * p.nTime + k] = (int)t.cgs[4 * m + 2];
        if (t.done[m] != 1) atomicOr(&p.status[m], t.done[m] == 2 ? HM_MEMBER_BAD_PIVOT : HM_MEMBER_NO_CONVERGENCE);
    }
}

}  // namespace

int launch_pressure_pcg(hm_fwd* f, const void* S, long long S_stride, int k) {
    const FwdParams& p = f->p;
    hipStream_t s = f->ctx->stream;
    if (f->dtype == 64) hipLaunchKernelGGL(k_pressure_pcg<double>, dim3(p.N), dim3(PT), 0, s, p, (const double*)S, S_stride, k);
    else hipLaunchKernelGGL(k_pressure_pcg<float>, dim3(p.N), dim3(PT), 0, s, p, (const float*)S, S_stride, k);
    HM_HIP(hipGetLastError());
    return 0;
}

// ---- two-level driver -------------------------------------------------------------------------------------------
int launch_coarse_factor_128(hipStream_t s, const FwdParams& pc);                                                // press128s.hip
int launch_coarse_solve_128(hipStream_t s, const FwdParams& pc, const double* b, double* x, const int* skip);

bool pressure_two_level_applies(const FwdParams& p) {
    if (p.Ny <= 128 || p.Ny % 128 != 0) return false;
    const int c = p.Ny / 128;
    return p.Nx % c == 0 && p.Nx / c >= 2;
}

// Synchronous within the time step: the host polls the number of converged members every few iterations.
int launch_pressure_two_level(hm_fwd* f, const void* S, long long S_stride, int k) {
    const FwdParams& p = f->p;
    hipStream_t s = f->ctx->stream;
    const int c = p.Ny / 128, Nxc = p.Nx / c;
    const size_t n = p.N, nc = (size_t)Nxc * 128;
    if (!f->tl_TXc.p || f->tl_n < p.N) {  // (first use, or sized for a smaller member block: the fall-back of the direct solver runs single members)
        DevBuf* bufs[] = {&f->tl_TXc, &f->tl_TYc, &f->tl_pin, &f->tl_rc, &f->tl_yc, &f->tl_yv, &f->tl_G, &f->tl_cgs, &f->tl_done, &f->tl_ndone, &f->tl_z1, &f->tl_dinv, &f->tl_parts};
        if (f->tl_TXc.p) HM_HIP(hipStreamSynchronize(s));
        for (DevBuf* b : bufs) hm_dev_free(*b);
        f->tl_n = p.N;
        int rc = 0;
#define A_(buf, bytes) if (!rc) rc = hm_dev_alloc(f->buf, (bytes))
        A_(tl_TXc, n * (Nxc + 1) * 128 * 8); A_(tl_TYc, n * Nxc * 129 * 8); A_(tl_pin, n * 8); A_(tl_rc, n * nc * 8); A_(tl_yc, n * nc * 8);
        A_(tl_yv, n * nc * 8); A_(tl_G, n * nc * 128 * 8); A_(tl_cgs, n * 4 * 8); A_(tl_done, n * 4); A_(tl_ndone, 16);
        A_(tl_z1, n * (size_t)p.Nxy * 8); A_(tl_dinv, n * (size_t)p.Nxy * 8); A_(tl_parts, (size_t)3 * 2 * n * 8 * 8);
#undef A_
        if (rc) return rc;
    }
    TlArgs t{(double*)f->tl_TXc.p, (double*)f->tl_TYc.p, (double*)f->tl_pin.p, (double*)f->tl_rc.p, (double*)f->tl_yc.p,
             (double*)f->tl_cgs.p, (int*)f->tl_done.p, (int*)f->tl_ndone.p, c, Nxc,
             (double*)f->tl_z1.p, (double*)f->tl_dinv.p, f->press_variant == 11 ? 0.0 : 0.8};
    FwdParams pc = p;  // the coarse system as the direct solver sees it
    pc.Nx = Nxc; pc.Ny = 128; pc.Nxy = (int)nc;
    pc.TX = t.TXc; pc.TY = t.TYc; pc.G = (double*)f->tl_G.p; pc.yv = (double*)f->tl_yv.p; pc.pin = t.pin;
    HM_HIP(hipMemsetAsync(t.ndone, 0, 4, s));
    if (f->dtype == 64) hipLaunchKernelGGL(k_tl_setup<double>, dim3(p.N), dim3(PT), 0, s, p, t, (const double*)S, S_stride, k);
    else hipLaunchKernelGGL(k_tl_setup<float>, dim3(p.N), dim3(PT), 0, s, p, t, (const float*)S, S_stride, k);
    HM_HIP(hipGetLastError());
    int rc = launch_coarse_factor_128(s, pc);
    if (rc) return rc > 0 ? rc : 1;
    if ((rc = launch_coarse_solve_128(s, pc, t.rc, t.yc, t.done))) return rc > 0 ? rc : 1;
    int ndone = 0;
    if (t.omega != 0.0) {
        // the cycle: six fine-grid kernels per iteration, G workgroups per member (enough to put two on every CU)
        const int G = std::max(1, std::min(8, (2 * f->ctx->num_cu + p.N - 1) / p.N));  // tl_parts holds 8 per member
        TgArgs a{(double*)f->tl_parts.p, G, 1};
        const dim3 grid(p.N, G);
        hipLaunchKernelGGL(k_tg_correct, grid, dim3(PT), 0, s, p, t, a);
        hipLaunchKernelGGL(k_tg_postsmooth, grid, dim3(PT), 0, s, p, t, a);
        hipLaunchKernelGGL(k_tg_direction, grid, dim3(PT), 0, s, p, t, a, 1);
        for (int it = 0; it < p.cg_max_iter; ++it) {
            a.par = it & 1;
            hipLaunchKernelGGL(k_tg_spmv, grid, dim3(PT), 0, s, p, t, a);
            hipLaunchKernelGGL(k_tg_update, grid, dim3(PT), 0, s, p, t, a);
            hipLaunchKernelGGL(k_tg_restrict, grid, dim3(PT), 0, s, p, t, a);
            if ((rc = launch_coarse_solve_128(s, pc, t.rc, t.yc, t.done))) return rc > 0 ? rc : 1;
            hipLaunchKernelGGL(k_tg_correct, grid, dim3(PT), 0, s, p, t, a);
            hipLaunchKernelGGL(k_tg_postsmooth, grid, dim3(PT), 0, s, p, t, a);
            hipLaunchKernelGGL(k_tg_direction, grid, dim3(PT), 0, s, p, t, a, 0);
            HM_HIP(hipGetLastError());
            if ((it & 7) == 7) {
                HM_HIP(hipMemcpyAsync(&ndone, t.ndone, 4, hipMemcpyDeviceToHost, s));
                HM_HIP(hipStreamSynchronize(s));
                if (ndone >= p.N) break;
            }
        }
    } else {
        hipLaunchKernelGGL(k_tl_dir, dim3(p.N), dim3(PT), 0, s, p, t, 1);
        HM_HIP(hipGetLastError());
        for (int it = 0; it < p.cg_max_iter; ++it) {
            hipLaunchKernelGGL(k_tl_iter, dim3(p.N), dim3(PT), 0, s, p, t);
            if ((rc = launch_coarse_solve_128(s, pc, t.rc, t.yc, t.done))) return rc > 0 ? rc : 1;
            hipLaunchKernelGGL(k_tl_dir, dim3(p.N), dim3(PT), 0, s, p, t, 0);
            HM_HIP(hipGetLastError());
            if ((it & 7) == 7) {
                HM_HIP(hipMemcpyAsync(&ndone, t.ndone, 4, hipMemcpyDeviceToHost, s));
                HM_HIP(hipStreamSynchronize(s));
                if (ndone >= p.N) break;
            }
        }
    }
    hipLaunchKernelGGL(k_tl_final, dim3(p.N), dim3(PT), 0, s, p, t, k);
    HM_HIP(hipGetLastError());
    return 0;
}
