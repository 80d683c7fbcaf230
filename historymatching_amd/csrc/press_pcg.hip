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
    double* TX = p.TX + (long long)m * (Nx + 1) * Ny;
    double* TY = p.TY + (long long)m * Nx * (Ny + 1);
    double* x = p.P + (long long)m * Nxy;
    double* r = p.cg_r + (long long)m * Nxy;
    double* pv = p.cg_p + (long long)m * Nxy;
    double* Ap = p.yv + (long long)m * Nxy;  // also the scratch of the assembly
    double* Vx = p.Vx + (long long)m * (Nx + 1) * Ny;
    double* Vy = p.Vy + (long long)m * Nx * (Ny + 1);
    const double* q = p.q + (long long)(p.q_cols > 1 ? k : 0) * Nxy;
    const double pin = Km[0] + Km[0];  // SPD pin: A[0,0] += Kx[0,0]+Ky[0,0]

    assemble_transmissibilities<TS>(p, S, Km, Ap /* scratch for L */, TX, TY, tid, PT);
    __syncthreads();

    auto diag = [&](int j, int ix, int iy) {
        const double y1 = TY[ix * (Ny + 1) + iy], y2 = TY[ix * (Ny + 1) + iy + 1];
        const double x1 = TX[ix * Ny + iy], x2 = TX[(ix + 1) * Ny + iy];
        double dg = y1 + y2 + x1 + x2;
        if (j == 0) dg += pin;
        return dg;
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

}  // namespace

int launch_pressure_pcg(hm_fwd* f, const void* S, long long S_stride, int k) {
    const FwdParams& p = f->p;
    hipStream_t s = f->ctx->stream;
    if (f->dtype == 64) hipLaunchKernelGGL(k_pressure_pcg<double>, dim3(p.N), dim3(PT), 0, s, p, (const double*)S, S_stride, k);
    else hipLaunchKernelGGL(k_pressure_pcg<float>, dim3(p.N), dim3(PT), 0, s, p, (const float*)S, S_stride, k);
    HM_HIP(hipGetLastError());
    return 0;
}
