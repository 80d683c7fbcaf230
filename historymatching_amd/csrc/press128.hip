// press128.hip -- Ny = 128 fp64 specialisation of the pressure step (listings TPFA, Pres).
//
// Same algorithm as the generic kernel (forward.hip): block elimination of the 5-point SPD system along ix with
// blocks of Ny = 128 unknowns, every Schur complement S_i = D_i - E G_{i-1} E inverted explicitly by symmetric
// Gauss-Jordan sweeps so that both substitution passes are dense mat-vecs.  What changes is where the data lives:
//   * the 128 x 128 block being inverted is REGISTER-resident: 1024 threads x 16 fp64 entries, thread (tr, tc)
//     owns rows tr+32a, columns tc+32b (a, b = 0..3), so every register index in the sweep is a compile-time
//     constant (the pivot loop is unrolled over the 4 column slabs);
//   * per pivot only the pivot column (= row, by symmetry) crosses threads: 128 doubles through a double-buffered
//     LDS line, one barrier per pivot;
//   * the inverses G_i (16.8 MB per member) stream to HBM in thread-major 16-byte chunks (fully coalesced) and
//     stream back in the same layout for the back substitution -- no transposition anywhere.
// Assembly of the transmissibilities and the face fluxes are the shared bit-exact routines of fwd_dev.h.
#include "fwd_dev.h"

namespace {

constexpr int NB = 128;  // block size = Ny
constexpr int NT = 1024;

__device__ __forceinline__ double rcp_newton(double d) {
    // reciprocal to fp64 accuracy: hardware estimate + two Newton steps (solver path: not a bit-exact path)
    double x = __builtin_amdgcn_rcp(d);
    double e = fma(-d, x, 1.0);
    x = fma(x, e, x);
    e = fma(-d, x, 1.0);
    x = fma(x, e, x);
    return x;
}

__device__ __forceinline__ double sum_over_tc(double v) {
    // sum across the 32 lanes that share tr (= one half-wave)
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m, 32);
    return v;
}

// t[r] = sum_c A[r][c] v[c] for the 4 rows of this thread; v in LDS; result valid in every lane of the half-wave
__device__ __forceinline__ void matvec4(const double (&A)[4][4], const double* __restrict__ v, int tc, double (&t)[4]) {
    double vv[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) vv[b] = v[tc + 32 * b];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        double s = A[a][0] * vv[0];
        s = fma(A[a][1], vv[1], s);
        s = fma(A[a][2], vv[2], s);
        s = fma(A[a][3], vv[3], s);
        t[a] = sum_over_tc(s);
    }
}

template <int KB>
__device__ __forceinline__ void sweep_slab(double (&A)[4][4], double* __restrict__ colbuf, int& cur, int tr, int tc, int& bad) {
    for (int kk = 0; kk < 32; ++kk) {
        double* cb = colbuf + cur * NB;
        if (tc == kk) {
#pragma unroll
            for (int a = 0; a < 4; ++a) cb[tr + 32 * a] = A[a][KB];
        }
        __syncthreads();
        double cr[4], tcv[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) cr[a] = cb[tr + 32 * a];
        const double d = cb[kk + 32 * KB];
        if (!(d > 0.0)) bad = 1;
        const double pinv = rcp_newton(d);
#pragma unroll
        for (int b = 0; b < 4; ++b) tcv[b] = cb[tc + 32 * b] * pinv;
        const bool rp = (tr == kk), cp = (tc == kk);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                double v = fma(-cr[a], tcv[b], A[a][b]);
                if (a == KB) v = rp ? tcv[b] : v;
                if (b == KB) v = cp ? cr[a] * pinv : v;
                if (a == KB && b == KB) v = (rp && cp) ? -pinv : v;
                A[a][b] = v;
            }
        cur ^= 1;
    }
}

template <typename TS>
__global__ __launch_bounds__(NT) void k_press128(FwdParams p, const TS* __restrict__ S_base, long long S_stride, int k) {
    __shared__ __attribute__((aligned(16))) double colbuf[2 * NB];
    __shared__ __attribute__((aligned(16))) double yprev[NB], ycur[NB], ev[NB], dgv[NB], tyv[NB + 1];
    const int m = blockIdx.x;
    const int tid = threadIdx.x;
    const int tr = tid >> 5, tc = tid & 31;
    const int Nx = p.Nx, Nxy = p.Nxy;

    const TS* S = S_base + (long long)m * S_stride;
    const double* Km = p.K + (long long)m * Nxy;
    double* TX = p.TX + (long long)m * (Nx + 1) * NB;
    double* TY = p.TY + (long long)m * Nx * (NB + 1);
    double2* G = reinterpret_cast<double2*>(p.G + (long long)m * Nx * NB * NB);
    double* yv = p.yv + (long long)m * Nxy;
    double* P = p.P + (long long)m * Nxy;
    double* Vx = p.Vx + (long long)m * (Nx + 1) * NB;
    double* Vy = p.Vy + (long long)m * Nx * (NB + 1);
    const double* q = p.q + (long long)m * p.q_mstride + (long long)(p.q_cols > 1 ? k : 0) * Nxy;

    assemble_transmissibilities<TS>(p, S, Km, P /* scratch for L */, TX, TY, tid, NT);

    double A[4][4];
    int bad = 0, cur = 0;
    for (int i = 0; i < Nx; ++i) {
        // per-block vectors: coupling e = TX[i], diagonal of D_i, off-diagonal TY[i]
        if (tid < NB) {
            const int j = tid;
            const double y1 = TY[i * (NB + 1) + j], y2 = TY[i * (NB + 1) + j + 1];
            const double x1 = TX[i * NB + j], x2 = TX[(i + 1) * NB + j];
            double dg = y1 + y2 + x1 + x2;
            if (i == 0 && j == 0) dg += Km[0] + Km[0];  // SPD pin: A[0,0] += Kx[0,0]+Ky[0,0]
            dgv[j] = dg;
            tyv[j] = y1;
            if (j == NB - 1) tyv[NB] = y2;
            ev[j] = x1;
        }
        __syncthreads();
        if (i > 0) {
            double t[4];
            matvec4(A, yprev, tc, t);
            if (tc == 0) {
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const int r = tr + 32 * a;
                    ycur[r] = q[i * NB + r] + ev[r] * t[a];
                }
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) A[a][b] = -(ev[tr + 32 * a] * A[a][b] * ev[tc + 32 * b]);
        } else {
            if (tid < NB) ycur[tid] = q[tid];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) A[a][b] = 0.0;
        }
        // add the tridiagonal D_i
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int r = tr + 32 * a, c = tc + 32 * b;
                if (r == c) A[a][b] += dgv[r];
                else if (c == r + 1) A[a][b] -= tyv[c];
                else if (r == c + 1) A[a][b] -= tyv[r];
            }
        __syncthreads();
        // symmetric sweeps over all 128 pivots: A <- -inv(A)
        sweep_slab<0>(A, colbuf, cur, tr, tc, bad);
        sweep_slab<1>(A, colbuf, cur, tr, tc, bad);
        sweep_slab<2>(A, colbuf, cur, tr, tc, bad);
        sweep_slab<3>(A, colbuf, cur, tr, tc, bad);
        // G_i = -A: keep in registers for the next block, stream to HBM (8 x 16-byte chunks, thread-major)
        double2* Gi = G + (long long)i * (NB * NB / 2);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                A[a][2 * h] = -A[a][2 * h];
                A[a][2 * h + 1] = -A[a][2 * h + 1];
                double2 v;
                v.x = A[a][2 * h];
                v.y = A[a][2 * h + 1];
                Gi[(a * 2 + h) * NT + tid] = v;
            }
        if (tid < NB) {
            yv[i * NB + tid] = ycur[tid];
            yprev[tid] = ycur[tid];
        }
        __syncthreads();
    }
    // back substitution: x_i = G_i (y_i + TX[i+1] * x_{i+1});  ycur holds x_{i+1}
    for (int i = Nx - 1; i >= 0; --i) {
        if (i < Nx - 1) {
            const double2* Gi = G + (long long)i * (NB * NB / 2);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    double2 v = Gi[(a * 2 + h) * NT + tid];
                    A[a][2 * h] = v.x;
                    A[a][2 * h + 1] = v.y;
                }
        }
        if (tid < NB) {
            double v = yv[i * NB + tid];
            if (i < Nx - 1) v += TX[(i + 1) * NB + tid] * ycur[tid];
            yprev[tid] = v;
        }
        __syncthreads();
        double t[4];
        matvec4(A, yprev, tc, t);
        __syncthreads();
        if (tc == 0) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int r = tr + 32 * a;
                ycur[r] = t[a];
                P[i * NB + r] = t[a];
            }
        }
        __syncthreads();
    }
    face_fluxes(p, P, TX, TY, Vx, Vy, tid, NT);
    if (bad && tc == 0 && tr == 0) atomicOr(&p.status[m], HM_MEMBER_BAD_PIVOT);
}

}  // namespace

// Returns 0 if launched, >0 on error, -1 if this specialisation does not apply.
int launch_pressure_128(hm_fwd* f, const void* S, long long S_stride, int k) {
    const FwdParams& p = f->p;
    if (p.Ny != NB) return -1;
    if (f->dtype == 64)
        hipLaunchKernelGGL(k_press128<double>, dim3(p.N), dim3(NT), 0, f->ctx->stream, p, (const double*)S, S_stride, k);
    else
        hipLaunchKernelGGL(k_press128<float>, dim3(p.N), dim3(NT), 0, f->ctx->stream, p, (const float*)S, S_stride, k);
    HM_HIP(hipGetLastError());
    return 0;
}
