// dgemm_mfma.hip -- small fp64 products of the ensemble-smoother update on the fp64 matrix cores
// (v_mfma_f64_16x16x4_f64): everything of size N x n_obs / n_obs x n_obs that stays fp64 in fp32 plans
//   SD  = [Y; D0] decorr        (2N x n_obs) . (n_obs x n_obs)        HistoryMatch.py:582-584
//   G   = S^T S                 split over row blocks, fixed-order reduction          :585
//   T1  = D C^-1                (+ transposed fp32 copy = the A operand of the apply)  :586
// One wave = one 32x32 tile of the result (2x2 MFMA tiles), operands straight from global memory (all of these
// matrices are L2 resident), 4 k-steps of loads in flight ahead of their MFMAs.  ~50 MFLOP each: a few microseconds
// instead of 40-55 us for the scalar row-block kernels they replace.
// Operand maps: A operand lane l holds A[l&15][l>>4], B operand B[l>>4][l&15]; C/D: col = l&15, row = (l>>4) + 4g.
#include "common.h"

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));

// C_z (M x N) = op(A) B over k in [z*kchunk, min(K, (z+1)*kchunk));  op(A)[i][k] = TA ? A[k*lda + i] : A[i*lda + k]
template <bool TA, int U = 4>
__global__ __launch_bounds__(64) void k_dgemm_mfma(int M, int N, int K, const double* __restrict__ A, int lda,
                                                   const double* __restrict__ B, int ldb, double* __restrict__ C, int ldc,
                                                   int kchunk, float* __restrict__ C32, int rows32, float* __restrict__ C32T) {
    const int l = threadIdx.x, lc = l & 15, lq = l >> 4;
    const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32, z = blockIdx.z;
    const int kbeg = z * kchunk, kend = min(K, kbeg + kchunk);
    d4 acc[2][2];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = d4{0.0, 0.0, 0.0, 0.0};
    int mrow[2], ncol[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        mrow[t] = min(m0 + 16 * t + lc, M - 1);
        ncol[t] = min(n0 + 16 * t + lc, N - 1);
    }
    for (int k0 = kbeg; k0 < kend; k0 += 4 * U) {
        double a[U][2], b[U][2];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // every load is unconditional on a clamped index (a predicated load makes the compiler serialise the 16
            // round trips of a batch); rows/columns past the edge are computed and never stored, the k tail is
            // removed by zeroing the B operand
            const int k = k0 + 4 * u + lq;
            const double kmask = k < kend ? 1.0 : 0.0;
            const int kc = min(k, kend - 1);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[u][t] = TA ? A[(size_t)kc * lda + mrow[t]] : A[(size_t)mrow[t] * lda + kc];
                b[u][t] = B[(size_t)kc * ldb + ncol[t]] * kmask;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][ti], b[u][tj], acc[ti][tj], 0, 0, 0);
    }
    double* Cz = C ? C + (size_t)z * M * ldc : nullptr;
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int row = m0 + 16 * ti + lq + 4 * g, col = n0 + 16 * tj + lc;
                if (row < M && col < N) {
                    const double v = acc[ti][tj][g];
                    if (Cz) Cz[(size_t)row * ldc + col] = v;
                    if (C32 && row < rows32) C32[(size_t)row * ldc + col] = (float)v;
                    if (C32T) C32T[(size_t)col * M + row] = (float)v;
                }
            }
}

// Lower triangle of G = A^T A (A: K x n, row-major), 16 x 16 tiles: one workgroup of 4 waves per tile (R >= C), each wave one
// quarter of the K range, partial tiles added in fixed order through LDS.  No split-K partials in memory, no reduction launch;
// the matrix-core inverse reads the lower tiles only.  Unsplit, a tile is a chain of K/4 dependent MFMAs (64 cycles each).
__global__ __launch_bounds__(256) void k_gram_lower(int n, int K, const double* __restrict__ A, int lda, double* __restrict__ G) {
    __shared__ double part[3][4][64];
    const int nt = n >> 4;
    int R = 0, t = blockIdx.x;
    while ((R + 1) * (R + 2) / 2 <= t) ++R;
    const int C = t - R * (R + 1) / 2;
    (void)nt;
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6, lc = l & 15, lq = l >> 4;
    const int kq = ((K + 3) / 4 + 3) / 4 * 4;  // k range of a wave, a multiple of 4
    const int kbeg = w * kq, kend = min(K, kbeg + kq);
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    constexpr int U = 8;
    for (int k0 = kbeg; k0 < kend; k0 += 4 * U) {
        double a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + 4 * u + lq;
            const double kmask = k < kend ? 1.0 : 0.0;
            const int kc = min(k, kend - 1);
            a[u] = A[(size_t)kc * lda + 16 * R + lc];
            b[u] = A[(size_t)kc * lda + 16 * C + lc] * kmask;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
    }
    if (w > 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) part[w - 1][g][l] = acc[g];
    }
    __syncthreads();
    if (w == 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const double v = ((acc[g] + part[0][g][l]) + part[1][g][l]) + part[2][g][l];
            G[(size_t)(16 * R + lq + 4 * g) * n + 16 * C + lc] = v;
        }
    }
}

// Centring of the observations AND the Gram matrix of the fused analysis step (hm_upd_run) in ONE launch: the two do not depend on
// each other once the Gram matrix is formed from shifted instead of centred observations,
//     Yc^T Yc = sum_k (y_k - y_0)(y_k - y_0)^T - N (ybar - y_0)(ybar - y_0)^T          (exact; y_0 = the first member's row)
// -- the rank-one term is subtracted by the inverse while it loads its tiles (spdinv.hip), like (N-1) R.  The shift by a member keeps
// the cancellation relative to the ensemble's spread (a collapsed ensemble late in an ES-MDA loses nothing), not to the magnitude
// of the observations.  Workgroups [0, ncb): 8 observation columns x 128 row lanes each -- column means in a fixed order (per-lane
// partial sums, then a tree), Yc in fp64 and fp32, the innovations D0 = obs - obs_ens - perturbs (HistoryMatch.py:582-584), and
// dmean = ybar - y_0.  Workgroups [ncb, ncb + tiles): one lower 16 x 16 tile each, 16 waves = 16 parts of the member range, partial
// tiles added in fixed order through LDS.
constexpr int CG_COLS = 8, CG_LANES = 1024 / CG_COLS;
__global__ __launch_bounds__(1024) void k_center_gram(const float* __restrict__ obs_ens, const float* __restrict__ perturbs,
                                                      const float* __restrict__ obs, int rows, int n_obs, int ncb,
                                                      double* __restrict__ YD, float* __restrict__ Yc32, double* __restrict__ dmean,
                                                      double* __restrict__ G, int* __restrict__ zero_me) {
    __shared__ double sh[15 * 4 * 64];  // centring: [128][9] partial sums; Gram: [15][4][64] partial tiles
    if (zero_me && blockIdx.x == 0 && threadIdx.x == 0) *zero_me = 0;  // the column counter of the factorisation launched next
    if ((int)blockIdx.x < ncb) {
        double (*part)[CG_COLS + 1] = reinterpret_cast<double (*)[CG_COLS + 1]>(sh);
        const int c = threadIdx.x & (CG_COLS - 1), g = threadIdx.x / CG_COLS;
        const int jraw = blockIdx.x * CG_COLS + c, j = min(jraw, n_obs - 1);
        constexpr int U = 8;
        float v[U], pv[U];
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const int r = min(g + CG_LANES * q, rows - 1);  // clamped, unconditional loads
            v[q] = obs_ens[(size_t)r * n_obs + j];
            pv[q] = perturbs[(size_t)r * n_obs + j];
        }
        double s0 = 0.0;
#pragma unroll
        for (int q = 0; q < U; ++q) s0 += (g + CG_LANES * q < rows) ? (double)v[q] : 0.0;
        for (int r = g + CG_LANES * U; r < rows; r += CG_LANES) s0 += (double)obs_ens[(size_t)r * n_obs + j];
        part[g][c] = s0;
        __syncthreads();
        for (int st = CG_LANES / 2; st > 0; st >>= 1) {  // fixed-order tree over the row lanes
            if (g < st) part[g][c] += part[g + st][c];
            __syncthreads();
        }
        const double mean = part[0][c] / (double)rows;
        if (jraw >= n_obs) return;
        if (g == 0) dmean[j] = mean - (double)obs_ens[j];
        const double ob = (double)obs[j];
        const size_t n = (size_t)rows * n_obs;
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const int r = g + CG_LANES * q;
            if (r < rows) {
                const size_t e = (size_t)r * n_obs + j;
                const double o = (double)v[q];
                YD[e] = o - mean;
                Yc32[e] = (float)(o - mean);
                YD[n + e] = ob - o - (double)pv[q];
            }
        }
        for (int r = g + CG_LANES * U; r < rows; r += CG_LANES) {
            const size_t e = (size_t)r * n_obs + j;
            const double o = (double)obs_ens[e];
            YD[e] = o - mean;
            Yc32[e] = (float)(o - mean);
            YD[n + e] = ob - o - (double)perturbs[e];
        }
        return;
    }
    double (*part)[4][64] = reinterpret_cast<double (*)[4][64]>(sh);
    int R = 0;
    const int t = blockIdx.x - ncb;
    while ((R + 1) * (R + 2) / 2 <= t) ++R;
    const int C = t - R * (R + 1) / 2;
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6, lc = l & 15, lq = l >> 4;
    const int kq = ((rows + 15) / 16 + 3) / 4 * 4;  // member range of a wave, a multiple of 4
    const int kbeg = w * kq, kend = min(rows, kbeg + kq);
    const double a0 = (double)obs_ens[16 * R + lc], b0 = (double)obs_ens[16 * C + lc];  // the shift: member 0
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    constexpr int U = 8;
    for (int k0 = kbeg; k0 < kend; k0 += 4 * U) {
        float a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kc = min(k0 + 4 * u + lq, kend - 1);
            a[u] = obs_ens[(size_t)kc * n_obs + 16 * R + lc];
            b[u] = obs_ens[(size_t)kc * n_obs + 16 * C + lc];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const double kmask = k0 + 4 * u + lq < kend ? 1.0 : 0.0;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)a[u] - a0, ((double)b[u] - b0) * kmask, acc, 0, 0, 0);
        }
    }
    if (w > 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) part[w - 1][g][l] = acc[g];
    }
    __syncthreads();
    if (w == 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            double v = acc[g];
#pragma unroll
            for (int z = 0; z < 15; ++z) v += part[z][g][l];
            G[(size_t)(16 * R + lq + 4 * g) * n_obs + 16 * C + lc] = v;
        }
    }
}

// Y = obs_ens - mean(obs_ens) (rows 0..N-1), D0 = obs - obs_ens - perturbs (rows N..2N-1)     HistoryMatch.py:582, 584
template <typename T>
__global__ void k_obs_prep(const T* __restrict__ obs_ens, const T* __restrict__ perturbs, const T* __restrict__ obs,
                           const double* __restrict__ colsum_y, double inv_n_total, int rows, int n_obs, double* __restrict__ YD,
                           const T* __restrict__ decorr, double* __restrict__ decorr64) {
    const size_t n = (size_t)rows * n_obs;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < (size_t)n_obs * n_obs; e += (size_t)gridDim.x * blockDim.x)
        decorr64[e] = (double)decorr[e];
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(e % n_obs);
        const double o = (double)obs_ens[e];
        YD[e] = o - colsum_y[j] * inv_n_total;
        YD[n + e] = (double)obs[j] - o - (double)perturbs[e];
    }
}

}  // namespace

// C (M x N, ldc) = op(A) B.  ksplit > 1: writes ksplit partial results C_z (z-major, M*ldc apart) for a fixed-order reduction
// by the caller.  C32 / C32T: optional fp32 copies (same layout for rows < rows32 / transposed N x M).
int dgemm_mfma(hipStream_t s, bool transA, int M, int N, int K, const double* A, int lda, const double* B, int ldb, double* C,
               int ldc, int ksplit, float* C32, int rows32, float* C32T) {
    const int kchunk = ksplit > 1 ? (((K + ksplit - 1) / ksplit + 3) / 4) * 4 : K;
    const dim3 grid((N + 31) / 32, (M + 31) / 32, ksplit > 1 ? (K + kchunk - 1) / kchunk : 1), block(64);
    // short K unsplit (the gain D0 B^-1: K = n_obs): 8 k-steps of loads in flight halve the number of dependent round trips
    const bool deep = ksplit <= 1 && K <= 512;
    if (transA) hipLaunchKernelGGL((k_dgemm_mfma<true, 4>), grid, block, 0, s, M, N, K, A, lda, B, ldb, C, ldc, kchunk, C32, rows32, C32T);
    else if (deep) hipLaunchKernelGGL((k_dgemm_mfma<false, 8>), grid, block, 0, s, M, N, K, A, lda, B, ldb, C, ldc, kchunk, C32, rows32, C32T);
    else hipLaunchKernelGGL((k_dgemm_mfma<false, 4>), grid, block, 0, s, M, N, K, A, lda, B, ldb, C, ldc, kchunk, C32, rows32, C32T);
    HM_HIP(hipGetLastError());
    return (int)grid.z > 0 ? 0 : 0;
}

// G (n x n, lower 16 x 16 tiles only) = A^T A, A: K x n.  Returns -1 if n is not a multiple of 16.
int gram_lower_mfma(hipStream_t s, int n, int K, const double* A, int lda, double* G) {
    if (n % 16 != 0 || K < 1) return -1;
    const int nt = n / 16;
    hipLaunchKernelGGL(k_gram_lower, dim3(nt * (nt + 1) / 2), dim3(256), 0, s, n, K, A, lda, G);
    HM_HIP(hipGetLastError());
    return 0;
}

// One launch: centred observations (YD rows [0, rows): Yc fp64, rows [rows, 2 rows): innovations D0; Yc32: Yc in fp32), dmean = column
// mean minus the first member's row, and the lower 16 x 16 tiles of sum_k (y_k - y_0)(y_k - y_0)^T; *zero_me = 0 if given.  Returns -1 if
// n_obs % 16 != 0.
int center_gram_mfma(hipStream_t s, const float* obs_ens, const float* perturbs, const float* obs, int rows, int n_obs, double* YD,
                     float* Yc32, double* dmean, double* G, int* zero_me) {
    if (n_obs % 16 != 0 || rows < 1) return -1;
    const int nt = n_obs / 16, ncb = (n_obs + CG_COLS - 1) / CG_COLS;
    hipLaunchKernelGGL(k_center_gram, dim3(ncb + nt * (nt + 1) / 2), dim3(1024), 0, s, obs_ens, perturbs, obs, rows, n_obs, ncb, YD, Yc32,
                       dmean, G, zero_me);
    HM_HIP(hipGetLastError());
    return 0;
}

int dgemm_mfma_splits(int K, int ksplit) {
    const int kchunk = (((K + ksplit - 1) / ksplit + 3) / 4) * 4;
    return (K + kchunk - 1) / kchunk;
}

template <typename T>
int obs_prep(hipStream_t s, const T* obs_ens, const T* perturbs, const T* obs, const double* colsum_y, double inv_n_total,
             int rows, int n_obs, double* YD, const T* decorr, double* decorr64) {
    const size_t n = (size_t)rows * n_obs;
    hipLaunchKernelGGL(k_obs_prep<T>, dim3((unsigned)std::min<size_t>(1024, (n + 255) / 256)), dim3(256), 0, s, obs_ens, perturbs, obs,
                       colsum_y, inv_n_total, rows, n_obs, YD, decorr, decorr64);
    HM_HIP(hipGetLastError());
    return 0;
}
template int obs_prep<float>(hipStream_t, const float*, const float*, const float*, const double*, double, int, int, double*, const float*, double*);
template int obs_prep<double>(hipStream_t, const double*, const double*, const double*, const double*, double, int, int, double*, const double*, double*);
