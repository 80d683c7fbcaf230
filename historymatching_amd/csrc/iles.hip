// iles.hip -- localised iterative ensemble smoother in ensemble subspace, partitioned over local domains, on the device.
//
// Replaces  ILES  notebooks/HistoryMatch.py:1007-1064  (per state element i one N x N weight matrix W_i, Gauss-Newton step
// restricted to the observations with sqrt(taper[i]) > 1e-2) in the batched form the reference points at (HistoryMatch.py:802-804
// "sequentially processing batches, i.e. subsets/domains rather than iterating over each single element";
// notebooks/tools/localization.py:95-145 rectangular_partitioning): the state elements of one batch share ONE weight matrix and
// one taper row.  With one element per batch this IS the reference's algorithm; its M N^2 weight storage is then B N^2.
//
// Per batch b and iterate (all fp64), with S = center(Eo decorr), D = (obs - Eo - perturbs) decorr (N x n_obs, from the host):
//     c = sqrt(taper_b), jj = c > cutoff;  S_b = S[:, jj] c[jj],  D_b = D[:, jj] c[jj]                 (HistoryMatch.py:1035-1041)
//     Y0 = center(pinv(W)) S_b = center(W^-1 S_b)     -- centring commutes with the right factor; W^-1 S_b by LU with partial
//                                                        pivoting on [W | S_b] (W = I + ... stays non-singular: pinv = inv)
//     the reference's step  dW = (D_b Y0^T + (N-1)(I - W)) (Y0 Y0^T + (N-1) I)^-1  (HistoryMatch.py:1046-1056, through the SVD of
//     Y0) is evaluated by the push-through identity  (Y0 Y0^T + (N-1) I)^-1 = (I - Y0 C^-1 Y0^T)/(N-1),  C = Y0^T Y0 + (N-1) I
//     (n_loc x n_loc, SPD: Cholesky in LDS), which collapses to
//         dW = (I - W) + [D_b - (I - W) Y0] C^-1 Y0^T
//     W <- W + xStep dW.
// Composition (HistoryMatch.py:1021-1022):  E[:, i] = x0[i] + W_b X0[:, i]  for the elements i of batch b.
//
// Two forms of the step.  k_iles_batch: one workgroup per batch, the N x N objects in HBM/L2 (a workspace per resident workgroup),
// the n_loc x n_loc Cholesky factor in LDS -- the form for many small batches (N ~ 100: 256 domains in 2 ms).  At N ~ 1000 one
// workgroup eliminating a 1000 x 1000 matrix column by column through memory takes 0.36 s per batch, so ensembles of 256..1024
// members take the BLOCKED form (k_ib_*, below): the same elimination with partial pivoting on the augmented matrix [W | S_b] in
// panels of 16 columns -- the panel factored in one workgroup's LDS, row swaps + triangular solve of the panel's rows, and the
// rank-16 update of the trailing matrix as launches over all resident batches at once (every element sees the same sequence of
// fused multiply-adds as in the unblocked form) -- the back substitution with eight rows of U at a time in LDS, one wave per
// right-hand side, and the products of size N x N x n_loc on the fp64 matrix cores (k_ib_gemm).
#include "common.h"

// spdinv.hip: W = inv(G + ridge I), n a multiple of 16 <= 256, lower 16 x 16 tiles of G read, full W written; *flag |= 1 on a bad pivot
int spd_inverse_mfma(hipStream_t s, const double* G, int nparts, int n, double ridge, double* W, int* flag, const double* add,
                     double add_scale, const double* rank1, double rank1_scale);

namespace {

constexpr int NT = 1024;

struct IlesArgs {
    int N, n_obs, B, b0;
    const double* taper_b;  // B x n_obs
    double cutoff;
    const double* S;        // N x n_obs
    const double* D;        // N x n_obs
    double* W;              // B x N x N
    double* ws;             // per resident workgroup: LU (N*N) | Z/Y0 (N*n) | Db (N*n) | R (N*n) | T (N*n) | Cinv (n*n)
    size_t ws_stride;       // doubles
    double xstep;
    int* flag;
};

// Cholesky factorisation (right-looking, in place) of the SPD matrix held as a packed lower triangle in LDS, then C^-1 column by
// column: thread q solves L L^T x = e_q; x goes to Cinv[i * ldc + q] (coalesced over q).  Whole workgroup of NT threads; returns
// non-zero (uniformly) on a non-positive pivot.
__device__ int chol_inverse_packed(double* L, int nl, double* __restrict__ Cinv, int ldc, int tid) {
    int bad = 0;
    for (int k = 0; k < nl; ++k) {
        const int kk = k * (k + 1) / 2;
        const double d = L[kk + k];
        if (!(d > 0.0)) bad = 1;
        const double dk = sqrt(d), inv = 1.0 / dk;
        __syncthreads();
        for (int r = k + 1 + tid; r < nl; r += NT) L[r * (r + 1) / 2 + k] *= inv;
        if (tid == 0) L[kk + k] = dk;
        __syncthreads();
        for (int r = k + 1 + (tid >> 5); r < nl; r += 32) {
            const int rbase = r * (r + 1) / 2;
            const double lrk = L[rbase + k];
            for (int c = k + 1 + (tid & 31); c <= r; c += 32) L[rbase + c] = fma(-lrk, L[c * (c + 1) / 2 + k], L[rbase + c]);
        }
        __syncthreads();
    }
    if (bad) return 1;
    if (tid < nl) {
        const int q = tid;
        for (int i = 0; i < nl; ++i) {
            double s = i == q ? 1.0 : 0.0;
            const int ib = i * (i + 1) / 2;
            for (int j = 0; j < i; ++j) s = fma(-L[ib + j], Cinv[(size_t)j * ldc + q], s);
            Cinv[(size_t)i * ldc + q] = s / L[ib + i];
        }
        for (int i = nl - 1; i >= 0; --i) {
            double s = Cinv[(size_t)i * ldc + q];
            for (int j = i + 1; j < nl; ++j) s = fma(-L[j * (j + 1) / 2 + i], Cinv[(size_t)j * ldc + q], s);
            Cinv[(size_t)i * ldc + q] = s / L[i * (i + 1) / 2 + i];
        }
    }
    __syncthreads();
    return 0;
}

__global__ __launch_bounds__(NT) void k_iles_batch(IlesArgs a) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int N = a.N, n = a.n_obs, tid = threadIdx.x;
    const int b = a.b0 + blockIdx.x;
    if (b >= a.B) return;
    double* L = sm;                                   // packed lower triangle, n(n+1)/2
    double* cvec = L + (size_t)n * (n + 1) / 2;       // n
    double* red = cvec + n;                           // NT
    double* mult = red + NT;                          // N multipliers / column means
    int* redi = reinterpret_cast<int*>(mult + N);     // NT
    int* jj = redi + NT;                              // n
    int* misc = jj + n;                               // [0] = n_loc, [1] = pivot row
    double* Wb = a.W + (size_t)b * N * N;
    double* ws = a.ws + (size_t)blockIdx.x * a.ws_stride;
    double* LU = ws;
    double* Z = LU + (size_t)N * N;
    double* Db = Z + (size_t)N * n;
    double* R = Db + (size_t)N * n;
    double* T = R + (size_t)N * n;
    double* Cinv = T + (size_t)N * n;

    for (int j = tid; j < n; j += NT) cvec[j] = sqrt(a.taper_b[(size_t)b * n + j]);
    __syncthreads();
    if (tid == 0) {  // order-preserving selection, as the reference's boolean mask
        int cnt = 0;
        for (int j = 0; j < n; ++j)
            if (cvec[j] > a.cutoff) jj[cnt++] = j;
        misc[0] = cnt;
    }
    __syncthreads();
    const int nl = misc[0];
    if (nl == 0) return;  // no observation in range: dW = 0 (HistoryMatch.py:1038-1039)

    for (int e = tid; e < N * N; e += NT) LU[e] = Wb[e];
    for (int e = tid; e < N * nl; e += NT) {
        const int r = e / nl, q = e - r * nl, j = jj[q];
        Z[e] = a.S[(size_t)r * n + j] * cvec[j];
        Db[e] = a.D[(size_t)r * n + j] * cvec[j];
    }
    __syncthreads();

    // ---- Z <- W^-1 S_b: elimination with partial pivoting on [LU | Z]
    for (int k = 0; k < N; ++k) {
        double best = -1.0;
        int bi = k;
        for (int r = k + tid; r < N; r += NT) {
            const double v = fabs(LU[(size_t)r * N + k]);
            if (v > best) { best = v; bi = r; }
        }
        red[tid] = best; redi[tid] = bi;
        __syncthreads();
        for (int s = NT / 2; s > 0; s >>= 1) {
            if (tid < s) {
                const double o = red[tid + s];
                const int oi = redi[tid + s];
                if (o > red[tid] || (o == red[tid] && oi < redi[tid])) { red[tid] = o; redi[tid] = oi; }
            }
            __syncthreads();
        }
        const int p = redi[0];
        const double pmax = red[0];
        __syncthreads();
        if (!(pmax > 0.0)) {  // singular weight matrix (or NaN): leave W_b as it is and report
            if (tid == 0) atomicOr(a.flag, 2);
            return;
        }
        if (p != k) {
            for (int c = k + tid; c < N; c += NT) {
                const double t = LU[(size_t)k * N + c];
                LU[(size_t)k * N + c] = LU[(size_t)p * N + c];
                LU[(size_t)p * N + c] = t;
            }
            for (int q = tid; q < nl; q += NT) {
                const double t = Z[(size_t)k * nl + q];
                Z[(size_t)k * nl + q] = Z[(size_t)p * nl + q];
                Z[(size_t)p * nl + q] = t;
            }
            __syncthreads();
        }
        const double piv = LU[(size_t)k * N + k];
        for (int r = k + 1 + tid; r < N; r += NT) mult[r] = LU[(size_t)r * N + k] / piv;
        __syncthreads();
        const int wa = N - k - 1, width = wa + nl;
        for (int e = tid; e < wa * width; e += NT) {
            const int r = k + 1 + e / width, cc = e % width;
            if (cc < wa) LU[(size_t)r * N + k + 1 + cc] -= mult[r] * LU[(size_t)k * N + k + 1 + cc];
            else Z[(size_t)r * nl + (cc - wa)] -= mult[r] * Z[(size_t)k * nl + (cc - wa)];
        }
        __syncthreads();
    }
    for (int k = N - 1; k >= 0; --k) {  // back substitution, column oriented
        const double ukk = LU[(size_t)k * N + k];
        for (int q = tid; q < nl; q += NT) Z[(size_t)k * nl + q] /= ukk;
        __syncthreads();
        for (int e = tid; e < k * nl; e += NT) {
            const int r = e / nl, q = e - r * nl;
            Z[e] -= LU[(size_t)r * N + k] * Z[(size_t)k * nl + q];
        }
        __syncthreads();
    }
    // ---- Y0 = center(Z)
    for (int q = tid; q < nl; q += NT) {
        double s = 0.0;
        for (int r = 0; r < N; ++r) s += Z[(size_t)r * nl + q];
        red[q] = s / (double)N;  // nl <= n_obs <= NT
    }
    __syncthreads();
    for (int e = tid; e < N * nl; e += NT) Z[e] -= red[e % nl];
    __syncthreads();
    double* Y0 = Z;
    // ---- C = Y0^T Y0 + (N-1) I, packed lower triangle in LDS
    for (int e = tid; e < nl * (nl + 1) / 2; e += NT) {
        int r = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
        while ((r + 1) * (r + 2) / 2 <= e) ++r;
        while (r * (r + 1) / 2 > e) --r;
        const int c = e - r * (r + 1) / 2;
        double s = 0.0;
        for (int m = 0; m < N; ++m) s = fma(Y0[(size_t)m * nl + r], Y0[(size_t)m * nl + c], s);
        L[e] = s + (r == c ? (double)(N - 1) : 0.0);
    }
    __syncthreads();
    // ---- Cholesky in place and C^-1
    if (chol_inverse_packed(L, nl, Cinv, nl, tid)) {
        if (tid == 0) atomicOr(a.flag, 1);
        return;
    }
    // ---- R = D_b - (I - W) Y0 = D_b - Y0 + W Y0
    for (int e = tid; e < N * nl; e += NT) {
        const int r = e / nl, q = e - r * nl;
        double s = 0.0;
        for (int m = 0; m < N; ++m) s = fma(Wb[(size_t)r * N + m], Y0[(size_t)m * nl + q], s);
        R[e] = Db[e] - Y0[e] + s;
    }
    __syncthreads();
    // ---- T = R C^-1
    for (int e = tid; e < N * nl; e += NT) {
        const int r = e / nl, q = e - r * nl;
        double s = 0.0;
        for (int j = 0; j < nl; ++j) s = fma(R[(size_t)r * nl + j], Cinv[(size_t)j * nl + q], s);
        T[e] = s;
    }
    __syncthreads();
    // ---- W <- W + xstep ((I - W) + T Y0^T)
    for (int e = tid; e < N * N; e += NT) {
        const int r = e / N, c = e - r * N;
        double s = 0.0;
        for (int q = 0; q < nl; ++q) s = fma(T[(size_t)r * nl + q], Y0[(size_t)c * nl + q], s);
        const double w = Wb[e];
        Wb[e] = w + a.xstep * (((r == c ? 1.0 : 0.0) - w) + s);
    }
}

// E[n, i] = x0[i] + sum_k W_b[n, k] X0[k, i] for the elements i of batch b = blockIdx.x
__global__ __launch_bounds__(256) void k_iles_compose(int N, int M, const int* __restrict__ boff, const int* __restrict__ bidx,
                                                      const double* __restrict__ W, const double* __restrict__ X0,
                                                      const double* __restrict__ x0, double* __restrict__ E) {
    const int b = blockIdx.x;
    const int lo = boff[b], cnt = boff[b + 1] - lo;
    const double* Wb = W + (size_t)b * N * N;
    for (int e = threadIdx.x; e < N * cnt; e += blockDim.x) {
        const int nrow = e / cnt, i = bidx[lo + (e - nrow * cnt)];
        double s = 0.0;
        for (int k = 0; k < N; ++k) s = fma(Wb[(size_t)nrow * N + k], X0[(size_t)k * M + i], s);
        E[(size_t)nrow * M + i] = x0[i] + s;
    }
}

__global__ void k_iles_center(int N, int M, const double* __restrict__ E, double* __restrict__ X0, double* __restrict__ x0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    double s = 0.0;
    for (int k = 0; k < N; ++k) s += E[(size_t)k * M + i];
    const double mean = s / (double)N;
    x0[i] = mean;
    for (int k = 0; k < N; ++k) X0[(size_t)k * M + i] = E[(size_t)k * M + i] - mean;
}

__global__ void k_iles_identity(int N, long long total, double* __restrict__ W) {
    long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x, nn = (long long)N * N;
    for (; e < total; e += stride) {
        const long long w = e % nn;
        W[e] = (w / N == w % N) ? 1.0 : 0.0;
    }
}


// ------------------------------------------------------------------------------------------------------------------------
// Blocked form of the step (256 <= N <= 1024): launches over all resident batches (blockIdx.y or .x = batch within the group).
// Workspace per resident batch (doubles): Aug = [LU | Z] (N x lda) | Db | R | T | Y0 (N x n each) | Y0T (n x N) | C (n x n) |
// Cinv (n x n) | means (n).  The products T Y0^T (N x N) go to the Aug area, which is free by then.
// ------------------------------------------------------------------------------------------------------------------------
constexpr int IB_PB = 16;      // panel width
constexpr int IB_PS = IB_PB + 1;  // row stride of the update's L21 tile in LDS (doubles)
constexpr int IB_BS_ROWS = 8;  // rows of U per LDS block of the back substitution = right-hand sides (waves) per workgroup

struct IbArgs {
    int N, n, lda, b0;
    double xstep;
    double* W;             // B x N x N
    double* ws;            // per resident batch
    size_t ws_stride;      // doubles
    const int* nloc;       // B: observations in range
    const int* jj;         // B x n: their indices, in order
    const double* cvec;    // B x n: sqrt(taper)
    const double* S;       // N x n
    const double* D;       // N x n
    int* ipiv;             // resident x N
    int* dead;             // resident: the batch's weight matrix turned out singular (W_b stays as it is)
    int* flag;
};
struct IbWs {
    double *aug, *Db, *R, *T, *Y0, *Y0T, *C, *Cinv, *means;
};
__device__ __host__ inline int ib_n16(int n) { return (n + 15) & ~15; }  // leading dimension of C and Cinv (spd_inverse_mfma wants whole 16 x 16 tiles)
__device__ __host__ inline size_t ib_ws_doubles(int N, int n, int lda) { return (size_t)N * lda + (size_t)5 * N * n + (size_t)2 * ib_n16(n) * ib_n16(n) + n; }
__device__ __host__ inline IbWs ib_ws(double* base, int N, int n, int lda) {
    IbWs w;
    w.aug = base;
    w.Db = w.aug + (size_t)N * lda;
    w.R = w.Db + (size_t)N * n;
    w.T = w.R + (size_t)N * n;
    w.Y0 = w.T + (size_t)N * n;
    w.Y0T = w.Y0 + (size_t)N * n;
    w.C = w.Y0T + (size_t)N * n;
    w.Cinv = w.C + (size_t)ib_n16(n) * ib_n16(n);
    w.means = w.Cinv + (size_t)ib_n16(n) * ib_n16(n);
    return w;
}

// c = sqrt(taper_b), the observations in range in order (the reference's boolean mask), their number -- once per plan
__global__ void k_ib_select(int n, const double* __restrict__ taper_b, double cutoff, double* __restrict__ cvec, int* __restrict__ jj,
                            int* __restrict__ nloc) {
    const int b = blockIdx.x;
    for (int j = threadIdx.x; j < n; j += blockDim.x) cvec[(size_t)b * n + j] = sqrt(taper_b[(size_t)b * n + j]);
    __syncthreads();
    if (threadIdx.x == 0) {
        int cnt = 0;
        for (int j = 0; j < n; ++j)
            if (cvec[(size_t)b * n + j] > cutoff) jj[(size_t)b * n + cnt++] = j;
        nloc[b] = cnt;
    }
}

// Aug = [W_b | S[:, jj] c[jj]], Db = D[:, jj] c[jj]; one workgroup per row
__global__ __launch_bounds__(256) void k_ib_fill(IbArgs a) {
    const int g = blockIdx.y, b = a.b0 + g, r = blockIdx.x, N = a.N, n = a.n;
    const int nl = a.nloc[b];
    if (nl == 0) return;
    const IbWs w = ib_ws(a.ws + (size_t)g * a.ws_stride, N, n, a.lda);
    double* row = w.aug + (size_t)r * a.lda;
    const double* Wr = a.W + ((size_t)b * N + r) * N;
    for (int c = threadIdx.x; c < N; c += 256) row[c] = Wr[c];
    for (int q = threadIdx.x; q < nl; q += 256) {
        const int j = a.jj[(size_t)b * n + q];
        const double cv = a.cvec[(size_t)b * n + j];
        row[N + q] = a.S[(size_t)r * n + j] * cv;
        w.Db[(size_t)r * n + q] = a.D[(size_t)r * n + j] * cv;
    }
    if (r == 0 && threadIdx.x == 0) a.dead[g] = 0;
}

// Panel [kb, kb + pw) of the elimination with partial pivoting, rows kb..N-1 (N <= 1024): ONE ROW PER THREAD, in registers.  Per
// column: pivot search (largest magnitude, lowest row on ties: wave shuffles + one LDS round), the pivot row and row j trade places
// through LDS, multipliers (kept in place of the eliminated entries) and the update of the row's later panel columns in registers --
// two workgroup barriers per column.
__global__ __launch_bounds__(NT) void k_ib_panel(IbArgs a, int kb) {
    __shared__ double prow[IB_PB], srow[IB_PB], redv[NT / 64];
    __shared__ int redi[NT / 64];
    const int g = blockIdx.x, b = a.b0 + g, N = a.N, tid = threadIdx.x;
    const int nl = a.nloc[b];
    if (nl == 0 || a.dead[g]) return;
    const IbWs w = ib_ws(a.ws + (size_t)g * a.ws_stride, N, a.n, a.lda);
    const int rows = N - kb, pw = min(IB_PB, rows);
    const bool live = tid < rows;
    double row[IB_PB];
    {
        const double* srcp = w.aug + (size_t)(kb + (live ? tid : 0)) * a.lda + kb;
#pragma unroll
        for (int c = 0; c < IB_PB; ++c) row[c] = (live && c < pw) ? srcp[c < pw ? c : 0] : 0.0;
    }
    for (int j = 0; j < pw; ++j) {  // (not unrolled: the compiler indexes the row registers by j)
        double best = (live && tid >= j) ? fabs(row[j]) : -1.0;
        if (!(best >= 0.0)) best = -1.0;  // NaN: never chosen; a column of NaN ends as "singular" below
        int bi = tid;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double o = __shfl_xor(best, off);
            const int oi = __shfl_xor(bi, off);
            if (o > best || (o == best && oi < bi)) { best = o; bi = oi; }
        }
        if ((tid & 63) == 0) { redv[tid >> 6] = best; redi[tid >> 6] = bi; }
        __syncthreads();
        best = redv[0]; bi = redi[0];
#pragma unroll
        for (int q = 1; q < NT / 64; ++q) {
            const double o = redv[q];
            const int oi = redi[q];
            if (o > best || (o == best && oi < bi)) { best = o; bi = oi; }
        }
        if (!(best > 0.0)) {  // singular weight matrix (or NaN): W_b stays as it is
            if (tid == 0) { a.dead[g] = 1; atomicOr(a.flag, 2); }
            return;
        }
        if (tid == 0) a.ipiv[(size_t)g * N + kb + j] = kb + bi;
        if (tid == bi) {
#pragma unroll
            for (int c = 0; c < IB_PB; ++c) prow[c] = row[c];
        }
        if (tid == j && bi != j) {
#pragma unroll
            for (int c = 0; c < IB_PB; ++c) srow[c] = row[c];
        }
        __syncthreads();
        if (bi != j) {
            if (tid == j) {
#pragma unroll
                for (int c = 0; c < IB_PB; ++c) row[c] = prow[c];
            } else if (tid == bi) {
#pragma unroll
                for (int c = 0; c < IB_PB; ++c) row[c] = srow[c];
            }
        }
        if (live && tid > j) {
            const double m = row[j] / prow[j];
            row[j] = m;
#pragma unroll
            for (int c = j + 1; c < IB_PB; ++c) row[c] = fma(-m, prow[c], row[c]);
        }
    }
    if (live) {
        double* dst = w.aug + (size_t)(kb + tid) * a.lda + kb;
#pragma unroll
        for (int c = 0; c < IB_PB; ++c)
            if (c < pw) dst[c] = row[c];
    }
}

// The panel's row swaps applied to the columns right of it as ONE gather per column (wave 0 composes the pw transpositions into "row
// key[i] receives what row src[i] held" by shuffles; every load of a column is in flight before its first store), then
// U12 = L11^-1 A12 (unit lower triangle of the panel): one thread per column.
__global__ __launch_bounds__(256) void k_ib_swap_trsm(IbArgs a, int kb) {
    __shared__ double L11[IB_PB][IB_PB];
    __shared__ int key[2 * IB_PB], src[2 * IB_PB], nkeys;
    const int g = blockIdx.y, b = a.b0 + g, N = a.N, tid = threadIdx.x;
    const int nl = a.nloc[b];
    if (nl == 0 || a.dead[g]) return;
    const IbWs w = ib_ws(a.ws + (size_t)g * a.ws_stride, N, a.n, a.lda);
    const int pw = min(IB_PB, N - kb);
    if (kb + pw + blockIdx.x * 256 >= N + nl) return;
    if (tid < pw * pw) L11[tid / pw][tid % pw] = w.aug[(size_t)(kb + tid / pw) * a.lda + kb + tid % pw];
    if (tid < 64) {  // wave 0, lane i = slot i of (key, src)
        const int pv = tid < pw ? a.ipiv[(size_t)g * N + kb + tid] : 0;
        int k_ = tid < pw ? kb + tid : -1, s_ = k_, nk = pw;
        for (int j = 0; j < pw; ++j) {
            const int pr = __shfl(pv, j);
            const unsigned long long hit = __ballot(k_ == pr);
            int ip;
            if (hit == 0ull) {
                if (tid == nk) { k_ = pr; s_ = pr; }
                ip = nk++;
            } else {
                ip = __ffsll((long long)hit) - 1;
            }
            const int sj = __shfl(s_, j), sip = __shfl(s_, ip);
            if (tid == j) s_ = sip;
            else if (tid == ip) s_ = sj;
        }
        if (tid < 2 * IB_PB) { key[tid] = tid < nk ? k_ : kb; src[tid] = tid < nk ? s_ : kb; }
        if (tid == 0) nkeys = nk;
    }
    __syncthreads();
    const int nk = nkeys;
    const int c = kb + pw + blockIdx.x * 256 + tid;
    if (c >= N + nl) return;
    double* col = w.aug + c;
    double v[2 * IB_PB];
#pragma unroll
    for (int i = 0; i < 2 * IB_PB; ++i) v[i] = col[(size_t)src[i] * a.lda];
#pragma unroll
    for (int t = 1; t < IB_PB; ++t) {
        if (t < pw) {
            double u = v[t];
#pragma unroll
            for (int q = 0; q < t; ++q) u = fma(-L11[t][q], v[q], u);
            v[t] = u;
        }
    }
#pragma unroll
    for (int i = 0; i < 2 * IB_PB; ++i)
        if (i < nk) col[(size_t)key[i] * a.lda] = v[i];
}

// A22 -= L21 U12: 64 x 64 tile per workgroup, 4 x 4 per thread; every element takes its pw fused multiply-adds in pivot order
__global__ __launch_bounds__(256) void k_ib_update(IbArgs a, int kb) {
    __shared__ double Lt[64][IB_PS];
    __shared__ double Ut[IB_PB][64];
    const int g = blockIdx.z, b = a.b0 + g, N = a.N, tid = threadIdx.x;
    const int nl = a.nloc[b];
    if (nl == 0 || a.dead[g]) return;
    const IbWs w = ib_ws(a.ws + (size_t)g * a.ws_stride, N, a.n, a.lda);
    const int pw = min(IB_PB, N - kb);
    const int r0 = kb + pw + blockIdx.y * 64, c0 = kb + pw + blockIdx.x * 64;
    if (r0 >= N || c0 >= N + nl) return;
    for (int e = tid; e < 64 * IB_PB; e += 256) {
        const int rr = e / IB_PB, pp = e % IB_PB;
        Lt[rr][pp] = (pp < pw && r0 + rr < N) ? w.aug[(size_t)(r0 + rr) * a.lda + kb + pp] : 0.0;
        const int p2 = e / 64, cc = e % 64;
        Ut[p2][cc] = (p2 < pw && c0 + cc < N + nl) ? w.aug[(size_t)(kb + p2) * a.lda + c0 + cc] : 0.0;
    }
    __syncthreads();
    const int tx = tid & 15, ty = tid >> 4;
    double acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = r0 + ty + 16 * i, c = c0 + tx + 16 * j;
            acc[i][j] = (r < N && c < N + nl) ? w.aug[(size_t)r * a.lda + c] : 0.0;
        }
    for (int p = 0; p < pw; ++p) {
        double l[4], u[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { l[i] = Lt[ty + 16 * i][p]; u[i] = Ut[p][tx + 16 * i]; }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = fma(-l[i], u[j], acc[i][j]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = r0 + ty + 16 * i, c = c0 + tx + 16 * j;
            if (r < N && c < N + nl) w.aug[(size_t)r * a.lda + c] = acc[i][j];
        }
}

// U Z = Z' in place, IB_BS_ROWS rows of U at a time in LDS (loaded by the whole workgroup), one wave per right-hand side.  Per block
// of rows [lo, hi): the part of the sums over the solved entries x[hi..N) for all rows of the block at once (independent chains), then
// the block's own small triangle by every lane alike.
__global__ __launch_bounds__(64 * IB_BS_ROWS) void k_ib_backsolve(IbArgs a) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int g = blockIdx.y, b = a.b0 + g, N = a.N, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nl = a.nloc[b];
    if (nl == 0 || a.dead[g] || blockIdx.x * IB_BS_ROWS >= nl) return;
    const IbWs w = ib_ws(a.ws + (size_t)g * a.ws_stride, N, a.n, a.lda);
    double* Ub = sm;                                   // IB_BS_ROWS x N
    double* x = Ub + (size_t)IB_BS_ROWS * N + (size_t)wv * N;  // this wave's solution
    const int q = min(blockIdx.x * IB_BS_ROWS + wv, nl - 1);    // (surplus waves redo the last column: same values)
    double* zcol = w.aug + N + q;
    // the rows of a block travel memory -> registers -> LDS; the NEXT block's are requested before the current one is computed, so their
    // round trip to memory runs beside the arithmetic (N <= 2 x 512 columns: two elements per thread and row)
    double pre[IB_BS_ROWS][2], zpre;
    auto fetch = [&](int hi_) {
        const int lo_ = max(hi_ - IB_BS_ROWS, 0), cnt_ = hi_ - lo_;
#pragma unroll
        for (int rr = 0; rr < IB_BS_ROWS; ++rr) {
            const double* src = w.aug + (size_t)(lo_ + (rr < cnt_ ? rr : 0)) * a.lda;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int c = lo_ + tid + it * 64 * IB_BS_ROWS;
                pre[rr][it] = src[c < N ? c : N - 1];
            }
        }
        zpre = zcol[(size_t)(lo_ + (lane < cnt_ ? lane : 0)) * a.lda];
    };
    fetch(N);
    for (int hi = N; hi > 0; hi -= IB_BS_ROWS) {
        const int lo = max(hi - IB_BS_ROWS, 0), cnt = hi - lo;
        __syncthreads();  // the previous block's rows are no longer read
#pragma unroll
        for (int rr = 0; rr < IB_BS_ROWS; ++rr)
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int c = lo + tid + it * 64 * IB_BS_ROWS;
                if (c < N) Ub[(size_t)rr * N + c] = pre[rr][it];
            }
        const double zl = zpre;
        if (lo > 0) fetch(lo);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // orders LDS only: __syncthreads() would wait for the loads just issued
        double sacc[IB_BS_ROWS];
#pragma unroll
        for (int r = 0; r < IB_BS_ROWS; ++r) sacc[r] = 0.0;
        for (int j = hi + lane; j < N; j += 64) {
            const double xj = x[j];
#pragma unroll
            for (int r = 0; r < IB_BS_ROWS; ++r) sacc[r] = fma(Ub[(size_t)(r < cnt ? r : 0) * N + j], xj, sacc[r]);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
#pragma unroll
            for (int r = 0; r < IB_BS_ROWS; ++r) sacc[r] += __shfl_xor(sacc[r], off);
        double xv[IB_BS_ROWS];
#pragma unroll
        for (int r = IB_BS_ROWS - 1; r >= 0; --r) {
            xv[r] = 0.0;
            if (r < cnt) {
                const double* Uk = Ub + (size_t)r * N + lo;
                double t = __shfl(zl, r) - sacc[r];
#pragma unroll
                for (int c = IB_BS_ROWS - 1; c > r; --c)
                    if (c < cnt) t = fma(-Uk[c], xv[c], t);
                xv[r] = t / Uk[r];
            }
        }
        double mine = xv[0];
#pragma unroll
        for (int r = 1; r < IB_BS_ROWS; ++r) mine = lane == r ? xv[r] : mine;
        if (lane < cnt) {
            x[lo + lane] = mine;
            zcol[(size_t)(lo + lane) * a.lda] = mine;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// column means of Z (summed in row order), then Y0 = Z - mean and its transpose
__global__ __launch_bounds__(256) void k_ib_means(IbArgs a) {
    const int g = blockIdx.x, b = a.b0 + g, N = a.N;
    const int nl = a.nloc[b];
    if (nl == 0 || a.dead[g]) return;
    const IbWs w = ib_ws(a.ws + (size_t)g * a.ws_stride, N, a.n, a.lda);
    // a column's sum in four row groups (each in row order), added in group order
    __shared__ double part[4][64];
    for (int q0 = 0; q0 < nl; q0 += 64) {
        const int q = q0 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
        const int r0 = (N * grp) / 4, r1 = (N * (grp + 1)) / 4;
        double s = 0.0;
        if (q < nl) {
#pragma unroll 8
            for (int r = r0; r < r1; ++r) s += w.aug[(size_t)r * a.lda + N + q];
        }
        part[grp][threadIdx.x & 63] = s;
        __syncthreads();
        if (grp == 0 && q < nl) w.means[q] = (((part[0][q - q0] + part[1][q - q0]) + part[2][q - q0]) + part[3][q - q0]) / (double)N;
        __syncthreads();
    }
    // C is inverted in whole 16 x 16 tiles: what lies past the n_loc x n_loc product stays zero (+ (N-1) on the diagonal)
    const int n16 = ib_n16(a.n);
    for (int e = threadIdx.x; e < n16 * n16; e += 256) w.C[e] = 0.0;
}
__global__ __launch_bounds__(256) void k_ib_y0(IbArgs a) {
    const int g = blockIdx.y, b = a.b0 + g, r = blockIdx.x, N = a.N, n = a.n;
    const int nl = a.nloc[b];
    if (nl == 0 || a.dead[g]) return;
    const IbWs w = ib_ws(a.ws + (size_t)g * a.ws_stride, N, n, a.lda);
    for (int q = threadIdx.x; q < nl; q += 256) {
        const double v = w.aug[(size_t)r * a.lda + N + q] - w.means[q];
        w.Y0[(size_t)r * n + q] = v;
        w.Y0T[(size_t)q * N + r] = v;
    }
}

// Cinv = (Y0^T Y0 + (N-1) I)^-1 from the product C (lower triangle used)
__global__ __launch_bounds__(NT) void k_ib_cinv(IbArgs a) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int g = blockIdx.x, b = a.b0 + g, N = a.N, n = a.n, tid = threadIdx.x;
    const int nl = a.nloc[b];
    if (nl == 0 || a.dead[g]) return;
    const IbWs w = ib_ws(a.ws + (size_t)g * a.ws_stride, N, n, a.lda);
    double* L = sm;
    for (int r = tid / 32; r < nl; r += NT / 32)
        for (int c = tid & 31; c <= r; c += 32) L[r * (r + 1) / 2 + c] = w.C[(size_t)r * ib_n16(n) + c] + (r == c ? (double)(N - 1) : 0.0);
    __syncthreads();
    if (chol_inverse_packed(L, nl, w.Cinv, ib_n16(n), tid)) {
        if (tid == 0) { a.dead[g] = 1; atomicOr(a.flag, 1); }
    }
}

// R = D_b - Y0 + (W Y0)  (the product is in R already)
__global__ __launch_bounds__(256) void k_ib_rfix(IbArgs a) {
    const int g = blockIdx.y, b = a.b0 + g, r = blockIdx.x, N = a.N, n = a.n;
    const int nl = a.nloc[b];
    if (nl == 0 || a.dead[g]) return;
    const IbWs w = ib_ws(a.ws + (size_t)g * a.ws_stride, N, n, a.lda);
    for (int q = threadIdx.x; q < nl; q += 256) {
        const size_t e = (size_t)r * n + q;
        w.R[e] = w.Db[e] - w.Y0[e] + w.R[e];
    }
}

// W <- W + xstep ((I - W) + T Y0^T)  (the product, N x N, is in the Aug area)
__global__ __launch_bounds__(256) void k_ib_wupdate(IbArgs a) {
    const int g = blockIdx.y, b = a.b0 + g, r = blockIdx.x, N = a.N;
    if (a.nloc[b] == 0 || a.dead[g]) return;
    const IbWs w = ib_ws(a.ws + (size_t)g * a.ws_stride, N, a.n, a.lda);
    double* Wr = a.W + ((size_t)b * N + r) * N;
    for (int c = threadIdx.x; c < N; c += 256) {
        const double wv = Wr[c];
        Wr[c] = wv + a.xstep * (((r == c ? 1.0 : 0.0) - wv) + w.aug[(size_t)r * N + c]);
    }
}

typedef double ib_d4 __attribute__((ext_vector_type(4)));

// The step's four products for every resident batch in one launch each (blockIdx.z = batch), one wave = one 32 x 32 tile on the fp64
// matrix cores as in dgemm_mfma.hip (operands from memory / L2, clamped unconditional loads, the k tail removed by zeroing B):
//   MODE 0: C (n_loc x n_loc, ld n16) = Y0^T Y0        MODE 1: R (N x n_loc) = W_b Y0
//   MODE 2: T (N x n_loc) = R Cinv                     MODE 3: Aug area (N x N, ld N) = T Y0^T  (Y0T: n_loc x N)
template <int MODE>
__global__ __launch_bounds__(64) void k_ib_gemm(IbArgs a) {
    const int g = blockIdx.z, b = a.b0 + g, N = a.N, n = a.n;
    const int nl = a.nloc[b];
    if (nl == 0 || a.dead[g]) return;
    const IbWs w = ib_ws(a.ws + (size_t)g * a.ws_stride, N, n, a.lda);
    const int n16 = ib_n16(n);
    constexpr bool TA = MODE == 0;
    const int Mr = MODE == 0 ? nl : N, Nc = MODE == 3 ? N : nl, K = MODE <= 1 ? N : nl;
    const double* A = MODE == 0 ? w.Y0 : MODE == 1 ? a.W + (size_t)b * N * N : MODE == 2 ? w.R : w.T;
    const int lda = MODE == 1 ? N : n;
    const double* Bm = MODE <= 1 ? w.Y0 : MODE == 2 ? w.Cinv : w.Y0T;
    const int ldb = MODE <= 1 ? n : MODE == 2 ? n16 : N;
    double* C = MODE == 0 ? w.C : MODE == 1 ? w.R : MODE == 2 ? w.T : w.aug;
    const int ldc = MODE == 0 ? n16 : MODE == 3 ? N : n;
    const int l = threadIdx.x, lc = l & 15, lq = l >> 4;
    const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
    if (n0 >= Nc || m0 >= Mr) return;
    ib_d4 acc[2][2];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = ib_d4{0.0, 0.0, 0.0, 0.0};
    int mrow[2], ncol[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        mrow[t] = min(m0 + 16 * t + lc, Mr - 1);
        ncol[t] = min(n0 + 16 * t + lc, Nc - 1);
    }
    constexpr int U = MODE <= 1 ? 4 : 8;
    for (int k0 = 0; k0 < K; k0 += 4 * U) {
        double av[U][2], bv[U][2];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + 4 * u + lq;
            const double kmask = k < K ? 1.0 : 0.0;
            const int kc = min(k, K - 1);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                av[u][t] = TA ? A[(size_t)kc * lda + mrow[t]] : A[(size_t)mrow[t] * lda + kc];
                bv[u][t] = Bm[(size_t)kc * ldb + ncol[t]] * kmask;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][ti], bv[u][tj], acc[ti][tj], 0, 0, 0);
    }
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int row = m0 + 16 * ti + lq + 4 * gq, col = n0 + 16 * tj + lc;
                if (row < Mr && col < Nc) C[(size_t)row * ldc + col] = acc[ti][tj][gq];
            }
}

// E[:, idx] = x0[idx] + W_b X0[:, idx] for the elements idx of batch b = blockIdx.z, on the fp64 matrix cores: one wave = one
// 32 x 32 tile (members x elements of the batch), operands from memory / L2 as in dgemm_mfma.hip
__global__ __launch_bounds__(64) void k_iles_compose_mfma(int N, int M, const int* __restrict__ boff, const int* __restrict__ bidx,
                                                          const double* __restrict__ W, const double* __restrict__ X0,
                                                          const double* __restrict__ x0, double* __restrict__ E) {
    const int b = blockIdx.z, l = threadIdx.x, lc = l & 15, lq = l >> 4;
    const int lo = boff[b], cnt = boff[b + 1] - lo;
    const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
    if (n0 >= cnt) return;
    const double* Wb = W + (size_t)b * N * N;
    ib_d4 acc[2][2];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = ib_d4{0.0, 0.0, 0.0, 0.0};
    int mrow[2], col[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        mrow[t] = min(m0 + 16 * t + lc, N - 1);
        col[t] = bidx[lo + min(n0 + 16 * t + lc, cnt - 1)];
    }
    constexpr int U = 4;
    for (int k0 = 0; k0 < N; k0 += 4 * U) {
        double av[U][2], bv[U][2];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + 4 * u + lq;
            const double kmask = k < N ? 1.0 : 0.0;
            const int kc = min(k, N - 1);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                av[u][t] = Wb[(size_t)mrow[t] * N + kc];
                bv[u][t] = X0[(size_t)kc * M + col[t]] * kmask;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][ti], bv[u][tj], acc[ti][tj], 0, 0, 0);
    }
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int row = m0 + 16 * ti + lq + 4 * gq, cc = n0 + 16 * tj + lc;
                if (row < N && cc < cnt) {
                    const int idx = bidx[lo + cc];
                    E[(size_t)row * M + idx] = x0[idx] + acc[ti][tj][gq];
                }
            }
}

}  // namespace

struct hm_iles {
    hm_ctx* ctx = nullptr;
    int N = 0, M = 0, n_obs = 0, B = 0, resident = 0;
    double cutoff = 1e-2;
    size_t ws_stride = 0, lds = 0;
    DevBuf boff, bidx, taper_b, W, X0, x0, E, S, D, ws, flag;
    // blocked form of the step (k_ib_*)
    int blocked = 0, lda = 0, max_cnt = 0;
    std::vector<int> nloc_host;
    DevBuf cvec, jj, nloc, ipiv, dead;
};
static bool ib_fits(int N) { return N >= 2 && N <= 1024; }  // a panel of N x 16 (+ padding) in LDS, one row per thread; the back substitution's blocks

extern "C" void hm_iles_destroy(hm_iles* p) {
    if (!p) return;
    (void)hipSetDevice(p->ctx->device);
    (void)hipStreamSynchronize(p->ctx->stream);
    DevBuf* bufs[] = {&p->boff, &p->bidx, &p->taper_b, &p->W, &p->X0, &p->x0, &p->E, &p->S, &p->D, &p->ws, &p->flag,
                      &p->cvec, &p->jj, &p->nloc, &p->ipiv, &p->dead};
    for (DevBuf* b : bufs) hm_dev_free(*b);
    delete p;
}

extern "C" int hm_iles_create(hm_ctx* ctx, int N, int M, int n_obs, int B, const int* batch_offsets, const int* batch_index,
                              const double* taper_b, double cutoff, const double* prior_ens, hm_iles** out) {
    HM_REQUIRE(ctx && batch_offsets && batch_index && taper_b && prior_ens && out, "hm_iles_create: NULL argument");
    HM_REQUIRE(N >= 2 && M >= 1 && n_obs >= 1 && B >= 1 && B <= M, "hm_iles_create: bad sizes");
    HM_REQUIRE(n_obs <= NT, "hm_iles_create: n_obs = %d exceeds %d", n_obs, NT);
    HM_REQUIRE(batch_offsets[0] == 0 && batch_offsets[B] == M, "hm_iles_create: the batches must partition the %d state elements", M);
    const size_t lds = ((size_t)n_obs * (n_obs + 1) / 2 + n_obs + NT + N) * 8 + ((size_t)NT + n_obs + 4) * 4;
    HM_REQUIRE(lds <= 160 * 1024, "hm_iles_create: N = %d, n_obs = %d need %zu bytes of LDS per batch (limit 160 KB)", N, n_obs, lds);
    HM_HIP(hipSetDevice(ctx->device));
    hm_iles* p = new hm_iles();
    p->ctx = ctx; p->N = N; p->M = M; p->n_obs = n_obs; p->B = B; p->cutoff = cutoff; p->lds = lds;
    p->lda = (N + n_obs + 1) & ~1;
    p->ws_stride = std::max((size_t)N * N + (size_t)4 * N * n_obs + (size_t)n_obs * n_obs, ib_ws_doubles(N, n_obs, p->lda));
    p->blocked = N >= 256 && ib_fits(N);  // large ensembles: the blocked form (hm_iles_set_option "blocked" overrides)
    for (int b = 0; b < B; ++b) p->max_cnt = std::max(p->max_cnt, batch_offsets[b + 1] - batch_offsets[b]);
    // workgroups of one launch = batches processed side by side, each with its own workspace: a few per CU's worth, bounded by 8 GB
    size_t res = std::min<size_t>((size_t)B, (size_t)4 * ctx->num_cu);
    while (res > 1 && res * p->ws_stride * 8 > ((size_t)8 << 30)) res /= 2;
    p->resident = (int)res;
    int rc = 0;
#define ALLOC(buf, bytes) do { rc = hm_dev_alloc(p->buf, (bytes)); if (rc) { hm_iles_destroy(p); return rc; } } while (0)
    ALLOC(boff, (size_t)(B + 1) * 4); ALLOC(bidx, (size_t)M * 4); ALLOC(taper_b, (size_t)B * n_obs * 8);
    ALLOC(W, (size_t)B * N * N * 8); ALLOC(X0, (size_t)N * M * 8); ALLOC(x0, (size_t)M * 8); ALLOC(E, (size_t)N * M * 8);
    ALLOC(S, (size_t)N * n_obs * 8); ALLOC(D, (size_t)N * n_obs * 8); ALLOC(ws, res * p->ws_stride * 8); ALLOC(flag, 16);
    ALLOC(cvec, (size_t)B * n_obs * 8); ALLOC(jj, (size_t)B * n_obs * 4); ALLOC(nloc, (size_t)B * 4); ALLOC(ipiv, res * (size_t)N * 4); ALLOC(dead, res * 4);
#undef ALLOC
    hipStream_t s = ctx->stream;
    HM_HIP(hipMemcpyAsync(p->boff.p, batch_offsets, (size_t)(B + 1) * 4, hipMemcpyHostToDevice, s));
    HM_HIP(hipMemcpyAsync(p->bidx.p, batch_index, (size_t)M * 4, hipMemcpyHostToDevice, s));
    HM_HIP(hipMemcpyAsync(p->taper_b.p, taper_b, (size_t)B * n_obs * 8, hipMemcpyHostToDevice, s));
    HM_HIP(hipMemcpyAsync(p->E.p, prior_ens, (size_t)N * M * 8, hipMemcpyHostToDevice, s));
    HM_HIP(hipMemsetAsync(p->flag.p, 0, 16, s));
    hipLaunchKernelGGL(k_iles_center, dim3((M + 255) / 256), dim3(256), 0, s, N, M, (const double*)p->E.p, (double*)p->X0.p, (double*)p->x0.p);
    hipLaunchKernelGGL(k_iles_identity, dim3(2048), dim3(256), 0, s, N, (long long)B * N * N, (double*)p->W.p);
    HM_HIP(hipGetLastError());
    HM_HIP(hipFuncSetAttribute((const void*)k_iles_batch, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HM_HIP(hipMemsetAsync(p->dead.p, 0, res * 4, s));
    hipLaunchKernelGGL(k_ib_select, dim3(B), dim3(64), 0, s, n_obs, (const double*)p->taper_b.p, cutoff, (double*)p->cvec.p, (int*)p->jj.p, (int*)p->nloc.p);
    HM_HIP(hipGetLastError());
    p->nloc_host.resize(B);
    HM_HIP(hipMemcpyAsync(p->nloc_host.data(), p->nloc.p, (size_t)B * 4, hipMemcpyDeviceToHost, s));
    HM_HIP(hipFuncSetAttribute((const void*)k_ib_backsolve, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HM_HIP(hipFuncSetAttribute((const void*)k_ib_cinv, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HM_HIP(hipStreamSynchronize(s));
    *out = p;
    return 0;
}

// E = x0 + W_b X0 per batch, to a host buffer (N x M) and/or left on the device (hm_iles_device_ptr "E").
extern "C" int hm_iles_compose(hm_iles* p, double* E_out) {
    HM_REQUIRE(p, "hm_iles_compose: NULL plan");
    HM_HIP(hipSetDevice(p->ctx->device));
    hipStream_t s = p->ctx->stream;
    if (p->max_cnt >= 16)  // batches of many elements: tiles of 32 members x 32 elements on the matrix cores
        hipLaunchKernelGGL(k_iles_compose_mfma, dim3((p->max_cnt + 31) / 32, (p->N + 31) / 32, p->B), dim3(64), 0, s, p->N, p->M,
                           (const int*)p->boff.p, (const int*)p->bidx.p, (const double*)p->W.p, (const double*)p->X0.p, (const double*)p->x0.p,
                           (double*)p->E.p);
    else
        hipLaunchKernelGGL(k_iles_compose, dim3(p->B), dim3(256), 0, s, p->N, p->M, (const int*)p->boff.p, (const int*)p->bidx.p,
                           (const double*)p->W.p, (const double*)p->X0.p, (const double*)p->x0.p, (double*)p->E.p);
    HM_HIP(hipGetLastError());
    if (E_out) HM_HIP(hipMemcpyAsync(E_out, p->E.p, (size_t)p->N * p->M * 8, hipMemcpyDeviceToHost, s));
    HM_HIP(hipStreamSynchronize(s));
    return 0;
}

// The blocked form of one step, every resident group of batches side by side (see the kernels above).
static int iles_step_blocked(hm_iles* p, double xstep) {
    hipStream_t s = p->ctx->stream;
    const int N = p->N, n = p->n_obs, lda = p->lda, n16 = ib_n16(n);
    IbArgs a{N, n, lda, 0, xstep, (double*)p->W.p, (double*)p->ws.p, p->ws_stride, (const int*)p->nloc.p, (const int*)p->jj.p,
             (const double*)p->cvec.p, (const double*)p->S.p, (const double*)p->D.p, (int*)p->ipiv.p, (int*)p->dead.p, (int*)p->flag.p};
    for (int b0 = 0; b0 < p->B; b0 += p->resident) {
        a.b0 = b0;
        const int G = std::min(p->resident, p->B - b0);
        int nlmax = 0;
        for (int g = 0; g < G; ++g) nlmax = std::max(nlmax, p->nloc_host[b0 + g]);
        if (nlmax == 0) continue;
        hipLaunchKernelGGL(k_ib_fill, dim3(N, G), dim3(256), 0, s, a);
        for (int kb = 0; kb < N; kb += IB_PB) {
            const int pw = std::min(IB_PB, N - kb), ncols = N + nlmax - kb - pw, nrows = N - kb - pw;
            hipLaunchKernelGGL(k_ib_panel, dim3(G), dim3(NT), 0, s, a, kb);
            hipLaunchKernelGGL(k_ib_swap_trsm, dim3((ncols + 255) / 256, G), dim3(256), 0, s, a, kb);
            if (nrows > 0) hipLaunchKernelGGL(k_ib_update, dim3((ncols + 63) / 64, (nrows + 63) / 64, G), dim3(256), 0, s, a, kb);
        }
        hipLaunchKernelGGL(k_ib_backsolve, dim3((nlmax + IB_BS_ROWS - 1) / IB_BS_ROWS, G), dim3(64 * IB_BS_ROWS), (size_t)2 * IB_BS_ROWS * N * 8, s, a);
        hipLaunchKernelGGL(k_ib_means, dim3(G), dim3(256), 0, s, a);
        hipLaunchKernelGGL(k_ib_y0, dim3(N, G), dim3(256), 0, s, a);
        HM_HIP(hipGetLastError());
        int rc = 0;
        auto ws_of = [&](int g) { return ib_ws((double*)p->ws.p + (size_t)g * p->ws_stride, N, n, lda); };
        const int tn = (nlmax + 31) / 32, tN = (N + 31) / 32;
        hipLaunchKernelGGL(k_ib_gemm<0>, dim3(tn, tn, G), dim3(64), 0, s, a);  // C = Y0^T Y0
        // C^-1: the matrix-core inverse of the analysis step (one workgroup, ~50 us per matrix) while there are few batches, the packed
        // Cholesky of one workgroup per batch side by side when there are many (or n_obs > 256)
        if (n16 <= 256 && G <= 16) {
            for (int g = 0; g < G; ++g) {
                if (!p->nloc_host[b0 + g]) continue;
                const IbWs w = ws_of(g);
                if ((rc = spd_inverse_mfma(s, w.C, 0, n16, (double)(N - 1), w.Cinv, (int*)p->flag.p, nullptr, 0.0, nullptr, 0.0)) > 0) return rc;
            }
        } else {
            hipLaunchKernelGGL(k_ib_cinv, dim3(G), dim3(NT), ((size_t)n * (n + 1) / 2 + 8) * 8, s, a);
        }
        hipLaunchKernelGGL(k_ib_gemm<1>, dim3(tn, tN, G), dim3(64), 0, s, a);  // R = W Y0 (then D_b - Y0 + R)
        hipLaunchKernelGGL(k_ib_rfix, dim3(N, G), dim3(256), 0, s, a);
        hipLaunchKernelGGL(k_ib_gemm<2>, dim3(tn, tN, G), dim3(64), 0, s, a);  // T = R C^-1
        hipLaunchKernelGGL(k_ib_gemm<3>, dim3(tN, tN, G), dim3(64), 0, s, a);  // T Y0^T into the Aug area (N x N)
        hipLaunchKernelGGL(k_ib_wupdate, dim3(N, G), dim3(256), 0, s, a);
        HM_HIP(hipGetLastError());
    }
    return 0;
}

// "blocked": 1 = the blocked form of the step (needs N <= 1024), 0 = one workgroup per batch; default: blocked from N = 256 on
extern "C" int hm_iles_set_option(hm_iles* p, const char* name, int value) {
    HM_REQUIRE(p && name, "hm_iles_set_option: NULL argument");
    if (std::string(name) == "blocked") {
        HM_REQUIRE(!value || ib_fits(p->N), "hm_iles_set_option: the blocked form holds a panel of N x 16 in LDS: N = %d is too large", p->N);
        p->blocked = value != 0;
        return 0;
    }
    hm_set_error("hm_iles_set_option: unknown option '%s'", name);
    return 2;
}

// One Gauss-Newton step of every batch's weight matrix from S = center(Eo decorr), D = (obs - Eo - perturbs) decorr (host, N x n_obs).
extern "C" int hm_iles_step(hm_iles* p, const double* S, const double* D, double xstep) {
    HM_REQUIRE(p && S && D, "hm_iles_step: NULL argument");
    HM_HIP(hipSetDevice(p->ctx->device));
    hipStream_t s = p->ctx->stream;
    const size_t nb = (size_t)p->N * p->n_obs * 8;
    HM_HIP(hipMemcpyAsync(p->S.p, S, nb, hipMemcpyHostToDevice, s));
    HM_HIP(hipMemcpyAsync(p->D.p, D, nb, hipMemcpyHostToDevice, s));
    if (p->blocked) {
        int rc = iles_step_blocked(p, xstep);
        if (rc) return rc;
    } else {
        IlesArgs a{p->N, p->n_obs, p->B, 0, (const double*)p->taper_b.p, p->cutoff, (const double*)p->S.p, (const double*)p->D.p,
                   (double*)p->W.p, (double*)p->ws.p, p->ws_stride, xstep, (int*)p->flag.p};
        for (int b0 = 0; b0 < p->B; b0 += p->resident) {
            a.b0 = b0;
            const int nb_launch = std::min(p->resident, p->B - b0);
            hipLaunchKernelGGL(k_iles_batch, dim3(nb_launch), dim3(NT), p->lds, s, a);
        }
    }
    HM_HIP(hipGetLastError());
    HM_HIP(hipStreamSynchronize(s));
    int flag = 0;
    HM_HIP(hipMemcpy(&flag, p->flag.p, 4, hipMemcpyDeviceToHost));
    if (flag) {
        HM_HIP(hipMemset(p->flag.p, 0, 16));
        hm_set_error("hm_iles_step: %s", (flag & 2) ? "a batch's weight matrix is singular (or holds NaN)" : "non-positive pivot in Y0^T Y0 + (N-1) I (NaN/Inf in the inputs?)");
        return 4;
    }
    return 0;
}

extern "C" int hm_iles_get_weights(hm_iles* p, int batch, double* W_out) {
    HM_REQUIRE(p && W_out && batch >= 0 && batch < p->B, "hm_iles_get_weights: bad arguments");
    HM_HIP(hipSetDevice(p->ctx->device));
    HM_HIP(hipStreamSynchronize(p->ctx->stream));
    HM_HIP(hipMemcpy(W_out, (const double*)p->W.p + (size_t)batch * p->N * p->N, (size_t)p->N * p->N * 8, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" void* hm_iles_device_ptr(hm_iles* p, const char* name) {
    if (!p || !name) return nullptr;
    std::string s(name);
    if (s == "E") return p->E.p;
    if (s == "W") return p->W.p;
    if (s == "X0") return p->X0.p;
    return nullptr;
}
