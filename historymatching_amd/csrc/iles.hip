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
// One workgroup per batch; the N x N objects live in HBM/L2 (a workspace per resident workgroup), the n_loc x n_loc Cholesky factor
// in LDS.  This is the callers' subspace algebra, sized by N^2 n_loc per batch -- written for clarity and
// batch parallelism, not for a roofline.
#include "common.h"

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
    // ---- Cholesky in place (right-looking)
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
    if (bad) {
        if (tid == 0) atomicOr(a.flag, 1);
        return;
    }
    // ---- C^-1 column by column: thread q solves L L^T x = e_q; x lives in Cinv[i*nl + q] (coalesced over q)
    if (tid < nl) {
        const int q = tid;
        for (int i = 0; i < nl; ++i) {
            double s = i == q ? 1.0 : 0.0;
            const int ib = i * (i + 1) / 2;
            for (int j = 0; j < i; ++j) s = fma(-L[ib + j], Cinv[(size_t)j * nl + q], s);
            Cinv[(size_t)i * nl + q] = s / L[ib + i];
        }
        for (int i = nl - 1; i >= 0; --i) {
            double s = Cinv[(size_t)i * nl + q];
            for (int j = i + 1; j < nl; ++j) s = fma(-L[j * (j + 1) / 2 + i], Cinv[(size_t)j * nl + q], s);
            Cinv[(size_t)i * nl + q] = s / L[i * (i + 1) / 2 + i];
        }
    }
    __syncthreads();
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

}  // namespace

struct hm_iles {
    hm_ctx* ctx = nullptr;
    int N = 0, M = 0, n_obs = 0, B = 0, resident = 0;
    double cutoff = 1e-2;
    size_t ws_stride = 0, lds = 0;
    DevBuf boff, bidx, taper_b, W, X0, x0, E, S, D, ws, flag;
};

extern "C" void hm_iles_destroy(hm_iles* p) {
    if (!p) return;
    (void)hipSetDevice(p->ctx->device);
    (void)hipStreamSynchronize(p->ctx->stream);
    DevBuf* bufs[] = {&p->boff, &p->bidx, &p->taper_b, &p->W, &p->X0, &p->x0, &p->E, &p->S, &p->D, &p->ws, &p->flag};
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
    p->ws_stride = (size_t)N * N + (size_t)4 * N * n_obs + (size_t)n_obs * n_obs;
    // workgroups of one launch = batches processed side by side, each with its own workspace: a few per CU's worth, bounded by 8 GB
    size_t res = std::min<size_t>((size_t)B, (size_t)4 * ctx->num_cu);
    while (res > 1 && res * p->ws_stride * 8 > ((size_t)8 << 30)) res /= 2;
    p->resident = (int)res;
    int rc = 0;
#define ALLOC(buf, bytes) do { rc = hm_dev_alloc(p->buf, (bytes)); if (rc) { hm_iles_destroy(p); return rc; } } while (0)
    ALLOC(boff, (size_t)(B + 1) * 4); ALLOC(bidx, (size_t)M * 4); ALLOC(taper_b, (size_t)B * n_obs * 8);
    ALLOC(W, (size_t)B * N * N * 8); ALLOC(X0, (size_t)N * M * 8); ALLOC(x0, (size_t)M * 8); ALLOC(E, (size_t)N * M * 8);
    ALLOC(S, (size_t)N * n_obs * 8); ALLOC(D, (size_t)N * n_obs * 8); ALLOC(ws, res * p->ws_stride * 8); ALLOC(flag, 16);
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
    HM_HIP(hipStreamSynchronize(s));
    *out = p;
    return 0;
}

// E = x0 + W_b X0 per batch, to a host buffer (N x M) and/or left on the device (hm_iles_device_ptr "E").
extern "C" int hm_iles_compose(hm_iles* p, double* E_out) {
    HM_REQUIRE(p, "hm_iles_compose: NULL plan");
    HM_HIP(hipSetDevice(p->ctx->device));
    hipStream_t s = p->ctx->stream;
    hipLaunchKernelGGL(k_iles_compose, dim3(p->B), dim3(256), 0, s, p->N, p->M, (const int*)p->boff.p, (const int*)p->bidx.p,
                       (const double*)p->W.p, (const double*)p->X0.p, (const double*)p->x0.p, (double*)p->E.p);
    HM_HIP(hipGetLastError());
    if (E_out) HM_HIP(hipMemcpyAsync(E_out, p->E.p, (size_t)p->N * p->M * 8, hipMemcpyDeviceToHost, s));
    HM_HIP(hipStreamSynchronize(s));
    return 0;
}

// One Gauss-Newton step of every batch's weight matrix from S = center(Eo decorr), D = (obs - Eo - perturbs) decorr (host, N x n_obs).
extern "C" int hm_iles_step(hm_iles* p, const double* S, const double* D, double xstep) {
    HM_REQUIRE(p && S && D, "hm_iles_step: NULL argument");
    HM_HIP(hipSetDevice(p->ctx->device));
    hipStream_t s = p->ctx->stream;
    const size_t nb = (size_t)p->N * p->n_obs * 8;
    HM_HIP(hipMemcpyAsync(p->S.p, S, nb, hipMemcpyHostToDevice, s));
    HM_HIP(hipMemcpyAsync(p->D.p, D, nb, hipMemcpyHostToDevice, s));
    IlesArgs a{p->N, p->n_obs, p->B, 0, (const double*)p->taper_b.p, p->cutoff, (const double*)p->S.p, (const double*)p->D.p,
               (double*)p->W.p, (double*)p->ws.p, p->ws_stride, xstep, (int*)p->flag.p};
    for (int b0 = 0; b0 < p->B; b0 += p->resident) {
        a.b0 = b0;
        const int nb_launch = std::min(p->resident, p->B - b0);
        hipLaunchKernelGGL(k_iles_batch, dim3(nb_launch), dim3(NT), p->lds, s, a);
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
