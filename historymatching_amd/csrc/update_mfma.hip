// update_mfma.hip -- fp32 matrix-core (v_mfma_f32_32x32x2_f32) kernels for the two contractions of the
// ensemble-smoother update that run over the state dimension M (everything else is N x n_obs sized and stays fp64):
//
//   k_gxt_mfma    Gxt (M x n_obs) = (E - mean)^T S          reads E exactly once (the HBM stream), K = N members
//   k_apply_mfma  E_out (N x M)   = E + A^T-operand  * B      A = (D C^-1) or D (n_obs x N, transposed copy),
//                                                           B = Gx or W (n_obs x M), K = n_obs
//
// Reference: the products in ens_update0 / ens_update0_loc, notebooks/HistoryMatch.py:581-586, 789-793, evaluated
// in the minimum-flop association.  v_mfma_f32_32x32x2_f32 is exact fp32 (k-ordered fmaf chain), so
// the fp32 error model of the tests (1e-4 of the max increment) is unchanged.
// Operand maps: A operand lane l holds A[l&31][l>>5], B operand B[l>>5][l&31]; C/D: col = l&31,
// row = (reg&3) + 8*(reg>>2) + 4*(l>>5)  (cdna_hip_programming.md section 3).
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// out[c][r] = (TO) in[r][c]
template <typename TI, typename TO>
__global__ void k_transpose(const TI* __restrict__ in, TO* __restrict__ out, int rows, int cols) {
    __shared__ TO tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 8 rows per pass
    for (int dy = ty; dy < 32; dy += 8) {
        int r = r0 + dy, c = c0 + tx;
        tile[dy][tx] = (r < rows && c < cols) ? (TO)in[(size_t)r * cols + c] : TO(0);
    }
    __syncthreads();
    for (int dy = ty; dy < 32; dy += 8) {
        int c = c0 + dy, r = r0 + tx;
        if (c < cols && r < rows) out[(size_t)c * rows + r] = tile[tx][dy];
    }
}

// Gxt[i][j] = sum_k (E[k][i] - mean_i) S[k][j].  One workgroup = 32 state elements x all NJ*32 observations; its 4
// waves split the member dimension K (interleaved pairs), keep 4 pairs of loads in flight each, and reduce their
// partial accumulators through LDS.  E is read exactly once (the HBM stream); S (N x n_obs) stays L2-resident.
template <int NJ>
__global__ __launch_bounds__(256) void k_gxt_mfma(int N, int M, int n_obs, const float* __restrict__ E,
                                                  const float* __restrict__ colsum, float inv_n,
                                                  const float* __restrict__ S, float* __restrict__ Gxt) {
    extern __shared__ __attribute__((aligned(16))) float red[];  // [4 waves][NJ*16 regs][64 lanes]
    constexpr int UN = 4;  // k-pairs in flight per wave (operands come straight from global/L2; 8 measured slower)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i0 = blockIdx.x * 32;
    const int il = lane & 31, kh = lane >> 5;
    const int i = min(i0 + il, M - 1);
    // any shift c_i leaves (E - c)^T S = X^T S unchanged when the columns of S sum to zero over ALL rows (S = centred
    // obs * decorr); single-rank plans pass colsum = nullptr and shift by the first member instead of the exact mean
    // (same conditioning, no extra pass over E)
    const float mean = colsum ? colsum[i] * inv_n : E[i];
    f32x16 acc[NJ];
#pragma unroll
    for (int t = 0; t < NJ; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    const int npairs = (N + 1) / 2;
    for (int kp = w; kp < npairs; kp += 4 * UN) {
        float b[UN], a[UN][NJ];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int k = 2 * (kp + 4 * u) + kh;
            const bool ok = k < N;
            const int kc = ok ? k : N - 1;
            b[u] = ok ? E[(size_t)kc * M + i] - mean : 0.0f;
            const float* Srow = S + (size_t)kc * n_obs + il;
#pragma unroll
            for (int t = 0; t < NJ; ++t) a[u][t] = ok ? Srow[32 * t] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int t = 0; t < NJ; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][t], b[u], acc[t], 0, 0, 0);
    }
    // cross-wave reduction of the K split
#pragma unroll
    for (int t = 0; t < NJ; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((w * NJ + t) * 16 + r) * 64 + lane] = acc[t][r];
    __syncthreads();
    if (i0 + il < M) {
        float* out = Gxt + (size_t)(i0 + il) * n_obs;
        for (int t = w; t < NJ; t += 4) {  // wave w finishes tiles t = w, w+4, ...
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = 4 * rq + q;
                    v[q] = (red[((0 * NJ + t) * 16 + r) * 64 + lane] + red[((1 * NJ + t) * 16 + r) * 64 + lane]) +
                           (red[((2 * NJ + t) * 16 + r) * 64 + lane] + red[((3 * NJ + t) * 16 + r) * 64 + lane]);
                }
                *reinterpret_cast<float4*>(out + 32 * t + 8 * rq + 4 * kh) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }
}

// E_out[n][i] = E[n][i] + sum_j At[j][n] B[j][i];  workgroup = 4 waves = 128 members x 128 state elements,
// B tile (n_obs x 128) staged once in LDS, every wave 32 members x 128 state elements (4 accumulator tiles).
__global__ __launch_bounds__(256) void k_apply_mfma(int N, int M, int n_obs, const float* __restrict__ E,
                                                    const float* __restrict__ At, const float* __restrict__ B,
                                                    float* __restrict__ Eout) {
    extern __shared__ __attribute__((aligned(16))) float Bs[];  // n_obs x 128
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i0 = blockIdx.x * 128, n0 = blockIdx.y * 128 + 32 * w;
    for (int e = threadIdx.x; e < n_obs * 32; e += 256) {  // float4 granules
        const int j = e >> 5, c4 = (e & 31) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i0 + c4 + 3 < M) v = *reinterpret_cast<const float4*>(B + (size_t)j * M + i0 + c4);
        else
            for (int q = 0; q < 4; ++q)
                if (i0 + c4 + q < M) (&v.x)[q] = B[(size_t)j * M + i0 + c4 + q];
        *reinterpret_cast<float4*>(Bs + j * 128 + c4) = v;
    }
    __syncthreads();
    const int nl = lane & 31, kh = lane >> 5;
    const int n = min(n0 + nl, N - 1);
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    const int npairs = (n_obs + 1) / 2;

    // the A operand comes straight from global memory (At is L2 resident): fetch UA k-pairs ahead of the MFMAs that
    // consume them, otherwise every group of 4 MFMAs waits for one load round trip
    constexpr int UA = 16;
    for (int jp0 = 0; jp0 < npairs; jp0 += UA) {
        float av[UA];
#pragma unroll
        for (int u = 0; u < UA; ++u) {
            const int j = 2 * (jp0 + u) + kh;
            av[u] = (j < n_obs) ? At[(size_t)j * N + n] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < UA; ++u) {
            const int j = 2 * (jp0 + u) + kh;
            const bool ok = j < n_obs;
            const float* brow = Bs + (ok ? j : 0) * 128 + nl;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float b = ok ? brow[32 * t] : 0.0f;
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], b, acc[t], 0, 0, 0);
            }
        }
    }
    // C/D: col = lane&31 -> state element, rows -> member
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int i = i0 + 32 * t + nl;
        if (i >= M) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int nn = n0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (nn < N) Eout[(size_t)nn * M + i] = E[(size_t)nn * M + i] + acc[t][r];
        }
    }
}


// ------------------------------------------------------------------------------------------------------------
// Second generation of the two contractions (single-rank fused run, hm_upd_run):
//
// k_gxt_lds:   Gx (n_obs x M) = S^T (E - c)      one workgroup = 64 state elements x all n_obs observations, K = N members.
//   E and S chunks of KC = 16 members are staged through LDS (float4 global loads one chunk ahead, in registers, while the
//   current chunk feeds the MFMAs), the 4 waves = 2 state halves x 2 interleaved halves of the k-pairs, partial
//   accumulators of the two k-halves added through LDS at the end.  Writes Gx directly in the (n_obs x M) layout the
//   apply kernel consumes (accumulator columns = state elements: 128-byte rows), so no transpose pass.
// k_apply_lds: E_out = E + A^T-operand * Gx      one workgroup = 128 members x 64 state elements: Gx tile (n_obs x 64) in
//   LDS (40 KB at n_obs = 160: 3 workgroups per CU overlap staging, MFMAs and the E read/modify/write), A operand
//   fetched 16 k-pairs ahead from L2, the E tile loaded before the MFMA loop.
// ------------------------------------------------------------------------------------------------------------
template <int NJ, int KC, int SH = 2, int DEPTH = 1>
__global__ __launch_bounds__(256 * SH, (KC == 64 || SH == 1) ? 2 : 4) void k_gxt_lds(int N, int M, int n_obs, const float* __restrict__ E,
                                                 const float* __restrict__ colsum, float inv_n,
                                                 const float* __restrict__ S, float* __restrict__ Gx) {
    // 4 SH waves = SH state halves (32 state elements each) x 4 interleaved quarters of the k-pairs.  SH = 2: one 8-wave
    // workgroup per CU (two waves per SIMD: one wave's LDS/barrier stalls are covered by the other's MFMAs); SH = 1: two
    // unsynchronised 4-wave workgroups per CU.  KC members per LDS chunk = per barrier.
    constexpr int NO = 32 * NJ, NKH = 4, NT = 256 * SH, SW = 32 * SH;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    // [2][KC][SW] E chunks, [2][KC][NO] S chunks; the final reduction buffers alias them
    float* Eb = sm;
    float* Sb = sm + 2 * KC * SW;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int sh = SH == 2 ? (w & 1) : 0, kh = SH == 2 ? (w >> 1) : w;
    const int il = lane & 31, kq = lane >> 5;
    const int i0 = blockIdx.x * SW;
    // staging roles: E chunk = KC rows x SW/4 float4 (EPT per thread); S chunk = KC rows x NO/4 float4
    constexpr int EPT = KC * (SW / 4) / NT, ERS = NT / (SW / 4);  // = KC / 32 float4 per thread; rows covered per pass = 32
    const int er = tid / (SW / 4), ec = (tid % (SW / 4)) * 4;
    const int ei = min(i0 + ec, M - 4);  // M % 4 == 0 (host-checked); columns past M are never stored
    float4 shift;
    if (colsum) shift = make_float4(colsum[ei] * inv_n, colsum[ei + 1] * inv_n, colsum[ei + 2] * inv_n, colsum[ei + 3] * inv_n);
    else shift = *reinterpret_cast<const float4*>(E + ei);
    constexpr int SV = KC * NO / 4;           // float4 per S chunk
    constexpr int SPT = (SV + NT - 1) / NT;   // per thread
    // DEPTH = 2: the global loads run TWO chunks ahead of the MFMAs (a second register set; one chunk = 2.1 us of matrix work at
    // KC = 64, about the latency of an HBM miss under load)
    float4 ereg[EPT], sreg[SPT], ereg2[DEPTH == 2 ? EPT : 1], sreg2[DEPTH == 2 ? SPT : 1];
    auto fetch_into = [&](int k0, float4* ereg, float4* sreg) {
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int k = k0 + er + ERS * q;
            ereg[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < N) {
                const float4 v = *reinterpret_cast<const float4*>(E + (size_t)k * M + ei);
                ereg[q] = make_float4(v.x - shift.x, v.y - shift.y, v.z - shift.z, v.w - shift.w);
            }
        }
#pragma unroll
        for (int q = 0; q < SPT; ++q) {
            const int e = tid + NT * q;
            const int r = e / (NO / 4), c4 = (e % (NO / 4)) * 4;
            sreg[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < SV && k0 + r < N) sreg[q] = *reinterpret_cast<const float4*>(S + (size_t)(k0 + r) * n_obs + c4);
        }
    };
    auto fetch = [&](int k0) { fetch_into(k0, ereg, sreg); };
    auto stash = [&](int buf) {
#pragma unroll
        for (int q = 0; q < EPT; ++q) *reinterpret_cast<float4*>(Eb + (buf * KC + er + ERS * q) * SW + ec) = ereg[q];
#pragma unroll
        for (int q = 0; q < SPT; ++q) {
            const int e = tid + NT * q;
            if (e < SV) *reinterpret_cast<float4*>(Sb + buf * KC * NO + e * 4) = sreg[q];
        }
    };
    f32x16 acc[NJ];
#pragma unroll
    for (int t = 0; t < NJ; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    const int nchunks = (N + KC - 1) / KC;
    const bool dbg_nostage = inv_n < 0.0f;  // diagnostic (wrong results): every chunk re-uses the first one, no loads, no stash
    fetch(0);
    stash(0);
    if (DEPTH == 2 && nchunks > 1) fetch(KC);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int buf = dbg_nostage ? 0 : (c & 1);
        if (DEPTH == 2) {
            if (c + 2 < nchunks) fetch_into((c + 2) * KC, ereg2, sreg2);
        } else if (c + 1 < nchunks && !dbg_nostage) fetch((c + 1) * KC);
        const float* eb = Eb + buf * KC * SW + 32 * sh + il;
        const float* sb = Sb + buf * KC * NO + il;
        // this wave's k-pairs of the chunk: pair p = NKH q + kh  (rows 2p, 2p+1).  The LDS operands of pair q + 1 are requested
        // BEFORE the MFMAs of pair q are issued (left to itself the compiler emits read -> wait -> two MFMAs -> read ...: every
        // MFMA pair then waits out the LDS latency of its own operands and the matrix pipe idles half the time).
        constexpr int NQ = KC / 2 / NKH;
        float a_cur[NJ], a_nxt[NJ], b_cur, b_nxt;
        {
            const int row = 2 * kh + kq;
            b_cur = eb[row * SW];
#pragma unroll
            for (int t = 0; t < NJ; ++t) a_cur[t] = sb[row * NO + 32 * t];
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (q + 1 < NQ) {
                const int row = 2 * (NKH * (q + 1) + kh) + kq;
                b_nxt = eb[row * SW];
#pragma unroll
                for (int t = 0; t < NJ; ++t) a_nxt[t] = sb[row * NO + 32 * t];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NJ; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[t], b_cur, acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (q + 1 < NQ) {
                b_cur = b_nxt;
#pragma unroll
                for (int t = 0; t < NJ; ++t) a_cur[t] = a_nxt[t];
            }
        }
        if (c + 1 < nchunks && !dbg_nostage) stash(buf ^ 1);
        if (!dbg_nostage) __syncthreads();
        if (DEPTH == 2) {
#pragma unroll
            for (int q = 0; q < EPT; ++q) ereg[q] = ereg2[q];
#pragma unroll
            for (int q = 0; q < SPT; ++q) sreg[q] = sreg2[q];
        }
    }
    // fixed-order tree over the 4 k-quarters through LDS: (0 + 1) and (2 + 3), then (0+1) + (2+3); buffers
    // [2 publishers][SH][NJ][16][64] alias the chunks
    float* red = sm;
    auto slot = [&](int pub, int t, int r) { return red + (((pub * SH + sh) * NJ + t) * 16 + r) * 64 + lane; };
    if (kh & 1) {
#pragma unroll
        for (int t = 0; t < NJ; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) *slot(kh >> 1, t, r) = acc[t][r];
    }
    __syncthreads();
    if (!(kh & 1)) {
#pragma unroll
        for (int t = 0; t < NJ; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] += *slot(kh >> 1, t, r);
    }
    __syncthreads();
    if (kh == 2) {
#pragma unroll
        for (int t = 0; t < NJ; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) *slot(0, t, r) = acc[t][r];
    }
    __syncthreads();
    if (kh == 0) {
        const int i = i0 + 32 * sh + il;
        if (i < M) {
#pragma unroll
            for (int t = 0; t < NJ; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int j = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * kq;  // accumulator row = observation
                    Gx[(size_t)j * M + i] = acc[t][r] + *slot(0, t, r);
                }
        }
    }
}

__global__ __launch_bounds__(256) void k_apply_lds(int N, int M, int n_obs, const float* __restrict__ E,
                                                   const float* __restrict__ At, const float* __restrict__ Gx,
                                                   float* __restrict__ Eout) {
    extern __shared__ __attribute__((aligned(16))) float Bs[];  // n_obs x 64
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i0 = blockIdx.x * 64, n0 = blockIdx.y * 128 + 32 * w;
    for (int e = threadIdx.x; e < n_obs * 16; e += 256) {  // float4 granules
        const int j = e >> 4, c4 = (e & 15) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i0 + c4 + 3 < M) v = *reinterpret_cast<const float4*>(Gx + (size_t)j * M + i0 + c4);
        *reinterpret_cast<float4*>(Bs + j * 64 + c4) = v;
    }
    const int nl = lane & 31, kh = lane >> 5;
    const int n = min(n0 + nl, N - 1);
    // the E tile of the epilogue is requested now and consumed after the MFMA loop
    float ev[2][16];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int nn = min(n0 + (r & 3) + 8 * (r >> 2) + 4 * kh, N - 1);
            const int i = min(i0 + 32 * t + nl, M - 1);
            ev[t][r] = E[(size_t)nn * M + i];
        }
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    __syncthreads();
    const int npairs = (n_obs + 1) / 2;
    constexpr int UA = 16;
    for (int jp0 = 0; jp0 < npairs; jp0 += UA) {
        float av[UA];
#pragma unroll
        for (int u = 0; u < UA; ++u) {
            const int j = min(2 * (jp0 + u) + kh, n_obs - 1);
            av[u] = At[(size_t)j * N + n];
        }
#pragma unroll
        for (int u = 0; u < UA; ++u) {
            const int j = 2 * (jp0 + u) + kh;
            const float a = j < n_obs ? av[u] : 0.0f;
            const float* brow = Bs + min(j, n_obs - 1) * 64 + nl;
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, brow[32 * t], acc[t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int i = i0 + 32 * t + nl;
        if (i >= M) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int nn = n0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (nn < N) Eout[(size_t)nn * M + i] = ev[t][r] + acc[t][r];
        }
    }
}

// k_gxt_dma: the first contraction with the chunks staged by LDS-DMA (global_load_lds_dwordx4: memory -> LDS without passing
// through registers, asynchronous, retired by the issuing wave's vmcnt).  Measured on k_gxt_lds: the matrix loop alone takes 48 us,
// register staging (global load -> VGPR -> ds_write_b128) + the barrier of every chunk another 20 us.  Same tiling (64 state
// elements x all observations per workgroup, 8 waves = 2 state halves x 4 k-quarters, KC members per chunk, double-buffered);
// one DMA piece = one wave instruction = 1 KB: 4 rows of the E chunk (256 B each) or 256 consecutive floats of the S chunk.
// The shift c (first member / exact mean) is subtracted when the E operand is read from LDS (one v_sub per 5 MFMAs).  Rows past
// N in the last chunk: sources clamped, the S rows zeroed in LDS before use.  Requires M % 64 == 0.
template <int NJ, int KC, int NKH, int NS = 2>
__global__ __launch_bounds__(128 * NKH, NKH / 2) void k_gxt_dma(int N, int M, int n_obs, const float* __restrict__ E,
                                                    const float* __restrict__ colsum, float inv_n,
                                                    const float* __restrict__ S, float* __restrict__ Gx) {
    constexpr int NO = 32 * NJ, SW = 64, NQ = KC / 2 / NKH, NWV = 2 * NKH, NT = 64 * NWV;  // NKH k-parts x 2 state halves
    constexpr int EP = KC / 4, SP = KC * NO / 256, NP = EP + SP;  // DMA pieces per chunk
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Eb = sm;                  // [2][KC][64]
    float* Sb = sm + 2 * KC * SW;    // [2][KC][NO]
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sh = w & 1, kh = w >> 1;
    const int il = lane & 31, kq = lane >> 5;
    const int i0 = blockIdx.x * SW;
    const int istate = i0 + 32 * sh + il;
    const float shift = colsum ? colsum[istate] * inv_n : E[istate];
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    auto stage = [&](int c, int buf) {
        const int k0 = c * KC;
#pragma unroll
        for (int q = 0; q < (NP + NWV - 1) / NWV; ++q) {
            const int piece = w + NWV * q;  // wave-uniform
            if (piece < EP) {
                const int row = min(k0 + 4 * piece + (lane >> 4), N - 1);
                const float* src = E + (size_t)row * M + i0 + (lane & 15) * 4;
                float* dst = Eb + (buf * KC + 4 * piece) * SW;
                __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)dst, 16, 0, 0);
            } else if (piece < NP) {
                const int sp = piece - EP;
                const size_t e = min((size_t)k0 * NO + (size_t)sp * 256 + lane * 4, (size_t)N * NO - 4);
                float* dst = Sb + buf * KC * NO + sp * 256;
                __builtin_amdgcn_global_load_lds((glb_ptr)(S + e), (lds_ptr)dst, 16, 0, 0);
            }
        }
    };
    f32x16 acc[NJ];
#pragma unroll
    for (int t = 0; t < NJ; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    const int nchunks = (N + KC - 1) / KC;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunks) stage(c + 1, buf ^ 1);
        if (c + 1 == nchunks && N % KC != 0) {  // rows past N: their S operands must be zero (the E rows are clamped copies)
            const int nvalid = N - c * KC;
            for (int e = tid; e < (KC - nvalid) * NO; e += NT) Sb[buf * KC * NO + nvalid * NO + e] = 0.0f;
            __syncthreads();
        }
        const float* eb = Eb + buf * KC * SW + 32 * sh + il;
        const float* sb = Sb + buf * KC * NO + il;
        float a_cur[NJ], a_nxt[NJ], b_cur, b_nxt;
        {
            const int row = 2 * kh + kq;
            b_cur = eb[row * SW];
#pragma unroll
            for (int t = 0; t < NJ; ++t) a_cur[t] = sb[row * NO + 32 * t];
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (q + 1 < NQ) {
                const int row = 2 * (NKH * (q + 1) + kh) + kq;
                b_nxt = eb[row * SW];
#pragma unroll
                for (int t = 0; t < NJ; ++t) a_nxt[t] = sb[row * NO + 32 * t];
            }
            __builtin_amdgcn_sched_barrier(0);
            const float bs = b_cur - shift;
#pragma unroll
            for (int t = 0; t < NJ; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[t], bs, acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (q + 1 < NQ) {
                b_cur = b_nxt;
#pragma unroll
                for (int t = 0; t < NJ; ++t) a_cur[t] = a_nxt[t];
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of the next chunk have landed
        __syncthreads();                                   // ... and everybody else's; the current buffer is free
    }
    // fixed-order tree over the NKH k-parts through LDS: step s = 1, 2, 4: part kh (kh % 2s == s) is added to part kh - s.
    // Two publisher slots (82 KB at n_obs = 160) alias the chunk buffers; a step with more than two publishers runs in rounds.
    float* red = sm;
    auto slot = [&](int pub, int t, int r) { return red + (((pub * 2 + sh) * NJ + t) * 16 + r) * 64 + lane; };
    __syncthreads();
#pragma unroll
    for (int st = 1; st < NKH; st *= 2) {
        const int npub = NKH / (2 * st);                 // publishers of this step: kh = st, 3 st, 5 st, ...
#pragma unroll
        for (int round = 0; round < (npub + NS - 1) / NS; ++round) {
            const int pidx = (kh / (2 * st));            // index of this wave's pair within the step
            const bool in_round = pidx / NS == round;
            if (in_round && kh % (2 * st) == st) {
#pragma unroll
                for (int t = 0; t < NJ; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) *slot(pidx % NS, t, r) = acc[t][r];
            }
            __syncthreads();
            if (in_round && kh % (2 * st) == 0 && st * 2 < NKH) {
#pragma unroll
                for (int t = 0; t < NJ; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][r] += *slot(pidx % NS, t, r);
            }
            if (st * 2 < NKH) __syncthreads();
        }
    }
    if (kh == 0) {
#pragma unroll
        for (int t = 0; t < NJ; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * kq;  // accumulator row = observation
                Gx[(size_t)j * M + istate] = acc[t][r] + *slot(0, t, r);
            }
    }
}

// k_gxt_dma2: as k_gxt_dma, but WITHOUT a workgroup barrier in the main loop.  4 waves per workgroup (one per SIMD), wave kh owns
// the members [16 kh, 16 kh + 16) of every 64-member chunk and BOTH state halves (64 state elements x all observations: 2 NJ
// accumulator tiles), and stages exactly the rows it consumes -- 16 rows of E (4 KB) and 16 rows of S (contiguous, 10 KB at
// n_obs = 160) -- by LDS-DMA into its own double buffer; it waits only on its own vmcnt.  The four partial sums are added in
// fixed order through LDS at the end.  Requires M % 64 == 0.
template <int NJ>
__global__ __launch_bounds__(256, 1) void k_gxt_dma2(int N, int M, int n_obs, const float* __restrict__ E,
                                                     const float* __restrict__ colsum, float inv_n,
                                                     const float* __restrict__ S, float* __restrict__ Gx) {
    constexpr int NO = 32 * NJ, KC = 64, KW = KC / 4, SW = 64, RW = SW + NO;  // rows per wave and chunk; floats per staged row pair
    constexpr int EP = KW * SW / 256, SP = KW * NO / 256;                      // DMA pieces per wave and chunk (4 + 10)
    static_assert(KW * NO % 256 == 0, "S slab of a wave must be whole 1 KB pieces");
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, kh = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* Ew = sm + kh * (2 * KW * RW);   // [2][KW][64] then [2][KW][NO] of this wave
    float* Sw = Ew + 2 * KW * SW;
    const int il = lane & 31, kq = lane >> 5;
    const int i0 = blockIdx.x * SW;
    float shift[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) shift[h] = colsum ? colsum[i0 + 32 * h + il] * inv_n : E[i0 + 32 * h + il];
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    auto stage = [&](int c, int buf) {
        const int k0 = c * KC + kh * KW;
#pragma unroll
        for (int pc = 0; pc < EP; ++pc) {
            const int row = min(k0 + 4 * pc + (lane >> 4), N - 1);
            __builtin_amdgcn_global_load_lds((glb_ptr)(E + (size_t)row * M + i0 + (lane & 15) * 4), (lds_ptr)(Ew + (buf * KW + 4 * pc) * SW), 16, 0, 0);
        }
#pragma unroll
        for (int pc = 0; pc < SP; ++pc) {
            const size_t e = min((size_t)k0 * NO + (size_t)pc * 256 + lane * 4, (size_t)N * NO - 4);
            __builtin_amdgcn_global_load_lds((glb_ptr)(S + e), (lds_ptr)(Sw + buf * KW * NO + pc * 256), 16, 0, 0);
        }
    };
    f32x16 acc[2][NJ];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int t = 0; t < NJ; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[h][t][r] = 0.0f;
    const int nchunks = (N + KC - 1) / KC;
    stage(0, 0);
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of chunk c (issued one chunk ago) have landed
        if (c + 1 < nchunks) stage(c + 1, buf ^ 1);
        const int nvalid = N - (c * KC + kh * KW);          // rows of this wave's slab that exist (wave-uniform)
        if (nvalid < KW) {                                  // last chunk: rows past N must not contribute -> zero their S rows
            const int first = max(nvalid, 0);
            for (int e = lane; e < (KW - first) * NO; e += 64) Sw[buf * KW * NO + first * NO + e] = 0.0f;
        }
        const float* eb = Ew + buf * KW * SW + il;
        const float* sb = Sw + buf * KW * NO + il;
        float a_cur[NJ], a_nxt[NJ], b_cur[2], b_nxt[2];
        b_cur[0] = eb[kq * SW]; b_cur[1] = eb[kq * SW + 32];
#pragma unroll
        for (int t = 0; t < NJ; ++t) a_cur[t] = sb[kq * NO + 32 * t];
#pragma unroll
        for (int q = 0; q < KW / 2; ++q) {
            if (q + 1 < KW / 2) {
                const int row = 2 * (q + 1) + kq;
                b_nxt[0] = eb[row * SW]; b_nxt[1] = eb[row * SW + 32];
#pragma unroll
                for (int t = 0; t < NJ; ++t) a_nxt[t] = sb[row * NO + 32 * t];
            }
            __builtin_amdgcn_sched_barrier(0);
            const float bs0 = b_cur[0] - shift[0], bs1 = b_cur[1] - shift[1];
#pragma unroll
            for (int t = 0; t < NJ; ++t) {
                acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[t], bs0, acc[0][t], 0, 0, 0);
                acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[t], bs1, acc[1][t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (q + 1 < KW / 2) {
                b_cur[0] = b_nxt[0]; b_cur[1] = b_nxt[1];
#pragma unroll
                for (int t = 0; t < NJ; ++t) a_cur[t] = a_nxt[t];
            }
        }
    }
    // fixed-order sum over the 4 waves through LDS: ((0 + 1) + 2) + 3; buffers [3][2][NJ][16][64] alias the chunk buffers
    __syncthreads();
    float* red = sm;
    auto slot = [&](int pub, int h, int t, int r) { return red + ((((pub * 2 + h) * NJ + t) * 16 + r) * 64) + lane; };
    if (kh > 0) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int t = 0; t < NJ; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) *slot(kh - 1, h, t, r) = acc[h][t][r];
    }
    __syncthreads();
    if (kh == 0) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int t = 0; t < NJ; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int j = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * kq;  // accumulator row = observation
                    const float v = ((acc[h][t][r] + *slot(0, h, t, r)) + *slot(1, h, t, r)) + *slot(2, h, t, r);
                    Gx[(size_t)j * M + i0 + 32 * h + il] = v;
                }
    }
}

// Third generation of the apply: one workgroup = 64 state elements x EVERY NG-th block of 128 members.  The Gx tile is staged
// once per workgroup (8 / NG times less often than one workgroup per member block), the grid is exactly two workgroups per CU
// (no partial last round), and the loads of a member block are issued one step ahead of their use: the A operand of the next
// 16 k-pairs while the current 16 feed the matrix cores, the E tile of the next member block behind the last A batch of the
// current one (vector-memory waits retire in issue order: an E tile requested in front of A loads would stall them).
template <int NG>
__global__ __launch_bounds__(256, 2) void k_apply_lds2(int N, int M, int n_obs, const float* __restrict__ E,
                                                       const float* __restrict__ At, const float* __restrict__ Gx,
                                                       float* __restrict__ Eout) {
    extern __shared__ __attribute__((aligned(16))) float Bs[];  // n_obs x 64
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i0 = blockIdx.x * 64;
    for (int e = threadIdx.x; e < n_obs * 16; e += 256) {  // float4 granules
        const int j = e >> 4, c4 = (e & 15) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i0 + c4 + 3 < M) v = *reinterpret_cast<const float4*>(Gx + (size_t)j * M + i0 + c4);
        *reinterpret_cast<float4*>(Bs + j * 64 + c4) = v;
    }
    const int nl = lane & 31, kh = lane >> 5;
    const int nblocks = (N + 127) / 128, npairs = (n_obs + 1) / 2;
    constexpr int UA = 8;
    float ev[2][16], evn[2][16], av[UA], avn[UA];
    // rows of a wave: n0 .. n0 + 31 with n0 clamped to N - 32 (host: N >= 32), so the ragged last block needs no per-row clamps --
    // its last waves recompute rows another wave also computes and store the same values.  M % 64 == 0 (host-checked).
    // The product is formed TRANSPOSED, (states x k) . (k x members): accumulator column = lane & 31 = member, registers = states
    // 32 t + 8 g + 4 kh + {0..3} -- four consecutive state elements per register quad, so the E tile moves as 16-byte accesses
    // (8 loads + 8 stores per thread and member block instead of 32 + 32 dword accesses: the scalar form was issue-bound in the
    // memory pipeline).  Addresses = wave-uniform pointer (scalar registers) + one 32-bit lane offset.
    const int wu = __builtin_amdgcn_readfirstlane(w);
    auto wave_row0 = [&](int blk) { return min(blk * 128 + 32 * wu, N - 32); };
    const int e_off = nl * M + 4 * kh;   // lane part of an E / E_out address (floats): member row nl, state quad kh
    const int a_off = kh * N + nl;       // lane part of an A-operand address
    auto load_E = [&](int blk, float (&dst)[2][16]) {
        const float* base = E + (size_t)wave_row0(blk) * M + i0 + e_off;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 v = *reinterpret_cast<const float4*>(base + 32 * t + 8 * g4);
                dst[t][4 * g4 + 0] = v.x; dst[t][4 * g4 + 1] = v.y; dst[t][4 * g4 + 2] = v.z; dst[t][4 * g4 + 3] = v.w;
            }
    };
    auto load_A = [&](int blk, int jp0, float (&dst)[UA]) {
        const float* base = At + wave_row0(blk) + (size_t)(2 * jp0) * N;
#pragma unroll
        for (int u = 0; u < UA; ++u) dst[u] = (base + (size_t)(2 * u) * N)[a_off];  // n_obs even: j = 2 (jp0 + u) + kh < n_obs
    };
    // (Measured and rejected: an explicit ping-pong of the A-operand registers with the batch loop unrolled by two and the batch's
    //  LDS reads grouped in front of its MFMAs -- 78 us against 67 us for this form, in which the compiler interleaves one
    //  ds_read2 per MFMA pair and waits for the A loads issued one batch earlier.)
    int b = blockIdx.y;
    if (b < nblocks) {
        load_A(b, 0, av);
        load_E(b, ev);
    }
    __syncthreads();
    for (; b < nblocks; b += NG) {
        f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
        const bool more = b + NG < nblocks;
        for (int jp0 = 0; jp0 < npairs; jp0 += UA) {  // npairs % UA == 0 (host-checked)
            if (jp0 + UA < npairs) {
                load_A(b, jp0 + UA, avn);
            } else if (more) {
                load_A(b + NG, 0, avn);
                load_E(b + NG, evn);
            }
            // the LDS operands of k-pair u + 2 are requested before the MFMAs of pair u are issued (two pairs = 256 cycles of
            // matrix work ahead: the LDS latency; see k_gxt_lds)
            const float* brow = Bs + (2 * jp0 + kh) * 64 + nl;
            float bq[UA + 2][2];
            bq[0][0] = brow[0]; bq[0][1] = brow[32];
            bq[1][0] = brow[128]; bq[1][1] = brow[128 + 32];
#pragma unroll
            for (int u = 0; u < UA; ++u) {
                if (u + 2 < UA) {
                    bq[u + 2][0] = brow[(u + 2) * 128];
                    bq[u + 2][1] = brow[(u + 2) * 128 + 32];
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[u][0], av[u], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[u][1], av[u], acc[1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int u = 0; u < UA; ++u) av[u] = avn[u];
        }
        float* obase = Eout + (size_t)wave_row0(b) * M + i0 + e_off;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float4 v;
                v.x = ev[t][4 * g4 + 0] + acc[t][4 * g4 + 0];
                v.y = ev[t][4 * g4 + 1] + acc[t][4 * g4 + 1];
                v.z = ev[t][4 * g4 + 2] + acc[t][4 * g4 + 2];
                v.w = ev[t][4 * g4 + 3] + acc[t][4 * g4 + 3];
                *reinterpret_cast<float4*>(obase + 32 * t + 8 * g4) = v;
            }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) ev[t][r] = evn[t][r];
    }
}

// k_apply_dma: the apply with its A operand (the gain, n_obs x N) staged by LDS-DMA into WAVE-PRIVATE double buffers: every wave
// streams the 32 member columns it needs in chunks of 16 k-pairs (4 KB = four 1 KB pieces of 8 rows x 128 B) and waits only on
// its own vmcnt -- no workgroup barrier in the main loop.  Everything else as k_apply_lds2 (Gx tile in LDS once per workgroup,
// transposed product, 16-byte E accesses, two workgroups per CU, every second block of 128 members per workgroup).  The member
// rows of a block start at min(128 b, N - 128): the ragged last block overlaps its predecessor and stores the same values again.
// Requires M % 64 == 0, N >= 128, N % 4 == 0, n_obs % 32 == 0.
template <int NG>
__global__ __launch_bounds__(256, 2) void k_apply_dma(int N, int M, int n_obs, const float* __restrict__ E,
                                                      const float* __restrict__ At, const float* __restrict__ Gx,
                                                      float* __restrict__ Eout) {
    extern __shared__ __attribute__((aligned(16))) float Bs[];  // [n_obs][64] Gx tile, then [4 waves][2][32][32] A chunks
    constexpr int UA = 16, AROWS = 2 * UA;                       // k-pairs / observation rows per chunk
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i0 = blockIdx.x * 64;
    float* Ab = Bs + n_obs * 64 + w * (2 * AROWS * 32);
    for (int e = threadIdx.x; e < n_obs * 16; e += 256) {  // float4 granules
        const int j = e >> 4, c4 = (e & 15) * 4;
        *reinterpret_cast<float4*>(Bs + j * 64 + c4) = *reinterpret_cast<const float4*>(Gx + (size_t)j * M + i0 + c4);
    }
    const int nl = lane & 31, kh = lane >> 5;
    const int nblocks = (N + 127) / 128, nchunk = n_obs / AROWS;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    auto row0 = [&](int blk) { return min(blk * 128, N - 128) + 32 * w; };  // this wave's first member row of block blk
    auto stage_A = [&](int blk, int ch, int buf) {  // rows [AROWS ch, +AROWS) of the gain, member columns row0 .. row0 + 31
        const float* base = At + (size_t)(AROWS * ch) * N + row0(blk);
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) {
            const float* src = base + (size_t)(8 * pc + (lane >> 3)) * N + (lane & 7) * 4;
            __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)(Ab + (buf * AROWS + 8 * pc) * 32), 16, 0, 0);
        }
    };
    const int e_off = nl * M + 4 * kh;
    float ev[2][16];
    auto load_E = [&](int blk, float (&dst)[2][16]) {
        const float* base = E + (size_t)row0(blk) * M + i0 + e_off;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 v = *reinterpret_cast<const float4*>(base + 32 * t + 8 * g4);
                dst[t][4 * g4 + 0] = v.x; dst[t][4 * g4 + 1] = v.y; dst[t][4 * g4 + 2] = v.z; dst[t][4 * g4 + 3] = v.w;
            }
    };
    int b = blockIdx.y, buf = 0;
    if (b < nblocks) stage_A(b, 0, 0);
    __syncthreads();  // Gx tile
    bool first = true;
    for (; b < nblocks; b += NG) {
        f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
        const bool more = b + NG < nblocks;
        for (int ch = 0; ch < nchunk; ++ch) {
            // Wait for THIS chunk's four pieces, issued one chunk ago.  Vector-memory operations retire in issue order, so what was
            // issued after them may stay in flight: at chunk 0 the 8 epilogue stores of the previous block, at chunk 1 the 8 loads of
            // this block's E tile (requested at chunk 0, consumed by the epilogue four chunks later).
            if ((ch == 0 && !first) || ch == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (ch + 1 < nchunk) stage_A(b, ch + 1, buf ^ 1);
            else if (more) stage_A(b + NG, 0, buf ^ 1);
            if (ch == 0) load_E(b, ev);
            const float* arow = Ab + (buf * AROWS + kh) * 32 + nl;       // A'^T[j = AROWS ch + 2 u + kh][member nl]
            const float* brow = Bs + (AROWS * ch + kh) * 64 + nl;        // Gx[j][state nl (+32)]
#pragma unroll
            for (int u = 0; u < UA; ++u) {
                const float a = arow[u * 64];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(brow[u * 128], a, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(brow[u * 128 + 32], a, acc[1], 0, 0, 0);
            }
            buf ^= 1;
        }
        first = false;
        float* obase = Eout + (size_t)row0(b) * M + i0 + e_off;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float4 v;
                v.x = ev[t][4 * g4 + 0] + acc[t][4 * g4 + 0];
                v.y = ev[t][4 * g4 + 1] + acc[t][4 * g4 + 1];
                v.z = ev[t][4 * g4 + 2] + acc[t][4 * g4 + 2];
                v.w = ev[t][4 * g4 + 3] + acc[t][4 * g4 + 3];
                *reinterpret_cast<float4*>(obase + 32 * t + 8 * g4) = v;
            }
    }
}

// ---- host entry points (return 0 launched, >0 error, -1 not applicable -> caller uses the generic GEMM) ----
int mfma_gxt(hipStream_t s, int N, int M, int n_obs, const float* E, const float* colsum, double inv_n, const float* S,
             float* Gxt) {
    if (n_obs % 32 != 0 || n_obs > 256 || n_obs % 4 != 0) return -1;
    const int nj = n_obs / 32;
    dim3 grid((M + 31) / 32), block(256);
    const size_t lds = (size_t)4 * nj * 16 * 64 * 4;
#define L(NJ) case NJ: HM_HIP(hipFuncSetAttribute((const void*)k_gxt_mfma<NJ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
                       hipLaunchKernelGGL(k_gxt_mfma<NJ>, grid, block, lds, s, N, M, n_obs, E, colsum, (float)inv_n, S, Gxt); break
    switch (nj) { L(1); L(2); L(3); L(4); L(5); L(6); L(7); L(8); default: return -1; }
#undef L
    HM_HIP(hipGetLastError());
    return 0;
}

int mfma_apply(hipStream_t s, int N, int M, int n_obs, const float* E, const float* At, const float* B, float* Eout) {
    const size_t lds = (size_t)n_obs * 128 * 4;
    if (lds > 150 * 1024 || M % 4 != 0) return -1;
    HM_HIP(hipFuncSetAttribute((const void*)k_apply_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid((M + 127) / 128, (N + 127) / 128), block(256);
    hipLaunchKernelGGL(k_apply_mfma, grid, block, lds, s, N, M, n_obs, E, At, B, Eout);
    HM_HIP(hipGetLastError());
    return 0;
}

int transpose_cast_d2f(hipStream_t s, const double* in, float* out, int rows, int cols) {
    dim3 grid((cols + 31) / 32, (rows + 31) / 32), block(256);
    hipLaunchKernelGGL((k_transpose<double, float>), grid, block, 0, s, in, out, rows, cols);
    HM_HIP(hipGetLastError());
    return 0;
}

int transpose_cast_f2d(hipStream_t s, const float* in, double* out, int rows, int cols) {
    dim3 grid((cols + 31) / 32, (rows + 31) / 32), block(256);
    hipLaunchKernelGGL((k_transpose<float, double>), grid, block, 0, s, in, out, rows, cols);
    HM_HIP(hipGetLastError());
    return 0;
}

int transpose_f2f(hipStream_t s, const float* in, float* out, int rows, int cols) {
    dim3 grid((cols + 31) / 32, (rows + 31) / 32), block(256);
    hipLaunchKernelGGL((k_transpose<float, float>), grid, block, 0, s, in, out, rows, cols);
    HM_HIP(hipGetLastError());
    return 0;
}

// second-generation kernels (hm_upd_run); Gx is (n_obs x M)
static int g_gxt_kc = 64;  // 64 members per LDS chunk: half the barriers of 32 (65.6 vs 68.6 us at C3)
void mfma_set_gxt_chunk(int kc) { g_gxt_kc = kc == 64 ? 64 : 32; }

static int g_gxt_dma = 1;  // chunks staged by LDS-DMA (k_gxt_dma) where it applies
void mfma_set_gxt_dma(int d) { g_gxt_dma = d; }
static int g_gxt_debug = 0;
void mfma_set_gxt_debug(int d) { g_gxt_debug = d; }
static int g_gxt_depth = 1;
void mfma_set_gxt_depth(int d) { g_gxt_depth = d == 2 ? 2 : 1; }
static int g_gxt_sh = 2;
void mfma_set_gxt_halves(int sh) { g_gxt_sh = sh == 1 ? 1 : 2; }

int mfma_gxt_lds(hipStream_t s, int N, int M, int n_obs, const float* E, const float* colsum, double inv_n, const float* S,
                 float* Gx) {
    if (n_obs % 32 != 0 || n_obs > 256 || M % 4 != 0 || M < 4) return -1;
    const int nj = n_obs / 32;
#ifdef HM_AB_VARIANTS  // superseded forms kept for A/B timing (make EXTRA=-DHM_AB_VARIANTS); the default build does not instantiate them
    if (g_gxt_dma == 2 && M % 64 == 0 && N >= 4 && nj <= 5 && (16 * n_obs) % 256 == 0) {
        dim3 grid(M / 64), block(256);
        const size_t chunks = (size_t)4 * 2 * 16 * (64 + n_obs) * 4, red = (size_t)3 * 2 * nj * 16 * 64 * 4;
        const size_t lds = chunks > red ? chunks : red;
        if (lds <= 160 * 1024) {
#define LD2(NJ) case NJ: HM_HIP(hipFuncSetAttribute((const void*)k_gxt_dma2<NJ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
                         hipLaunchKernelGGL((k_gxt_dma2<NJ>), grid, block, lds, s, N, M, n_obs, E, colsum, (float)inv_n, S, Gx); break
            switch (nj) { LD2(1); LD2(2); LD2(3); LD2(4); LD2(5); default: goto no_dma; }
#undef LD2
            HM_HIP(hipGetLastError());
            return 0;
        }
    }
    if (g_gxt_dma && g_gxt_kc == 32 && M % 64 == 0 && N >= 4) {
        // 32-member chunks and ONE publisher slot in the final sum: 56 KB of LDS at n_obs = 160 instead of 112 -- a few percent slower
        // alone, but a second workgroup (the fp64 chain of hm_upd_run on its own stream) fits on the CU beside it
        constexpr int KC = 32;
        dim3 grid(M / 64), block(512);
        const size_t chunks = (size_t)2 * KC * (64 + n_obs) * 4, red = (size_t)2 * (n_obs / 32) * 16 * 64 * 4;
        const size_t lds = chunks > red ? chunks : red;
#define LD(NJ) case NJ: HM_HIP(hipFuncSetAttribute((const void*)k_gxt_dma<NJ, KC, 4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
                        hipLaunchKernelGGL((k_gxt_dma<NJ, KC, 4, 1>), grid, block, lds, s, N, M, n_obs, E, colsum, (float)inv_n, S, Gx); break
        switch (nj) { LD(1); LD(2); LD(3); LD(4); LD(5); default: goto no_dma; }
#undef LD
        HM_HIP(hipGetLastError());
        return 0;
    }
#endif
    if (g_gxt_dma && M % 64 == 0 && N >= 4) {
        constexpr int KC = 64;
        dim3 grid(M / 64), block(512);
        const size_t lds = (size_t)2 * KC * (64 + n_obs) * 4;
        if (lds <= 160 * 1024) {
#ifdef HM_AB_VARIANTS
#define LD(NJ) case NJ: if (g_gxt_dma == 3) { HM_HIP(hipFuncSetAttribute((const void*)k_gxt_dma<NJ, KC, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
                        hipLaunchKernelGGL((k_gxt_dma<NJ, KC, 8>), grid, dim3(1024), lds, s, N, M, n_obs, E, colsum, (float)inv_n, S, Gx); } else { \
                        HM_HIP(hipFuncSetAttribute((const void*)k_gxt_dma<NJ, KC, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
                        hipLaunchKernelGGL((k_gxt_dma<NJ, KC, 4>), grid, block, lds, s, N, M, n_obs, E, colsum, (float)inv_n, S, Gx); } break
#else
#define LD(NJ) case NJ: HM_HIP(hipFuncSetAttribute((const void*)k_gxt_dma<NJ, KC, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
                        hipLaunchKernelGGL((k_gxt_dma<NJ, KC, 4>), grid, block, lds, s, N, M, n_obs, E, colsum, (float)inv_n, S, Gx); break
#endif
            switch (nj) { LD(1); LD(2); LD(3); LD(4); LD(5); default: goto no_dma; }
#undef LD
            HM_HIP(hipGetLastError());
            return 0;
        }
    }
no_dma:
#ifdef HM_AB_VARIANTS
    const int sh = g_gxt_sh;
#else
    const int sh = 2;
#endif
    const int kc = sh == 1 ? 32 : g_gxt_kc, sw = 32 * sh;
    dim3 grid((M + sw - 1) / sw), block(256 * sh);
    const size_t chunks = (size_t)2 * kc * (sw + n_obs) * 4, red = (size_t)2 * sh * nj * 16 * 64 * 4;
    const size_t lds = chunks > red ? chunks : red;
    if (lds > 160 * 1024) return -1;
#define L3(NJ, KC, SH) do { HM_HIP(hipFuncSetAttribute((const void*)k_gxt_lds<NJ, KC, SH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
                           hipLaunchKernelGGL((k_gxt_lds<NJ, KC, SH>), grid, block, lds, s, N, M, n_obs, E, colsum, (float)(g_gxt_debug ? -inv_n : inv_n), S, Gx); } while (0)
#define L4(NJ, KC) do { HM_HIP(hipFuncSetAttribute((const void*)k_gxt_lds<NJ, KC, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
                        hipLaunchKernelGGL((k_gxt_lds<NJ, KC, 2, 2>), grid, block, lds, s, N, M, n_obs, E, colsum, (float)inv_n, S, Gx); } while (0)
#ifdef HM_AB_VARIANTS
#define L(NJ) case NJ: if (sh == 1) L3(NJ, 32, 1); else if (g_gxt_depth == 2 && kc == 64) L4(NJ, 64); else if (kc == 64) L3(NJ, 64, 2); else L3(NJ, 32, 2); break
#else
#define L(NJ) case NJ: if (kc == 64) L3(NJ, 64, 2); else L3(NJ, 32, 2); break
#endif
    switch (nj) { L(1); L(2); L(3); L(4); L(5); L(6); L(7); L(8); default: return -1; }
#undef L
#undef L4
#undef L3
    HM_HIP(hipGetLastError());
    return 0;
}

static int g_apply_variant = 3;  // 3: k_apply_dma | 2: k_apply_lds2 | 1: k_apply_lds
void mfma_set_apply_variant(int v) { g_apply_variant = v; }

int mfma_apply_lds(hipStream_t s, int N, int M, int n_obs, const float* E, const float* At, const float* Gx, float* Eout) {
    const size_t lds = (size_t)n_obs * 64 * 4;
    if (lds > 150 * 1024 || M % 4 != 0) return -1;
    if (g_apply_variant == 3 && lds + 32 * 1024 <= 80 * 1024 && M % 64 == 0 && N >= 128 && N % 4 == 0 && n_obs % 32 == 0) {
        const size_t lds3 = lds + (size_t)4 * 2 * 32 * 32 * 4;  // + wave-private A chunks
        HM_HIP(hipFuncSetAttribute((const void*)k_apply_dma<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
        dim3 grid(M / 64, 2), block(256);
        hipLaunchKernelGGL(k_apply_dma<2>, grid, block, lds3, s, N, M, n_obs, E, At, Gx, Eout);
        HM_HIP(hipGetLastError());
        return 0;
    }
#ifdef HM_AB_VARIANTS
    if (g_apply_variant >= 2 && lds <= 78 * 1024 && M % 64 == 0 && N >= 32 && n_obs % 16 == 0) {  // two workgroups per CU
        HM_HIP(hipFuncSetAttribute((const void*)k_apply_lds2<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        dim3 grid((M + 63) / 64, 2), block(256);
        hipLaunchKernelGGL(k_apply_lds2<2>, grid, block, lds, s, N, M, n_obs, E, At, Gx, Eout);
        HM_HIP(hipGetLastError());
        return 0;
    }
#endif
    HM_HIP(hipFuncSetAttribute((const void*)k_apply_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid((M + 63) / 64, (N + 127) / 128), block(256);
    hipLaunchKernelGGL(k_apply_lds, grid, block, lds, s, N, M, n_obs, E, At, Gx, Eout);
    HM_HIP(hipGetLastError());
    return 0;
}
