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
template <int NJ, int KC, int SH = 2>
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
    float4 ereg[EPT], sreg[SPT];
    auto fetch = [&](int k0) {
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
    fetch(0);
    stash(0);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunks) fetch((c + 1) * KC);
        const float* eb = Eb + buf * KC * SW + 32 * sh + il;
        const float* sb = Sb + buf * KC * NO + il;
        // this wave's k-pairs of the chunk: pair p = NKH q + kh  (rows 2p, 2p+1)
#pragma unroll
        for (int q = 0; q < KC / 2 / NKH; ++q) {
            const int row = 2 * (NKH * q + kh) + kq;
            const float b = eb[row * SW];
#pragma unroll
            for (int t = 0; t < NJ; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(sb[row * NO + 32 * t], b, acc[t], 0, 0, 0);
        }
        if (c + 1 < nchunks) stash(buf ^ 1);
        __syncthreads();
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
            const float* brow = Bs + (2 * jp0 + kh) * 64 + nl;
#pragma unroll
            for (int u = 0; u < UA; ++u) {
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(brow[u * 128 + 32 * t], av[u], acc[t], 0, 0, 0);
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

static int g_gxt_sh = 2;
void mfma_set_gxt_halves(int sh) { g_gxt_sh = sh == 1 ? 1 : 2; }

int mfma_gxt_lds(hipStream_t s, int N, int M, int n_obs, const float* E, const float* colsum, double inv_n, const float* S,
                 float* Gx) {
    if (n_obs % 32 != 0 || n_obs > 256 || M % 4 != 0 || M < 4) return -1;
    const int nj = n_obs / 32;
    const int sh = g_gxt_sh, kc = sh == 1 ? 32 : g_gxt_kc, sw = 32 * sh;
    dim3 grid((M + sw - 1) / sw), block(256 * sh);
    const size_t chunks = (size_t)2 * kc * (sw + n_obs) * 4, red = (size_t)2 * sh * nj * 16 * 64 * 4;
    const size_t lds = chunks > red ? chunks : red;
    if (lds > 160 * 1024) return -1;
#define L3(NJ, KC, SH) do { HM_HIP(hipFuncSetAttribute((const void*)k_gxt_lds<NJ, KC, SH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
                           hipLaunchKernelGGL((k_gxt_lds<NJ, KC, SH>), grid, block, lds, s, N, M, n_obs, E, colsum, (float)inv_n, S, Gx); } while (0)
#define L(NJ) case NJ: if (sh == 1) L3(NJ, 32, 1); else if (kc == 64) L3(NJ, 64, 2); else L3(NJ, 32, 2); break
    switch (nj) { L(1); L(2); L(3); L(4); L(5); L(6); L(7); L(8); default: return -1; }
#undef L
#undef L3
    HM_HIP(hipGetLastError());
    return 0;
}

static int g_apply_variant = 2;
void mfma_set_apply_variant(int v) { g_apply_variant = v; }

int mfma_apply_lds(hipStream_t s, int N, int M, int n_obs, const float* E, const float* At, const float* Gx, float* Eout) {
    const size_t lds = (size_t)n_obs * 64 * 4;
    if (lds > 150 * 1024 || M % 4 != 0) return -1;
    if (g_apply_variant == 2 && lds <= 78 * 1024 && M % 64 == 0 && N >= 32 && n_obs % 16 == 0) {  // two workgroups per CU
        HM_HIP(hipFuncSetAttribute((const void*)k_apply_lds2<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        dim3 grid((M + 63) / 64, 2), block(256);
        hipLaunchKernelGGL(k_apply_lds2<2>, grid, block, lds, s, N, M, n_obs, E, At, Gx, Eout);
        HM_HIP(hipGetLastError());
        return 0;
    }
    HM_HIP(hipFuncSetAttribute((const void*)k_apply_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid((M + 63) / 64, (N + 127) / 128), block(256);
    hipLaunchKernelGGL(k_apply_lds, grid, block, lds, s, N, M, n_obs, E, At, Gx, Eout);
    HM_HIP(hipGetLastError());
    return 0;
}
