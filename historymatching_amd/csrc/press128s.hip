// press128s.hip -- Ny = 128 fp64 pressure step, SYMMETRIC tile storage + a SIMD reserved for the pivot chain.
//
// Block elimination (block-Thomas along ix, every 128x128 Schur complement
// inverted explicitly by blocked symmetric Gauss-Jordan sweeps with rank-16 panels), re-laid-out around two
// measured facts of gfx950 (profiles/diag/inv16.hip):
//   * fp64 MFMA and the VALU share the SIMD's double-precision lanes: a wave's VALU instruction waits for every
//     in-flight v_mfma_f64 of the co-resident waves (64 cycles each).  The in-wave 16x16 pivot sweep (2.5k cycles
//     alone) takes 11k-19k cycles next to one or two MFMA streams, so it cannot be hidden under matrix-core work on
//     the same SIMD -- but a sweep on SIMD 0 is not slowed at all by MFMAs on SIMDs 1..3.
//   * fp64 MFMA peak equals fp64 VALU peak (78 TF): the matrix cores only pay if the flop count is minimal.
// Hence:
//   1. only the 36 lower-triangle 16x16 tiles of the symmetric block are stored and updated (-44% flops, -44% of
//      the G = inv(S_i) HBM stream, 72 KB instead of 128 KB per block);
//   2. waves are dealt round-robin to the SIMDs, so waves w = 0, 4, ... (SIMD 0) are SERVICE waves -- wave 0 is the
//      sweeper, together they do the 128-long vector work -- and only waves on SIMDs 1..3 hold tiles and issue
//      MFMAs.  The sweep of panel p+1's diagonal tile runs on SIMD 0 while SIMDs 1..3 do the rank-16 update of
//      panel p (look-ahead through LDS, released by an LDS flag instead of a workgroup barrier);
//   3. 8 waves per workgroup (2 service + 6 compute x 6 tiles), two workgroups per CU.
//
// Tile (R, C), R >= C, in the MFMA accumulator layout: lane (lq = l >> 4, lc = l & 15), register r <-> entry
// (16R + lq + 4r, 16C + lc).  Panel Cp (16 pivots = tile column/row Cp), with U = A[:, 16Cp..] (all 128 rows: rows
// above the diagonal come from the TRANSPOSED tiles of tile row Cp) and P = inv(A[Cp, Cp]):
//   B. tile (R, Cp), R > Cp:  W_R = U_R P           tile (Cp, C), C < Cp:  W_C^T = P U_C^T        diagonal: -P
//   C. every other tile (R, C):  A_RC -= W_R U_C^T  (both operands from LDS in operand layout).
#include "fwd_dev.h"
#include "sweep16.h"

#ifdef HM_PRESS_PROF
// Cycle stamps of workgroup 0: lane 0 of compute wave 1 (slots 0..15) and of the sweeper (slots 16..31).
__device__ long long hm_press_prof_s_buf[32];
#define PROF_DECL long long prof_t = clock64(), prof_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define PROF(i) do { const long long now_ = clock64(); prof_acc[i] += now_ - prof_t; prof_t = now_; } while (0)
#else
#define PROF_DECL
#define PROF(i)
#endif

namespace {

constexpr int NB = 128;

template <int NW>
struct SCfg {
    static constexpr int NS = NW / 4;     // service waves (SIMD 0)
    static constexpr int NC = NW - NS;    // compute waves (SIMDs 1..3)
    static constexpr int TPW = 36 / NC;   // lower-triangle tiles per compute wave
    static constexpr int NT = 64 * NW;
    static constexpr int NCT = 64 * NC;   // compute threads
};

// Tile -> (compute wave, slot).  Found by local search (every panel: the 7 phase-B tiles on at most 3 per SIMD and 2
// per wave; phase-C tiles 8..10 per SIMD).  Compute wave c sits on SIMD 1 + c % 3.  Entries are 16*R + C.
__constant__ unsigned char TILE_TAB8[6][6] = {
    {0x21, 0x31, 0x43, 0x50, 0x66, 0x77},
    {0x11, 0x30, 0x53, 0x60, 0x64, 0x72},
    {0x20, 0x44, 0x52, 0x54, 0x63, 0x71},
    {0x10, 0x33, 0x42, 0x65, 0x70, 0x76},
    {0x00, 0x51, 0x55, 0x62, 0x73, 0x74},
    {0x22, 0x32, 0x40, 0x41, 0x61, 0x75},
};
__constant__ unsigned char TILE_TAB16[12][3] = {
    {0x10, 0x11, 0x75}, {0x21, 0x65, 0x77}, {0x50, 0x55, 0x64}, {0x30, 0x52, 0x74}, {0x22, 0x41, 0x53}, {0x51, 0x60, 0x73},
    {0x33, 0x43, 0x76}, {0x20, 0x44, 0x71}, {0x00, 0x54, 0x62}, {0x32, 0x61, 0x66}, {0x42, 0x63, 0x70}, {0x31, 0x40, 0x72},
};

struct __attribute__((aligned(16))) SLds {
    // (static LDS must stay below 64 KB; rows padded to 17 doubles against bank conflicts of the operand reads)
    double U[2][NB][17];  // the 16 pivot columns, all 128 rows; double buffered (panel p+1 is published during panel p)
    double W[NB][17];     // U P
    double P[16][17];     // inverse of the diagonal tile
    double Dg[16][17];    // diagonal tile handed to the sweeper
    double ev[NB], dgv[NB], tyv[NB + 8], yprev[NB], ycur[NB];
    int flag;             // token of the diagonal tile currently in Dg
    int pad[3];
    // mat-vec scratch aliases the panel buffers (idle during the substitution mat-vecs)
    __device__ double* transpose_buf(int c) { return &U[0][0][0] + c * (16 * 17); }  // [NC][16][17]
    __device__ double* partial(int c) { return &W[0][0] + c * NB; }                  // [NC][NB]
};

struct SGeo {
    int lane, lc, lq;
};

// The tile coordinates of a slot are wave-uniform run-time values; left alone, the compiler hoists every LDS address
// derived from them out of the block/panel loops (dozens of VGPRs per slot) and spills.  Passing them through an
// empty asm at the point of use keeps the address arithmetic (a few scalar/vector ops) inside the loops.
__device__ __forceinline__ int opaque(int x) {
    asm volatile("" : "+s"(x));
    return x;
}

__device__ __forceinline__ void wave_lds_fence() {
    // lanes of ONE wave exchange data through LDS: the hardware keeps a wave's LDS operations in order, the compiler
    // must be told not to move the loads above the other lanes' stores
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Partial products of t = A v for the tiles of one compute wave (A symmetric, lower tiles stored):
//   tile (R, C) gives t[16C + j] += sum_i tile[i][j] v[16R + i]      (column-wise: no cross-lane reduction)
//   and, if R != C,  t[16R + i] += sum_j tile[i][j] v[16C + j]      (the same on the tile transposed through LDS)
// Lanes differing in lq hold partial sums of the same output: they are combined by LDS atomic adds into the wave's
// private vector, which the service threads sum over the compute waves after a barrier.
template <int TPW>
__device__ __forceinline__ void matvec_partial(const d4 (&acc)[TPW], const int (&tR)[TPW], const int (&tC)[TPW],
                                               const double* __restrict__ v, double* __restrict__ tw, double* __restrict__ tb,
                                               const SGeo& g) {
    tw[g.lane] = 0.0;
    tw[64 + g.lane] = 0.0;
    wave_lds_fence();
#pragma unroll
    for (int s = 0; s < TPW; ++s) {
        const int R = opaque(tR[s]), C = opaque(tC[s]);
        double sc = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) sc = fma(acc[s][r], v[16 * R + g.lq + 4 * r], sc);
        atomicAdd(&tw[16 * C + g.lc], sc);
        if (R != C) {
#pragma unroll
            for (int r = 0; r < 4; ++r) tb[(g.lq + 4 * r) * 17 + g.lc] = acc[s][r];
            wave_lds_fence();
            double sr = 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) sr = fma(tb[g.lc * 17 + g.lq + 4 * r], v[16 * C + g.lq + 4 * r], sr);
            atomicAdd(&tw[16 * R + g.lc], sr);
            wave_lds_fence();
        }
    }
}

template <int NC>
__device__ __forceinline__ double matvec_total(SLds& L, int j) {
    double a = L.partial(0)[j], b = L.partial(1)[j];
#pragma unroll
    for (int c = 2; c < NC; c += 2) {
        a += L.partial(c)[j];
        b += L.partial(c + 1)[j];
    }
    return a + b;
}

// FACTOR = true: factor-only entry for the coarse level of the two-level CG preconditioner (press_pcg.hip): the face
// transmissibilities TX, TY and the SPD pin are GIVEN (p.TX, p.TY, p.pin), nothing is assembled, no right-hand side is
// eliminated, only the inverse Schur complements G_i are produced (p.G) for k_coarse_solve below.
template <typename TS, int NW, bool FACTOR = false>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 4 : 2) void k_press128s(FwdParams p, const TS* __restrict__ S_base, long long S_stride,
                                                                       int k) {
    using Cf = SCfg<NW>;
    constexpr int NC = Cf::NC, TPW = Cf::TPW, NT = Cf::NT, NCT = Cf::NCT;
    __shared__ SLds L;
    const int m = blockIdx.x;
    const int tid = threadIdx.x;
    SGeo g;
    g.lane = tid & 63;
    g.lc = g.lane & 15;
    g.lq = g.lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Roles by wave index: waves 0, 4, ... (one SIMD under round-robin placement) are the service waves.  (Roles by PHYSICAL
    // SIMD -- the service waves of both workgroups of a CU on SIMD 0, read from HW_REG_HW_ID -- were measured in round 2: the
    // sweep drops from 4.4 k to 3.3 k cycles, but both workgroups' matrix-core waves then share three SIMDs instead of
    // spilling onto the other workgroup's service SIMD, and the launch takes 12.5 ms instead of 11.6.)
    // (Also measured, end of round 2: the next pivot tile's last update on the second service wave -- W^T = P U^T straight into the
    // B-operand layout, before the panel's first barrier instead of after it, as in spdinv.hip's factorisation -- with the two
    // workgroup barriers per panel kept: 12.4 ms instead of 11.45.  The barriers tie the sweeper to the tile waves either way;
    // taking them apart needs the flag / counter synchronisation of spdinv.hip in this kernel too.)
    // (Round 2, later, both measured with tests/tools/press_time_only.py at 1000 members, 11.45 ms as it stands:
    //   * the flag / counter synchronisation itself -- sweeper and a pivot wave coupled by LDS flags, the six tile waves on a counter
    //     barrier of their own, P double buffered, the pre-pivot tile handed over through the free P slot: correct, 13.1 ms.  With two
    //     workgroups per CU the matrix pipes of SIMDs 1..3 are 86 % busy as it is; off the barriers the tile waves (5.9 k busy cycles per
    //     panel + 2.4 k at their own barrier) become the chain instead of the sweep (3.5 k) + pivot update (0.8 k).
    //   * the forward substitution's mat-vec G_{i-1} y_{i-1} moved off the chain, onto the second service wave, from the G_{i-1} the
    //     tile waves stored (one tile row or five tiles per panel; sums as LDS adds without return; tiles prefetched a panel ahead):
    //     correct, 12.3-13.2 ms in four forms, 10.8 ms with the wave's work compiled out -- every form made the wave the last at the
    //     panel's barriers (LDS round trips of ~300 cycles under the operand traffic, L2 latency of 1-2 us under the launch's stores).
    //   * the block prologue on its own: the mat-vec's two cross-lane sums finished in registers (v_permlane16/32_swap over lq, a
    //     register-halving DPP exchange over lc) instead of through a transposed LDS copy and colliding LDS adds: correct, 11.67 ms;
    //     the next block's vectors brought by LDS-DMA during the panel loop instead of by the service threads' register prefetch
    //     (which the register allocator spills, waiting for the loads where they are issued): correct, 11.53 ms.  With two workgroups
    //     per CU one workgroup's prologue runs under the other's panels; neither change moves the launch.)
    const bool service = (w & 3) == 0;
    const bool sweeper = w == 0;
    const int c = service ? 0 : (w >> 2) * 3 + (w & 3) - 1;  // compute wave index
    const int j = service ? (w >> 2) * 64 + g.lane : NB;       // vector element of a service thread (< NB: active)
    const int ct = c * 64 + g.lane;                           // compute thread index (G layout)
    const int Nx = p.Nx, Nxy = p.Nxy;

    const TS* S = S_base + (long long)m * S_stride;
    const double* Km = p.K + (long long)m * Nxy;
    const double* Kym = (!FACTOR && p.Ky) ? p.Ky + (long long)m * Nxy : Km;
    double* TX = p.TX + (long long)m * (Nx + 1) * NB;
    double* TY = p.TY + (long long)m * Nx * (NB + 1);
    double2* G = reinterpret_cast<double2*>(p.G + (long long)m * Nx * NB * NB);
    double* yv = p.yv + (long long)m * Nxy;
    double* P = p.P + (long long)m * Nxy;
    double* Vx = p.Vx + (long long)m * (Nx + 1) * NB;
    double* Vy = p.Vy + (long long)m * Nx * (NB + 1);
    const double* q = p.q + (long long)m * p.q_mstride + (long long)(p.q_cols > 1 ? k : 0) * Nxy;

    int tR[TPW], tC[TPW];
#pragma unroll
    for (int s = 0; s < TPW; ++s) {
        const int e = NW == 8 ? TILE_TAB8[c % 6][s % 6] : TILE_TAB16[c % 12][s % 3];
        tR[s] = __builtin_amdgcn_readfirstlane(e >> 4);
        tC[s] = __builtin_amdgcn_readfirstlane(e & 15);
    }

    if (sweeper) __builtin_amdgcn_s_setprio(3);  // the pivot chain outranks whatever else is issued on its SIMD
    PROF_DECL;
    if constexpr (!FACTOR) assemble_transmissibilities<TS>(p, S, Km, Kym, P /* scratch for L */, TX, TY, tid, NT);
    const double pin = FACTOR ? p.pin[m] : Km[0] + Kym[0];  // SPD pin: A[0,0] += Kx[0,0]+Ky[0,0]
    if (tid == 0) L.flag = 0;
    __syncthreads();  // TX/TY entries written by other threads are read below
    PROF(5);

    d4 acc[TPW];
    int bad = 0, cur = 0;
    // vectors of block i+1 are fetched by the service threads one block ahead
    double pf_y1 = 0.0, pf_y2 = 0.0, pf_x1 = 0.0, pf_x2 = 0.0, pf_q = 0.0, q_cur = 0.0;
    if (j < NB) {
        pf_y1 = TY[j]; pf_y2 = TY[j + 1]; pf_x1 = TX[j]; pf_x2 = TX[NB + j]; pf_q = FACTOR ? 0.0 : q[j];
    }

    // ---- one rank-16 panel --------------------------------------------------------------------------------------
    auto update_tile = [&](int s, int Cp, double (*U)[17]) {
        const int R = opaque(tR[s]), C = opaque(tC[s]);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
            acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(-L.W[16 * R + g.lc][4 * kk + g.lq], U[16 * C + g.lc][4 * kk + g.lq], acc[s], 0, 0, 0);
    };
    // hand tile s over as part of pivot column Cn: natural rows for a column tile, transposed for a row tile, the
    // diagonal tile to the sweeper (released by the flag)
    auto publish_tile = [&](int s, int Cn, double (*Un)[17], int token) {
        const int R = opaque(tR[s]), C = opaque(tC[s]);
        if (R == Cn && C == Cn) {
#pragma unroll
            for (int r = 0; r < 4; ++r) L.Dg[g.lq + 4 * r][g.lc] = acc[s][r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (g.lane == 0) __hip_atomic_store(&L.flag, token, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (C == Cn) {
#pragma unroll
            for (int r = 0; r < 4; ++r) Un[16 * R + g.lq + 4 * r][g.lc] = acc[s][r];
        } else {  // R == Cn, C < Cn
#pragma unroll
            for (int r = 0; r < 4; ++r) Un[16 * C + g.lc][g.lq + 4 * r] = acc[s][r];
        }
    };
    auto sweep_published = [&](int token) {
        // bounded spin: the tile arrives within a few thousand cycles; if it never does (a defect), flag the member and go
        // on rather than hang the GPU -- the workgroup barriers below still match
        for (int spins = 0; __hip_atomic_load(&L.flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != token; ++spins) {
            if (spins > (1 << 24)) { bad = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        PROF(13);
        d4 t;
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] = L.Dg[g.lq + 4 * r][g.lc];
        sweep16_inwave(t, g, bad);
#pragma unroll
        for (int r = 0; r < 4; ++r) L.P[g.lq + 4 * r][g.lc] = -t[r];  // t = -inv(tile)
        PROF(14);
    };

    for (int i = 0; i < Nx; ++i) {
        if (j < NB) {
            double dg = pf_y1 + pf_y2 + pf_x1 + pf_x2;
            if (i == 0 && j == 0) dg += pin;
            L.dgv[j] = dg;
            L.tyv[j] = pf_y1;
            if (j == NB - 1) L.tyv[NB] = pf_y2;
            L.ev[j] = pf_x1;
            q_cur = pf_q;
        }
        __syncthreads();
        PROF(6);
        if (i > 0) {
            if constexpr (!FACTOR) {
                if (!service) matvec_partial<TPW>(acc, tR, tC, L.yprev, L.partial(c), L.transpose_buf(c), g);
                __syncthreads();
                if (j < NB) L.ycur[j] = q_cur + L.ev[j] * matvec_total<NC>(L, j);
            }
            if (!service) {
#pragma unroll
                for (int s = 0; s < TPW; ++s) {
                    const int R = opaque(tR[s]), C = opaque(tC[s]);
                    const double ec = L.ev[16 * C + g.lc];
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[s][r] = -(L.ev[16 * R + g.lq + 4 * r] * acc[s][r] * ec);
                }
            }
        } else {
            if (j < NB) L.ycur[j] = q_cur;
#pragma unroll
            for (int s = 0; s < TPW; ++s) acc[s] = d4{0.0, 0.0, 0.0, 0.0};
        }
        if (j < NB && i + 1 < Nx) {  // next block's vectors
            const int in = i + 1;
            pf_y1 = TY[in * (NB + 1) + j]; pf_y2 = TY[in * (NB + 1) + j + 1];
            pf_x1 = TX[in * NB + j]; pf_x2 = TX[(in + 1) * NB + j];
            pf_q = FACTOR ? 0.0 : q[in * NB + j];
        }
        __syncthreads();  // the mat-vec scratch (aliases U, W) is free again
        PROF(7);
        if (!service) {
            // add the tridiagonal D_i (diagonal tiles, and the corner entry of the sub-diagonal tiles), then hand over
            // panel 0: column 0 and the first diagonal tile
#pragma unroll
            for (int s = 0; s < TPW; ++s) {
                const int R = opaque(tR[s]), C = opaque(tC[s]);
                if (R == C) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int lrow = g.lq + 4 * r, row = 16 * R + lrow, col = 16 * C + g.lc;
                        double add = 0.0;
                        if (g.lc == lrow) add = L.dgv[row];
                        else if (g.lc == lrow + 1) add = -L.tyv[col];
                        else if (lrow == g.lc + 1) add = -L.tyv[row];
                        acc[s][r] += add;
                    }
                } else if (R == C + 1) {  // entry (16R, 16C+15): row == col + 1
                    if (g.lane == 15) acc[s][0] -= L.tyv[16 * R];
                }
                if (C == 0) publish_tile(s, 0, L.U[cur], 8 * i + 1);
            }
        } else if (sweeper) {
            sweep_published(8 * i + 1);
        }
        PROF(8);
        __syncthreads();  // U_0, P_0 visible
        PROF(9);
        for (int Cp = 0; Cp < 8; ++Cp) {
            double (*U)[17] = L.U[cur];
            double (*Un)[17] = L.U[cur ^ 1];
            const int Cn = Cp + 1, token = 8 * i + Cn + 1;
            if (!service) {
                // ---- phase B: the swept tile column / tile row
#pragma unroll
                for (int s = 0; s < TPW; ++s) {
                    const int R = opaque(tR[s]), C = opaque(tC[s]);
                    if (C == Cp) {
                        if (R == Cp) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[s][r] = -L.P[g.lq + 4 * r][g.lc];
                        } else {
                            d4 wv = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                            for (int kk = 0; kk < 4; ++kk)
                                wv = __builtin_amdgcn_mfma_f64_16x16x4f64(U[16 * R + g.lc][4 * kk + g.lq], L.P[4 * kk + g.lq][g.lc], wv, 0, 0, 0);
#pragma unroll
                            for (int r = 0; r < 4; ++r) L.W[16 * R + g.lq + 4 * r][g.lc] = wv[r];
                            acc[s] = wv;
                        }
                    } else if (R == Cp) {  // C < Cp: W_C^T = P U_C^T
                        d4 wv = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk)
                            wv = __builtin_amdgcn_mfma_f64_16x16x4f64(L.P[g.lc][4 * kk + g.lq], U[16 * C + g.lc][4 * kk + g.lq], wv, 0, 0, 0);
#pragma unroll
                        for (int r = 0; r < 4; ++r) L.W[16 * C + g.lc][g.lq + 4 * r] = wv[r];
                        acc[s] = wv;
                    }
                }
            }
            PROF(0);
            __syncthreads();  // W visible; P_p and Dg free
            PROF(1);
            if (!service) {
                // ---- phase C, pass 1: tiles of the NEXT pivot column/row first, handed over at once
                if (Cn < 8) {
                    // the next DIAGONAL tile before anything else: its in-wave sweep on the service SIMD is the critical path
                    // of the panel (stamps: the sweeper waited 1.9 k cycles per panel for it when it came in slot order)
#pragma unroll
                    for (int s = 0; s < TPW; ++s) {
                        const int R = tR[s], C = tC[s];
                        if (R == Cn && C == Cn) {
                            __builtin_amdgcn_s_setprio(3);  // ahead of the other waves' rank-16 updates on this SIMD
                            update_tile(s, Cp, U);
                            publish_tile(s, Cn, Un, token);
                            __builtin_amdgcn_s_setprio(0);
                        }
                    }
#pragma unroll
                    for (int s = 0; s < TPW; ++s) {
                        const int R = tR[s], C = tC[s];
                        if ((R == Cn) != (C == Cn)) {
                            if (C != Cp) update_tile(s, Cp, U);
                            publish_tile(s, Cn, Un, token);
                        }
                    }
                }
                PROF(2);
                // ---- pass 2: the rest
#pragma unroll
                for (int s = 0; s < TPW; ++s) {
                    const int R = tR[s], C = tC[s];
                    if (R == Cp || C == Cp || R == Cn || C == Cn) continue;
                    update_tile(s, Cp, U);
                }
            } else if (sweeper && Cn < 8) {
                sweep_published(token);
            }
            PROF(3);
            __syncthreads();  // U_{p+1}, P_{p+1} visible; W free
            PROF(4);
            cur ^= 1;
        }
        // G_i = -A: keep in the accumulators for the next block, stream the 36 tiles to HBM (thread-major 16-byte chunks)
        if (!service) {
            double2* Gi = G + (long long)i * (NB * NB / 2);
#pragma unroll
            for (int s = 0; s < TPW; ++s) {
                acc[s] = -acc[s];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    double2 v;
                    v.x = acc[s][2 * h];
                    v.y = acc[s][2 * h + 1];
                    Gi[(s * 2 + h) * NCT + ct] = v;
                }
            }
        }
        if (!FACTOR && j < NB) {
            yv[i * NB + j] = L.ycur[j];
            L.yprev[j] = L.ycur[j];
        }
        __syncthreads();
        PROF(10);
    }
    if constexpr (FACTOR) {
        if (bad && g.lane == 0) atomicOr(&p.status[m], HM_MEMBER_BAD_PIVOT);
        return;
    }
    // back substitution: x_i = G_i (y_i + TX[i+1] * x_{i+1});  ycur holds x_{i+1};  G_{Nx-1} is still in the accumulators
    for (int i = Nx - 1; i >= 0; --i) {
        if (!service && i < Nx - 1) {
            const double2* Gi = G + (long long)i * (NB * NB / 2);
#pragma unroll
            for (int s = 0; s < TPW; ++s)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const double2 v = Gi[(s * 2 + h) * NCT + ct];
                    acc[s][2 * h] = v.x;
                    acc[s][2 * h + 1] = v.y;
                }
        }
        if (j < NB) {
            double v = yv[i * NB + j];
            if (i < Nx - 1) v += TX[(i + 1) * NB + j] * L.ycur[j];
            L.yprev[j] = v;
        }
        __syncthreads();
        if (!service) matvec_partial<TPW>(acc, tR, tC, L.yprev, L.partial(c), L.transpose_buf(c), g);
        __syncthreads();
        if (j < NB) {
            const double t = matvec_total<NC>(L, j);
            L.ycur[j] = t;
            P[i * NB + j] = t;
        }
        __syncthreads();
    }
    PROF(11);
    face_fluxes(p, P, TX, TY, Vx, Vy, tid, NT);
    PROF(12);
#ifdef HM_PRESS_PROF
    if (m == 0 && g.lane == 0 && (w == 0 || w == 1))
        for (int q_ = 0; q_ < 16; ++q_) hm_press_prof_s_buf[16 * (1 - w) + q_] = prof_acc[q_];
#endif
    if (bad && g.lane == 0) atomicOr(&p.status[m], HM_MEMBER_BAD_PIVOT);
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------
// x = A^-1 b for a system factored by k_press128s<.., FACTOR = true>: the two substitution passes of the block
// elimination with the stored inverse Schur complements (36 tiles per block, 72 KB), one right-hand side per member.
//   forward   y_0 = b_0,  y_i = b_i + e_i o (G_{i-1} y_{i-1})          e_i = TX[i] (coupling to block i-1)
//   backward  x_{Nx-1} = G_{Nx-1} y_{Nx-1},  x_i = G_i (y_i + e_{i+1} o x_{i+1})
// Coarse solve of the two-level CG preconditioner (press_pcg.hip): 2 Nx mat-vecs, the G stream read twice.
// `skip[m] != 0`: member already converged, nothing to do.
// ------------------------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(64 * NW, 4) void k_coarse_solve(FwdParams p, const double* __restrict__ b_base, double* __restrict__ x_base,
                                                              const int* __restrict__ skip) {
    using Cf = SCfg<NW>;
    constexpr int NC = Cf::NC, TPW = Cf::TPW, NCT = Cf::NCT;
    __shared__ SLds L;
    const int m = blockIdx.x;
    if (skip && skip[m]) return;
    const int tid = threadIdx.x;
    SGeo g;
    g.lane = tid & 63;
    g.lc = g.lane & 15;
    g.lq = g.lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool service = (w & 3) == 0;
    const int c = service ? 0 : (w >> 2) * 3 + (w & 3) - 1;
    const int j = service ? (w >> 2) * 64 + g.lane : NB;
    const int ct = c * 64 + g.lane;
    const int Nx = p.Nx, Nxy = p.Nxy;
    const double* TX = p.TX + (long long)m * (Nx + 1) * NB;
    const double2* G = reinterpret_cast<const double2*>(p.G + (long long)m * Nx * NB * NB);
    double* yv = p.yv + (long long)m * Nxy;
    const double* b = b_base + (long long)m * Nxy;
    double* x = x_base + (long long)m * Nxy;
    int tR[TPW], tC[TPW];
#pragma unroll
    for (int s = 0; s < TPW; ++s) {
        const int e = NW == 8 ? TILE_TAB8[c % 6][s % 6] : TILE_TAB16[c % 12][s % 3];
        tR[s] = __builtin_amdgcn_readfirstlane(e >> 4);
        tC[s] = __builtin_amdgcn_readfirstlane(e & 15);
    }
    d4 acc[TPW];
    auto load_G = [&](int i) {
        const double2* Gi = G + (long long)i * (NB * NB / 2);
#pragma unroll
        for (int s = 0; s < TPW; ++s)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const double2 v = Gi[(s * 2 + h) * NCT + ct];
                acc[s][2 * h] = v.x;
                acc[s][2 * h + 1] = v.y;
            }
    };
    if (j < NB) {
        const double v = b[j];
        yv[j] = v;
        L.yprev[j] = v;
    }
    for (int i = 1; i < Nx; ++i) {
        if (!service) load_G(i - 1);
        __syncthreads();
        if (!service) matvec_partial<TPW>(acc, tR, tC, L.yprev, L.partial(c), L.transpose_buf(c), g);
        __syncthreads();
        if (j < NB) {
            const double v = b[i * NB + j] + TX[i * NB + j] * matvec_total<NC>(L, j);
            yv[i * NB + j] = v;
            L.yprev[j] = v;
        }
    }
    for (int i = Nx - 1; i >= 0; --i) {
        if (!service) load_G(i);
        if (j < NB && i < Nx - 1) L.yprev[j] = yv[i * NB + j] + TX[(i + 1) * NB + j] * L.ycur[j];
        __syncthreads();
        if (!service) matvec_partial<TPW>(acc, tR, tC, L.yprev, L.partial(c), L.transpose_buf(c), g);
        __syncthreads();
        if (j < NB) {
            const double t = matvec_total<NC>(L, j);
            L.ycur[j] = t;
            x[i * NB + j] = t;
        }
    }
}

#ifdef HM_PRESS_PROF
extern "C" int hm_debug_press_prof_s(long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(hm_press_prof_s_buf), sizeof(long long) * 32);
}
#endif

// Returns 0 if launched, >0 on error, -1 if this specialisation does not apply.
int launch_pressure_128s(hm_fwd* f, const void* S, long long S_stride, int k) {
    const FwdParams& p = f->p;
    if (p.Ny != NB) return -1;
    hipStream_t s = f->ctx->stream;
#define LAUNCH(TS, NW) hipLaunchKernelGGL((k_press128s<TS, NW>), dim3(p.N), dim3(64 * NW), 0, s, p, (const TS*)S, S_stride, k)
    if (f->dtype == 64) {
        if (f->press_variant == 7) LAUNCH(double, 16);
        else LAUNCH(double, 8);
    } else {
        if (f->press_variant == 7) LAUNCH(float, 16);
        else LAUNCH(float, 8);
    }
#undef LAUNCH
    HM_HIP(hipGetLastError());
    return 0;
}

// Coarse level of the two-level CG preconditioner (press_pcg.hip).  `pc`: parameter block of the coarse system (Ny = 128,
// TX/TY/pin given, G and yv scratch).  Factor once per time step, then one solve per CG iteration.
int launch_coarse_factor_128(hipStream_t s, const FwdParams& pc) {
    if (pc.Ny != NB) return -1;
    hipLaunchKernelGGL((k_press128s<double, 8, true>), dim3(pc.N), dim3(512), 0, s, pc, (const double*)nullptr, 0LL, 0);
    HM_HIP(hipGetLastError());
    return 0;
}

int launch_coarse_solve_128(hipStream_t s, const FwdParams& pc, const double* b, double* x, const int* skip) {
    if (pc.Ny != NB) return -1;
    hipLaunchKernelGGL(k_coarse_solve<8>, dim3(pc.N), dim3(512), 0, s, pc, b, x, skip);
    HM_HIP(hipGetLastError());
    return 0;
}

// Self-test hook: D(16x16) = A(16x4) B(4x16) through one v_mfma_f64_16x16x4_f64 with the operand/result lane maps this
// file assumes.  Host buffers.
namespace {
__global__ void k_mfma_f64_probe(const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ D) {
    const int l = threadIdx.x;
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}
}  // namespace

extern "C" int hm_debug_mfma_f64(hm_ctx* ctx, const double* A, const double* B, double* D) {
    HM_REQUIRE(ctx && A && B && D, "hm_debug_mfma_f64: NULL argument");
    HM_HIP(hipSetDevice(ctx->device));
    double *dA, *dB, *dD;
    HM_HIP(hipMalloc(&dA, 64 * 8));
    HM_HIP(hipMalloc(&dB, 64 * 8));
    HM_HIP(hipMalloc(&dD, 256 * 8));
    HM_HIP(hipMemcpy(dA, A, 64 * 8, hipMemcpyHostToDevice));
    HM_HIP(hipMemcpy(dB, B, 64 * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_mfma_f64_probe, dim3(1), dim3(64), 0, ctx->stream, dA, dB, dD);
    HM_HIP(hipGetLastError());
    HM_HIP(hipStreamSynchronize(ctx->stream));
    HM_HIP(hipMemcpy(D, dD, 256 * 8, hipMemcpyDeviceToHost));
    (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dD);
    return 0;
}
