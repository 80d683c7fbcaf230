// spdinv.hip -- explicit inverse of ONE symmetric positive definite matrix of order n = 16 nt <= 256 in fp64:
// W = inv(G + ridge I), the C^-1 of the ensemble-smoother update (C = S^T S + (N-1) I, HistoryMatch.py:585-586; n = n_obs
// = 160 in the reference's case).
//
// Same machinery as the pressure kernel press128s.hip: blocked symmetric Gauss-Jordan sweeps with rank-16 panels on
// the fp64 matrix cores, only the lower-triangle 16x16 tiles stored (accumulator layout: lane (lq, lc), register r
// <-> entry (16R + lq + 4r, 16C + lc)), the 16x16 pivot tile inverted inside one wave (sweep16.h) that sits on SIMD 0
// where no MFMA is issued, next panel's pivot tile handed over through LDS and released by an LDS flag.  One
// workgroup of 16 waves: waves 0, 4, 8, 12 are service waves (wave 0 sweeps), the other 12 hold SLOTS tiles each
// (tile t -> wave t % 12, slot t / 12).  ~10 panels x ~5k cycles at n = 160: ~25 us against 129 us for the rank-1
// register sweeps (k_invert_C_reg).
#include <algorithm>
#include <vector>

#include "common.h"
#include "sweep16.h"

namespace {

struct IGeo {
    int lane, lc, lq;
};

__device__ __forceinline__ int opaque_s(int x) {
    asm volatile("" : "+s"(x));
    return x;
}

template <int SLOTS, int NW>
__global__ __launch_bounds__(64 * NW, 4) void k_spd_inverse(const double* __restrict__ G, int nparts, int n, double ridge,
                                                      double* __restrict__ Wout, int* __restrict__ flag,
                                                      const double* __restrict__ add, double add_scale,
                                                      const double* __restrict__ rank1, double rank1_scale) {
    extern __shared__ __attribute__((aligned(16))) double lds_d[];
    const int nt = n >> 4;
    // LDS: U[2][n][17], W[n][17], P[16][17], Dg[16][17], flag
    double (*U0)[17] = reinterpret_cast<double (*)[17]>(lds_d);
    double (*U1)[17] = U0 + n;
    double (*Wp)[17] = U1 + n;
    double (*P)[17] = Wp + n;
    double (*Dg)[17] = P + 16;
    int* lflag = reinterpret_cast<int*>(Dg + 16);

    const int tid = threadIdx.x;
    IGeo g;
    g.lane = tid & 63;
    g.lc = g.lane & 15;
    g.lq = g.lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool service = (w & 3) == 0, sweeper = w == 0;
    constexpr int NC = NW - NW / 4;                            // compute waves (SIMDs 1..3)
    const int c = service ? 0 : (w >> 2) * 3 + (w & 3) - 1;  // compute wave index 0..NC-1
    const int ntiles = nt * (nt + 1) / 2;

    int tR[SLOTS], tC[SLOTS];
    d4 acc[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int t = s * NC + c;
        int R = -1, C = -1;
        if (!service && t < ntiles) {
            R = 0;
            while ((R + 1) * (R + 2) / 2 <= t) ++R;
            C = t - R * (R + 1) / 2;
        }
        tR[s] = __builtin_amdgcn_readfirstlane(R);
        tC[s] = __builtin_amdgcn_readfirstlane(C);
        acc[s] = d4{0.0, 0.0, 0.0, 0.0};
        if (R >= 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * R + g.lq + 4 * r, col = 16 * C + g.lc;
                // G may arrive as `nparts` partial Gram matrices (split over row blocks): summed here in fixed order
                // (nparts == 0: one matrix the caller has already made exactly symmetric -- no transposed read, which is a
                //  64-line gather per wave instruction through this one CU: ~25 of the kernel's 53 us at n = 160)
                double gs = 0.0;
                if (nparts == 0) gs = G[(size_t)row * n + col] + (add ? add_scale * add[(size_t)row * n + col] : 0.0);
                for (int z = 0; z < nparts; ++z) {
                    const double* Gz = G + (size_t)z * n * n;
                    gs += 0.5 * (Gz[(size_t)row * n + col] + Gz[(size_t)col * n + row]);
                }
                acc[s][r] = gs + (row == col ? ridge : 0.0);
            }
        }
    }
    if (tid == 0) *lflag = 0;
    int bad = 0, cur = 0;
#ifdef HM_INV_PROF
    unsigned long long pt, pa[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define ISTAMP(k) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); pa[k] += t_ - pt; pt = t_; } while (0)
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pt) :: "memory");
#else
#define ISTAMP(k)
#endif
    if (rank1) {  // Gram matrix of shifted columns -> of centred ones: the vector goes through LDS (W's buffer, not in use yet); read
                  // element by element from global memory beside the tile loads it cost this one CU 6.7 us
        double* r1 = reinterpret_cast<double*>(Wp);
        if (tid < n) r1[tid] = rank1[tid];
        __syncthreads();
#pragma unroll
        for (int s = 0; s < SLOTS; ++s)
            if (tR[s] >= 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[s][r] -= rank1_scale * (r1[16 * tR[s] + g.lq + 4 * r] * r1[16 * tC[s] + g.lc]);
            }
    }

    auto publish_tile = [&](int s, int Cn, double (*Un)[17], int token) {
        const int R = opaque_s(tR[s]), C = opaque_s(tC[s]);
        if (R == Cn && C == Cn) {
#pragma unroll
            for (int r = 0; r < 4; ++r) Dg[g.lq + 4 * r][g.lc] = acc[s][r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (g.lane == 0) __hip_atomic_store(lflag, token, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (C == Cn) {
#pragma unroll
            for (int r = 0; r < 4; ++r) Un[16 * R + g.lq + 4 * r][g.lc] = acc[s][r];
        } else {  // R == Cn, C < Cn: transposed
#pragma unroll
            for (int r = 0; r < 4; ++r) Un[16 * C + g.lc][g.lq + 4 * r] = acc[s][r];
        }
    };
    auto sweep_published = [&](int token) {
        // bounded spin: the tile arrives within a few thousand cycles; if it never does (a defect), flag the member and go
        // on rather than hang the GPU -- the workgroup barriers below still match
        for (int spins = 0; __hip_atomic_load(lflag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != token; ++spins) {
            if (spins > (1 << 24)) { bad = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        d4 t;
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] = Dg[g.lq + 4 * r][g.lc];
        sweep16_inwave(t, g, bad);
#pragma unroll
        for (int r = 0; r < 4; ++r) P[g.lq + 4 * r][g.lc] = -t[r];  // t = -inv(tile)
    };
    auto update_tile = [&](int s, double (*U)[17]) {
        const int R = opaque_s(tR[s]), C = opaque_s(tC[s]);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
            acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(-Wp[16 * R + g.lc][4 * kk + g.lq], U[16 * C + g.lc][4 * kk + g.lq], acc[s], 0, 0, 0);
    };

    __syncthreads();
    ISTAMP(0);
    // panel 0: column 0 and the first diagonal tile
    if (!service) {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s)
            if (tC[s] == 0) publish_tile(s, 0, U0, 1);
    } else if (sweeper) {
        sweep_published(1);
    }
    __syncthreads();
    ISTAMP(1);
    for (int Cp = 0; Cp < nt; ++Cp) {
        double (*U)[17] = cur ? U1 : U0;
        double (*Un)[17] = cur ? U0 : U1;
        const int Cn = Cp + 1, token = Cn + 1;
        if (!service) {
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                const int R = opaque_s(tR[s]), C = opaque_s(tC[s]);
                if (C == Cp) {
                    if (R == Cp) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[s][r] = -P[g.lq + 4 * r][g.lc];
                    } else {  // W_R = U_R P
                        d4 wv = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk)
                            wv = __builtin_amdgcn_mfma_f64_16x16x4f64(U[16 * R + g.lc][4 * kk + g.lq], P[4 * kk + g.lq][g.lc], wv, 0, 0, 0);
#pragma unroll
                        for (int r = 0; r < 4; ++r) Wp[16 * R + g.lq + 4 * r][g.lc] = wv[r];
                        acc[s] = wv;
                    }
                } else if (R == Cp) {  // C < Cp: W_C^T = P U_C^T
                    d4 wv = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
                        wv = __builtin_amdgcn_mfma_f64_16x16x4f64(P[g.lc][4 * kk + g.lq], U[16 * C + g.lc][4 * kk + g.lq], wv, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) Wp[16 * C + g.lc][g.lq + 4 * r] = wv[r];
                    acc[s] = wv;
                }
            }
        }
        ISTAMP(2);
        __syncthreads();  // W visible; P and Dg free
        ISTAMP(3);
        if (!service) {
            if (Cn < nt) {
#pragma unroll
                for (int s = 0; s < SLOTS; ++s) {
                    const int R = opaque_s(tR[s]), C = opaque_s(tC[s]);
                    if (R == Cn && C == Cn) {  // the next diagonal tile first: its sweep is the panel's critical path
                        update_tile(s, U);
                        publish_tile(s, Cn, Un, token);
                    }
                }
#pragma unroll
                for (int s = 0; s < SLOTS; ++s) {
                    const int R = opaque_s(tR[s]), C = opaque_s(tC[s]);
                    if (R >= 0 && ((R == Cn) != (C == Cn))) {
                        if (C != Cp) update_tile(s, U);
                        publish_tile(s, Cn, Un, token);
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                const int R = opaque_s(tR[s]), C = opaque_s(tC[s]);
                if (R < 0 || R == Cp || C == Cp || R == Cn || C == Cn) continue;
                update_tile(s, U);
            }
        } else if (sweeper && Cn < nt) {
            sweep_published(token);
        }
        ISTAMP(4);
        __syncthreads();
        ISTAMP(5);
        cur ^= 1;
    }
    // acc = -inv: write both triangles.  The mirrored tile goes through a wave-private 16 x 17 LDS block so that its rows are
    // stored contiguously too (a direct transposed store is a 64-line scatter per wave instruction through this one CU).
    if (!service) {
        double (*T)[17] = reinterpret_cast<double (*)[17]>(lds_d) + 16 * w;  // all panel buffers are free after the last barrier
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int R = tR[s], C = tC[s];
            if (R < 0) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * R + g.lq + 4 * r, col = 16 * C + g.lc;
                const double v = -acc[s][r];
                Wout[(size_t)row * n + col] = v;
                T[g.lq + 4 * r][g.lc] = v;
            }
            if (R != C) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < 4; ++r) Wout[(size_t)(16 * C + g.lq + 4 * r) * n + 16 * R + g.lc] = T[g.lc][g.lq + 4 * r];
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    ISTAMP(6);
#ifdef HM_INV_PROF
    if (g.lane == 0 && (w == 0 || w == 1 || w == 5 || w == 15))
        printf("spd_inverse wave %d: load %llu, panel0 %llu, phaseB %llu, barrier1 %llu, phaseC/sweep %llu, barrier2 %llu, store %llu cycles (n = %d)\n", w, pa[0], pa[1], pa[2], pa[3], pa[4], pa[5], pa[6], n);
#endif
    if (bad && g.lane == 0) atomicOr(flag, 1);
}

// ------------------------------------------------------------------------------------------------------------
// Block L D L^T factorisation instead of the explicit inverse, for the fused analysis step (hm_upd_run), which only needs the
// gain A' = D0 B^-1:   B = Lt Dt Lt^T,  Lt unit block-lower-triangular (16 x 16 blocks), Dt = diag(S_0 .. S_{nt-1}) the Schur pivots.
// Panel j:  P_j = S_j^-1 (the same in-wave sweep),  Lt_ij = A_ij P_j  (i > j),  A_ik -= Lt_ij A_kj^T  for i >= k > j -- the panel
// machinery of k_spd_inverse restricted to the TRAILING tiles: 210 tile products instead of 650, and per panel only the chain
// pivot sweep -> W = U P -> next diagonal tile is serial.  Output F (n x n, row-major): Lt_ij in the blocks below the diagonal, P_j in
// the diagonal blocks (the blocks above the diagonal are not written).
// colflag (may be NULL): a word in global memory that counts the finished columns of F (value j + 1: block column j -- the tiles
// Lt_ij, i > j, and P_j -- is in memory and visible to the device), for consumers that run beside the factorisation (k_ldl_chain).
template <int SLOTS, int NW>
__device__ __forceinline__ void ldl_factor_body(const double* __restrict__ G, int n, double* __restrict__ F, int* __restrict__ flag,
                                                const double* __restrict__ add, double add_scale,
                                                const double* __restrict__ rank1, double rank1_scale, int* __restrict__ colflag,
                                                double ridge = 0.0) {
    // Synchronisation: the chain  pivot sweep -> W^T = P U^T and update of the next pivot tile -> next sweep  is all that is serial
    // per panel, so the sweeper (wave 0, alone on SIMD 0: waves 4, 8, 12 only fix the wave placement and leave at once) and the
    // owner of the next pivot tile talk through two LDS flags and never meet the others at a barrier; the twelve tile waves
    // synchronise among themselves with an LDS counter (a workgroup barrier would make them wait for the sweeper, and it for them).
    extern __shared__ __attribute__((aligned(16))) double lds_d[];
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool service = (w & 3) == 0, sweeper = w == 0, pivot_wave = w == 4;
    if (service && !sweeper && !pivot_wave) return;
    const int nt = n >> 4;
    double (*U0)[17] = reinterpret_cast<double (*)[17]>(lds_d);
    double (*U1)[17] = U0 + n;
    double (*Wp0)[17] = U1 + n;  // W of even panels, then W of odd panels
    double (*P0)[17] = Wp0 + 2 * n;  // P of even panels, then P of odd panels
    double (*Dg)[17] = P0 + 32;
    double (*Dp0)[17] = Dg + 16;                 // the next pivot tile before its last update: tile j in Dp[j & 1]
    int* fDg = reinterpret_cast<int*>(Dp0 + 32); // token: pivot tile j is in Dg        (value j + 1)
    int* fP = fDg + 1;                           // token: P_j = S_j^-1 is in P[j & 1]   (value j + 1)
    int* bar = fDg + 2;                          // arrivals of the tile waves at their barriers
    int* colcnt = fDg + 3;                       // tile waves whose stores of the finished columns have been acknowledged
    int* fDp = fDg + 4;                          // token: tile j with the updates of panels 0 .. j-2 is in Dp[j & 1]   (value j + 1)
    int* fU = fDg + 5;                           // token: tile (j + 1, j) of block column j is in its U buffer          (value j + 1)

    IGeo g;
    g.lane = tid & 63;
    g.lc = g.lane & 15;
    g.lq = g.lane >> 4;
    constexpr int NC = NW - NW / 4;
    const int c = service ? 0 : (w >> 2) * 3 + (w & 3) - 1;
    const int ntiles = nt * (nt + 1) / 2;

    int tR[SLOTS], tC[SLOTS];
    d4 acc[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int t = s * NC + c;
        int R = -1, C = -1;
        if (!service && t < ntiles) {
            R = 0;
            while ((R + 1) * (R + 2) / 2 <= t) ++R;
            C = t - R * (R + 1) / 2;
        }
        tR[s] = __builtin_amdgcn_readfirstlane(R);
        tC[s] = __builtin_amdgcn_readfirstlane(C);
    }
    // Every load of the wave's tiles in ONE basic block (an empty slot loads tile (0, 0) and drops it): behind a branch per slot the
    // loads of slot s + 1 were issued after those of slot s had returned -- five round trips to memory (the producer ran on other
    // XCDs: nothing of G is in this L2), 8.5 us before the first pivot tile instead of 5
    const double r1v = rank1 ? rank1[min(tid, n - 1)] : 0.0;  // in flight together with the tiles
    {
        double gv[SLOTS][4], av[SLOTS][4];
#pragma unroll
        for (int s = 0; s < SLOTS; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const size_t e = (size_t)(16 * max(tR[s], 0) + g.lq + 4 * r) * n + 16 * max(tC[s], 0) + g.lc;
                gv[s][r] = G[e];
                av[s][r] = add ? add[e] : 0.0;
            }
#pragma unroll
        for (int s = 0; s < SLOTS; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                acc[s][r] = tR[s] >= 0 ? gv[s][r] + add_scale * av[s][r] + (tR[s] == tC[s] && g.lq + 4 * r == g.lc ? ridge : 0.0) : 0.0;
    }
    if (tid == 0) { *fDg = 0; *fP = 0; *bar = 0; *colcnt = 0; *fDp = 0; *fU = 0; }
    int bad = 0, cur = 0;
    if (rank1) {  // Gram matrix of shifted columns -> of centred ones (the vector goes through LDS: W's buffer, not in use yet)
        double* r1 = reinterpret_cast<double*>(Wp0);
        if (tid < n) r1[tid] = r1v;
        __syncthreads();
#pragma unroll
        for (int s = 0; s < SLOTS; ++s)
            if (tR[s] >= 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[s][r] -= rank1_scale * (r1[16 * tR[s] + g.lq + 4 * r] * r1[16 * tC[s] + g.lc]);
            }
    }
    __syncthreads();  // flags zeroed, r1 consumed: from here on no workgroup barrier

    auto wait_for = [&](int* f, int token) {
        for (int spins = 0; __hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < token; ++spins) {
            if (spins > (1 << 26)) { bad = 1; break; }  // a defect: flag it and go on rather than hang the GPU
        }
    };
    (void)wait_for;  // (not every instantiation has a busy-waiting role)
    auto wait_quietly = [&](int* f, int token) {  // sweeper and pivot wave share a SIMD: the one that waits must not take issue slots
        for (int spins = 0; __hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < token; ++spins) {
            if (spins > (1 << 24)) { bad = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
    };
    auto post = [&](int* f, int token) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (g.lane == 0) __hip_atomic_store(f, token, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    };

    if (sweeper) {
#ifdef HM_INV_PROF
        unsigned long long pt, pa[3] = {0, 0, 0};
#define LSTAMP(k) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); pa[k] += t_ - pt; pt = t_; } while (0)
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pt) :: "memory");
#else
#define LSTAMP(k)
#endif
        for (int j = 0; j < nt; ++j) {
            wait_quietly(fDg, j + 1);
            LSTAMP(0);
            d4 t;
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r] = Dg[g.lq + 4 * r][g.lc];
            sweep16_inwave(t, g, bad);  // t = -inv(tile)
            LSTAMP(1);
            double (*P)[17] = P0 + 16 * (j & 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) P[g.lq + 4 * r][g.lc] = -t[r];
            post(fP, j + 1);
            LSTAMP(2);
        }
#ifdef HM_INV_PROF
        if (g.lane == 0) printf("ldl_factor sweeper: waiting for pivot tiles %llu, sweeps %llu, posting %llu cycles (%d panels)\n", pa[0], pa[1], pa[2], nt);
#endif
        if (bad && g.lane == 0) atomicOr(flag, 1);
        return;
    }
    if (pivot_wave) {
        // The serial chain's other half, on a wave with nothing else to do (it shares SIMD 0 with the sweeper; one of the two is
        // always waiting for the other): the next pivot tile arrives from its owner with every update but the current panel's,
        // W^T = P U^T of its row block goes straight into the B-operand layout (a tile in accumulator layout is one), the tile's
        // last update A -= U W^T, and on to the sweeper.  As part of the tile waves' panel loop the same work waited for their
        // trailing updates and barriers: 3.1 k cycles per panel between two sweeps instead of 1.5 k.
#ifdef HM_INV_PROF
        unsigned long long qt, qa[4] = {0, 0, 0, 0};
#define QSTAMP(k) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); qa[k] += t_ - qt; qt = t_; } while (0)
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(qt) :: "memory");
#else
#define QSTAMP(k)
#endif
        for (int Cp = 0; Cp + 1 < nt; ++Cp) {
            const int Cn = Cp + 1;
            double (*U)[17] = (Cp & 1) ? U1 : U0;
            double (*P)[17] = P0 + 16 * (Cp & 1);
            double (*Dp)[17] = Dp0 + 16 * (Cn & 1);
            wait_quietly(fU, Cp + 1);  // the row block of column Cp this tile needs is in U (not the whole column: that is the tile waves' barrier)
            QSTAMP(0);
            wait_quietly(fDp, Cn + 1);
            d4 t;
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r] = Dp[g.lq + 4 * r][g.lc];
            QSTAMP(1);
            wait_quietly(fP, Cp + 1);
            QSTAMP(2);
            d4 wt = {0.0, 0.0, 0.0, 0.0}, wz = wt, tz = wt;
            double un[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) un[kk] = U[16 * Cn + g.lc][4 * kk + g.lq];
            wt = __builtin_amdgcn_mfma_f64_16x16x4f64(P[g.lc][g.lq], un[0], wt, 0, 0, 0);
            wz = __builtin_amdgcn_mfma_f64_16x16x4f64(P[g.lc][4 + g.lq], un[1], wz, 0, 0, 0);
            wt = __builtin_amdgcn_mfma_f64_16x16x4f64(P[g.lc][8 + g.lq], un[2], wt, 0, 0, 0);
            wz = __builtin_amdgcn_mfma_f64_16x16x4f64(P[g.lc][12 + g.lq], un[3], wz, 0, 0, 0);
            wt += wz;
            t = __builtin_amdgcn_mfma_f64_16x16x4f64(-un[0], wt[0], t, 0, 0, 0);
            tz = __builtin_amdgcn_mfma_f64_16x16x4f64(-un[1], wt[1], tz, 0, 0, 0);
            t = __builtin_amdgcn_mfma_f64_16x16x4f64(-un[2], wt[2], t, 0, 0, 0);
            tz = __builtin_amdgcn_mfma_f64_16x16x4f64(-un[3], wt[3], tz, 0, 0, 0);
            t += tz;
#pragma unroll
            for (int r = 0; r < 4; ++r) Dg[g.lq + 4 * r][g.lc] = t[r];
            post(fDg, Cn + 1);
            QSTAMP(3);
        }
#ifdef HM_INV_PROF
        if (g.lane == 0) printf("ldl_factor pivot wave: waiting for U row block %llu, for the pre-pivot tile %llu, for P %llu, W^T + update + post %llu cycles\n", qa[0], qa[1], qa[2], qa[3]);
#endif
        if (bad && g.lane == 0) atomicOr(flag, 1);
        return;
    }

    int gen = 0;
    auto tile_barrier = [&]() {  // the NC tile waves only
        ++gen;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (g.lane == 0) __hip_atomic_fetch_add(bar, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        wait_quietly(bar, NC * gen);
    };
    auto publish_column = [&](int s, double (*Un)[17]) {  // a tile of the next panel's column, below the pivot tile
        const int R = opaque_s(tR[s]);
#pragma unroll
        for (int r = 0; r < 4; ++r) Un[16 * R + g.lq + 4 * r][g.lc] = acc[s][r];
    };
    auto publish_pivot = [&](int s, int token) {
#pragma unroll
        for (int r = 0; r < 4; ++r) Dg[g.lq + 4 * r][g.lc] = acc[s][r];
        post(fDg, token);
    };
    // A tile product is four matrix instructions on one accumulator, each waiting for the one before (~100 cycles); two accumulators
    // of two instructions each and one addition halve the chain -- these chains, not the matrix pipe, are what the panel loop waits for.
    auto update_tile = [&](int s, double (*U)[17], double (*Wp)[17]) {
        const int R = opaque_s(tR[s]), C = opaque_s(tC[s]);
        d4 z = {0.0, 0.0, 0.0, 0.0};
        acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(-Wp[16 * R + g.lc][g.lq], U[16 * C + g.lc][g.lq], acc[s], 0, 0, 0);
        z = __builtin_amdgcn_mfma_f64_16x16x4f64(-Wp[16 * R + g.lc][4 + g.lq], U[16 * C + g.lc][4 + g.lq], z, 0, 0, 0);
        acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(-Wp[16 * R + g.lc][8 + g.lq], U[16 * C + g.lc][8 + g.lq], acc[s], 0, 0, 0);
        z = __builtin_amdgcn_mfma_f64_16x16x4f64(-Wp[16 * R + g.lc][12 + g.lq], U[16 * C + g.lc][12 + g.lq], z, 0, 0, 0);
        acc[s] += z;
    };

    // Telling the other workgroups that block columns 0 .. ncols-1 of F are in memory: every tile wave waits for the acknowledgements
    // of its own write-through stores (issued a panel ago: no stall) and counts itself in; the last one in publishes the count.  Kept off the tile waves' barriers: waiting there put the store latency into every panel.
    auto publish_columns = [&](int ncols) {
        if (!colflag || ncols < 1) return;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (g.lane == 0 && __hip_atomic_fetch_add(colcnt, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP) == NC * ncols - 1)
            __hip_atomic_store(colflag, ncols, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // the last one in tells the others
    };
    auto publish_prepivot = [&](int s, int j) {  // tile (j, j) with every update but panel j-1's, for the pivot wave
        double (*Dp)[17] = Dp0 + 16 * (j & 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) Dp[g.lq + 4 * r][g.lc] = acc[s][r];
        post(fDp, j + 1);
    };
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        if (tR[s] == 1 && tC[s] == 1) publish_prepivot(s, 1);
        if (tC[s] != 0) continue;
        if (tR[s] == 0) publish_pivot(s, 1);
        else {
            publish_column(s, U0);
            if (tR[s] == 1) post(fU, 1);
        }
    }
    tile_barrier();
#ifdef HM_INV_PROF
    unsigned long long tt, ta[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define TSTAMP(k) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); ta[k] += t_ - tt; tt = t_; } while (0)
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt) :: "memory");
#else
#define TSTAMP(k)
#endif
    for (int Cp = 0; Cp < nt; ++Cp) {
        double (*U)[17] = cur ? U1 : U0;
        double (*Un)[17] = cur ? U0 : U1;
        double (*P)[17] = P0 + 16 * (Cp & 1);
        double (*Wp)[17] = Wp0 + n * (Cp & 1);
        const int Cn = Cp + 1;
        wait_quietly(fP, Cp + 1);
        TSTAMP(0);
        publish_columns(Cp);
        TSTAMP(1);  // columns 0 .. Cp-1 (stored during the previous panels)
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int R = opaque_s(tR[s]), C = opaque_s(tC[s]);
            if (C != Cp) continue;
            if (R == Cp) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[s][r] = P[g.lq + 4 * r][g.lc];  // final: P_j
            } else {  // Lt_R = U_R P: final, and the panel's W operand
                d4 wv = {0.0, 0.0, 0.0, 0.0}, wz = wv;
                wv = __builtin_amdgcn_mfma_f64_16x16x4f64(U[16 * R + g.lc][g.lq], P[g.lq][g.lc], wv, 0, 0, 0);
                wz = __builtin_amdgcn_mfma_f64_16x16x4f64(U[16 * R + g.lc][4 + g.lq], P[4 + g.lq][g.lc], wz, 0, 0, 0);
                wv = __builtin_amdgcn_mfma_f64_16x16x4f64(U[16 * R + g.lc][8 + g.lq], P[8 + g.lq][g.lc], wv, 0, 0, 0);
                wz = __builtin_amdgcn_mfma_f64_16x16x4f64(U[16 * R + g.lc][12 + g.lq], P[12 + g.lq][g.lc], wz, 0, 0, 0);
                wv += wz;
#pragma unroll
                for (int r = 0; r < 4; ++r) Wp[16 * R + g.lq + 4 * r][g.lc] = wv[r];
                acc[s] = wv;
            }
            // final: out it goes -- write-through (sc1) when another XCD's workgroups read it while this kernel runs: every XCD has an
            // L2 of its own, and an agent-scope release fence (a write-back of this L2) cost 1.4 us per panel
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double* dst = F + (size_t)(16 * R + g.lq + 4 * r) * n + 16 * C + g.lc;
                if (colflag) __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst), (unsigned long long)__double_as_longlong(acc[s][r]),
                                                __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else *dst = acc[s][r];
            }
        }
        TSTAMP(2);
        tile_barrier();  // W visible
        TSTAMP(3);
        if (Cn < nt) {
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                const int R = opaque_s(tR[s]), C = opaque_s(tC[s]);
                if (C == Cn && R > Cn) {  // the next panel's column first
                    update_tile(s, U, Wp);
                    publish_column(s, Un);
                    if (R == Cn + 1) post(fU, Cn + 1);
                }
            }
        }
        TSTAMP(4);
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {  // the pivot tile after the next: the pivot wave wants it before the next sweep ends
            const int R = opaque_s(tR[s]), C = opaque_s(tC[s]);
            if (R == Cn + 1 && C == Cn + 1) {
                update_tile(s, U, Wp);
                publish_prepivot(s, Cn + 1);
            }
        }
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int R = opaque_s(tR[s]), C = opaque_s(tC[s]);
            if (C > Cn && !(R == Cn + 1 && C == Cn + 1)) update_tile(s, U, Wp);  // the rest of the trailing matrix (the next pivot tile is the pivot wave's)
        }
        TSTAMP(5);
        // No second barrier: phase B of the next panel reads only what this wave published itself (its own tiles of the next
        // column) and P; W has a buffer per panel parity; everything else of the next column is read after the next barrier.
        cur ^= 1;
    }
#ifdef HM_INV_PROF
    if (g.lane == 0 && (c == 0 || c == 5 || c == 11)) printf("ldl_factor tile wave %d: waiting for P %llu, publishing columns %llu, phase B %llu, barrier %llu, next column %llu, rest of phase C %llu, barrier %llu cycles\n", c, ta[0], ta[1], ta[2], ta[3], ta[4], ta[5], ta[6]);
#endif
    publish_columns(nt);
    if (bad && g.lane == 0) atomicOr(flag, 1);
}

template <int SLOTS, int NW>
__global__ __launch_bounds__(64 * NW, 4) void k_ldl_factor(const double* __restrict__ G, int n, double* __restrict__ F, int* __restrict__ flag,
                                                           const double* __restrict__ add, double add_scale,
                                                           const double* __restrict__ rank1, double rank1_scale) {
    ldl_factor_body<SLOTS, NW>(G, n, F, flag, add, add_scale, rank1, rank1_scale, nullptr);
}

// Gain of the fused analysis step from the block L D L^T factors:  A' = X B^-1 = X Lt^-T Dt^-1 Lt^-1  (X = D0: N x n innovations),
// written as its fp32 transpose A_T (n x N) -- the operand layout of the apply kernel.  Transposed, every step is a 16 x 16 tile
// product on the fp64 matrix cores with BOTH the running tile and the result in accumulator layout (a tile in accumulator layout is
// the B operand of v_mfma_f64_16x16x4 as it stands: register kk <-> rows 4 kk + lq).  One workgroup of 4 waves = 16 members: the nt
// tiles T_j = (X^T)_j are dealt to the waves (tile j -> wave j mod 4) and the solve runs right-looking,
//     forward   T_j -= Lt_jk T_k   (k < j),      scaling   T_j = P_j T_j,      backward   T_j -= Lt_kj^T T_k   (k > j),
// the owner of T_k handing it to the others through LDS at every step (18 workgroup barriers at n = 160; one wave running all 100
// tile products of its members in sequence took 22 us, the products of a step spread over four waves take 9).  The factor's tiles
// are A operands from LDS (the whole lower triangle, 2 KB per tile, staged once per workgroup).  No triangular solves inside tiles.
// colflag != NULL: the factorisation runs beside this workgroup (k_ldl_chain) -- block column k of F is staged when the counter says
// it is there, right before forward step k, so that only the last forward step, the scaling and the backward sweep are left when the
// factorisation ends.
template <int NT>
__device__ __forceinline__ void ldl_gain_body(const double* __restrict__ F, int n, const double* __restrict__ X, int N,
                                              float* __restrict__ A_T, int wg, const int* __restrict__ colflag, int* __restrict__ flag) {
    extern __shared__ __attribute__((aligned(16))) double lds_d[];
    constexpr int NTILES = NT * (NT + 1) / 2, NS = (NT + 3) / 4;  // tiles of the factor; tiles of T per wave
    double (*Ft)[16][17] = reinterpret_cast<double (*)[16][17]>(lds_d);  // tile (R, C), R >= C, at index R (R + 1) / 2 + C
    double (*Tk)[4][64] = reinterpret_cast<double (*)[4][64]>(lds_d + (size_t)NTILES * 16 * 17);  // [2][4][64]: the step's T_k
    const int tid = threadIdx.x, lane = tid & 63, lc = lane & 15, lq = lane >> 4, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const double* Fl = F + (size_t)(tid >> 4) * n + (tid & 15);
    if (!colflag) {  // stage the lower tiles: every load of a tile row of the factor in flight before the first LDS write of that row
#pragma unroll
        for (int R = 0; R < NT; ++R) {
            double v[NT];
#pragma unroll
            for (int C = 0; C <= R; ++C) v[C] = Fl[(size_t)16 * R * n + 16 * C];
#pragma unroll
            for (int C = 0; C <= R; ++C) Ft[R * (R + 1) / 2 + C][tid >> 4][tid & 15] = v[C];
        }
    }
    int stalled = 0;
    auto stage_column = [&](int k) {  // block column k of F (tiles (R, k), R >= k) once the factorisation has published it
        // ONE wave of the workgroup polls, at a leisurely rate: the counter lives in one L2 channel, and 250 waves asking for it every
        // few hundred cycles slowed the factorisation itself down to less than half its speed
        if (w == 0) {
            for (int spins = 0; __hip_atomic_load(colflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < k + 1; ++spins) {
                if (spins > (1 << 20)) { stalled = 1; break; }  // the factorisation never got there (a defect): flag it, do not hang
                __builtin_amdgcn_s_sleep(16);
            }
        }
        __syncthreads();
        double v[NT];  // sc1 loads: past this XCD's caches, where the factorising workgroup's write-through stores went
#pragma unroll
        for (int R = 0; R < NT; ++R)
            if (R >= k)
                v[R] = __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(Fl + (size_t)16 * R * n + 16 * k),
                                                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
#pragma unroll
        for (int R = 0; R < NT; ++R)
            if (R >= k) Ft[R * (R + 1) / 2 + k][tid >> 4][tid & 15] = v[R];
    };
    const int m0 = wg * 16, m = min(m0 + lc, N - 1);
    d4 T[NS];  // slot s <-> tile j = 4 s + w
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r) T[s][r] = 4 * s + w < NT ? X[(size_t)m * n + 16 * (4 * s + w) + lq + 4 * r] : 0.0;
    auto tile = [&](int R, int C) -> double (*)[17] { return Ft[R * (R + 1) / 2 + C]; };
    // hand-over of T_k: its owner writes the four registers, everybody reads them back after the barrier (two buffers: a step's
    // readers may still be reading while the next owner writes)
    auto hand_over = [&](int k, int step, d4& tk, bool forward) {
        if (colflag && forward) stage_column(k);
        if ((k & 3) == w) {
#pragma unroll
            for (int r = 0; r < 4; ++r) Tk[step & 1][r][lane] = T[k >> 2][r];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) tk[r] = Tk[step & 1][r][lane];
    };
    __syncthreads();
    int step = 0;
    // forward: Lt T = X^T (unit diagonal blocks)
#pragma unroll
    for (int k = 0; k + 1 < NT; ++k, ++step) {
        d4 tk;
        hand_over(k, step, tk, true);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int j = 4 * s + w;  // wave-uniform
                if (j > k && j < NT) T[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(-tile(j, k)[lc][4 * kk + lq], tk[kk], T[s], 0, 0, 0);
            }
    }
    if (colflag) {  // the last block column is the last pivot's inverse alone
        stage_column(NT - 1);
        __syncthreads();
    }
    // scaling by the inverse pivots (symmetric tiles)
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int j = 4 * s + w;
        if (j < NT) {
            d4 z = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) z = __builtin_amdgcn_mfma_f64_16x16x4f64(tile(j, j)[lc][4 * kk + lq], T[s][kk], z, 0, 0, 0);
            T[s] = z;
        }
    }
    // backward: Lt^T A'^T = T
#pragma unroll
    for (int k = NT - 1; k > 0; --k, ++step) {
        d4 tk;
        hand_over(k, step, tk, false);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int j = 4 * s + w;
                if (j < k) T[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(-tile(k, j)[4 * kk + lq][lc], tk[kk], T[s], 0, 0, 0);
            }
    }
    if (m0 + lc < N) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int j = 4 * s + w;
            if (j < NT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) A_T[(size_t)(16 * j + lq + 4 * r) * N + m0 + lc] = (float)T[s][r];
            }
        }
    }
    if (stalled && lane == 0) atomicOr(flag, 2);
}

template <int NT>
__global__ __launch_bounds__(256) void k_ldl_gain(const double* __restrict__ F, int n, const double* __restrict__ X, int N,
                                                  float* __restrict__ A_T) {
    ldl_gain_body<NT>(F, n, X, N, A_T, blockIdx.x, nullptr, nullptr);
}

// Factorisation and gain in ONE launch: workgroup 0 factorises, workgroups 1.. (four of their sixteen waves) run the gain of 16
// members each and take the factor's block columns as they are published -- the forward sweep runs behind the factorisation,
// the launch boundary, the staging of the factor and the loads of the innovations are off the serial path.  Workgroup 0 is
// dispatched first and waits for nobody, so the others' waits cannot deadlock; they are bounded all the same.
template <int SLOTS, int NW, int NT>
__global__ __launch_bounds__(64 * NW, 4) void k_ldl_chain(const double* __restrict__ G, int n, double* __restrict__ F, int* __restrict__ flag,
                                                          const double* __restrict__ add, double add_scale,
                                                          const double* __restrict__ rank1, double rank1_scale, int* __restrict__ colflag,
                                                          const double* __restrict__ X, int N, float* __restrict__ A_T, double ridge) {
    if (blockIdx.x == 0) {
        ldl_factor_body<SLOTS, NW>(G, n, F, flag, add, add_scale, rank1, rank1_scale, colflag, ridge);
    } else {
        if (threadIdx.x >= 256) return;
        ldl_gain_body<NT>(F, n, X, N, A_T, blockIdx.x - 1, colflag, flag);
    }
}

// ------------------------------------------------------------------------------------------------------------
// Localised analysis on the matrix cores (fp32 plans): one workgroup per state element i          HistoryMatch.py:783-793
//   jj = { j : c_j = sqrt(taper[i][j]) > cutoff },  Ci = (c c^T) o G[jj,jj] + (N-1) I,  w = Ci^-1 (c o Gxt[i,jj]),
//   Wt[i, jj] = c o w          (elements without an observation in range: Wt = 0, the element stays unchanged)
// Same panel machinery as k_spd_inverse, used as a SOLVE: the right-hand side rides along as one extra tile row below
// the (zero-padded to 16 nt) matrix that is never pivoted.  Sweeping the nt pivot panels of [[C, b], [b^T, .]] leaves
// b^T C^-1 = w^T in that row (the sweep operator), so no inverse is formed and no mat-vec is needed afterwards.
// fp64 plans keep the packed LDS Cholesky (k_local_analysis in update.hip): the explicit 16x16 pivot-tile inverses cost
// a factor cond(tile) of forward accuracy, inside the fp32 bar only.
// ------------------------------------------------------------------------------------------------------------
template <int SLOTS>
__global__ __launch_bounds__(1024, 4) void k_local_analysis_mfma(int M, int n_obs, int N_total, double cutoff, const float* __restrict__ taper,
                                                                 const double* __restrict__ G, const float* __restrict__ Gxt,
                                                                 float* __restrict__ Wt, int* __restrict__ flag) {
    extern __shared__ __attribute__((aligned(16))) double lds_d[];
    const int i = blockIdx.x;
    const int nmax = ((n_obs + 15) / 16) * 16 + 16;  // rows of U / W: padded matrix + the right-hand-side tile row
    double (*U0)[17] = reinterpret_cast<double (*)[17]>(lds_d);
    double (*U1)[17] = U0 + nmax;
    double (*Wp)[17] = U1 + nmax;
    double (*P)[17] = Wp + nmax;
    double (*Dg)[17] = P + 16;
    double* cv = reinterpret_cast<double*>(Dg + 16);   // n_obs: taper weights of the selected observations
    double* rhs = cv + n_obs;                           // n_obs
    int* jj = reinterpret_cast<int*>(rhs + n_obs);      // n_obs
    int* misc = jj + n_obs;                             // [0] n_loc  [1] flag token  [2..5] per-wave selection counts

    const int tid = threadIdx.x;
    IGeo g;
    g.lane = tid & 63;
    g.lc = g.lane & 15;
    g.lq = g.lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool service = (w & 3) == 0, sweeper = w == 0;
    const int c = service ? 0 : (w >> 2) * 3 + (w & 3) - 1;

    // selection, order-preserving (the reference's observation order): ballot + prefix count over the first 4 waves
    for (int j = tid; j < n_obs; j += 1024) Wt[(size_t)i * n_obs + j] = 0.0f;
    {
        const bool inr = tid < n_obs;  // n_obs <= 256 (host-checked)
        const double cj = inr ? sqrt((double)taper[(size_t)i * n_obs + tid]) : 0.0;
        const double gx = inr ? (double)Gxt[(size_t)i * n_obs + tid] : 0.0;
        const bool sel = inr && cj > cutoff;
        const unsigned long long mask = __ballot(sel);
        if (g.lane == 0 && w < 4) misc[2 + w] = __popcll(mask);
        if (tid == 0) misc[1] = 0;
        __syncthreads();
        if (w < 4) {
            int base = 0;
            for (int q = 0; q < w; ++q) base += misc[2 + q];
            const int pos = base + __popcll(mask & ((1ull << g.lane) - 1ull));
            if (sel) {
                jj[pos] = tid;
                cv[pos] = cj;
                rhs[pos] = cj * gx;
            }
        }
        if (tid == 0) misc[0] = misc[2] + misc[3] + misc[4] + misc[5];
    }
    __syncthreads();
    const int nloc = misc[0];
    if (nloc == 0) return;  // no observation in range: element unchanged (HistoryMatch.py:787-788)
    const int nt = (nloc + 15) >> 4;                  // pivot panels
    const int ntiles = (nt + 1) * (nt + 2) / 2;       // lower triangle of nt + 1 tile rows (the last = right-hand side)
    const double ridge = (double)(N_total - 1);
    int* lflag = misc + 1;

    int tR[SLOTS], tC[SLOTS];
    d4 acc[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int t = s * 12 + c;
        int R = -1, C = -1;
        if (!service && t < ntiles) {
            R = 0;
            while ((R + 1) * (R + 2) / 2 <= t) ++R;
            C = t - R * (R + 1) / 2;
        }
        tR[s] = __builtin_amdgcn_readfirstlane(R);
        tC[s] = __builtin_amdgcn_readfirstlane(C);
        acc[s] = d4{0.0, 0.0, 0.0, 0.0};
        if (R >= 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int lrow = g.lq + 4 * r, col = 16 * C + g.lc;
                double v = 0.0;
                if (R < nt) {
                    const int row = 16 * R + lrow;
                    if (row < nloc && col < nloc) {
                        const int a = jj[row], b = jj[col];  // row >= col of a lower tile and the selection keeps the order: a >= b
                        v = cv[row] * G[(size_t)max(a, b) * n_obs + min(a, b)] * cv[col];  // the lower triangle of the Gram matrix only
                    }
                    if (row == col) v += ridge;
                } else if (lrow == 0 && C < nt && col < nloc) {
                    v = rhs[col];
                }
                acc[s][r] = v;
            }
        }
    }
    int bad = 0, cur = 0;

    auto publish_tile = [&](int s, int Cn, double (*Un)[17], int token) {
        const int R = opaque_s(tR[s]), C = opaque_s(tC[s]);
        if (R == Cn && C == Cn) {
#pragma unroll
            for (int r = 0; r < 4; ++r) Dg[g.lq + 4 * r][g.lc] = acc[s][r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (g.lane == 0) __hip_atomic_store(lflag, token, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (C == Cn) {
#pragma unroll
            for (int r = 0; r < 4; ++r) Un[16 * R + g.lq + 4 * r][g.lc] = acc[s][r];
        } else {  // R == Cn, C < Cn: transposed
#pragma unroll
            for (int r = 0; r < 4; ++r) Un[16 * C + g.lc][g.lq + 4 * r] = acc[s][r];
        }
    };
    auto sweep_published = [&](int token) {
        // bounded spin: the tile arrives within a few thousand cycles; if it never does (a defect), flag the member and go
        // on rather than hang the GPU -- the workgroup barriers below still match
        for (int spins = 0; __hip_atomic_load(lflag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != token; ++spins) {
            if (spins > (1 << 24)) { bad = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        d4 t;
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] = Dg[g.lq + 4 * r][g.lc];
        sweep16_inwave(t, g, bad);
#pragma unroll
        for (int r = 0; r < 4; ++r) P[g.lq + 4 * r][g.lc] = -t[r];
    };
    auto update_tile = [&](int s, double (*U)[17]) {
        const int R = opaque_s(tR[s]), C = opaque_s(tC[s]);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
            acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(-Wp[16 * R + g.lc][4 * kk + g.lq], U[16 * C + g.lc][4 * kk + g.lq], acc[s], 0, 0, 0);
    };

    __syncthreads();
    if (!service) {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s)
            if (tC[s] == 0) publish_tile(s, 0, U0, 1);
    } else if (sweeper) {
        sweep_published(1);
    }
    __syncthreads();
    for (int Cp = 0; Cp < nt; ++Cp) {
        double (*U)[17] = cur ? U1 : U0;
        double (*Un)[17] = cur ? U0 : U1;
        const int Cn = Cp + 1, token = Cn + 1;
        if (!service) {
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                const int R = opaque_s(tR[s]), C = opaque_s(tC[s]);
                if (C == Cp) {
                    if (R == Cp) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[s][r] = -P[g.lq + 4 * r][g.lc];
                    } else {
                        d4 wv = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk)
                            wv = __builtin_amdgcn_mfma_f64_16x16x4f64(U[16 * R + g.lc][4 * kk + g.lq], P[4 * kk + g.lq][g.lc], wv, 0, 0, 0);
#pragma unroll
                        for (int r = 0; r < 4; ++r) Wp[16 * R + g.lq + 4 * r][g.lc] = wv[r];
                        acc[s] = wv;
                    }
                } else if (R == Cp) {
                    d4 wv = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
                        wv = __builtin_amdgcn_mfma_f64_16x16x4f64(P[g.lc][4 * kk + g.lq], U[16 * C + g.lc][4 * kk + g.lq], wv, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) Wp[16 * C + g.lc][g.lq + 4 * r] = wv[r];
                    acc[s] = wv;
                }
            }
        }
        __syncthreads();
        if (!service) {
            if (Cn < nt) {
#pragma unroll
                for (int s = 0; s < SLOTS; ++s) {
                    const int R = opaque_s(tR[s]), C = opaque_s(tC[s]);
                    if (R == Cn && C == Cn) {  // the next diagonal tile first: its sweep is the panel's critical path
                        update_tile(s, U);
                        publish_tile(s, Cn, Un, token);
                    }
                }
#pragma unroll
                for (int s = 0; s < SLOTS; ++s) {
                    const int R = opaque_s(tR[s]), C = opaque_s(tC[s]);
                    if (R >= 0 && ((R == Cn) != (C == Cn))) {
                        if (C != Cp) update_tile(s, U);
                        publish_tile(s, Cn, Un, token);
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                const int R = opaque_s(tR[s]), C = opaque_s(tC[s]);
                if (R < 0 || R == Cp || C == Cp || (Cn < nt && (R == Cn || C == Cn))) continue;
                update_tile(s, U);
            }
        } else if (sweeper && Cn < nt) {
            sweep_published(token);
        }
        __syncthreads();
        cur ^= 1;
    }
    // the right-hand-side tile row now holds w^T = b^T Ci^-1 in its first row
    if (!service) {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int R = tR[s], C = tC[s];
            if (R != nt || C >= nt || g.lq != 0) continue;
            const int col = 16 * C + g.lc;
            if (col < nloc) Wt[(size_t)i * n_obs + jj[col]] = (float)(cv[col] * acc[s][0]);
        }
    }
    if (bad && g.lane == 0) atomicOr(flag, 1);
}

}  // namespace

static int g_spd_small = 0;
void spd_inverse_set_small(int v) { g_spd_small = v; }

// W = inv(G + ridge I) for n a multiple of 16, n <= 256.  Returns 0 if launched, -1 if not applicable, >0 on error.
// nparts > 0: G is the sum of `nparts` partial matrices, symmetrised while loading; nparts == 0: one matrix of which only the
// lower 16 x 16 tiles are read, plus add_scale * add (lower tiles of a second matrix) if `add` is given.
// rank1 (n values) with rank1_scale: G - rank1_scale * rank1 rank1^T is inverted (the Gram matrix of shifted columns turned into
// the Gram matrix of centred ones while loading; nparts == 0 only).
int spd_inverse_mfma(hipStream_t s, const double* G, int nparts, int n, double ridge, double* W, int* flag, const double* add,
                     double add_scale, const double* rank1, double rank1_scale) {
    if (n % 16 != 0 || n < 16 || n > 256) return -1;
    const int nt = n / 16, ntiles = nt * (nt + 1) / 2;
    // panel buffers U[2][n][17], W[n][17], P, Dg + flag; the final mirrored store wants a 16 x 17 block per wave (16 waves at most)
    const size_t lds = std::max(((size_t)3 * n * 17 + 2 * 16 * 17) * 8 + 16, (size_t)16 * 16 * 17 * 8);
#define L(S, NW)                                                                                                              \
    do {                                                                                                                      \
        HM_HIP(hipFuncSetAttribute((const void*)k_spd_inverse<S, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((k_spd_inverse<S, NW>), dim3(1), dim3(64 * NW), lds, s, G, nparts, n, ridge, W, flag, add, add_scale, rank1, rank1_scale);  \
    } while (0)
    if (ntiles <= 12) L(2, 8);
    else if (ntiles <= 24) L(2, 16);
    else if (ntiles <= 60 && g_spd_small) L(10, 8);  // 8 waves, <= 128 registers: fits on a CU beside a contraction workgroup
    else if (ntiles <= 60) L(5, 16);
    else if (ntiles <= 96) L(8, 16);
    else L(12, 16);
#undef L
    HM_HIP(hipGetLastError());
    return 0;
}

// Self-test hook (host buffers): W = inv(G + ridge I) through k_spd_inverse.
// F = block L D L^T factors of G - rank1_scale rank1 rank1^T + add_scale add (lower tiles of G and add are read).  Returns 0 if
// launched, -1 if not applicable (n not a multiple of 16, or the factor does not fit the gain kernel's LDS), > 0 on error.
int ldl_factor_mfma(hipStream_t s, const double* G, int n, double* F, int* flag, const double* add, double add_scale,
                    const double* rank1, double rank1_scale) {
    if (n % 16 != 0 || n < 16 || n > 176) return -1;
    const int nt = n / 16, ntiles = nt * (nt + 1) / 2;
    const size_t lds = ((size_t)4 * n * 17 + 5 * 16 * 17) * 8 + 32;
#define L(S, NW)                                                                                                             \
    do {                                                                                                                     \
        HM_HIP(hipFuncSetAttribute((const void*)k_ldl_factor<S, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((k_ldl_factor<S, NW>), dim3(1), dim3(64 * NW), lds, s, G, n, F, flag, add, add_scale, rank1, rank1_scale); \
    } while (0)
    if (ntiles <= 12) L(2, 8);
    else if (ntiles <= 24) L(2, 16);
    else if (ntiles <= 60) L(5, 16);
    else L(6, 16);
#undef L
    HM_HIP(hipGetLastError());
    return 0;
}

// A_T (n x N, fp32) = (X B^-1)^T from the factors F of B.  X: N x n (fp64, row-major).
int ldl_gain_mfma(hipStream_t s, const double* F, int n, const double* X, int N, float* A_T) {
    if (n % 16 != 0 || n < 16 || n > 176 || N < 1) return -1;
    const int nt = n / 16;
    const size_t lds = (size_t)(nt * (nt + 1) / 2) * 16 * 17 * 8 + 2 * 4 * 64 * 8;
    const dim3 grid((N + 15) / 16), block(256);
#define L(NT) case NT: HM_HIP(hipFuncSetAttribute((const void*)k_ldl_gain<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
                       hipLaunchKernelGGL(k_ldl_gain<NT>, grid, block, lds, s, F, n, X, N, A_T); break
    switch (nt) { L(1); L(2); L(3); L(4); L(5); L(6); L(7); L(8); L(9); L(10); L(11); default: return -1; }
#undef L
    HM_HIP(hipGetLastError());
    return 0;
}

// Both in one launch (k_ldl_chain), of G - rank1_scale rank1 rank1^T + add_scale add + ridge I.  colflag: one int of device memory that is
// ZERO when the kernel starts (the caller resets it on the same stream).  Returns -1 where the pair above applies but this form does not (n = 16 nt with nt outside [2, 11]).
int ldl_chain_mfma(hipStream_t s, const double* G, int n, double* F, int* flag, const double* add, double add_scale, const double* rank1,
                   double rank1_scale, int* colflag, const double* X, int N, float* A_T, double ridge) {
    if (n % 16 != 0 || n < 32 || n > 176 || N < 1) return -1;
    const int nt = n / 16, ntiles = nt * (nt + 1) / 2;
    const size_t lds_f = ((size_t)4 * n * 17 + 5 * 16 * 17) * 8 + 32, lds_g = (size_t)ntiles * 16 * 17 * 8 + 2 * 4 * 64 * 8;
    const size_t lds = std::max(lds_f, lds_g);
    const dim3 grid(1 + (N + 15) / 16), block(1024);
#define L(S, NT)                                                                                                                  \
    case NT:                                                                                                                      \
        HM_HIP(hipFuncSetAttribute((const void*)k_ldl_chain<S, 16, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));   \
        hipLaunchKernelGGL((k_ldl_chain<S, 16, NT>), grid, block, lds, s, G, n, F, flag, add, add_scale, rank1, rank1_scale, colflag, X, N, A_T, ridge); \
        break
    switch (nt) { L(2, 2); L(2, 3); L(2, 4); L(2, 5); L(2, 6); L(5, 7); L(5, 8); L(5, 9); L(5, 10); L(6, 11); default: return -1; }
#undef L
    HM_HIP(hipGetLastError());
    return 0;
}

// test hook: A_T = (X inv(G + ridge I))^T through the factorisation
extern "C" int hm_debug_ldl_gain(hm_ctx* ctx, int n, int N, const double* G, double ridge, const double* X, float* A_T) {
    HM_REQUIRE(ctx && G && X && A_T, "hm_debug_ldl_gain: NULL argument");
    HM_REQUIRE(n % 16 == 0 && n >= 16 && n <= 176 && N >= 1, "hm_debug_ldl_gain: n must be a multiple of 16 in [16, 176]");
    HM_HIP(hipSetDevice(ctx->device));
    std::vector<double> Gr(G, G + (size_t)n * n);
    for (int i = 0; i < n; ++i) Gr[(size_t)i * n + i] += ridge;
    double *dG, *dF, *dX;
    float* dA;
    int* dflag;
    HM_HIP(hipMalloc(&dG, (size_t)n * n * 8));
    HM_HIP(hipMalloc(&dF, (size_t)n * n * 8));
    HM_HIP(hipMalloc(&dX, (size_t)N * n * 8));
    HM_HIP(hipMalloc(&dA, (size_t)N * n * 4));
    HM_HIP(hipMalloc(&dflag, 4));
    HM_HIP(hipMemset(dflag, 0, 4));
    HM_HIP(hipMemcpy(dG, Gr.data(), (size_t)n * n * 8, hipMemcpyHostToDevice));
    HM_HIP(hipMemcpy(dX, X, (size_t)N * n * 8, hipMemcpyHostToDevice));
    int rc = ldl_factor_mfma(ctx->stream, dG, n, dF, dflag, nullptr, 0.0, nullptr, 0.0);
    if (rc == 0) rc = ldl_gain_mfma(ctx->stream, dF, n, dX, N, dA);
    int flag = 0;
    if (rc == 0 && (hipStreamSynchronize(ctx->stream) != hipSuccess || hipMemcpy(A_T, dA, (size_t)N * n * 4, hipMemcpyDeviceToHost) != hipSuccess ||
                    hipMemcpy(&flag, dflag, 4, hipMemcpyDeviceToHost) != hipSuccess)) {
        hm_set_error("hm_debug_ldl_gain: device error");
        rc = 1;
    }
    (void)hipFree(dG); (void)hipFree(dF); (void)hipFree(dX); (void)hipFree(dA); (void)hipFree(dflag);
    if (rc < 0) { hm_set_error("hm_debug_ldl_gain: not applicable"); rc = 2; }
    if (!rc && flag) { hm_set_error("hm_debug_ldl_gain: non-positive pivot"); rc = 4; }
    return rc;
}

extern "C" int hm_debug_spd_inverse(hm_ctx* ctx, int n, const double* G, double ridge, double* W) {
    HM_REQUIRE(ctx && G && W, "hm_debug_spd_inverse: NULL argument");
    HM_REQUIRE(n % 16 == 0 && n >= 16 && n <= 256, "hm_debug_spd_inverse: n must be a multiple of 16 in [16, 256]");
    HM_HIP(hipSetDevice(ctx->device));
    double *dG, *dW;
    int* dflag;
    const size_t bytes = (size_t)n * n * 8;
    HM_HIP(hipMalloc(&dG, bytes));
    HM_HIP(hipMalloc(&dW, bytes));
    HM_HIP(hipMalloc(&dflag, 4));
    HM_HIP(hipMemset(dflag, 0, 4));
    HM_HIP(hipMemcpy(dG, G, bytes, hipMemcpyHostToDevice));
    int rc = spd_inverse_mfma(ctx->stream, dG, 1, n, ridge, dW, dflag, nullptr, 0.0, nullptr, 0.0);
    int flag = 0;
    if (rc == 0 && (hipStreamSynchronize(ctx->stream) != hipSuccess || hipMemcpy(W, dW, bytes, hipMemcpyDeviceToHost) != hipSuccess ||
                    hipMemcpy(&flag, dflag, 4, hipMemcpyDeviceToHost) != hipSuccess)) {
        hm_set_error("hm_debug_spd_inverse: device error");
        rc = 1;
    }
    (void)hipFree(dG); (void)hipFree(dW); (void)hipFree(dflag);
    if (!rc && flag) { hm_set_error("hm_debug_spd_inverse: non-positive pivot"); rc = 4; }
    return rc;
}

// Localised analysis of an fp32 plan on the matrix cores.  Returns 0 if launched, -1 if not applicable, >0 on error.
int local_analysis_mfma(hipStream_t s, int M, int n_obs, int N_total, double cutoff, const float* taper, const double* G,
                        const float* Gxt, float* Wt, int* flag) {
    if (n_obs > 240) return -1;  // selection by the first 4 waves; LDS
    const int nmax = ((n_obs + 15) / 16) * 16 + 16;
    const int ntmax = nmax / 16, ntiles = ntmax * (ntmax + 1) / 2;
    const size_t lds = ((size_t)3 * nmax * 17 + 2 * 16 * 17 + 2 * n_obs) * 8 + (size_t)n_obs * 4 + 32;
    if (lds > 160 * 1024 - 256) return -1;
#define L(S)                                                                                                                    \
    do {                                                                                                                        \
        HM_HIP(hipFuncSetAttribute((const void*)k_local_analysis_mfma<S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL(k_local_analysis_mfma<S>, dim3(M), dim3(1024), lds, s, M, n_obs, N_total, cutoff, taper, G, Gxt, Wt, flag); \
    } while (0)
    if (ntiles <= 36) L(3);
    else if (ntiles <= 72) L(6);
    else if (ntiles <= 108) L(9);
    else L(12);
#undef L
    HM_HIP(hipGetLastError());
    return 0;
}
