// press_nd.hip -- fp64 pressure step of the 128 x 128 grid by NESTED DISSECTION (multifrontal), SURVEY.md A.3's system.
//
// Replaces the sparse direct solve inside ResSim.sim (notebooks/HistoryMatch.py:362; TPFA-ResSim) for
// Nx = Ny = 128: the block elimination along ix of press128s.hip costs 302 Mflop per member and time step and has a serial
// chain of 1024 pivot tiles; the geometric dissection of the grid (nd.h) needs about 70 Mflop (profiles/tools/nd_flops.py) and
// a chain of 24.  Numerically it is a block L D L^T factorisation in another elimination order: same pin, same pivot-tile
// inverses (sweep16.h), same fp64 matrix-core products.
//
// Data flow per member and time step (four launches):
//   k_nd_assemble   TX, TY (bit-exact with the oracle, fwd_dev.h) and the matrix diagonal dg
//   k_nd_sub        levels 10..7: ONE WAVE per level-7 subtree (8 x 16 cells, 15 fronts in post-order).  A front with one
//                   pivot tile lives entirely in the wave's registers: with the pivot panel held TRANSPOSED (V_R = F21_R^T,
//                   16 pivots x 16 front rows, accumulator layout) every product the factorisation needs has the form
//                   Y^T Z of two register tiles, which v_mfma_f64_16x16x4_f64 computes straight from the accumulator
//                   layout (register kk of Y as A operand = Y^T's k-slice, register kk of Z as B operand):
//                       W_R^T = P V_R  (P = inverse pivot tile, symmetric),   F22[R, C] -= (W_R^T)^T V_C.
//                   No operand staging through LDS at all.  Children's update matrices (packed lower triangles, levels 8..10
//                   in per-wave LDS slots) are GATHERED into the parent's tiles through the position tables.
//   k_nd_wave       levels 6 and 5: one wave per front, children and update in the member's arena (global memory)
//   k_nd_top        levels 4..0: one WORKGROUP per member; fronts with 2..8 pivot tiles, their tiles dealt to the waves'
//                   registers, the current panel (W^T and V per row tile, register images) broadcast through LDS
//   k_nd_solve      back substitution root -> leaves, x1 = -W^T [x2; -1] per panel, then the face fluxes
// The right-hand side rides along as one extra boundary row of every front (its W^T column is z1 = P r1; the (rhs, boundary)
// entries of the update are the reduced right-hand side), so forward elimination costs nothing extra.
#include "fwd_dev.h"
#include "nd_build.h"
#include "sweep16.h"

namespace {

constexpr int NB = 128;

struct NdDev {
    const int* fronts;
    const int* cells;
    const short* cpos;
    double* fact;
    double* arena;
    double* dg;
    long long fact_stride, arena_stride;
    int slot8, slot9, slot10;  // doubles per LDS update slot of levels 8, 9, 10
};

struct NdGeo {
    int lane, lc, lq;
};

struct NdMem {  // one member's arrays
    const double *dg, *TX, *TY, *q;
    double *fact, *arena;
};

__device__ __forceinline__ int tri(int a, int b) {
    const int hi = a > b ? a : b, lo = a > b ? b : a;
    return ((hi * (hi + 1)) >> 1) + lo;
}

// A[cm, ck] (five-point system; cell = ix * 128 + iy) or, for cm = -2, the right-hand side q[ck]; identity on padded pivots.
__device__ __forceinline__ double nd_coef(const NdMem& mm, int cm, int ck, bool same_pos) {
    double v = 0.0;
    if (ck >= 0) {
        if (cm >= 0) {
            const int d = cm - ck;
            const double* src = nullptr;
            if (d == 0) src = mm.dg + ck;
            else if (d == NB) src = mm.TX + ck + NB;
            else if (d == -NB) src = mm.TX + ck;
            else if (d == 1) src = mm.TY + ck + (ck >> 7) + 1;
            else if (d == -1) src = mm.TY + ck + (ck >> 7);
            if (src) {
                v = *src;
                if (d != 0) v = -v;
            }
        } else if (cm == -2) v = mm.q[ck];
    } else if (same_pos) v = 1.0;
    return v;
}

template <typename GEO>
__device__ __forceinline__ void sweep16_partial(d4& t, const GEO& g, int& bad, int kreg) {
    double mypinv = 0.0;
#define S(K) sweep16_step<K, GEO>(t, mypinv, g, bad);
    S(0) S(1) S(2) S(3)
    if (kreg > 1) { S(4) S(5) S(6) S(7) }
    if (kreg > 2) { S(8) S(9) S(10) S(11) }
    if (kreg > 3) { S(12) S(13) S(14) S(15) }
#undef S
#pragma unroll
    for (int r = 0; r < 4; ++r) t[r] *= mypinv;
}

__device__ __forceinline__ void nd_wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ------------------------------------------------------------------------------------------------------------------------
// One front with ONE pivot tile, processed by one wave in registers.  ch0 / ch1: the children's packed update matrices
// (KIDS), out: this front's packed update matrix ((b + 1)(b + 2) / 2 doubles; nullptr for none).
// ------------------------------------------------------------------------------------------------------------------------
template <int MAXBT, bool KIDS>
__device__ __forceinline__ void nd_wave_front(const NdDev& nd, const NdMem& mm, int f, const double* ch0, const double* ch1,
                                              double* out, const NdGeo& g, int& bad) {
    const int* F = nd.fronts + f * ND_FRONT_INTS;
    const int b = __builtin_amdgcn_readfirstlane(F[NDF_B]);
    const int bt = __builtin_amdgcn_readfirstlane(F[NDF_BT]);
    const int co = __builtin_amdgcn_readfirstlane(F[NDF_CELLS]);
    const int kreg = __builtin_amdgcn_readfirstlane(F[NDF_KREG]);
    const int* cl = nd.cells + co;
    const short* cp0 = nd.cpos + 2 * co;
    const short* cp1 = cp0 + 16 * (1 + bt);
    int ck[4], pk0[4], pk1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        ck[r] = cl[4 * r + g.lq];
        pk0[r] = KIDS ? cp0[4 * r + g.lq] : -1;
        pk1[r] = KIDS ? cp1[4 * r + g.lq] : -1;
    }
    // ---- the pivot panel, transposed: V[R][r] = F[front row 16 R + lc][pivot 4 r + lq]
    d4 V[MAXBT + 1];
#pragma unroll
    for (int R = 0; R <= MAXBT; ++R) {
        if (R <= bt) {
            const int pm = 16 * R + g.lc;
            const int cm = cl[pm];
            const int pm0 = KIDS ? cp0[pm] : -1, pm1 = KIDS ? cp1[pm] : -1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double v = nd_coef(mm, cm, ck[r], R == 0 && g.lc == 4 * r + g.lq);
                if (KIDS) {
                    if (pk0[r] >= 0 && pm0 >= 0) v += ch0[tri(pk0[r], pm0)];
                    if (pk1[r] >= 0 && pm1 >= 0) v += ch1[tri(pk1[r], pm1)];
                }
                V[R][r] = v;
            }
        } else {
            V[R] = d4{0.0, 0.0, 0.0, 0.0};
        }
    }
    // ---- P = inverse of the pivot tile
    d4 P = V[0];
    sweep16_partial(P, g, bad, kreg);  // P = -inv
    // ---- W_R^T = P V_R (stored negated: the products below subtract), factor rows to memory
    double* fa = mm.fact + F[NDF_FACT];
    d4 WTn[MAXBT];
#pragma unroll
    for (int R = 1; R <= MAXBT; ++R) {
        d4 w = {0.0, 0.0, 0.0, 0.0};
        if (R <= bt) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                if (kk < kreg) w = __builtin_amdgcn_mfma_f64_16x16x4f64(P[kk], V[R][kk], w, 0, 0, 0);
            // w = -W^T (P is the negated inverse)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r < kreg) fa[((R - 1) * kreg + r) * 64 + g.lane] = -w[r];
        }
        WTn[R - 1] = w;
    }
    if (!out) return;
    // ---- trailing tiles: update = children - W V^T, packed lower
#pragma unroll
    for (int R = 1; R <= MAXBT; ++R) {
        if (R > bt) break;
        int pr0[4], pr1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pr0[r] = KIDS ? cp0[16 * R + 4 * r + g.lq] : -1;
            pr1[r] = KIDS ? cp1[16 * R + 4 * r + g.lq] : -1;
        }
#pragma unroll
        for (int C = 1; C <= R; ++C) {
            d4 acc = {0.0, 0.0, 0.0, 0.0};
            if (KIDS) {
                const int pc0 = cp0[16 * C + g.lc], pc1 = cp1[16 * C + g.lc];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double v = 0.0;
                    if (pr0[r] >= 0 && pc0 >= 0) v += ch0[tri(pr0[r], pc0)];
                    if (pr1[r] >= 0 && pc1 >= 0) v += ch1[tri(pr1[r], pc1)];
                    acc[r] = v;
                }
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                if (kk < kreg) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(WTn[R - 1][kk], V[C][kk], acc, 0, 0, 0);
            const int j = 16 * (C - 1) + g.lc;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * (R - 1) + 4 * r + g.lq;
                if (j <= i && i <= b) out[((i * (i + 1)) >> 1) + j] = acc[r];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// TX, TY and the matrix diagonal (one workgroup per member).
// ------------------------------------------------------------------------------------------------------------------------
template <typename TS>
__global__ __launch_bounds__(1024) void k_nd_assemble(FwdParams p, NdDev nd, const TS* __restrict__ S_base, long long S_stride) {
    const int m = blockIdx.x, tid = threadIdx.x;
    const int Nx = p.Nx, Nxy = p.Nxy;
    const TS* S = S_base + (long long)m * S_stride;
    const double* Km = p.K + (long long)m * Nxy;
    const double* Kym = p.Ky ? p.Ky + (long long)m * Nxy : Km;
    double* TX = p.TX + (long long)m * (Nx + 1) * NB;
    double* TY = p.TY + (long long)m * Nx * (NB + 1);
    double* dg = nd.dg + (long long)m * Nxy;
    assemble_transmissibilities<TS>(p, S, Km, Kym, p.P + (long long)m * Nxy, TX, TY, tid, 1024);
    for (int c = tid; c < Nxy; c += 1024) {
        const int ty = c + (c >> 7);
        double d = TY[ty] + TY[ty + 1] + TX[c] + TX[c + NB];
        if (c == 0) d += Km[0] + Kym[0];  // SPD pin: A[0,0] += Kx[0,0] + Ky[0,0]
        dg[c] = d;
    }
}

__device__ __forceinline__ NdMem nd_member(const FwdParams& p, const NdDev& nd, int m, int k) {
    NdMem mm;
    mm.dg = nd.dg + (long long)m * p.Nxy;
    mm.TX = p.TX + (long long)m * (p.Nx + 1) * NB;
    mm.TY = p.TY + (long long)m * p.Nx * (NB + 1);
    mm.q = p.q + (long long)m * p.q_mstride + (long long)(p.q_cols > 1 ? k : 0) * p.Nxy;
    mm.fact = nd.fact + (long long)m * nd.fact_stride;
    mm.arena = nd.arena + (long long)m * nd.arena_stride;
    return mm;
}

// ------------------------------------------------------------------------------------------------------------------------
// Levels 10..7: one wave per level-7 subtree, 4 waves per workgroup, 32 workgroups per member.
// ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 3) void k_nd_sub(FwdParams p, NdDev nd, int k) {
    extern __shared__ double nd_lds[];
    const int m = blockIdx.x >> 5, g32 = blockIdx.x & 31;
    const int tid = threadIdx.x;
    NdGeo g;
    g.lane = tid & 63;
    g.lc = g.lane & 15;
    g.lq = g.lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const NdMem mm = nd_member(p, nd, m, k);
    double* base = nd_lds + w * 2 * (nd.slot8 + nd.slot9 + nd.slot10);
    double* s8 = base;
    double* s9 = base + 2 * nd.slot8;
    double* s10 = s9 + 2 * nd.slot9;
    int bad = 0;
    const int i7 = 4 * g32 + w;
    for (int a = 0; a < 2; ++a) {
        const int i8 = 2 * i7 + a;
        for (int bq = 0; bq < 2; ++bq) {
            const int i9 = 2 * i8 + bq;
            for (int cq = 0; cq < 2; ++cq) {
                const int i10 = 2 * i9 + cq;
                nd_wave_front<1, false>(nd, mm, 1023 + i10, nullptr, nullptr, s10 + cq * nd.slot10, g, bad);
            }
            nd_wave_fence();
            nd_wave_front<2, true>(nd, mm, 511 + i9, s10, s10 + nd.slot10, s9 + bq * nd.slot9, g, bad);
            nd_wave_fence();
        }
        nd_wave_front<2, true>(nd, mm, 255 + i8, s9, s9 + nd.slot9, s8 + a * nd.slot8, g, bad);
        nd_wave_fence();
    }
    const int f7 = 127 + i7;
    nd_wave_front<3, true>(nd, mm, f7, s8, s8 + nd.slot8, mm.arena + nd.fronts[f7 * ND_FRONT_INTS + NDF_UPD], g, bad);
    if (bad && g.lane == 0) atomicOr(&p.status[m], HM_MEMBER_BAD_PIVOT);
}

// Levels 6 and 5: one wave per front, children and update in the arena.
template <int LEVEL, int MAXBT>
__global__ __launch_bounds__(256) void k_nd_wave(FwdParams p, NdDev nd, int k) {
    constexpr int NF = 1 << LEVEL, WPB = 4, BPM = NF / WPB;
    const int m = blockIdx.x / BPM, blk = blockIdx.x % BPM;
    const int tid = threadIdx.x;
    NdGeo g;
    g.lane = tid & 63;
    g.lc = g.lane & 15;
    g.lq = g.lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const NdMem mm = nd_member(p, nd, m, k);
    const int f = NF - 1 + blk * WPB + w;
    const int* F = nd.fronts + f * ND_FRONT_INTS;
    const double* ch0 = mm.arena + nd.fronts[F[NDF_C0] * ND_FRONT_INTS + NDF_UPD];
    const double* ch1 = mm.arena + nd.fronts[F[NDF_C1] * ND_FRONT_INTS + NDF_UPD];
    int bad = 0;
    nd_wave_front<MAXBT, true>(nd, mm, f, ch0, ch1, mm.arena + F[NDF_UPD], g, bad);
    if (bad && g.lane == 0) atomicOr(&p.status[m], HM_MEMBER_BAD_PIVOT);
}

// ------------------------------------------------------------------------------------------------------------------------
// Levels 4..0: one workgroup (8 waves) per member, front after front; the tiles of a front are dealt to the waves.
//   V tiles  (q, R), q < st, R >= q:   the transposed panel tile  F[rows of R][pivots of q]^T     (index = q-major)
//   trailing (R, C), st <= C <= R < T: the update matrix, accumulated over the panels in registers
// Per panel p:  S1 the owner of V(p, p) inverts it -> Pimg;  S2 the owners of V(p, R), R > p, form W_R^T = P V and publish
// W_R^T and V_R as register images (lane-major, conflict-free 8-byte reads) and store the factor;  S3 every later tile is
// updated with Y^T Z products of two published images.
// ------------------------------------------------------------------------------------------------------------------------
constexpr int TOP_NW = 8, TOP_NVS = 6, TOP_NTS = 7, TOP_MAXT = 13;

__device__ __forceinline__ d4 img_load(const double* img, int lane) {
    d4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = img[r * 64 + lane];
    return v;
}
__device__ __forceinline__ void img_store(double* img, int lane, const d4& v) {
#pragma unroll
    for (int r = 0; r < 4; ++r) img[r * 64 + lane] = v[r];
}

__global__ __launch_bounds__(64 * TOP_NW, 2) void k_nd_top(FwdParams p, NdDev nd, int k) {
    __shared__ double Pimg[256];
    __shared__ double Wimg[TOP_MAXT][256];
    __shared__ double Vimg[TOP_MAXT][256];
    const int m = blockIdx.x, tid = threadIdx.x;
    NdGeo g;
    g.lane = tid & 63;
    g.lc = g.lane & 15;
    g.lq = g.lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const NdMem mm = nd_member(p, nd, m, k);
    int bad = 0;
    for (int lv = 4; lv >= 0; --lv) {
        for (int fi = 0; fi < (1 << lv); ++fi) {
            const int f = (1 << lv) - 1 + fi;
            const int* F = nd.fronts + f * ND_FRONT_INTS;
            const int b = __builtin_amdgcn_readfirstlane(F[NDF_B]);
            const int st = __builtin_amdgcn_readfirstlane(F[NDF_ST]);
            const int bt = __builtin_amdgcn_readfirstlane(F[NDF_BT]);
            const int T = st + bt;
            const int co = __builtin_amdgcn_readfirstlane(F[NDF_CELLS]);
            const int kreg_last = __builtin_amdgcn_readfirstlane(F[NDF_KREG]);
            const int* cl = nd.cells + co;
            const short* cp0 = nd.cpos + 2 * co;
            const short* cp1 = cp0 + 16 * T;
            const double* ch0 = mm.arena + nd.fronts[F[NDF_C0] * ND_FRONT_INTS + NDF_UPD];
            const double* ch1 = mm.arena + nd.fronts[F[NDF_C1] * ND_FRONT_INTS + NDF_UPD];
            double* fa = mm.fact + F[NDF_FACT];
            const int nV = st * T - ((st * (st - 1)) >> 1);
            const int nT = b > 0 ? ((bt * (bt + 1)) >> 1) : 0;
            // ---- my tiles
            d4 vt[TOP_NVS], tr[TOP_NTS];
            int vq[TOP_NVS], vR[TOP_NVS], tR[TOP_NTS], tC[TOP_NTS];
#pragma unroll
            for (int s = 0; s < TOP_NVS; ++s) {
                const int idx = s * TOP_NW + w;
                int q = -1, R = -1;
                if (idx < nV) {
                    int rem = idx;
                    q = 0;
                    while (rem >= T - q) { rem -= T - q; ++q; }
                    R = q + rem;
                }
                vq[s] = q; vR[s] = R;
                vt[s] = d4{0.0, 0.0, 0.0, 0.0};
                if (q >= 0) {
                    const int pm = 16 * R + g.lc;
                    const int cm = cl[pm];
                    const int pm0 = cp0[pm], pm1 = cp1[pm];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int pk = 16 * q + 4 * r + g.lq;
                        const int ck = cl[pk];
                        double v = nd_coef(mm, cm, ck, pk == pm);
                        const int a0 = cp0[pk], a1 = cp1[pk];
                        if (a0 >= 0 && pm0 >= 0) v += ch0[tri(a0, pm0)];
                        if (a1 >= 0 && pm1 >= 0) v += ch1[tri(a1, pm1)];
                        vt[s][r] = v;
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < TOP_NTS; ++s) {
                const int idx = s * TOP_NW + w;
                int R = -1, C = -1;
                if (idx < nT) {
                    int rem = idx;
                    R = 0;
                    while (rem > R) { rem -= R + 1; ++R; }
                    C = rem;
                    R += st; C += st;
                }
                tR[s] = R; tC[s] = C;
                tr[s] = d4{0.0, 0.0, 0.0, 0.0};
                if (R >= 0) {
                    const int pc0 = cp0[16 * C + g.lc], pc1 = cp1[16 * C + g.lc];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int pr = 16 * R + 4 * r + g.lq;
                        const int a0 = cp0[pr], a1 = cp1[pr];
                        double v = 0.0;
                        if (a0 >= 0 && pc0 >= 0) v += ch0[tri(a0, pc0)];
                        if (a1 >= 0 && pc1 >= 0) v += ch1[tri(a1, pc1)];
                        tr[s][r] = v;
                    }
                }
            }
            // ---- panels
            int fo = 0;  // factor offset of panel p, in 64-double register rows
            for (int pp = 0; pp < st; ++pp) {
                const int kreg = pp == st - 1 ? kreg_last : 4;
                const int ipp = pp * T - ((pp * (pp - 1)) >> 1);  // index of V(pp, pp)
                if (w == ipp % TOP_NW) {
                    const int sl = ipp / TOP_NW;
#pragma unroll
                    for (int s = 0; s < TOP_NVS; ++s)
                        if (s == sl) {
                            d4 t = vt[s];
                            sweep16_partial(t, g, bad, kreg);
#pragma unroll
                            for (int r = 0; r < 4; ++r) Pimg[r * 64 + g.lane] = t[r];  // -inv
                        }
                }
                __syncthreads();
                {
                    const d4 Pn = img_load(Pimg, g.lane);
#pragma unroll
                    for (int s = 0; s < TOP_NVS; ++s) {
                        if (vq[s] == pp && vR[s] > pp) {
                            const int R = vR[s];
                            d4 wv = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                            for (int kk = 0; kk < 4; ++kk)
                                if (kk < kreg) wv = __builtin_amdgcn_mfma_f64_16x16x4f64(Pn[kk], vt[s][kk], wv, 0, 0, 0);
                            img_store(Wimg[R], g.lane, wv);  // -W^T
                            img_store(Vimg[R], g.lane, vt[s]);
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (r < kreg) fa[(fo + (R - pp - 1) * kreg + r) * 64 + g.lane] = -wv[r];
                        }
                    }
                }
                __syncthreads();
#pragma unroll
                for (int s = 0; s < TOP_NVS; ++s) {
                    if (vq[s] > pp) {  // V(q, R) -= V(pp, q)^T W(pp, R)^T
                        const d4 Y = img_load(Vimg[vq[s]], g.lane), Z = img_load(Wimg[vR[s]], g.lane);
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk)
                            if (kk < kreg) vt[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(Y[kk], Z[kk], vt[s], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int s = 0; s < TOP_NTS; ++s) {
                    if (tR[s] >= 0) {  // F22(R, C) -= W(pp, R) V(pp, C)^T
                        const d4 Y = img_load(Wimg[tR[s]], g.lane), Z = img_load(Vimg[tC[s]], g.lane);
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk)
                            if (kk < kreg) tr[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(Y[kk], Z[kk], tr[s], 0, 0, 0);
                    }
                }
                fo += (T - pp - 1) * kreg;
                // (the next panel's S1 writes Pimg only; its S2 rewrites the images behind the next barrier)
            }
            // ---- the update matrix to the arena
            if (nT > 0) {
                double* out = mm.arena + F[NDF_UPD];
#pragma unroll
                for (int s = 0; s < TOP_NTS; ++s) {
                    if (tR[s] >= 0) {
                        const int j = 16 * (tC[s] - st) + g.lc;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int i = 16 * (tR[s] - st) + 4 * r + g.lq;
                            if (j <= i && i <= b) out[((i * (i + 1)) >> 1) + j] = tr[s][r];
                        }
                    }
                }
            }
            __syncthreads();  // children before parents; the images are free again
        }
    }
    if (bad && g.lane == 0) atomicOr(&p.status[m], HM_MEMBER_BAD_PIVOT);
}

// ------------------------------------------------------------------------------------------------------------------------
// Back substitution, root to leaves: per front and panel (last first)  x1 = -W^T [x of the rows below; -1 for the rhs row],
// one wave per front, levels separated by workgroup barriers; pressures in P; then the face fluxes.
// ------------------------------------------------------------------------------------------------------------------------
constexpr int SOL_NW = 16;

__global__ __launch_bounds__(64 * SOL_NW) void k_nd_solve(FwdParams p, NdDev nd, int k) {
    __shared__ double xe_all[SOL_NW][16 * TOP_MAXT];
    const int m = blockIdx.x, tid = threadIdx.x;
    NdGeo g;
    g.lane = tid & 63;
    g.lc = g.lane & 15;
    g.lq = g.lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Nx = p.Nx, Nxy = p.Nxy;
    double* P = p.P + (long long)m * Nxy;
    const double* fact = nd.fact + (long long)m * nd.fact_stride;
    double* xe = xe_all[w];
    for (int lv = 0; lv < ND_LEVELS; ++lv) {
        const int nf = 1 << lv;
        for (int fi = w; fi < nf; fi += SOL_NW) {
            const int f = nf - 1 + fi;
            const int* F = nd.fronts + f * ND_FRONT_INTS;
            const int st = __builtin_amdgcn_readfirstlane(F[NDF_ST]);
            const int bt = __builtin_amdgcn_readfirstlane(F[NDF_BT]);
            const int T = st + bt;
            const int kreg_last = __builtin_amdgcn_readfirstlane(F[NDF_KREG]);
            const int* cl = nd.cells + __builtin_amdgcn_readfirstlane(F[NDF_CELLS]);
            const double* fa = fact + F[NDF_FACT];
            // boundary values (ancestors' pivots, already known), -1 on the right-hand-side row
            for (int pos = 16 * st + g.lane; pos < 16 * T; pos += 64) {
                const int c = cl[pos];
                xe[pos] = c >= 0 ? P[c] : (c == -2 ? -1.0 : 0.0);
            }
            nd_wave_fence();
            int fo = 0;
            for (int pp = 0; pp < st - 1; ++pp) fo += (T - pp - 1) * 4;
            for (int pp = st - 1; pp >= 0; --pp) {
                const int kreg = pp == st - 1 ? kreg_last : 4;
                double acc[4] = {0.0, 0.0, 0.0, 0.0};
                for (int R = pp + 1; R < T; ++R) {
                    const double xv = xe[16 * R + g.lc];
                    const double* tl = fa + (long long)(fo + (R - pp - 1) * kreg) * 64 + g.lane;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (r < kreg) acc[r] = fma(tl[r * 64], xv, acc[r]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double v = acc[r];
                    v += __shfl_xor(v, 8);
                    v += __shfl_xor(v, 4);
                    v += __shfl_xor(v, 2);
                    v += __shfl_xor(v, 1);
                    acc[r] = -v;
                }
                if (g.lc == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int pos = 16 * pp + 4 * r + g.lq;
                        const int c = cl[pos];
                        const double v = (r < kreg && c >= 0) ? acc[r] : 0.0;
                        xe[pos] = v;
                        if (c >= 0) P[c] = v;
                    }
                }
                nd_wave_fence();
                if (pp > 0) fo -= (T - pp) * 4;
            }
        }
        __syncthreads();
    }
    face_fluxes(p, P, p.TX + (long long)m * (Nx + 1) * NB, p.TY + (long long)m * Nx * (NB + 1), p.Vx + (long long)m * (Nx + 1) * NB,
                p.Vy + (long long)m * Nx * (NB + 1), tid, 64 * SOL_NW);
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------------
// Host side: tables built once per plan, buffers sized by the builder.
// ------------------------------------------------------------------------------------------------------------------------
struct hm_nd {
    NdInfo info{};
    DevBuf fronts, cells, cpos, fact, arena, dg;
    NdDev dev{};
};

bool pressure_nd_applies(const FwdParams& p) { return p.Nx == NB && p.Ny == NB; }

void hm_nd_free(hm_nd* n) {
    if (!n) return;
    DevBuf* bufs[] = {&n->fronts, &n->cells, &n->cpos, &n->fact, &n->arena, &n->dg};
    for (DevBuf* b : bufs) hm_dev_free(*b);
    delete n;
}

static int nd_setup(hm_fwd* f) {
    const FwdParams& p = f->p;
    NdTablesHost t;
    HM_REQUIRE(nd_build_tables(p.Nx, p.Ny, t), "nested-dissection tables: %s", t.error.c_str());
    HM_REQUIRE(t.info.max_bt[10] <= 1 && t.info.max_bt[9] <= 2 && t.info.max_bt[8] <= 2 && t.info.max_bt[7] <= 3 && t.info.max_bt[6] <= 4 &&
                   t.info.max_bt[5] <= 6, "nested-dissection tables: a wave-level front is larger than the kernels' register arrays");
    for (int lv = 0; lv <= 4; ++lv) {
        const int T = t.info.max_st[lv] + t.info.max_bt[lv];
        HM_REQUIRE(T <= TOP_MAXT, "nested-dissection tables: a top-level front has %d tile rows (max %d)", T, TOP_MAXT);
    }
    for (int lv = 5; lv < ND_LEVELS; ++lv) HM_REQUIRE(t.info.max_st[lv] == 1, "nested-dissection tables: level %d has several pivot tiles", lv);
    for (int fI = 0; fI < t.info.n_fronts; ++fI) {
        const int* F = &t.fronts[(size_t)fI * ND_FRONT_INTS];
        if (F[NDF_LEVEL] > 4) continue;
        const int st = F[NDF_ST], bt = F[NDF_BT], T = st + bt;
        const int nV = st * T - st * (st - 1) / 2, nT = F[NDF_B] > 0 ? bt * (bt + 1) / 2 : 0;
        HM_REQUIRE(nV <= TOP_NVS * TOP_NW && nT <= TOP_NTS * TOP_NW, "nested-dissection tables: front %d has %d + %d tiles", fI, nV, nT);
    }
    hm_nd* n = new hm_nd();
    n->info = t.info;
    int rc = 0;
    const size_t N = p.N;
    if ((rc = hm_dev_alloc(n->fronts, t.fronts.size() * 4)) || (rc = hm_dev_alloc(n->cells, t.cells.size() * 4)) ||
        (rc = hm_dev_alloc(n->cpos, t.cpos.size() * 2)) || (rc = hm_dev_alloc(n->fact, N * t.info.fact_doubles * 8)) ||
        (rc = hm_dev_alloc(n->arena, N * t.info.arena_doubles * 8)) || (rc = hm_dev_alloc(n->dg, N * p.Nxy * 8))) {
        hm_nd_free(n);
        return rc;
    }
    HM_HIP(hipMemcpy(n->fronts.p, t.fronts.data(), t.fronts.size() * 4, hipMemcpyHostToDevice));
    HM_HIP(hipMemcpy(n->cells.p, t.cells.data(), t.cells.size() * 4, hipMemcpyHostToDevice));
    HM_HIP(hipMemcpy(n->cpos.p, t.cpos.data(), t.cpos.size() * 2, hipMemcpyHostToDevice));
    NdDev& d = n->dev;
    d.fronts = (const int*)n->fronts.p;
    d.cells = (const int*)n->cells.p;
    d.cpos = (const short*)n->cpos.p;
    d.fact = (double*)n->fact.p;
    d.arena = (double*)n->arena.p;
    d.dg = (double*)n->dg.p;
    d.fact_stride = t.info.fact_doubles;
    d.arena_stride = t.info.arena_doubles;
    d.slot8 = t.info.lds_slot_doubles[0];
    d.slot9 = t.info.lds_slot_doubles[1];
    d.slot10 = t.info.lds_slot_doubles[2];
    f->nd = n;
    return 0;
}

// Returns 0 if launched, >0 on error, -1 if this specialisation does not apply.
int launch_pressure_nd(hm_fwd* f, const void* S, long long S_stride, int k) {
    const FwdParams& p = f->p;
    if (!pressure_nd_applies(p)) return -1;
    if (!f->nd) {
        int rc = nd_setup(f);
        if (rc) return rc;
    }
    hipStream_t s = f->ctx->stream;
    const NdDev& nd = f->nd->dev;
    if (f->dtype == 64) hipLaunchKernelGGL(k_nd_assemble<double>, dim3(p.N), dim3(1024), 0, s, p, nd, (const double*)S, S_stride);
    else hipLaunchKernelGGL(k_nd_assemble<float>, dim3(p.N), dim3(1024), 0, s, p, nd, (const float*)S, S_stride);
    const size_t lds = (size_t)4 * 2 * (nd.slot8 + nd.slot9 + nd.slot10) * 8;
    hipLaunchKernelGGL(k_nd_sub, dim3(p.N * 32), dim3(256), lds, s, p, nd, k);
    hipLaunchKernelGGL((k_nd_wave<6, 4>), dim3(p.N * 16), dim3(256), 0, s, p, nd, k);
    hipLaunchKernelGGL((k_nd_wave<5, 6>), dim3(p.N * 8), dim3(256), 0, s, p, nd, k);
    hipLaunchKernelGGL(k_nd_top, dim3(p.N), dim3(64 * TOP_NW), 0, s, p, nd, k);
    hipLaunchKernelGGL(k_nd_solve, dim3(p.N), dim3(64 * SOL_NW), 0, s, p, nd, k);
    HM_HIP(hipGetLastError());
    return 0;
}

extern "C" int hm_debug_nd_tables(int Nx, int Ny, long long* info, int* fronts, int* cells, short* cpos) {
    HM_REQUIRE(info, "hm_debug_nd_tables: NULL info");
    NdTablesHost t;
    if (!nd_build_tables(Nx, Ny, t)) {
        hm_set_error("hm_debug_nd_tables: %s (%d x %d)", t.error.c_str(), Nx, Ny);
        return 2;
    }
    info[0] = t.info.n_fronts;
    info[1] = t.info.n_cells;
    info[2] = t.info.fact_doubles;
    info[3] = t.info.arena_doubles;
    for (int i = 0; i < 3; ++i) info[4 + i] = t.info.lds_slot_doubles[i];
    for (int i = 0; i < ND_LEVELS; ++i) info[8 + i] = t.info.max_bt[i] * 16 + t.info.max_st[i];
    if (fronts) memcpy(fronts, t.fronts.data(), t.fronts.size() * sizeof(int));
    if (cells) memcpy(cells, t.cells.data(), t.cells.size() * sizeof(int));
    if (cpos) memcpy(cpos, t.cpos.data(), t.cpos.size() * sizeof(short));
    return 0;
}
