// press128m.hip -- Ny = 128 fp64 pressure step with the Schur-complement inversion on the MATRIX CORES.
//
// Same block elimination as press128.hip / the generic kernel, but the symmetric Gauss-Jordan
// sweep is blocked: 4 pivots at a time.  With U = A[:,K] the 128x4 panel of the 4 pivot columns and
// P = A[K,K]^-1 (4x4), one block sweep is
//        A      <- A - (U P) U^T        rank-4 update of the whole 128x128 block  -> v_mfma_f64_16x16x4_f64
//        A[:,K] <- U P,  A[K,:] <- (U P)^T,  A[K,K] <- -P
// and after the 32 panels A = -inv(S_i).  The block lives in the MFMA accumulators, spread over the NW waves of the
// workgroup as a WR x WC grid of waves, each holding TRW x TCW tiles of 16x16 (4 fp64 per lane per tile).  Per panel
// only the panel (128x4 doubles) and P (16 doubles) cross waves, through a double-buffered LDS area; every register
// index is a compile-time constant (the panel loop is unrolled 16x over the tile-column residue and column group).
//
// The kernel is bound by the per-panel dependency chain (publish -> 4x4 inverse -> fragments -> MFMA -> swept rows/
// columns), not by MFMA throughput: cycle stamps put the chain at ~4700 cycles per panel with 16 waves, of which
// the two LDS hand-offs through 1024-thread barriers are ~600 cycles each (diag/lat.hip).  (A look-ahead variant -- next panel's columns updated and published before the
// bulk MFMAs, one barrier per panel -- was measured slower: the fp64 VALU work of the chain stalls behind the other
// waves' fp64 MFMAs, which share the DP units; it lives in the git history.)
//
// C/D layout of v_mfma_f64_16x16x4_f64: lane l, reg g -> row (l>>4)+4g, col l&15; A operand: lane l holds
// A[l&15][l>>4]; B operand: B[l>>4][l&15] (checked on hardware by tests/test_forward_gpu.py::test_mfma_f64_layout).
#include <type_traits>

#include "fwd_dev.h"

namespace {

constexpr int NB = 128;
typedef double d4 __attribute__((ext_vector_type(4)));

struct __attribute__((aligned(16))) PressLds {
    double U[2][NB][4];  // panel columns (current values), double buffered           (rank-4 panels)
    double Pm[2][16];    // inverse of the 4x4 pivot block
    double ev[NB], dgv[NB], tyv[NB + 8], yprev[NB], ycur[NB];
    double red[4][NB];
};

// LDS of the rank-16 pipeline: one panel = one whole tile column (16 pivots)
struct __attribute__((aligned(16))) PressLds16 {
    // (static LDS must stay below 64 KB: beyond that the kernel needs the dynamic-LDS opt-in, and out-of-range
    //  LDS reads silently return 0)
    // rows padded to 17 doubles: with 16 the 16 lanes of an operand read (row = lane&15) are 8-way bank conflicted
    double U[2][NB][17];   // the 16 pivot columns (current values); double buffered: panel j+1 publishes while
                           // slower waves still read panel j's U in their rank-16 update
    double W[NB][17];      // U P      (written after barrier 1, read after barrier 2: single buffer is safe)
    double P[16][17];      // inverse of the 16x16 diagonal tile (written before barrier 1, read before barrier 2)
    double cb[16];         // pivot column of the in-wave 16x16 inversion
    double ev[NB], dgv[NB], tyv[NB + 8], yprev[NB], ycur[NB];
    double red[4][NB];
};

template <int NW>
struct Cfg {
    static constexpr int WC = 4;                      // wave grid columns
    static constexpr int WR = NW / WC;                // wave grid rows
    static constexpr int TRW = 8 / WR;                // tile rows per wave
    static constexpr int TCW = 8 / WC;                // tile cols per wave
    static constexpr int NT = 64 * NW;
};

struct Geo {
    int w, wr, wc, lane, lc, lq;
};

__device__ __forceinline__ double rcp_newton(double d) {
    double x = __builtin_amdgcn_rcp(d);
    double e = fma(-d, x, 1.0);
    x = fma(x, e, x);
    e = fma(-d, x, 1.0);
    x = fma(x, e, x);
    return x;
}

__device__ __forceinline__ double dot4(const double* __restrict__ u, const double (&p)[4]) {
    double s = u[0] * p[0];
    s = fma(u[1], p[1], s);
    s = fma(u[2], p[2], s);
    s = fma(u[3], p[3], s);
    return s;
}

// t[row] = sum_col acc[row][col] v[col]; result returned to threads tid < 128 (row = tid). Contains barriers.
template <int NW, typename LDS>
__device__ __forceinline__ double matvec_tiles(const d4 (&acc)[Cfg<NW>::TRW][Cfg<NW>::TCW], const double* __restrict__ v,
                                               LDS& L, const Geo& g, int tid) {
    constexpr int TRW = Cfg<NW>::TRW, TCW = Cfg<NW>::TCW, WC = Cfg<NW>::WC;
    double vv[TCW];
#pragma unroll
    for (int tj = 0; tj < TCW; ++tj) vv[tj] = v[16 * (TCW * g.wc + tj) + g.lc];
#pragma unroll
    for (int ti = 0; ti < TRW; ++ti)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double s = acc[ti][0][r] * vv[0];
#pragma unroll
            for (int tj = 1; tj < TCW; ++tj) s = fma(acc[ti][tj][r], vv[tj], s);
#pragma unroll
            for (int msk = 8; msk >= 1; msk >>= 1) s += __shfl_xor(s, msk, 16);
            if (g.lc == 0) L.red[g.wc][16 * (TRW * g.wr + ti) + g.lq + 4 * r] = s;
        }
    __syncthreads();
    double t = 0.0;
    if (tid < NB) {
        t = L.red[0][tid];
#pragma unroll
        for (int c = 1; c < WC; ++c) t += L.red[c][tid];
    }
    __syncthreads();
    return t;
}

// 4x4 SPD inverse by four rank-1 sweeps on the packed lower triangle (all lanes of the calling wave redundantly);
// s[r*(r+1)/2 + c], c <= r.  Returns P = A^-1 in the same packing.
__device__ __forceinline__ void inv4_sym(double (&s)[10], int& bad) {
#define SY(r, c) s[((r) >= (c)) ? ((r) * ((r) + 1) / 2 + (c)) : ((c) * ((c) + 1) / 2 + (r))]
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const double d = SY(kk, kk);
        if (!(d > 0.0)) bad = 1;
        const double pinv = rcp_newton(d);
        double col[4], tcl[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            col[r] = SY(r, kk);
            tcl[r] = col[r] * pinv;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c <= r; ++c) {
                double v = fma(-col[r], tcl[c], SY(r, c));
                if (r == kk) v = (c == kk) ? -pinv : tcl[c];
                else if (c == kk) v = tcl[r];
                SY(r, c) = v;
            }
    }
#pragma unroll
    for (int e = 0; e < 10; ++e) s[e] = -s[e];
#undef SY
}

// One block-sweep panel: pivot columns k0 .. k0+3 of tile column Cp = 4*cq + CP4, k0 = 16*Cp + 4*GQ.
template <int NW, int CP4, int GQ>
__device__ __forceinline__ void panel(d4 (&acc)[Cfg<NW>::TRW][Cfg<NW>::TCW], PressLds& L, int& cur, int cq, const Geo& g, int& bad) {
    constexpr int TRW = Cfg<NW>::TRW, TCW = Cfg<NW>::TCW;
    constexpr int TJ = CP4 % TCW;                              // tile column inside the owning wave
    constexpr int TI = CP4 % TRW;                              // tile row inside the owning wave
    const int wc_role = cq * (4 / TCW) + CP4 / TCW;            // wave column owning tile column Cp
    const int wr_role = cq * (4 / TRW) + CP4 / TRW;            // wave row owning tile row Rp = Cp
    const int k0 = 16 * (4 * cq + CP4) + 4 * GQ;
    double (*U)[4] = L.U[cur];
    double* Pm = L.Pm[cur];
    // A: publish the panel columns (owners: waves of tile-column Cp, lanes holding columns k0..k0+3)
    if (g.wc == wc_role && (g.lc >> 2) == GQ) {
#pragma unroll
        for (int ti = 0; ti < TRW; ++ti)
#pragma unroll
            for (int r = 0; r < 4; ++r) U[16 * (TRW * g.wr + ti) + g.lq + 4 * r][g.lc & 3] = acc[ti][TJ][r];
    }
    __syncthreads();
    // B: P = inverse of the 4x4 pivot block, by the wave owning the diagonal tile (all its lanes redundantly)
    if (g.wr == wr_role && g.wc == wc_role) {
        double a[10];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) a[i * (i + 1) / 2 + j] = U[k0 + i][j];
        inv4_sym(a, bad);
        if (g.lane == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) Pm[4 * i + j] = a[(i >= j) ? (i * (i + 1) / 2 + j) : (j * (j + 1) / 2 + i)];
        }
    }
    __syncthreads();
    // C: rank-4 update on the matrix cores
    double Prow[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) Prow[j] = Pm[4 * g.lq + j];  // P symmetric: P[j][lq] = P[lq][j]
    double ufr[TCW];
#pragma unroll
    for (int tj = 0; tj < TCW; ++tj) ufr[tj] = U[16 * (TCW * g.wc + tj) + g.lc][g.lq];
#pragma unroll
    for (int ti = 0; ti < TRW; ++ti) {
        const double wfr = -dot4(U[16 * (TRW * g.wr + ti) + g.lc], Prow);  // -(U P)[16R + lc][lq]
#pragma unroll
        for (int tj = 0; tj < TCW; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(wfr, ufr[tj], acc[ti][tj], 0, 0, 0);
    }
    // D: rows and columns of the panel take their swept values
    if (g.wr == wr_role) {  // tile row Rp (ti = TI), register GQ: rows k0+lq, all columns: A[k][c] = (U P)[c][k-k0]
#pragma unroll
        for (int tj = 0; tj < TCW; ++tj) acc[TI][tj][GQ] = dot4(U[16 * (TCW * g.wc + tj) + g.lc], Prow);
    }
    if (g.wc == wc_role) {  // tile column Cp (tj = TJ), lanes with columns k0..k0+3: A[r][k] = (U P)[r][k-k0]
        const bool mine = (g.lc >> 2) == GQ;
        double Pc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) Pc[j] = Pm[4 * (g.lc & 3) + j];
#pragma unroll
        for (int ti = 0; ti < TRW; ++ti)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double s = dot4(U[16 * (TRW * g.wr + ti) + g.lq + 4 * r], Pc);
                acc[ti][TJ][r] = mine ? s : acc[ti][TJ][r];
            }
        if (g.wr == wr_role && mine) acc[TI][TJ][GQ] = -Pm[4 * g.lq + (g.lc & 3)];  // pivot block itself: -P
    }
    cur ^= 1;
}

// ------------------------------------------------------------------------------------------------------------
// Rank-16 panels (16 waves, 2x2 tiles per wave): one panel = one whole tile column Cp = 2*cp2 + CPP.
//   1. the 4 waves owning tile column Cp publish U = A[:, 16Cp..16Cp+15]; the wave owning the diagonal tile inverts
//      it IN-WAVE (16 symmetric sweeps; the pivot column goes through a wave-private LDS line: LDS operations of one
//      wave are ordered, so no workgroup barrier inside the 16-pivot chain) and publishes P           -> barrier
//   2. the owners form W = U P with 4 MFMAs per tile (operands re-read from LDS in operand layout), publish W and
//      keep it as their swept tile column (the diagonal tile becomes -P)                              -> barrier
//   3. everybody: A <- A - W U^T on the remaining tiles (4 MFMAs per tile); the tile row takes W^T.
// Two barriers per 16 pivots instead of two per 4: the sequential pivot chain stays inside one wave.
// ------------------------------------------------------------------------------------------------------------
template <int CPP>
__device__ __forceinline__ void panel16(d4 (&acc)[2][2], PressLds16& L, int& cur, int cp2, const Geo& g, int& bad) {
    constexpr int TJ = CPP, TI = CPP;
    const int Cp = 2 * cp2 + CPP;
    const bool col_owner = g.wc == cp2, row_owner = g.wr == cp2;
    double (*U)[17] = L.U[cur];
    double (*W)[17] = L.W;
    double (*P)[17] = L.P;
    if (col_owner) {
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int r = 0; r < 4; ++r) U[16 * (2 * g.wr + ti) + g.lq + 4 * r][g.lc] = acc[ti][TJ][r];
        if (row_owner) {
            // in-wave inverse of the diagonal tile: entry (row = lq + 4r, col = lc)
            d4 t = acc[TI][TJ];
            for (int k = 0; k < 16; ++k) {
                if (g.lc == k) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) L.cb[g.lq + 4 * r] = t[r];
                }
                // lanes exchange data through LDS inside one wave: the hardware keeps a wave's LDS operations in
                // order, but the compiler must be told that the other lanes' stores precede these loads (otherwise
                // it may run the "else" lanes' loads ahead of the "then" lanes' stores)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                double cr[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) cr[r] = L.cb[g.lq + 4 * r];
                const double cc = L.cb[g.lc], d = L.cb[k];
                if (!(d > 0.0)) bad = 1;
                const double pinv = rcp_newton(d);
                const double tc = cc * pinv;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = g.lq + 4 * r;
                    double v = fma(-cr[r], tc, t[r]);
                    if (row == k) v = (g.lc == k) ? -pinv : tc;
                    else if (g.lc == k) v = cr[r] * pinv;
                    t[r] = v;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            // t = -inv(diagonal tile): publish P = -t and keep -P = t as the swept diagonal tile
#pragma unroll
            for (int r = 0; r < 4; ++r) P[g.lq + 4 * r][g.lc] = -t[r];
            acc[TI][TJ] = t;
        }
    }
    __syncthreads();
    if (col_owner) {
        // W tile = U tile * P  (rows 16R.., R = 2*wr + ti);  the diagonal tile's rows are never used
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
            if (row_owner && ti == TI) continue;
            const int R = 2 * g.wr + ti;
            d4 w = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                w = __builtin_amdgcn_mfma_f64_16x16x4f64(U[16 * R + g.lc][4 * kk + g.lq], P[4 * kk + g.lq][g.lc], w, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) W[16 * R + g.lq + 4 * r][g.lc] = w[r];
            acc[ti][TJ] = w;  // swept tile column: A[r][K] = (U P)[r]
        }
    }
    __syncthreads();
    // rank-16 update of every tile outside tile row / tile column Cp; the tile row takes W^T
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
        const int R = 2 * g.wr + ti;
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
            const int C = 2 * g.wc + tj;
            if (C == Cp) continue;  // swept column (done by the owners above)
            if (R == Cp) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[ti][tj][r] = W[16 * C + g.lc][g.lq + 4 * r];
            } else {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                    acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(-W[16 * R + g.lc][4 * kk + g.lq], U[16 * C + g.lc][4 * kk + g.lq],
                                                                       acc[ti][tj], 0, 0, 0);
            }
        }
    }
    cur ^= 1;
}

template <typename TS, int NW, bool R16>
__global__ __launch_bounds__(64 * NW) void k_press128m(FwdParams p, const TS* __restrict__ S_base, long long S_stride, int k) {
    constexpr int TRW = Cfg<NW>::TRW, TCW = Cfg<NW>::TCW, WC = Cfg<NW>::WC;
    constexpr int NT = Cfg<NW>::NT;
    __shared__ typename std::conditional<R16, PressLds16, PressLds>::type L;
    const int m = blockIdx.x;
    const int tid = threadIdx.x;
    Geo g;
    g.lane = tid & 63;
    g.w = tid >> 6;
    g.wr = g.w / WC;
    g.wc = g.w % WC;
    g.lc = g.lane & 15;
    g.lq = g.lane >> 4;
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
    const double* q = p.q + (long long)(p.q_cols > 1 ? k : 0) * Nxy;

    assemble_transmissibilities<TS>(p, S, Km, P /* scratch for L */, TX, TY, tid, NT);

    d4 acc[TRW][TCW];
    int bad = 0, cur = 0;
    for (int i = 0; i < Nx; ++i) {
        for (int j = tid; j < NB; j += NT) {
            const double y1 = TY[i * (NB + 1) + j], y2 = TY[i * (NB + 1) + j + 1];
            const double x1 = TX[i * NB + j], x2 = TX[(i + 1) * NB + j];
            double dg = y1 + y2 + x1 + x2;
            if (i == 0 && j == 0) dg += Km[0] + Km[0];  // SPD pin: A[0,0] += Kx[0,0]+Ky[0,0]
            L.dgv[j] = dg;
            L.tyv[j] = y1;
            if (j == NB - 1) L.tyv[NB] = y2;
            L.ev[j] = x1;
        }
        __syncthreads();
        if (i > 0) {
            const double t = matvec_tiles<NW>(acc, L.yprev, L, g, tid);
            if (tid < NB) L.ycur[tid] = q[i * NB + tid] + L.ev[tid] * t;
#pragma unroll
            for (int ti = 0; ti < TRW; ++ti)
#pragma unroll
                for (int tj = 0; tj < TCW; ++tj) {
                    const double ec = L.ev[16 * (TCW * g.wc + tj) + g.lc];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        acc[ti][tj][r] = -(L.ev[16 * (TRW * g.wr + ti) + g.lq + 4 * r] * acc[ti][tj][r] * ec);
                }
        } else {
            if (tid < NB) L.ycur[tid] = q[tid];
#pragma unroll
            for (int ti = 0; ti < TRW; ++ti)
#pragma unroll
                for (int tj = 0; tj < TCW; ++tj) acc[ti][tj] = d4{0.0, 0.0, 0.0, 0.0};
        }
        // add the tridiagonal D_i
#pragma unroll
        for (int ti = 0; ti < TRW; ++ti)
#pragma unroll
            for (int tj = 0; tj < TCW; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * (TRW * g.wr + ti) + g.lq + 4 * r, col = 16 * (TCW * g.wc + tj) + g.lc;
                    if (row == col) acc[ti][tj][r] += L.dgv[row];
                    else if (col == row + 1) acc[ti][tj][r] -= L.tyv[col];
                    else if (row == col + 1) acc[ti][tj][r] -= L.tyv[row];
                }
        __syncthreads();
        // 32 block-sweep panels: A <- -inv(A)
        if constexpr (R16) {
            for (int cp2 = 0; cp2 < 4; ++cp2) {
                panel16<0>(acc, L, cur, cp2, g, bad);
                panel16<1>(acc, L, cur, cp2, g, bad);
            }
        } else {
#define PANEL(a, b) panel<NW, a, b>(acc, L, cur, cq, g, bad)
            for (int cq = 0; cq < 2; ++cq) {
                PANEL(0, 0); PANEL(0, 1); PANEL(0, 2); PANEL(0, 3);
                PANEL(1, 0); PANEL(1, 1); PANEL(1, 2); PANEL(1, 3);
                PANEL(2, 0); PANEL(2, 1); PANEL(2, 2); PANEL(2, 3);
                PANEL(3, 0); PANEL(3, 1); PANEL(3, 2); PANEL(3, 3);
            }
#undef PANEL
        }
        // G_i = -A: keep in the accumulators for the next block, stream to HBM (16-byte chunks, thread-major)
        double2* Gi = G + (long long)i * (NB * NB / 2);
#pragma unroll
        for (int ti = 0; ti < TRW; ++ti)
#pragma unroll
            for (int tj = 0; tj < TCW; ++tj) {
                acc[ti][tj] = -acc[ti][tj];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    double2 v;
                    v.x = acc[ti][tj][2 * h];
                    v.y = acc[ti][tj][2 * h + 1];
                    Gi[(((ti * TCW + tj) * 2) + h) * NT + tid] = v;
                }
            }
        if (tid < NB) {
            yv[i * NB + tid] = L.ycur[tid];
            L.yprev[tid] = L.ycur[tid];
        }
        __syncthreads();
    }
    // back substitution: x_i = G_i (y_i + TX[i+1] * x_{i+1});  ycur holds x_{i+1}
    for (int i = Nx - 1; i >= 0; --i) {
        if (i < Nx - 1) {
            const double2* Gi = G + (long long)i * (NB * NB / 2);
#pragma unroll
            for (int ti = 0; ti < TRW; ++ti)
#pragma unroll
                for (int tj = 0; tj < TCW; ++tj)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        double2 v = Gi[(((ti * TCW + tj) * 2) + h) * NT + tid];
                        acc[ti][tj][2 * h] = v.x;
                        acc[ti][tj][2 * h + 1] = v.y;
                    }
        }
        if (tid < NB) {
            double v = yv[i * NB + tid];
            if (i < Nx - 1) v += TX[(i + 1) * NB + tid] * L.ycur[tid];
            L.yprev[tid] = v;
        }
        __syncthreads();
        const double t = matvec_tiles<NW>(acc, L.yprev, L, g, tid);
        if (tid < NB) {
            L.ycur[tid] = t;
            P[i * NB + tid] = t;
        }
        __syncthreads();
    }
    face_fluxes(p, P, TX, TY, Vx, Vy, tid, NT);
    if (bad && g.lane == 0) atomicOr(&p.status[m], HM_MEMBER_BAD_PIVOT);
}

__global__ void k_mfma_f64_probe(const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ D) {
    const int l = threadIdx.x;
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}

}  // namespace

// Returns 0 if launched, >0 on error, -1 if this specialisation does not apply.
// press_variant 0 (and any other value): 16 waves, rank-16 panels;  4: 16 waves, rank-4 panels;  3: 8 waves, rank-4.
int launch_pressure_128m(hm_fwd* f, const void* S, long long S_stride, int k) {
    const FwdParams& p = f->p;
    if (p.Ny != NB) return -1;
    hipStream_t s = f->ctx->stream;
    const int v = f->press_variant;
#define LAUNCH(TS, NW, R16) hipLaunchKernelGGL((k_press128m<TS, NW, R16>), dim3(p.N), dim3(64 * NW), 0, s, p, (const TS*)S, S_stride, k)
    if (f->dtype == 64) {
        if (v == 3) LAUNCH(double, 8, false);
        else if (v == 4) LAUNCH(double, 16, false);
        else LAUNCH(double, 16, true);
    } else {
        if (v == 3) LAUNCH(float, 8, false);
        else if (v == 4) LAUNCH(float, 16, false);
        else LAUNCH(float, 16, true);
    }
#undef LAUNCH
    HM_HIP(hipGetLastError());
    return 0;
}

// Self-test hook: D(16x16) = A(16x4) B(4x16) through one v_mfma_f64_16x16x4_f64 with the operand/result lane maps this
// file assumes.  Host buffers.
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
