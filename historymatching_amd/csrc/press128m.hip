// press128m.hip -- Ny = 128 fp64 pressure step with the Schur-complement inversion on the MATRIX CORES.
//
// Same block elimination as press128.hip / the generic kernel, but the symmetric Gauss-Jordan
// sweep is blocked: 4 pivots at a time.  With U = A[:,K] the 128x4 panel of the 4 pivot columns and
// P = A[K,K]^-1 (4x4), one block sweep is
//        A      <- A - (U P) U^T        rank-4 update of the whole 128x128 block  -> v_mfma_f64_16x16x4_f64
//        A[:,K] <- U P,  A[K,:] <- (U P)^T,  A[K,K] <- -P
// and after the 32 panels A = -inv(S_i).  The block lives in the MFMA accumulators: 16 waves x (2x2 tiles of
// 16x16) x 4 fp64 per lane.  Per panel only the panel (128x4 doubles) and P (16 doubles) cross waves, through a
// double-buffered LDS area; every register index is a compile-time constant (the panel loop is unrolled 8x over
// the tile parity and the within-tile column group).
// C/D layout of v_mfma_f64_16x16x4_f64: lane l, reg g -> row (l>>4)+4g, col l&15; A operand: lane l holds
// A[l&15][l>>4]; B operand: B[l>>4][l&15] (checked on hardware by tests/test_forward_gpu.py::test_mfma_f64_layout).
#include <type_traits>

#include "fwd_dev.h"

namespace {

constexpr int NB = 128;
typedef double d4 __attribute__((ext_vector_type(4)));

struct __attribute__((aligned(16))) PressLds {
    double U[2][NB][4];  // panel columns (current values), double buffered
    double Pm[2][16];    // inverse of the 4x4 pivot block
    double ev[NB], dgv[NB], tyv[NB + 8], yprev[NB], ycur[NB];
    double red[4][NB];
};

__device__ __forceinline__ double rcp_newton(double d) {
    double x = __builtin_amdgcn_rcp(d);
    double e = fma(-d, x, 1.0);
    x = fma(x, e, x);
    e = fma(-d, x, 1.0);
    x = fma(x, e, x);
    return x;
}

struct Geo {
    int w, wr, wc, lane, lc, lq;
};

// Wave grid: NW = 16 waves -> 4 (tile rows) x 4 (tile cols) waves, 2x2 tiles each;
//            NW =  8 waves -> 2 x 4 waves, 4x2 tiles each (two such workgroups share a CU and hide each other's
//            per-panel latency chain: pivot-block inverse -> MFMA -> fix-ups -> publish).
template <int NW>
struct Cfg {
    static constexpr int TRW = (NW == 16) ? 2 : 4;  // tile rows per wave
    static constexpr int TCW = 2;                   // tile cols per wave
    static constexpr int NT = 64 * NW;
};

// t[row] = sum_col acc[row][col] v[col]; result returned to threads tid < 128 (row = tid). Contains barriers.
template <int NW>
__device__ __forceinline__ double matvec_tiles(const d4 (&acc)[Cfg<NW>::TRW][2], const double* __restrict__ v, PressLds& L,
                                               const Geo& g, int tid) {
    constexpr int TRW = Cfg<NW>::TRW;
    const double v0 = v[16 * (2 * g.wc) + g.lc], v1 = v[16 * (2 * g.wc + 1) + g.lc];
#pragma unroll
    for (int ti = 0; ti < TRW; ++ti)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double s = fma(acc[ti][1][r], v1, acc[ti][0][r] * v0);
#pragma unroll
            for (int msk = 8; msk >= 1; msk >>= 1) s += __shfl_xor(s, msk, 16);
            if (g.lc == 0) L.red[g.wc][16 * (TRW * g.wr + ti) + g.lq + 4 * r] = s;
        }
    __syncthreads();
    double t = 0.0;
    if (tid < NB) t = (L.red[0][tid] + L.red[1][tid]) + (L.red[2][tid] + L.red[3][tid]);
    __syncthreads();
    return t;
}

__device__ __forceinline__ double dot4(const double* __restrict__ u, const double (&p)[4]) {
    double s = u[0] * p[0];
    s = fma(u[1], p[1], s);
    s = fma(u[2], p[2], s);
    s = fma(u[3], p[3], s);
    return s;
}

// One block-sweep panel WITHOUT look-ahead (publish -> barrier -> 4x4 inverse by one wave -> barrier -> update): pivot columns k0 .. k0+3 of tile column Cp = 4*cq + CP4, k0 = 16*Cp + 4*GQ.
template <int NW, int CP4, int GQ>
__device__ __forceinline__ void panel_simple(d4 (&acc)[Cfg<NW>::TRW][2], PressLds& L, int& cur, int cq, const Geo& g, int& bad) {
    constexpr int TRW = Cfg<NW>::TRW;
    constexpr int TJ = CP4 & 1;                          // tile column inside the owning wave
    constexpr int TI = (TRW == 2) ? (CP4 & 1) : CP4;     // tile row inside the owning wave
    const int wc_role = 2 * cq + (CP4 >> 1);             // wave column owning tile column Cp
    const int wr_role = (TRW == 2) ? wc_role : cq;       // wave row owning tile row Rp = Cp
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
    // B: P = inverse of the 4x4 pivot block, by one wave (all its lanes redundantly), via 4 rank-1 sweeps
    if (g.wr == wr_role && g.wc == wc_role) {
        double a[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) a[i][j] = U[k0 + i][j];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const double d = a[kk][kk];
            if (!(d > 0.0)) bad = 1;
            const double pinv = rcp_newton(d);
            double col[4], tcl[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                col[r] = a[r][kk];
                tcl[r] = col[r] * pinv;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    double v = fma(-col[r], tcl[c], a[r][c]);
                    if (r == kk) v = (c == kk) ? -pinv : tcl[c];
                    else if (c == kk) v = tcl[r];
                    a[r][c] = v;
                }
        }
        if (g.lane == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) Pm[4 * i + j] = -a[i][j];
        }
    }
    __syncthreads();
    // C: rank-4 update on the matrix cores
    double Prow[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) Prow[j] = Pm[4 * g.lq + j];  // P symmetric: P[j][lq] = P[lq][j]
    double ufr[2];
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) ufr[tj] = U[16 * (2 * g.wc + tj) + g.lc][g.lq];
#pragma unroll
    for (int ti = 0; ti < TRW; ++ti) {
        const double wfr = -dot4(U[16 * (TRW * g.wr + ti) + g.lc], Prow);  // -(U P)[16R + lc][lq]
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(wfr, ufr[tj], acc[ti][tj], 0, 0, 0);
    }
    // D: rows and columns of the panel take their swept values
    if (g.wr == wr_role) {  // tile row Rp (ti = TI), register GQ: rows k0+lq, all columns: A[k][c] = (U P)[c][k-k0]
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) acc[TI][tj][GQ] = dot4(U[16 * (2 * g.wc + tj) + g.lc], Prow);
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
}

// Publish the 4 pivot columns k0..k0+3 (tile column Cp = 4*cq + CP4, column group GQ) into LDS buffer `buf`, and let
// the wave that owns the diagonal tile invert the 4x4 pivot block right away: it reads back rows it has just written
// itself (LDS operations of one wave are ordered), so no workgroup barrier is needed before the inversion.
template <int NW, int CP4, int GQ>
__device__ __forceinline__ void publish_panel(const d4 (&acc)[Cfg<NW>::TRW][2], PressLds& L, int buf, int cq, const Geo& g, int& bad) {
    constexpr int TRW = Cfg<NW>::TRW;
    constexpr int TJ = CP4 & 1;
    const int wc_role = 2 * cq + (CP4 >> 1);
    const int wr_role = (TRW == 2) ? wc_role : cq;
    const int k0 = 16 * (4 * cq + CP4) + 4 * GQ;
    double (*U)[4] = L.U[buf];
    if (g.wc == wc_role) {
        if ((g.lc >> 2) == GQ) {
#pragma unroll
            for (int ti = 0; ti < TRW; ++ti)
#pragma unroll
                for (int r = 0; r < 4; ++r) U[16 * (TRW * g.wr + ti) + g.lq + 4 * r][g.lc & 3] = acc[ti][TJ][r];
        }
        if (g.wr == wr_role) {
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
                    for (int j = 0; j < 4; ++j) L.Pm[buf][4 * i + j] = a[(i >= j) ? (i * (i + 1) / 2 + j) : (j * (j + 1) / 2 + i)];
            }
        }
    }
}

// One block-sweep panel with look-ahead.  On entry (after a barrier) the panel U_j and P_j = inv(pivot block) are in
// LDS buffer `cur`.  The waves that own the NEXT panel's tile column update those tiles first, give them their
// swept rows/columns, publish U_{j+1} (and P_{j+1}) into buffer cur^1, and only then issue the rest of their MFMAs;
// everybody else streams its MFMAs.  One barrier per panel: the latency chain
//     fragments -> 2 MFMAs -> fix-ups -> publish -> 4x4 inverse
// runs concurrently with the 64 MFMAs of the rank-4 update.
#ifdef PRESS_STAMPS
#define STAMP(i) do { long long t_ = clock64(); stamps[i] += t_ - tprev; tprev = t_; } while (0)
#define STAMP_ARGS , long long (&stamps)[16], long long& tprev
#define STAMP_PASS , stamps, tprev
#else
#define STAMP(i) do {} while (0)
#define STAMP_ARGS
#define STAMP_PASS
#endif

template <int NW, int CP4, int GQ>
__device__ __forceinline__ void panel(d4 (&acc)[Cfg<NW>::TRW][2], PressLds& L, int& cur, int cq, const Geo& g, int& bad STAMP_ARGS) {
    constexpr int TRW = Cfg<NW>::TRW;
    constexpr int TJ = CP4 & 1;                          // tile column inside the owning wave
    constexpr int TI = (TRW == 2) ? (CP4 & 1) : CP4;     // tile row inside the owning wave
    constexpr bool SAME_COL = GQ < 3;                    // next panel in the same tile column?
    constexpr int CP4n = SAME_COL ? CP4 : ((CP4 + 1) & 3);
    constexpr int GQn = SAME_COL ? GQ + 1 : 0;
    constexpr int TJn = CP4n & 1;
    const int cqn = (!SAME_COL && CP4 == 3) ? cq + 1 : cq;
    const bool has_next = cqn < 2;
    const int wc_role = 2 * cq + (CP4 >> 1);             // wave column owning tile column Cp
    const int wr_role = (TRW == 2) ? wc_role : cq;       // wave row owning tile row Rp = Cp
    const int wc_next = 2 * cqn + (CP4n >> 1);
    const bool la = has_next && g.wc == wc_next;         // this wave owns the next panel's tile column
    double (*U)[4] = L.U[cur];
    const double* Pm = L.Pm[cur];

    double Prow[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) Prow[j] = Pm[4 * g.lq + j];  // P symmetric: P[j][lq] = P[lq][j]
    double ufr[2], wfr[TRW];
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) ufr[tj] = U[16 * (2 * g.wc + tj) + g.lc][g.lq];
#pragma unroll
    for (int ti = 0; ti < TRW; ++ti) wfr[ti] = -dot4(U[16 * (TRW * g.wr + ti) + g.lc], Prow);  // -(U P)[16R + lc][lq]

    // swept values of panel j for the tiles of tile-column slot tjs of this wave
    auto fix = [&](auto tjs_c) {
        constexpr int tjs = decltype(tjs_c)::value;
        if (g.wr == wr_role)  // rows k0+lq (register GQ of tile row TI): A[k][c] = (U P)[c][k-k0]
            acc[TI][tjs][GQ] = dot4(U[16 * (2 * g.wc + tjs) + g.lc], Prow);
        if (tjs == TJ && g.wc == wc_role) {  // columns k0..k0+3: A[r][k] = (U P)[r][k-k0]
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
    };
    using std::integral_constant;
    STAMP(0);  // fragments
    if (la) {
#pragma unroll
        for (int ti = 0; ti < TRW; ++ti)
            acc[ti][TJn] = __builtin_amdgcn_mfma_f64_16x16x4f64(wfr[ti], ufr[TJn], acc[ti][TJn], 0, 0, 0);
        STAMP(1);  // la: issue look-ahead MFMAs
        fix(integral_constant<int, TJn>{});
        STAMP(2);  // la: wait for them + fix-ups
        publish_panel<NW, CP4n, GQn>(acc, L, cur ^ 1, cqn, g, bad);
        STAMP(3);  // la: publish (+ 4x4 inverse on the diagonal wave)
#pragma unroll
        for (int ti = 0; ti < TRW; ++ti)
            acc[ti][1 - TJn] = __builtin_amdgcn_mfma_f64_16x16x4f64(wfr[ti], ufr[1 - TJn], acc[ti][1 - TJn], 0, 0, 0);
        fix(integral_constant<int, 1 - TJn>{});
        STAMP(4);  // la: remaining MFMAs + fix-ups
    } else {
#pragma unroll
        for (int ti = 0; ti < TRW; ++ti)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(wfr[ti], ufr[tj], acc[ti][tj], 0, 0, 0);
        STAMP(5);  // other: issue MFMAs
        fix(integral_constant<int, 0>{});
        fix(integral_constant<int, 1>{});
        STAMP(6);  // other: fix-ups (incl. waiting for MFMA results when this wave has a role)
    }
    __syncthreads();
#ifdef PRESS_STAMPS
    { long long t_ = clock64(); stamps[la ? 7 : 8] += t_ - tprev; tprev = t_; stamps[la ? 9 : 10] += 1; }
#endif
    cur ^= 1;
}

template <typename TS, int NW, bool LA>
__global__ __launch_bounds__(64 * NW) void k_press128m(FwdParams p, const TS* __restrict__ S_base, long long S_stride, int k) {
    constexpr int TRW = Cfg<NW>::TRW;
    constexpr int NT = Cfg<NW>::NT;
    constexpr int NCH = TRW * 2 * 2;  // 16-byte chunks per lane
    __shared__ PressLds L;
    const int m = blockIdx.x;
    const int tid = threadIdx.x;
    Geo g;
    g.lane = tid & 63;
    g.w = tid >> 6;
    g.wr = g.w >> 2;
    g.wc = g.w & 3;
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

    d4 acc[TRW][2];
    int bad = 0, cur = 0;
#ifdef PRESS_STAMPS
    long long stamps[16] = {0}, tprev = 0, tstart = clock64();
#endif
    for (int i = 0; i < Nx; ++i) {
        if (tid < NB) {
            const int j = tid;
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
                for (int tj = 0; tj < 2; ++tj) {
                    const double ec = L.ev[16 * (2 * g.wc + tj) + g.lc];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        acc[ti][tj][r] = -(L.ev[16 * (TRW * g.wr + ti) + g.lq + 4 * r] * acc[ti][tj][r] * ec);
                }
        } else {
            if (tid < NB) L.ycur[tid] = q[tid];
#pragma unroll
            for (int ti = 0; ti < TRW; ++ti)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = d4{0.0, 0.0, 0.0, 0.0};
        }
        // add the tridiagonal D_i
#pragma unroll
        for (int ti = 0; ti < TRW; ++ti)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * (TRW * g.wr + ti) + g.lq + 4 * r, col = 16 * (2 * g.wc + tj) + g.lc;
                    if (row == col) acc[ti][tj][r] += L.dgv[row];
                    else if (col == row + 1) acc[ti][tj][r] -= L.tyv[col];
                    else if (row == col + 1) acc[ti][tj][r] -= L.tyv[row];
                }
        // 32 block-sweep panels: A <- -inv(A)
        if (LA) {
            publish_panel<NW, 0, 0>(acc, L, cur, 0, g, bad);
            __syncthreads();
        }
#ifdef PRESS_STAMPS
        tprev = clock64();
#endif
#define PANEL(a, b) do { if (LA) panel<NW, a, b>(acc, L, cur, cq, g, bad STAMP_PASS); else panel_simple<NW, a, b>(acc, L, cur, cq, g, bad); } while (0)
        for (int cq = 0; cq < 2; ++cq) {
            PANEL(0, 0); PANEL(0, 1); PANEL(0, 2); PANEL(0, 3);
            PANEL(1, 0); PANEL(1, 1); PANEL(1, 2); PANEL(1, 3);
            PANEL(2, 0); PANEL(2, 1); PANEL(2, 2); PANEL(2, 3);
            PANEL(3, 0); PANEL(3, 1); PANEL(3, 2); PANEL(3, 3);
        }
        // G_i = -A: keep in the accumulators for the next block, stream to HBM (16-byte chunks, thread-major)
        double2* Gi = G + (long long)i * (NB * NB / 2);
#pragma unroll
        for (int ti = 0; ti < TRW; ++ti)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj) {
                acc[ti][tj] = -acc[ti][tj];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    double2 v;
                    v.x = acc[ti][tj][2 * h];
                    v.y = acc[ti][tj][2 * h + 1];
                    Gi[(((ti * 2 + tj) * 2) + h) * NT + tid] = v;
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
            for (int c = 0; c < NCH; ++c) {
                double2 v = Gi[c * NT + tid];
                acc[c >> 2][(c >> 1) & 1][2 * (c & 1)] = v.x;
                acc[c >> 2][(c >> 1) & 1][2 * (c & 1) + 1] = v.y;
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
#ifdef PRESS_STAMPS
    // diagnostic build only: per-phase cycle sums of three waves of member 0 -> tail of the TX scratch
    if (m == 0 && g.lane == 0 && (g.w == 0 || g.w == 5 || g.w == 3)) {
        stamps[11] = clock64() - tstart;
        long long* dbg = reinterpret_cast<long long*>(p.TX + (long long)p.N * (Nx + 1) * NB) - 64 + (g.w == 0 ? 0 : (g.w == 5 ? 16 : 32));
        for (int s_ = 0; s_ < 16; ++s_) dbg[s_] = stamps[s_];
    }
#endif
}

__global__ void k_mfma_f64_probe(const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ D) {
    const int l = threadIdx.x;
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}

}  // namespace

// Returns 0 if launched, >0 on error, -1 if this specialisation does not apply.
// press_variant 0: 16-wave workgroups, two barriers per panel (fastest today);  4: 16-wave with look-ahead
// (one barrier per panel; slower today: the fp64 VALU work of the latency chain stalls behind the other waves'
// fp64 MFMAs);  3: 8-wave look-ahead.
int launch_pressure_128m(hm_fwd* f, const void* S, long long S_stride, int k) {
    const FwdParams& p = f->p;
    if (p.Ny != NB) return -1;
    hipStream_t s = f->ctx->stream;
    const int v = f->press_variant;
#define LAUNCH(TS, NW, LA) hipLaunchKernelGGL((k_press128m<TS, NW, LA>), dim3(p.N), dim3(64 * NW), 0, s, p, (const TS*)S, S_stride, k)
    if (f->dtype == 64) {
        if (v == 3) LAUNCH(double, 8, true);
        else if (v == 4) LAUNCH(double, 16, true);
        else LAUNCH(double, 16, false);
    } else {
        if (v == 3) LAUNCH(float, 8, true);
        else if (v == 4) LAUNCH(float, 16, true);
        else LAUNCH(float, 16, false);
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
