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
#include "sweep16.h"

#ifdef HM_PRESS_PROF
// Cycle stamps of workgroup 0 / thread 0 (diag/press_prof.py): built only with -DHM_PRESS_PROF.
__device__ long long hm_press_prof_buf[16];
#define PROF_DECL long long prof_t = clock64(), prof_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define PROF(i) do { const long long now_ = clock64(); prof_acc[i] += now_ - prof_t; prof_t = now_; } while (0)
#define PROF_ARGS , long long (&prof_acc)[16], long long& prof_t
#define PROF_PASS , prof_acc, prof_t
#else
#define PROF_DECL
#define PROF(i)
#define PROF_ARGS
#define PROF_PASS
#endif

namespace {

constexpr int NB = 128;

struct __attribute__((aligned(16))) PressLds {
    double U[2][NB][4];  // panel columns (current values), double buffered           (rank-4 panels)
    double Pm[2][16];    // inverse of the 4x4 pivot block
    double ev[NB], dgv[NB], tyv[NB + 8], yprev[NB], ycur[NB];
    double red[16][NB];  // mat-vec partial sums
    __device__ double* redbuf() { return &red[0][0]; }
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
    double ev[NB], dgv[NB], tyv[NB + 8], yprev[NB], ycur[NB];
    // mat-vec partial sums (16 x NB doubles) live in U: the panels are idle during the substitution mat-vecs
    __device__ double* redbuf() { return &U[0][0][0]; }
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

// t = A v for the symmetric block held in the tiles, evaluated column-wise: t[col] = sum_row A[row][col] v[row].
// In the accumulator layout a lane owns ONE column per tile column and 4*TRW rows of it, so its partial sums need no
// cross-lane reduction at all: 4*TRW fmas per tile column, one LDS store, and after a barrier thread `col` adds the
// 4*WR partials (lane-rows x wave-rows) of its column.  (The row-wise form reduced over the 16 lanes of a row with 4
// shuffle stages per output: 8.2k cycles per mat-vec against ~1k.)  Result returned to threads tid < 128.
template <int NW, typename LDS>
__device__ __forceinline__ double matvec_tiles(const d4 (&acc)[Cfg<NW>::TRW][Cfg<NW>::TCW], const double* __restrict__ v,
                                               LDS& L, const Geo& g, int tid) {
    constexpr int TRW = Cfg<NW>::TRW, TCW = Cfg<NW>::TCW, WR = Cfg<NW>::WR;
    double* red = L.redbuf();  // [WR * 4][NB]
    double s[TCW];
#pragma unroll
    for (int tj = 0; tj < TCW; ++tj) s[tj] = 0.0;
#pragma unroll
    for (int ti = 0; ti < TRW; ++ti)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double vr = v[16 * (TRW * g.wr + ti) + g.lq + 4 * r];
#pragma unroll
            for (int tj = 0; tj < TCW; ++tj) s[tj] = fma(acc[ti][tj][r], vr, s[tj]);
        }
#pragma unroll
    for (int tj = 0; tj < TCW; ++tj) red[(4 * g.wr + g.lq) * NB + 16 * (TCW * g.wc + tj) + g.lc] = s[tj];
    __syncthreads();
    double t = 0.0;
    if (tid < NB) {
        double a = red[tid], b = red[NB + tid];
#pragma unroll
        for (int c = 2; c < 4 * WR; c += 2) {
            a += red[c * NB + tid];
            b += red[(c + 1) * NB + tid];
        }
        t = a + b;
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
// Rank-16 panels (NW = 16 waves: 2x2 tiles per wave; NW = 8: 4x2 tiles per wave, two workgroups per CU):
// one panel = one whole tile column Cp = TRW*cpo + CPM.
//   1. the WR waves owning tile column Cp publish U = A[:, 16Cp..16Cp+15]; the wave owning the diagonal tile inverts
//      it IN-WAVE (sweep16_inwave: no workgroup barrier inside the 16-pivot chain) and publishes P       -> barrier
//   2. the owners form W = U P with 4 MFMAs per tile (operands re-read from LDS in operand layout), publish W and
//      keep it as their swept tile column (the diagonal tile becomes -P)                                 -> barrier
//   3. everybody: A <- A - W U^T on the remaining tiles (4 MFMAs per tile); the tile row takes W^T.
// Two barriers per 16 pivots instead of two per 4: the sequential pivot chain stays inside one wave.
// ------------------------------------------------------------------------------------------------------------
template <int NW, int CPM>
__device__ __forceinline__ void panel16(d4 (&acc)[Cfg<NW>::TRW][Cfg<NW>::TCW], PressLds16& L, int& cur, int cpo, const Geo& g,
                                        int& bad PROF_ARGS) {
    constexpr int TRW = Cfg<NW>::TRW, TCW = Cfg<NW>::TCW;
    constexpr int TI = CPM % TRW, TJ = CPM % TCW;
    const int Cp = TRW * cpo + CPM;
    const bool col_owner = g.wc == Cp / TCW, row_owner = g.wr == cpo;
    double (*U)[17] = L.U[cur];
    double (*W)[17] = L.W;
    double (*P)[17] = L.P;
    if (col_owner) {
#pragma unroll
        for (int ti = 0; ti < TRW; ++ti)
#pragma unroll
            for (int r = 0; r < 4; ++r) U[16 * (TRW * g.wr + ti) + g.lq + 4 * r][g.lc] = acc[ti][TJ][r];
        if (row_owner) {
            // in-wave inverse of the diagonal tile: entry (row = lq + 4r, col = lc)
            d4 t = acc[TI][TJ];
            sweep16_inwave(t, g, bad);
            // t = -inv(diagonal tile): publish P = -t and keep -P = t as the swept diagonal tile
#pragma unroll
            for (int r = 0; r < 4; ++r) P[g.lq + 4 * r][g.lc] = -t[r];
            acc[TI][TJ] = t;
        }
    }
    PROF(0);
    __syncthreads();
    PROF(1);
    if (col_owner) {
        // W tile = U tile * P  (rows 16R.., R = TRW*wr + ti);  the diagonal tile's rows are never used
#pragma unroll
        for (int ti = 0; ti < TRW; ++ti) {
            if (row_owner && ti == TI) continue;
            const int R = TRW * g.wr + ti;
            d4 w = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                w = __builtin_amdgcn_mfma_f64_16x16x4f64(U[16 * R + g.lc][4 * kk + g.lq], P[4 * kk + g.lq][g.lc], w, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) W[16 * R + g.lq + 4 * r][g.lc] = w[r];
            acc[ti][TJ] = w;  // swept tile column: A[r][K] = (U P)[r]
        }
    }
    PROF(2);
    __syncthreads();
    PROF(3);
    // rank-16 update of every tile outside tile row / tile column Cp; the tile row takes W^T.  k-slices outermost:
    // one slice needs TRW operands of W and TCW of U, each read from LDS once and shared by the wave's tiles.
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        double wv[TRW], uv[TCW];
#pragma unroll
        for (int ti = 0; ti < TRW; ++ti) wv[ti] = -W[16 * (TRW * g.wr + ti) + g.lc][4 * kk + g.lq];
#pragma unroll
        for (int tj = 0; tj < TCW; ++tj) uv[tj] = U[16 * (TCW * g.wc + tj) + g.lc][4 * kk + g.lq];
#pragma unroll
        for (int ti = 0; ti < TRW; ++ti)
#pragma unroll
            for (int tj = 0; tj < TCW; ++tj) {
                if (TCW * g.wc + tj == Cp || TRW * g.wr + ti == Cp) continue;  // swept column / row (wave-uniform)
                acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(wv[ti], uv[tj], acc[ti][tj], 0, 0, 0);
            }
    }
    if (row_owner) {
#pragma unroll
        for (int tj = 0; tj < TCW; ++tj) {
            const int C = TCW * g.wc + tj;
            if (C == Cp) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[TI][tj][r] = W[16 * C + g.lc][g.lq + 4 * r];
        }
    }
    PROF(4);
    cur ^= 1;
}

template <typename TS, int NW, bool R16>
__global__ __launch_bounds__(64 * NW, (R16 && NW == 8) ? 4 : 1) void k_press128m(FwdParams p, const TS* __restrict__ S_base, long long S_stride, int k) {
    constexpr int TRW = Cfg<NW>::TRW, TCW = Cfg<NW>::TCW, WC = Cfg<NW>::WC;
    constexpr int NT = Cfg<NW>::NT;
    __shared__ typename std::conditional<R16, PressLds16, PressLds>::type L;
    const int m = blockIdx.x;
    const int tid = threadIdx.x;
    Geo g;
    g.lane = tid & 63;
    g.w = tid >> 6;
    if constexpr (R16) {
        // waves are dealt round-robin to the 4 SIMDs: with wr = w % 4 the four owners of a tile column (equal wc), who
        // alone run the W = U P MFMAs, sit on four different SIMDs instead of queueing on one
        g.wr = g.w % (NW / WC);
        g.wc = g.w / (NW / WC);
    } else {
        g.wr = g.w / WC;
        g.wc = g.w % WC;
    }
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
    const double* q = p.q + (long long)m * p.q_mstride + (long long)(p.q_cols > 1 ? k : 0) * Nxy;

    PROF_DECL;
    assemble_transmissibilities<TS>(p, S, Km, P /* scratch for L */, TX, TY, tid, NT);
    __syncthreads();  // TX/TY entries written by other threads are read below
    PROF(5);

    d4 acc[TRW][TCW];
    int bad = 0, cur = 0;
    // The per-column vectors of block i+1 (transmissibilities, source) are fetched at the END of block i, before the
    // 128 KB store of G_i is issued: vector-memory operations retire in order, so a load issued after those stores
    // would stall the next block on the whole store drain (~10k cycles at one CU's share of HBM bandwidth).
    double pf_y1 = 0.0, pf_y2 = 0.0, pf_x1 = 0.0, pf_x2 = 0.0, pf_q = 0.0, q_cur = 0.0;
    if (tid < NB) {
        pf_y1 = TY[tid]; pf_y2 = TY[tid + 1]; pf_x1 = TX[tid]; pf_x2 = TX[NB + tid]; pf_q = q[tid];
    }
    for (int i = 0; i < Nx; ++i) {
        if (tid < NB) {
            double dg = pf_y1 + pf_y2 + pf_x1 + pf_x2;
            if (i == 0 && tid == 0) dg += Km[0] + Km[0];  // SPD pin: A[0,0] += Kx[0,0]+Ky[0,0]
            L.dgv[tid] = dg;
            L.tyv[tid] = pf_y1;
            if (tid == NB - 1) L.tyv[NB] = pf_y2;
            L.ev[tid] = pf_x1;
            q_cur = pf_q;
        }
        __syncthreads();
        PROF(10);
        if (i > 0) {
            const double t = matvec_tiles<NW>(acc, L.yprev, L, g, tid);
            PROF(11);
            if (tid < NB) L.ycur[tid] = q_cur + L.ev[tid] * t;
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
            if (tid < NB) L.ycur[tid] = q_cur;
#pragma unroll
            for (int ti = 0; ti < TRW; ++ti)
#pragma unroll
                for (int tj = 0; tj < TCW; ++tj) acc[ti][tj] = d4{0.0, 0.0, 0.0, 0.0};
        }
        PROF(12);
        // add the tridiagonal D_i: only diagonal tiles and the corner entries of the two adjacent tile diagonals
        // are touched; the tile tests are wave-uniform
#pragma unroll
        for (int ti = 0; ti < TRW; ++ti)
#pragma unroll
            for (int tj = 0; tj < TCW; ++tj) {
                const int R = TRW * g.wr + ti, C = TCW * g.wc + tj;
                if (R == C) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int lrow = g.lq + 4 * r, row = 16 * R + lrow, col = 16 * C + g.lc;
                        double add = 0.0;
                        if (g.lc == lrow) add = L.dgv[row];
                        else if (g.lc == lrow + 1) add = -L.tyv[col];
                        else if (lrow == g.lc + 1) add = -L.tyv[row];
                        acc[ti][tj][r] += add;
                    }
                } else if (C == R + 1) {  // entry (16R+15, 16C): col == row + 1
                    if (g.lane == 48) acc[ti][tj][3] -= L.tyv[16 * C];
                } else if (R == C + 1) {  // entry (16R, 16C+15): row == col + 1
                    if (g.lane == 15) acc[ti][tj][0] -= L.tyv[16 * R];
                }
            }
        __syncthreads();
        PROF(6);
        // 32 block-sweep panels: A <- -inv(A)
        if constexpr (R16) {
            for (int cpo = 0; cpo < 8 / TRW; ++cpo) {
                panel16<NW, 0>(acc, L, cur, cpo, g, bad PROF_PASS);
                panel16<NW, 1>(acc, L, cur, cpo, g, bad PROF_PASS);
                if constexpr (TRW == 4) {
                    panel16<NW, 2>(acc, L, cur, cpo, g, bad PROF_PASS);
                    panel16<NW, 3>(acc, L, cur, cpo, g, bad PROF_PASS);
                }
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
        if (i + 1 < Nx && tid < NB) {  // next block's vectors, ahead of the G_i stores (see above)
            const int in = i + 1;
            pf_y1 = TY[in * (NB + 1) + tid]; pf_y2 = TY[in * (NB + 1) + tid + 1];
            pf_x1 = TX[in * NB + tid]; pf_x2 = TX[(in + 1) * NB + tid];
            pf_q = q[in * NB + tid];
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
        PROF(7);
    }
    // back substitution: x_i = G_i (y_i + TX[i+1] * x_{i+1});  ycur holds x_{i+1}.  G_{Nx-1} is still in the
    // accumulators; G_{i-1} (128 KB, the HBM stream of this phase) is fetched into a second register set while the
    // mat-vec of block i runs.
    constexpr bool PF = NW == 16;  // with 8 waves the second register set does not fit; the co-resident workgroup
                                   // hides the load latency instead
    d4 nxt[PF ? TRW : 1][PF ? TCW : 1];
    double nyv = 0.0, ntx = 0.0;
    if (PF && tid < NB) nyv = yv[(Nx - 1) * NB + tid];
    for (int i = Nx - 1; i >= 0; --i) {
        if constexpr (!PF) {
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
                nyv = yv[i * NB + tid];
                if (i < Nx - 1) ntx = TX[(i + 1) * NB + tid];
            }
        }
        if (tid < NB) {
            double v = nyv;
            if (i < Nx - 1) v += ntx * L.ycur[tid];
            L.yprev[tid] = v;
        }
        if constexpr (PF) {
            if (i > 0) {
                const double2* Gi = G + (long long)(i - 1) * (NB * NB / 2);
#pragma unroll
                for (int ti = 0; ti < TRW; ++ti)
#pragma unroll
                    for (int tj = 0; tj < TCW; ++tj)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            double2 v = Gi[(((ti * TCW + tj) * 2) + h) * NT + tid];
                            nxt[ti][tj][2 * h] = v.x;
                            nxt[ti][tj][2 * h + 1] = v.y;
                        }
                if (tid < NB) {
                    nyv = yv[(i - 1) * NB + tid];
                    ntx = TX[i * NB + tid];
                }
            }
        }
        __syncthreads();
        const double t = matvec_tiles<NW>(acc, L.yprev, L, g, tid);
        if (tid < NB) {
            L.ycur[tid] = t;
            P[i * NB + tid] = t;
        }
        if constexpr (PF) {
            if (i > 0) {
#pragma unroll
                for (int ti = 0; ti < TRW; ++ti)
#pragma unroll
                    for (int tj = 0; tj < TCW; ++tj) acc[ti][tj] = nxt[ti][tj];
            }
        }
        __syncthreads();
    }
    PROF(8);
    face_fluxes(p, P, TX, TY, Vx, Vy, tid, NT);
    PROF(9);
#ifdef HM_PRESS_PROF
    if (m == 0 && tid == 0)
        for (int c = 0; c < 16; ++c) hm_press_prof_buf[c] = prof_acc[c];
#endif
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
// press_variant 0 (and any other value): 8 waves x 2 workgroups per CU, rank-16 panels;  5: 16 waves, rank-16 panels;
// 4: 16 waves, rank-4 panels;  3: 8 waves, rank-4.
int launch_pressure_128m(hm_fwd* f, const void* S, long long S_stride, int k) {
    const FwdParams& p = f->p;
    if (p.Ny != NB) return -1;
    hipStream_t s = f->ctx->stream;
    const int v = f->press_variant;
#define LAUNCH(TS, NW, R16) hipLaunchKernelGGL((k_press128m<TS, NW, R16>), dim3(p.N), dim3(64 * NW), 0, s, p, (const TS*)S, S_stride, k)
    if (f->dtype == 64) {
        if (v == 3) LAUNCH(double, 8, false);
        else if (v == 4) LAUNCH(double, 16, false);
        else if (v == 5) LAUNCH(double, 16, true);
        else LAUNCH(double, 8, true);  // v == 8
    } else {
        if (v == 3) LAUNCH(float, 8, false);
        else if (v == 4) LAUNCH(float, 16, false);
        else if (v == 5) LAUNCH(float, 16, true);
        else LAUNCH(float, 8, true);
    }
#undef LAUNCH
    HM_HIP(hipGetLastError());
    return 0;
}

#ifdef HM_PRESS_PROF
extern "C" int hm_debug_press_prof(long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(hm_press_prof_buf), sizeof(long long) * 16);
}
#endif

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
