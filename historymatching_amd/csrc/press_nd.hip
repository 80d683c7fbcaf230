// press_nd.hip -- fp64 pressure step of the 128 x 128, 256 x 256 and 512 x 512 grids by NESTED DISSECTION (multifrontal), SURVEY.md A.3's system.
//
// Replaces the sparse direct solve inside ResSim.sim (notebooks/HistoryMatch.py:362; TPFA-ResSim): the block
// elimination along ix of press128s.hip costs 302 Mflop per member and time step at 128 x 128 and has a serial chain of 1024 pivot tiles;
// the geometric dissection of the grid (nd.h) needs about 55 Mflop and a chain of 24 (0.45 Gflop at 256 x 256, 3.6 at 512 x 512, where the
// alternative was 26 / 48 iterations of a two-level CG).  Numerically it is a block L D L^T factorisation in another elimination order:
// same pin, same pivot-tile inverses (sweep16.h), same fp64 matrix-core products.  One source, one object per grid size (ND_LG below).
//
// Data flow per member and time step (level numbers of the 128 x 128 tree; a larger grid's levels are LO = 2 / 4 higher):
//   k_nd_assemble / k_ndl_assemble   TX, TY (bit-exact with the oracle, fwd_dev.h), the coefficient block [dg | -TX | -TY | q]; the plan of the
//                   time step (which fronts keep the results they have: k_nd_plan / k_ndl_plan)
//   k_nd_leaf       level 10: one LANE per leaf, banded L D L^T in registers
//   k_nd_sub        levels 9, 8: ONE WAVE per level-8 subtree.  A front with one pivot tile lives entirely in the wave's registers: with
//                   the pivot panel held TRANSPOSED (V_R = F21_R^T, 16 pivots x 16 front rows, accumulator layout) every product the
//                   factorisation needs has the form Y^T Z of two register tiles, which v_mfma_f64_16x16x4_f64 computes straight from the
//                   accumulator layout (register kk of Y as A operand = Y^T's k-slice, register kk of Z as B operand):
//                       W_R^T = P V_R  (P = inverse pivot tile, symmetric),   F22[R, C] -= (W_R^T)^T V_C.
//                   No operand staging through LDS at all.  Children's update matrices (packed lower triangles) are GATHERED into the
//                   parent's tiles through host-built recipes.
//   k_nd_wave       levels 7, 6, 5: one wave per front, children staged from the member's arena into LDS
//   k_nd_top        128 x 128: levels 4..0, one WORKGROUP per member; fronts with 2..8 pivot tiles, their tiles dealt to the waves' registers,
//                   the current panel (W^T and V per row tile, register images) broadcast through LDS.  Larger grids: levels LO + 4 and
//                   LO + 3, one front per workgroup
//   k_big_*         larger grids, levels LO + 2 .. 0: fronts of up to 25 / 49 tile rows, eliminated left-looking out of global memory
//   k_nd_solve*     back substitution root -> leaves, x1 = -W^T [x2; -1] per panel, then the face fluxes (k_nd_flux; on the larger grids
//                   with the a-posteriori check of the solve)
// The right-hand side rides along as one extra boundary row of every front (its W^T column is z1 = P r1; the (rhs, boundary)
// entries of the update are the reduced right-hand side), so forward elimination costs nothing extra.
#include "fwd_dev.h"
#include "nd_build.h"
#include "nd_plan.h"
#include "sweep16.h"

#ifdef HM_ND_PROF
// cycle stamps: block 0, wave 0, lane 0.  [0..15] k_nd_top phases, [16..31] k_nd_sub (wave-front phases by level), [32..47] counts
__device__ long long hm_nd_prof_buf[64];
#ifndef HM_ND_PROF_TOP_WAVE
#define HM_ND_PROF_TOP_WAVE 0  // the wave of k_nd_top's block 0 whose stamps are kept
#endif
#ifndef HM_ND_PROF_SUB_BLOCK
#define HM_ND_PROF_SUB_BLOCK 0
#endif
#define NPROF_DECL long long prof_t = clock64(), prof_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define NPROF(i) do { const long long now_ = clock64(); prof_acc[i] += now_ - prof_t; prof_t = now_; } while (0)
#else
#define NPROF_DECL
#define NPROF(i)
#endif

// One source, three objects: -DND_LG=7 (128 x 128, the default object press_nd.o), 8 (256 x 256, press_nd256.o), 9 (512 x 512,
// press_nd512.o).  Every grid constant below follows from it, so the 128 x 128 object compiles to what it was.
#ifndef ND_LG
#define ND_LG 7
#endif

namespace {

constexpr int LG = ND_LG;
constexpr int NB = 1 << LG;        // cells a side
constexpr int LO = 2 * (LG - 7);   // levels above the 128 x 128 tree's (nd.h)
constexpr int ND_LEVELS = 11 + LO;
// first front id of the level that is level `l` of the 128 x 128 tree
__host__ __device__ constexpr int FID(int l) { return (1 << (l + LO)) - 1; }
constexpr int NF8 = 256 << LO, NF7 = 128 << LO, NF6 = 64 << LO, NF5 = 32 << LO;  // fronts of levels 8, 7, 6, 5 (128 x 128 numbering)

constexpr int ND_W8 = 4, ND_W7 = ND_W8 + NF8, ND_W6 = ND_W7 + NF7, ND_W5 = ND_W6 + NF6, ND_WT = ND_W5 + NF5;  // ND_WT: n, then the fronts of levels 4..0 in order
constexpr int ND_WORK_INTS = ND_WT + 64;
constexpr int NCACHE = 512 << LO;              // >= fronts of levels 0 .. LO + 8
constexpr int NTODO = 32 << LO;                // >= fronts of levels 0 .. LO + 4
constexpr int WETW = NB / 64;                  // 64-bit words of a grid row's wet-cell bitmap
__device__ __forceinline__ const int* nd_work(const NdDev& nd, int m) { return nd.work + (long long)m * ND_WORK_INTS; }
// Flat tables (nd_plan.h: NdDev::leaft, ssub; built by nd_build_flat_tables below).  A leaf: [0] x0, [1] y0, [2] w, [3] h of its box, [4] b,
// [5] offset of its update matrix in the arena, then for boundary entry j < 12: [6 + 3 j] the local cell 4 lx + ly it touches, [7 + 3 j] the
// index of the face between them in the coefficient block (-1: no such entry), [8 + 3 j] the boundary cell (-1).  A level-8 subtree (fronts
// i = 0: the level-8 front, 1, 2: its level-9 children): header [i] boundary tiles, [3 + i] pivot register rows, [6 + i] factor offset,
// [9] x0, [10] y0 of its region, [11] the LDS plane's row length; per lane [2 i + R] the cell of boundary position 16 R + lc (tile R < 2; a
// tile beyond bt repeats tile 0), [6], [7] the level-8 front's pivot cells 4 r + lq (r < 2), [8], [9] those of the level-9 fronts (r = 0).
constexpr int NLEAF = 1024 << LO;
constexpr int ND_LEAF_INTS = 44;
constexpr int ND_SSUB_HDR = 16, ND_SSUB_LANE = 10, ND_SSUB_INTS = ND_SSUB_HDR + ND_SSUB_LANE * 64;

struct NdGeo {
    int lane, lc, lq;
};

__device__ __forceinline__ int tri(int a, int b) {
    const int hi = a > b ? a : b, lo = a > b ? b : a;
    return ((hi * (hi + 1)) >> 1) + lo;
}

// The member's coefficient block cf = [dg | -TX | -TY | q] (k_nd_assemble): A[cm, ck] of the five-point system (cell = ix * 128
// + iy) is dg[ck], -TX[face between] or -TY[face between]; the right-hand-side row (cm = -2) reads q[ck].
constexpr int CF_OX = NB * NB, CF_OY = CF_OX + (NB + 1) * NB, CF_OQ = CF_OY + NB * (NB + 1), CF_STRIDE = CF_OQ + NB * NB;

// (index arithmetic only and ONE load used unconditionally: behind `if`s the compiler sinks each load into its branch and waits for
// it there -- a round trip to memory per entry instead of one for all the entries a lane assembles)
__device__ __forceinline__ double nd_coef_global(const double* __restrict__ cf, int cm, int ck, bool same_pos) {
    const int d = cm - ck;
    const bool cell = ck >= 0 && cm >= 0;
    int off = -1;
    off = (cell && d == 0) ? ck : off;
    off = (cell && d == NB) ? CF_OX + ck + NB : off;
    off = (cell && d == -NB) ? CF_OX + ck : off;
    off = (cell && d == 1) ? CF_OY + ck + (ck >> LG) + 1 : off;
    off = (cell && d == -1) ? CF_OY + ck + (ck >> LG) : off;
    off = (ck >= 0 && cm == -2) ? CF_OQ + ck : off;
    const double l = cf[off >= 0 ? off : 0];
    return l * (off >= 0 ? 1.0 : 0.0) + ((ck < 0 && same_pos) ? 1.0 : 0.0);  // padded pivot: identity
}

// nd_coef_global in two halves: the load alone (flags: bit 0 = the entry is a coefficient, bit 1 = identity entry of a padded pivot), and
// the arithmetic on the loaded value later -- k_nd_top issues the loads of all its panel tiles, gathers its trailing tiles out of LDS and
// only then touches the values (one trip to memory for all of them, and the stores of the front before have drained meanwhile)
__device__ __forceinline__ double nd_coef_issue(const double* __restrict__ cf, int cm, int ck, bool same_pos, int& flags) {
    const int d = cm - ck;
    const bool cell = ck >= 0 && cm >= 0;
    int off = -1;
    off = (cell && d == 0) ? ck : off;
    off = (cell && d == NB) ? CF_OX + ck + NB : off;
    off = (cell && d == -NB) ? CF_OX + ck : off;
    off = (cell && d == 1) ? CF_OY + ck + (ck >> LG) + 1 : off;
    off = (cell && d == -1) ? CF_OY + ck + (ck >> LG) : off;
    off = (ck >= 0 && cm == -2) ? CF_OQ + ck : off;
    flags = (off >= 0 ? 1 : 0) | ((ck < 0 && same_pos) ? 2 : 0);
    return cf[off >= 0 ? off : 0];
}
__device__ __forceinline__ double nd_coef_finish(double l, int flags) { return l * ((flags & 1) ? 1.0 : 0.0) + ((flags & 2) ? 1.0 : 0.0); }

// The same from an LDS copy of the coefficients around a box of cells [x0, x1) x [y0, y1): four planes (dg, -TX of the cell's
// west face, -TY of its south face, q) over the box plus a ring of one cell, local index (ix - x0 + 1) * ld + (iy - y0 + 1),
// ld = y1 - y0 + 2.  Every pivot of the fronts that use the copy lies inside the box.
struct NdCfl {
    const double* p;  // LDS
    int plane, ld, x0, y0;
};
// a front record's pivot box (slot NDF_PBOX) or region (NDF_RBOX): x range in the slot, y range five slots on (nd.h)
struct NdBox {
    int x0, x1, y0, y1;
};
__device__ __host__ __forceinline__ NdBox nd_box(const int* F, int slot) {
    const int bx = F[slot], by = F[slot + (NDF_PBOY - NDF_PBOX)];
    return NdBox{bx & 0xffff, bx >> 16, by & 0xffff, by >> 16};
}
static_assert(NDF_RBOY - NDF_RBOX == NDF_PBOY - NDF_PBOX, "box slots");
__device__ __forceinline__ void nd_stage_cf(const double* __restrict__ cf, double* cfl, int plane, const NdBox& box, int lane, NdCfl& out) {
    const int x0 = box.x0, y0 = box.y0, x1 = box.x1, y1 = box.y1;
    const int ld = y1 - y0 + 2, n = (x1 - x0 + 2) * ld;
    for (int i = lane; i < n; i += 64) {
        const int lx = i / ld, ly = i - lx * ld;
        const int ix = x0 - 1 + lx, iy = y0 - 1 + ly;
        const bool xin = ix >= 0 && ix < NB, yin = iy >= 0 && iy < NB;
        const bool vx = yin && ix >= 0 && ix <= NB, vy = xin && iy >= 0 && iy <= NB;
        const int c = ix * NB + iy;
        // four unconditional loads (a harmless address where the cell lies outside the grid), then the masks: under selects the
        // compiler branches around each load and waits for it there, four round trips to memory instead of one
        const double l0 = cf[(xin && yin) ? c : 0], l1 = cf[CF_OX + (vx ? c : 0)], l2 = cf[CF_OY + (vy ? ix * (NB + 1) + iy : 0)],
                     l3 = cf[CF_OQ + ((xin && yin) ? c : 0)];
        cfl[i] = l0 * ((xin && yin) ? 1.0 : 0.0);
        cfl[plane + i] = l1 * (vx ? 1.0 : 0.0);
        cfl[2 * plane + i] = l2 * (vy ? 1.0 : 0.0);
        cfl[3 * plane + i] = l3 * ((xin && yin) ? 1.0 : 0.0);
    }
    out.p = cfl; out.plane = plane; out.ld = ld; out.x0 = x0; out.y0 = y0;
}
// (index arithmetic only and the loaded value used unconditionally: with `if`s the lanes of a wave take up to six divergent paths,
// each with its own LDS round trip; with a select around the load the compiler sinks the load into a branch and waits there)
__device__ __forceinline__ double nd_coef_lds(const NdCfl& L, int cm, int ck, bool same_pos) {
    const int li = ((ck >> LG) - L.x0 + 1) * L.ld + ((ck & (NB - 1)) - L.y0 + 1);
    const int d = cm - ck;
    const bool cell = ck >= 0 && cm >= 0;
    int off = -1;
    off = (cell && d == 0) ? li : off;
    off = (cell && d == NB) ? L.plane + li + L.ld : off;
    off = (cell && d == -NB) ? L.plane + li : off;
    off = (cell && d == 1) ? 2 * L.plane + li + 1 : off;
    off = (cell && d == -1) ? 2 * L.plane + li : off;
    off = (ck >= 0 && cm == -2) ? 3 * L.plane + li : off;
    const double l = L.p[off >= 0 ? off : 0];
    return l * (off >= 0 ? 1.0 : 0.0) + ((ck < 0 && same_pos) ? 1.0 : 0.0);
}
// children's updates in LDS (wave-level fronts): unconditional load, see above
__device__ __forceinline__ double nd_gather(const double* ch, int a, int c) {
    const bool ok = a >= 0 && c >= 0;
    const double l = ch[ok ? tri(a, c) : 0];
    return l * (ok ? 1.0 : 0.0);
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

// n doubles (n even, both 16-byte aligned) from global memory to LDS by one wave through LDS-DMA (global_load_lds_dwordx4: no registers,
// every piece of 64 double2 in flight at once, retired by the wave's vmcnt -- the caller waits for vmcnt(0) before it reads); the last
// n2 % 64 double2 by one ordinary load and store (a clamped last piece would write past the destination).
typedef __attribute__((address_space(3))) void* nd_lds_ptr;
typedef const __attribute__((address_space(1))) void* nd_glb_ptr;
__device__ __forceinline__ void nd_wave_copy(double* dst, const double* __restrict__ src, int n, int lane) {
    const double2* s2 = reinterpret_cast<const double2*>(src);
    double2* d2 = reinterpret_cast<double2*>(dst);
    const int n2 = n >> 1, full = n2 >> 6;
    for (int pc = 0; pc < full; ++pc)
        __builtin_amdgcn_global_load_lds((nd_glb_ptr)(s2 + pc * 64 + lane), (nd_lds_ptr)(d2 + pc * 64), 16, 0, 0);
    const int i = full * 64 + lane;
    if (i < n2) d2[i] = s2[i];
}

// ------------------------------------------------------------------------------------------------------------------------
// One front with ONE pivot tile, processed by one wave in registers.  What each lane reads, adds and writes is spelled out in
// the front's assembly RECIPES (nd.h: offsets precomputed on the host): no index arithmetic here.  cfl: the wave's LDS copy of
// the coefficient planes; ch0 / ch1: the children's packed update matrices in LDS (KIDS); out: this front's packed update
// matrix (LDS or global); fa: the front's factor (global).
// ------------------------------------------------------------------------------------------------------------------------
typedef short s4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ s4 rec_load(const short* __restrict__ rec, int blk, int lane) {
    return *reinterpret_cast<const s4*>(rec + ((long long)blk * 64 + lane) * 4);
}
// value at a recipe's byte offset into the wave's LDS block ("none" points at a cell holding 0.0, "identity" at one holding 1.0:
// no clamp, no mask, no select -- one ds_read per entry)
__device__ __forceinline__ double rec_val(const double* blk, int byte_off) {
    return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(blk) + byte_off);
}

// The panel's recipes, requested ahead of the front that uses them (a front's first instructions would otherwise wait a round trip
// to L2 for them): while front i computes, front i + 1's are in flight.
template <int MAXBT, bool KIDS>
struct NdPanelRec {
    s4 rc[MAXBT + 1], r0[MAXBT + 1], r1[MAXBT + 1];
};
// kidm / cofm: the front's NDF_KIDM / NDF_COFM (nd.h): tile rows that hold anything of child 0 / child 1 / a coefficient against a pivot.
// Recipes of what a tile does not have are neither loaded nor followed (they would point at the zero cell: one ds_read and one add each).
template <int MAXBT, bool KIDS>
__device__ __forceinline__ void nd_panel_rec_load(NdPanelRec<MAXBT, KIDS>& pr, const short* __restrict__ rec, int bt, int lane, int kidm, int cofm) {
    constexpr int NK = KIDS ? 3 : 1;
#pragma unroll
    for (int R = 0; R <= MAXBT; ++R) {
        const int Rc = R <= bt ? R : bt;
        if ((cofm >> Rc) & 1) pr.rc[R] = rec_load(rec, NK * Rc, lane);
        if (KIDS) {
            if ((kidm >> Rc) & 1) pr.r0[R] = rec_load(rec, NK * Rc + 1, lane);
            if ((kidm >> (16 + Rc)) & 1) pr.r1[R] = rec_load(rec, NK * Rc + 2, lane);
        }
    }
}

template <int MAXBT, bool KIDS>
__device__ __forceinline__ void nd_wave_front(int bt, int kreg, const NdPanelRec<MAXBT, KIDS>& pr, const short* __restrict__ rec, const double* blk,
                                              double* out, double* __restrict__ fa, const NdGeo& g, int& bad, int kidm, int cofm) {
    constexpr int NK = KIDS ? 3 : 1;
    // a trailing tile (R, C) takes child c's update where both its rows and its columns hold something of it
    const int both0 = kidm & 0xffff, both1 = (kidm >> 16) & 0xffff;
    // ---- recipes of the trailing tiles, one tile row at a time: row 1 requested now (used after the sweep), row R + 1 while row R
    // is computed (all rows at once would be 126 registers for a level-5 front)
    const short* rect = rec + (long long)NK * (bt + 1) * 256;
    s4 t0[2][MAXBT], t1[2][MAXBT], to[2][MAXBT];
    auto load_row = [&](int R, int buf) {  // R static at every call site
#pragma unroll
        for (int C = 1; C <= MAXBT; ++C) {
            if (C > R) break;
            const int t = R * (R - 1) / 2 + C - 1;
            const int tc = R <= bt ? t : 0;
            if (KIDS) {
                if ((both0 >> R) & (both0 >> C) & 1) t0[buf][C - 1] = rec_load(rect, NK * tc, g.lane);
                if ((both1 >> R) & (both1 >> C) & 1) t1[buf][C - 1] = rec_load(rect, NK * tc + 1, g.lane);
            }
            to[buf][C - 1] = rec_load(rect, NK * tc + NK - 1, g.lane);
        }
    };
    load_row(1, 1);
    // ---- the pivot panel, transposed: V[R][r] = F[front row 16 R + lc][pivot 4 r + lq]
    d4 V[MAXBT + 1];
#pragma unroll
    for (int R = 0; R <= MAXBT; ++R) {
        if (R > bt) break;  // (tile rows beyond bt are never used below)
        const bool hc = (cofm >> R) & 1, h0 = KIDS && ((both0 >> R) & 1), h1 = KIDS && ((both1 >> R) & 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double v = hc ? rec_val(blk, pr.rc[R][r]) : 0.0;
            if (h0 && h1) v += rec_val(blk, pr.r0[R][r]) + rec_val(blk, pr.r1[R][r]);
            else if (h0) v += rec_val(blk, pr.r0[R][r]);
            else if (h1) v += rec_val(blk, pr.r1[R][r]);
            V[R][r] = v;
        }
    }
    // ---- P = inverse of the pivot tile
    d4 P = V[0];
    sweep16_partial(P, g, bad, kreg);  // P = -inv
    // ---- W_R^T = P V_R (kept negated: the products below subtract), factor rows to memory
    d4 WTn[MAXBT];
#pragma unroll
    for (int R = 1; R <= MAXBT; ++R) {
        d4 w = {0.0, 0.0, 0.0, 0.0};
        if (R <= bt) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                if (kk < kreg) w = __builtin_amdgcn_mfma_f64_16x16x4f64(P[kk], V[R][kk], w, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r < kreg) fa[((R - 1) * kreg + r) * 64 + g.lane] = -w[r];
        }
        WTn[R - 1] = w;
    }
    // ---- trailing tiles: update = children - W V^T, packed lower
#pragma unroll
    for (int R = 1; R <= MAXBT; ++R) {
        if (R > bt) break;
        if (R < MAXBT) load_row(R + 1, (R + 1) & 1);
#pragma unroll
        for (int C = 1; C <= R; ++C) {
            d4 acc = {0.0, 0.0, 0.0, 0.0};
            if (KIDS) {
                const bool a0 = (both0 >> R) & (both0 >> C) & 1, a1 = (both1 >> R) & (both1 >> C) & 1;
                if (a0 && a1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[r] = rec_val(blk, t0[R & 1][C - 1][r]) + rec_val(blk, t1[R & 1][C - 1][r]);
                } else if (a0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[r] = rec_val(blk, t0[R & 1][C - 1][r]);
                } else if (a1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[r] = rec_val(blk, t1[R & 1][C - 1][r]);
                }
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                if (kk < kreg) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(WTn[R - 1][kk], V[C][kk], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = to[R & 1][C - 1][r];
                if (o >= 0) out[o] = acc[r];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// TX, TY (bit-exact: fwd_dev.h) and the coefficient block (one workgroup per member).  Isotropic K (every reference call site): one
// pass -- 1 / (mobility K) of the member goes to LDS (128 KB), every face and the diagonal are formed from there (the diagonal takes
// its four faces again: the same expressions, the same bits), and the plan of the time step (below) is made in the same launch from a
// wet-cell bitmap the first loop collects by ballots: 0.36 + 0.06 -> 0.24 ms for 1000 members.  Anisotropic K: 1 / (mobility K) through
// memory as on the other grids (fwd_dev.h), the plan as a launch of its own.
// ------------------------------------------------------------------------------------------------------------------------
#if ND_LG == 7
__device__ __forceinline__ void nd_plan_body(const NdDev& nd, int m, int t, const unsigned long long (*wet)[2], unsigned char* dry, int* wcount);  // below

template <typename TS>
__global__ __launch_bounds__(1024, 2) void k_nd_assemble(FwdParams p, NdDev nd, const TS* __restrict__ S_base, long long S_stride, int k) {
    const int m = blockIdx.x, tid = threadIdx.x;
    const int Nx = p.Nx, Nxy = p.Nxy;
    const TS* S = S_base + (long long)m * S_stride;
    const double* Km = p.K + (long long)m * Nxy;
    const double* Kym = p.Ky ? p.Ky + (long long)m * Nxy : Km;
    double* TX = p.TX + (long long)m * (Nx + 1) * NB;
    double* TY = p.TY + (long long)m * Nx * (NB + 1);
    double* cf = nd.cf + (long long)m * CF_STRIDE;
    const double* q = p.q + (long long)m * p.q_mstride + (long long)(p.q_cols > 1 ? k : 0) * Nxy;
    if (Kym == Km) {  // isotropic (every reference call site): 1 / (mobility K) of the member in LDS, every face and the diagonal from there
        extern __shared__ double nd_lds[];
        __shared__ unsigned char dry[512];
        __shared__ unsigned long long wet[NB][2];
        __shared__ int wcount[4];
        // (round 6) in TWO passes of 64 grid rows, 1 / (mobility K) of rows ix0 - 1 .. ix0 + 64 in 66 KB of LDS instead of the whole member in
        // 128 KB: two workgroups fit a CU, and one's loads run beside the other's stores (the kernel is nothing but 1 MB of memory traffic
        // a member: 0.29 -> 0.24 ms per 1000 members, profiles/r06/nd_assemble_two_pass.txt).  The same expressions on the same operands.
        double* L = nd_lds;
        constexpr int HALF = NB / 2;
        for (int h = 0; h < 2; ++h) {
            const int ix0 = h * HALF, lo = ix0 > 0 ? ix0 - 1 : 0, hi = ix0 + HALF < Nx ? ix0 + HALF : Nx - 1;  // rows held: lo .. hi
            const double* Lr = L - lo * NB;  // Lr[cell] for the cells of rows lo .. hi
            if (h) __syncthreads();  // (the first pass's reads are done)
            for (int j = lo * NB + tid; j < (hi + 1) * NB; j += 1024) {  // (a wave = 64 consecutive cells of one grid row: its ballot is a word of the wet-cell bitmap)
                const double sj = (double)S[j];
                double mw, mo;
                rel_perm<double>(p, sj, mw, mo);
                L[j - lo * NB] = 1.0 / ((mw + mo) * Km[j]);
                const unsigned long long bits = __ballot(S[j] != (TS)0);
                if ((tid & 63) == 0) wet[j >> 7][(j >> 6) & 1] = bits;
            }
            __syncthreads();
            const int fx1 = h ? (Nx + 1) * NB : (ix0 + HALF) * NB;  // x faces ix0 .. ix0 + 63 (the last pass: the boundary face Nx as well)
            for (int f = ix0 * NB + tid; f < fx1; f += 1024) {
                const int ix = f >> 7;
                const double tx = (ix == 0 || ix == Nx) ? 0.0 : p.cx / (Lr[f - NB] + Lr[f]);
                TX[f] = tx;
                cf[CF_OX + f] = -tx;
            }
            for (int f = ix0 * (NB + 1) + tid; f < (ix0 + HALF) * (NB + 1); f += 1024) {
                const int ix = f / (NB + 1), iy = f - ix * (NB + 1), c = ix * NB + iy;
                const double ty = (iy == 0 || iy == NB) ? 0.0 : p.cy / (Lr[c - 1] + Lr[c]);
                TY[f] = ty;
                cf[CF_OY + f] = -ty;
            }
            for (int c = ix0 * NB + tid; c < (ix0 + HALF) * NB; c += 1024) {  // the diagonal: the four faces again, the same expressions (the same bits)
                const int ix = c >> 7, iy = c & (NB - 1);
                const double ty0 = iy == 0 ? 0.0 : p.cy / (Lr[c - 1] + Lr[c]), ty1 = iy == NB - 1 ? 0.0 : p.cy / (Lr[c] + Lr[c + 1]);
                const double tx0 = ix == 0 ? 0.0 : p.cx / (Lr[c - NB] + Lr[c]), tx1 = ix == Nx - 1 ? 0.0 : p.cx / (Lr[c] + Lr[c + NB]);
                double d = ty0 + ty1 + tx0 + tx1;
                if (c == 0) d += Km[0] + Kym[0];
                if (d == 0.0) d = 1.0;  // a cell of zero permeability (the padding of an embedded grid, forward.hip): every face closed, its equation is 1 p = 0
                cf[c] = d;
                cf[CF_OQ + c] = q[c];
            }
        }
        nd_plan_body(nd, m, tid, wet, dry, wcount);  // what has to be eliminated this time step
        return;
    }
    assemble_transmissibilities<TS>(p, S, Km, Kym, p.P + (long long)m * Nxy, TX, TY, tid, 1024);
    for (int c = tid; c < Nxy; c += 1024) {
        const int ty = c + (c >> 7);
        double d = TY[ty] + TY[ty + 1] + TX[c] + TX[c + NB];
        if (c == 0) d += Km[0] + Kym[0];  // SPD pin: A[0,0] += Kx[0,0] + Ky[0,0]
        cf[c] = d;
        cf[CF_OQ + c] = q[c];
    }
    for (int i = tid; i < (NB + 1) * NB; i += 1024) {
        cf[CF_OX + i] = -TX[i];
        cf[CF_OY + i] = -TY[i];
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// What has to be eliminated this time step.  Ahead of the water front the saturation is exactly zero, the mobilities and with them the
// coefficients of every cell there are what they were the step before, and a front whose whole subtree lies there would reproduce its
// factor rows and its update matrix bit for bit -- both are still in memory (the arena keeps one slot per front of levels 1..8 and per
// leaf; level 9 lives in LDS inside k_nd_sub, whose level-8 subtree is skipped or redone as a whole).  A front of levels 8..0 is SKIPPED
// when (a) every cell of the box its coefficients come from (its subtree's region for level 8, its separator above that, plus a ring of
// one cell) is dry now, (b) its children are skipped, (c) the rates of the wells in its subtree -- they are its right-hand side -- are
// those of the time step its results were stored at (`wells_ok`: constant rates, or the same run of equal columns of a rate schedule;
// the producers' corners are the regions that stay dry longest) and (d) its stored results were computed from such an all-dry state
// since the plan's inputs last changed (`cached`, cleared whenever K, wells or kernel selection change: hm_fwd::inputs_gen).  The
// back substitution always runs in full.
// One workgroup of 256 threads per member; the surviving fronts are written as compacted lists that the elimination kernels index.
// ------------------------------------------------------------------------------------------------------------------------
// The plan from the wet-cell bitmap (one bit per cell: S != 0; -0.0 == 0: dry; 128 bits per grid row); every thread of the
// workgroup calls it (the first 256 do the work, all take the barriers).
__device__ __forceinline__ void nd_plan_body(const NdDev& nd, int m, int t, const unsigned long long (*wet)[2], unsigned char* dry, int* wcount) {
    int* work = nd.work + (long long)m * ND_WORK_INTS;
    unsigned char* cached = nd.cached + (long long)m * 512;
    auto box_dry = [&](const NdBox& box) {
        const int x0 = max(box.x0 - 1, 0), y0 = max(box.y0 - 1, 0), x1 = min(box.x1 + 1, NB), y1 = min(box.y1 + 1, NB);
        // bits [y0, y1) of a 128-bit row
        const unsigned long long lo = y0 < 64 ? (~0ull << y0) & (y1 >= 64 ? ~0ull : ~(~0ull << y1)) : 0ull;
        const unsigned long long hi = y1 > 64 ? (y0 > 64 ? ~0ull << (y0 - 64) : ~0ull) & (y1 == 128 ? ~0ull : ~(~0ull << (y1 - 64))) : 0ull;
        unsigned long long any = 0ull;
        for (int ix = x0; ix < x1; ++ix) any |= (wet[ix][0] & lo) | (wet[ix][1] & hi);
        return any == 0ull;
    };
    // level 8: the subtree's region; levels 7..5: the separator, and both children
    if (t < 256) {
        const int f = 255 + t;
        dry[f] = nd.reuse && box_dry(nd_box(nd.fronts + f * ND_FRONT_INTS, NDF_RBOX));
    }
    __syncthreads();
    for (int lv = 7; lv >= 0; --lv) {
        const int nf = 1 << lv, f = nf - 1 + t;
        if (t < nf) dry[f] = dry[2 * f + 1] && dry[2 * f + 2] && box_dry(nd_box(nd.fronts + f * ND_FRONT_INTS, NDF_PBOX));
        __syncthreads();
    }
    // skip = dry and cached; what is computed now is the state of the cache afterwards
    for (int lv = 8; lv >= 5; --lv) {
        const int nf = 1 << lv, f = nf - 1 + t;
        const int base = lv == 8 ? ND_W8 : lv == 7 ? ND_W7 : lv == 6 ? ND_W6 : ND_W5;
        bool todo = false;
        if (t < nf) {
            todo = !(dry[f] && cached[f] && (nd.wells_ok || !nd.wells[f]));
            cached[f] = dry[f];
        }
        // order-preserving compaction over the (at most four) waves that hold fronts
        const unsigned long long mask = __ballot(todo);
        const int w = t >> 6, lane = t & 63;
        if (lane == 0 && w < 4) wcount[w] = __popcll(mask);
        __syncthreads();
        int off = 0;
        for (int q = 0; q < w && q < 4; ++q) off += wcount[q];
        if (todo) work[base + off + __popcll(mask & ((1ull << lane) - 1ull))] = t;
        if (t == 0) work[8 - lv] = wcount[0] + wcount[1] + wcount[2] + wcount[3];
        __syncthreads();
    }
    // levels 4..0 (k_nd_top): the fronts to eliminate, in that kernel's order
    if (t == 0) {
        int nt = 0;
        for (int lv = 4; lv >= 0; --lv)
            for (int f = (1 << lv) - 1; f < (2 << lv) - 1; ++f) {
                if (!(dry[f] && cached[f] && (nd.wells_ok || !nd.wells[f]))) work[ND_WT + 1 + nt++] = f;
                cached[f] = dry[f];
            }
        work[ND_WT] = nt;
    }
}

// One workgroup of 256 threads per member (the anisotropic plans; the isotropic ones plan inside k_nd_assemble).
template <typename TS>
__global__ __launch_bounds__(256) void k_nd_plan(FwdParams p, NdDev nd, const TS* __restrict__ S_base, long long S_stride) {
    __shared__ unsigned char dry[512];  // by front id (0..510)
    __shared__ unsigned long long wet[NB][2];
    __shared__ int wcount[4];
    const int m = blockIdx.x, t = threadIdx.x;
    const TS* S = S_base + (long long)m * S_stride;
    for (int h = t; h < 2 * NB; h += 256) {  // thread t takes the half rows t and t + 256 (64 cells each)
        const TS* row = S + (h >> 1) * NB + (h & 1) * 64;
        unsigned long long bits = 0ull;
        for (int c = 0; c < 64; ++c) bits |= (unsigned long long)(row[c] != (TS)0) << c;
        wet[h >> 1][h & 1] = bits;
    }
    __syncthreads();
    nd_plan_body(nd, m, t, wet, dry, wcount);
}

#else  // ND_LG > 7
// TX, TY (bit-exact: the expressions of fwd_dev.h / the kernel above, operand for operand) and the coefficient block of the larger
// grids: one thread per cell, one workgroup per grid row, 1 / (mobility K) of the cell and its four neighbours recomputed on the fly
// (five divisions per cell instead of a 0.5 / 2 MB image per member that no LDS holds).  A cell writes its west and south faces; the
// last row / column also the zero boundary faces behind them.
template <typename TS>
__global__ __launch_bounds__(NB) void k_ndl_assemble(FwdParams p, NdDev nd, const TS* __restrict__ S_base, long long S_stride, int k) {
    const int m = blockIdx.x % p.N, ix = blockIdx.x / p.N, iy = threadIdx.x;
    const int Nx = p.Nx, Nxy = p.Nxy;
    const TS* S = S_base + (long long)m * S_stride;
    const double* Km = p.K + (long long)m * Nxy;
    const double* Kym = p.Ky ? p.Ky + (long long)m * Nxy : Km;
    double* TX = p.TX + (long long)m * (Nx + 1) * NB;
    double* TY = p.TY + (long long)m * Nx * (NB + 1);
    double* cf = nd.cf + (long long)m * CF_STRIDE;
    const double* q = p.q + (long long)m * p.q_mstride + (long long)(p.q_cols > 1 ? k : 0) * Nxy;
    const int c = ix * NB + iy;
    {   // the row's wet-cell bitmap (one bit per cell: S != 0; -0.0 == 0: dry), a word per wave, for the plan of the time step
        const unsigned long long bits = __ballot(S[c] != (TS)0);
        if ((iy & 63) == 0) nd.wet[((long long)m * NB + ix) * WETW + (iy >> 6)] = bits;
    }
    auto linv = [&](int cell, const double* Kf) {  // 1 / (mobility K): the expression of assemble_transmissibilities (fwd_dev.h)
        double mw, mo;
        rel_perm<double>(p, (double)S[cell], mw, mo);
        return 1.0 / ((mw + mo) * Kf[cell]);
    };
    const double lxc = linv(c, Km), lyc = Kym == Km ? lxc : linv(c, Kym);
    const double lw = ix > 0 ? linv(c - NB, Km) : 0.0, le = ix < Nx - 1 ? linv(c + NB, Km) : 0.0;
    const double ls = iy > 0 ? linv(c - 1, Kym) : 0.0, ln = iy < NB - 1 ? linv(c + 1, Kym) : 0.0;
    const double tx0 = ix == 0 ? 0.0 : p.cx / (lw + lxc), tx1 = ix == Nx - 1 ? 0.0 : p.cx / (lxc + le);
    const double ty0 = iy == 0 ? 0.0 : p.cy / (ls + lyc), ty1 = iy == NB - 1 ? 0.0 : p.cy / (lyc + ln);
    TX[c] = tx0;
    cf[CF_OX + c] = -tx0;
    if (ix == Nx - 1) { TX[c + NB] = 0.0; cf[CF_OX + c + NB] = -0.0; }
    const int fy = ix * (NB + 1) + iy;
    TY[fy] = ty0;
    cf[CF_OY + fy] = -ty0;
    if (iy == NB - 1) { TY[fy + 1] = 0.0; cf[CF_OY + fy + 1] = -0.0; }
    double d = ty0 + ty1 + tx0 + tx1;
    if (c == 0) d += Km[0] + Kym[0];  // SPD pin: A[0,0] += Kx[0,0] + Ky[0,0]
    if (d == 0.0) d = 1.0;  // a cell of zero permeability (the padding of an embedded grid, forward.hip): its equation is 1 p = 0
    cf[c] = d;
    cf[CF_OQ + c] = q[c];
}

// What has to be eliminated this time step on the larger grids: the plan of the 128 x 128 tree (nd_plan_body above: a front is skipped while
// its box + ring is dry, its children are skipped, the rates of the wells in its subtree are unchanged and its stored results were
// computed from such a state since the inputs last changed) for 4 / 16 times the fronts.  One workgroup of 1024 threads per member; the
// lists of levels 8..5 (128 x 128 numbering) as there, one `todo` byte per front of levels 0 .. LO + 4 (k_nd_top's and the big fronts'
// launches are indexed by front: a skipped front's workgroups return at once).  Only where the whole ensemble is one member block (the
// factor and update matrices of every member stay in memory); else, and for press_variant 14, nd.reuse = 0 makes every front due.
__global__ __launch_bounds__(1024) void k_ndl_plan(FwdParams p, NdDev nd) {
    extern __shared__ unsigned long long plan_lds[];
    unsigned long long* wet = plan_lds;                                             // [NB][WETW]
    unsigned char* dry = reinterpret_cast<unsigned char*>(plan_lds + NB * WETW);     // by front id, levels 0 .. LO + 8
    __shared__ int wcount[16];
    const int m = blockIdx.x, t = threadIdx.x;
    int* work = nd.work + (long long)m * ND_WORK_INTS;
    unsigned char* cached = nd.cached + (long long)m * NCACHE;
    for (int i = t; i < NB * WETW; i += 1024) wet[i] = nd.wet[(long long)m * NB * WETW + i];
    __syncthreads();
    auto box_dry = [&](const NdBox& box) {
        const int x0 = max(box.x0 - 1, 0), y0 = max(box.y0 - 1, 0), x1 = min(box.x1 + 1, NB), y1 = min(box.y1 + 1, NB);
        unsigned long long any = 0ull;
        for (int wd = y0 >> 6; wd <= (y1 - 1) >> 6; ++wd) {
            const int lo = max(y0 - 64 * wd, 0), hi = min(y1 - 64 * wd, 64);  // bits [lo, hi) of word wd
            const unsigned long long msk = (hi == 64 ? ~0ull : ~(~0ull << hi)) & (~0ull << lo);
            for (int ix = x0; ix < x1; ++ix) any |= wet[ix * WETW + wd] & msk;
        }
        return any == 0ull;
    };
    for (int i = t; i < NF8; i += 1024) {
        const int f = FID(8) + i;
        dry[f] = nd.reuse && box_dry(nd_box(nd.fronts + f * ND_FRONT_INTS, NDF_RBOX));
    }
    __syncthreads();
    for (int lv = LO + 7; lv >= 0; --lv) {
        const int nf = 1 << lv;
        for (int i = t; i < nf; i += 1024) {
            const int f = nf - 1 + i;
            dry[f] = dry[2 * f + 1] && dry[2 * f + 2] && box_dry(nd_box(nd.fronts + f * ND_FRONT_INTS, NDF_PBOX));
        }
        __syncthreads();
    }
    auto due = [&](int f) {  // skip = dry and cached; what is computed now is the state of the cache afterwards
        const bool todo = !(dry[f] && cached[f] && (nd.wells_ok || !nd.wells[f]));
        cached[f] = dry[f];
        return todo;
    };
    for (int lvp = 8; lvp >= 5; --lvp) {  // order-preserving compaction, 1024 fronts at a time
        const int nf = 1 << (lvp + LO);
        const int base = lvp == 8 ? ND_W8 : lvp == 7 ? ND_W7 : lvp == 6 ? ND_W6 : ND_W5;
        int running = 0;
        for (int c0 = 0; c0 < nf; c0 += 1024) {
            const int i = c0 + t;
            const bool todo = i < nf && due(nf - 1 + i);
            const unsigned long long mask = __ballot(todo);
            const int w = t >> 6, lane = t & 63;
            if (lane == 0) wcount[w] = __popcll(mask);
            __syncthreads();
            int off = running, tot = 0;
            for (int qq = 0; qq < 16; ++qq) {
                off += qq < w ? wcount[qq] : 0;
                tot += wcount[qq];
            }
            if (todo) work[base + off + __popcll(mask & ((1ull << lane) - 1ull))] = i;
            running += tot;
            __syncthreads();
        }
        if (t == 0) work[8 - lvp] = running;
    }
    unsigned char* todo_b = nd.todo + (long long)m * NTODO;
    for (int f = t; f < (2 << (LO + 4)) - 1; f += 1024) todo_b[f] = due(f);
}
#endif  // ND_LG

// ------------------------------------------------------------------------------------------------------------------------
// The leaves (level 10: 3 x 3 .. 4 x 4 cells, at most 12 boundary cells), ONE LANE PER LEAF.  A leaf's pivot block is the five-point
// matrix of a tiny grid -- banded, bandwidth 4 in the padded 4 x 4 ordering i = 4 lx + ly -- and every boundary cell touches exactly one
// of its cells.  As a front on the matrix cores a leaf costs a whole wave ~900 instructions (a 16-pivot in-wave sweep, two padded
// tiles); as a banded L D L^T in one lane's registers the 64 leaves of a wave cost ~3 k instructions together.  Per lane:
//   band of A_II from the coefficient block, L D L^T (16 x 4 band), z = A_II^-1 q_I and the columns g = A_II^-1 e_i of the cells next
//   to the boundary; update matrix U[j][l] = -t_j t_l G[i_l][i_j], right-hand-side row U[rhs][l] = -t_l z[i_l]   (t = A[boundary, cell])
// written packed to the arena (k_nd_sub stages them for the level-9 fronts).  No factor is stored: k_nd_leaf_solve, the last step of the
// back substitution, repeats the band factorisation and solves A_II x_I = q_I - A_Ib x_b with the boundary pressures known.
// ------------------------------------------------------------------------------------------------------------------------
struct NdLeaf {
    double l[16][4];   // L[i][i-1-k], k = 0..3
    double invd[16];
    int cell[16];      // global cell of local cell i = 4 lx + ly, -1: padding
};

// band + factorisation of the leaf whose cells are the box [x0, x0 + w) x [y0, y0 + h) (per lane); returns false for a non-positive pivot
__device__ __forceinline__ bool nd_leaf_factor(NdLeaf& Lf, const double* __restrict__ cf, int x0, int y0, int w, int h) {
    double d[16], e1[16], e4[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int lx = i >> 2, ly = i & 3;
        const bool alive = lx < w && ly < h;
        const int c = alive ? (x0 + lx) * NB + y0 + ly : 0;
        Lf.cell[i] = alive ? c : -1;
        const double dg = cf[c], tn = cf[CF_OY + c + (c >> LG) + 1], te = cf[CF_OX + c + NB];
        d[i] = alive ? dg : 1.0;
        e1[i] = (alive && ly + 1 < h) ? tn : 0.0;   // A[i+1][i]: the north face (cf holds -TY)
        e4[i] = (alive && lx + 1 < w) ? te : 0.0;   // A[i+4][i]: the east face (cf holds -TX)
    }
    bool ok = true;
    double ld[16][4];  // L[i][j] D[j]
#pragma unroll
    for (int i = 0; i < 16; ++i) {
#pragma unroll
        for (int k = 3; k >= 0; --k) {  // j = i - 1 - k ascending
            const int j = i - 1 - k;
            if (j < 0) { Lf.l[i][k] = 0.0; ld[i][k] = 0.0; continue; }
            double sacc = k == 0 ? e1[j] : (k == 3 ? e4[j] : 0.0);
            if (k == 0 && (i & 3) == 0) sacc = 0.0;  // i and i - 1 lie in different columns of the 4 x 4 layout
#pragma unroll
            for (int mm = i - 4; mm < j; ++mm) {
                if (mm < 0) continue;
                sacc -= ld[i][i - 1 - mm] * Lf.l[j][j - 1 - mm];
            }
            ld[i][k] = sacc;
            Lf.l[i][k] = sacc * Lf.invd[j];
        }
        double dd = d[i];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i - 1 - k >= 0) dd -= Lf.l[i][k] * ld[i][k];
        ok = ok && dd > 0.0;
        Lf.invd[i] = 1.0 / dd;
    }
    return ok;
}
// x = A_II^-1 v in place
__device__ __forceinline__ void nd_leaf_solve(const NdLeaf& Lf, double (&v)[16]) {
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i - 1 - k >= 0) v[i] -= Lf.l[i][k] * v[i - 1 - k];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] *= Lf.invd[i];
#pragma unroll
    for (int i = 15; i >= 0; --i)
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i + 1 + k < 16) v[i] -= Lf.l[i + 1 + k][k] * v[i + 1 + k];
}
// boundary entry j of a leaf from the leaf table (one lane per leaf; entry k of leaf l at leaft[k * NLEAF + l]): the local cell it touches
// and the coefficient A[boundary cell, that cell] (0: no such entry)
__device__ __forceinline__ void nd_leaf_boundary(const int* __restrict__ LT, const double* __restrict__ cf, int j, int& il, double& t, int& bcell) {
    il = LT[(6 + 3 * j) * NLEAF];
    const int fidx = LT[(7 + 3 * j) * NLEAF];
    bcell = LT[(8 + 3 * j) * NLEAF];
    const double tv = cf[fidx >= 0 ? fidx : 0];  // (unconditional load, then the mask: see nd_coef_global)
    t = fidx >= 0 ? tv : 0.0;
}

__global__ __launch_bounds__(256, 2) void k_nd_leaf(FwdParams p, NdDev nd, int k) {
    __shared__ double gsh[4][16][64];  // per wave: one solution vector per lane, [cell][lane]
    const int m = blockIdx.x % p.N, bidx = blockIdx.x / p.N;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // the leaves of the level-8 subtrees that are eliminated this step (k_nd_plan's list), four lanes per subtree
    const int* work = nd_work(nd, m);
    const int e = (bidx * 256 + tid) >> 2;
    if (e >= work[0]) return;  // (no workgroup barrier below)
    // (round 6: box, boundary cells and faces of the leaf come from the flat leaf table -- one coalesced read instead of the chain front
    // record -> position table -> cell, two dependent trips to memory less per wave)
    const int i8 = work[ND_W8 + e];
    const int* LT = nd.leaft + 4 * i8 + (tid & 3);
    const double* cf = nd.cf + (long long)m * CF_STRIDE;
    // the subtree's block of interleaved leaf updates (nd_plan.h: leafu): entry e of this leaf at out[4 e]
    double* out = nd.leafu + (long long)m * nd.leafu_stride + (long long)i8 * (4 * nd.slot10) + (tid & 3);
    const int b = LT[4 * NLEAF];
    NdLeaf Lf;
    const bool ok = nd_leaf_factor(Lf, cf, LT[0], LT[NLEAF], LT[2 * NLEAF], LT[3 * NLEAF]);
    int il[12];
    double t[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        int bc;
        nd_leaf_boundary(LT, cf, j, il[j], t[j], bc);
    }
    double (*gl)[64] = gsh[w];
    // right-hand-side row: z = A_II^-1 q_I
    {
        double v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = Lf.cell[i] >= 0 ? cf[CF_OQ + Lf.cell[i]] : 0.0;
        nd_leaf_solve(Lf, v);
#pragma unroll
        for (int i = 0; i < 16; ++i) gl[i][lane] = v[i];
#pragma unroll
        for (int l = 0; l < 12; ++l)
            if (l < b) out[4 * (((b * (b + 1)) >> 1) + l)] = -t[l] * gl[il[l]][lane];
        out[4 * (((b * (b + 1)) >> 1) + b)] = 0.0;
    }
    // boundary rows: g = A_II^-1 e_{il[j]}
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        double v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = (i == il[j]) ? 1.0 : 0.0;
        nd_leaf_solve(Lf, v);
#pragma unroll
        for (int i = 0; i < 16; ++i) gl[i][lane] = v[i];
#pragma unroll
        for (int l = 0; l <= j; ++l)
            if (j < b) out[4 * (((j * (j + 1)) >> 1) + l)] = -(t[j] * t[l]) * gl[il[l]][lane];
    }
    if (!ok) atomicOr(&p.status[m], HM_MEMBER_BAD_PIVOT);
}

// The last step of the back substitution: the leaves' own cells, with every separator pressure known.
__global__ __launch_bounds__(256) void k_nd_leaf_solve(FwdParams p, NdDev nd, int k) {
    const int m = blockIdx.x % p.N, bidx = blockIdx.x / p.N;
    const int* LT = nd.leaft + bidx * 256 + threadIdx.x;  // the leaf's row of the flat leaf table (see k_nd_leaf)
    const double* cf = nd.cf + (long long)m * CF_STRIDE;
    double* P = p.P + (long long)m * p.Nxy;
    // box, boundary cells and faces arrive with ONE (coalesced) table read; the leaf's coefficients, its right-hand side, the faces towards
    // the separators and the separator pressures are all requested behind it -- the lane waits for two trips to memory, not four
    const int x0 = LT[0], y0 = LT[NLEAF], w = LT[2 * NLEAF], h = LT[3 * NLEAF];
    double v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int lx = i >> 2, ly = i & 3;
        const bool alive = lx < w && ly < h;
        const double qv = cf[CF_OQ + (alive ? (x0 + lx) * NB + y0 + ly : 0)];
        v[i] = alive ? qv : 0.0;
    }
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        int il, bc;
        double t;
        nd_leaf_boundary(LT, cf, j, il, t, bc);
        const double s = t * P[bc >= 0 ? bc : 0];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] -= (i == il) ? s : 0.0;
    }
    NdLeaf Lf;
    nd_leaf_factor(Lf, cf, x0, y0, w, h);
    nd_leaf_solve(Lf, v);
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (Lf.cell[i] >= 0) P[Lf.cell[i]] = v[i];
}

// ------------------------------------------------------------------------------------------------------------------------
// Levels 10..8: one wave per level-8 subtree (8 x 8 cells: 4 leaves, 2 level-9 fronts, the level-8 front), 4 waves per
// workgroup, 64 workgroups per member.  Per wave in LDS: the subtree's position tables and coefficients (staged once, one
// round trip to memory), two update slots each for levels 10 and 9.  The level-8 update goes to the arena.
// ------------------------------------------------------------------------------------------------------------------------
constexpr int SUB_CF_PLANE = ND_CF_PLANE_SUB;

__device__ __host__ __forceinline__ int nd_sub_lds_doubles(const NdDev& nd) { return ND_LDS_DATA + 2 * (nd.slot9 + nd.slot10) + 4 * SUB_CF_PLANE; }

#ifndef SUB_WPB
#define SUB_WPB 4  // waves (level-8 subtrees) per workgroup
#endif
__global__ __launch_bounds__(64 * SUB_WPB) void k_nd_sub(FwdParams p, NdDev nd, int k) {
    extern __shared__ double nd_lds[];
    // the waves of a workgroup work on the SAME level-8 subtree of SUB_WPB consecutive members: they read the same recipes, front
    // records and position tables at the same time (one trip to L2 for the workgroup instead of one per wave); subtree-major grid
    const int tid = threadIdx.x;
    NdGeo g;
    g.lane = tid & 63;
    g.lc = g.lane & 15;
    g.lq = g.lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mgroups = (p.N + SUB_WPB - 1) / SUB_WPB;
    const int m = SUB_WPB * (blockIdx.x % mgroups) + w, e8 = blockIdx.x / mgroups;
    if (m >= p.N) return;  // (no workgroup barrier below)
    const int* work = nd_work(nd, m);
    if (e8 >= __builtin_amdgcn_readfirstlane(work[0])) return;  // the member's subtrees that are eliminated this step: k_nd_plan's list
    const int i8 = __builtin_amdgcn_readfirstlane(work[ND_W8 + e8]);
    double* blk = nd_lds + w * nd_sub_lds_doubles(nd);  // the wave's LDS block (nd.h): the recipes' offsets refer to it
    double* s9 = blk + ND_LDS_DATA;
    double* s10 = s9 + 2 * nd.slot9;
    double* cfl = s10 + 2 * nd.slot10;
    if (g.lane == 0) { blk[ND_LDS_ZERO] = 0.0; blk[ND_LDS_ONE] = 1.0; }
    const double* cf = nd.cf + (long long)m * CF_STRIDE;
    double* fact = nd.fact + (long long)m * nd.fact_stride;
    double* arena = nd.arena + (long long)m * nd.arena_stride;
    const int f8 = FID(8) + i8, f9 = FID(9) + 2 * i8, f10 = FID(10) + 4 * i8;
    int bad = 0;
    NPROF_DECL;
    const int* F8 = nd.fronts + f8 * ND_FRONT_INTS;
    // per-front scalars of the 7 fronts (wave-uniform): boundary tiles, pivot register rows, recipe and factor offsets
    auto recp = [&](const int* F) { return nd.rec + (long long)__builtin_amdgcn_readfirstlane(F[NDF_REC]) * 256; };
    auto btof = [&](const int* F) { return __builtin_amdgcn_readfirstlane(F[NDF_BT]); };
    auto krof = [&](const int* F) { return __builtin_amdgcn_readfirstlane(F[NDF_KREG]); };
    auto kidof = [&](const int* F) { return __builtin_amdgcn_readfirstlane(F[NDF_KIDM]); };
    auto cofof = [&](const int* F) { return __builtin_amdgcn_readfirstlane(F[NDF_COFM]); };
    const int* F10[4];
    const int* F9[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) F10[i] = nd.fronts + (f10 + i) * ND_FRONT_INTS;
#pragma unroll
    for (int i = 0; i < 2; ++i) F9[i] = nd.fronts + (f9 + i) * ND_FRONT_INTS;
    NdPanelRec<2, true> pp;
    // the four leaves' update matrices: the subtree's block of 4 * slot10 doubles, interleaved by leaf (nd_plan.h: leafu), as one contiguous
    // read of 2 * slot10 double2 (three per lane).  double2 k holds entry k / 2 of leaves 0, 1 (k even) or 2, 3 (k odd): even lanes fill the
    // level-10 slots now, odd lanes keep theirs in registers until the first level-9 front has read the slots
    static_assert(true, "slot10 <= 128 doubles is checked on the host (nd_setup)");
    const int lu_n2 = 2 * nd.slot10;
    double2 lu[3];
    {
        const double2* src = reinterpret_cast<const double2*>(nd.leafu + (long long)m * nd.leafu_stride + (long long)i8 * (4 * nd.slot10));
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int kq = g.lane + 64 * q;
            lu[q] = src[kq < lu_n2 ? kq : lu_n2 - 1];
        }
    }
    auto leaf_slots = [&](int odd) {  // the kept double2 of the even (leaves 0, 1) or odd (leaves 2, 3) lanes -> level-10 slots A, B
        if ((g.lane & 1) == odd) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int kq = g.lane + 64 * q;
                if (kq < lu_n2) { s10[kq >> 1] = lu[q].x; s10[nd.slot10 + (kq >> 1)] = lu[q].y; }
            }
        }
    };
    nd_panel_rec_load(pp, recp(F9[0]), btof(F9[0]), g.lane, kidof(F9[0]), cofof(F9[0]));
    NdCfl L;
    nd_stage_cf(cf, cfl, SUB_CF_PLANE, nd_box(F8, NDF_RBOX), g.lane, L);
    leaf_slots(0);
    nd_wave_fence();
    NPROF(0);
    nd_wave_front<2, true>(btof(F9[0]), krof(F9[0]), pp, recp(F9[0]), blk, s9, fact + F9[0][NDF_FACT], g, bad, kidof(F9[0]), cofof(F9[0]));
    nd_panel_rec_load(pp, recp(F9[1]), btof(F9[1]), g.lane, kidof(F9[1]), cofof(F9[1]));
    nd_wave_fence();
    leaf_slots(1);
    nd_wave_fence();
    NPROF(1);
    nd_wave_front<2, true>(btof(F9[1]), krof(F9[1]), pp, recp(F9[1]), blk, s9 + nd.slot9, fact + F9[1][NDF_FACT], g, bad, kidof(F9[1]), cofof(F9[1]));
    nd_panel_rec_load(pp, recp(F8), btof(F8), g.lane, kidof(F8), cofof(F8));
    nd_wave_fence();
    NPROF(2);
    nd_wave_front<2, true>(btof(F8), krof(F8), pp, recp(F8), blk, arena + F8[NDF_UPD], fact + F8[NDF_FACT], g, bad, kidof(F8), cofof(F8));
    NPROF(3);
#ifdef HM_ND_PROF
    if (blockIdx.x == HM_ND_PROF_SUB_BLOCK && tid == 0)
        for (int i = 0; i < 16; ++i) hm_nd_prof_buf[16 + i] = prof_acc[i];
#endif
    if (bad && g.lane == 0) atomicOr(&p.status[m], HM_MEMBER_BAD_PIVOT);
}

// ------------------------------------------------------------------------------------------------------------------------
// Levels 7, 6, 5: one wave per front.  The two children's update matrices (arena) and the coefficients around the front's
// separator are staged into the wave's LDS first (bulk copies, one round trip); the update goes to the arena.
// ------------------------------------------------------------------------------------------------------------------------
constexpr int WAVE_CF_PLANE = ND_CF_PLANE_WAVE;

template <int LEVEL, int MAXBT, int WPB>
__global__ __launch_bounds__(64 * WPB) void k_nd_wave(FwdParams p, NdDev nd, int k) {
    extern __shared__ double nd_lds[];
    constexpr int NF = 1 << (LEVEL + LO);
    const int m = blockIdx.x % p.N, bidx = blockIdx.x / p.N;
    const int tid = threadIdx.x;
    NdGeo g;
    g.lane = tid & 63;
    g.lc = g.lane & 15;
    g.lq = g.lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int chd = nd.child_doubles[7 - LEVEL];
    double* blk = nd_lds + w * (ND_LDS_DATA + 2 * chd + 4 * WAVE_CF_PLANE);  // the wave's LDS block (nd.h)
    double* c0l = blk + ND_LDS_DATA;
    double* c1l = c0l + chd;
    double* cfl = c1l + chd;
    if (g.lane == 0) { blk[ND_LDS_ZERO] = 0.0; blk[ND_LDS_ONE] = 1.0; }
    const double* cf = nd.cf + (long long)m * CF_STRIDE;
    double* fact = nd.fact + (long long)m * nd.fact_stride;
    double* arena = nd.arena + (long long)m * nd.arena_stride;
    const int* work = nd_work(nd, m);
    const int e = bidx * WPB + w;
    if (e >= __builtin_amdgcn_readfirstlane(work[8 - LEVEL])) return;  // the level's fronts that are eliminated this step (k_nd_plan); no workgroup barrier below
    const int f = NF - 1 + __builtin_amdgcn_readfirstlane(work[(LEVEL == 7 ? ND_W7 : LEVEL == 6 ? ND_W6 : ND_W5) + e]);
    const int* F = nd.fronts + f * ND_FRONT_INTS;
    const int bc0 = F[NDF_BC0], bc1 = F[NDF_BC1];
    const int n0 = (((bc0 + 1) * (bc0 + 2) >> 1) + 1) & ~1, n1 = (((bc1 + 1) * (bc1 + 2) >> 1) + 1) & ~1;
    const int bt = __builtin_amdgcn_readfirstlane(F[NDF_BT]);
    const short* rec = nd.rec + (long long)__builtin_amdgcn_readfirstlane(F[NDF_REC]) * 256;
    const int kidm = __builtin_amdgcn_readfirstlane(F[NDF_KIDM]), cofm = __builtin_amdgcn_readfirstlane(F[NDF_COFM]);
    NPROF_DECL;
    NdPanelRec<MAXBT, true> pr;
    nd_panel_rec_load(pr, rec, bt, g.lane, kidm, cofm);  // in flight beside the bulk copies below
    nd_wave_copy(c0l, arena + F[NDF_UC0], n0, g.lane);
    nd_wave_copy(c1l, arena + F[NDF_UC1], n1, g.lane);
    NdCfl L;
    nd_stage_cf(cf, cfl, WAVE_CF_PLANE, nd_box(F, NDF_PBOX), g.lane, L);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the DMA pieces have landed
    nd_wave_fence();
    NPROF(0);
    int bad = 0;
    nd_wave_front<MAXBT, true>(bt, __builtin_amdgcn_readfirstlane(F[NDF_KREG]), pr, rec, blk, arena + F[NDF_UPD], fact + F[NDF_FACT], g, bad, kidm, cofm);
    NPROF(1);
#ifdef HM_ND_PROF
    if (blockIdx.x == HM_ND_PROF_SUB_BLOCK && tid == 0) { hm_nd_prof_buf[54 + 2 * (7 - LEVEL)] = prof_acc[0]; hm_nd_prof_buf[55 + 2 * (7 - LEVEL)] = prof_acc[1]; }
#endif
    if (bad && g.lane == 0) atomicOr(&p.status[m], HM_MEMBER_BAD_PIVOT);
}

// ------------------------------------------------------------------------------------------------------------------------
// Levels 4..0: one workgroup (16 waves) per member, front after front; the tiles of a front are dealt to the waves.
//   V tiles  (q, R), q < st, R >= q:   the transposed panel tile  F[rows of R][pivots of q]^T     (index = q-major)
//   trailing (R, C), st <= C <= R < T: the update matrix, accumulated over the panels in registers
// Assembly: the children's update matrices are staged into LDS one after the other (bulk copy by the whole workgroup) and
// gathered from there.  Per panel p:  the owner of V(p, p) inverts it -> Pimg (for p > 0 right after that tile's own update:
// the sweep runs beside the other waves' updates);  the owners of V(p, R), R > p, form W_R^T = P V and publish W_R^T and V_R
// as register images (lane-major, conflict-free 8-byte reads) and store the factor;  every later tile is updated with
// Y^T Z products of two published images.
// ------------------------------------------------------------------------------------------------------------------------
constexpr int TOP_NW = 16;
constexpr int TOP_MAXT = LG == 7 ? 13 : (LG == 8 ? 25 : 49);  // tile rows of the largest front of the grid (sizes the back substitution)
// k_nd_top<TOP_NVS, TOP_NTS, TOPK_MAXT, TOPK_LEVEL>: panel / trailing tiles per wave, tile rows of the largest front it takes, and (larger
// grids) the one level it eliminates, a workgroup per front.  128 x 128: <3, 4, 13, 4> (levels 4..0 of a member in one workgroup);
// larger grids: <3, 4, 13, LO + 4> and <2, 6, 15, LO + 3>.

// A child's packed update matrix (n2 double2), arena -> LDS, by the whole workgroup through LDS-DMA (global_load_lds_dwordx4: no
// registers, every piece in flight at once, retired by the issuing wave's vmcnt): pieces of 64 double2 = 1 KB, wave w takes pieces
// w, w + nw, ...; the lanes of the last piece past n2 re-read the last element (the destination is padded to whole pieces).
__device__ __host__ __forceinline__ int top_pad(int doubles) { return (doubles + 127) & ~127; }
__device__ __forceinline__ void top_dma(double* dst, const double* __restrict__ src, int n2, int w, int lane, int nw) {
    const double2* s2 = reinterpret_cast<const double2*>(src);
    double2* d2 = reinterpret_cast<double2*>(dst);
    for (int pc = w; pc * 64 < n2; pc += nw) {
        const int i = pc * 64 + lane;
#if defined(HM_EXP_TOP) && (HM_EXP_TOP & 2)
        __builtin_amdgcn_global_load_lds((nd_glb_ptr)(s2 + lane), (nd_lds_ptr)(d2 + pc * 64), 16, 0, 0);
#else
        __builtin_amdgcn_global_load_lds((nd_glb_ptr)(s2 + (i < n2 ? i : n2 - 1)), (nd_lds_ptr)(d2 + pc * 64), 16, 0, 0);
#endif
    }
}

// Workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding vector-memory operation of the wave,
// which would end the flight of the next front's DMA pieces at the first barrier behind their issue.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

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

// Larger grids (ND_LG > 7): the same kernel eliminates the fronts of level LO + 4 -- the shape of the 128 x 128 tree's level 4: two pivot
// tiles, up to 8 boundary tiles, children packed by k_nd_wave<5> -- and, with six trailing tiles a wave, those of level LO + 3 (two pivot
// tiles, up to 13 boundary tiles), ONE FRONT PER WORKGROUP (grid: members x fronts of the level); the levels above are the big fronts
// (k_big_*).  As big fronts the 64 / 256 fronts of level LO + 4 cost 3.0 ms per 512 members at 256 x 256 (one wave per tile row or 2 x 2
// tiles, every operand through L2); here 1.4.
// NW: waves per workgroup.  ONE: one front per workgroup -- front blockIdx.x / N of level TOPK_LEVEL of member blockIdx.x % N -- instead of
// a member's fronts in turn.  Round 5: the 16 (x 4^h) fronts of level LO + 4, two pivot tiles and at most 8 boundary tiles each, run as
// workgroups of EIGHT waves with 75 KB of LDS (children staged one after the other), TWO of them to a CU: a front is a sequence of phases
// that each load a different unit of the CU (LDS issue in the gathers, one wave's dependent chain in the pivot-tile inverse, the matrix
// pipe in the updates) with barriers in between -- inside one workgroup nothing overlaps them (profiles/r05/nd_top_ablation.txt), a
// second workgroup on the CU does.  128 x 128: that launch first, then levels 3..0 of a member in one workgroup of 16 waves as before.
constexpr __host__ __device__ int top_lds_doubles(int maxt, int child_doubles) { return 256 * (2 * maxt - 1) + 256 + ((child_doubles + 127) & ~127); }
template <int TOP_NVS, int TOP_NTS, int TOPK_MAXT, int TOPK_LEVEL, int NW = TOP_NW, bool ONE = (ND_LG > 7)>
__global__ __launch_bounds__(64 * NW, (NW <= 8 ? 2 : 1)) void k_nd_top(FwdParams p, NdDev nd, int k, int top_child_doubles) {
    extern __shared__ double nd_lds[];
    // images: P, then W(R) and V(R) for tile rows R = 1 .. TOPK_MAXT - 1 (row 0 is never published: a panel's images are those of the
    // rows BELOW its pivot tile), addressed as Wimg + 256 R / Vimg + 256 R
    double* Pimg = nd_lds;                       // 256
    double* Wimg = Pimg;                         // rows 1 .. TOPK_MAXT - 1 behind P
    double* Vimg = Wimg + (TOPK_MAXT - 1) * 256;
    int* cl_s = reinterpret_cast<int*>(Vimg + TOPK_MAXT * 256);  // 16 T ints (<= 832 B) in one DMA piece of 1 KB
    short* cp_s0 = reinterpret_cast<short*>(cl_s + 256);        // 16 T shorts for child 0, then 16 T for child 1, as in memory: one piece
    double* chl = reinterpret_cast<double*>(cp_s0 + 512);       // the children's packed updates
    const int tid = threadIdx.x;
    const int m = ONE ? blockIdx.x % p.N : blockIdx.x;
    NdGeo g;
    g.lane = tid & 63;
    g.lc = g.lane & 15;
    g.lq = g.lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const double* cf = nd.cf + (long long)m * CF_STRIDE;
    double* fact = nd.fact + (long long)m * nd.fact_stride;
    double* arena = nd.arena + (long long)m * nd.arena_stride;
    int bad = 0;
    NPROF_DECL;
    // the records of fronts 0..30 in LDS, read once (they carry their children's sizes and offsets): per front they would otherwise
    // cost two dependent round trips to memory (the front's record, then its children's) before anything else can start
    const int chl_cap = top_pad(top_child_doubles);  // doubles: whole DMA pieces
    int* frec = reinterpret_cast<int*>(chl + chl_cap);  // 31 records behind the child buffer
    // 128 x 128: the fronts of levels 4..0 that are eliminated this step, in order (k_nd_plan: those whose subtree is still dry keep the
    // results they have); larger grids: a `todo` byte per front (k_ndl_plan)
    const int* wlist = nd.work + (long long)m * ND_WORK_INTS + ND_WT;
    // ... kept in a REGISTER, lane i = entry i (count + at most 31 fronts; the block has 64 ints): read from memory per front, the entry came
    // back behind every store the wave had in flight (vmcnt retires in order: the factor rows of the panel before, the update matrix of the
    // front before) -- 3 k cycles between two fronts and 6 k in the first panel of each, where the next front's DMA is issued (round 6,
    // profiles/r06/nd_top_deal.txt)
    const int wl = ND_LG == 7 ? wlist[g.lane] : 0;  // (larger grids: a `todo` byte per front instead)
    auto wl_at = [&](int i) { return __builtin_amdgcn_readlane(wl, i); };  // (i wave-uniform)
    const int f0 = (1 << TOPK_LEVEL) - 1 + (ONE ? blockIdx.x / p.N : 0);
    int first = 0;  // (not ONE) the list entries in front of `first` belong to the levels a launch of their own has eliminated
    if (ONE) {
        // one front: its record at frec[0] (the code below indexes the records by front id: `frec - f0 * ND_FRONT_INTS` makes that this one)
#if ND_LG == 7
        const int nl = wl_at(0);  // (at most 31 entries: levels 4 .. 0 in order)
        if (__ballot(g.lane >= 1 && g.lane <= nl && wl == f0) == 0) return;
#else
        if (!nd.todo[(long long)m * NTODO + f0]) return;  // (the front keeps the results it has; the whole workgroup leaves)
#endif
        if (tid < ND_FRONT_INTS) frec[tid] = nd.fronts[f0 * ND_FRONT_INTS + tid];
        __syncthreads();
        frec -= f0 * ND_FRONT_INTS;
    } else {
        for (int i = tid; i < ((2 << TOPK_LEVEL) - 1) * ND_FRONT_INTS; i += 64 * NW) frec[i] = nd.fronts[i];
        __syncthreads();
        const int nl = wl_at(0);
        while (first < nl && wl_at(1 + first) >= (2 << TOPK_LEVEL) - 1) ++first;
    }
    const int nt = ONE ? 1 : wl_at(0);
    bool prefetched = false;  // (wave-uniform) this front's tables and children were issued during the previous front's panels
    {
        for (int idx = first; idx < nt; ++idx) {
            const int f = ONE ? f0 : wl_at(1 + idx);
#ifdef HM_ND_PROF
            if (blockIdx.x == 0 && tid == 64 * HM_ND_PROF_TOP_WAVE)
                for (int lv = 4; lv >= 0; --lv)
                    if (f == (1 << lv) - 1 || (idx == first && f >= (1 << lv) - 1 && f < (2 << lv) - 1)) hm_nd_prof_buf[48 + lv] = clock64();
#endif
            const int* F = frec + f * ND_FRONT_INTS;
            const int b = __builtin_amdgcn_readfirstlane(F[NDF_B]);
            const int st = __builtin_amdgcn_readfirstlane(F[NDF_ST]);
            const int bt = __builtin_amdgcn_readfirstlane(F[NDF_BT]);
            const int T = st + bt;
            const int kreg_last = __builtin_amdgcn_readfirstlane(F[NDF_KREG]);
            double* fa = fact + __builtin_amdgcn_readfirstlane(F[NDF_FACT]);
            const int nV = st * T - ((st * (st - 1)) >> 1);
            const int nT = b > 0 ? ((bt * (bt + 1)) >> 1) : 0;
            NPROF(0);
            // ---- the front's position tables and its children's updates to LDS: issued during the PREVIOUS front's panels (below) except for
            // the first front; both children at once where they fit the buffer together (level 4), else one after the other
            const int bch[2] = {__builtin_amdgcn_readfirstlane(F[NDF_BC0]), __builtin_amdgcn_readfirstlane(F[NDF_BC1])};
            const int n2c[2] = {(((bch[0] + 1) * (bch[0] + 2) >> 1) + 1) >> 1, (((bch[1] + 1) * (bch[1] + 2) >> 1) + 1) >> 1};
            const int c1off = top_pad(2 * n2c[0]);                       // where the second child goes when both are staged at once
            const bool both = c1off + top_pad(2 * n2c[1]) <= chl_cap;
            auto stage_front = [&](const int* Fn) {  // tables + child 0 (+ child 1) of front record Fn; wave-uniform arguments
                const int con = __builtin_amdgcn_readfirstlane(Fn[NDF_CELLS]);
                const int b0 = __builtin_amdgcn_readfirstlane(Fn[NDF_BC0]), b1 = __builtin_amdgcn_readfirstlane(Fn[NDF_BC1]);
                const int m0 = (((b0 + 1) * (b0 + 2) >> 1) + 1) >> 1, m1 = (((b1 + 1) * (b1 + 2) >> 1) + 1) >> 1;
                if (w == 0) __builtin_amdgcn_global_load_lds((nd_glb_ptr)(nd.cells + con + 4 * g.lane), (nd_lds_ptr)cl_s, 16, 0, 0);
                if (w == 1) __builtin_amdgcn_global_load_lds((nd_glb_ptr)(nd.cpos + 2 * con + 8 * g.lane), (nd_lds_ptr)cp_s0, 16, 0, 0);
                top_dma(chl, arena + __builtin_amdgcn_readfirstlane(Fn[NDF_UC0]), m0, w, g.lane, NW);
                const int o1 = top_pad(2 * m0);
                if (o1 + top_pad(2 * m1) <= chl_cap) top_dma(chl + o1, arena + __builtin_amdgcn_readfirstlane(Fn[NDF_UC1]), m1, w, g.lane, NW);
            };
            // The NEXT front's pieces, TRICKLED (round 6): a CU takes in the 100 KB of a front's children at about 15 bytes a cycle, and a
            // wave's vector-memory instructions issue in order -- issued in one go behind the first panel's barrier, the prefetch held every
            // wave in its issue loop for 7 k cycles, the owner of the next pivot tile among them (13 % of the kernel; wave 0 alone issuing
            // them all only moved the wait to the next barrier: profiles/r06/nd_top_deal.txt).  So a wave issues ONE piece behind each tile
            // update of the panels -- pieces w, w + NW, ... of [cells | positions | child 0 | child 1] -- and what is left after the last panel.
            int nx_gi = 0, nx_tot = 0, nx_con = 0, nx_m0 = 0, nx_m1 = 0, nx_np0 = 0, nx_o1 = 0, nx_u0 = 0, nx_u1 = 0;
            auto dma_setup = [&](const int* Fn) {
                nx_con = __builtin_amdgcn_readfirstlane(Fn[NDF_CELLS]);
                const int b0 = __builtin_amdgcn_readfirstlane(Fn[NDF_BC0]), b1 = __builtin_amdgcn_readfirstlane(Fn[NDF_BC1]);
                nx_m0 = (((b0 + 1) * (b0 + 2) >> 1) + 1) >> 1;
                nx_m1 = (((b1 + 1) * (b1 + 2) >> 1) + 1) >> 1;
                nx_u0 = __builtin_amdgcn_readfirstlane(Fn[NDF_UC0]);
                nx_u1 = __builtin_amdgcn_readfirstlane(Fn[NDF_UC1]);
                nx_o1 = top_pad(2 * nx_m0);
                nx_np0 = (nx_m0 + 63) >> 6;
                nx_tot = 2 + nx_np0 + (nx_o1 + top_pad(2 * nx_m1) <= chl_cap ? (nx_m1 + 63) >> 6 : 0);
                nx_gi = w;
            };
            auto dma_one = [&]() {
                if (nx_gi >= nx_tot) return;
                const int gi = nx_gi;
                nx_gi += NW;
                if (gi == 0) {
                    __builtin_amdgcn_global_load_lds((nd_glb_ptr)(nd.cells + nx_con + 4 * g.lane), (nd_lds_ptr)cl_s, 16, 0, 0);
                } else if (gi == 1) {
                    __builtin_amdgcn_global_load_lds((nd_glb_ptr)(nd.cpos + 2 * nx_con + 8 * g.lane), (nd_lds_ptr)cp_s0, 16, 0, 0);
                } else {
                    int pc = gi - 2, n2 = nx_m0, uo = nx_u0, lo = 0;
                    if (pc >= nx_np0) { pc -= nx_np0; n2 = nx_m1; uo = nx_u1; lo = nx_o1; }
                    const double2* s2 = reinterpret_cast<const double2*>(arena + uo);
                    double2* d2 = reinterpret_cast<double2*>(chl + lo);
                    const int i = pc * 64 + g.lane;
#if defined(HM_EXP_TOP) && (HM_EXP_TOP & 2)
                    __builtin_amdgcn_global_load_lds((nd_glb_ptr)(s2 + g.lane), (nd_lds_ptr)(d2 + pc * 64), 16, 0, 0);
#else
                    __builtin_amdgcn_global_load_lds((nd_glb_ptr)(s2 + (i < n2 ? i : n2 - 1)), (nd_lds_ptr)(d2 + pc * 64), 16, 0, 0);
#endif
                }
            };
            const bool staged_now = !prefetched;
            if (!prefetched) stage_front(F);
            prefetched = false;
            const short* cp_s1 = cp_s0 + 16 * T;
            // ---- my tiles: decode (scalar)
            d4 vt[TOP_NVS], tr[TOP_NTS];
            int vq[TOP_NVS], vR[TOP_NVS], tR[TOP_NTS], tC[TOP_NTS];
#pragma unroll
            for (int s = 0; s < TOP_NVS; ++s) {
                const int idx = s * NW + w;
                int q = -1, R = -1;
                if (idx < nV) {
                    int rem = idx;
                    q = 0;
                    while (rem >= T - q) { rem -= T - q; ++q; }
                    R = q + rem;
                }
                vq[s] = q; vR[s] = R;
            }
            // trailing tiles (round 6): ROW-MAJOR RUNS -- a wave's tiles are neighbours in a tile row, so one W image serves the run -- dealt to
            // the waves that own NO pivot tile V(q, q) where those can hold them all: the owner of the next pivot tile has that tile's update
            // and its 16 dependent pivots between the two barriers of a panel, and with four trailing updates behind them it was the wave
            // every other one waited for (profiles/r06/nd_top_deal.txt).  Which wave holds a tile changes nothing in its arithmetic.
            unsigned ownmask = 0;
            for (int q = 0; q < st; ++q) ownmask |= 1u << ((q * T - ((q * (q - 1)) >> 1)) % NW);
            int tper = (nT + (NW - __builtin_popcount(ownmask)) - 1) / (NW - __builtin_popcount(ownmask));
            if (tper > TOP_NTS || !nd.top_deal) {
                ownmask = 0;
                tper = (nT + NW - 1) / NW;
            }
            const bool ttake = !((ownmask >> w) & 1);
            const int tbase = __builtin_popcount(~ownmask & ((1u << w) - 1)) * tper;
#pragma unroll
            for (int s = 0; s < TOP_NTS; ++s) {
                const int idx = nd.top_deal ? tbase + s : s * NW + w;
                int R = -1, C = -1;
                if (idx < nT && (!nd.top_deal || (ttake && s < tper))) {
                    int rem = idx;
                    R = 0;
                    while (rem > R) { rem -= R + 1; ++R; }
                    C = rem;
                    R += st; C += st;
                }
                tR[s] = R; tC[s] = C;
            }
            // (a prefetched front: every wave has waited for its own pieces in front of the barrier that ended the front before -- below)
            if (staged_now || !nd.top_deal) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA pieces have landed
                __syncthreads();
            }
            NPROF(1);
            // ---- coefficients + child 0.  The boundary rows of a front are ordered by child (nd.h): a tile whose rows or columns hold
            // nothing of a child skips that child's gather, a panel tile with no cell next to a pivot its coefficient loads (round 5:
            // 114 -> 56 tile gathers for a level-4 front)
            const int kidm = __builtin_amdgcn_readfirstlane(F[NDF_KIDM]), cofm = __builtin_amdgcn_readfirstlane(F[NDF_COFM]);
            auto has = [&](int c, int R) { return ((kidm >> (16 * c + R)) & 1) != 0; };  // (tile row R < 16 whenever kidm != -1)
            // (the panel tiles' coefficient loads: all of them issued here, used behind the trailing tiles' gather)
            int cflags = 0;  // two flag bits per entry (4 s + r)
#pragma unroll
            for (int s = 0; s < TOP_NVS; ++s) {
                vt[s] = d4{0.0, 0.0, 0.0, 0.0};
                if (vq[s] >= 0 && ((cofm >> vR[s]) & 1)) {
                    const int pm = 16 * vR[s] + g.lc;
                    const int cm = cl_s[pm];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int pk = 16 * vq[s] + 4 * r + g.lq;
                        int fl;
                        vt[s][r] = nd_coef_issue(cf, cm, cl_s[pk], pk == pm, fl);
                        cflags |= fl << (2 * (4 * s + r));
                    }
                }
            }
            // (trailing tiles first: no loads from memory in them -- the stores of the front before drain meanwhile)
#pragma unroll
            for (int s = 0; s < TOP_NTS; ++s) {
                tr[s] = d4{0.0, 0.0, 0.0, 0.0};
                if (tR[s] >= 0 && has(0, tR[s]) && has(0, tC[s])) {
                    const int pc0 = cp_s0[16 * tC[s] + g.lc];
#pragma unroll
                    for (int r = 0; r < 4; ++r) tr[s][r] = nd_gather(chl, cp_s0[16 * tR[s] + 4 * r + g.lq], pc0);
                }
            }
#pragma unroll
            for (int s = 0; s < TOP_NVS; ++s) {
                if (vq[s] >= 0 && ((cofm >> vR[s]) & 1)) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) vt[s][r] = nd_coef_finish(vt[s][r], (cflags >> (2 * (4 * s + r))) & 3);
                }
                if (vq[s] >= 0 && has(0, vR[s])) {
                    const int pm0 = cp_s0[16 * vR[s] + g.lc];
#pragma unroll
                    for (int r = 0; r < 4; ++r) vt[s][r] += nd_gather(chl, cp_s0[16 * vq[s] + 4 * r + g.lq], pm0);
                }
            }
            NPROF(2);
            if (!both) {
                __syncthreads();
                top_dma(chl, arena + __builtin_amdgcn_readfirstlane(F[NDF_UC1]), n2c[1], w, g.lane, NW);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            const double* ch1 = both ? chl + c1off : chl;
            NPROF(3);
            // ---- child 1
#pragma unroll
            for (int s = 0; s < TOP_NVS; ++s) {
                if (vq[s] >= 0 && has(1, vR[s])) {
                    const int pm1 = cp_s1[16 * vR[s] + g.lc];
#pragma unroll
                    for (int r = 0; r < 4; ++r) vt[s][r] += nd_gather(ch1, cp_s1[16 * vq[s] + 4 * r + g.lq], pm1);
                }
            }
#pragma unroll
            for (int s = 0; s < TOP_NTS; ++s) {
                if (tR[s] >= 0 && has(1, tR[s]) && has(1, tC[s])) {
                    const int pc1 = cp_s1[16 * tC[s] + g.lc];
#pragma unroll
                    for (int r = 0; r < 4; ++r) tr[s][r] += nd_gather(ch1, cp_s1[16 * tR[s] + 4 * r + g.lq], pc1);
                }
            }
            NPROF(4);
            // ---- panels.  The inverse of the NEXT pivot tile is formed by its owner right after that tile's own update (look-ahead):
            // the in-wave sweep then runs beside the other waves' updates instead of in front of a barrier.
            auto sweep_diag = [&](int pp, int kreg) {
                const int ipp = pp * T - ((pp * (pp - 1)) >> 1);  // index of V(pp, pp)
                if (w == ipp % NW) {
                    const int sl = ipp / NW;
#pragma unroll
                    for (int s = 0; s < TOP_NVS; ++s)
                        if (s == sl) {
                            d4 t = vt[s];
                            sweep16_partial(t, g, bad, kreg);
#pragma unroll
                            for (int r = 0; r < 4; ++r) Pimg[r * 64 + g.lane] = t[r];  // -inv
                        }
                }
            };
            auto update_v = [&](int s, int kreg) {  // V(q, R) -= V(pp, q)^T W(pp, R)^T
                const d4 Y = img_load(Vimg + 256 * vq[s], g.lane), Z = img_load(Wimg + 256 * vR[s], g.lane);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                    if (kk < kreg) vt[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(Y[kk], Z[kk], vt[s], 0, 0, 0);
            };
            int fo = 0;  // factor offset of panel p, in 64-double register rows
            sweep_diag(0, st == 1 ? kreg_last : 4);
            NPROF(5);
            for (int pp = 0; pp < st; ++pp) {
                const int kreg = pp == st - 1 ? kreg_last : 4;
                if (pp == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (my stores of the front before are out: see the end of the front)
                lds_barrier();  // P(pp) published; every read of the previous panel's images is done
                NPROF(6);
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
                            img_store(Wimg + 256 * R, g.lane, wv);  // -W^T
                            img_store(Vimg + 256 * R, g.lane, vt[s]);
#pragma unroll
                            for (int r = 0; r < 4; ++r)
#if !(defined(HM_EXP_TOP) && (HM_EXP_TOP & 1))
                                if (r < kreg) __builtin_nontemporal_store(-wv[r], &fa[(fo + (R - pp - 1) * kreg + r) * 64 + g.lane]);
#else
                                ;
#endif
                        }
                    }
                }
                NPROF(7);
                lds_barrier();  // images of panel pp visible; P(pp) no longer read
                NPROF(8);
                if (pp == 0 && idx + 1 < nt) {
                    // every wave is past its gathers: the tables and the child buffer are free -- the NEXT front's go in now, beside the
                    // panels, unless this front is one of its children (possible since fronts are skipped: its update is not written yet)
                    const int fnext = wl_at(2 + idx);  // (not ONE: nt = 1 there)
                    if (f != 2 * fnext + 1 && f != 2 * fnext + 2) {
                        if (nd.top_deal) dma_setup(frec + fnext * ND_FRONT_INTS);
                        else stage_front(frec + fnext * ND_FRONT_INTS);
                        prefetched = true;
                    }
                }
                NPROF(13);
                if (pp + 1 < st) {
                    const int inx = (pp + 1) * T - (((pp + 1) * pp) >> 1);
#ifdef HM_ND_PROF
                    if (w == inx % NW) prof_acc[14] += 1;
#endif
                    // (the owner of the next pivot tile is the wave every other one will wait for: its 16 dependent pivots share the SIMD's
                    // double-precision pipe with three waves of matrix instructions -- it goes first whenever it has an instruction ready)
                    if (w == inx % NW) {
                        __builtin_amdgcn_s_setprio(3);
                        const int sl = inx / NW;
#pragma unroll
                        for (int s = 0; s < TOP_NVS; ++s)
                            if (s == sl) update_v(s, kreg);
                    }
                    sweep_diag(pp + 1, pp + 1 == st - 1 ? kreg_last : 4);
                    if (w == inx % NW) __builtin_amdgcn_s_setprio(0);
                    NPROF(12);
#pragma unroll
                    for (int s = 0; s < TOP_NVS; ++s) {
                        const int idx = s * NW + w;
                        if (vq[s] > pp && idx != inx) {
                            update_v(s, kreg);
                            dma_one();
                        }
                    }
                }
                d4 Y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s = 0; s < TOP_NTS; ++s) {
                    if (tR[s] >= 0) {  // F22(R, C) -= W(pp, R) V(pp, C)^T
#if defined(HM_EXP_S3) && HM_EXP_S3 == 2  // (timing experiment, wrong results: no image loads in the trailing updates)
                        Y = tr[s];
                        const d4 Z = tr[s];
#else
                        if (s == 0 || tR[s] != tR[s > 0 ? s - 1 : 0]) Y = img_load(Wimg + 256 * tR[s], g.lane);  // (a run in one tile row: one W image)
                        const d4 Z = img_load(Vimg + 256 * tC[s], g.lane);
#endif
#if defined(HM_EXP_S3) && HM_EXP_S3 == 1  // (timing experiment, wrong results: no matrix instructions in the trailing updates)
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) asm volatile("" ::"v"(Y[kk]), "v"(Z[kk]));
#else
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk)
                            if (kk < kreg) tr[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(Y[kk], Z[kk], tr[s], 0, 0, 0);
#endif
                        dma_one();
                    }
                }
                if (pp == 0)
                    while (nx_gi < nx_tot) dma_one();  // (a wave with few tiles has pieces left: all out in the first panel, a panel ahead of the wait)
                fo += (T - pp - 1) * kreg;
                NPROF(9);
            }
            while (nx_gi < nx_tot) dma_one();  // (the pieces the panels did not get to)
            // A wave's vector-memory operations retire IN ORDER: whatever waits for a load behind the update's stores waits for the stores to
            // drain at the CU's share of the memory bandwidth (13 bytes a cycle: 100 KB = 8 k cycles; without the stores of this kernel a
            // member takes 592 k cycles instead of 819 k, profiles/r06/nd_top_deal.txt).  So the wait for the next front's pieces comes IN
            // FRONT of the stores, the barrier behind them orders LDS only, and the stores drain beside the next front's gathers out of LDS
            // (its trailing tiles first: the panel tiles' coefficient loads are the first thing that queues behind the stores).
            const bool drain_aside = prefetched && nd.top_deal;
            if (drain_aside) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // ---- the update matrix to the arena
            if (nT > 0) {
                double* out = arena + F[NDF_UPD];
#pragma unroll
                for (int s = 0; s < TOP_NTS; ++s) {
                    if (tR[s] >= 0) {
                        const int j = 16 * (tC[s] - st) + g.lc;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int i = 16 * (tR[s] - st) + 4 * r + g.lq;
#if !(defined(HM_EXP_TOP) && (HM_EXP_TOP & 1))
                            if (j <= i && i <= b) out[((i * (i + 1)) >> 1) + j] = tr[s][r];
#else
                            ;
#endif
                        }
                    }
                }
            }
            NPROF(10);
            // children before parents; the images are free again.  drain_aside: the next front is no parent of this one, and every wave
            // waits for its stores (vmcnt(0) in front of the first panel's barrier) before a later front's children are fetched
            if (drain_aside) lds_barrier();
            else __syncthreads();
            NPROF(11);
        }
    }
#ifdef HM_ND_PROF
    if (blockIdx.x == 0 && tid == 64 * HM_ND_PROF_TOP_WAVE) {
        for (int i = 0; i < 16; ++i) hm_nd_prof_buf[i] = prof_acc[i];
        hm_nd_prof_buf[53] = clock64();
    }
#endif
    if (bad && g.lane == 0) atomicOr(&p.status[m], HM_MEMBER_BAD_PIVOT);
}

#if ND_LG > 7
// ------------------------------------------------------------------------------------------------------------------------
// The BIG fronts of the larger grids (levels 0 .. LO + 2: up to 25 tile rows at 256 x 256, 49 at 512 x 512 -- 0.65 / 2.5 MB a front,
// more than a CU's register file).  Same elimination as k_nd_top -- same tiles, same products, same order of additions per tile --
// but LEFT-LOOKING out of global memory, one WAVE per task, no synchronisation inside a launch:
//   pivot tile columns are taken in blocks of BIG_PB.  A tile (q, R) of block g (transposed panel tile: pivots of tile column q x front
//   rows of tile row R) is assembled -- coefficients + the children's update matrices gathered through the position tables -- and
//   takes the updates of every earlier block at once, V(q, R) += V(p, q)^T W(p, R)^T for p < g BIG_PB, straight from the stored
//   panels:  `fact` holds +W^T (the factor rows the back substitution reads), `vfac` the NEGATED panel tiles -V(p, R) at the same
//   offsets, so that (-V)(−Wn) = V Wn is the product of two stored images.
//     k_big_diag   one wave per front: the block's own BIG_PB x BIG_PB tiles; in-wave inverses of its pivot tiles (-> `pimg`), its
//                  off-diagonal tiles eliminated in registers
//     k_big_rows   one wave per row tile R below the block: W(p, R)^T = P(p) V(p, R) for the block's pivot tiles in turn, the later
//                  tiles of the row updated with the diagonal block's stored tiles
//     k_big_trail  one wave per 2 x 2 block of trailing tiles: children's updates gathered + the sum over ALL pivot tile columns, written
//                  as whole tiles (nd.h) for the parent's gathers
// A launch covers every front of a level and every member; levels run leaves-to-root.  Each tile is assembled, written and read
// once; what a front costs in memory traffic is its factor (written once, read by the later blocks and the trailing products).
// ------------------------------------------------------------------------------------------------------------------------
constexpr int BIG_PB = 4;

struct BigFront {  // wave-uniform
    int b, st, bt, T;
    const int* cl;         // position -> cell
    const short* cp[2];    // position -> position in child c's boundary list
    const double* ch[2];   // children's update matrices
    bool tiles[2];         // ... stored as whole tiles (big children) or packed (children of level LO + 5)
    bool kids;
    double* fa;            // factor (+W^T tiles)
    double* vf;            // -V tiles, same offsets
    double* pi;            // -inverse pivot tiles
    double* upd;           // this front's update matrix (whole tiles)
};
__device__ __forceinline__ void big_front(BigFront& B, const NdDev& nd, int m, int f) {
    const int* F = nd.fronts + f * ND_FRONT_INTS;
    B.b = __builtin_amdgcn_readfirstlane(F[NDF_B]);
    B.st = __builtin_amdgcn_readfirstlane(F[NDF_ST]);
    B.bt = __builtin_amdgcn_readfirstlane(F[NDF_BT]);
    B.T = B.st + B.bt;
    const int con = __builtin_amdgcn_readfirstlane(F[NDF_CELLS]);
    B.cl = nd.cells + con;
    B.cp[0] = nd.cpos + 2 * (long long)con;
    B.cp[1] = B.cp[0] + 16 * B.T;
    const double* arena = nd.arena + (long long)m * nd.arena_stride;
    B.ch[0] = arena + __builtin_amdgcn_readfirstlane(F[NDF_UC0]);
    B.ch[1] = arena + __builtin_amdgcn_readfirstlane(F[NDF_UC1]);
    const int lv = __builtin_amdgcn_readfirstlane(F[NDF_LEVEL]);
    B.tiles[0] = B.tiles[1] = lv + 1 <= LO + 2;  // (levels LO + 3, LO + 4: packed by k_nd_top)
    B.kids = true;  // (every big front has children)
    const long long fo = __builtin_amdgcn_readfirstlane(F[NDF_FACT]);
    B.fa = nd.fact + (long long)m * nd.fact_stride + fo;
    B.vf = nd.vfac + (long long)m * nd.vfac_stride + fo;
    B.pi = nd.pimg + (long long)m * nd.pimg_stride + __builtin_amdgcn_readfirstlane(F[NDF_PIMG]);
    B.upd = nd.arena + (long long)m * nd.arena_stride + __builtin_amdgcn_readfirstlane(F[NDF_UPD]);
}
// offset (doubles) of the stored tile (p, R), R > p, of a front with T tile rows (every pivot tile of a big front is full: kreg = 4)
__device__ __forceinline__ int big_img(int T, int p, int R) { return 256 * (p * T - ((p * (p + 1)) >> 1) + (R - p - 1)); }
// entry (hi, lo), hi >= lo, of an update matrix stored as whole tiles (nd.h)
__device__ __forceinline__ int big_tile_off(int hi, int lo) {
    const int R = hi >> 4, C = lo >> 4;
    return ((((R * (R + 1)) >> 1) + C) << 8) + (((hi & 15) >> 2) << 6) + ((hi & 3) << 4) + (lo & 15);
}
__device__ __forceinline__ double big_gather(const double* __restrict__ ch, bool tiles, int a, int c) {
    const bool ok = a >= 0 && c >= 0;
    const int hi = a > c ? a : c, lo = a > c ? c : a;
    const int off = tiles ? big_tile_off(hi, lo) : ((hi * (hi + 1)) >> 1) + lo;
    const double l = ch[ok ? off : 0];  // (unconditional load, then the mask: see nd_coef_global)
    return l * (ok ? 1.0 : 0.0);
}
// t += the children's entries of the tile whose rows are the front positions rowpos0 + 4 r + lq and whose columns are colpos0 + lc; a child
// that touches none of them (wave-uniform) costs two table reads
__device__ __forceinline__ void big_gather_tile(d4& t, const BigFront& B, int rowpos0, int colpos0, const NdGeo& g) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const short* cp = B.cp[c];
        const int cc = cp[colpos0 + g.lc];
        int a[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] = cp[rowpos0 + 4 * r + g.lq];
        const bool any = cc >= 0 && (a[0] >= 0 || a[1] >= 0 || a[2] >= 0 || a[3] >= 0);
        if (__ballot(any) == 0ull) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] += big_gather(B.ch[c], B.tiles[c], a[r], cc);
    }
}
// the assembled panel tile (q, R): coefficients of the five-point system + the children
__device__ __forceinline__ d4 big_assemble_v(const BigFront& B, const double* __restrict__ cf, int q, int R, const NdGeo& g) {
    d4 t;
    const int pm = 16 * R + g.lc, cm = B.cl[pm];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int pk = 16 * q + 4 * r + g.lq;
        t[r] = nd_coef_global(cf, cm, B.cl[pk], pk == pm);
    }
    big_gather_tile(t, B, 16 * q, 16 * R, g);
    return t;
}
__device__ __forceinline__ d4 big_mfma4(const d4& Y, const d4& Z, d4 acc) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Y[kk], Z[kk], acc, 0, 0, 0);
    return acc;
}
__device__ __forceinline__ d4 big_neg(const d4& v) { return d4{-v[0], -v[1], -v[2], -v[3]}; }
// Workgroup -> (member, front of the level, workgroup of the front), XCD-aware: workgroups are dealt round-robin to the 8 XCDs, each with an L2
// of its own, and the `wgs` workgroups of one (member, front) read the same panel tiles (a trailing tile's operands are shared by a whole tile
// row / column of the front) -- so they get CONSECUTIVE slots of ONE XCD: they run at the same time behind the same L2.  (Member-fastest
// order put a front's workgroups N x fronts apart: every task fetched its operands from HBM -- k_big_trail moved 15 MB per member-step.)
__device__ __forceinline__ bool big_task(int N, int nf, int wgs, int& m, int& fi, int& wg) {
    const int xcd = blockIdx.x & 7, t = blockIdx.x >> 3;
    wg = t % wgs;
    const int pr = (t / wgs) * 8 + xcd;
    m = pr % N;
    fi = pr / N;
    return pr < N * nf;
}
__host__ inline unsigned big_grid(int N, int nf, int wgs) { return 8u * (unsigned)((N * nf + 7) / 8) * (unsigned)wgs; }

// The diagonal block of pivot block g0 (grid: members x fronts of the level, one wave each).
__global__ __launch_bounds__(64) void k_big_diag(FwdParams p, NdDev nd, int level, int g0) {
    const int m = blockIdx.x % p.N, f = (1 << level) - 1 + blockIdx.x / p.N;
    NdGeo g;
    g.lane = threadIdx.x;
    g.lc = g.lane & 15;
    g.lq = g.lane >> 4;
    if (!nd.todo[(long long)m * NTODO + f]) return;  // (k_ndl_plan)
    BigFront B;
    big_front(B, nd, m, f);
    const int q0 = g0 * BIG_PB;
    if (q0 >= B.st) return;
    const int nq = min(BIG_PB, B.st - q0);
    const double* cf = nd.cf + (long long)m * CF_STRIDE;
    d4 D[BIG_PB][BIG_PB];  // D[a][c], a <= c: tile (q0 + a, q0 + c)
#pragma unroll
    for (int a = 0; a < BIG_PB; ++a)
#pragma unroll
        for (int c = a; c < BIG_PB; ++c)
            if (c < nq) D[a][c] = big_assemble_v(B, cf, q0 + a, q0 + c, g);
    for (int pp = 0; pp < q0; ++pp) {  // the earlier blocks' updates
        d4 Y[BIG_PB], Z[BIG_PB];
#pragma unroll
        for (int a = 0; a < BIG_PB; ++a)
            if (a < nq) {
                Y[a] = img_load(B.vf + big_img(B.T, pp, q0 + a), g.lane);
                Z[a] = img_load(B.fa + big_img(B.T, pp, q0 + a), g.lane);
            }
#pragma unroll
        for (int a = 0; a < BIG_PB; ++a)
#pragma unroll
            for (int c = a; c < BIG_PB; ++c)
                if (c < nq) D[a][c] = big_mfma4(Y[a], Z[c], D[a][c]);
    }
    int bad = 0;
#pragma unroll
    for (int a = 0; a < BIG_PB; ++a) {
        if (a >= nq) break;
        const int pp = q0 + a;
        d4 Pn = D[a][a];
        sweep16_partial(Pn, g, bad, 4);  // -inverse
        img_store(B.pi + 256 * pp, g.lane, Pn);
        d4 W[BIG_PB];
#pragma unroll
        for (int c = a + 1; c < BIG_PB; ++c)
            if (c < nq) {
                W[c] = big_mfma4(Pn, D[a][c], d4{0.0, 0.0, 0.0, 0.0});  // -W^T
                img_store(B.fa + big_img(B.T, pp, q0 + c), g.lane, big_neg(W[c]));
                img_store(B.vf + big_img(B.T, pp, q0 + c), g.lane, big_neg(D[a][c]));
            }
#pragma unroll
        for (int a2 = a + 1; a2 < BIG_PB; ++a2)
#pragma unroll
            for (int c = a2; c < BIG_PB; ++c)
                if (c < nq) D[a2][c] = big_mfma4(D[a][a2], W[c], D[a2][c]);
    }
    if (bad && g.lane == 0) atomicOr(&p.status[m], HM_MEMBER_BAD_PIVOT);
}

// The rows below the diagonal block of pivot block g0 (grid: members x fronts x ceil(rows / 4) workgroups of four waves, one row tile each).
__global__ __launch_bounds__(256) void k_big_rows(FwdParams p, NdDev nd, int level, int g0, int wgs) {
    const int nf = 1 << level;
    int m, fi, wg;
    if (!big_task(p.N, nf, wgs, m, fi, wg)) return;
    const int f = nf - 1 + fi;
    NdGeo g;
    g.lane = threadIdx.x & 63;
    g.lc = g.lane & 15;
    g.lq = g.lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (!nd.todo[(long long)m * NTODO + f]) return;  // (k_ndl_plan)
    BigFront B;
    big_front(B, nd, m, f);
    const int q0 = g0 * BIG_PB;
    if (q0 >= B.st) return;
    const int nq = min(BIG_PB, B.st - q0);
    const int R = q0 + nq + 4 * wg + w;
    if (R >= B.T) return;
    const double* cf = nd.cf + (long long)m * CF_STRIDE;
    // (requesting the diagonal block's tiles up front instead of inside the elimination below was measured: 200 instead of 128 registers,
    // two waves per SIMD instead of four, 6.3 against 5.0 ms per time step over the levels)
    d4 V[BIG_PB];
#pragma unroll
    for (int a = 0; a < BIG_PB; ++a)
        if (a < nq) V[a] = big_assemble_v(B, cf, q0 + a, R, g);
    for (int pp = 0; pp < q0; ++pp) {
        const d4 Z = img_load(B.fa + big_img(B.T, pp, R), g.lane);
#pragma unroll
        for (int a = 0; a < BIG_PB; ++a)
            if (a < nq) V[a] = big_mfma4(img_load(B.vf + big_img(B.T, pp, q0 + a), g.lane), Z, V[a]);
    }
#pragma unroll
    for (int a = 0; a < BIG_PB; ++a) {
        if (a >= nq) break;
        const int pp = q0 + a;
        const d4 Pn = img_load(B.pi + 256 * pp, g.lane);
        const d4 W = big_mfma4(Pn, V[a], d4{0.0, 0.0, 0.0, 0.0});  // -W^T
        const d4 Wp = big_neg(W);
        img_store(B.fa + big_img(B.T, pp, R), g.lane, Wp);
        img_store(B.vf + big_img(B.T, pp, R), g.lane, big_neg(V[a]));
#pragma unroll
        for (int a2 = a + 1; a2 < BIG_PB; ++a2)
            if (a2 < nq) V[a2] = big_mfma4(img_load(B.vf + big_img(B.T, pp, q0 + a2), g.lane), Wp, V[a2]);  // (-V)(+W^T) = V (-W^T)
    }
}

// The update matrix (grid: members x fronts x ceil(blocks / 4) workgroups of four waves, one 2 x 2 block of lower tiles each).
__global__ __launch_bounds__(256) void k_big_trail(FwdParams p, NdDev nd, int level, int wgs) {
    const int nf = 1 << level;
    int m, fi, wg;
    if (!big_task(p.N, nf, wgs, m, fi, wg)) return;
    const int f = nf - 1 + fi;
    NdGeo g;
    g.lane = threadIdx.x & 63;
    g.lc = g.lane & 15;
    g.lq = g.lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (!nd.todo[(long long)m * NTODO + f]) return;  // (k_ndl_plan)
    BigFront B;
    big_front(B, nd, m, f);
    const int hb = (B.bt + 1) >> 1;
    int rem = 4 * wg + w, Rb = 0;
    if (rem >= ((hb * (hb + 1)) >> 1)) return;
    while (rem > Rb) { rem -= Rb + 1; ++Rb; }
    const int Cb = rem;
    const int R0 = B.st + 2 * Rb, C0 = B.st + 2 * Cb;
    const int R1 = R0 + 1 < B.T ? R0 + 1 : R0, C1 = C0 + 1 < B.T ? C0 + 1 : C0;  // (a surplus tile recomputes its neighbour and is not stored)
    // the first panel's images are requested before the children's entries are gathered (two dependent round trips to memory: position
    // tables, then the entries), every later panel's while the previous one's products run
    d4 Y0 = img_load(B.fa + big_img(B.T, 0, R0), g.lane), Y1 = img_load(B.fa + big_img(B.T, 0, R1), g.lane);
    d4 Z0 = img_load(B.vf + big_img(B.T, 0, C0), g.lane), Z1 = img_load(B.vf + big_img(B.T, 0, C1), g.lane);
    d4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};
            if (R0 + i < B.T && C0 + j <= R0 + i) big_gather_tile(acc[i][j], B, 16 * (R0 + i), 16 * (C0 + j), g);
        }
    for (int pp = 0; pp < B.st; ++pp) {
        const int pn = pp + 1 < B.st ? pp + 1 : pp;
        const d4 Y0n = img_load(B.fa + big_img(B.T, pn, R0), g.lane), Y1n = img_load(B.fa + big_img(B.T, pn, R1), g.lane);
        const d4 Z0n = img_load(B.vf + big_img(B.T, pn, C0), g.lane), Z1n = img_load(B.vf + big_img(B.T, pn, C1), g.lane);
        acc[0][0] = big_mfma4(Y0, Z0, acc[0][0]);
        acc[0][1] = big_mfma4(Y0, Z1, acc[0][1]);
        acc[1][0] = big_mfma4(Y1, Z0, acc[1][0]);
        acc[1][1] = big_mfma4(Y1, Z1, acc[1][1]);
        Y0 = Y0n; Y1 = Y1n; Z0 = Z0n; Z1 = Z1n;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int Rt = R0 + i - B.st, Ct = C0 + j - B.st;
            if (R0 + i < B.T && Ct <= Rt) img_store(B.upd + ((((Rt * (Rt + 1)) >> 1) + Ct) << 8), g.lane, acc[i][j]);
        }
}
#endif  // ND_LG

// ------------------------------------------------------------------------------------------------------------------------
// Back substitution, root to leaves: per front and panel (last first)  x1 = -W^T [x of the rows below; -1 for the rhs row],
// one wave per front, levels separated by workgroup barriers; pressures in P; then the face fluxes.
// ------------------------------------------------------------------------------------------------------------------------
constexpr int SOL_NW = 4;
#ifndef SOL_OCC
#define SOL_OCC 4
#endif

// Fronts with one pivot tile (levels >= 5), UNR of them per wave at a time: the cell indices of all of them, then the known
// pressures and the factor tiles of all of them are requested before anything is used -- two round trips to memory per UNR
// fronts (the back substitution does 16 multiply-adds per 8 bytes: it is a stream of the factor, bound by loads in flight).
template <int UNR, int MAXBT>
struct NdSolveIdx {
    int bt[UNR], kreg[UNR], cb[UNR][MAXBT], cpv[UNR][4];
    const double* fa[UNR];
};
template <int UNR, int MAXBT>
__device__ __forceinline__ void nd_solve_idx(NdSolveIdx<UNR, MAXBT>& I, const NdDev& nd, const double* __restrict__ fact, int nf, int fi0, const NdGeo& g) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
        const int fi = fi0 + u < nf ? fi0 + u : nf - 1;  // (a wave's surplus slots redo its last front: same values, harmless)
        const int* F = nd.fronts + (nf - 1 + fi) * ND_FRONT_INTS;
        I.bt[u] = __builtin_amdgcn_readfirstlane(F[NDF_BT]);
        I.kreg[u] = __builtin_amdgcn_readfirstlane(F[NDF_KREG]);
        const int* cl = nd.cells + __builtin_amdgcn_readfirstlane(F[NDF_CELLS]);
        I.fa[u] = fact + F[NDF_FACT];
#pragma unroll
        for (int R = 0; R < MAXBT; ++R) I.cb[u][R] = cl[16 * (1 + (R < I.bt[u] ? R : 0)) + g.lc];
#pragma unroll
        for (int r = 0; r < 4; ++r) I.cpv[u][r] = cl[4 * r + g.lq];
    }
}
template <int UNR, int MAXBT>
__device__ __forceinline__ void nd_solve_single(const NdDev& nd, const double* __restrict__ fact, double* P, int nf, int w_in, const NdGeo& g, int nwaves = SOL_NW) {
    const int w = w_in;  // this wave's index among the `nwaves` waves that share the level's fronts (one workgroup, or several: k_nd_solve_level)
    // software pipeline: the cell indices of batch i + 1 are requested before batch i's values are waited for, so a batch costs one
    // round trip to memory, not two
    NdSolveIdx<UNR, MAXBT> I, In;
    nd_solve_idx(I, nd, fact, nf, w * UNR < nf ? w * UNR : 0, g);
    for (int fi0 = w * UNR; fi0 < nf; fi0 += nwaves * UNR) {
        double xv[UNR][MAXBT], t[UNR][MAXBT][4];
#pragma unroll
        for (int u = 0; u < UNR; ++u)
#pragma unroll
            for (int R = 0; R < MAXBT; ++R) {
                const int c = I.cb[u][R];
                const double l = P[c >= 0 ? c : 0];
                xv[u][R] = R < I.bt[u] ? l * (c >= 0 ? 1.0 : 0.0) + (c == -2 ? -1.0 : 0.0) : 0.0;
                const int Rc = R < I.bt[u] ? R : 0;
#pragma unroll
#ifdef HM_ND_SOLVE_NOLOAD
                for (int r = 0; r < 4; ++r) t[u][R][r] = 1e-3 * (Rc + r);
#else
                for (int r = 0; r < 4; ++r) t[u][R][r] = I.fa[u][(Rc * I.kreg[u] + (r < I.kreg[u] ? r : 0)) * 64 + g.lane];
#endif
            }
        const int fin = fi0 + nwaves * UNR;
        nd_solve_idx(In, nd, fact, nf, fin < nf ? fin : fi0, g);
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int R = 0; R < MAXBT; ++R)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = fma(t[u][R][r], xv[u][R], acc[r]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double v = acc[r];
#ifndef HM_ND_SOLVE_NOSHFL
                v += __shfl_xor(v, 8);
                v += __shfl_xor(v, 4);
                v += __shfl_xor(v, 2);
                v += __shfl_xor(v, 1);
#endif
                if (g.lc == 0 && r < I.kreg[u] && I.cpv[u][r] >= 0) P[I.cpv[u][r]] = -v;
            }
        }
        I = In;
    }
}

#ifndef SOL_DEPTH
#define SOL_DEPTH 8  // factor tiles a wave keeps in flight in the fronts of levels 0 .. 4 (64 of its 128 registers)
#endif
__global__ __launch_bounds__(64 * SOL_NW, SOL_OCC) void k_nd_solve(FwdParams p, NdDev nd, int k) {
    __shared__ double xe_all[SOL_NW][16 * TOP_MAXT];
    __shared__ int pcell_all[SOL_NW][16 * TOP_MAXT];  // the current front's pivot cells
    const int m = blockIdx.x, tid = threadIdx.x;
    NdGeo g;
    g.lane = tid & 63;
    g.lc = g.lane & 15;
    g.lq = g.lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Nxy = p.Nxy;
    double* P = p.P + (long long)m * Nxy;
    const double* fact = nd.fact + (long long)m * nd.fact_stride;
    double* xe = xe_all[w];
    NPROF_DECL;
    for (int lvg = 0; lvg < (LO == 0 ? 8 : LO + 5); ++lvg) {  // levels 8..10 (128 x 128 numbering): k_nd_solve_sub; larger grids: levels 5..7 by k_nd_solve_level
        const int nf = 1 << lvg, lv = lvg - LO;
        NPROF(lv < 5 ? 12 : lv - 5);
        if (lv >= 5) {
            if (lv == 5) nd_solve_single<1, 6>(nd, fact, P, nf, w, g);
            else if (lv == 6) nd_solve_single<2, 4>(nd, fact, P, nf, w, g);
            else if (lv == 7) nd_solve_single<2, 3>(nd, fact, P, nf, w, g);
            else if (lv == 10) nd_solve_single<4, 1>(nd, fact, P, nf, w, g);
            else nd_solve_single<4, 2>(nd, fact, P, nf, w, g);
            NPROF(lv - 5 + 6 > 11 ? 11 : lv);
            __syncthreads();
            continue;
        }
        for (int fi = w; fi < nf; fi += SOL_NW) {
            const int f = nf - 1 + fi;
            const int* F = nd.fronts + f * ND_FRONT_INTS;
            const int st = __builtin_amdgcn_readfirstlane(F[NDF_ST]);
            const int bt = __builtin_amdgcn_readfirstlane(F[NDF_BT]);
            const int T = st + bt;
            const int kreg_last = __builtin_amdgcn_readfirstlane(F[NDF_KREG]);
            const int* cl = nd.cells + __builtin_amdgcn_readfirstlane(F[NDF_CELLS]);
            const double* fa = fact + F[NDF_FACT];
            // boundary values (ancestors' pivots, already known), -1 on the right-hand-side row; the front's own pivot cells to LDS in the same
            // trip to memory (round 6: every panel used to fetch its 16 from the position table when it was done -- a trip per panel)
            int* pcell = pcell_all[w];
            for (int pos = g.lane; pos < 16 * st; pos += 64) pcell[pos] = cl[pos];
            for (int pos = 16 * st + g.lane; pos < 16 * T; pos += 64) {
                const int c = cl[pos];
                xe[pos] = c >= 0 ? P[c] : (c == -2 ? -1.0 : 0.0);
            }
            nd_wave_fence();
            int fo = 0;
            for (int pp = 0; pp < st - 1; ++pp) fo += (T - pp - 1) * 4;
            if (kreg_last == 4) {
                // Round 6: the front's factor as ONE stream.  The tiles are consumed in a fixed order -- panels st - 1 .. 0, within a panel the
                // tile rows pp + 1 .. T - 1 -- at addresses that do not depend on the solution, so SOL_DEPTH tiles are kept in flight in
                // registers across panel boundaries: the chain of dependent trips to memory (one per four tiles, two per panel: 95 for levels
                // 0 .. 4 of a member) becomes one trip per front plus the stream.  With a whole ensemble resident the phase is bound by HBM
                // bandwidth either way (1.28 MB of factor per member at 4.9 TB/s: profiles/r06/nd_cycle_stamps_mid_grid.txt); a SMALL member
                // shard -- 125 members: one rank's share of config 2 over 8 GPUs -- does not fill the memory system, and there the chain was
                // the kernel's time (0.27 of the shard's 1.13 ms pressure step).  Same products, same order of additions per pivot (tile rows
                // ascending): the same bits.
                const int ntot = st * T - ((st * (st + 1)) >> 1);
                d4 buf[SOL_DEPTH];
                int ppf = st - 1, Rf = st, fof = fo;  // the prefetch stream's panel, tile row, factor offset
                auto fetch = [&](d4& b, bool real) {  // (`real`: wave-uniform; a slot past the end re-reads the front's first tile: the load
                                                      // count per iteration stays static, which is what lets the compiler wait by count)
                    const double* tl = fa + (real ? (long long)(fof + (Rf - ppf - 1) * 4) * 64 : 0LL) + g.lane;
#pragma unroll
                    for (int r = 0; r < 4; ++r) b[r] = tl[r * 64];
                    if (real && ++Rf == T) {
                        --ppf;
                        fof -= (T - ppf - 1) * 4;
                        Rf = ppf + 1;
                    }
                };
#pragma unroll
                for (int j = 0; j < SOL_DEPTH; ++j) fetch(buf[j], j < ntot);
                int pp = st - 1, R = st;
                double acc[4] = {0.0, 0.0, 0.0, 0.0};
                for (int base = 0; base < ntot; base += SOL_DEPTH) {
#pragma unroll
                    for (int j = 0; j < SOL_DEPTH; ++j) {
                        const int i = base + j;
                        const bool live = i < ntot;  // (wave-uniform)
                        const double xv = xe[16 * (live ? R : 0) + g.lc];
                        if (live) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[r] = fma(buf[j][r], xv, acc[r]);
                        }
                        fetch(buf[j], i + SOL_DEPTH < ntot);
                        if (live && ++R == T) {  // the panel is complete: its 16 pivots
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
                                    const int c = pcell[pos];
                                    const double v = c >= 0 ? acc[r] : 0.0;
                                    xe[pos] = v;
                                    if (c >= 0) P[c] = v;
                                }
                            }
                            nd_wave_fence();
                            --pp;
                            R = pp + 1;
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[r] = 0.0;
                        }
                    }
                }
                continue;
            }
            for (int pp = st - 1; pp >= 0; --pp) {
                const int kreg = pp == st - 1 ? kreg_last : 4;
                double acc[4] = {0.0, 0.0, 0.0, 0.0};
                if (kreg == 4) {
#pragma unroll 4
                    for (int R = pp + 1; R < T; ++R) {
                        const double xv = xe[16 * R + g.lc];
                        const double* tl = fa + (long long)(fo + (R - pp - 1) * 4) * 64 + g.lane;
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[r] = fma(tl[r * 64], xv, acc[r]);
                    }
                } else {
#pragma unroll 2
                    for (int R = pp + 1; R < T; ++R) {
                        const double xv = xe[16 * R + g.lc];
                        const double* tl = fa + (long long)(fo + (R - pp - 1) * kreg) * 64 + g.lane;
#pragma unroll
                        for (int r = 0; r < 3; ++r) {
                            const double l = tl[(r < kreg ? r : 0) * 64];
                            acc[r] = fma(l * (r < kreg ? 1.0 : 0.0), xv, acc[r]);
                        }
                    }
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
    NPROF(13);
#ifdef HM_ND_PROF
    if (blockIdx.x == 0 && tid == 0)
        for (int i = 0; i < 16; ++i) hm_nd_prof_buf[32 + i] = prof_acc[i];
#endif
}

#if ND_LG > 7
// Larger grids: the back substitution of ONE of the levels 0 .. LO + 4 as a launch of its own (inside k_nd_solve's one workgroup per member the
// 4 waves took a level's fronts in turn: 256 fronts of level 8 at 512 x 512 are 64 rounds, and 125 members are 125 workgroups on 256 CUs).
// WPF = 4: the four waves of a workgroup share one front, each taking every fourth tile of a panel's row (the fronts of levels 0 .. LO + 2:
// up to 48 tiles a panel, 32 panels); WPF = 1: a front per wave.  Every pivot tile of these levels is full (kreg = 4, checked by nd_setup).
template <int WPF>
__global__ __launch_bounds__(256) void k_nd_solve_front(FwdParams p, NdDev nd, int level) {
    __shared__ double xe_all[WPF == 4 ? 1 : 4][16 * TOP_MAXT];
    __shared__ double part[4][16];
    const int nf = 1 << level, tid = threadIdx.x;
    const int m = blockIdx.x % p.N, b = blockIdx.x / p.N;
    NdGeo g;
    g.lane = tid & 63;
    g.lc = g.lane & 15;
    g.lq = g.lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fi = WPF == 4 ? b : 4 * b + w;
    if (fi >= nf) return;  // (WPF = 4: the whole workgroup; WPF = 1: no workgroup barrier below)
    double* P = p.P + (long long)m * p.Nxy;
    const double* fact = nd.fact + (long long)m * nd.fact_stride;
    double* xe = xe_all[WPF == 4 ? 0 : w];
    const int* F = nd.fronts + (nf - 1 + fi) * ND_FRONT_INTS;
    const int st = __builtin_amdgcn_readfirstlane(F[NDF_ST]);
    const int T = st + __builtin_amdgcn_readfirstlane(F[NDF_BT]);
    const int* cl = nd.cells + __builtin_amdgcn_readfirstlane(F[NDF_CELLS]);
    const double* fa = fact + F[NDF_FACT];
    // boundary values (ancestors' pivots, already known), -1 on the right-hand-side row
    for (int pos = 16 * st + (WPF == 4 ? tid : g.lane); pos < 16 * T; pos += WPF == 4 ? 256 : 64) {
        const int c = cl[pos];
        xe[pos] = c >= 0 ? P[c] : (c == -2 ? -1.0 : 0.0);
    }
    if (WPF == 4) __syncthreads();
    else nd_wave_fence();
    for (int pp = st - 1; pp >= 0; --pp) {
        const long long fo = 4LL * (pp * T - ((pp * (pp + 1)) >> 1));  // register rows in front of panel pp (big_img)
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
        for (int R = pp + 1 + (WPF == 4 ? w : 0); R < T; R += WPF) {
            const double xv = xe[16 * R + g.lc];
            const double* tl = fa + (fo + (R - pp - 1) * 4) * 64 + g.lane;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = fma(tl[r * 64], xv, acc[r]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double v = acc[r];
            v += __shfl_xor(v, 8);
            v += __shfl_xor(v, 4);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 1);
            acc[r] = v;
        }
        if (WPF == 4) {
            if (g.lc == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) part[w][4 * r + g.lq] = acc[r];
            }
            __syncthreads();
            if (tid < 16) {  // (fixed order of the four partial sums)
                const double v = -(((part[0][tid] + part[1][tid]) + part[2][tid]) + part[3][tid]);
                const int pos = 16 * pp + tid, c = cl[pos];
                const double x = c >= 0 ? v : 0.0;
                xe[pos] = x;
                if (c >= 0) P[c] = x;
            }
            __syncthreads();
        } else {
            if (g.lc == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int pos = 16 * pp + 4 * r + g.lq, c = cl[pos];
                    const double x = c >= 0 ? -acc[r] : 0.0;
                    xe[pos] = x;
                    if (c >= 0) P[c] = x;
                }
            }
            nd_wave_fence();
        }
    }
}

// Larger grids: the back substitution of one of the levels 5, 6, 7 (128 x 128 numbering) as a launch of its own, SOLL_WGS workgroups of four
// waves per member sharing the level's fronts (a member has 4 / 16 times the fronts of the 128 x 128 tree there: inside k_nd_solve's one
// workgroup per member they were 1.1 of its 1.9 ms at 256 x 256).
constexpr int SOLL_WGS = 4;
template <int LV>
__global__ __launch_bounds__(256, SOL_OCC) void k_nd_solve_level(FwdParams p, NdDev nd) {
    const int m = blockIdx.x % p.N, wg = blockIdx.x / p.N, tid = threadIdx.x;
    NdGeo g;
    g.lane = tid & 63;
    g.lc = g.lane & 15;
    g.lq = g.lane >> 4;
    const int w = wg * 4 + __builtin_amdgcn_readfirstlane(tid >> 6);
    double* P = p.P + (long long)m * p.Nxy;
    const double* fact = nd.fact + (long long)m * nd.fact_stride;
    constexpr int nf = 1 << (LV + LO);
    if (LV == 5) nd_solve_single<1, 6>(nd, fact, P, nf, w, g, 4 * SOLL_WGS);
    else if (LV == 6) nd_solve_single<2, 4>(nd, fact, P, nf, w, g, 4 * SOLL_WGS);
    else nd_solve_single<2, 3>(nd, fact, P, nf, w, g, 4 * SOLL_WGS);
}
#endif

// ------------------------------------------------------------------------------------------------------------------------
// Back substitution of levels 8..10: one wave per level-8 subtree (the workgroups of k_nd_sub), top-down.  Every factor row of the
// subtree's 7 fronts and every cell index is requested up front (one round trip to memory), the pressures live in an LDS plane
// over the subtree's region and its ring of separator cells (known from the levels above: one gather), and the region's 8 x 8
// pressures go to memory at the end.
// ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 4) void k_nd_solve_sub(FwdParams p, NdDev nd, int k) {
    __shared__ double xl_all[4][ND_CF_PLANE_SUB];
    const int m = blockIdx.x % p.N, bidx = blockIdx.x / p.N;
    const int tid = threadIdx.x;
    NdGeo g;
    g.lane = tid & 63;
    g.lc = g.lane & 15;
    g.lq = g.lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    double* xl = xl_all[w];
    double* P = p.P + (long long)m * p.Nxy;
    const double* fact = nd.fact + (long long)m * nd.fact_stride;
    const int i8 = 4 * bidx + w;
    constexpr int NFR = 3;  // the level-8 front and its two level-9 children (the leaves: k_nd_leaf_solve)
    constexpr int MB[NFR] = {2, 2, 2}, MK[NFR] = {2, 1, 1};  // most boundary tiles / pivot register rows per front
    // (round 6) the subtree's flat record: shapes and factor offsets of the three fronts as wave-uniform ints, boundary and pivot cells per
    // lane -- one read, and the factor tiles and the ring's pressures are requested right behind it (two dependent trips to memory per
    // wave instead of front record -> position table -> cell -> pressure: four)
    const int* H = nd.ssub + (long long)i8 * ND_SSUB_INTS;
    const int* TL = H + ND_SSUB_HDR + g.lane;
    int bt[NFR], kreg[NFR], cb[NFR][2], cpv[NFR][4];
    double t[NFR][2][4];
    const int x0 = __builtin_amdgcn_readfirstlane(H[9]), y0 = __builtin_amdgcn_readfirstlane(H[10]), ld = __builtin_amdgcn_readfirstlane(H[11]);
#pragma unroll
    for (int i = 0; i < NFR; ++i) {
        bt[i] = __builtin_amdgcn_readfirstlane(H[i]);
        kreg[i] = __builtin_amdgcn_readfirstlane(H[3 + i]);
        const double* fa = fact + __builtin_amdgcn_readfirstlane(H[6 + i]);
#pragma unroll
        for (int R = 0; R < MB[i]; ++R) {
            cb[i][R] = TL[(2 * i + R) * 64];
#pragma unroll
            for (int r = 0; r < MK[i]; ++r) t[i][R][r] = fa[((R < bt[i] ? R : 0) * kreg[i] + (r < kreg[i] ? r : 0)) * 64 + g.lane];
        }
#pragma unroll
        for (int r = 0; r < MK[i]; ++r) cpv[i][r] = TL[(i == 0 ? 6 + r : 7 + i) * 64];
    }
    auto li = [&](int c) { return ((c >> LG) - x0 + 1) * ld + ((c & (NB - 1)) - y0 + 1); };
    // the ring: the level-8 front's boundary cells, solved by the levels above
#pragma unroll
    for (int R = 0; R < 2; ++R) {
        const int c = cb[0][R];
        if (R < bt[0] && c >= 0) xl[li(c)] = P[c];
    }
    nd_wave_fence();
#pragma unroll
    for (int i = 0; i < NFR; ++i) {
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int R = 0; R < MB[i]; ++R) {
            const int c = cb[i][R];
            const double xv = (R < bt[i]) ? (c >= 0 ? xl[li(c >= 0 ? c : 0)] : (c == -2 ? -1.0 : 0.0)) : 0.0;
#pragma unroll
            for (int r = 0; r < MK[i]; ++r) acc[r] = fma(t[i][R][r], xv, acc[r]);
        }
#pragma unroll
        for (int r = 0; r < MK[i]; ++r) {
            double v = acc[r];
            v += __shfl_xor(v, 8);
            v += __shfl_xor(v, 4);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 1);
            if (g.lc == 0 && r < kreg[i] && cpv[i][r] >= 0) {
                xl[li(cpv[i][r])] = -v;
                P[cpv[i][r]] = -v;
            }
        }
        if (i == 0) nd_wave_fence();  // level boundary 8 | 9
    }
}

// Face fluxes from the pressures (fwd_dev.h), one workgroup per member.
__global__ __launch_bounds__(1024) void k_nd_flux(FwdParams p, int k) {
    const int m = blockIdx.x, Nx = p.Nx;
    double* Vx = p.Vx + (long long)m * (Nx + 1) * NB;
    double* Vy = p.Vy + (long long)m * Nx * (NB + 1);
    face_fluxes(p, p.P + (long long)m * p.Nxy, p.TX + (long long)m * (Nx + 1) * NB, p.TY + (long long)m * Nx * (NB + 1), Vx, Vy, threadIdx.x, 1024);
#if ND_LG > 7
    // A posteriori check of the direct solve (the larger grids only): the fluxes must reproduce the wells, max |div V - q| <= 1e-4 max |q|.
    // A gross check on purpose.  What a healthy solve leaves there is set by the member's largest transmissibility, not by the solver:
    // a flux T (p_c - p_nb) carries T eps |p| of rounding, 1e-7 for T = 1e9, and over 1024 members of BASELINE config 4's prior the residual
    // ranges from 1e-13 to 1e-5 for the nested dissection and the two-level CG alike (tests/tools/nd_residual_stats.py; normwise both are
    // at 1e-15 of ||A|| ||p||).  An elimination without pivoting BREAKS DOWN on the rare member whose permeability spans ten orders of magnitude
    // (cond(A) beyond 1 / eps: the diagonal of a strongly coupled cluster cancels to nothing): measured on a member with K = 0.1 ... 1.2e9,
    // the residual climbs 1e-5 -> 4e-3 -> 1e-2 over the three time steps before the first non-positive pivot shows.  Such a member -- this
    // check, or a non-positive pivot flagged by the elimination kernels -- is handed to the two-level CG for this time step by the host
    // (nd_check_and_fall_back).
    __shared__ double red[2][16];
    __syncthreads();
    const double* q = p.q + (long long)m * p.q_mstride + (long long)(p.q_cols > 1 ? k : 0) * p.Nxy;
    double worst = 0.0, qmax = 0.0;
    const double pin0 = (p.K[(long long)m * p.Nxy] + (p.Ky ? p.Ky : p.K)[(long long)m * p.Nxy]) * p.P[(long long)m * p.Nxy];
    for (int c = threadIdx.x; c < p.Nxy; c += 1024) {
        const int ix = c >> LG, iy = c & (NB - 1), fy = ix * (NB + 1) + iy;
        const double div = (Vx[c + NB] - Vx[c]) + (Vy[fy + 1] - Vy[fy]);
        // (cell 0 carries the SPD pin, A[0,0] += Kx[0] + Ky[0]: the system solved is div V + pin P[0] = q there -- without the term rates
        // with |sum q| > 1e-4 max |q|, which the reference accepts up to np.isclose, would flag every member)
        const double e = fabs(div - q[c] + (c == 0 ? pin0 : 0.0));
        worst = (e > worst || e != e) ? e : worst;  // (a NaN wins)
        qmax = fmax(qmax, fabs(q[c]));
    }
    for (int o = 32; o > 0; o >>= 1) {
        const double w2 = __shfl_xor(worst, o), q2 = __shfl_xor(qmax, o);
        worst = (w2 > worst || w2 != w2) ? w2 : worst;
        qmax = fmax(qmax, q2);
    }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = worst; red[1][threadIdx.x >> 6] = qmax; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) {
            worst = (red[0][w] > worst || red[0][w] != red[0][w]) ? red[0][w] : worst;
            qmax = fmax(qmax, red[1][w]);
        }
        if (!(worst <= 1e-4 * qmax)) atomicOr(&p.status[m], HM_MEMBER_BAD_PIVOT);
    }
#else
    (void)k;
#endif
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------------
// Host side: tables built once per plan, buffers sized by the builder.
// ------------------------------------------------------------------------------------------------------------------------
// the entry points of this object (one per grid size, fwd.h)
#if ND_LG == 7
#define ND_ENTRY(name) name
#elif ND_LG == 8
#define ND_ENTRY(name) name##256
#else
#define ND_ENTRY(name) name##512
#endif

bool ND_ENTRY(pressure_nd_applies)(const FwdParams& p) { return p.Nx == NB && p.Ny == NB; }

#if ND_LG == 7
void hm_nd_free(hm_nd* n) {
    if (!n) return;
    DevBuf* bufs[] = {&n->fronts, &n->cells, &n->cpos, &n->rec, &n->fact, &n->arena, &n->dg, &n->work, &n->cached, &n->wells, &n->vfac, &n->pimg, &n->wet, &n->todo, &n->leaft, &n->ssub, &n->leafu};
    for (DevBuf* b : bufs) hm_dev_free(*b);
    delete n;
}
#endif

// The flat per-leaf and per-subtree tables of the leaf kernels and k_nd_solve_sub (layout: the constants at the top of this file), from the
// symbolic tables: what the kernels of rounds 3-5 derived per lane from front record -> position table -> cell.
static void nd_build_flat_tables(const NdTablesHost& t, std::vector<int>& leaft, std::vector<int>& ssub) {
    leaft.assign((size_t)ND_LEAF_INTS * NLEAF, -1);
    for (int l = 0; l < NLEAF; ++l) {
        const int* F = &t.fronts[(size_t)(FID(10) + l) * ND_FRONT_INTS];
        const NdBox box = nd_box(F, NDF_RBOX);
        const int x0 = box.x0, y0 = box.y0, x1 = box.x1, y1 = box.y1, b = F[NDF_B];
        auto put = [&](int k, int v) { leaft[(size_t)k * NLEAF + l] = v; };
        put(0, x0); put(1, y0); put(2, x1 - x0); put(3, y1 - y0); put(4, b); put(5, F[NDF_UPD]);
        for (int j = 0; j < 12; ++j) {
            if (j >= b) { put(6 + 3 * j, 0); put(7 + 3 * j, -1); put(8 + 3 * j, -1); continue; }
            const int c = t.cells[F[NDF_CELLS] + 16 + j], bx = c >> LG, by = c & (NB - 1);
            int ix = bx, iy = by, fidx;  // the one region cell next to the boundary cell, and the face between them
            if (bx < x0) { ix = x0; fidx = CF_OX + ix * NB + iy; }                 // west side: the cell's west face
            else if (bx >= x1) { ix = x1 - 1; fidx = CF_OX + bx * NB + by; }      // east side: the boundary cell's west face
            else if (by < y0) { iy = y0; fidx = CF_OY + ix * (NB + 1) + iy; }      // south side: the cell's south face
            else { iy = y1 - 1; fidx = CF_OY + bx * (NB + 1) + by; }              // north side: the boundary cell's south face
            put(6 + 3 * j, (ix - x0) * 4 + (iy - y0)); put(7 + 3 * j, fidx); put(8 + 3 * j, c);
        }
    }
    ssub.assign((size_t)ND_SSUB_INTS * NF8, 0);
    for (int i8 = 0; i8 < NF8; ++i8) {
        int* H = &ssub[(size_t)i8 * ND_SSUB_INTS];
        const int fid[3] = {FID(8) + i8, FID(9) + 2 * i8, FID(9) + 1 + 2 * i8};
        const NdBox box = nd_box(&t.fronts[(size_t)fid[0] * ND_FRONT_INTS], NDF_RBOX);
        H[9] = box.x0; H[10] = box.y0; H[11] = box.y1 - box.y0 + 2;
        for (int i = 0; i < 3; ++i) {
            const int* F = &t.fronts[(size_t)fid[i] * ND_FRONT_INTS];
            const int* cl = &t.cells[F[NDF_CELLS]];
            const int bt = F[NDF_BT];
            H[i] = bt; H[3 + i] = F[NDF_KREG]; H[6 + i] = F[NDF_FACT];
            for (int lane = 0; lane < 64; ++lane) {
                const int lc = lane & 15, lq = lane >> 4;
                for (int R = 0; R < 2; ++R) H[ND_SSUB_HDR + (2 * i + R) * 64 + lane] = cl[16 * (1 + (R < bt ? R : 0)) + lc];
                for (int r = 0; r < (i == 0 ? 2 : 1); ++r) H[ND_SSUB_HDR + (i == 0 ? 6 + r : 7 + i) * 64 + lane] = cl[4 * r + lq];
            }
        }
    }
}

static int nd_setup(hm_fwd* f) {
    const FwdParams& p = f->p;
    NdTablesHost t;
    HM_REQUIRE(nd_build_tables(p.Nx, p.Ny, t), "nested-dissection tables: %s", t.error.c_str());
    HM_REQUIRE(t.info.lo == LO && t.info.levels == ND_LEVELS, "nested-dissection tables: %d levels for a kernel family of %d", t.info.levels, ND_LEVELS);
    const int* mbt = t.info.max_bt + LO;  // by the 128 x 128 tree's level numbers
    HM_REQUIRE(mbt[10] <= 1 && mbt[9] <= 2 && mbt[8] <= 2 && mbt[7] <= 3 && mbt[6] <= 4 && mbt[5] <= 6,
               "nested-dissection tables: a wave-level front is larger than the kernels' register arrays");
    for (int lv = 0; lv <= LO + 4; ++lv) {
        const int T = t.info.max_st[lv] + t.info.max_bt[lv];
        HM_REQUIRE(T <= TOP_MAXT, "nested-dissection tables: a top-level front has %d tile rows (max %d)", T, TOP_MAXT);
    }
    HM_REQUIRE(t.info.upd_doubles[LO + 10] <= 128 && mbt[10] == 1, "nested-dissection tables: a leaf has more than 12 boundary cells");
    for (int lv = LO + 5; lv < ND_LEVELS; ++lv) HM_REQUIRE(t.info.max_st[lv] == 1, "nested-dissection tables: level %d has several pivot tiles", lv);
    for (int fI = 0; fI < t.info.n_fronts; ++fI) {
        const int* F = &t.fronts[(size_t)fI * ND_FRONT_INTS];
        if (F[NDF_LEVEL] > LO + 4) continue;
        if (LO > 0) HM_REQUIRE(F[NDF_KREG] == 4, "nested-dissection tables: front %d of the top levels has a partial last pivot tile", fI);
        if (LO > 0 && F[NDF_LEVEL] <= LO + 2) continue;
        // the three instances of k_nd_top: level LO + 4 <3, 5, 10, ., 8 waves>, level LO + 3 of the larger grids <2, 6, 15, ., 16>, levels 3..0 at 128 x 128 <3, 4, 13, ., 16>
        const bool l4 = F[NDF_LEVEL] == LO + 4, l3big = LO > 0 && F[NDF_LEVEL] == LO + 3;
        const int top_nts = l4 ? 5 : l3big ? 6 : 4, top_nvs = l3big ? 2 : 3, top_maxt = l4 ? 10 : l3big ? 15 : 13, top_nw = l4 ? 8 : TOP_NW;
        HM_REQUIRE(F[NDF_ST] + F[NDF_BT] <= top_maxt, "nested-dissection tables: front %d has %d tile rows (k_nd_top takes %d)", fI, F[NDF_ST] + F[NDF_BT], top_maxt);
        const int st = F[NDF_ST], bt = F[NDF_BT], T = st + bt;
        const int nV = st * T - st * (st - 1) / 2, nT = F[NDF_B] > 0 ? bt * (bt + 1) / 2 : 0;
        HM_REQUIRE(nV <= top_nvs * top_nw && nT <= top_nts * top_nw, "nested-dissection tables: front %d has %d + %d tiles", fI, nV, nT);
    }
    // the LDS-DMA copies move 16-byte pieces: every update matrix starts on an even double of an even-strided, 16-byte aligned arena
    HM_REQUIRE(t.info.arena_doubles % 2 == 0, "nested dissection: odd arena stride %lld", (long long)t.info.arena_doubles);
    for (int f = 1; f < (int)(t.fronts.size() / ND_FRONT_INTS); ++f)
        HM_REQUIRE(t.fronts[f * ND_FRONT_INTS + NDF_UPD] < 0 || t.fronts[f * ND_FRONT_INTS + NDF_UPD] % 2 == 0,
                   "nested dissection: the update matrix of front %d starts on an odd double", f);
    hm_nd* n = new hm_nd();
    n->info = t.info;
    int rc = 0;
    // members per block: the whole ensemble where its factor, update matrices and panels fit the budget below (always at 128 x 128:
    // 9.6 MB a member), else blocks of that many members, one after the other through the same buffers
    const size_t leafu_doubles = (size_t)NF8 * 4 * t.info.upd_doubles[LO + 10];  // interleaved leaf updates (nd_plan.h: leafu)
    const size_t per_member = (size_t)(t.info.fact_doubles + t.info.arena_doubles + t.info.big_fact_doubles + t.info.pimg_doubles + CF_STRIDE + leafu_doubles) * 8;
    size_t cap = p.N;
    if (LO > 0) {
        size_t free_b = 0, total_b = 0;
        HM_HIP(hipMemGetInfo(&free_b, &total_b));
        // Blocks of members through the same buffers are no faster when larger (23.3 ms per 512 members at 512, 1024 or 2048 a block) --
        // but an ensemble that is ONE block keeps the results of its dry fronts from step to step (16.0 instead of 22.6 ms per 512 members
        // at 256 x 256).  So: the whole ensemble where its buffers fit the free memory with 16 GB to spare (BASELINE config 4 whole on one
        // GPU: 4096 x 58 MB = 238 GB of the 288), else blocks within 64 GB.
        const size_t whole = (size_t)p.N * per_member, spare = (size_t)16 << 30;
        const size_t budget = whole + spare <= free_b ? whole : std::min<size_t>(free_b / 2, (size_t)64 << 30);
        cap = std::max<size_t>(1, std::min<size_t>(p.N, budget / per_member));
        if (f->dbg_nd_cap > 0) cap = std::max<size_t>(1, std::min<size_t>(cap, (size_t)f->dbg_nd_cap));  // (hm_fwd_set_debug "nd_cap": tests, experiments)
    }
    n->cap = (int)cap;
    const size_t N = cap;
    const size_t n_cached = NCACHE;
    if ((rc = hm_dev_alloc(n->fronts, t.fronts.size() * 4)) || (rc = hm_dev_alloc(n->cells, t.cells.size() * 4)) ||
        (rc = hm_dev_alloc(n->cpos, t.cpos.size() * 2)) || (rc = hm_dev_alloc(n->rec, t.rec.size() * 2)) || (rc = hm_dev_alloc(n->fact, N * t.info.fact_doubles * 8)) ||
        (rc = hm_dev_alloc(n->arena, N * t.info.arena_doubles * 8)) || (rc = hm_dev_alloc(n->dg, N * (size_t)CF_STRIDE * 8)) ||
        (rc = hm_dev_alloc(n->leafu, N * leafu_doubles * 8)) || (rc = hm_dev_alloc(n->work, N * (size_t)ND_WORK_INTS * 4)) || (rc = hm_dev_alloc(n->cached, N * n_cached)) || (rc = hm_dev_alloc(n->wells, n_cached)) ||
        (LO > 0 && ((rc = hm_dev_alloc(n->vfac, N * (size_t)t.info.big_fact_doubles * 8)) || (rc = hm_dev_alloc(n->pimg, N * (size_t)t.info.pimg_doubles * 8)) ||
                    (rc = hm_dev_alloc(n->wet, N * (size_t)NB * WETW * 8)) || (rc = hm_dev_alloc(n->todo, N * (size_t)NTODO))))) {
        hm_nd_free(n);
        return rc;
    }
    HM_HIP(hipMemcpy(n->fronts.p, t.fronts.data(), t.fronts.size() * 4, hipMemcpyHostToDevice));
    HM_HIP(hipMemcpy(n->cells.p, t.cells.data(), t.cells.size() * 4, hipMemcpyHostToDevice));
    HM_HIP(hipMemcpy(n->cpos.p, t.cpos.data(), t.cpos.size() * 2, hipMemcpyHostToDevice));
    HM_HIP(hipMemcpy(n->rec.p, t.rec.data(), t.rec.size() * 2, hipMemcpyHostToDevice));
    {
        std::vector<int> leaft, ssub;
        nd_build_flat_tables(t, leaft, ssub);
        if ((rc = hm_dev_alloc(n->leaft, leaft.size() * 4)) || (rc = hm_dev_alloc(n->ssub, ssub.size() * 4))) {
            hm_nd_free(n);
            return rc;
        }
        HM_HIP(hipMemcpy(n->leaft.p, leaft.data(), leaft.size() * 4, hipMemcpyHostToDevice));
        HM_HIP(hipMemcpy(n->ssub.p, ssub.data(), ssub.size() * 4, hipMemcpyHostToDevice));
    }
    NdDev& d = n->dev;
    d.leafu = (double*)n->leafu.p;
    d.leafu_stride = (long long)leafu_doubles;
    d.leaft = (const int*)n->leaft.p;
    d.ssub = (const int*)n->ssub.p;
    d.fronts = (const int*)n->fronts.p;
    d.cells = (const int*)n->cells.p;
    d.cpos = (const short*)n->cpos.p;
    d.rec = (const short*)n->rec.p;
    d.fact = (double*)n->fact.p;
    d.arena = (double*)n->arena.p;
    d.cf = (double*)n->dg.p;
    d.work = (int*)n->work.p;
    d.cached = (unsigned char*)n->cached.p;
    d.wells = (const unsigned char*)n->wells.p;
    d.vfac = (double*)n->vfac.p;
    d.pimg = (double*)n->pimg.p;
    d.wet = (unsigned long long*)n->wet.p;
    d.todo = (unsigned char*)n->todo.p;
    d.vfac_stride = t.info.big_fact_doubles;
    d.pimg_stride = t.info.pimg_doubles;
    d.reuse = 1;
    {   // fronts with a well in their subtree: the region of a level-8 subtree, above that the separator and both children (+ 1 ring)
        std::vector<unsigned char> wf(NCACHE, 0);
        for (int fr = FID(9) - 1; fr >= 0; --fr) {
            if (fr < FID(8)) wf[fr] = wf[2 * fr + 1] | wf[2 * fr + 2];
            const NdBox box = nd_box(&t.fronts[fr * ND_FRONT_INTS], fr >= FID(8) ? NDF_RBOX : NDF_PBOX);
            const int x0 = box.x0 - 1, y0 = box.y0 - 1, x1 = box.x1 + 1, y1 = box.y1 + 1;
            for (int cell : f->well_cells_host) {
                const int ix = cell / NB, iy = cell % NB;
                if (ix >= x0 && ix < x1 && iy >= y0 && iy < y1) wf[fr] = 1;
            }
        }
        HM_HIP(hipMemcpy(n->wells.p, wf.data(), NCACHE, hipMemcpyHostToDevice));
    }
    HM_HIP(hipMemset(n->cached.p, 0, N * n_cached));
#if ND_LG > 7
    HM_HIP(hipFuncSetAttribute((const void*)k_ndl_plan, hipFuncAttributeMaxDynamicSharedMemorySize, NB * WETW * 8 + NCACHE));
#endif
    d.fact_stride = t.info.fact_doubles;
    d.arena_stride = t.info.arena_doubles;
    d.slot9 = t.info.upd_doubles[LO + 9];
    d.slot10 = t.info.upd_doubles[LO + 10];
    d.child_doubles[0] = t.info.upd_doubles[LO + 8];
    d.child_doubles[1] = t.info.upd_doubles[LO + 7];
    d.child_doubles[2] = t.info.upd_doubles[LO + 6];
    // dynamic LDS beyond 64 KB must be requested per kernel
    for (int f = LO == 0 ? 0 : (1 << (LO + 3)) - 1; f < (2 << (LO + 4)) - 1; ++f)  // (k_nd_top's fronts: their 1 KB table pieces)
        HM_REQUIRE(t.fronts[f * ND_FRONT_INTS + NDF_CELLS] + 256 <= (int)t.cells.size(), "nested dissection: the table piece of front %d runs past the tables", f);
    d.top_child_doubles = 0;  // (128 x 128: the largest child of a front of levels 3..0)
    for (int lv = 1; lv <= 4 && LO == 0; ++lv) d.top_child_doubles = std::max(d.top_child_doubles, t.info.upd_doubles[lv]);
    HM_HIP(hipFuncSetAttribute((const void*)k_nd_top<3, 5, 10, LO + 4, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#ifdef ND_EXP_TOP4_16
    HM_HIP(hipFuncSetAttribute((const void*)k_nd_top<3, 4, 10, LO + 4, 16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#endif
#if ND_LG > 7
    HM_HIP(hipFuncSetAttribute((const void*)k_nd_top<2, 6, 15, LO + 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#else
    HM_HIP(hipFuncSetAttribute((const void*)k_nd_top<3, 4, 13, 3, TOP_NW, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HM_HIP(hipFuncSetAttribute((const void*)k_nd_top<3, 4, 13, 3, TOP_NW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HM_HIP(hipFuncSetAttribute((const void*)k_nd_top<3, 4, 13, 2, TOP_NW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HM_HIP(hipFuncSetAttribute((const void*)k_nd_top<3, 4, 13, 1, TOP_NW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HM_HIP(hipFuncSetAttribute((const void*)k_nd_top<3, 4, 13, 0, TOP_NW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#endif
#if ND_LG == 7
    HM_HIP(hipFuncSetAttribute((const void*)k_nd_assemble<double>, hipFuncAttributeMaxDynamicSharedMemorySize, NB * NB * 8));
    HM_HIP(hipFuncSetAttribute((const void*)k_nd_assemble<float>, hipFuncAttributeMaxDynamicSharedMemorySize, NB * NB * 8));
#endif
    HM_HIP(hipFuncSetAttribute((const void*)k_nd_wave<5, 6, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HM_HIP(hipFuncSetAttribute((const void*)k_nd_wave<6, 4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HM_HIP(hipFuncSetAttribute((const void*)k_nd_wave<7, 3, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HM_HIP(hipFuncSetAttribute((const void*)k_nd_sub, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    f->nd = n;
    return 0;
}

// The launches of one pressure step for the members the parameter block `p` describes (the whole ensemble, or one block of it).
static int nd_launch_block(hm_fwd* f, const FwdParams& p, const void* S, long long S_stride, int k) {
    hipStream_t s = f->ctx->stream;
    NdDev& nd = f->nd->dev;
    const size_t lds_sub = (size_t)SUB_WPB * nd_sub_lds_doubles(nd) * 8;
#if ND_LG == 7
    const bool iso = p.Ky == nullptr;  // isotropic: assembly from an LDS copy of 1 / (mobility K), the plan in the same launch
    const size_t lds_asm = iso ? (size_t)(NB / 2 + 2) * NB * 8 : 0;  // (two passes of 64 rows + a row either side)
    if (f->dtype == 64) {
        hipLaunchKernelGGL(k_nd_assemble<double>, dim3(p.N), dim3(1024), lds_asm, s, p, nd, (const double*)S, S_stride, k);
        if (!iso) hipLaunchKernelGGL(k_nd_plan<double>, dim3(p.N), dim3(256), 0, s, p, nd, (const double*)S, S_stride);
    } else {
        hipLaunchKernelGGL(k_nd_assemble<float>, dim3(p.N), dim3(1024), lds_asm, s, p, nd, (const float*)S, S_stride, k);
        if (!iso) hipLaunchKernelGGL(k_nd_plan<float>, dim3(p.N), dim3(256), 0, s, p, nd, (const float*)S, S_stride);
    }
#else
    if (f->dtype == 64) hipLaunchKernelGGL(k_ndl_assemble<double>, dim3(p.N * p.Nx), dim3(NB), 0, s, p, nd, (const double*)S, S_stride, k);
    else hipLaunchKernelGGL(k_ndl_assemble<float>, dim3(p.N * p.Nx), dim3(NB), 0, s, p, nd, (const float*)S, S_stride, k);
    hipLaunchKernelGGL(k_ndl_plan, dim3(p.N), dim3(1024), (size_t)NB * WETW * 8 + NCACHE, s, p, nd);
#endif
    hipLaunchKernelGGL(k_nd_leaf, dim3(p.N * (4 << LO)), dim3(256), 0, s, p, nd, k);
    hipLaunchKernelGGL(k_nd_sub, dim3(((p.N + SUB_WPB - 1) / SUB_WPB) * NF8), dim3(64 * SUB_WPB), lds_sub, s, p, nd, k);
    // (levels LO + 5, LO + 6 as workgroups of ONE wave: a level-5 wave's LDS block is 31.5 KB -- five fit a CU, workgroups of two put four there --
    // and a wave that is done frees its block at once instead of when its workgroup's slowest is: 0.55 -> 0.52 and 0.43 -> 0.41 ms per 1000 members
    // at 128 x 128 (round 6); level 7 is faster as workgroups of four, 0.38 against 0.39: its waves share a front's records and recipes in L1)
    hipLaunchKernelGGL((k_nd_wave<7, 3, 4>), dim3(p.N * (NF7 / 4)), dim3(256), (size_t)4 * (ND_LDS_DATA + 2 * nd.child_doubles[0] + 4 * WAVE_CF_PLANE) * 8, s, p, nd, k);
    hipLaunchKernelGGL((k_nd_wave<6, 4, 1>), dim3(p.N * NF6), dim3(64), (size_t)1 * (ND_LDS_DATA + 2 * nd.child_doubles[1] + 4 * WAVE_CF_PLANE) * 8, s, p, nd, k);
    hipLaunchKernelGGL((k_nd_wave<5, 6, 1>), dim3(p.N * NF5), dim3(64), (size_t)1 * (ND_LDS_DATA + 2 * nd.child_doubles[2] + 4 * WAVE_CF_PLANE) * 8, s, p, nd, k);
    // level LO + 4: one front per workgroup of 8 waves, two workgroups a CU (images, 2 KB of tables, ONE child, records)
    const int chd4 = f->nd->info.upd_doubles[LO + 5];
    const size_t lds_top4 = (size_t)top_lds_doubles(10, chd4) * 8 + 31 * ND_FRONT_INTS * 4;
#ifdef ND_EXP_TOP4_16  // (A/B: the level as workgroups of 16 waves, one to a CU -- round 4's form)
    hipLaunchKernelGGL((k_nd_top<3, 4, 10, LO + 4, 16, true>), dim3(p.N << (LO + 4)), dim3(64 * 16), lds_top4 + 70000, s, p, nd, k, chd4);
#else
    hipLaunchKernelGGL((k_nd_top<3, 5, 10, LO + 4, 8, true>), dim3(p.N << (LO + 4)), dim3(64 * 8), lds_top4, s, p, nd, k, chd4);
#endif
    if (hipError_t e_ = hipGetLastError()) { hm_set_error("k_nd_top<level LO + 4> launch with %zu bytes of LDS: %s", lds_top4, hipGetErrorString(e_)); return 1; }
#if ND_LG == 7
    const size_t lds_top = (size_t)top_lds_doubles(13, nd.top_child_doubles) * 8 + 31 * ND_FRONT_INTS * 4;
    if (4 * p.N <= 3 * f->ctx->num_cu && f->dbg_top_per_level != 0) {
        // A SMALL member shard (at most three quarters as many members as CUs -- one rank's share of a strong-scaled ensemble; measured: 64
        // members 0.90 -> 0.77 ms a pressure step, 125 members 1.11 -> 1.03, 250 members no gain): a workgroup per member would leave CUs
        // idle while each member's 15 fronts of levels 3 .. 0 run one after the other (0.29 of the 1.13 ms pressure step at 125 members).  So a
        // launch per level, ONE FRONT PER WORKGROUP -- the larger grids' form of this kernel: 8, 4, 2, 1 fronts a member side by side.  Same
        // tiles, same products, same order of additions: the same bits (hm_fwd_set_debug "top_per_level" 0: the member-per-workgroup form).
        hipLaunchKernelGGL((k_nd_top<3, 4, 13, 3, TOP_NW, true>), dim3(p.N << 3), dim3(64 * TOP_NW), lds_top, s, p, nd, k, nd.top_child_doubles);
        hipLaunchKernelGGL((k_nd_top<3, 4, 13, 2, TOP_NW, true>), dim3(p.N << 2), dim3(64 * TOP_NW), lds_top, s, p, nd, k, nd.top_child_doubles);
        hipLaunchKernelGGL((k_nd_top<3, 4, 13, 1, TOP_NW, true>), dim3(p.N << 1), dim3(64 * TOP_NW), lds_top, s, p, nd, k, nd.top_child_doubles);
        hipLaunchKernelGGL((k_nd_top<3, 4, 13, 0, TOP_NW, true>), dim3(p.N), dim3(64 * TOP_NW), lds_top, s, p, nd, k, nd.top_child_doubles);
    } else {
        hipLaunchKernelGGL((k_nd_top<3, 4, 13, 3, TOP_NW, false>), dim3(p.N), dim3(64 * TOP_NW), lds_top, s, p, nd, k, nd.top_child_doubles);  // levels 3..0 of a member
    }
    if (hipError_t e_ = hipGetLastError()) { hm_set_error("k_nd_top launch with %zu bytes of LDS: %s", lds_top, hipGetErrorString(e_)); return 1; }
#endif
#if ND_LG > 7
    const NdInfo& I = f->nd->info;
    {   // level LO + 3: one front per workgroup as well, six trailing tiles a wave; its children (level LO + 4) staged one after the other
        const int chd = I.upd_doubles[LO + 4];
        const size_t lds3 = (size_t)top_lds_doubles(15, chd) * 8 + 31 * ND_FRONT_INTS * 4;
        hipLaunchKernelGGL((k_nd_top<2, 6, 15, LO + 3>), dim3(p.N << (LO + 3)), dim3(64 * TOP_NW), lds3, s, p, nd, k, chd);
        if (hipError_t e_ = hipGetLastError()) { hm_set_error("k_nd_top<level LO + 3> launch with %zu bytes of LDS: %s", lds3, hipGetErrorString(e_)); return 1; }
    }
    for (int lv = LO + 2; lv >= 0; --lv) {  // the big fronts, leaves-to-root; per level: pivot blocks in turn, then the update matrix
        const int nf = 1 << lv, st = I.max_st[lv], bt = I.max_bt[lv], T = st + bt;
        for (int g0 = 0; g0 * BIG_PB < st; ++g0) {
            hipLaunchKernelGGL(k_big_diag, dim3(p.N * nf), dim3(64), 0, s, p, nd, lv, g0);
            const int rows = T - std::min(st, (g0 + 1) * BIG_PB), wgs = (rows + 3) / 4;
            if (wgs > 0) hipLaunchKernelGGL(k_big_rows, dim3(big_grid(p.N, nf, wgs)), dim3(256), 0, s, p, nd, lv, g0, wgs);
        }
        if (lv > 0) {
            const int hb = (bt + 1) / 2, wgs = (hb * (hb + 1) / 2 + 3) / 4;
            hipLaunchKernelGGL(k_big_trail, dim3(big_grid(p.N, nf, wgs)), dim3(256), 0, s, p, nd, lv, wgs);
        }
    }
#endif
#if ND_LG == 7
    hipLaunchKernelGGL(k_nd_solve, dim3(p.N), dim3(64 * SOL_NW), 0, s, p, nd, k);
#else
    for (int lv = 0; lv <= LO + 4; ++lv) {  // root to leaves, a launch per level
        if (lv <= LO + 2) hipLaunchKernelGGL(k_nd_solve_front<4>, dim3(p.N << lv), dim3(256), 0, s, p, nd, lv);
        else hipLaunchKernelGGL(k_nd_solve_front<1>, dim3(p.N * (((1 << lv) + 3) / 4)), dim3(256), 0, s, p, nd, lv);
    }
    hipLaunchKernelGGL(k_nd_solve_level<5>, dim3(p.N * SOLL_WGS), dim3(256), 0, s, p, nd);
    hipLaunchKernelGGL(k_nd_solve_level<6>, dim3(p.N * SOLL_WGS), dim3(256), 0, s, p, nd);
    hipLaunchKernelGGL(k_nd_solve_level<7>, dim3(p.N * SOLL_WGS), dim3(256), 0, s, p, nd);
#endif
    hipLaunchKernelGGL(k_nd_solve_sub, dim3(p.N * (64 << LO)), dim3(256), 0, s, p, nd, k);
    hipLaunchKernelGGL(k_nd_leaf_solve, dim3(p.N * (4 << LO)), dim3(256), 0, s, p, nd, k);
#if ND_LG == 7
    if (f->dbg_lazy_flux && p.N == f->p.N) {  // (the whole ensemble in one block: always at 128 x 128)
        f->flux_pending = true;               // fwd.h: the fluxes are formed by whoever needs them
        HM_HIP(hipGetLastError());
        return 0;
    }
#endif
    hipLaunchKernelGGL(k_nd_flux, dim3(p.N), dim3(1024), 0, s, p, k);
    HM_HIP(hipGetLastError());
    return 0;
}

#if ND_LG == 7
int nd128_materialize_fluxes(hm_fwd* f) {
    if (!f->flux_pending) return 0;
    hipLaunchKernelGGL(k_nd_flux, dim3(f->p.N), dim3(1024), 0, f->ctx->stream, f->p, 0);
    HM_HIP(hipGetLastError());
    f->flux_pending = false;
    return 0;
}
#endif

#if ND_LG > 7
// The direct solver's safety net on the larger grids.  A member the elimination could not solve -- a non-positive pivot, or fluxes that
// miss the wells grossly (k_nd_flux) -- is solved again for this time step by the two-level conjugate-gradient solver
// (press_pcg.hip: it works on the matrix itself and stops on its residual), as a member block of one.  Members are independent, so nothing
// else is touched.  Costs one stream synchronisation and a read of the status words per time step (a step is ~25 ms of pressure solve and
// ~75 ms of sweep per 512 members); a flagged member costs a single-member CG solve (a few ms).  More than ND_MAX_FALLBACK flagged members
// in one step are a defect of the inputs (K <= 0, NaN), not of conditioning: their flags stay.
constexpr int ND_MAX_FALLBACK = 32;
static int nd_check_and_fall_back(hm_fwd* f, const void* S, long long S_stride, int k) {
    if (f->press_variant == 12) return 0;  // the fully asynchronous form: no host synchronisation per time step; a member the elimination
                                           // cannot solve keeps its HM_MEMBER_BAD_PIVOT flag (the behaviour of the 128 x 128 solver)
    const FwdParams p = f->p;
    hipStream_t s = f->ctx->stream;
    std::vector<int> st((size_t)p.N);
    HM_HIP(hipMemcpyAsync(st.data(), p.status, (size_t)p.N * 4, hipMemcpyDeviceToHost, s));
    HM_HIP(hipStreamSynchronize(s));
    std::vector<int> bad;
    for (int m = 0; m < p.N; ++m)
        if (st[m] & HM_MEMBER_BAD_PIVOT) bad.push_back(m);
    {   // (hm_fwd_set_debug "nd_force_fallback": a test makes one member take the hand-over every time step, whatever its solve was like)
        const int m = f->dbg_nd_force_fallback;
        if (m >= 0 && m < p.N && !(st[m] & HM_MEMBER_BAD_PIVOT)) bad.push_back(m);
    }
    if (bad.empty() || (int)bad.size() > ND_MAX_FALLBACK || !pressure_two_level_applies(p)) return 0;
    int rc = 0;
    for (int m : bad) {
        const long long o = m;
        const int cleared = st[m] & ~HM_MEMBER_BAD_PIVOT;
        HM_HIP(hipMemcpyAsync(p.status + o, &cleared, 4, hipMemcpyHostToDevice, s));
        HM_HIP(hipMemsetAsync(p.P + o * p.Nxy, 0, (size_t)p.Nxy * 8, s));  // the CG starts from the pressures it finds: not from a failed solve's
        HM_HIP(hipStreamSynchronize(s));                                    // (`cleared` leaves scope)
        FwdParams pb = p;
        pb.N = 1;
        pb.K = p.K + o * p.Nxy;
        if (p.Ky) pb.Ky = p.Ky + o * p.Nxy;
        pb.q = p.q + o * p.q_mstride;
        pb.TX = p.TX + o * (p.Nx + 1) * NB;
        pb.TY = p.TY + o * p.Nx * (NB + 1);
        pb.P = p.P + o * p.Nxy;
        pb.Vx = p.Vx + o * (p.Nx + 1) * NB;
        pb.Vy = p.Vy + o * p.Nx * (NB + 1);
        pb.status = p.status + o;
        pb.n_cg = p.n_cg + o * p.nTime;
        f->p = pb;  // (launch_pressure_two_level reads the plan's parameter block)
        // (the inner plan of an embedded grid, forward.hip: the padding's empty rows would make the two-level method's coarse matrix singular)
        rc = f->is_inner ? launch_pressure_pcg(f, (const char*)S + (size_t)(o * S_stride) * f->esz, S_stride, k)
                         : launch_pressure_two_level(f, (const char*)S + (size_t)(o * S_stride) * f->esz, S_stride, k);
        f->p = p;
        if (rc) return rc;
        f->nd_fallbacks++;
    }
    return 0;
}
#endif

// Tables and buffers of the plan, if they are not there yet: hm_fwd_run calls this before it starts the clock of a run (building the
// tables and allocating the factor / update / panel buffers of a large ensemble takes seconds, once per plan).
int ND_ENTRY(prepare_pressure_nd)(hm_fwd* f) {
    if (!ND_ENTRY(pressure_nd_applies)(f->p) || f->nd) return 0;
    return nd_setup(f);
}

// Returns 0 if launched, >0 on error, -1 if this specialisation does not apply.
int ND_ENTRY(launch_pressure_nd)(hm_fwd* f, const void* S, long long S_stride, int k) {
    const FwdParams& p = f->p;
    if (!ND_ENTRY(pressure_nd_applies)(p)) return -1;
    if (!f->nd) {
        int rc = nd_setup(f);
        if (rc) return rc;
    }
    hipStream_t s = f->ctx->stream;
    NdDev& nd = f->nd->dev;
    // results kept from earlier time steps are only good for the inputs they were computed from; press_variant 14: no reuse at all; per-member
    // wells: none either (the well flags are per plan); an ensemble solved in several member blocks through the same buffers keeps nothing
    nd.reuse = f->press_variant != 14 && p.q_mstride == 0 && !f->raw_field_exposed && p.N <= f->nd->cap;
    nd.top_deal = f->dbg_top_deal;
    if (f->nd->cached_gen != f->inputs_gen) {
        HM_HIP(hipMemsetAsync(f->nd->cached.p, 0, (size_t)std::min(p.N, f->nd->cap) * NCACHE, s));
        f->nd->cached_gen = f->inputs_gen;
    }
    {   // the right-hand side rows of fronts with wells in their subtree: kept while the rates stay what they were
        const int ep = p.q_cols > 1 ? f->q_epoch[k] : 0;
        nd.wells_ok = ep == f->nd->cached_q_epoch;
        f->nd->cached_q_epoch = ep;
    }
    const int cap = f->nd->cap;
#if ND_LG > 7
    if (p.N <= cap) {
        if (int rc = nd_launch_block(f, p, S, S_stride, k)) return rc;
        return nd_check_and_fall_back(f, S, S_stride, k);
    }
#else
    if (p.N <= cap) return nd_launch_block(f, p, S, S_stride, k);
#endif
    for (int m0 = 0; m0 < p.N; m0 += cap) {  // blocks of members through the same factor / update / panel buffers, one after the other
        FwdParams pb = p;
        const long long o = m0;
        pb.N = std::min(cap, p.N - m0);
        pb.K = p.K + o * p.Nxy;
        if (p.Ky) pb.Ky = p.Ky + o * p.Nxy;
        pb.q = p.q + o * p.q_mstride;
        pb.TX = p.TX + o * (p.Nx + 1) * NB;
        pb.TY = p.TY + o * p.Nx * (NB + 1);
        pb.P = p.P + o * p.Nxy;
        pb.Vx = p.Vx + o * (p.Nx + 1) * NB;
        pb.Vy = p.Vy + o * p.Nx * (NB + 1);
        pb.status = p.status + o;
        const char* Sb = (const char*)S + (size_t)(o * S_stride) * f->esz;
        if (int rc = nd_launch_block(f, pb, Sb, S_stride, k)) return rc;
    }
#if ND_LG > 7
    return nd_check_and_fall_back(f, S, S_stride, k);
#else
    return 0;
#endif
}

#ifdef HM_ND_PROF
extern "C" int hm_debug_nd_prof(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(hm_nd_prof_buf), sizeof(long long) * 64); }
#endif

#if ND_LG == 7
// The symbolic phase on the host (no device needed).  info (64 entries): [0] fronts, [1] table entries, [2] factor doubles, [3] arena doubles,
// [4..6] largest update of levels 8..10 (128 x 128 numbering), [7] recipe blocks, [19] ints per front record, [20] levels, [21] LO,
// [22] panel-image doubles, [23] pivot-image doubles, [24 + level] max_bt * 64 + max_st; the first 11 levels also at [8 + level] as max_bt * 16 + max_st
// (the 128 x 128 form of the call).
extern "C" int hm_debug_nd_tables(int Nx, int Ny, long long* info, int* fronts, int* cells, short* cpos, short* rec) {
    HM_REQUIRE(info, "hm_debug_nd_tables: NULL info");
    NdTablesHost t;
    if (!nd_build_tables(Nx, Ny, t)) {
        hm_set_error("hm_debug_nd_tables: %s (%d x %d)", t.error.c_str(), Nx, Ny);
        return 2;
    }
    const int lo = t.info.lo;
    info[0] = t.info.n_fronts;
    info[1] = t.info.n_cells;
    info[2] = t.info.fact_doubles;
    info[3] = t.info.arena_doubles;
    for (int i = 0; i < 3; ++i) info[4 + i] = t.info.upd_doubles[lo + 8 + i];
    if (lo == 0)
        for (int i = 0; i < 11; ++i) info[8 + i] = t.info.max_bt[i] * 16 + t.info.max_st[i];
    if (fronts) memcpy(fronts, t.fronts.data(), t.fronts.size() * sizeof(int));
    if (cells) memcpy(cells, t.cells.data(), t.cells.size() * sizeof(int));
    info[7] = t.info.n_rec_blocks;
    info[19] = ND_FRONT_INTS;
    info[20] = t.info.levels;
    info[21] = lo;
    info[22] = t.info.big_fact_doubles;
    info[23] = t.info.pimg_doubles;
    for (int i = 0; i < t.info.levels; ++i) info[24 + i] = t.info.max_bt[i] * 64 + t.info.max_st[i];
    if (cpos) memcpy(cpos, t.cpos.data(), t.cpos.size() * sizeof(short));
    if (rec) memcpy(rec, t.rec.data(), t.rec.size() * sizeof(short));
    return 0;
}
#endif
