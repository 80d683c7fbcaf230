// sat128e.hip -- 128x128 fp64 explicit upwind saturation sweep, EDGE-EXCHANGE form.
//
// Same decomposition as sat128.hip (one workgroup of 512 threads = one ensemble member on one CU for all ~615 explicit sub-steps,
// every thread an 8 (ix) x 4 (iy) patch of cells, lanes = consecutive patches along iy), different data flow:
//   * the fractional flow fw(S) of a patch never leaves its thread except for the two EDGE rows (patch rows 0 and 7), which the
//     ix-neighbours need: those go through LDS (2 x 32 KB, double-buffered by sub-step parity -> ONE workgroup barrier per
//     sub-step instead of two, 2 rows written + 4 read per thread instead of 8 + 24).  Inside the patch the sweep walks the rows
//     with a three-row register window (fw of row i-1, i, i+1); fw of row i+1 is evaluated while row i is updated, so the
//     division chains of one row interleave with the multiply-add chains of the other;
//   * the LDS that the full fw image occupied (128 KB) now holds 20 of a thread's 68 face fluxes (80 KB): the register state is
//     S (32 doubles) + 48 fluxes, which leaves room for the window without scratch spills;
//   * wells: the owner thread's register copy of the well cell is kept EXACT -- the row that contains a well re-evaluates that
//     one cell with the well's own coefficients (source terms included, from a small LDS record) under a wave-uniform branch that
//     only waves with a well in that row take.  No fw patching, no side state.
// Arithmetic per cell is exactly the generic kernel's (oracle/ressim.py:saturation_step_upwind):
//     S_c <- S_c + (((((cE fE + cN fN) + cC fC) + cS fS) + cW fW) + fi_c dtx)      (E,N,C,S,W = CSR order)
// with the upwind coefficients re-derived from the face fluxes every sub-step; results are bit-identical.
// Dry waves (band of 16 x 128 cells all zero, no injector, both halo rows zero) skip the sub-step under a wave-uniform branch; their
// edge rows in both parity buffers hold the zeros written before the loop.
//
// Compiled with -ffp-contract=off (no FMA contraction: every product and sum is rounded separately, as NumPy does).
#include "fracflow.h"

namespace {

constexpr int N128 = 128;
constexpr int PX = 8, PY = 4;
constexpr int NPY = N128 / PY;             // 32 patches along iy = 32 lanes
constexpr int NPX = N128 / PX;             // 16 patches along ix
constexpr int NT = NPX * NPY;              // 512 threads
constexpr int RX = 6;                      // flux rows 0..RX-1 of Vx and Vy live in registers, the rest in LDS
constexpr int EDGE_PAR = NPX * 2 * 1024;   // one parity: 16 patches x {row 0, row 7} x 1 KB
constexpr int EDGE_BYTES = 2 * EDGE_PAR;   // 64 KB
constexpr int REC_BYTES = 64;              // well record: -, cE, cN, cC, cS, cW, fid, -
constexpr int MAX_WELLS = 16;
constexpr int REC_BASE = EDGE_BYTES;
constexpr int VSP_BASE = REC_BASE + (MAX_WELLS + 1) * REC_BYTES;
constexpr int NCH = ((PX + 1 - RX) + (PX - RX)) * 2;  // 16-byte chunks per thread: Vx rows RX..8, Vy rows RX..7
constexpr int VSP_BYTES = NCH * NT * 16;   // 80 KB

__device__ __forceinline__ double from_next_lane(double v) {  // value of lane+1 (whole-wave shift; out-of-wave source reads 0)
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, true);  // wave_shl:1
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double from_prev_lane(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true);  // wave_shr:1
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

template <bool FD>
__global__ __launch_bounds__(NT) void k_sat128e(FwdParams p, const double* __restrict__ Sin_base,
                                                double* __restrict__ Sout_base, long long S_stride,
                                                double* __restrict__ prods, int k) {
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x;
    const int m = blockIdx.x;
    const int py = tid & (NPY - 1), px = tid >> 5;
    const int ix0 = px * PX, iy0 = py * PY;

    const double* Sin = Sin_base + (long long)m * S_stride;
    double* Sout = Sout_base + (long long)m * S_stride;
    const double* gVx = p.Vx + (long long)m * (N128 + 1) * N128;
    const double* gVy = p.Vy + (long long)m * N128 * (N128 + 1);
    const double* q = p.q + (long long)(p.q_cols > 1 ? k : 0) * p.Nxy;

    // ---------------- member state -> registers / LDS
    double S[PX][PY], Vx[RX][PY], Vy[RX][PY];
    double2* vsp = reinterpret_cast<double2*>(lds + VSP_BASE) + tid;  // chunk c at vsp[c * NT]
#pragma unroll
    for (int i = 0; i < PX; ++i)
#pragma unroll
        for (int j = 0; j < PY; j += 2) {
            double2 v = *reinterpret_cast<const double2*>(Sin + (ix0 + i) * N128 + iy0 + j);
            S[i][j] = v.x;
            S[i][j + 1] = v.y;
        }
#pragma unroll
    for (int i = 0; i < RX; ++i)
#pragma unroll
        for (int j = 0; j < PY; j += 2) {
            double2 v = *reinterpret_cast<const double2*>(gVx + (ix0 + i) * N128 + iy0 + j);
            Vx[i][j] = v.x;
            Vx[i][j + 1] = v.y;
        }
#pragma unroll
    for (int i = 0; i < RX; ++i)
#pragma unroll
        for (int j = 0; j < PY; ++j) Vy[i][j] = gVy[(ix0 + i) * (N128 + 1) + iy0 + j];
    // chunks: Vx row r (RX <= r <= 8) -> 2 (r - RX) + {0, 1};  Vy row r (RX <= r <= 7) -> 2 (9 - RX) + 2 (r - RX) + {0, 1}
    constexpr int CVY = 2 * (PX + 1 - RX);
#pragma unroll
    for (int r = RX; r <= PX; ++r)
#pragma unroll
        for (int h = 0; h < 2; ++h)
            vsp[(2 * (r - RX) + h) * NT] = *reinterpret_cast<const double2*>(gVx + (ix0 + r) * N128 + iy0 + 2 * h);
#pragma unroll
    for (int r = RX; r < PX; ++r)
#pragma unroll
        for (int h = 0; h < 2; ++h)
            vsp[(CVY + 2 * (r - RX) + h) * NT] =
                make_double2(gVy[(ix0 + r) * (N128 + 1) + iy0 + 2 * h], gVy[(ix0 + r) * (N128 + 1) + iy0 + 2 * h + 1]);

    // ---------------- the (at most one) well of this patch
    int wcell = -1, wrec = REC_BASE + MAX_WELLS * REC_BYTES;  // non-owners read a shared all-zero record
    double wq = 0.0;
    const int nW = min(p.nInj + p.nPrd, MAX_WELLS);
    for (int w = 0; w < nW; ++w) {
        int cell = p.well_cells[w];
        if (((cell >> 7) >> 3) == px && ((cell & 127) >> 2) == py && q[cell] != 0.0) {
            wcell = cell;
            wq = q[cell];
            wrec = REC_BASE + w * REC_BYTES;
        }
    }
    const bool has_well = wcell >= 0;
    const int wrow = has_well ? (wcell >> 7) - ix0 : -1, wcol = has_well ? (wcell & 127) - iy0 : -1;
    unsigned wellrows = 0;  // wave-uniform: patch rows in which some lane of this wave owns a well
#pragma unroll
    for (int i = 0; i < PX; ++i) wellrows |= __ballot(wrow == i) != 0ull ? 1u << i : 0u;
    wellrows = __builtin_amdgcn_readfirstlane(wellrows);

    // ---------------- CFL: pm = min over cells of pv / (Vi + fi)          
    const double pv = p.h2 * 1.0;
    double lmin = INFINITY;
#pragma unroll
    for (int i = 0; i < PX; ++i) {
        const double vy0 = i < RX ? Vy[i < RX ? i : 0][0] : gVy[(ix0 + i) * (N128 + 1) + iy0];
        const double vyn3 = from_next_lane(vy0);  // north face of column 3 (0 on the domain boundary)
#pragma unroll
        for (int j = 0; j < PY; ++j) {
            const double vxw = i < RX ? Vx[i < RX ? i : 0][j] : gVx[(ix0 + i) * N128 + iy0 + j];
            const double vxe = i + 1 < RX ? Vx[i + 1 < RX ? i + 1 : 0][j] : gVx[(ix0 + i + 1) * N128 + iy0 + j];
            const double vys = i < RX ? Vy[i < RX ? i : 0][j] : gVy[(ix0 + i) * (N128 + 1) + iy0 + j];
            const double vyn = j + 1 < PY ? (i < RX ? Vy[i < RX ? i : 0][j + 1 < PY ? j + 1 : 0] : gVy[(ix0 + i) * (N128 + 1) + iy0 + j + 1]) : vyn3;
            double xp = fmax(vxw, 0.0), yp = fmax(vys, 0.0);
            double xn = fmin(vxe, 0.0), yn = fmin(vyn, 0.0);
            double Vi = xp + yp - xn - yn;
            lmin = fmin(lmin, pv / (Vi + 0.0));  // fi = 0 for every cell without an injector
        }
    }
    double wVxW = 0, wVxE = 0, wVyS = 0, wVyN = 0;
    if (has_well) {
        const int wix = wcell >> 7, wiy = wcell & 127;
        wVxW = gVx[wix * N128 + wiy];
        wVxE = gVx[(wix + 1) * N128 + wiy];
        wVyS = gVy[wix * (N128 + 1) + wiy];
        wVyN = gVy[wix * (N128 + 1) + wiy + 1];
        double Vi = fmax(wVxW, 0.0) + fmax(wVyS, 0.0) - fmin(wVxE, 0.0) - fmin(wVyN, 0.0);
        lmin = fmin(lmin, pv / (Vi + fmax(wq, 0.0)));
    }
    double* red = reinterpret_cast<double*>(lds);  // the edge buffers are not in use yet
    red[tid] = lmin;
    __syncthreads();
    for (int s = NT / 2; s > 0; s >>= 1) {
        if (tid < s) red[tid] = fmin(red[tid], red[tid + s]);
        __syncthreads();
    }
    const double pm = red[0];
    __syncthreads();
    const double sat = p.swc + p.sor;
    const double cfl = ((1.0 - sat) / 3.0) * pm;
    const double ntsd = ceil(p.dt / cfl);
    const bool bad = !(ntsd >= 1.0 && ntsd <= 1.0e7);
    const int Nts = bad ? 0 : (int)ntsd;
    if (tid == 0) {
        p.nts[(long long)m * p.nTime + k] = Nts;
        if (bad) atomicOr(&p.status[m], HM_MEMBER_BAD_CFL);
    }
    const double d = bad ? 0.0 : (p.dt / (double)Nts) / pv;

    // well record: the cell's coefficients including the source terms (the shared record of the non-owners is all zeros)
    if (tid < 8) reinterpret_cast<double*>(lds + REC_BASE + MAX_WELLS * REC_BYTES)[tid] = 0.0;
    if (has_well) {
        double* rec = reinterpret_cast<double*>(lds + wrec);
        double fpq = fmin(wq, 0.0), fiq = fmax(wq, 0.0);
        double x1 = fmin(wVxW, 0.0), x2 = fmax(wVxE, 0.0), y1 = fmin(wVyS, 0.0), y2 = fmax(wVyN, 0.0);
        rec[1] = d * (-fmin(wVxE, 0.0));             // cE
        rec[2] = d * (-fmin(wVyN, 0.0));             // cN
        rec[3] = d * (fpq + x1 - x2 + y1 - y2);      // cC
        rec[4] = d * fmax(wVyS, 0.0);                // cS
        rec[5] = d * fmax(wVxW, 0.0);                // cW
        rec[6] = fiq * d;                            // fid
    }

    // ---------------- edge rows: byte offsets inside one parity buffer
    const int swz = (py >> 3) & 1;
    const int seg = py * 32;  // this thread's 32-byte segment of a 1 KB row: two 16-byte chunks, order flipped for every other
                              // group of 8 lanes -> the b128 accesses of a row are bank-conflict-free
    const int own0 = (px * 2 + 0) * 1024 + seg, own7 = (px * 2 + 1) * 1024 + seg;
    const int haloW = px > 0 ? ((px - 1) * 2 + 1) * 1024 + seg : own0;        // domain boundary: the coefficient is 0, any
    const int haloE = px + 1 < NPX ? ((px + 1) * 2 + 0) * 1024 + seg : own7;  // finite value will do
    auto load_row = [&](const char* base, double (&f)[PY]) {
#ifdef HM_ABL_NOEDGE
        f[0] = S[0][0]; f[1] = S[0][1]; f[2] = S[0][2]; f[3] = S[0][3];
        return;
#endif
        double2 a = *reinterpret_cast<const double2*>(base + (swz * 16));
        double2 b = *reinterpret_cast<const double2*>(base + ((1 ^ swz) * 16));
        f[0] = a.x; f[1] = a.y; f[2] = b.x; f[3] = b.y;
    };
    auto store_row = [&](char* base, const double (&f)[PY]) {
#ifdef HM_ABL_NOEDGE
        if (f[0] != 12345.0) return;
#endif
        *reinterpret_cast<double2*>(base + (swz * 16)) = make_double2(f[0], f[1]);
        *reinterpret_cast<double2*>(base + ((1 ^ swz) * 16)) = make_double2(f[2], f[3]);
    };
    auto fw_row = [&](const double (&s)[PY], double (&f)[PY]) {
#pragma unroll
        for (int j = 0; j < PY; ++j) f[j] = frac_flow<FD>(p, s[j]);
    };
    {   // both parities start from fw of the initial state (a dry wave never writes its edge rows again)
        double e[PY];
        fw_row(S[0], e);
        store_row(lds + own0, e);
        store_row(lds + EDGE_PAR + own0, e);
        fw_row(S[PX - 1], e);
        store_row(lds + own7, e);
        store_row(lds + EDGE_PAR + own7, e);
    }

    // wave-uniform: every cell of the band is exactly zero and no lane owns an injector (a producer in a dry band sits at S = 0 and
    // stays there: its row is re-evaluated with fw = 0 everywhere)
    int dry;
    {
        unsigned long long bits = 0ull;
#pragma unroll
        for (int i = 0; i < PX; ++i)
#pragma unroll
            for (int j = 0; j < PY; ++j) bits |= (unsigned long long)__double_as_longlong(S[i][j]) << 1;  // -0.0 counts as zero
        dry = __ballot(bits != 0ull || (has_well && wq > 0.0)) == 0ull;
    }

    // ---------------- explicit sub-steps
    for (int it = 0; it < Nts; ++it) {
        // The upwind coefficients are pure functions of (Vx, Vy, d): left alone, the compiler hoists them out of the sub-step loop
        // and spills.  Routing the constants d and 0.0 through an empty asm makes every coefficient depend on a per-iteration
        // opaque value (no instruction is emitted).
        double dd = d, z = 0.0;
        asm volatile("" : "+v"(dd), "+v"(z));
        const char* eb = lds + (it & 1) * EDGE_PAR;        // edge rows of the current state
        char* en = lds + ((it & 1) ^ 1) * EDGE_PAR;        // edge rows of the state this sub-step produces
#ifndef HM_ABL_NOBAR
        __syncthreads();
#endif

        if (dry) {  // the band only changes if a fractional flow just outside it is non-zero: its west / east halo rows
            const unsigned long long* hw = reinterpret_cast<const unsigned long long*>(eb + haloW);
            const unsigned long long* he = reinterpret_cast<const unsigned long long*>(eb + haloE);
            const unsigned long long o = (hw[0] | hw[1]) | (hw[2] | hw[3]) | (he[0] | he[1]) | (he[2] | he[3]);  // fw >= +0: bit test
            if (__ballot(o != 0ull) == 0ull) continue;
            dry = 0;  // water at the border: the band is wet from now on
        }

        double fprev[PY], fcur[PY], fnext[PY];
        load_row(eb + haloW, fprev);
        load_row(eb + own0, fcur);
#pragma unroll
        for (int i = 0; i < PX; ++i) {
            // fw of the next row: evaluated from the not yet updated S for the inner rows, own edge row / east halo from LDS
            if (i + 1 == PX - 1) load_row(eb + own7, fnext);
            else if (i + 1 == PX) load_row(eb + haloE, fnext);
            const double fS = from_prev_lane(fcur[PY - 1]);  // f(ix, iy0-1): its coefficient is 0 on the boundary
            const double fN = from_next_lane(fcur[0]);       // f(ix, iy0+PY)
            double vxw[PY], vxe[PY], vys[PY];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (i < RX) { vxw[2 * h] = Vx[i < RX ? i : 0][2 * h]; vxw[2 * h + 1] = Vx[i < RX ? i : 0][2 * h + 1]; }
                else { double2 v = vsp[(2 * (i - RX) + h) * NT]; vxw[2 * h] = v.x; vxw[2 * h + 1] = v.y; }
                if (i + 1 < RX) { vxe[2 * h] = Vx[i + 1 < RX ? i + 1 : 0][2 * h]; vxe[2 * h + 1] = Vx[i + 1 < RX ? i + 1 : 0][2 * h + 1]; }
                else { double2 v = vsp[(2 * (i + 1 - RX) + h) * NT]; vxe[2 * h] = v.x; vxe[2 * h + 1] = v.y; }
                if (i < RX) { vys[2 * h] = Vy[i < RX ? i : 0][2 * h]; vys[2 * h + 1] = Vy[i < RX ? i : 0][2 * h + 1]; }
                else { double2 v = vsp[(CVY + 2 * (i - RX) + h) * NT]; vys[2 * h] = v.x; vys[2 * h + 1] = v.y; }
            }
            const double vyn3 = from_next_lane(vys[0]);
            // two cells at a time: two division chains of row i+1 next to two multiply-add chains of row i (four and four need
            // more registers than there are)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (i + 1 < PX - 1) {
                    fnext[2 * h] = frac_flow<FD>(p, S[i + 1 < PX ? i + 1 : 0][2 * h]);
                    fnext[2 * h + 1] = frac_flow<FD>(p, S[i + 1 < PX ? i + 1 : 0][2 * h + 1]);
                }
                double acc[2];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int j = 2 * h + jj;
                    const double vyn = j + 1 < PY ? vys[j + 1 < PY ? j + 1 : 0] : vyn3;
                    const double x1 = fmin(vxw[j], z), x2 = fmax(vxe[j], z), y1 = fmin(vys[j], z), y2 = fmax(vyn, z);
                    const double cC = dd * (x1 - x2 + y1 - y2);
                    const double cW = dd * fmax(vxw[j], z);
                    const double cE = dd * (-fmin(vxe[j], z));
                    const double cS = dd * fmax(vys[j], z);
                    const double cN = dd * (-fmin(vyn, z));
                    const double fs = j > 0 ? fcur[j > 0 ? j - 1 : 0] : fS;
                    const double fn = j + 1 < PY ? fcur[j + 1 < PY ? j + 1 : 0] : fN;
                    double a = cE * fnext[j];
                    a = a + cN * fn;
                    a = a + cC * fcur[j];
                    a = a + cS * fs;
                    a = a + cW * fprev[j];
                    acc[jj] = a;
                }
                if (wellrows & (1u << i)) {  // some lane of this wave owns a well in this row: that one cell with its own coefficients
                    const double* rec = reinterpret_cast<const double*>(lds + wrec);
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int j = 2 * h + jj;
                        const double fs = j > 0 ? fcur[j > 0 ? j - 1 : 0] : fS;
                        const double fn = j + 1 < PY ? fcur[j + 1 < PY ? j + 1 : 0] : fN;
                        double a = rec[1] * fnext[j];
                        a = a + rec[2] * fn;
                        a = a + rec[3] * fcur[j];
                        a = a + rec[4] * fs;
                        a = a + rec[5] * fprev[j];
                        a = a + rec[6];
                        acc[jj] = (wrow == i && wcol == j) ? a : acc[jj];
                    }
                }
                S[i][2 * h] = S[i][2 * h] + acc[0];
                S[i][2 * h + 1] = S[i][2 * h + 1] + acc[1];
                __builtin_amdgcn_sched_barrier(0);
            }
            if (i == 0 || i == PX - 1) {  // the new state's edge rows for the neighbours (and for this thread's next sub-step)
                double e[PY];
                fw_row(S[i], e);
                store_row(en + (i == 0 ? own0 : own7), e);
            }
#pragma unroll
            for (int j = 0; j < PY; ++j) {
                fprev[j] = fcur[j];
                fcur[j] = fnext[j];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---------------- write back
    int nonfinite = 0;
#pragma unroll
    for (int i = 0; i < PX; ++i)
#pragma unroll
        for (int j = 0; j < PY; j += 2) {
            double2 v;
            v.x = S[i][j];
            v.y = S[i][j + 1];
            *reinterpret_cast<double2*>(Sout + (ix0 + i) * N128 + iy0 + j) = v;
            nonfinite |= !isfinite(v.x) || !isfinite(v.y);
        }
    if (nonfinite) atomicOr(&p.status[m], HM_MEMBER_NONFINITE);
    __syncthreads();
    if (tid < p.nPrd) prods[((long long)m * p.nTime + k) * p.nPrd + tid] = Sout[p.prd_ind[tid]];
}

template <bool FD>
int launch(hm_fwd* f, const void* S_in, void* S_out, long long S_stride, int k) {
    const size_t lds = (size_t)VSP_BASE + VSP_BYTES;
    auto kern = k_sat128e<FD>;
    HM_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(f->p.N), dim3(NT), lds, f->ctx->stream, f->p, (const double*)S_in, (double*)S_out,
                       S_stride, (double*)f->prods.p, k);
    HM_HIP(hipGetLastError());
    return 0;
}

}  // namespace

// Returns 0 if launched, >0 on error, -1 if this specialisation does not apply.
int launch_saturation_128e(hm_fwd* f, const void* S_in, void* S_out, long long S_stride, int k) {
    const FwdParams& p = f->p;
    if (p.q_mstride != 0) return -1;  // per-member wells: the well rows are found from one shared well list
    if (p.Nx != N128 || p.Ny != N128 || f->dtype != 64 || p.por != nullptr) return -1;
    if ((int)f->well_cells_host.size() > MAX_WELLS) return -1;
    std::vector<int> seen;  // at most one well per 8x4 patch
    for (int cell : f->well_cells_host) {
        int id = ((cell >> 7) >> 3) * 1000 + ((cell & 127) >> 2);
        for (int s : seen)
            if (s == id) return -1;
        seen.push_back(id);
    }
    return p.fluid_default ? launch<true>(f, S_in, S_out, S_stride, k) : launch<false>(f, S_in, S_out, S_stride, k);
}
