// sat128.hip -- 128x128 fp64 specialisation of the explicit upwind saturation sweep.
//
// One workgroup (512 threads = 8 waves) = one ensemble member, resident on one CU for all Nts (~615) explicit
// sub-steps:
//   * registers: every thread owns an 8 (ix) x 4 (iy) patch of cells: their saturations S and the face fluxes
//     Vx (9x4) and Vy (8x4 south faces) -- the whole member state S,Vx,Vy = 384 KB lives in the CU's 512 KB
//     register file; the north face of a patch's last column is the south face of the next lane's first column
//     and is fetched with a DPP wave shift (lanes = consecutive patches along iy);
//   * LDS (128 KB): the fractional-flow field fw(S) of the current sub-step, the only thing neighbouring
//     patches exchange; 16-byte chunks are XOR-swizzled so the b128 row reads are bank-conflict-free;
//     the iy-halo of a row comes from the neighbouring lane by DPP instead of LDS;
//   * HBM: S, Vx, Vy are read once and S written once per launch (per member-step).
// Upwind coefficients are re-derived from the face fluxes every sub-step (there is no register room for a
// fourth/fifth cell-sized array in fp64), with exactly the reference's operations, so results are bit-identical
// to the generic kernel and to oracle/ressim.py:saturation_step_upwind:
//     S_c <- S_c + (((((cE fE + cN fN) + cC fC) + cS fS) + cW fW) + fi_c dtx)      (E,N,C,S,W = CSR order)
// Wells (cells with q != 0) are handled by the thread that owns the cell in a branch-free side path that keeps the
// exact state of that cell in a small LDS record and overwrites the LDS fw entry (threads without a well run the
// same instructions on a dummy record: divergent control flow inside the sub-step loop makes the register
// allocator spill the whole flux state), so the straight-line code needs no per-cell source terms (x + 0.0 == x).
// Requires <= 1 well per patch and uniform porosity; otherwise the host falls back to the generic kernel.
//
// Dry waves: a wave whose 16 x 128 band holds S == 0 everywhere (and no injector) has fw == 0 there, so its phase A would store
// the zeros its rows of the LDS image already hold, and while the fw rows just above and below the band are zero as well its phase
// B adds exact zeros (x + (+-0) == x).  Such a wave skips both phases under a wave-uniform branch -- it only reads its two halo
// rows and ballots -- and the wet wave that shares its SIMD gets the whole issue rate: the early time steps, when the water has
// reached 2-4 of the 8 bands, run up to twice as fast.  Once water arrives (a non-zero halo row) the wave runs phase B and is wet
// for good.  Results are bit-identical (tests/test_forward_gpu.py::test_saturation_step_bitexact_given_fluxes and the rest).
// The same skip per patch row (a wet-row mask, phase A for wet rows, phase B for wet rows and their neighbours, frontier rows tested
// for water afterwards) is bit-identical too and executes fewer instructions, but the branch per row makes the register allocator
// spill three times as much (76 instead of 21 registers, 32 scratch reloads per sub-step): 19.8 ms instead of 13.85.
// Neighbour flags in LDS instead of the two workgroup barriers per sub-step (a wave exchanges fw only with the bands above and
// below its own) are bit-identical as well and slower: 14.3 ms, with or without s_sleep in the polls -- a wave at s_barrier takes
// no issue slots from the wave it shares its SIMD with, a polling one does.
// The 11 scratch instructions left in the sub-step loop (7 spilled doubles reloaded at the start of phase B) are not what bounds it
// either: built with `-mllvm -amdgpu-use-amdgpu-trackers=1` the loop has 5 and the launch takes 14.05 ms instead of 13.92.
//
// Compiled with -ffp-contract=off (no FMA contraction: every product and sum is rounded separately, as NumPy does).
#include "fracflow.h"

namespace {

constexpr int N128 = 128;
constexpr int PX = 8, PY = 4;
constexpr int NPY = N128 / PY;            // 32 patches along iy = 32 lanes
constexpr int NT = (N128 / PX) * NPY;     // 512 threads
constexpr int FW_BYTES = N128 * N128 * 8; // 128 KB
constexpr int REC_BYTES = 64;             // well record: S, cE, cN, cC, cS, cW, fid, fw
constexpr int MAX_WELLS = 16;
constexpr int VSP_BYTES = 3 * NT * 16;       // per-thread LDS home of 6 face fluxes (3 x 16 B): Vx[8][0..3], Vy[7][0..1]
constexpr int VSP_BASE = FW_BYTES + MAX_WELLS * REC_BYTES + 2 * REC_BYTES;

__device__ __forceinline__ int lds_off(int ix, int iy) {
    // byte offset of fw(ix, iy): rows of 1 KB; each thread's 32-byte row segment = two 16-byte chunks whose
    // order is flipped for every other group of 8 lanes -> ds_read_b128 of a row is conflict-free
    int py = iy >> 2;
    int chunk = (iy >> 1) ^ ((py >> 3) & 1);
    return ix * 1024 + chunk * 16 + (iy & 1) * 8;
}

// value of lane+1 / lane-1 (whole-wave shift); out-of-wave source reads as 0
__device__ __forceinline__ double from_next_lane(double v) {
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

// max(a, z) / min(a, z) as the one instruction they are: fmax() / fmin() put a canonicalising v_max_f64 x, x in front of every operand
// the compiler cannot prove free of signalling NaNs (the fluxes: they live in registers across the sub-step loop).  Same result for
// every operand that is not a signalling NaN.
__device__ __forceinline__ double vmax(double a, double z) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(z));
    return r;
}
__device__ __forceinline__ double vmin(double a, double z) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(z));
    return r;
}

template <bool FD>
__global__ __launch_bounds__(NT) void k_sat128(FwdParams p, const double* __restrict__ Sin_base,
                                               double* __restrict__ Sout_base, long long S_stride,
                                               double* __restrict__ prods, int k) {
    extern __shared__ __attribute__((aligned(16))) char lds[];  // [0,128K) fw field; then well records

    const int tid = threadIdx.x;
    const int m = blockIdx.x;
    const int py = tid & (NPY - 1), px = tid >> 5;
    const int ix0 = px * PX, iy0 = py * PY;

    const double* Sin = Sin_base + (long long)m * S_stride;
    double* Sout = Sout_base + (long long)m * S_stride;
    const double* gVx = p.Vx + (long long)m * (N128 + 1) * N128;
    const double* gVy = p.Vy + (long long)m * N128 * (N128 + 1);
    const double* q = p.q + (long long)(p.q_cols > 1 ? k : 0) * p.Nxy;

    // ---------------- member state -> registers
    // Vx row 8 (east boundary faces) and Vy[7][0..1] live in LDS (no register room): VX8(j), VY7(j)
    double S[PX][PY], Vx[PX][PY], Vy[PX][PY];
    double2* vsp = reinterpret_cast<double2*>(lds + VSP_BASE) + tid;  // chunk c at vsp[c * NT]
#define VX8(j) ((j) < 2 ? ((j) == 0 ? vx8a.x : vx8a.y) : ((j) == 2 ? vx8b.x : vx8b.y))
#define VY7(j) ((j) == 0 ? vy7a.x : vy7a.y)
#pragma unroll
    for (int i = 0; i < PX; ++i)
#pragma unroll
        for (int j = 0; j < PY; j += 2) {
            double2 v = *reinterpret_cast<const double2*>(Sin + (ix0 + i) * N128 + iy0 + j);
            S[i][j] = v.x;
            S[i][j + 1] = v.y;
        }
#pragma unroll
    for (int i = 0; i < PX; ++i)
#pragma unroll
        for (int j = 0; j < PY; j += 2) {
            double2 v = *reinterpret_cast<const double2*>(gVx + (ix0 + i) * N128 + iy0 + j);
            Vx[i][j] = v.x;
            Vx[i][j + 1] = v.y;
        }
#pragma unroll
    for (int i = 0; i < PX; ++i)
#pragma unroll
        for (int j = 0; j < PY; ++j) Vy[i][j] = gVy[(ix0 + i) * (N128 + 1) + iy0 + j];
    {   // park them in LDS right away (the region behind the fw field is not used by the CFL reduction)
        vsp[0] = *reinterpret_cast<const double2*>(gVx + (ix0 + PX) * N128 + iy0);
        vsp[NT] = *reinterpret_cast<const double2*>(gVx + (ix0 + PX) * N128 + iy0 + 2);
        vsp[2 * NT] = make_double2(Vy[PX - 1][0], Vy[PX - 1][1]);
    }

    // ---------------- the (at most one) well of this patch
    int wcell = -1, wrec = FW_BYTES + MAX_WELLS * REC_BYTES;  // non-owners work on a shared dummy record
    double wq = 0.0;
    const int nW = min(p.nInj + p.nPrd, MAX_WELLS);
    for (int w = 0; w < nW; ++w) {
        int cell = p.well_cells[w];
        if (((cell >> 7) >> 3) == px && ((cell & 127) >> 2) == py && q[cell] != 0.0) {
            wcell = cell;
            wq = q[cell];
            wrec = FW_BYTES + w * REC_BYTES;
        }
    }
    const bool has_well = wcell >= 0;

    // ---------------- CFL: pm = min over cells of pv / (Vi + fi)          
    const double pv = p.h2 * 1.0;
    double lmin = INFINITY;
#pragma unroll
    for (int i = 0; i < PX; ++i) {
        const double vyn3 = from_next_lane(Vy[i][0]);  // north face of column 3 (0 on the domain boundary)
#pragma unroll
        for (int j = 0; j < PY; ++j) {
            const double vyn = j + 1 < PY ? Vy[i][j + 1 < PY ? j + 1 : 0] : vyn3;
            double xp = fmax(Vx[i][j], 0.0), yp = fmax(Vy[i][j], 0.0);
            const double vxe = i + 1 < PX ? Vx[i + 1 < PX ? i + 1 : 0][j] : gVx[(ix0 + PX) * N128 + iy0 + j];
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
    double* red = reinterpret_cast<double*>(lds);
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

    // well record: exact coefficients including the source terms, exact S
    // (the dummy record is all zeros: S=0 -> fw=0, coefficients 0 -> stays 0; its "cell" slots live behind it)
    if (tid < 16) reinterpret_cast<double*>(lds + FW_BYTES + MAX_WELLS * REC_BYTES)[tid] = 0.0;
    __syncthreads();
    if (has_well) {
        double* rec = reinterpret_cast<double*>(lds + wrec);
        double fpq = fmin(wq, 0.0), fiq = fmax(wq, 0.0);
        double x1 = fmin(wVxW, 0.0), x2 = fmax(wVxE, 0.0), y1 = fmin(wVyS, 0.0), y2 = fmax(wVyN, 0.0);
        rec[0] = Sin[wcell];
        rec[1] = d * (-fmin(wVxE, 0.0));             // cE
        rec[2] = d * (-fmin(wVyN, 0.0));             // cN
        rec[3] = d * (fpq + x1 - x2 + y1 - y2);      // cC
        rec[4] = d * fmax(wVyS, 0.0);                // cS
        rec[5] = d * fmax(wVxW, 0.0);                // cW
        rec[6] = fiq * d;                            // fid
    }

    // LDS byte addresses of the well cell and its 4 neighbours are recomputed from `wcell` where they are used
    // (threads without a well point every one of them at a dummy slot behind the records)
    const int dummy = FW_BYTES + MAX_WELLS * REC_BYTES + REC_BYTES;
    auto well_addr = [&](int dx, int dy) {
        const int wix = wcell >> 7, wiy = wcell & 127;
        const int a = lds_off(min(max(wix + dx, 0), N128 - 1), min(max(wiy + dy, 0), N128 - 1));
        return has_well ? a : dummy;
    };
    __syncthreads();

    const int swz = (py >> 3) & 1;
    const int seg = py * 32;                       // byte offset of this thread's segment within a 1 KB row
    const int ixW = max(ix0 - 1, 0), ixE = min(ix0 + PX, N128 - 1);

    auto load_row = [&](int ix, double (&f)[PY]) {
        const char* base = lds + ix * 1024 + seg;
        double2 a = *reinterpret_cast<const double2*>(base + (swz * 16));
        double2 b = *reinterpret_cast<const double2*>(base + ((1 ^ swz) * 16));
        f[0] = a.x; f[1] = a.y; f[2] = b.x; f[3] = b.y;
    };

    // Register allocation of this kernel sits at the 256-VGPR edge and a scratch reload costs hundreds of cycles.  Wrapping
    // phase B in a branch on a run-time-true, compiler-opaque scalar changes where the allocator splits live ranges: 24
    // scratch instructions in the kernel instead of 64, 15.8 instead of 16.5 ms per launch (found by counting scratch
    // instructions over such perturbations; a guard around phase A as well is worse again).  No effect on results.
    int always = __builtin_amdgcn_readfirstlane(Nts > 0);
    asm volatile("" : "+s"(always));
    // wave-uniform: every cell of the band is exactly zero and no lane owns an injector (a producer in a dry band sits at S = 0:
    // its record stays 0 and its fw entry 0 until the band itself becomes wet)
    auto band_is_dry = [&]() {
        unsigned long long bits = 0ull;
#pragma unroll
        for (int i = 0; i < PX; ++i)
#pragma unroll
            for (int j = 0; j < PY; ++j) bits |= (unsigned long long)__double_as_longlong(S[i][j]) << 1;  // -0.0 counts as zero
        return p.swc == 0.0 && __ballot(bits != 0ull || (has_well && (wq > 0.0 || Sin[wcell] != 0.0))) == 0ull;  // swc > 0: fw(0) != 0, nothing is dry
    };
    int dry = band_is_dry() ? 1 : 0;
    const int wave_well = __builtin_amdgcn_readfirstlane(__ballot(has_well) != 0ull);
    // ---------------- explicit sub-steps
    for (int it = 0; it < Nts; ++it) {
        // The upwind coefficients are pure functions of (Vx, Vy, d): left alone, the compiler hoists all of them
        // (5 extra cell-sized arrays) out of the sub-step loop and spills.  Routing the constants d and 0.0 through
        // an empty asm makes every coefficient depend on a per-iteration opaque value (no instruction is emitted).
        double dd = d, z = 0.0;
        asm volatile("" : "+v"(dd), "+v"(z));

        // phase A: fractional flow of every own cell -> LDS (a dry band's rows already hold its zeros after the first sub-step)
        if (!(dry && it > 0)) {
#pragma unroll
        for (int i = 0; i < PX; ++i) {
            char* base = lds + (ix0 + i) * 1024 + seg;
            double2 a, b;
            a.x = frac_flow<FD>(p, S[i][0]);
            a.y = frac_flow<FD>(p, S[i][1]);
            b.x = frac_flow<FD>(p, S[i][2]);
            b.y = frac_flow<FD>(p, S[i][3]);
            *reinterpret_cast<double2*>(base + (swz * 16)) = a;
            *reinterpret_cast<double2*>(base + ((1 ^ swz) * 16)) = b;
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        if (wave_well) {  // well side path: lane-branch-free (lanes without a well run it on the dummy record), skipped by waves without wells
            double* rec = reinterpret_cast<double*>(lds + wrec);
            double wf = frac_flow<FD>(p, rec[0]);
            rec[7] = wf;
            *reinterpret_cast<double*>(lds + well_addr(0, 0)) = wf;  // after this thread's own row write: ordered
        }
        __syncthreads();

        int run_b = always;
        if (dry) {  // the band only changes if a fractional flow just outside it is non-zero: its west / east halo rows
            const unsigned long long* hw = reinterpret_cast<const unsigned long long*>(lds + ixW * 1024 + seg);
            const unsigned long long* he = reinterpret_cast<const unsigned long long*>(lds + ixE * 1024 + seg);
            const unsigned long long o = (hw[0] | hw[1]) | (hw[2] | hw[3]) | (he[0] | he[1]) | (he[2] | he[3]);  // fw >= +0: bit test
            run_b = __ballot(o != 0ull) != 0ull;
            dry = !run_b;  // water at the border: the band is wet from now on
        }
        if (run_b) {
        // phase B: upwind update row by row
#pragma unroll
        for (int i = 0; i < PX; ++i) {
            const int ix = ix0 + i;
            double fc[PY];
            load_row(ix, fc);
            const char* rowW = lds + (i > 0 ? ix - 1 : ixW) * 1024 + seg;
            const char* rowE = lds + (i + 1 < PX ? ix + 1 : ixE) * 1024 + seg;
            // (these two values also sit in the image; reading them from LDS instead of the neighbour lane's registers removes 32 DPP
            //  moves per sub-step from the VALU and was slower: 14.26 against 13.85 ms)
            const double fS = from_prev_lane(fc[PY - 1]);  // f(ix, iy0-1): its coefficient is 0 on the boundary
            const double fN = from_next_lane(fc[0]);       // f(ix, iy0+PY)
            double2 vx8a = make_double2(0, 0), vx8b = vx8a, vy7a = vx8a;
            if (i == PX - 1) {
                vx8a = vsp[0];
                vx8b = vsp[NT];
                vy7a = vsp[2 * NT];
            }
            const double vyn3 = i == PX - 1 ? from_next_lane(VY7(0)) : from_next_lane(Vy[i][0]);
#pragma unroll
            for (int jp = 0; jp < PY; jp += 2) {
                const double2 fwp = *reinterpret_cast<const double2*>(rowW + (((jp >> 1) ^ swz) * 16));
                const double2 fep = *reinterpret_cast<const double2*>(rowE + (((jp >> 1) ^ swz) * 16));
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int j = jp + jj;
                    const double vxw = Vx[i][j], vxe = (i == PX - 1) ? VX8(j) : Vx[i + 1 < PX ? i + 1 : 0][j];
                    const double vys = (i == PX - 1 && j < 2) ? VY7(j) : Vy[i][j];
                    const double vyn = j + 1 < PY ? ((i == PX - 1 && j + 1 < 2) ? VY7(j + 1 < 2 ? j + 1 : 0) : Vy[i][j + 1 < PY ? j + 1 : 0]) : vyn3;
                    const double x1 = vmin(vxw, z), x2 = vmax(vxe, z), y1 = vmin(vys, z), y2 = vmax(vyn, z);
                    const double cC = dd * (x1 - x2 + y1 - y2);
                    const double cW = dd * vmax(vxw, z);
                    const double cE = dd * (-vmin(vxe, z));
                    const double cS = dd * vmax(vys, z);
                    const double cN = dd * (-vmin(vyn, z));
                    const double fs = j > 0 ? fc[j > 0 ? j - 1 : 0] : fS;
                    const double fn = j + 1 < PY ? fc[j + 1 < PY ? j + 1 : 0] : fN;
                    double acc = cE * (jj ? fep.y : fep.x);
                    acc = acc + cN * fn;
                    acc = acc + cC * fc[j];
                    acc = acc + cS * fs;
                    acc = acc + cW * (jj ? fwp.y : fwp.x);
                    S[i][j] = S[i][j] + acc;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        if (wave_well) {
            double* rec = reinterpret_cast<double*>(lds + wrec);
            double acc = rec[1] * *reinterpret_cast<const double*>(lds + well_addr(1, 0));
            acc = acc + rec[2] * *reinterpret_cast<const double*>(lds + well_addr(0, 1));
            acc = acc + rec[3] * rec[7];
            acc = acc + rec[4] * *reinterpret_cast<const double*>(lds + well_addr(0, -1));
            acc = acc + rec[5] * *reinterpret_cast<const double*>(lds + well_addr(-1, 0));
            rec[0] = rec[0] + (acc + rec[6]);
        }
        __syncthreads();
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
            const int c0 = (ix0 + i) * N128 + iy0 + j;
            nonfinite |= (c0 != wcell && !isfinite(v.x)) || (c0 + 1 != wcell && !isfinite(v.y));
        }
    if (has_well) {
        const double wS = *reinterpret_cast<const double*>(lds + wrec);
        Sout[wcell] = wS;  // after this thread's own store of the patch: ordered
        nonfinite |= !isfinite(wS);
    }
    if (nonfinite) atomicOr(&p.status[m], HM_MEMBER_NONFINITE);
    __syncthreads();
    if (tid < p.nPrd) prods[((long long)m * p.nTime + k) * p.nPrd + tid] = Sout[p.prd_ind[tid]];
}

template <bool FD>
int launch(hm_fwd* f, const void* S_in, void* S_out, long long S_stride, int k) {
    const size_t lds = (size_t)VSP_BASE + VSP_BYTES;
    auto kern = k_sat128<FD>;
    HM_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(f->p.N), dim3(NT), lds, f->ctx->stream, f->p, (const double*)S_in, (double*)S_out,
                       S_stride, (double*)f->prods.p, k);
    HM_HIP(hipGetLastError());
    return 0;
}

}  // namespace

// Returns 0 if launched, >0 on error, -1 if this specialisation does not apply.
int launch_saturation_128(hm_fwd* f, const void* S_in, void* S_out, long long S_stride, int k) {
    const FwdParams& p = f->p;
    if (p.q_mstride != 0) return -1;  // per-member wells: the well side path works from one shared well list
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
