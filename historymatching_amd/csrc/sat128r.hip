// sat128r.hip -- 128x128 fp64 explicit upwind saturation sweep, fractional flow held in registers.
//
// Same decomposition as sat128.hip (one workgroup of 512 threads = one member, resident on one CU for all Nts sub-steps; a thread owns
// an 8 (ix) x 4 (iy) patch; lanes = consecutive patches along iy), with two changes that cut the instructions per cell and sub-step
// from 39 to 28 and the LDS traffic per thread and sub-step from 1 KB to 0.4 KB:
//   * the fractional-flow field no longer goes through an LDS image.  A thread sweeps its patch row by row with a rolling window of
//     three fw rows in registers (the row above, the row itself, the row below; the next row's fw is computed from the not-yet-updated
//     S while the current row is updated); the iy-neighbours come from the adjacent lanes by DPP wave shifts as before; the only
//     exchange through LDS is the ix-halo -- every thread publishes 64 bytes (below) and reads 64 bytes;
//   * the registers hold the SCALED face fluxes d Vx, d Vy (d = dtx / pore volume), from which the four off-diagonal upwind
//     coefficients are one v_max / v_min each: d max(v, 0) == max(d v, 0) and d (-min(v, 0)) == -min(d v, 0) bit for bit (d > 0: the
//     rounding of a product is sign-symmetric; zeros and NaNs go the same way through the same instruction).  The diagonal coefficient
//     c_C = d ((((fp + x1) - x2) + y1) - y2) is not a function of the scaled fluxes alone; it is computed once per launch, exactly as
//     the reference does, into the 128 KB of LDS that the fw image used to occupy (thread-private, read one row at a time).
// The east term of a patch's last row, c_E f_E, has its coefficient on the west face of the patch BELOW and its fw in that patch's
// row 0: the patch below computes the product (the same two operands, the same instruction) and publishes it instead of f, so no
// thread holds a ninth row of x-fluxes.  The fluxes through the domain boundary are zero (the pressure kernels write them so): the
// boundary terms are (+-0) x (a finite fw), as in sat128.hip.
// Wells: c_C of a producer's cell includes its rate (it is data now), so producers need nothing else.  An injector adds fi d to the
// update, S + (acc + fi d): on patch rows holding an injector (a wave-uniform mask) the addend is selected per lane and column
// before S is updated -- for every other cell of such a row it is 0.0, which is what the reference adds.
// Dry waves (every S of the wave's 16 x 128 band exactly zero, no injector) publish zeros and skip the sweep until something non-zero
// arrives from a neighbouring band, as in sat128.hip.
// Bit-identical to sat128.hip, the generic kernel and oracle/ressim.py:saturation_step_upwind (tests/test_forward_gpu.py).
// Compiled with -ffp-contract=off.
#include "fracflow.h"

#ifdef HM_SAT_PROF
__device__ long long hm_sat_prof_buf[64];  // workgroup 0: [wave][publish, barrier 1, sweep, barrier 2, loop cycles, loop time (10 ns), dry sub-steps, Nts]
#define SPROF(slot) do { const long long t_ = clock64(); sprof[slot] += t_ - sprof_t; sprof_t = t_; } while (0)
#else
#define SPROF(slot) do { } while (0)
#endif

namespace {

constexpr int N128 = 128;
constexpr int PX = 8, PY = 4;
constexpr int NPY = N128 / PY;         // 32 lanes along iy
constexpr int NT = (N128 / PX) * NPY;  // 512 threads
constexpr int CHUNK = NT * 16;         // one 16-byte chunk per thread: consecutive lanes, consecutive chunks (conflict-free b128)
constexpr int ARR_BYTES = 2 * PX * CHUNK;     // c_C: chunk (2 i + c) holds columns 2c, 2c+1 of patch row i
constexpr int HW_BASE = ARR_BYTES;            // fw of patch row 7 (the west halo of the patch below), chunks c = 0, 1
constexpr int HE_BASE = HW_BASE + 2 * CHUNK;  // c_E f_E for row 7 of the patch above, computed from this patch's west faces and row 0
constexpr int LDS_BYTES = HE_BASE + 2 * CHUNK;  // 160 KB
constexpr int MAX_WELLS = 16;

__device__ __forceinline__ double next_lane(double v) {  // value of lane + 1; 0 beyond the wave
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double prev_lane(double v) {  // value of lane - 1
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// c f for an off-diagonal upwind coefficient c = max(+-v, 0) of a scaled face flux v = d V: ONE instruction, the product with the VOP3
// `clamp` output modifier (and the `neg` input modifier): clamp(x) = min(max(x, +0), 1), every product of the sweep is below 1 (the CFL
// bound keeps the coefficients below 1/3, f <= 1), f >= +0, so  max(v, 0) f == max(v f, +0) == clamp(v f)  bit for bit, denormal products
// included (profiles/r05/fp32_rate.txt, clamp_f64.txt).  Where the reference's coefficient is -0 (-min of a positive flux) the product here
// is +0 instead of -0: the sum of the five terms is the same -- a zero sum is +0 either way because the c_S and c_W terms are never -0 --
// and nothing else sees the term.  (Rounds 2-4 formed the coefficient first, one v_max_f64 / v_min_f64 each: 4 of the 26 instructions per
// cell and sub-step.)
__device__ __forceinline__ double mulc(double v, double f) {
    double r;
    asm("v_mul_f64 %0, %1, %2 clamp" : "=v"(r) : "v"(v), "v"(f));
    return r;
}
__device__ __forceinline__ double nmulc(double v, double f) {
    double r;
    asm("v_mul_f64 %0, -%1, %2 clamp" : "=v"(r) : "v"(v), "v"(f));
    return r;
}

// FROMP (round 6, lazy face fluxes: fwd.h): the pressure step left P, TX, TY and no Vx, Vy -- the thread forms the fluxes of its patch's faces
// itself, V = (P_upwind-side - P_this) T with zero on the domain boundary: face_fluxes' expression (fwd_dev.h), the same bits; k_nd_flux's
// launch, its 0.26 MB of writes per member and this kernel's read of them are gone.
template <bool FD, bool FROMP>
__global__ __launch_bounds__(NT) void k_sat128r(FwdParams p, const double* __restrict__ Sin_base, double* __restrict__ Sout_base,
                                                long long S_stride, double* __restrict__ prods, int k) {
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

    double S[PX][PY], Vx[PX][PY], Vy[PX][PY], Vx8[PY];
#pragma unroll
    for (int i = 0; i < PX; ++i)
#pragma unroll
        for (int j = 0; j < PY; j += 2) {
            const double2 v = *reinterpret_cast<const double2*>(Sin + (ix0 + i) * N128 + iy0 + j);
            S[i][j] = v.x;
            S[i][j + 1] = v.y;
        }
    if constexpr (!FROMP) {
#pragma unroll
        for (int i = 0; i <= PX; ++i)
#pragma unroll
            for (int j = 0; j < PY; j += 2) {
                const double2 v = *reinterpret_cast<const double2*>(gVx + (ix0 + i) * N128 + iy0 + j);
                if (i < PX) { Vx[i < PX ? i : 0][j] = v.x; Vx[i < PX ? i : 0][j + 1] = v.y; }
                else { Vx8[j] = v.x; Vx8[j + 1] = v.y; }  // the east faces of the last row: for the CFL bound and c_C only
            }
#pragma unroll
        for (int i = 0; i < PX; ++i)
#pragma unroll
            for (int j = 0; j < PY; ++j) Vy[i][j] = gVy[(ix0 + i) * (N128 + 1) + iy0 + j];
    } else {
        const double* gP = p.P + (long long)m * p.Nxy;
        const double* gTX = p.TX + (long long)m * (N128 + 1) * N128;
        const double* gTY = p.TY + (long long)m * N128 * (N128 + 1);
        // pressures of rows ix0 - 1 .. ix0 + 8 (clamped into the grid: a clamped row only meets a boundary face, whose flux is set to zero)
        double Pm[PY], Pc[PY];
        {
            const int r = ix0 > 0 ? ix0 - 1 : 0;
#pragma unroll
            for (int j = 0; j < PY; j += 2) {
                const double2 v = *reinterpret_cast<const double2*>(gP + r * N128 + iy0 + j);
                Pm[j] = v.x; Pm[j + 1] = v.y;
            }
        }
#pragma unroll
        for (int i = 0; i <= PX; ++i) {
            const int ix = ix0 + i, r = ix < N128 ? ix : N128 - 1;
            double tx[PY];
#pragma unroll
            for (int j = 0; j < PY; j += 2) {
                const double2 v = *reinterpret_cast<const double2*>(gP + r * N128 + iy0 + j);
                Pc[j] = v.x; Pc[j + 1] = v.y;
                const double2 t = *reinterpret_cast<const double2*>(gTX + ix * N128 + iy0 + j);
                tx[j] = t.x; tx[j + 1] = t.y;
            }
            const bool edge = ix == 0 || ix == N128;
#pragma unroll
            for (int j = 0; j < PY; ++j) {
                const double v = edge ? 0.0 : (Pm[j] - Pc[j]) * tx[j];  // face_fluxes (fwd_dev.h)
                if (i < PX) Vx[i < PX ? i : 0][j] = v;
                else Vx8[j] = v;
            }
            if (i < PX) {  // the south faces of row ix: its own pressures and the one of the cell south of the patch
                const double ps = gP[r * N128 + (iy0 > 0 ? iy0 - 1 : 0)];
#pragma unroll
                for (int j = 0; j < PY; ++j) {
                    const int iy = iy0 + j;
                    const double pl = j > 0 ? Pc[j > 0 ? j - 1 : 0] : ps;
                    const double ty = gTY[ix * (N128 + 1) + iy];
                    Vy[i][j] = iy == 0 ? 0.0 : (pl - Pc[j]) * ty;  // (iy = 128 is no cell's south face)
                }
            }
#pragma unroll
            for (int j = 0; j < PY; ++j) Pm[j] = Pc[j];
        }
    }
#define VXE(i, j) ((i) + 1 < PX ? Vx[(i) + 1 < PX ? (i) + 1 : 0][j] : Vx8[j])

    // ---------------- the (at most one) well of this patch
    int wcell = -1;
    double wq = 0.0;
    const int nW = min(p.nInj + p.nPrd, MAX_WELLS);
    for (int w = 0; w < nW; ++w) {
        const int cell = p.well_cells[w];
        if (((cell >> 7) >> 3) == px && ((cell & 127) >> 2) == py && q[cell] != 0.0) {
            wcell = cell;
            wq = q[cell];
        }
    }
    const bool has_well = wcell >= 0;
    const int wrow = has_well ? ((wcell >> 7) & (PX - 1)) : -1, wcol = wcell & (PY - 1);
    const double fpq = fmin(wq, 0.0), fiq = fmax(wq, 0.0);

    // ---------------- CFL: pm = min over cells of pv / (Vi + fi)          
    const double pv = p.h2 * 1.0;
    double lmin = INFINITY;
#pragma unroll
    for (int i = 0; i < PX; ++i) {
        const double vyn3 = next_lane(Vy[i][0]);  // north face of column 3 (0 on the domain boundary)
#pragma unroll
        for (int j = 0; j < PY; ++j) {
            const double vyn = j + 1 < PY ? Vy[i][j + 1 < PY ? j + 1 : 0] : vyn3;
            const double xp = fmax(Vx[i][j], 0.0), yp = fmax(Vy[i][j], 0.0);
            const double xn = fmin(VXE(i, j), 0.0), yn = fmin(vyn, 0.0);
            const double Vi = xp + yp - xn - yn;
            lmin = fmin(lmin, pv / (Vi + ((wrow == i && wcol == j) ? fiq : 0.0)));
        }
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

    // ---------------- c_C -> LDS (thread-private), then the fluxes are scaled in place
    char* arr = lds + tid * 16;
#pragma unroll
    for (int i = 0; i < PX; ++i) {
        double a[PY];
        const double vyn3 = next_lane(Vy[i][0]);
#pragma unroll
        for (int j = 0; j < PY; ++j) {
            const double vyn = j + 1 < PY ? Vy[i][j + 1 < PY ? j + 1 : 0] : vyn3;
            const double x1 = fmin(Vx[i][j], 0.0), x2 = fmax(VXE(i, j), 0.0), y1 = fmin(Vy[i][j], 0.0), y2 = fmax(vyn, 0.0);
            a[j] = (wrow == i && wcol == j) ? d * (fpq + x1 - x2 + y1 - y2) : d * (x1 - x2 + y1 - y2);
        }
        *reinterpret_cast<double2*>(arr + (2 * i) * CHUNK) = make_double2(a[0], a[1]);
        *reinterpret_cast<double2*>(arr + (2 * i + 1) * CHUNK) = make_double2(a[2], a[3]);
    }
#pragma unroll
    for (int i = 0; i < PX; ++i)
#pragma unroll
        for (int j = 0; j < PY; ++j) {
            Vx[i][j] = d * Vx[i][j];
            Vy[i][j] = d * Vy[i][j];
        }

    // halo slots: written by their owner, read by the patch below (HW) / above (HE).  The first patch row has nobody above it: its HE
    // slots hold the east terms of the LAST patch row instead, (-0) x fw = -0 (no flux through the boundary).  The first patch row's
    // own west halo is its own HW slot: any finite fw against the coefficient max(0, 0).
    char* pubW = lds + HW_BASE + tid * 16;
    char* pubE = lds + HE_BASE + tid * 16;
    const char* getW = px > 0 ? pubW - 32 * 16 : pubW;
    const char* getE = px + 1 < N128 / PX ? pubE + 32 * 16 : lds + HE_BASE + py * 16;
    if (px == 0) {
        *reinterpret_cast<double2*>(pubE) = make_double2(-0.0, -0.0);
        *reinterpret_cast<double2*>(pubE + CHUNK) = make_double2(-0.0, -0.0);
    }

    // the injector of this wave (the host admits at most one per wave): its patch row and the four per-column addends
    // (fi d in the injector's column, 0.0 elsewhere) are wave-uniform, its lane is a mask
    const bool inj = has_well && wq > 0.0;
    const unsigned long long injb = __ballot(inj);
    const int injl = injb ? __ffsll((long long)injb) - 1 : 0;
    const int irow = injb ? __builtin_amdgcn_readlane(wrow, injl) : -1, icol = __builtin_amdgcn_readlane(wcol, injl);
    const double fid = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(fiq * d), injl), __builtin_amdgcn_readlane(__double2loint(fiq * d), injl));
    const double fi0 = icol == 0 ? fid : 0.0, fi1 = icol == 1 ? fid : 0.0, fi2 = icol == 2 ? fid : 0.0, fi3 = icol == 3 ? fid : 0.0;
    int dry;
    {
        unsigned long long bits = 0ull;
#pragma unroll
        for (int i = 0; i < PX; ++i)
#pragma unroll
            for (int j = 0; j < PY; ++j) bits |= (unsigned long long)__double_as_longlong(S[i][j]) << 1;  // -0.0 counts as zero
        dry = p.swc == 0.0 && __ballot(bits != 0ull || inj) == 0ull;  // swc > 0: fw(0) != 0, nothing is dry
    }
    auto ff4 = [&](const double (&s)[PY], double (&f)[PY]) {
#pragma unroll
        for (int j = 0; j < PY; ++j) f[j] = frac_flow<FD>(p, s[j]);
    };
    auto ld4 = [&](const char* a, double (&f)[PY]) {
        const double2 u = *reinterpret_cast<const double2*>(a), v = *reinterpret_cast<const double2*>(a + CHUNK);
        f[0] = u.x; f[1] = u.y; f[2] = v.x; f[3] = v.y;
    };
    auto st4 = [&](char* a, const double (&f)[PY]) {
        *reinterpret_cast<double2*>(a) = make_double2(f[0], f[1]);
        *reinterpret_cast<double2*>(a + CHUNK) = make_double2(f[2], f[3]);
    };

#ifdef HM_SAT_PROF
    long long sprof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long sprof_t = clock64();
    const long long sprof_c0 = sprof_t, sprof_w0 = wall_clock64();
#endif
    // ---------------- explicit sub-steps
    for (int it = 0; it < Nts; ++it) {
        double fc[PY], fm[PY], fn[PY];
        {
            double f7[PY], te[PY];
            if (!dry) {
                ff4(S[0], fc);
                ff4(S[PX - 1], f7);
            } else {
#pragma unroll
                for (int j = 0; j < PY; ++j) fc[j] = f7[j] = 0.0;
            }
#pragma unroll
            for (int j = 0; j < PY; ++j) te[j] = nmulc(Vx[0][j], fc[j]);  // c_E f_E of the cell above, (ix0 - 1, iy0 + j)
            if (px > 0) st4(pubE, te);
            st4(pubW, f7);
        }
        SPROF(0);
        __syncthreads();
        SPROF(1);
#ifdef HM_SAT_PROF
        sprof[6] += dry;
#endif
        ld4(getW, fm);
        if (dry) {  // the band only changes once something non-zero arrives from just outside it
            double he[PY];
            ld4(getE, he);
            unsigned long long o = 0ull;
#pragma unroll
            for (int j = 0; j < PY; ++j)
                o |= (unsigned long long)__double_as_longlong(fm[j]) | ((unsigned long long)__double_as_longlong(he[j]) << 1);
            dry = __ballot(o != 0ull) == 0ull;  // fw >= +0: bit test; an east term can be -0
        }
        if (!dry) {
#pragma unroll
            for (int i = 0; i < PX; ++i) {
                if (i + 2 < PX) ff4(S[i + 1], fn);
                else if (i + 2 == PX) ld4(pubW, fn);  // this thread's own row 7, as published
                else ld4(getE, fn);                   // row 7: the east TERMS, not fw
                double ar[PY];
                ld4(arr + (2 * i) * CHUNK, ar);
                const double fS = prev_lane(fc[PY - 1]);  // f(ix, iy0 - 1): its coefficient is 0 on the boundary
                // c_N f_N of column 3 has both operands in the NEXT lane (its column 0: the south face flux and the fw): that lane forms the
                // product (same operands, same instruction) and the product is shifted -- 2 DPP moves instead of 4
                const double tN3 = next_lane(nmulc(Vy[i][0], fc[0]));
                double acc[PY];
#pragma unroll
                for (int j = 0; j < PY; ++j) {
                    const double fs = j > 0 ? fc[j > 0 ? j - 1 : 0] : fS;
                    double a = i + 1 < PX ? nmulc(Vx[i + 1 < PX ? i + 1 : 0][j], fn[j]) : fn[j];  // c_E f_E = max(-d Vx_e, 0) f_E
                    a = a + (j + 1 < PY ? nmulc(Vy[i][j + 1 < PY ? j + 1 : 0], fc[j + 1 < PY ? j + 1 : 0]) : tN3);  // c_N f_N
                    a = a + ar[j] * fc[j];
                    a = a + mulc(Vy[i][j], fs);    // c_S f_S = max(d Vy, 0) f_S
                    acc[j] = a + mulc(Vx[i][j], fm[j]);
                }
#ifndef HM_SAT_NOWELL
                // the injector's row (wave-uniform): its lane adds fi d in its column before S is updated.  A scalar branch inside the
                // asm: a branch the compiler sees costs this loop some forty scratch reloads per sub-step.
                asm volatile("s_cmp_lg_u32 %[ir], %[i]\n\t"
                             "s_cbranch_scc1 .Lsat128r_noinj_%=\n\t"
                             "s_mov_b64 exec, %[m]\n\t"
                             "v_add_f64 %[a0], %[a0], %[f0]\n\t"
                             "v_add_f64 %[a1], %[a1], %[f1]\n\t"
                             "v_add_f64 %[a2], %[a2], %[f2]\n\t"
                             "v_add_f64 %[a3], %[a3], %[f3]\n\t"
                             "s_mov_b64 exec, -1\n"
                             ".Lsat128r_noinj_%=:"
                             : [a0] "+v"(acc[0]), [a1] "+v"(acc[1]), [a2] "+v"(acc[2]), [a3] "+v"(acc[3])
                             : [ir] "s"(irow), [i] "s"(i), [m] "s"(injb), [f0] "s"(fi0), [f1] "s"(fi1), [f2] "s"(fi2), [f3] "s"(fi3)
                             : "scc");
#endif
#pragma unroll
                for (int j = 0; j < PY; ++j) S[i][j] = S[i][j] + acc[j];
#pragma unroll
                for (int j = 0; j < PY; ++j) { fm[j] = fc[j]; fc[j] = fn[j]; }
            }
        }
        SPROF(2);
        __syncthreads();
        SPROF(3);
    }
#ifdef HM_SAT_PROF
    if (blockIdx.x == 0 && (tid & 63) == 0) {
        sprof[4] = clock64() - sprof_c0;
        sprof[5] = wall_clock64() - sprof_w0;
        sprof[7] = Nts;
        for (int i = 0; i < 8; ++i) hm_sat_prof_buf[(tid >> 6) * 8 + i] = sprof[i];
    }
#endif

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
    __threadfence_block();
    __syncthreads();
    if (tid < p.nPrd) prods[((long long)m * p.nTime + k) * p.nPrd + tid] = Sout[p.prd_ind[tid]];
}

template <bool FD, bool FROMP>
int launch(hm_fwd* f, const void* S_in, void* S_out, long long S_stride, int k) {
    auto kern = k_sat128r<FD, FROMP>;
    HM_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
    hipLaunchKernelGGL(kern, dim3(f->p.N), dim3(NT), LDS_BYTES, f->ctx->stream, f->p, (const double*)S_in, (double*)S_out, S_stride,
                       (double*)f->prods.p, k);
    HM_HIP(hipGetLastError());
    return 0;
}

}  // namespace

#ifdef HM_SAT_PROF
extern "C" int hm_debug_sat_prof(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(hm_sat_prof_buf), sizeof(long long) * 64); }
#endif

// Returns 0 if launched, >0 on error, -1 if this specialisation does not apply.
int launch_saturation_128r(hm_fwd* f, const void* S_in, void* S_out, long long S_stride, int k) {
    const FwdParams& p = f->p;
    if (p.q_mstride != 0) return -1;  // per-member wells: the well cells come from one shared well list
    if (p.Nx != N128 || p.Ny != N128 || f->dtype != 64 || p.por != nullptr) return -1;
    if ((int)f->well_cells_host.size() > MAX_WELLS) return -1;
    std::vector<int> seen;  // at most one well per 8x4 patch
    for (int cell : f->well_cells_host) {
        const int id = ((cell >> 7) >> 3) * 1000 + ((cell & 127) >> 2);
        for (int s : seen)
            if (s == id) return -1;
        seen.push_back(id);
    }
    // at most one injector (a well with q > 0 in this time column) per wave = per band of 16 grid rows
    const double* qk = f->q_host.data() + (size_t)(p.q_cols > 1 ? k : 0) * p.Nxy;
    int inj_waves = 0;
    for (int cell : f->well_cells_host)
        if (qk[cell] > 0.0) {
            const int bit = 1 << ((cell >> 7) >> 4);
            if (inj_waves & bit) return -1;
            inj_waves |= bit;
        }
    if (f->flux_pending)  // (lazy face fluxes, fwd.h: P, TX, TY are current, Vx / Vy are not -- and stay so: they are materialised on demand)
        return p.fluid_default ? launch<true, true>(f, S_in, S_out, S_stride, k) : launch<false, true>(f, S_in, S_out, S_stride, k);
    return p.fluid_default ? launch<true, false>(f, S_in, S_out, S_stride, k) : launch<false, false>(f, S_in, S_out, S_stride, k);
}
