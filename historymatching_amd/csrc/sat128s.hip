// sat128s.hip -- fp64 explicit upwind saturation sweep of the 128 x 128 grid for SMALL member shards: the register sweep of
// sat128r.hip on SLABS of 64 (or 32) rows, a member = a TEAM of two (four) workgroups of 256 (128) threads.
//
// sat128r.hip gives a member one workgroup on one CU: an ensemble shard below 256 members leaves CUs idle -- 125 members (BASELINE config 2's
// N_e = 1000 over 8 GPUs) occupy 125 of 256, and the time step takes what it takes with 256 (profiles/r06/small_shards_before.txt: 39 k
// ensemble-steps/s at 125 members against 75 k at 1000: 4.2x of the ideal 8x when config 2 is split over 8 GPUs).  Here the member's 128
// rows are cut into slabs, one workgroup each, every workgroup on a CU of its own: four waves on four SIMDs sweep a slab of 64 rows in the
// time sat128r's SIMDs take for ONE of their two waves (profiles/r06/sat128r_cycle_stamps.txt: 5.2 k against 8.0 k cycles a sub-step), and
// the slabs trade one row per sub-step the way sat256s.hip's do: the last patch row hands the fw of its last row DOWN, the first patch row
// the east terms of the row above it UP, as GRANULES (sat_team.h).  A patch row is 32 lanes here -- half a wave -- so the hand-off is a
// per-lane branch of the first / last wave, not a wave-uniform one.  Everything else -- thread patch 8 x 4, scaled fluxes in registers, the
// diagonal coefficient as data in LDS, rolling window of three fw rows, clamp-modifier products, the injector's scalar branch, dry bands,
// the team's CFL minimum through granules -- is sat256s.hip / sat128r.hip: the same arithmetic per cell, bit-identical to both and to
// oracle/ressim.py:saturation_step_upwind (tests/test_forward_gpu.py).  Chosen by forward.hip when the shard's teams fit the CUs
// (members x slabs <= CUs); spins are bounded, a timed-out member is redone by the device-gated tiled sweep (forward.hip).
// Compiled with -ffp-contract=off.
#include "sat_team.h"
#include "fracflow.h"

namespace {

using sat_team::u64;

constexpr int NY = 128;
constexpr int PX = 8, PY = 4;
constexpr int NPY = NY / PY;           // 32 lanes = one patch row across the grid: a wave holds two
// NPR: patch rows per slab (8: slabs of 64 rows, teams of two; 4: slabs of 32 rows, teams of four)
template <int NPR>
struct Geo {
    static constexpr int SLAB = NPR * PX;          // rows per workgroup
    static constexpr int NT = NPR * NPY;           // 256 / 128 threads
    static constexpr int CHUNK = NT * 16;
    static constexpr int ARR_BYTES = 2 * PX * CHUNK;     // c_C: chunk (2 i + c) holds columns 2c, 2c+1 of patch row i
    static constexpr int HW_BASE = ARR_BYTES;            // fw of patch row 7 (the west halo of the patch below), chunks c = 0, 1
    static constexpr int HE_BASE = HW_BASE + 2 * CHUNK;  // c_E f_E for row 7 of the patch above
    static constexpr int LDS_BYTES = HE_BASE + 2 * CHUNK + 16;  // 80 / 40 KB (+ the team minimum)
};
constexpr int MAX_WELLS = 16;
constexpr int SPIN_LIMIT = 1 << 22;

// team block: CFL granules [T][2] (padded to 512 B), then per slab [2 parities][2 directions: 0 = down (f7), 1 = up (east terms)][8 granules][32 lanes]
__host__ __device__ inline size_t team_pub_off() { return 512; }
__host__ __device__ inline size_t team_bytes(int T) { return team_pub_off() + (size_t)T * 2 * 2 * 8 * NPY * 8; }

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

// four doubles of this lane as eight granules; slot = [8][NPY] u64
__device__ __forceinline__ void put4(u64* slot, int lane, const double (&v)[PY], unsigned tag) {
#pragma unroll
    for (int j = 0; j < PY; ++j) {
        sat_team::put_granule(slot + (2 * j) * NPY + lane, (unsigned)__double2loint(v[j]), tag);
        sat_team::put_granule(slot + (2 * j + 1) * NPY + lane, (unsigned)__double2hiint(v[j]), tag);
    }
}
// poll until all eight carry `tag` (the wave leaves together); `failed`: a wait of this wave has timed out -- no more waiting
__device__ __forceinline__ void get4(const u64* slot, int lane, double (&v)[PY], unsigned tag, int& failed) {
#ifdef SLAB_NOPOLL  // timing experiment (wrong results): no hand-off at all -- what the polls cost, see the header
    for (int j = 0; j < PY; ++j) v[j] = 0.0;
    return;
#endif
    u64 g[2 * PY];
    for (int spins = 0;; ++spins) {
        bool ok = true;
#pragma unroll
        for (int q = 0; q < 2 * PY; ++q) {
            g[q] = __hip_atomic_load(slot + q * NPY + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = ok && (unsigned)(g[q] >> 32) == tag;
        }
        if (__all(ok) || failed) break;
        if (spins > SPIN_LIMIT) { failed = 1; break; }
        __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int j = 0; j < PY; ++j) v[j] = __hiloint2double((int)(unsigned)g[2 * j + 1], (int)(unsigned)g[2 * j]);
}


// (launch bound 512 threads although 256 / 128 are launched: with a bound of 256 the compiler may give a thread 512 registers and parks what
// does not fit the 256 architectural ones in accumulator registers -- 330 v_accvgpr moves inside the sub-step loop, a third more vector
// instructions; bounded like sat128r.hip it keeps the loop in 256 registers with its spills outside)
template <int NPR, bool FD>
__global__ __launch_bounds__(512) void k_sat128s(FwdParams p, const double* __restrict__ Sin_base, double* __restrict__ Sout_base,
                                                long long S_stride, double* __restrict__ prods, int k, char* team_mem, int T, int first_member) {
    constexpr int SLAB = Geo<NPR>::SLAB, NT = Geo<NPR>::NT, CHUNK = Geo<NPR>::CHUNK, HW_BASE = Geo<NPR>::HW_BASE, HE_BASE = Geo<NPR>::HE_BASE;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    int team, slab;
    sat_team::team_of_block(T, team, slab);
    const int m = first_member + team;
    if (m >= p.N) return;
    const int px = tid >> 5;          // patch row of the slab (two to a wave)
    const int lt = tid;
    const int py = tid & (NPY - 1);
    const int gx0 = slab * SLAB + px * PX, iy0 = py * PY;  // global row of the patch's first row
    const bool hasPrev = slab > 0, hasNext = slab + 1 < T;

    char* tm = team_mem + (size_t)team * team_bytes(T);
    u64* cflg = reinterpret_cast<u64*>(tm);
    u64* pub = reinterpret_cast<u64*>(tm + team_pub_off());
    auto slot = [&](int sl, int par, int dir) { return pub + (((size_t)sl * 2 + par) * 2 + dir) * (8 * NPY); };
    int failed = 0;

    const double* Sin = Sin_base + (long long)m * S_stride;
    double* Sout = Sout_base + (long long)m * S_stride;
    const double* gVx = p.Vx + (long long)m * (p.Nx + 1) * NY;
    const double* gVy = p.Vy + (long long)m * p.Nx * (NY + 1);
    const double* q = p.q + (long long)(p.q_cols > 1 ? k : 0) * p.Nxy;

    double S[PX][PY], Vx[PX][PY], Vy[PX][PY], Vx8[PY];
#pragma unroll
    for (int i = 0; i < PX; ++i)
#pragma unroll
        for (int j = 0; j < PY; j += 2) {
            const double2 v = *reinterpret_cast<const double2*>(Sin + (long long)(gx0 + i) * NY + iy0 + j);
            S[i][j] = v.x;
            S[i][j + 1] = v.y;
        }
#pragma unroll
    for (int i = 0; i <= PX; ++i)
#pragma unroll
        for (int j = 0; j < PY; j += 2) {
            const double2 v = *reinterpret_cast<const double2*>(gVx + (long long)(gx0 + i) * NY + iy0 + j);
            if (i < PX) { Vx[i < PX ? i : 0][j] = v.x; Vx[i < PX ? i : 0][j + 1] = v.y; }
            else { Vx8[j] = v.x; Vx8[j + 1] = v.y; }  // the east faces of the last row: for the CFL bound and c_C only
        }
#pragma unroll
    for (int i = 0; i < PX; ++i)
#pragma unroll
        for (int j = 0; j < PY; ++j) Vy[i][j] = gVy[(long long)(gx0 + i) * (NY + 1) + iy0 + j];
#define VXE(i, j) ((i) + 1 < PX ? Vx[(i) + 1 < PX ? (i) + 1 : 0][j] : Vx8[j])

    // ---------------- the (at most one) well of this patch
    int wcell = -1;
    double wq = 0.0;
    const int nW = min(p.nInj + p.nPrd, MAX_WELLS);
    for (int w = 0; w < nW; ++w) {
        const int cell = p.well_cells[w];
        if ((cell / NY - gx0) >= 0 && (cell / NY - gx0) < PX && ((cell % NY) >> 2) == py && q[cell] != 0.0) {
            wcell = cell;
            wq = q[cell];
        }
    }
    const bool has_well = wcell >= 0;
    const int wrow = has_well ? wcell / NY - gx0 : -1, wcol = wcell & (PY - 1);
    const double fpq = fmin(wq, 0.0), fiq = fmax(wq, 0.0);

    // ---------------- CFL: pm = min over cells of pv / (Vi + fi), over the whole member = the team      
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
    if (tid == 0) sat_team::put_double(cflg + slab * 2, cflg + slab * 2 + 1, red[0], 1u);
    if (tid < 64) {  // wave 0: lane t collects slab t's minimum
        const int t = tid < T ? tid : 0;
        u64 x = 0, y = 0;
        for (int spins = 0;; ++spins) {
            x = __hip_atomic_load(cflg + t * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            y = __hip_atomic_load(cflg + t * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all((unsigned)(x >> 32) == 1u && (unsigned)(y >> 32) == 1u)) break;
            if (spins > SPIN_LIMIT) { failed = 1; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        double v = failed ? INFINITY : __hiloint2double((int)(unsigned)y, (int)(unsigned)x);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off));
        if (tid == 0) red[NT] = v;
    }
    __syncthreads();
    const double pm = red[NT];
    __syncthreads();
    const double sat = p.swc + p.sor;
    const double cfl = ((1.0 - sat) / 3.0) * pm;
    const double ntsd = ceil(p.dt / cfl);
    const bool bad = !(ntsd >= 1.0 && ntsd <= 1.0e7);
    const int Nts = bad ? 0 : (int)ntsd;
    if (tid == 0 && slab == 0) {
        p.nts[(long long)m * p.nTime + k] = Nts;
        if (bad) atomicOr(&p.status[m], HM_MEMBER_BAD_CFL);
    }
    const double d = bad ? 0.0 : (p.dt / (double)Nts) / pv;

    // ---------------- c_C -> LDS (thread-private), then the fluxes are scaled in place
    char* arr = lds + lt * 16;
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

    // halo slots inside the slab as in sat128r.hip.  The first wave's HE slots are read by the LAST wave as the east terms of its row 7:
    // the next slab's (wave 7 parks what it polled there) or, at the end of the domain, (-0) x fw = -0.  The first wave's own west halo:
    // the previous slab's (polled into registers) or, at the start of the domain, its own HW slot (any finite fw against max(0, 0)).
    char* pubW = lds + HW_BASE + lt * 16;
    char* pubE = lds + HE_BASE + lt * 16;
    const char* getW = px > 0 ? pubW - NPY * 16 : pubW;
    char* edgeE = lds + HE_BASE + py * 16;  // the first wave's HE slot of this lane
    const char* getE = px + 1 < NPR ? pubE + NPY * 16 : edgeE;
    if (px == 0) {
        *reinterpret_cast<double2*>(pubE) = make_double2(-0.0, -0.0);
        *reinterpret_cast<double2*>(pubE + CHUNK) = make_double2(-0.0, -0.0);
    }

    // the injector of this wave (the host admits at most one per wave)
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
    const bool first = px == 0 && hasPrev, last = px == NPR - 1 && hasNext;  // (per half wave) the patch rows that talk to a neighbouring slab

    // ---------------- explicit sub-steps
    for (int it = 0; it < Nts; ++it) {
        const unsigned tag = (unsigned)it + 2u;
        const int par = it & 1;
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
            for (int j = 0; j < PY; ++j) te[j] = nmulc(Vx[0][j], fc[j]);  // c_E f_E of the cell above, (gx0 - 1, iy0 + j)
            if (px > 0) st4(pubE, te);
            st4(pubW, f7);
            if (first) put4(slot(slab, par, 1), py, te, tag);             // up: the previous slab's last wave wants them
            if (last) put4(slot(slab, par, 0), py, f7, tag);              // down: the next slab's first wave wants them
        }
        __syncthreads();
        if (last) {  // the next slab's east terms for this wave's row 7 -> the LDS slots the row loop reads them from
            double he[PY];
            get4(slot(slab + 1, par, 1), py, he, tag, failed);
            st4(edgeE, he);
        }
        if (first) get4(slot(slab - 1, par, 0), py, fm, tag, failed);
        else ld4(getW, fm);
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
                // the injector's row (wave-uniform): a scalar branch inside the asm, as in sat128r.hip
                asm volatile("s_cmp_lg_u32 %[ir], %[i]\n\t"
                             "s_cbranch_scc1 .Lsat128s_noinj_%=\n\t"
                             "s_mov_b64 exec, %[m]\n\t"
                             "v_add_f64 %[a0], %[a0], %[f0]\n\t"
                             "v_add_f64 %[a1], %[a1], %[f1]\n\t"
                             "v_add_f64 %[a2], %[a2], %[f2]\n\t"
                             "v_add_f64 %[a3], %[a3], %[f3]\n\t"
                             "s_mov_b64 exec, -1\n"
                             ".Lsat128s_noinj_%=:"
                             : [a0] "+v"(acc[0]), [a1] "+v"(acc[1]), [a2] "+v"(acc[2]), [a3] "+v"(acc[3])
                             : [ir] "s"(irow), [i] "s"(i), [m] "s"(injb), [f0] "s"(fi0), [f1] "s"(fi1), [f2] "s"(fi2), [f3] "s"(fi3)
                             : "scc");
#pragma unroll
                for (int j = 0; j < PY; ++j) S[i][j] = S[i][j] + acc[j];
#pragma unroll
                for (int j = 0; j < PY; ++j) { fm[j] = fc[j]; fc[j] = fn[j]; }
            }
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
            *reinterpret_cast<double2*>(Sout + (long long)(gx0 + i) * NY + iy0 + j) = v;
            nonfinite |= !isfinite(v.x) || !isfinite(v.y);
        }
    if (nonfinite) atomicOr(&p.status[m], HM_MEMBER_NONFINITE);
    if (__ballot(failed) != 0ull && (tid & 63) == 0) atomicOr(&p.status[m], HM_MEMBER_SYNC_TIMEOUT);
    __threadfence_block();
    __syncthreads();
    if (tid < p.nPrd) {
        const int cell = p.prd_ind[tid];
        if (cell / NY >= slab * SLAB && cell / NY < (slab + 1) * SLAB) prods[((long long)m * p.nTime + k) * p.nPrd + tid] = Sout[cell];
    }
}

template <int NPR, bool FD>
int launch(hm_fwd* f, const void* S_in, void* S_out, long long S_stride, int k, int T) {
    const FwdParams& p = f->p;
    const size_t need = team_bytes(T) * (size_t)p.N;
    if (f->team_mem.bytes < need) {
        hm_dev_free(f->team_mem);
        int rc = hm_dev_alloc(f->team_mem, need);
        if (rc) return rc;
    }
    hipStream_t s = f->ctx->stream;
    auto kern = k_sat128s<NPR, FD>;
    constexpr int NT = Geo<NPR>::NT, LDS_BYTES = Geo<NPR>::LDS_BYTES;
    HM_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
    int resident = 0;
    HM_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, kern, NT, LDS_BYTES));
    if (resident < 1) return -1;
    const int used_per_xcd = (p.N + 7) / 8;  // every team of the shard in ONE launch: members x slabs <= CUs (the caller's condition)
    HM_HIP(hipMemsetAsync(f->team_mem.p, 0, team_bytes(T) * (size_t)p.N, s));  // tags restart at 0 every launch
    hipLaunchKernelGGL(kern, dim3(8 * used_per_xcd * T), dim3(NT), LDS_BYTES, s, f->p, (const double*)S_in, (double*)S_out, S_stride,
                       (double*)f->prods.p, k, (char*)f->team_mem.p, T, 0);
    HM_HIP(hipGetLastError());
    return 0;
}

}  // namespace

// Returns 0 if launched, >0 on error, -1 if this specialisation does not apply: the 128 x 128 fp64 sweep as workgroup teams, for member shards
// whose teams all find a CU of their own at once (hm_fwd_set_debug "sat_teams": 0 = never, 2 / 4 = that many slabs whatever the shard).
// slabs per member the team sweep would run this plan's step k with: 2, 4, or 0 (it does not apply)
static int sat128s_teams(const hm_fwd* f, int k) {
    const FwdParams& p = f->p;
    if (p.q_mstride != 0) return 0;  // per-member wells: the well cells come from one shared well list
    if (p.Nx != NY || p.Ny != NY || f->dtype != 64 || p.por != nullptr) return 0;
    const int cus = f->ctx->num_cu;
    int T = f->dbg_sat_teams;
    if (T < 0) T = 4 * p.N <= cus ? 4 : 2 * p.N <= cus ? 2 : 0;  // automatic: the most slabs whose teams all fit the CUs
    if (T != 2 && T != 4) return 0;
    if (T * ((p.N + 7) / 8) * 8 > cus) return 0;                 // (teams are placed by XCD: sat_team.h team_of_block)
    if (!sat_team::wells_fit_patches(f, MAX_WELLS)) return 0;
    // at most one injector (a well with q > 0 in this time column) per wave = per band of 16 grid rows
    const double* qk = f->q_host.data() + (size_t)(p.q_cols > 1 ? k : 0) * p.Nxy;
    std::vector<int> bands;
    for (int cell : f->well_cells_host)
        if (qk[cell] > 0.0) {
            const int band = (cell / NY) / (2 * PX);
            for (int b : bands)
                if (b == band) return 0;
            bands.push_back(band);
        }
    return T;
}
bool sat128s_applies(const hm_fwd* f, int k) { return sat128s_teams(f, k) != 0; }

int launch_saturation_128s(hm_fwd* f, const void* S_in, void* S_out, long long S_stride, int k) {
    const FwdParams& p = f->p;
    const int T = sat128s_teams(f, k);
    if (!T) return -1;
    if (T == 2) return p.fluid_default ? launch<8, true>(f, S_in, S_out, S_stride, k, T) : launch<8, false>(f, S_in, S_out, S_stride, k, T);
    return p.fluid_default ? launch<4, true>(f, S_in, S_out, S_stride, k, T) : launch<4, false>(f, S_in, S_out, S_stride, k, T);
}
