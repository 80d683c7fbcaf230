// sat128t.hip -- fp64 explicit upwind saturation sweep for grids made of 128 x 128-cell tiles
// (256 x 256, 512 x 512, 256 x 128 ...): TEAMS of workgroups, one workgroup per tile, one team per ensemble member.
//
// A member of a 256^2 / 512^2 grid (0.5 / 2 MB of saturation + 1 / 4 MB of face fluxes) does not fit one CU, and one
// workgroup per member streaming its state through L2 leaves the sweep 5-8x slower per cell than the register-resident
// 128 x 128 kernel (sat128.hip) -- and most of the chip idle at config 5's 125 members per GPU.  Here every tile is
// handled exactly as sat128.hip handles a whole 128 x 128 member (state in the CU's register file, fractional flow
// exchanged through LDS), and the tiles of a member additionally exchange their edge rows/columns of the fractional
// flow once per sub-step through a small L2-resident buffer:
//   * phase A: fw of own cells -> LDS (exactly sat128.hip's);
//   * hand-off: the 512 threads = 4 edges x 128 values.  Every thread publishes one value of its tile's edge (read back
//     from LDS, so a well on the edge travels with its exact fw) as two GRANULES -- 8-byte {tag = event number, 32 payload
//     bits} words written write-through (sc1) -- and polls the matching pair of the neighbouring tile with sc1 loads until
//     both tags match, then writes the value to the LDS halo (rows 128/129 of the fw image for W/E, a 128-entry
//     {fw, Vy_north} column for S/N).  The data is its own flag: no counters, no drain, no cache-wide release/acquire
//     (measured: 1.7 us per sub-step at full occupancy against 4-10 us with flag + agent-scope fences);
//   * phase B: as sat128.hip; lanes on the S/N tile border take their halo value from the LDS column instead of DPP.
// The CFL minimum is reduced over the team the same way (one all-tiles event per member), so every tile derives the
// same sub-step count.  The arithmetic per cell is unchanged -> bit-identical to the generic kernels and the oracle.
//
// Workgroups spin on each other, so all tiles of a team must be resident at once: a launch is at most one workgroup per
// CU (160 KB of LDS each) = 64 teams of 4 tiles or 16 teams of 16 (the host launches rounds of that many members; a
// member loop inside the kernel costs registers the sweep does not have), and the tiles of a team have workgroup ids
// that differ by multiples of 8 (workgroups are dealt round-robin to the 8 XCDs) so a team shares one L2 -- a speed
// bonus only, correctness does not depend on placement.  Spins are bounded: on a timeout the member is flagged
// HM_MEMBER_SYNC_TIMEOUT and the workgroup stops waiting.
//
// Compiled with -ffp-contract=off (no FMA contraction: every product and sum is rounded separately, as NumPy does).
#include "sat_team.h"
#include "fracflow.h"

namespace {

using namespace sat_team;

constexpr int PX = 8, PY = 4;
constexpr int NPY = TS / PY;                // 32 patches along iy = 32 lanes
constexpr int NT = (TS / PX) * NPY;         // 512 threads
constexpr int FW_BYTES = (TS + 2) * TS * 8; // fw rows 0..127 of the tile, row 128 = west halo, row 129 = east halo
constexpr int REC_BYTES = 64;               // well record: S, cE, cN, cC, cS, cW, fid, fw
constexpr int MAX_WELLS = 16;
constexpr int REC_BASE = FW_BYTES;
constexpr int EDGE_BASE = REC_BASE + MAX_WELLS * REC_BYTES + 2 * REC_BYTES;  // {fw halo, Vy north face}[2][128]: south, north
constexpr int EDGE_BYTES = 2 * TS * 16;
constexpr int VSP_BASE = EDGE_BASE + EDGE_BYTES;
constexpr int VSP_BYTES = 3 * NT * 16;      // per-thread LDS home of 6 face fluxes (3 x 16 B): Vx[8][0..3], Vy[7][0..1]
constexpr int MISC_BASE = VSP_BASE + VSP_BYTES;  // team CFL minima (32 doubles)
constexpr int LDS_TOTAL = MISC_BASE + 32 * 8;
static_assert(LDS_TOTAL <= 160 * 1024, "LDS budget");

__device__ __forceinline__ int lds_off(int ix, int iy) {
    // byte offset of fw(ix, iy): rows of 1 KB; each thread's 32-byte row segment = two 16-byte chunks whose
    // order is flipped for every other group of 8 lanes -> ds_read_b128 of a row is conflict-free
    int py = iy >> 2;
    int chunk = (iy >> 1) ^ ((py >> 3) & 1);
    return ix * 1024 + chunk * 16 + (iy & 1) * 8;
}

// LDS byte address of fw at tile-local (lix, liy), lix/liy in [-1, 128]: outside the tile -> the halo
__device__ __forceinline__ int tile_addr(int lix, int liy) {
    if (liy < 0) return EDGE_BASE + min(max(lix, 0), TS - 1) * 16;
    if (liy >= TS) return EDGE_BASE + (TS + min(max(lix, 0), TS - 1)) * 16;
    if (lix < 0) return lds_off(TS, liy);
    if (lix >= TS) return lds_off(TS + 1, liy);
    return lds_off(lix, liy);
}

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
__global__ __launch_bounds__(NT) void k_sat128t(FwdParams p, const double* __restrict__ Sin_base,
                                                double* __restrict__ Sout_base, long long S_stride,
                                                double* __restrict__ prods, int k, char* team_mem, int TXn, int TYn,
                                                int first_member) {
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x;
    const int T = TXn * TYn;
    int team, tile;
    team_of_block(T, team, tile);
    const int m = first_member + team;
    if (m >= p.N) return;
    const int tx = tile / TYn, ty = tile % TYn;
    const int gx0 = tx * TS, gy0 = ty * TS;
    const int Ny = p.Ny;
    const bool hasW = tx > 0, hasE = tx + 1 < TXn, hasS = ty > 0, hasN = ty + 1 < TYn;

    const TeamLayout<2> lay{T};
    char* tm = team_mem + (size_t)team * lay.bytes();
    u64* cflg = reinterpret_cast<u64*>(tm + lay.cfl_off());
    u64* pub = reinterpret_cast<u64*>(tm + lay.pub_off());
    __shared__ int dead_word;  // set once a wait has timed out: the workgroup stops waiting
    int* dead = &dead_word;
    double* team_min = reinterpret_cast<double*>(lds + MISC_BASE);

    const int py = tid & (NPY - 1), px = tid >> 5;
    const int ix0 = px * PX, iy0 = py * PY;      // tile-local origin of this thread's patch
    const bool isS = py == 0, isN = py == NPY - 1;
    // halos without a neighbour stay 0 (their coefficients are 0: boundary faces carry no flux)
    for (int i = tid; i < 2 * TS; i += NT) reinterpret_cast<double*>(lds + TS * 1024)[i] = 0.0;
    for (int i = tid; i < 4 * TS; i += NT) reinterpret_cast<double*>(lds + EDGE_BASE)[i] = 0.0;
    if (tid == 0) dead_word = 0;
    int ev = 0;      // events published by this tile so far (identical sequence in every tile of the team)
    __syncthreads();

    const double* Sin = Sin_base + (long long)m * S_stride;
    double* Sout = Sout_base + (long long)m * S_stride;
    const double* gVx = p.Vx + (long long)m * (p.Nx + 1) * Ny;
    const double* gVy = p.Vy + (long long)m * p.Nx * (Ny + 1);
    const double* q = p.q + (long long)(p.q_cols > 1 ? k : 0) * p.Nxy;

    // ---------------- tile state -> registers
    // Vx row 8 of the patch and Vy[7][0..1] live in LDS (no register room): VX8(j), VY7(j)
    double S[PX][PY], Vx[PX][PY], Vy[PX][PY];
    double2* vsp = reinterpret_cast<double2*>(lds + VSP_BASE) + tid;  // chunk c at vsp[c * NT]
#define VX8(j) ((j) < 2 ? ((j) == 0 ? vx8a.x : vx8a.y) : ((j) == 2 ? vx8b.x : vx8b.y))
#define VY7(j) ((j) == 0 ? vy7a.x : vy7a.y)
#pragma unroll
    for (int i = 0; i < PX; ++i)
#pragma unroll
        for (int j = 0; j < PY; j += 2) {
            double2 v = *reinterpret_cast<const double2*>(Sin + (long long)(gx0 + ix0 + i) * Ny + gy0 + iy0 + j);
            S[i][j] = v.x;
            S[i][j + 1] = v.y;
        }
#pragma unroll
    for (int i = 0; i < PX; ++i)
#pragma unroll
        for (int j = 0; j < PY; j += 2) {
            double2 v = *reinterpret_cast<const double2*>(gVx + (long long)(gx0 + ix0 + i) * Ny + gy0 + iy0 + j);
            Vx[i][j] = v.x;
            Vx[i][j + 1] = v.y;
        }
#pragma unroll
    for (int i = 0; i < PX; ++i)
#pragma unroll
        for (int j = 0; j < PY; ++j) Vy[i][j] = gVy[(long long)(gx0 + ix0 + i) * (Ny + 1) + gy0 + iy0 + j];
    {
        vsp[0] = *reinterpret_cast<const double2*>(gVx + (long long)(gx0 + ix0 + PX) * Ny + gy0 + iy0);
        vsp[NT] = *reinterpret_cast<const double2*>(gVx + (long long)(gx0 + ix0 + PX) * Ny + gy0 + iy0 + 2);
        vsp[2 * NT] = make_double2(Vy[PX - 1][0], Vy[PX - 1][1]);
    }
    // north face flux of the tile's last column (the next tile's first south face; 0 on the domain boundary)
    if (tid < TS) reinterpret_cast<double*>(lds + EDGE_BASE + (TS + tid) * 16)[1] = gVy[(long long)(gx0 + tid) * (Ny + 1) + gy0 + TS];

    // ---------------- the (at most one) well of this patch
    int wlx = -1, wly = -1, wcell = -1, wrec = REC_BASE + MAX_WELLS * REC_BYTES;  // non-owners: shared dummy record
    double wq = 0.0;
    const int nW = min(p.nInj + p.nPrd, MAX_WELLS);
    for (int w = 0; w < nW; ++w) {
        int cell = p.well_cells[w];
        int lx = cell / Ny - gx0, ly = cell % Ny - gy0;
        if (lx >= 0 && lx < TS && ly >= 0 && ly < TS && (lx >> 3) == px && (ly >> 2) == py && q[cell] != 0.0) {
            wcell = cell;
            wlx = lx;
            wly = ly;
            wq = q[cell];
            wrec = REC_BASE + w * REC_BYTES;
        }
    }
    const bool has_well = wcell >= 0;
    __syncthreads();  // the north-face column is in LDS

    // ---------------- CFL: pm = min over cells of pv / (Vi + fi)          
    const double pv = p.h2 * 1.0;
    double lmin = INFINITY;
#pragma unroll
    for (int i = 0; i < PX; ++i) {
        const double vyn_edge = reinterpret_cast<const double*>(lds + EDGE_BASE + (TS + ix0 + i) * 16)[1];
        const double vyn_dpp = from_next_lane(Vy[i][0]);
        const double vyn3 = isN ? vyn_edge : vyn_dpp;  // north face of column 3
#pragma unroll
        for (int j = 0; j < PY; ++j) {
            const double vyn = j + 1 < PY ? Vy[i][j + 1 < PY ? j + 1 : 0] : vyn3;
            double xp = fmax(Vx[i][j], 0.0), yp = fmax(Vy[i][j], 0.0);
            const double vxe = i + 1 < PX ? Vx[i + 1 < PX ? i + 1 : 0][j] : gVx[(long long)(gx0 + ix0 + PX) * Ny + gy0 + iy0 + j];
            double xn = fmin(vxe, 0.0), yn = fmin(vyn, 0.0);
            double Vi = xp + yp - xn - yn;
            lmin = fmin(lmin, pv / (Vi + 0.0));  // fi = 0 for every cell without an injector
        }
    }
    double wVxW = 0, wVxE = 0, wVyS = 0, wVyN = 0;
    if (has_well) {
        const long long wix = gx0 + wlx, wiy = gy0 + wly;
        wVxW = gVx[wix * Ny + wiy];
        wVxE = gVx[(wix + 1) * Ny + wiy];
        wVyS = gVy[wix * (Ny + 1) + wiy];
        wVyN = gVy[wix * (Ny + 1) + wiy + 1];
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
    // team-wide minimum: one all-tiles event
    if (tid == 0) put_double(cflg + (tile) * 2, cflg + (tile) * 2 + 1, red[0], ev + 1);
    if (tid < 64) {   // wave 0: lane t collects tile t's minimum
        const int t = tid < T ? tid : 0;
        double v = INFINITY;
        if (!get_double(cflg + (t) * 2, cflg + (t) * 2 + 1, ev + 1, v, dead)) v = INFINITY;
        if (tid < T) team_min[tid] = v;
    }
    ++ev;
    __syncthreads();
    double pm = team_min[0];
    for (int t = 1; t < T; ++t) pm = fmin(pm, team_min[t]);
    __syncthreads();
    const double sat = p.swc + p.sor;
    const double cfl = ((1.0 - sat) / 3.0) * pm;
    const double ntsd = ceil(p.dt / cfl);
    const bool bad = !(ntsd >= 1.0 && ntsd <= 1.0e7);
    const int Nts = bad ? 0 : (int)ntsd;
    if (tid == 0 && tile == 0) {
        p.nts[(long long)m * p.nTime + k] = Nts;
        if (bad) atomicOr(&p.status[m], HM_MEMBER_BAD_CFL);
    }
    const double d = bad ? 0.0 : (p.dt / (double)Nts) / pv;

    // well record: exact coefficients including the source terms, exact S
    if (tid < 16) reinterpret_cast<double*>(lds + REC_BASE + MAX_WELLS * REC_BYTES)[tid] = 0.0;
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
    const int dummy = REC_BASE + MAX_WELLS * REC_BYTES + REC_BYTES;
    auto well_addr = [&](int dx, int dy) { return has_well ? tile_addr(wlx + dx, wly + dy) : dummy; };
    __syncthreads();

    const int swz = (py >> 3) & 1;
    const int seg = py * 32;                       // byte offset of this thread's segment within a 1 KB row
    const int ixW = ix0 > 0 ? ix0 - 1 : (hasW ? TS : 0), ixE = ix0 + PX < TS ? ix0 + PX : (hasE ? TS + 1 : TS - 1);
    const char* edge_row0 = lds + EDGE_BASE + (isN ? TS * 16 : 0) + ix0 * 16;  // {fw halo, Vy north} entry of this patch's first row

    auto load_row = [&](int ix, double (&f)[PY]) {
        const char* base = lds + ix * 1024 + seg;
        double2 a = *reinterpret_cast<const double2*>(base + (swz * 16));
        double2 b = *reinterpret_cast<const double2*>(base + ((1 ^ swz) * 16));
        f[0] = a.x; f[1] = a.y; f[2] = b.x; f[3] = b.y;
    };

    int always = __builtin_amdgcn_readfirstlane(Nts > 0);
    asm volatile("" : "+s"(always));
    // Dry bands (as in sat128.hip): a wave whose 16 x 128 band of the tile holds S == 0 everywhere and owns no injector skips
    // phase A after the first sub-step (its rows of the fw image already hold its zeros; the hand-off publishes them from there)
    // and phase B while every fractional flow around the band is zero: the rows above and below it (the west / east halo rows
    // of the tile included) and, for the lanes on the tile's south / north border, the column halo.  Bit-identical.
    int dry;
    {
        unsigned long long bits = 0ull;
#pragma unroll
        for (int i = 0; i < PX; ++i)
#pragma unroll
            for (int j = 0; j < PY; ++j) bits |= (unsigned long long)__double_as_longlong(S[i][j]) << 1;  // -0.0 counts as zero
        dry = p.swc == 0.0 && __ballot(bits != 0ull || (has_well && (wq > 0.0 || Sin[wcell] != 0.0))) == 0ull;  // swc > 0: fw(0) != 0, nothing is dry
    }
    const int wave_well = __builtin_amdgcn_readfirstlane(__ballot(has_well) != 0ull);
#ifdef HM_SAT_PROF
    unsigned long long prof_a = 0, prof_h = 0, prof_b = 0, prof_t;
#define STAMP(acc) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); acc += t_ - prof_t; prof_t = t_; } while (0)
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(prof_t) :: "memory");
#else
#define STAMP(acc)
#endif
    // ---------------- explicit sub-steps
    for (int it = 0; it < Nts; ++it, ++ev) {
        double dd = d, z = 0.0;
        asm volatile("" : "+v"(dd), "+v"(z));
        // phase A: fractional flow of every own cell -> LDS
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
        STAMP(prof_a);
        {
            int t = tid;
            asm volatile("" : "+v"(t));
            const int e = t >> 7, idx = t & (TS - 1);
            const int src = e == 0 ? lds_off(0, idx) : e == 1 ? lds_off(TS - 1, idx) : e == 2 ? lds_off(idx, 0) : lds_off(idx, TS - 1);
            const int dst = e == 0 ? lds_off(TS, idx) : e == 1 ? lds_off(TS + 1, idx) : EDGE_BASE + ((e - 2) * TS + idx) * 16;
            const bool has = e == 0 ? hasW : e == 1 ? hasE : e == 2 ? hasS : hasN;
            const int nb = e == 0 ? tile - TYn : e == 1 ? tile + TYn : e == 2 ? tile - 1 : tile + 1;
            u64* mine = pub + ((size_t)(tile * 2 + (ev & 1)) * 4 + e) * 2 * TS + idx;
            put_double(mine, mine + TS, *reinterpret_cast<const double*>(lds + src), ev + 1);
            if (has) {
                const u64* theirs = pub + ((size_t)(nb * 2 + (ev & 1)) * 4 + (e ^ 1)) * 2 * TS + idx;
                double v;
                if (get_double(theirs, theirs + TS, ev + 1, v, dead)) *reinterpret_cast<double*>(lds + dst) = v;
            }
        }
        __syncthreads();
        STAMP(prof_h);

        int run_b = always;
        if (dry) {
            const unsigned long long* hw = reinterpret_cast<const unsigned long long*>(lds + ixW * 1024 + seg);
            const unsigned long long* he = reinterpret_cast<const unsigned long long*>(lds + ixE * 1024 + seg);
            unsigned long long o = (hw[0] | hw[1]) | (hw[2] | hw[3]) | (he[0] | he[1]) | (he[2] | he[3]);  // fw >= +0: bit test
            unsigned long long oc = 0ull;  // column halo {fw, Vy north}: the fw entries of this patch's 8 rows
#pragma unroll
            for (int i = 0; i < PX; ++i) oc |= *reinterpret_cast<const unsigned long long*>(edge_row0 + i * 16);
            if (isS || isN) o |= oc;
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
            const double2 eh = *reinterpret_cast<const double2*>(edge_row0 + i * 16);  // {fw halo, Vy north} of this row
            const double fSd = from_prev_lane(fc[PY - 1]);  // f(ix, iy0-1): its coefficient is 0 on the boundary
            const double fNd = from_next_lane(fc[0]);       // f(ix, iy0+PY)
            const double fS = isS ? eh.x : fSd;
            const double fN = isN ? eh.x : fNd;
            double2 vx8a = make_double2(0, 0), vx8b = vx8a, vy7a = vx8a;
            if (i == PX - 1) {
                vx8a = vsp[0];
                vx8b = vsp[NT];
                vy7a = vsp[2 * NT];
            }
            const double vyn3d = i == PX - 1 ? from_next_lane(VY7(0)) : from_next_lane(Vy[i][0]);
            const double vyn3 = isN ? eh.y : vyn3d;
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
        STAMP(prof_b);
    }
#ifdef HM_SAT_PROF
    if (tid == 0 && (m == 0 || m == p.N - 1))
        printf("sat128t member %d tile %d: Nts %d, cycles per sub-step: phase A %.0f, hand-off %.0f, phase B %.0f\n", m, tile, Nts,
               (double)prof_a / Nts, (double)prof_h / Nts, (double)prof_b / Nts);
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
            const long long c0 = (long long)(gx0 + ix0 + i) * Ny + gy0 + iy0 + j;
            *reinterpret_cast<double2*>(Sout + c0) = v;
            nonfinite |= (c0 != wcell && !isfinite(v.x)) || (c0 + 1 != wcell && !isfinite(v.y));
        }
    if (has_well) {
        const double wS = *reinterpret_cast<const double*>(lds + wrec);
        Sout[wcell] = wS;  // after this thread's own store of the patch: ordered
        nonfinite |= !isfinite(wS);
    }
    if (nonfinite) atomicOr(&p.status[m], HM_MEMBER_NONFINITE);
    if (tid == 0 && dead_word) atomicOr(&p.status[m], HM_MEMBER_SYNC_TIMEOUT);
    __syncthreads();
    if (tid < p.nPrd) {
        const int cell = p.prd_ind[tid];
        const int lx = cell / Ny - gx0, ly = cell % Ny - gy0;
        if (lx >= 0 && lx < TS && ly >= 0 && ly < TS) prods[((long long)m * p.nTime + k) * p.nPrd + tid] = Sout[cell];
    }
}

template <bool FD>
int launch(hm_fwd* f, const void* S_in, void* S_out, long long S_stride, int k, int TXn, int TYn, int max_teams) {
    const FwdParams& p = f->p;
    const int T = TXn * TYn;
    const TeamLayout<2> lay{T};
    const size_t need = lay.bytes() * (size_t)max_teams;
    if (f->team_mem.bytes < need) {
        hm_dev_free(f->team_mem);
        int rc = hm_dev_alloc(f->team_mem, need);
        if (rc) return rc;
    }
    hipStream_t s = f->ctx->stream;
    auto kern = k_sat128t<FD>;
    HM_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL));
    int resident = 0;  // the runtime's own answer: can a workgroup of this kernel be resident on a CU at all?
    HM_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, kern, NT, LDS_TOTAL));
    if (resident < 1) return -1;  // no: the caller falls back to the single-workgroup tiled sweep
    // one launch per round of max_teams members (a member loop inside the kernel costs registers the sweep does not have)
    for (int first = 0; first < p.N; first += max_teams) {
        const int nteams = std::min(max_teams, p.N - first);
        const int used_per_xcd = (nteams + 7) / 8;
        HM_HIP(hipMemsetAsync(f->team_mem.p, 0, lay.bytes() * (size_t)nteams, s));  // tags restart at 0 every launch
        hipLaunchKernelGGL(kern, dim3(8 * used_per_xcd * T), dim3(NT), LDS_TOTAL, s, f->p, (const double*)S_in, (double*)S_out,
                           S_stride, (double*)f->prods.p, k, (char*)f->team_mem.p, TXn, TYn, first);
    }
    HM_HIP(hipGetLastError());
    return 0;
}

}  // namespace

// Returns 0 if launched, >0 on error, -1 if this specialisation does not apply.
int launch_saturation_128t(hm_fwd* f, const void* S_in, void* S_out, long long S_stride, int k) {
    const FwdParams& p = f->p;
    int TXn, TYn, max_teams;
    if (p.q_mstride != 0) return -1;  // per-member wells: the well side path works from one shared well list
    if (f->dtype != 64 || p.por != nullptr || !tiles_of(f, TXn, TYn, max_teams) || !wells_fit_patches(f, MAX_WELLS)) return -1;
    return p.fluid_default ? launch<true>(f, S_in, S_out, S_stride, k, TXn, TYn, max_teams)
                           : launch<false>(f, S_in, S_out, S_stride, k, TXn, TYn, max_teams);
}
