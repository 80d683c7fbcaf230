// sat32s.hip -- float32 explicit upwind saturation sweep of dtype = 32 plans on grids 128, 256 or 512 cells wide,
// fractional flow in registers, the saturation as the compensated pair of sat32.h: the register-window sweep of sat128r.hip /
// sat256s.hip re-cut for single precision.
//
// A thread owns a 4 (ix) x 8 (iy) patch; the NY / 8 lanes of a patch row span the grid WIDTH (16 / 32 / 64 lanes: four, two or one patch
// rows per wave), a workgroup of 8 waves is a SLAB of 128 / 64 / 32 rows = 16 384 cells, and the Nx / SLAB slabs of a member form a
// TEAM of workgroups, all resident at once (128 x 128: a team of one, no traffic between workgroups at all).  Per cell the registers
// hold base and dS (sat32.h), the SCALED west- and south-face fluxes fx = (float)(d Vx), fy = (float)(d Vy) and the diagonal coefficient
// c_C: five words, 160 of the 256 registers -- the four off-diagonal coefficients are not stored.  The reference's
//     c_W f_W,   c_W = (float)(d max(Vx, 0))         is      clamp(fx f_W)       (the VOP3 `clamp` output modifier: one v_mul_f32)
//     c_E f_E,   c_E = (float)(d (-min(Vx_e, 0)))    is      clamp(-fx_e f_E)    (`neg` input modifier)
// bit for bit: d > 0 and rounding are sign-symmetric and monotone, so max(.., 0) commutes with the scaling and the conversion; f >= +0,
// so it commutes with the product as well; every product of the sweep is below 1, where clamp(x) == max(x, +0) including denormals
// (profiles/r05/fp32_rate.txt).  Where the reference's coefficient is -0 (a positive flux under -min) the product here is +0: the
// sign of a zero term never reaches the state (dS starts at +0; (+-0) + x = x; base + (+0)).  c_C = (float)(d ((((fp + x1) - x2) +
// y1) - y2)) is not a function of the rounded fluxes: it is formed once per launch in fp64 as the reference does and kept as data.
// A sub-step:  fw of the patch's rows 0 and 3 from s = base + dS; row 3's fw (the west halo of the patch below) and the east TERMS
// clamp(-fx[0] f[0]) of the row above are published in LDS (32 bytes each per thread; two parities, so ONE workgroup barrier per
// sub-step); then the four rows in turn with a rolling window of three fw rows (row i+1's fw from its not-yet-updated state), the
// iy-neighbours by DPP wave shifts (past a patch row's end the shift brings the neighbouring row's value against a zero boundary flux).
// Between slabs the first and the last patch row trade the same two 8-value records per sub-step as GRANULES (sat_team.h: 8-byte
// {tag, float} words, write-through stores, polled until the tag matches); the last row polls right before its row 3, behind three
// rows of work.  Wells: a producer's rate is part of c_C; the injector's lane adds fi d in a scalar branch inside asm (sat128r.hip).
// Dry waves (every base and dS of the wave zero, no injector) publish zeros and skip the sweep until something non-zero arrives.
// Arithmetic per cell and sub-step: 19 VALU instructions (s, fw: 8 -- the division in four, fracflow.h --, 5 products, 4 + 1 sums) + 0.25
// DPP moves.
// Bit-identical to k_saturation_generic<float> / _stream / _tiled and to oracle/ressim.py:saturation_step_stencil_f32c.
// Spins are bounded: on a timeout the member is flagged HM_MEMBER_SYNC_TIMEOUT, and the device-gated launch of the single-workgroup tiled
// sweep that follows every team launch (forward.hip: launch_saturation) redoes the member's step -- nothing waits for the host.
// Compiled with -ffp-contract=off (and without the SLP vectoriser: Makefile).
#include "sat_team.h"
#include "fracflow.h"
#include "sat32.h"

#ifdef HM_SAT_PROF
__device__ long long hm_sat32_prof_buf[64];  // workgroup 0: [wave][publish, barrier, halo (polls), sweep, fold, loop cycles, dry sub-steps, Nts]
#define SPROF(slot) do { const long long t_ = clock64(); sprof[slot] += t_ - sprof_t; sprof_t = t_; } while (0)
#else
#define SPROF(slot) do { } while (0)
#endif

namespace {

using sat_team::u64;

constexpr int PX = 4, PY = 8;
constexpr int NW = 8;                  // waves per workgroup
constexpr int NT = NW * 64;            // 512 threads
constexpr int CHUNK = NT * 16;         // one 16-byte chunk per thread: consecutive lanes, consecutive chunks (conflict-free b128)
// LDS: per parity [HW chunk 0, 1 | HE chunk 0, 1]; then the edge slots of the first / last patch row (what they polled, or zeros)
constexpr int HALO_BYTES = 2 * 4 * CHUNK;       // 64 KB
constexpr int EDGE_BASE = HALO_BYTES;           // [W | E][parity][chunk][64 lanes] x 16 bytes
constexpr int LDS_BYTES = EDGE_BASE + 2 * 2 * 2 * 64 * 16;  // 72 KB
constexpr int MAX_WELLS = 16;
constexpr int MAX_SLABS = 32;
constexpr int SPIN_LIMIT = 1 << 22;

__device__ __forceinline__ float next_lane(float v) {  // value of lane + 1; 0 beyond the wave
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));
}
__device__ __forceinline__ float prev_lane(float v) {  // value of lane - 1
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));
}
// clamp(a b) and clamp(-a b): max(+-a b, +0) for products below 1 (see the header)
__device__ __forceinline__ float mulc(float a, float b) {
    float r;
    asm("v_mul_f32_e64 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float nmulc(float a, float b) {
    float r;
    asm("v_mul_f32_e64 %0, -%1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// team block: CFL granules [T][2] (padded to 512 B), then per slab [2 parities][2 directions: 0 = down (fw of the last row), 1 = up (east terms)][8 values][NY / 8 lanes]
__host__ __device__ inline size_t team_pub_off() { return 512; }
__host__ __device__ inline size_t team_bytes(int T, int NY) { return team_pub_off() + (size_t)T * 2 * 2 * NY * 8; }

template <int LPR>
__device__ __forceinline__ void put8(u64* slot, int py, const float (&v)[PY], unsigned tag) {
#pragma unroll
    for (int j = 0; j < PY; ++j) sat_team::put_granule(slot + j * LPR + py, __float_as_uint(v[j]), tag);
}
// poll until all eight carry `tag` (the active lanes leave together); `failed`: a wait of this thread's wave has timed out -- no more waiting
template <int LPR>
__device__ __forceinline__ void get8(const u64* slot, int py, float (&v)[PY], unsigned tag, int& failed) {
    u64 g[PY];
    for (int spins = 0;; ++spins) {
        bool ok = true;
#pragma unroll
        for (int j = 0; j < PY; ++j) {
            g[j] = __hip_atomic_load(slot + j * LPR + py, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = ok && (unsigned)(g[j] >> 32) == tag;
        }
        if (__all(ok) || failed) break;
        if (spins > SPIN_LIMIT) { failed = 1; break; }
        __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int j = 0; j < PY; ++j) v[j] = __uint_as_float((unsigned)g[j]);
}

template <int NY, bool FD>
__global__ __launch_bounds__(NT) void k_sat32s(FwdParams p, const float* __restrict__ Sin_base, float* __restrict__ Sout_base,
                                               long long S_stride, float* __restrict__ prods, int k, char* team_mem, int T, int first_member,
                                               const unsigned char* __restrict__ wet_in, unsigned char* __restrict__ wet_out, int all_active, int* counters) {
    constexpr int LPR = NY / PY;          // lanes per patch row
    constexpr int NROWS = NT / LPR;       // patch rows per slab
    constexpr int SLAB = NROWS * PX;      // grid rows per slab
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    int team, slab;
    sat_team::team_of_block(T, team, slab);
    const int m = first_member + team;
    if (m >= p.N) return;
    // all_active == 2: the REDO launch that follows every launch in which slabs may sit out -- only the members flagged HM_MEMBER_REDO_STEP
    // (water reached a slab that sat the step out), all their slabs; everybody else leaves here.  The flag is cleared by slab 0 once every
    // workgroup of the member has published its CFL minimum, i.e. has read it.
    const bool redo = all_active == 2;
    if (redo && !(p.status[m] & HM_MEMBER_REDO_STEP)) return;
    // Which wave takes which band of the slab: the slab's LAST band (it hands its row 3 down to the next slab) goes to hardware wave 1,
    // not 7.  Of the two waves on a SIMD the earlier-dispatched one (0..3) wins the issue arbitration and is through its rows in 2.7 k of
    // the sub-step's 6.2 k cycles (profiles/r05/sat32_prof_128.txt; waves 4..7: 4.6 k), so its next record leaves two thousand cycles
    // earlier -- about the latency of a granule hand-off, which the next slab's first band otherwise waits out every sub-step.
    const int lw = (tid >> 6) == 0 ? 0 : (tid >> 6) == 1 ? NW - 1 : (tid >> 6) - 1;  // logical wave: position of this wave's band in the slab
    const int lt = lw * 64 + (tid & 63);                                             // logical thread id: geometry and LDS slots follow it
    const int py = lt % LPR, prow = lt / LPR;
    const int gx0 = slab * SLAB + prow * PX, iy0 = py * PY;  // global row / column of the patch's first cell
    // ACTIVE slabs.  A slab whose cells are all dry, with dry neighbouring slabs and no injector, cannot change within one time step (the
    // front would have to cross a whole slab of 16 384 / Ny rows): its workgroup contributes its cells to the CFL bound, writes zeros and
    // LEAVES -- the CU goes to the next workgroup of the launch, so a launch costs what its wet slabs cost (at 512 x 512 the injector's slab
    // is wet from the first step, the outer ones only late in a run).  What is wet is recorded per member and slab at the end of every
    // launch (wet_out) and read by the next (wet_in; all_active: the record is not valid -- first step of a run, inputs changed).
    // Every workgroup of a member evaluates the same rule on the same record, so neighbours agree on who takes part; towards an inactive
    // neighbour a slab behaves as at the domain boundary (zero halo), and checks at the end that its border row is still dry.  It need
    // not be: ahead of the visible front the saturation decays doubly exponentially but stays bitwise non-zero down to 1e-45, and in a
    // high-flux channel that frontier of denormal values can run a hundred cells within the 9 831 sub-steps of one time step (observed on 1
    // of 125 members of config 5's prior for six steps in a row).  Such a member is flagged HM_MEMBER_REDO_STEP and its step redone with
    // every slab by the REDO launch that follows (below): 26 ms, bit-identical.
    const unsigned char* wet = wet_in + (size_t)m * T;
    auto slab_has_injector = [&](int sl) {
        bool any = false;
        const int nW0 = min(p.nInj + p.nPrd, MAX_WELLS);
        for (int w = 0; w < nW0; ++w) {
            const int cell = p.well_cells[w];
            any = any || (cell / NY / SLAB == sl && p.q[(long long)(p.q_cols > 1 ? k : 0) * p.Nxy + cell] > 0.0);
        }
        return any;
    };
    const bool narrow = all_active == 3;  // (hm_fwd_set_debug "slab_margin" 0: the neighbours of a wet slab sit out as well -- tests of the REDO launch)
    auto slab_active = [&](int sl) {
        if (sl < 0 || sl >= T) return false;
        if (all_active == 1 || all_active == 2 || T == 1) return true;  // (1: record void; 2: the redo launch)
        return wet[sl] != 0 || (!narrow && ((sl > 0 && wet[sl - 1] != 0) || (sl + 1 < T && wet[sl + 1] != 0))) || slab_has_injector(sl);
    };
    const bool active = slab_active(slab);
    const bool hasPrev = slab_active(slab - 1), hasNext = slab_active(slab + 1);  // (an active neighbour: the one this slab trades rows with)
    const bool first = prow == 0 && hasPrev, last = prow == NROWS - 1 && hasNext;  // the patch rows that talk to a neighbouring slab

    char* tm = team_mem + (size_t)team * team_bytes(T, NY);
    u64* cflg = reinterpret_cast<u64*>(tm);
    u64* pub = reinterpret_cast<u64*>(tm + team_pub_off());
    auto slot = [&](int sl, int par, int dir) { return pub + (((size_t)sl * 2 + par) * 2 + dir) * NY; };
    int failed = 0;

    const float* Sin = Sin_base + (long long)m * S_stride;
    float* Sout = Sout_base + (long long)m * S_stride;
    const double* gVx = p.Vx + (long long)m * (p.Nx + 1) * NY;
    const double* gVy = p.Vy + (long long)m * p.Nx * (NY + 1);
    const double* q = p.q + (long long)(p.q_cols > 1 ? k : 0) * p.Nxy;

    // ---------------- the (at most one) well of this patch
    int wcell = -1;
    double wq = 0.0;
    const int nWl = min(p.nInj + p.nPrd, MAX_WELLS);
    for (int w = 0; w < nWl; ++w) {
        const int cell = p.well_cells[w];
        const int r = cell / NY - gx0;
        if (r >= 0 && r < PX && ((cell % NY) >> 3) == py && q[cell] != 0.0) {
            wcell = cell;
            wq = q[cell];
        }
    }
    const bool has_well = wcell >= 0;
    const int wrow = has_well ? wcell / NY - gx0 : -1, wcol = wcell & (PY - 1);
    const double fpq = fmin(wq, 0.0), fiq = fmax(wq, 0.0);

    // ---------------- CFL: pm = min over cells of pv / (Vi + fi), over the whole member = the team, in fp64      
    const double pv = p.h2 * 1.0;
    double lmin = INFINITY;
#pragma unroll
    for (int i = 0; i < PX; ++i)
#pragma unroll
        for (int j = 0; j < PY; ++j) {
            const long long ix = gx0 + i, iy = iy0 + j;
            const double Vi = fmax(gVx[ix * NY + iy], 0.0) + fmax(gVy[ix * (NY + 1) + iy], 0.0) - fmin(gVx[(ix + 1) * NY + iy], 0.0) -
                              fmin(gVy[ix * (NY + 1) + iy + 1], 0.0);
            lmin = fmin(lmin, pv / (Vi + ((wrow == i && wcol == j) ? fiq : 0.0)));
        }
    double* red = reinterpret_cast<double*>(lds);
    red[tid] = lmin;
    __syncthreads();
    for (int s = NT / 2; s > 0; s >>= 1) {
        if (tid < s) red[tid] = fmin(red[tid], red[tid + s]);
        __syncthreads();
    }
    if (T > 1) {
        if (tid == 0) sat_team::put_double(cflg + slab * 2, cflg + slab * 2 + 1, red[0], 1u);
        if (!active && slab != 0) {  // (slab 0 records the member's sub-step count: it stays for the team minimum)
            for (int i = tid; i < SLAB * NY / 4; i += NT) reinterpret_cast<float4*>(Sout + (long long)slab * SLAB * NY)[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (tid < p.nPrd) {
                const int cell = p.prd_ind[tid];
                if (cell / NY >= slab * SLAB && cell / NY < (slab + 1) * SLAB) prods[((long long)m * p.nTime + k) * p.nPrd + tid] = 0.0f;
            }
            if (tid == 0) wet_out[(size_t)m * T + slab] = 0;
            return;
        }
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
    }
    const double pm = T > 1 ? red[NT] : red[0];
    __syncthreads();
    const double sat = p.swc + p.sor;
    const double cfl = ((1.0 - sat) / 3.0) * pm;
    const double ntsd = ceil(p.dt / cfl);
    const bool bad = !(ntsd >= 1.0 && ntsd <= 1.0e7);
    const int Nts = bad ? 0 : (int)ntsd;
    if (tid == 0 && slab == 0) {
        p.nts[(long long)m * p.nTime + k] = Nts;
        if (bad) atomicOr(&p.status[m], HM_MEMBER_BAD_CFL);
        if (redo) {
            atomicAnd(&p.status[m], ~HM_MEMBER_REDO_STEP);
            if (counters) atomicAdd(counters + 1, 1);  // hm_fwd_slab_redos
        }
    }
    if (!active) {  // slab 0, inactive: as above
        for (int i = tid; i < SLAB * NY / 4; i += NT) reinterpret_cast<float4*>(Sout)[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (tid < p.nPrd) {
            const int cell = p.prd_ind[tid];
            if (cell / NY < SLAB) prods[((long long)m * p.nTime + k) * p.nPrd + tid] = 0.0f;
        }
        if (tid == 0) wet_out[(size_t)m * T] = 0;
        if (__ballot(failed) != 0ull && (tid & 63) == 0) atomicOr(&p.status[m], HM_MEMBER_SYNC_TIMEOUT);
        return;
    }
    const double d = bad ? 0.0 : (p.dt / (double)Nts) / pv;

    // ---------------- the state and its coefficients: fp64 arithmetic on the fp64 fluxes, rounded to float32 once (= the generic kernel)
    float base[PX][PY], dS[PX][PY], fx[PX][PY], fy[PX][PY], cC[PX][PY];
    float fidw = 0.0f;  // fi d of this patch's well (an injector), float32
#pragma unroll
    for (int i = 0; i < PX; ++i) {
        const long long ix = gx0 + i;
        const float4 u = *reinterpret_cast<const float4*>(Sin + ix * NY + iy0), v = *reinterpret_cast<const float4*>(Sin + ix * NY + iy0 + 4);
        base[i][0] = u.x; base[i][1] = u.y; base[i][2] = u.z; base[i][3] = u.w;
        base[i][4] = v.x; base[i][5] = v.y; base[i][6] = v.z; base[i][7] = v.w;
#pragma unroll
        for (int j = 0; j < PY; ++j) {
            const long long iy = iy0 + j;
            const double vxw = gVx[ix * NY + iy], vxe = gVx[(ix + 1) * NY + iy];
            const double vys = gVy[ix * (NY + 1) + iy], vyn = gVy[ix * (NY + 1) + iy + 1];
            const double x1 = fmin(vxw, 0.0), x2 = fmax(vxe, 0.0), y1 = fmin(vys, 0.0), y2 = fmax(vyn, 0.0);
            const double cC64 = d * (((wrow == i && wcol == j) ? fpq : 0.0) + x1 - x2 + y1 - y2);
            cC[i][j] = (float)cC64;
            float fidc = 0.0f;
            if (wrow == i && wcol == j) fidc = fidw = source32(cC64, cC[i][j], fiq, d);  // (sat32.h: the well's source, rounded jointly with its c_C)
            fx[i][j] = (float)(d * vxw);
            fy[i][j] = (float)(d * vys);
            // (sat32.h: a saturated neighbourhood gains nothing; the four off-diagonal coefficients as the sweep forms them from the rounded
            // scaled fluxes: max(fx, 0), max(fy, 0), max(-fx_east, 0), max(-fy_north, 0))
            cC[i][j] = diag32(cC[i][j], fmaxf(-(float)(d * vxe), 0.0f), fmaxf(-(float)(d * vyn), 0.0f), fmaxf(fy[i][j], 0.0f), fmaxf(fx[i][j], 0.0f), fidc, base[i][j]);
            dS[i][j] = 0.0f;
        }
    }

    // halo slots: HW = fw of this patch's row 3 (read by the patch row below as its west halo), HE = east terms for the row above this
    // patch (read by the patch row above).  The first / last patch row of the slab read their halo from an edge slot instead: what they
    // polled from the neighbouring slab, or (at the domain boundary) zeros -- against a zero boundary flux.
    char* own = lds + lt * 16;
    const char* getW = prow > 0 ? own - LPR * 16 : lds + EDGE_BASE + py * 16;
    const char* getE = prow + 1 < NROWS ? own + 2 * CHUNK + LPR * 16 : lds + EDGE_BASE + 4 * 1024 + py * 16;
    constexpr int PAR_W = 4 * CHUNK;  // parity stride of the halo slots
    constexpr int PAR_E = 2 * 1024;   // ... of the edge slots (chunk stride there: 1024)
    if (tid < 64)
#pragma unroll
        for (int c = 0; c < 8; ++c) *reinterpret_cast<float4*>(lds + EDGE_BASE + c * 1024 + tid * 16) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const int cstrW = prow > 0 ? CHUNK : 1024, cstrE = prow + 1 < NROWS ? CHUNK : 1024;
    const int pstrW = prow > 0 ? PAR_W : PAR_E, pstrE = prow + 1 < NROWS ? PAR_W : PAR_E;

    // the injector of this wave (the host admits at most one per wave): its patch row and the eight per-column addends
    // (fi d in the injector's column, 0.0 elsewhere) are wave-uniform, its lane is a mask
    const bool inj = has_well && wq > 0.0;
    const unsigned long long injb = __ballot(inj);
    const int injl = injb ? __ffsll((long long)injb) - 1 : 0;
    const int irow = injb ? __builtin_amdgcn_readlane(wrow, injl) : -1, icol = __builtin_amdgcn_readlane(wcol, injl);
    const float fid = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(fidw), injl));
    float fi[PY];
#pragma unroll
    for (int j = 0; j < PY; ++j) fi[j] = icol == j ? fid : 0.0f;
    int dry;
    {
        unsigned bits = 0u;
#pragma unroll
        for (int i = 0; i < PX; ++i)
#pragma unroll
            for (int j = 0; j < PY; ++j) bits |= __float_as_uint(base[i][j]) << 1;  // -0.0 counts as zero
        dry = p.swc == 0.0 && __ballot(bits != 0u || inj) == 0ull;  // swc > 0: fw(0) != 0, nothing is dry
    }
    auto ff8 = [&](const float (&b)[PY], const float (&e)[PY], float (&f)[PY]) {
#pragma unroll
        for (int j = 0; j < PY; ++j) f[j] = frac_flow<FD>(p, b[j] + e[j]);
    };
    auto ld8 = [&](const char* a, int cstr, float (&f)[PY]) {
        const float4 u = *reinterpret_cast<const float4*>(a), v = *reinterpret_cast<const float4*>(a + cstr);
        f[0] = u.x; f[1] = u.y; f[2] = u.z; f[3] = u.w; f[4] = v.x; f[5] = v.y; f[6] = v.z; f[7] = v.w;
    };
    auto st8 = [&](char* a, int cstr, const float (&f)[PY]) {
        *reinterpret_cast<float4*>(a) = make_float4(f[0], f[1], f[2], f[3]);
        *reinterpret_cast<float4*>(a + cstr) = make_float4(f[4], f[5], f[6], f[7]);
    };
    __syncthreads();

#ifdef HM_SAT_PROF
    long long sprof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long sprof_t = clock64();
    const long long sprof_c0 = sprof_t;
#endif
    // ---------------- explicit sub-steps
    for (int it = 0; it < Nts; ++it) {
        const unsigned tag = (unsigned)it + 2u;
        const int par = it & 1;
        float fc[PY], fm[PY], fn[PY];
        {
            float f3[PY], te[PY];
            // (the record that leaves the workgroup is formed and sent first: its flight overlaps the rest of this phase)
            if (!dry) ff8(base[PX - 1], dS[PX - 1], f3);
            else
#pragma unroll
                for (int j = 0; j < PY; ++j) f3[j] = 0.0f;
            if (last) put8<LPR>(slot(slab, par, 0), py, f3, tag);    // down: the next slab's first patch row wants them
            if (!dry) ff8(base[0], dS[0], fc);
            else
#pragma unroll
                for (int j = 0; j < PY; ++j) fc[j] = 0.0f;
#pragma unroll
            for (int j = 0; j < PY; ++j) te[j] = nmulc(fx[0][j], fc[j]);  // c_E f_E of the cell above, (gx0 - 1, iy0 + j)
            if (first) put8<LPR>(slot(slab, par, 1), py, te, tag);   // up: the previous slab's last patch row wants them
            st8(own + par * PAR_W, CHUNK, f3);
            st8(own + par * PAR_W + 2 * CHUNK, CHUNK, te);
        }
        SPROF(0);
        __syncthreads();
        SPROF(1);
#ifdef HM_SAT_PROF
        sprof[6] += dry;
#endif
        if (first) {  // the previous slab's last fw row -> this patch row's edge slot (thread-private)
            float hw[PY];
            get8<LPR>(slot(slab - 1, par, 0), py, hw, tag, failed);
            st8(lds + EDGE_BASE + par * PAR_E + py * 16, 1024, hw);
        }
        ld8(getW + par * pstrW, cstrW, fm);
        if (dry) {  // the band only changes once something non-zero arrives from just outside it
            float he[PY];
            if (last) {  // (a band that wakes up polls the same record again before its row 3: same tag, same values)
                get8<LPR>(slot(slab + 1, par, 1), py, he, tag, failed);
                st8(lds + EDGE_BASE + 4 * 1024 + par * PAR_E + py * 16, 1024, he);
            }
            ld8(getE + par * pstrE, cstrE, he);
            unsigned o = 0u;
#pragma unroll
            for (int j = 0; j < PY; ++j) o |= __float_as_uint(fm[j]) | __float_as_uint(he[j]);  // fw >= +0, east terms >= +0: bit test
            dry = __ballot(o != 0u) == 0ull;
        }
        SPROF(2);
        if (!dry) {
#pragma unroll
            for (int i = 0; i < PX; ++i) {
                if (i + 2 < PX) ff8(base[i + 1], dS[i + 1], fn);
                else if (i + 2 == PX) ld8(own + par * PAR_W, CHUNK, fn);  // this thread's own row 3, as published
                else {
                    if (last) {  // the next slab's east terms for this patch row's row 3, polled behind three rows of work
                        float he[PY];
                        get8<LPR>(slot(slab + 1, par, 1), py, he, tag, failed);
                        st8(lds + EDGE_BASE + 4 * 1024 + par * PAR_E + py * 16, 1024, he);
                    }
                    ld8(getE + par * pstrE, cstrE, fn);  // row 3: the east TERMS, not fw
                }
                const float fS = prev_lane(fc[PY - 1]);  // f(ix, iy0 - 1): its flux is 0 on the boundary
                const float tN7 = next_lane(nmulc(fy[i][0], fc[0]));  // c_N f_N of column 7: both operands live in the next lane (its column 0)
                float acc[PY];
#pragma unroll
                for (int j = 0; j < PY; ++j) {
                    float a = i + 1 < PX ? nmulc(fx[i + 1 < PX ? i + 1 : 0][j], fn[j]) : fn[j];
                    a = a + (j + 1 < PY ? nmulc(fy[i][j + 1 < PY ? j + 1 : 0], fc[j + 1 < PY ? j + 1 : 0]) : tN7);
                    a = a + cC[i][j] * fc[j];
                    a = a + mulc(fy[i][j], j > 0 ? fc[j > 0 ? j - 1 : 0] : fS);
                    acc[j] = a + mulc(fx[i][j], fm[j]);
                }
                // the injector's row (wave-uniform): its lane adds fi d in its column before the state is updated.  A scalar branch inside
                // the asm: a branch the compiler sees costs the loop its register allocation (sat128r.hip).
                asm volatile("s_cmp_lg_u32 %[ir], %[i]\n\t"
                             "s_cbranch_scc1 .Lsat32s_noinj_%=\n\t"
                             "s_mov_b64 exec, %[m]\n\t"
                             "v_add_f32 %[a0], %[f0], %[a0]\n\t"
                             "v_add_f32 %[a1], %[f1], %[a1]\n\t"
                             "v_add_f32 %[a2], %[f2], %[a2]\n\t"
                             "v_add_f32 %[a3], %[f3], %[a3]\n\t"
                             "v_add_f32 %[a4], %[f4], %[a4]\n\t"
                             "v_add_f32 %[a5], %[f5], %[a5]\n\t"
                             "v_add_f32 %[a6], %[f6], %[a6]\n\t"
                             "v_add_f32 %[a7], %[f7], %[a7]\n\t"
                             "s_mov_b64 exec, -1\n"
                             ".Lsat32s_noinj_%=:"
                             : [a0] "+v"(acc[0]), [a1] "+v"(acc[1]), [a2] "+v"(acc[2]), [a3] "+v"(acc[3]), [a4] "+v"(acc[4]), [a5] "+v"(acc[5]),
                               [a6] "+v"(acc[6]), [a7] "+v"(acc[7])
                             : [ir] "s"(irow), [i] "s"(i), [m] "s"(injb), [f0] "s"(fi[0]), [f1] "s"(fi[1]), [f2] "s"(fi[2]), [f3] "s"(fi[3]),
                               [f4] "s"(fi[4]), [f5] "s"(fi[5]), [f6] "s"(fi[6]), [f7] "s"(fi[7])
                             : "scc");
#pragma unroll
                for (int j = 0; j < PY; ++j) dS[i][j] = dS[i][j] + acc[j];
#pragma unroll
                for (int j = 0; j < PY; ++j) { fm[j] = fc[j]; fc[j] = fn[j]; }
            }
            SPROF(3);
            if ((it & (F32_FOLD - 1)) == F32_FOLD - 1) {
#pragma unroll
                for (int i = 0; i < PX; ++i)
#pragma unroll
                    for (int j = 0; j < PY; ++j) fold32(base[i][j], dS[i][j]);
            }
            SPROF(4);
        }
    }
#ifdef HM_SAT_PROF
    if (blockIdx.x == HM_SAT_PROF && (tid & 63) == 0) {
        sprof[5] = clock64() - sprof_c0;
        sprof[7] = Nts;
        for (int i = 0; i < 8; ++i) hm_sat32_prof_buf[(tid >> 6) * 8 + i] = sprof[i];
    }
#endif

    // ---------------- write back: the state of the next time step is fl(base + dS)
    int nonfinite = 0;
#pragma unroll
    for (int i = 0; i < PX; ++i) {
        float s[PY];
#pragma unroll
        for (int j = 0; j < PY; ++j) {
            s[j] = base[i][j] + dS[i][j];
            nonfinite |= !isfinite(s[j]);
        }
        float* o = Sout + (long long)(gx0 + i) * NY + iy0;
        *reinterpret_cast<float4*>(o) = make_float4(s[0], s[1], s[2], s[3]);
        *reinterpret_cast<float4*>(o + 4) = make_float4(s[4], s[5], s[6], s[7]);
    }
    if (nonfinite) atomicOr(&p.status[m], HM_MEMBER_NONFINITE);
    {   // what the next launch reads: is anything in this slab wet; and towards an inactive neighbour the border row must still be dry
        unsigned wbits = 0u, border = 0u;
#pragma unroll
        for (int i = 0; i < PX; ++i)
#pragma unroll
            for (int j = 0; j < PY; ++j) {
                const unsigned b = __float_as_uint(base[i][j] + dS[i][j]) << 1;
                wbits |= b;
                if ((i == 0 && prow == 0 && slab > 0 && !hasPrev) || (i == PX - 1 && prow == NROWS - 1 && slab + 1 < T && !hasNext)) border |= b;
            }
        if (__ballot(border != 0u) != 0ull && (tid & 63) == 0) atomicOr(&p.status[m], HM_MEMBER_REDO_STEP);  // water reached a slab that sat the step out
        int* wsum = reinterpret_cast<int*>(lds);
        __syncthreads();
        if (tid == 0) wsum[0] = 0;
        __syncthreads();
        if (__ballot(wbits != 0u) != 0ull && (tid & 63) == 0) atomicOr(wsum, 1);
        __syncthreads();
        if (tid == 0) wet_out[(size_t)m * T + slab] = (unsigned char)wsum[0];
    }
    if (__ballot(failed) != 0ull && (tid & 63) == 0) atomicOr(&p.status[m], HM_MEMBER_SYNC_TIMEOUT);
    __threadfence_block();
    __syncthreads();
    if (tid < p.nPrd) {
        const int cell = p.prd_ind[tid];
        if (cell / NY >= slab * SLAB && cell / NY < (slab + 1) * SLAB) prods[((long long)m * p.nTime + k) * p.nPrd + tid] = Sout[cell];
    }
}

template <int NY, bool FD>
int launch(hm_fwd* f, const void* S_in, void* S_out, long long S_stride, int k, int T, int max_teams) {
    const FwdParams& p = f->p;
    // One launch for the whole ensemble (teams of more than one slab): workgroups of slabs that sit the step out leave at once, so the
    // launch costs what the wet slabs cost.  A team's workgroups have consecutive places in their XCD's share of the grid (sat_team.h:
    // team_of_block); with workgroups started in grid order -- observed, not promised by HIP -- a resident workgroup only ever waits for
    // workgroups that are resident or next in line, whatever the number of teams.  Should that order not hold, the bounded spins end the
    // wait, the member is flagged and its step redone by the gated tiled sweep (forward.hip): slower, never wrong.  hm_fwd_set_debug
    // "team_rounds" = 1 restores round 4's form: rounds of as many teams as are resident at once whatever the order.
    const bool rounds = T > 1 && f->dbg_team_rounds != 0;
    const int per_launch = T > 1 && rounds ? max_teams : p.N;
    const size_t need = T > 1 ? team_bytes(T, NY) * (size_t)per_launch : 0;  // a team of one trades nothing
    if (f->team_mem.bytes < need) {
        hm_dev_free(f->team_mem);
        int rc = hm_dev_alloc(f->team_mem, need);
        if (rc) return rc;
    }
    if (!f->retried.p) {
        int rc = hm_dev_alloc(f->retried, 8);
        if (rc) return rc;
        HM_HIP(hipMemsetAsync(f->retried.p, 0, 8, f->ctx->stream));
    }
    const size_t nflags = (size_t)p.N * T;
    if (f->slab_wet.bytes < 2 * nflags) {
        hm_dev_free(f->slab_wet);
        int rc = hm_dev_alloc(f->slab_wet, 2 * nflags);
        if (rc) return rc;
        f->slab_wet_step = -1;
    }
    // the record of wet slabs written by the launch of time index k - 1 is good for this one if nothing touched the plan's inputs or state since
    const bool record_void = !(f->slab_wet_step == k && f->slab_wet_gen == f->inputs_gen) || p.swc != 0.0 || f->raw_state_exposed;  // (swc > 0: fw(0) != 0, no slab is inert)
    // 0: slabs sit out by the record; 1: everybody takes part; 3: by the record WITHOUT the margin of one slab around the wet ones (a test knob:
    // the front then crosses into a sitting-out slab within a few steps, which the border check flags and the REDO launch below repairs)
    const int all_active = record_void ? 1 : (f->dbg_slab_margin == 0 ? 3 : 0);
    const unsigned char* wet_in = (const unsigned char*)f->slab_wet.p + (size_t)(k & 1) * nflags;
    unsigned char* wet_out = (unsigned char*)f->slab_wet.p + (size_t)((k + 1) & 1) * nflags;
    hipStream_t s = f->ctx->stream;
    auto kern = k_sat32s<NY, FD>;
    HM_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
    if (T > 1) {
        int resident = 0;  // the runtime's own answer: can a workgroup of this kernel be resident on a CU at all?
        HM_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, kern, NT, LDS_BYTES));
        if (resident < 1) return -1;
    }
    for (int first = 0; first < p.N; first += per_launch) {
        const int nteams = std::min(per_launch, p.N - first);
        const int used_per_xcd = (nteams + 7) / 8;
        if (T > 1) HM_HIP(hipMemsetAsync(f->team_mem.p, 0, team_bytes(T, NY) * (size_t)nteams, s));  // tags restart at 0 every launch
        hipLaunchKernelGGL(kern, dim3(8 * used_per_xcd * T), dim3(NT), LDS_BYTES, s, f->p, (const float*)S_in, (float*)S_out, S_stride,
                           (float*)f->prods.p, k, (char*)f->team_mem.p, T, first, wet_in, wet_out, all_active, (int*)f->retried.p);
    }
    if (T > 1 && all_active != 1) {  // slabs may have sat out: the gated redo launch (only flagged members do anything)
        for (int first = 0; first < p.N; first += per_launch) {
            const int nteams = std::min(per_launch, p.N - first);
            const int used_per_xcd = (nteams + 7) / 8;
            HM_HIP(hipMemsetAsync(f->team_mem.p, 0, team_bytes(T, NY) * (size_t)nteams, s));
            hipLaunchKernelGGL(kern, dim3(8 * used_per_xcd * T), dim3(NT), LDS_BYTES, s, f->p, (const float*)S_in, (float*)S_out, S_stride,
                               (float*)f->prods.p, k, (char*)f->team_mem.p, T, first, wet_in, wet_out, 2, (int*)f->retried.p);
        }
    }
    HM_HIP(hipGetLastError());
    f->slab_wet_step = k + 1;
    f->slab_wet_gen = f->inputs_gen;
    return 0;
}

}  // namespace

#ifdef HM_SAT_PROF
extern "C" int hm_debug_sat32_prof(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(hm_sat32_prof_buf), sizeof(long long) * 64); }
#endif

// Hardware self-test of the float32 fractional flow (fracflow.h): frac_flow<true> as the sweeps compute it -- v_rcp_f32, q = n r, one residual
// correction -- against the compiler's IEEE division on EVERY float s (the function has one operand: 2^32 cases, a second of GPU time).
// out[0]: cases with |s| < 2^62 whose bits differ (the claim is 0: that is the bit-for-bit equality with the NumPy float32 specification);
// out[1]: cases that differ anywhere (operands beyond 6.5e18, where s^2 + (1 - s)^2 leaves the range the short form is exact on).
namespace {
__global__ __launch_bounds__(256) void k_fracflow32_check(FwdParams p, unsigned long long* out) {
    const unsigned long long gid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x, stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long bad = 0, bad_in = 0;
    for (unsigned long long b = gid; b < (1ull << 32); b += stride) {
        const float s = __uint_as_float((unsigned)b);
        const float got = frac_flow<true>(p, s), want = frac_flow_ieee<true>(p, s);
        if (__float_as_uint(got) != __float_as_uint(want) && !(got != got && want != want)) {
            ++bad;
            if (fabsf(s) < 4.611686e18f) ++bad_in;
        }
    }
    if (bad_in) atomicAdd(out, bad_in);
    if (bad) atomicAdd(out + 1, bad);
}
}  // namespace

// The same for the fp64 fractional flow (fracflow.h: v_rcp_f64, ONE cubic refinement of the reciprocal, the quotient and one residual correction --
// seven instructions against the compiler's eleven).  A double has 2^64 bit patterns: not all of them can be tried, so the claim "the IEEE quotient,
// bit for bit" is pinned on the operands a saturation can take, densely: (i) 2^33 values spread over [0, 1 + 2^-9) by a 64-bit hash, every one with
// a full 52-bit mantissa; (ii) the same count next to 0, 1/2 and 1 (|s - c| = 2^-k u, k = 0..63: where 1 - s, s^2 and the denominator cancel or
// lose bits); (iii) every binade from 2^-1074 up to 2^-1 with 2^18 mantissas each (the doubly exponentially small values ahead of the front,
// denormals included: fracflow.h's |S| < 2^-480 case).  out[0] = operands whose result differs from the compiler's IEEE division in any bit
// (claimed: 0), out[1] = operands tried.
namespace {
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {  // splitmix64 finaliser
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
__global__ __launch_bounds__(256) void k_fracflow64_check(FwdParams p, unsigned long long* out, unsigned long long per_class) {
    const unsigned long long gid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x, stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long bad = 0, tried = 0;
    auto check = [&](double s) {
        const double got = frac_flow<true>(p, s), want = frac_flow_ieee<true>(p, s);
        if (__double_as_longlong(got) != __double_as_longlong(want) && !(got != got && want != want)) ++bad;
        ++tried;
    };
    for (unsigned long long i = gid; i < per_class; i += stride) {
        const unsigned long long h = mix64(i), h2 = mix64(h);
        const double u = (double)(h >> 11) * 0x1.0p-53;  // [0, 1), 53 random bits
        check(u * (1.0 + 0x1.0p-9));                                                        // (i)
        const int k = (int)(h2 & 63), which = (int)((h2 >> 6) % 3);                         // (ii)
        const double c = which == 0 ? 0.0 : which == 1 ? 0.5 : 1.0;
        const double dlt = ldexp(u, -k);
        check(which == 0 ? dlt : ((h2 >> 8) & 1) ? c + dlt * 0x1.0p-10 : c - dlt * 0x1.0p-1);
    }
    for (unsigned long long i = gid; i < (1075ull << 18); i += stride) {                     // (iii)
        const int e = (int)(i >> 18);  // binade: value in [2^(e - 1075), 2^(e - 1074)); e = 0: the denormals
        const unsigned long long man = mix64(i) & ((1ull << 52) - 1);
        const unsigned long long bits = e == 0 ? man : (((unsigned long long)e) << 52) | man;
        check(__longlong_as_double((long long)bits));
    }
    if (bad) atomicAdd(out, bad);
    atomicAdd(out + 1, tried);
}
}  // namespace

extern "C" int hm_debug_fracflow64_check(hm_ctx* ctx, unsigned long long* out) {
    HM_REQUIRE(ctx && out, "hm_debug_fracflow64_check: NULL argument");
    HM_HIP(hipSetDevice(ctx->device));
    unsigned long long* d;
    HM_HIP(hipMalloc(&d, 16));
    HM_HIP(hipMemsetAsync(d, 0, 16, ctx->stream));
    FwdParams p{};
    hipLaunchKernelGGL(k_fracflow64_check, dim3(8 * ctx->num_cu), dim3(256), 0, ctx->stream, p, d, 1ull << 33);
    HM_HIP(hipGetLastError());
    HM_HIP(hipStreamSynchronize(ctx->stream));
    HM_HIP(hipMemcpy(out, d, 16, hipMemcpyDeviceToHost));
    (void)hipFree(d);
    return 0;
}

extern "C" int hm_debug_fracflow32_check(hm_ctx* ctx, unsigned long long* out) {
    HM_REQUIRE(ctx && out, "hm_debug_fracflow32_check: NULL argument");
    HM_HIP(hipSetDevice(ctx->device));
    unsigned long long* d;
    HM_HIP(hipMalloc(&d, 16));
    HM_HIP(hipMemsetAsync(d, 0, 16, ctx->stream));
    FwdParams p{};
    hipLaunchKernelGGL(k_fracflow32_check, dim3(8 * ctx->num_cu), dim3(256), 0, ctx->stream, p, d);
    HM_HIP(hipGetLastError());
    HM_HIP(hipStreamSynchronize(ctx->stream));
    HM_HIP(hipMemcpy(out, d, 16, hipMemcpyDeviceToHost));
    (void)hipFree(d);
    return 0;
}

// Returns 0 if launched, >0 on error, -1 if this specialisation does not apply.
int launch_saturation_32s(hm_fwd* f, const void* S_in, void* S_out, long long S_stride, int k) {
    const FwdParams& p = f->p;
    if (p.q_mstride != 0) return -1;  // per-member wells: the well cells come from one shared well list
    if (f->dtype != 32 || p.por != nullptr) return -1;
    if (p.Ny != 128 && p.Ny != 256 && p.Ny != 512) return -1;
    const int slab = 16384 / p.Ny;  // rows per workgroup
    if (p.Nx % slab != 0) return -1;
    const int T = p.Nx / slab, slots = f->ctx->num_cu / 8;
    if (T > MAX_SLABS || (T > 1 && T > slots)) return -1;
    const int max_teams = T > 1 ? 8 * (slots / T) : p.N;  // a team of one needs nobody resident beside it
    if ((int)f->well_cells_host.size() > MAX_WELLS) return -1;
    // at most one well per 4 x 8 patch; at most one injector (a well with q > 0 in this time column) per wave = per band of 4 * (512 / Ny) rows
    const double* qk = f->q_host.data() + (size_t)(p.q_cols > 1 ? k : 0) * p.Nxy;
    std::vector<long long> patches;
    std::vector<int> bands;
    for (int cell : f->well_cells_host) {
        const long long id = (long long)((cell / p.Ny) >> 2) * 100000 + ((cell % p.Ny) >> 3);
        for (long long s : patches)
            if (s == id) return -1;
        patches.push_back(id);
        if (qk[cell] > 0.0) {
            const int band = (cell / p.Ny) / (PX * 512 / p.Ny);
            for (int b : bands)
                if (b == band) return -1;
            bands.push_back(band);
        }
    }
#define HM_SAT32S(NYV) (p.fluid_default ? launch<NYV, true>(f, S_in, S_out, S_stride, k, T, max_teams) : launch<NYV, false>(f, S_in, S_out, S_stride, k, T, max_teams))
    return p.Ny == 128 ? HM_SAT32S(128) : p.Ny == 256 ? HM_SAT32S(256) : HM_SAT32S(512);
#undef HM_SAT32S
}
