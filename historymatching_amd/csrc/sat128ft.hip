// sat128ft.hip -- fp32 explicit upwind saturation sweep for grids made of 128 x 128-cell tiles (256 x 256,
// 512 x 512 ...) in plans created with dtype = 32: the fp32 twin of sat128t.hip = sat128f.hip per tile + the team
// hand-off of sat_team.h.  One workgroup per tile keeps S and the five upwind coefficients of its 128 x 128 cells in
// registers (formed once per launch in fp64 from the fp64 face fluxes and rounded to fp32 exactly like the generic
// kernel), exchanges the fractional flow of the current sub-step through a (128 + 2) x 128 LDS image -- rows 128/129 are
// the west/east halo -- plus two 128-entry halo columns for south/north, and trades tile edges with the neighbouring
// tiles' workgroups once per sub-step: one granule per value (a float is the 32-bit payload of one granule).
// Bit-identical to k_saturation_generic<float> / k_saturation_tiled<float>.  Compiled with -ffp-contract=off.
#include "sat_team.h"
#include "fracflow.h"

namespace {

using namespace sat_team;

constexpr int PX = 8, PY = 4;
constexpr int NPY = TS / PY;           // 32 patches along iy
constexpr int NT = (TS / PX) * NPY;    // 512 threads
constexpr int FW_FLOATS = (TS + 2) * TS;   // rows 0..127 of the tile, 128 = west halo, 129 = east halo
constexpr int REC_FLOATS = 8;          // S, cE, cN, cC, cS, cW, fid, fw
constexpr int MAX_WELLS = 16;
constexpr int REC_BASE = FW_FLOATS;                             // float indices
constexpr int EDGE_BASE = REC_BASE + (MAX_WELLS + 2) * REC_FLOATS;  // south halo column [128], north halo column [128]
constexpr int CC_BASE = EDGE_BASE + 2 * TS;                     // the diagonal coefficient c_C, thread-private: chunk i of thread t at CC_BASE + (i * NT + t) * 4
                                                                // (consecutive lanes, consecutive 16-byte chunks: conflict-free b128) -- 64 KB of the
                                                                // LDS the tile does not otherwise use, for 32 registers of a register-full loop
                                                                // (round 4: 264 -> 255 ms per 125-member step at 512 x 512)
constexpr int MISC_BASE = CC_BASE + PX * NT * PY;               // team CFL minima (32 doubles)
constexpr int LDS_BYTES = MISC_BASE * 4 + MAX_TILES * 8;
static_assert(MISC_BASE % 2 == 0, "double alignment");

__device__ __forceinline__ float next_lane(float v) {  // value of lane+1 (0 past the wave)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));
}
__device__ __forceinline__ float prev_lane(float v) {  // value of lane-1
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));
}

// float index of fw at tile-local (lix, liy), lix/liy in [-1, 128]: outside the tile -> the halo
__device__ __forceinline__ int tile_at(int lix, int liy) {
    if (liy < 0) return EDGE_BASE + min(max(lix, 0), TS - 1);
    if (liy >= TS) return EDGE_BASE + TS + min(max(lix, 0), TS - 1);
    if (lix < 0) return TS * TS + liy;
    if (lix >= TS) return (TS + 1) * TS + liy;
    return lix * TS + liy;
}

template <bool FD>
__global__ __launch_bounds__(NT) void k_sat128ft(FwdParams p, const float* __restrict__ Sin_base, float* __restrict__ Sout_base,
                                                 long long S_stride, float* __restrict__ prods, int k, char* team_mem, int TXn,
                                                 int TYn, int first_member) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    __shared__ int dead_word;  // set once a wait has timed out: the workgroup stops waiting
    int* dead = &dead_word;
    float* fwf = reinterpret_cast<float*>(lds);
    float* recs = fwf + REC_BASE;
    double* team_min = reinterpret_cast<double*>(fwf + MISC_BASE);

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

    const TeamLayout<1> lay{T};
    char* tm = team_mem + (size_t)team * lay.bytes();
    u64* cflg = reinterpret_cast<u64*>(tm + lay.cfl_off());
    u64* pub = reinterpret_cast<u64*>(tm + lay.pub_off());

    const int py = tid & (NPY - 1), px = tid >> 5;
    const int ix0 = px * PX, iy0 = py * PY;
    const bool isS = py == 0, isN = py == NPY - 1;
    const float* Sin = Sin_base + (long long)m * S_stride;
    float* Sout = Sout_base + (long long)m * S_stride;
    const double* gVx = p.Vx + (long long)m * (p.Nx + 1) * Ny;
    const double* gVy = p.Vy + (long long)m * p.Nx * (Ny + 1);
    const double* q = p.q + (long long)(p.q_cols > 1 ? k : 0) * p.Nxy;

    // halos without a neighbour stay 0 (their coefficients are 0: boundary faces carry no flux)
    for (int i = tid; i < 2 * TS; i += NT) {
        fwf[TS * TS + i] = 0.0f;
        fwf[EDGE_BASE + i] = 0.0f;
    }
    if (tid == 0) dead_word = 0;
    int ev = 0;  // events published by this tile so far (identical sequence in every tile of the team)

    float S[PX][PY];
#pragma unroll
    for (int i = 0; i < PX; ++i) {
        const float4 v = *reinterpret_cast<const float4*>(Sin + (long long)(gx0 + ix0 + i) * Ny + gy0 + iy0);
        S[i][0] = v.x; S[i][1] = v.y; S[i][2] = v.z; S[i][3] = v.w;
    }

    // the (at most one) well of this patch
    int wlx = -1, wly = -1, wcell = -1, wslot = MAX_WELLS;  // non-owners work on a dummy record
    double wq = 0.0;
    const int nW = min(p.nInj + p.nPrd, MAX_WELLS);
    for (int w = 0; w < nW; ++w) {
        const int cell = p.well_cells[w];
        const int lx = cell / Ny - gx0, ly = cell % Ny - gy0;
        if (lx >= 0 && lx < TS && ly >= 0 && ly < TS && (lx >> 3) == px && (ly >> 2) == py && q[cell] != 0.0) {
            wcell = cell;
            wlx = lx;
            wly = ly;
            wq = q[cell];
            wslot = w;
        }
    }
    const bool has_well = wcell >= 0;

    // CFL: pm = min over cells of pv / (Vi + fi), in fp64                                  
    const double pv = p.h2 * 1.0;
    double lmin = INFINITY;
#pragma unroll
    for (int i = 0; i < PX; ++i)
#pragma unroll
        for (int j = 0; j < PY; ++j) {
            const long long ix = gx0 + ix0 + i, iy = gy0 + iy0 + j;
            const double Vi = fmax(gVx[ix * Ny + iy], 0.0) + fmax(gVy[ix * (Ny + 1) + iy], 0.0) -
                              fmin(gVx[(ix + 1) * Ny + iy], 0.0) - fmin(gVy[ix * (Ny + 1) + iy + 1], 0.0);
            const double fi = (has_well && ix * Ny + iy == wcell) ? fmax(wq, 0.0) : 0.0;
            lmin = fmin(lmin, pv / (Vi + fi));
        }
    double* red = reinterpret_cast<double*>(lds);
    red[tid] = lmin;
    __syncthreads();
    for (int s = NT / 2; s > 0; s >>= 1) {
        if (tid < s) red[tid] = fmin(red[tid], red[tid + s]);
        __syncthreads();
    }
    // team-wide minimum: one all-tiles event
    if (tid == 0) put_double(cflg + tile * 2, cflg + tile * 2 + 1, red[0], ev + 1);
    if (tid < 64) {   // wave 0: lane t collects tile t's minimum
        const int t = tid < T ? tid : 0;
        double v = INFINITY;
        if (!get_double(cflg + t * 2, cflg + t * 2 + 1, ev + 1, v, dead)) v = INFINITY;
        if (tid < T) team_min[tid] = v;
    }
    ++ev;
    __syncthreads();
    double pm = team_min[0];
    for (int t = 1; t < T; ++t) pm = fmin(pm, team_min[t]);
    __syncthreads();
    const double cfl = ((1.0 - (p.swc + p.sor)) / 3.0) * pm;
    const double ntsd = ceil(p.dt / cfl);
    const bool bad = !(ntsd >= 1.0 && ntsd <= 1.0e7);
    const int Nts = bad ? 0 : (int)ntsd;
    if (tid == 0 && tile == 0) {
        p.nts[(long long)m * p.nTime + k] = Nts;
        if (bad) atomicOr(&p.status[m], HM_MEMBER_BAD_CFL);
    }
    const double d = bad ? 0.0 : (p.dt / (double)Nts) / pv;

    // upwind coefficients of the own cells: fp64 arithmetic on the fp64 fluxes, rounded to fp32 once (= the generic kernel)
    float cE[PX][PY], cN[PX][PY], cS[PX][PY], cW[PX][PY];
    float* ccl = fwf + CC_BASE + tid * PY;
#pragma unroll
    for (int i = 0; i < PX; ++i)
#pragma unroll
        for (int j = 0; j < PY; ++j) {
            const long long ix = gx0 + ix0 + i, iy = gy0 + iy0 + j;
            const double vxw = gVx[ix * Ny + iy], vxe = gVx[(ix + 1) * Ny + iy];
            const double vys = gVy[ix * (Ny + 1) + iy], vyn = gVy[ix * (Ny + 1) + iy + 1];
            const double x1 = fmin(vxw, 0.0), x2 = fmax(vxe, 0.0), y1 = fmin(vys, 0.0), y2 = fmax(vyn, 0.0);
            ccl[i * NT * PY + j] = (float)(d * (0.0 + x1 - x2 + y1 - y2));
            cW[i][j] = (float)(d * fmax(vxw, 0.0));
            cE[i][j] = (float)(d * (-fmin(vxe, 0.0)));
            cS[i][j] = (float)(d * fmax(vys, 0.0));
            cN[i][j] = (float)(d * (-fmin(vyn, 0.0)));
        }
    // well record (exact coefficients including the source terms); the dummy record is all zeros
    if (tid < 2 * REC_FLOATS) recs[MAX_WELLS * REC_FLOATS + tid] = 0.0f;
    __syncthreads();
    float* rec = recs + wslot * REC_FLOATS;
    if (has_well) {
        const long long wix = gx0 + wlx, wiy = gy0 + wly;
        const double vxw = gVx[wix * Ny + wiy], vxe = gVx[(wix + 1) * Ny + wiy];
        const double vys = gVy[wix * (Ny + 1) + wiy], vyn = gVy[wix * (Ny + 1) + wiy + 1];
        const double fpq = fmin(wq, 0.0), fiq = fmax(wq, 0.0);
        const double x1 = fmin(vxw, 0.0), x2 = fmax(vxe, 0.0), y1 = fmin(vys, 0.0), y2 = fmax(vyn, 0.0);
        rec[0] = Sin[wcell];
        rec[1] = (float)(d * (-fmin(vxe, 0.0)));
        rec[2] = (float)(d * (-fmin(vyn, 0.0)));
        rec[3] = (float)(d * (fpq + x1 - x2 + y1 - y2));
        rec[4] = (float)(d * fmax(vys, 0.0));
        rec[5] = (float)(d * fmax(vxw, 0.0));
        rec[6] = (float)(fiq * d);
    }
    // float index of the well cell and its 4 neighbours in the fw image / halo (threads without a well: a dummy slot)
    const int dummy = REC_BASE + (MAX_WELLS + 1) * REC_FLOATS;
    auto well_at = [&](int dx, int dy) { return has_well ? tile_at(wlx + dx, wly + dy) : dummy; };
    __syncthreads();

    const int ixW = ix0 > 0 ? ix0 - 1 : (hasW ? TS : 0), ixE = ix0 + PX < TS ? ix0 + PX : (hasE ? TS + 1 : TS - 1);
    const float* edge_row0 = fwf + EDGE_BASE + (isN ? TS : 0) + ix0;  // halo column entry of this patch's first row
    auto load_row = [&](int ix, float (&f)[PY]) {
        const float4 v = *reinterpret_cast<const float4*>(fwf + ix * TS + iy0);
        f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    };

    for (int it = 0; it < Nts; ++it, ++ev) {
        // phase A: fractional flow of every own cell -> LDS
#pragma unroll
        for (int i = 0; i < PX; ++i) {
            float4 v;
            v.x = frac_flow<FD>(p, S[i][0]);
            v.y = frac_flow<FD>(p, S[i][1]);
            v.z = frac_flow<FD>(p, S[i][2]);
            v.w = frac_flow<FD>(p, S[i][3]);
            *reinterpret_cast<float4*>(fwf + (ix0 + i) * TS + iy0) = v;
        }
        {   // well side path, branch-free (threads without a well run it on the dummy record)
            const float wf = frac_flow<FD>(p, rec[0]);
            rec[7] = wf;
            fwf[well_at(0, 0)] = wf;  // after this thread's own row write: ordered
        }
        __syncthreads();
        // hand-off: publish this tile's edges (4 x 128 values, one per thread; a well on the edge already carries its exact fw
        // in LDS) and collect the neighbours' edges into the LDS halo
        {
            int t = tid;
            asm volatile("" : "+v"(t));  // roles re-derived here: no registers held across the sweep for them
            const int e = t >> 7, idx = t & (TS - 1);   // edge 0 W, 1 E, 2 S, 3 N
            const int src = e == 0 ? idx : e == 1 ? (TS - 1) * TS + idx : e == 2 ? idx * TS : idx * TS + TS - 1;
            const int dst = e == 0 ? TS * TS + idx : e == 1 ? (TS + 1) * TS + idx : EDGE_BASE + (e - 2) * TS + idx;
            const bool has = e == 0 ? hasW : e == 1 ? hasE : e == 2 ? hasS : hasN;
            const int nb = e == 0 ? tile - TYn : e == 1 ? tile + TYn : e == 2 ? tile - 1 : tile + 1;
            put_granule(pub + ((size_t)(tile * 2 + (ev & 1)) * 4 + e) * TS + idx, __float_as_uint(fwf[src]), ev + 1);
            if (has) {
                float v;
                if (get_float(pub + ((size_t)(nb * 2 + (ev & 1)) * 4 + (e ^ 1)) * TS + idx, ev + 1, v, dead)) fwf[dst] = v;
            }
        }
        __syncthreads();

        // phase B: upwind update, a sliding window of three fw rows
        float fp[PY], fc[PY], fn[PY];
        load_row(ixW, fp);
        load_row(ix0, fc);
#pragma unroll
        for (int i = 0; i < PX; ++i) {
            load_row(i + 1 < PX ? ix0 + i + 1 : ixE, fn);
            const float eh = edge_row0[i];
            const float4 cc4 = *reinterpret_cast<const float4*>(ccl + i * NT * PY);
            const float cC[PY] = {cc4.x, cc4.y, cc4.z, cc4.w};
            const float fSd = prev_lane(fc[PY - 1]);  // fw(ix, iy0 - 1): its coefficient is 0 on the boundary
            const float fNd = next_lane(fc[0]);       // fw(ix, iy0 + PY)
            const float fS = isS ? eh : fSd;
            const float fN = isN ? eh : fNd;
#pragma unroll
            for (int j = 0; j < PY; ++j) {
                const float fs = j > 0 ? fc[j > 0 ? j - 1 : 0] : fS;
                const float fnn = j + 1 < PY ? fc[j + 1 < PY ? j + 1 : 0] : fN;
                float acc = cE[i][j] * fn[j];
                acc = acc + cN[i][j] * fnn;
                acc = acc + cC[j] * fc[j];
                acc = acc + cS[i][j] * fs;
                acc = acc + cW[i][j] * fp[j];
                S[i][j] = S[i][j] + acc;
            }
#pragma unroll
            for (int j = 0; j < PY; ++j) {
                fp[j] = fc[j];
                fc[j] = fn[j];
            }
        }
        {
            float acc = rec[1] * fwf[well_at(1, 0)];
            acc = acc + rec[2] * fwf[well_at(0, 1)];
            acc = acc + rec[3] * rec[7];
            acc = acc + rec[4] * fwf[well_at(0, -1)];
            acc = acc + rec[5] * fwf[well_at(-1, 0)];
            rec[0] = rec[0] + (acc + rec[6]);
        }
        __syncthreads();
    }

    // write back
    int nonfinite = 0;
#pragma unroll
    for (int i = 0; i < PX; ++i) {
        float4 v;
        v.x = S[i][0]; v.y = S[i][1]; v.z = S[i][2]; v.w = S[i][3];
        const long long c0 = (long long)(gx0 + ix0 + i) * Ny + gy0 + iy0;
        *reinterpret_cast<float4*>(Sout + c0) = v;
#pragma unroll
        for (int j = 0; j < PY; ++j) nonfinite |= (c0 + j != wcell) && !isfinite(S[i][j]);
    }
    if (has_well) {
        const float wS = rec[0];
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
    const TeamLayout<1> lay{T};
    const size_t need = lay.bytes() * (size_t)max_teams;
    if (f->team_mem.bytes < need) {
        hm_dev_free(f->team_mem);
        int rc = hm_dev_alloc(f->team_mem, need);
        if (rc) return rc;
    }
    hipStream_t s = f->ctx->stream;
    auto kern = k_sat128ft<FD>;
    const int lds_req = LDS_BYTES;
    HM_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_req));
    int resident = 0;  // the runtime's own answer: can a workgroup of this kernel be resident on a CU at all?
    HM_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, kern, NT, lds_req));
    if (resident < 1) return -1;  // no: the caller falls back to the single-workgroup tiled sweep
    for (int first = 0; first < p.N; first += max_teams) {
        const int nteams = std::min(max_teams, p.N - first);
        const int used_per_xcd = (nteams + 7) / 8;
        HM_HIP(hipMemsetAsync(f->team_mem.p, 0, lay.bytes() * (size_t)nteams, s));  // tags restart at 0 every launch
        hipLaunchKernelGGL(kern, dim3(8 * used_per_xcd * T), dim3(NT), lds_req, s, f->p, (const float*)S_in, (float*)S_out, S_stride,
                           (float*)f->prods.p, k, (char*)f->team_mem.p, TXn, TYn, first);
    }
    HM_HIP(hipGetLastError());
    return 0;
}

}  // namespace

// Returns 0 if launched, >0 on error, -1 if this specialisation does not apply.
int launch_saturation_128ft(hm_fwd* f, const void* S_in, void* S_out, long long S_stride, int k) {
    const FwdParams& p = f->p;
    int TXn, TYn, max_teams;
    if (p.q_mstride != 0) return -1;  // per-member wells: the well side path works from one shared well list
    if (f->dtype != 32 || p.por != nullptr || !tiles_of(f, TXn, TYn, max_teams) || !wells_fit_patches(f, MAX_WELLS)) return -1;
    return p.fluid_default ? launch<true>(f, S_in, S_out, S_stride, k, TXn, TYn, max_teams)
                           : launch<false>(f, S_in, S_out, S_stride, k, TXn, TYn, max_teams);
}
