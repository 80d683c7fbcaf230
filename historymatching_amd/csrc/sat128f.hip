// sat128f.hip -- 128x128 fp32 specialisation of the explicit upwind saturation sweep for plans created
// with dtype = 32 (saturation storage and arithmetic in fp32, pressure and fluxes fp64, Nts in fp64): the fp32 twin of
// sat128.hip, bit-identical to k_saturation_generic<float>.
//
// One workgroup (512 threads) = one member, resident for all Nts sub-steps; every thread owns an 8 (ix) x 4 (iy) patch.
// In fp32 the whole per-member state S + the five upwind coefficients (6 x 64 KB) fits the register file, so the
// coefficients are formed ONCE per launch -- in fp64 from the fp64 face fluxes, then rounded to fp32 exactly like the
// generic kernel (`(T)(d * ...)`) -- instead of being re-derived every sub-step as the fp64 kernel must.  The
// fractional-flow field of the current sub-step is exchanged through a 64 KB LDS image (one float4 per thread and row:
// conflict-free without swizzling), the iy halo through DPP wave shifts.  Per cell and sub-step: fw (one fp32 division),
// 5 multiplies, 5 adds.  Wells: branch-free side path on a small LDS record, as in sat128.hip.
// Summation order = CSR row order of the reference's matrix form (E, N, C, S, W), compiled with -ffp-contract=off.
#include "fracflow.h"

namespace {

constexpr int N128 = 128;
constexpr int PX = 8, PY = 4;
constexpr int NPY = N128 / PY;          // 32 patches along iy
constexpr int NT = (N128 / PX) * NPY;   // 512 threads
constexpr int FW_BYTES = N128 * N128 * 4;
constexpr int REC_FLOATS = 8;           // S, cE, cN, cC, cS, cW, fid, fw
constexpr int MAX_WELLS = 16;
constexpr int LDS_BYTES = FW_BYTES + (MAX_WELLS + 2) * REC_FLOATS * 4 + 64;

__device__ __forceinline__ float next_lane(float v) {  // value of lane+1 (0 past the wave)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));
}
__device__ __forceinline__ float prev_lane(float v) {  // value of lane-1
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));
}

template <bool FD>
__global__ __launch_bounds__(NT) void k_sat128f(FwdParams p, const float* __restrict__ Sin_base, float* __restrict__ Sout_base,
                                                long long S_stride, float* __restrict__ prods, int k) {
    extern __shared__ __attribute__((aligned(16))) char lds[];  // [0, 64K) fw field; then well records
    float* fwf = reinterpret_cast<float*>(lds);
    float* recs = reinterpret_cast<float*>(lds + FW_BYTES);

    const int tid = threadIdx.x, m = blockIdx.x;
    const int py = tid & (NPY - 1), px = tid >> 5;
    const int ix0 = px * PX, iy0 = py * PY;
    const float* Sin = Sin_base + (long long)m * S_stride;
    float* Sout = Sout_base + (long long)m * S_stride;
    const double* gVx = p.Vx + (long long)m * (N128 + 1) * N128;
    const double* gVy = p.Vy + (long long)m * N128 * (N128 + 1);
    const double* q = p.q + (long long)(p.q_cols > 1 ? k : 0) * p.Nxy;

    float S[PX][PY];
#pragma unroll
    for (int i = 0; i < PX; ++i) {
        const float4 v = *reinterpret_cast<const float4*>(Sin + (ix0 + i) * N128 + iy0);
        S[i][0] = v.x; S[i][1] = v.y; S[i][2] = v.z; S[i][3] = v.w;
    }

    // the (at most one) well of this patch
    int wcell = -1, wslot = MAX_WELLS;  // non-owners work on a dummy record
    double wq = 0.0;
    const int nW = min(p.nInj + p.nPrd, MAX_WELLS);
    for (int w = 0; w < nW; ++w) {
        const int cell = p.well_cells[w];
        if (((cell >> 7) >> 3) == px && ((cell & 127) >> 2) == py && q[cell] != 0.0) {
            wcell = cell;
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
            const int ix = ix0 + i, iy = iy0 + j;
            const double Vi = fmax(gVx[ix * N128 + iy], 0.0) + fmax(gVy[ix * (N128 + 1) + iy], 0.0) -
                              fmin(gVx[(ix + 1) * N128 + iy], 0.0) - fmin(gVy[ix * (N128 + 1) + iy + 1], 0.0);
            const double fi = (has_well && ix * N128 + iy == wcell) ? fmax(wq, 0.0) : 0.0;
            lmin = fmin(lmin, pv / (Vi + fi));
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
    const double cfl = ((1.0 - (p.swc + p.sor)) / 3.0) * pm;
    const double ntsd = ceil(p.dt / cfl);
    const bool bad = !(ntsd >= 1.0 && ntsd <= 1.0e7);
    const int Nts = bad ? 0 : (int)ntsd;
    if (tid == 0) {
        p.nts[(long long)m * p.nTime + k] = Nts;
        if (bad) atomicOr(&p.status[m], HM_MEMBER_BAD_CFL);
    }
    const double d = bad ? 0.0 : (p.dt / (double)Nts) / pv;

    // upwind coefficients of the own cells: fp64 arithmetic on the fp64 fluxes, rounded to fp32 once (= the generic kernel)
    float cE[PX][PY], cN[PX][PY], cC[PX][PY], cS[PX][PY], cW[PX][PY];
#pragma unroll
    for (int i = 0; i < PX; ++i)
#pragma unroll
        for (int j = 0; j < PY; ++j) {
            const int ix = ix0 + i, iy = iy0 + j;
            const double vxw = gVx[ix * N128 + iy], vxe = gVx[(ix + 1) * N128 + iy];
            const double vys = gVy[ix * (N128 + 1) + iy], vyn = gVy[ix * (N128 + 1) + iy + 1];
            const double x1 = fmin(vxw, 0.0), x2 = fmax(vxe, 0.0), y1 = fmin(vys, 0.0), y2 = fmax(vyn, 0.0);
            cC[i][j] = (float)(d * (0.0 + x1 - x2 + y1 - y2));
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
        const int wix = wcell >> 7, wiy = wcell & 127;
        const double vxw = gVx[wix * N128 + wiy], vxe = gVx[(wix + 1) * N128 + wiy];
        const double vys = gVy[wix * (N128 + 1) + wiy], vyn = gVy[wix * (N128 + 1) + wiy + 1];
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
    // float index of the well cell and its 4 neighbours in the fw image (threads without a well: a dummy slot)
    const int dummy = FW_BYTES / 4 + (MAX_WELLS + 1) * REC_FLOATS;
    auto well_at = [&](int dx, int dy) {
        const int wix = wcell >> 7, wiy = wcell & 127;
        const int a = min(max(wix + dx, 0), N128 - 1) * N128 + min(max(wiy + dy, 0), N128 - 1);
        return has_well ? a : dummy;
    };
    __syncthreads();

    const int ixW = max(ix0 - 1, 0), ixE = min(ix0 + PX, N128 - 1);
    auto load_row = [&](int ix, float (&f)[PY]) {
        const float4 v = *reinterpret_cast<const float4*>(fwf + ix * N128 + iy0);
        f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    };

    for (int it = 0; it < Nts; ++it) {
        // phase A: fractional flow of every own cell -> LDS
#pragma unroll
        for (int i = 0; i < PX; ++i) {
            float4 v;
            v.x = frac_flow<FD>(p, S[i][0]);
            v.y = frac_flow<FD>(p, S[i][1]);
            v.z = frac_flow<FD>(p, S[i][2]);
            v.w = frac_flow<FD>(p, S[i][3]);
            *reinterpret_cast<float4*>(fwf + (ix0 + i) * N128 + iy0) = v;
        }
        {   // well side path, branch-free (threads without a well run it on the dummy record)
            const float wf = frac_flow<FD>(p, rec[0]);
            rec[7] = wf;
            fwf[well_at(0, 0)] = wf;  // after this thread's own row write: ordered
        }
        __syncthreads();

        // phase B: upwind update, a sliding window of three fw rows
        float fp[PY], fc[PY], fn[PY];
        load_row(ixW, fp);
        load_row(ix0, fc);
#pragma unroll
        for (int i = 0; i < PX; ++i) {
            load_row(i + 1 < PX ? ix0 + i + 1 : ixE, fn);
            const float fS = prev_lane(fc[PY - 1]);  // fw(ix, iy0 - 1): its coefficient is 0 on the boundary
            const float fN = next_lane(fc[0]);       // fw(ix, iy0 + PY)
#pragma unroll
            for (int j = 0; j < PY; ++j) {
                const float fs = j > 0 ? fc[j > 0 ? j - 1 : 0] : fS;
                const float fnn = j + 1 < PY ? fc[j + 1 < PY ? j + 1 : 0] : fN;
                float acc = cE[i][j] * fn[j];
                acc = acc + cN[i][j] * fnn;
                acc = acc + cC[i][j] * fc[j];
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
        *reinterpret_cast<float4*>(Sout + (ix0 + i) * N128 + iy0) = v;
#pragma unroll
        for (int j = 0; j < PY; ++j) nonfinite |= ((ix0 + i) * N128 + iy0 + j != wcell) && !isfinite(S[i][j]);
    }
    if (has_well) {
        const float wS = rec[0];
        Sout[wcell] = wS;  // after this thread's own store of the patch: ordered
        nonfinite |= !isfinite(wS);
    }
    if (nonfinite) atomicOr(&p.status[m], HM_MEMBER_NONFINITE);
    __syncthreads();
    if (tid < p.nPrd) prods[((long long)m * p.nTime + k) * p.nPrd + tid] = Sout[p.prd_ind[tid]];
}

template <bool FD>
int launch(hm_fwd* f, const void* S_in, void* S_out, long long S_stride, int k) {
    auto kern = k_sat128f<FD>;
    HM_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    hipLaunchKernelGGL(kern, dim3(f->p.N), dim3(NT), LDS_BYTES, f->ctx->stream, f->p, (const float*)S_in, (float*)S_out, S_stride,
                       (float*)f->prods.p, k);
    HM_HIP(hipGetLastError());
    return 0;
}

}  // namespace

// Returns 0 if launched, >0 on error, -1 if this specialisation does not apply.
int launch_saturation_128f(hm_fwd* f, const void* S_in, void* S_out, long long S_stride, int k) {
    const FwdParams& p = f->p;
    if (p.q_mstride != 0) return -1;  // per-member wells: the well side path works from one shared well list
    if (p.Nx != N128 || p.Ny != N128 || f->dtype != 32 || p.por != nullptr) return -1;
    if ((int)f->well_cells_host.size() > MAX_WELLS) return -1;
    std::vector<int> seen;  // at most one well per 8x4 patch
    for (int cell : f->well_cells_host) {
        const int id = ((cell >> 7) >> 3) * 1000 + ((cell & 127) >> 2);
        for (int s : seen)
            if (s == id) return -1;
        seen.push_back(id);
    }
    return p.fluid_default ? launch<true>(f, S_in, S_out, S_stride, k) : launch<false>(f, S_in, S_out, S_stride, k);
}
