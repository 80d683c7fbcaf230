// mfma32_mix.hip -- what an fp32 32x32x2 matrix instruction costs a SIMD when its operands come from LDS between the instructions
// (the inner loops of k_gxt_dma / k_apply_dma), against the bare stream.   hipcc -O3 --offload-arch=gfx950 mfma32_mix.hip -o mfma32_mix
// Variants (one workgroup per CU, NW waves): 0 bare MFMAs on NT accumulators; 1 + one ds_read_b32 per MFMA for the A operand and one
// per group for B, prefetched one group ahead; 2 as 1 with the loads of a group issued together before its MFMAs (no prefetch);
// 3 as 1 + the VALU work of the real loop (a subtraction and NT register moves per group).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int VAR, int NT>
__global__ void k(float* out, long long* cyc, int iters) {
    extern __shared__ float sm[];
    const int lane = threadIdx.x & 63;
    for (int e = threadIdx.x; e < 16384; e += blockDim.x) sm[e] = 0.001f * (e & 255);
    __syncthreads();
    f32x16 acc[NT];
    for (int t = 0; t < NT; ++t)
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    const float* sa = sm + lane;
    float a_cur[NT], a_nxt[NT], b_cur = 1.0f, b_nxt = 1.0f;
    for (int t = 0; t < NT; ++t) a_cur[t] = sa[64 * t];
    long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int it = 0; it < iters; ++it) {
        const float* row = sa + ((it * 7) & 63) * 128;
        if (VAR == 1 || VAR == 3) {
            b_nxt = row[64 * NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) a_nxt[t] = row[64 * t];
        }
        if (VAR == 2) {
            b_cur = row[64 * NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) a_cur[t] = row[64 * t];
        }
        __builtin_amdgcn_sched_barrier(0);
        const float bs = VAR == 3 ? b_cur - 0.5f : b_cur;
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[t], bs, acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (VAR == 1 || VAR == 3) {
            b_cur = b_nxt;
#pragma unroll
            for (int t = 0; t < NT; ++t) a_cur[t] = a_nxt[t];
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    float s = 0;
    for (int t = 0; t < NT; ++t)
        for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int VAR, int NT>
void run(const char* name) {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 256 * 1024 * 4); (void)hipMalloc(&cyc, 8);
    (void)hipFuncSetAttribute((const void*)k<VAR, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4000;
    for (int threads : {256, 512, 1024}) {
        k<VAR, NT><<<256, threads, 65536>>>(out, cyc, iters);
        (void)hipEventRecord(e0);
        k<VAR, NT><<<256, threads, 65536>>>(out, cyc, iters);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double mfma_per_simd = (double)iters * NT * (threads / 256);
        printf("%-44s NT=%d %d waves/SIMD: %6.1f cycles per MFMA per SIMD (wall, 2.4 GHz) = %5.1f %% of the matrix peak\n", name, NT, threads / 256,
               ms * 1e-3 * 2.4e9 / mfma_per_simd, 100.0 * 64.0 / (ms * 1e-3 * 2.4e9 / mfma_per_simd));
    }
    (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    run<0, 5>("bare");
    run<1, 5>("LDS operands, prefetched one group ahead");
    run<2, 5>("LDS operands, loaded right before use");
    run<3, 5>("prefetched + subtraction and moves");
    run<0, 2>("bare");
    run<1, 2>("LDS operands, prefetched one group ahead");
    run<3, 2>("prefetched + subtraction and moves");
    return 0;
}
