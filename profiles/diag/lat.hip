// Diagnostic microbenchmark: latencies of the primitives the blocked sweep chains per panel (gfx950, one workgroup).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double rcp_newton(double d) {
    double x = __builtin_amdgcn_rcp(d);
    double e = fma(-d, x, 1.0); x = fma(x, e, x); e = fma(-d, x, 1.0); x = fma(x, e, x);
    return x;
}
__global__ void k(double* out, long long* res, int iters) {
    __shared__ double sh[2048];
    const int tid = threadIdx.x;
    for (int i = tid; i < 2048; i += blockDim.x) sh[i] = 1.0 + 1e-3 * i;
    __syncthreads();
    long long t0, t1;
    // 1. barrier
    t0 = clock64();
    for (int i = 0; i < iters; ++i) __syncthreads();
    t1 = clock64();
    if (tid == 0) res[0] = (t1 - t0) / iters;
    // 2. dependent LDS read chain (index from the loaded value)
    int idx = tid & 1023;
    double acc = 0;
    t0 = clock64();
    for (int i = 0; i < iters; ++i) { double v = sh[idx]; acc += v; idx = (idx + (int)v) & 1023; }
    t1 = clock64();
    if (tid == 0) res[1] = (t1 - t0) / iters;
    // 3. dependent rcp_newton chain
    double x = 1.5 + tid * 1e-6;
    t0 = clock64();
    for (int i = 0; i < iters; ++i) x = rcp_newton(x) + 1.0;
    t1 = clock64();
    if (tid == 0) res[2] = (t1 - t0) / iters;
    // 4. dependent fma chain
    double y = 1.0 + tid * 1e-9;
    t0 = clock64();
    for (int i = 0; i < iters; ++i) { y = fma(y, 1.0000001, 1e-9); y = fma(y, 1.0000001, 1e-9); y = fma(y, 1.0000001, 1e-9); y = fma(y, 1.0000001, 1e-9); }
    t1 = clock64();
    if (tid == 0) res[3] = (t1 - t0) / (4 * iters);
    // 5. MFMA then dependent VALU read of the result
    d4 c = {0, 0, 0, 0};
    double a = 1e-3 * tid, b = 1.0;
    t0 = clock64();
    for (int i = 0; i < iters; ++i) { c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); a = c[0] * 1e-9 + 1e-3; }
    t1 = clock64();
    if (tid == 0) res[4] = (t1 - t0) / iters;
    // 6. LDS write -> barrier -> LDS read by another wave -> barrier (one hand-off)
    t0 = clock64();
    for (int i = 0; i < iters; ++i) { sh[tid] = acc + i; __syncthreads(); acc += sh[(tid + 64) & (blockDim.x - 1)]; __syncthreads(); }
    t1 = clock64();
    if (tid == 0) res[5] = (t1 - t0) / iters;
    // 7. scratch-like global round trip (dependent load from global, L2-hit)
    long long t2 = clock64();
    int gi = tid;
    for (int i = 0; i < iters; ++i) { gi = (int)out[gi & 1023] & 1023; }
    long long t3 = clock64();
    if (tid == 0) res[6] = (t3 - t2) / iters;
    out[tid + 2048] = acc + x + y + c[1] + gi;
}
int main() {
    double* out; long long* res;
    hipMalloc(&out, 1 << 20); hipMalloc(&res, 64);
    hipMemset(out, 0, 1 << 20);
    long long h[8];
    for (int threads : {64, 256, 1024}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(threads), 0, 0, out, res, 2000);
        hipDeviceSynchronize(); hipMemcpy(h, res, 56, hipMemcpyDeviceToHost);
        printf("%4d threads: barrier %lld | dep LDS read %lld | rcp_newton+add chain %lld | dep fma %lld | mfma->valu->mfma %lld | lds handoff (2 barriers) %lld | dep global load (L2) %lld  [ticks]\n",
               threads, h[0], h[1], h[2], h[3], h[4], h[5], h[6]);
    }
    return 0;
}
