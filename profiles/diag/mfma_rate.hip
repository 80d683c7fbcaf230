// Diagnostic microbenchmark: cycles per v_mfma_f64_16x16x4_f64 (and per v_fma_f64) on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k_mfma(double* out, long long* cyc, int iters) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = clock64();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x % 64 == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
__global__ void k_fma(double* out, long long* cyc, int iters) {
    double x[16];
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 1e-3 + i;
    double a = 1.0000001, b = 1e-9;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = fma(x[i], a, b);
    }
    long long t1 = clock64();
    double s = 0;
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x % 64 == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
int main() {
    double* out; long long* cyc;
    hipMalloc(&out, 1 << 26); hipMalloc(&cyc, 1 << 16);
    long long h[64];
    const int iters = 2000;
    for (int threads : {64, 256, 512, 1024}) {
        hipLaunchKernelGGL(k_mfma<4>, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
        hipDeviceSynchronize(); hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        printf("mfma_f64_16x16x4, 4 acc, %4d threads (1 WG): wave0 %.1f cycles per MFMA-per-wave; per SIMD: %.1f\n", threads,
               (double)h[0] / (iters * 4), (double)h[0] / (iters * 4) / ((threads / 64 + 3) / 4));
        hipLaunchKernelGGL(k_mfma<1>, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
        hipDeviceSynchronize(); hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        printf("mfma_f64_16x16x4, 1 acc (dependent), %4d threads: %.1f cycles per MFMA\n", threads, (double)h[0] / iters);
        hipLaunchKernelGGL(k_fma, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
        hipDeviceSynchronize(); hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        printf("v_fma_f64 x16 independent, %4d threads: %.2f cycles per FMA-per-wave\n", threads, (double)h[0] / (iters * 16));
    }
    // chip-wide wall-clock rates
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int big = 20000, blocks = 2048;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(1024), 0, 0, out, cyc, big);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flop = 2.0 * blocks * 1024.0 * big * 16;
        printf("chip v_fma_f64: %.2f ms -> %.1f TFLOP/s\n", ms, flop / ms / 1e9);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_mfma<4>, dim3(blocks), dim3(1024), 0, 0, out, cyc, big);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        flop = 2.0 * blocks * 16.0 * big * 4 * 1024;
        printf("chip mfma_f64_16x16x4: %.2f ms -> %.1f TFLOP/s\n", ms, flop / ms / 1e9);
    }
    return 0;
}
