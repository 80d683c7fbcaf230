// exec_mask_rate.hip -- does a double-precision VALU instruction cost less when half of the wave's lanes are masked off?  (gfx950)
//   hipcc -O3 --offload-arch=gfx950 exec_mask_rate.hip -o exec_mask_rate && ./exec_mask_rate
// The question behind VERDICT round 3 item 8 (half-wave dry skips in the saturation sweep by exec narrowing instead of control flow): a
// DP instruction occupies the SIMD's 16 DP lanes for 4 passes of a 64-lane wave -- does the hardware skip the passes whose lanes are all
// inactive?  Each kernel runs ITER trips of 32 independent v_fma_f64 in every wave of a 512-thread workgroup (2 waves per SIMD, one
// workgroup per CU) with exec = all 64 lanes, the lower 32, the lower 16, or every fourth lane, and reports shader cycles per instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_rate(double* out, long long* cyc, int iters, unsigned long long mask) {
    double a[8], b[8], c[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = 1.0 + 0.001 * (threadIdx.x + i);
        b[i] = 1.0 + 1e-9 * i;
        c[i] = 1e-12 * i;
    }
    __syncthreads();
    const unsigned long long saved = __builtin_amdgcn_read_exec();
    long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    asm volatile("s_mov_b64 exec, %0" ::"s"(mask));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]));
    }
    asm volatile("s_mov_b64 exec, %0" ::"s"(saved));
    long long t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    double s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    double* out;
    long long* cyc;
    hipMalloc(&out, 256 * 512 * 8);
    hipMalloc(&cyc, 8);
    const int iters = 4000;
    const struct { const char* name; unsigned long long m; } cases[] = {
        {"all 64 lanes", ~0ull}, {"lower 32 lanes", 0xffffffffull}, {"lower 16 lanes", 0xffffull}, {"lanes 16..31", 0xffff0000ull}, {"every 4th lane", 0x1111111111111111ull}};
    for (auto& c : cases) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_rate, dim3(256), dim3(512), 0, 0, out, cyc, iters, c.m);
        hipDeviceSynchronize();
        long long h;
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        // s_memtime counts at 100 MHz on gfx950: convert with the 2.4 GHz shader clock assumption only as a ratio between the cases
        printf("%-16s: %lld memtime ticks for %d x 32 instructions of each of 2 waves per SIMD -> %.3f ticks / instruction / SIMD\n", c.name, h, iters,
               (double)h / (iters * 32.0 * 2.0));
    }
    return 0;
}
