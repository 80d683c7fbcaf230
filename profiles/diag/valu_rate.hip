// valu_rate.hip -- issue cost of the double-precision vector instructions the saturation sweep is made of (gfx950).
//   hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate && ./valu_rate
// Every kernel runs ITER trips of 32 independent instructions of one kind (8 register sets, 4 rounds) in every wave of a
// 256- or 512-thread workgroup (1 or 2 waves per SIMD), one workgroup per CU, and reports shader cycles per instruction per
// SIMD (s_memtime around the loop, wave 0 of workgroup 0).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ void k_rate(double* out, long long* cyc, int iters, double seed) {
    double a[8], b[8], c[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = seed + 0.001 * (threadIdx.x + i);
        b[i] = 1.0 + 0.01 * i + 1e-3 * threadIdx.x;
        c[i] = 0.5 + 0.02 * i;
    }
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#define ONE(i)                                                                                                              \
    if (OP == 0) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]));                              \
    if (OP == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));                                             \
    if (OP == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));                                             \
    if (OP == 3) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));                                             \
    if (OP == 4) asm volatile("v_min_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));                                             \
    if (OP == 5) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[i]));                                                             \
    if (OP == 6) asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]) : "vcc");                      \
    if (OP == 7) asm volatile("v_div_fmas_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]) : "vcc");                 \
    if (OP == 8) asm volatile("v_div_fixup_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]));                        \
    if (OP == 9) { int lo = __double2loint(a[i]); asm volatile("v_mov_b32_dpp %0, %0 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(lo)); a[i] = __hiloint2double(__double2hiint(a[i]), lo); } \
    if (OP == 10) { int lo = __double2loint(a[i]); asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo) : "v"(__double2loint(b[i]))); a[i] = __hiloint2double(__double2hiint(a[i]), lo); } \
    if (OP == 11) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(*(int*)&a[i]) : "v"(*(int*)&b[i]) : "vcc");            \
    if (OP == 12) asm volatile("v_rcp_f32 %0, %0" : "+v"(*(float*)&a[i]));                                                   \
    if (OP == 13) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(*(float*)&a[i]) : "v"(b[i]));                                   \
    if (OP == 14) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[i]) : "v"(*(float*)&b[i]));                                   \
    if (OP == 15) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]));                            \
    if (OP == 16) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a[i]) : "v"(__double2loint(b[i])));                           \
    if (OP == 17) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(a[i]), "v"(b[i]) : "vcc");                                 \
    if (OP == 18) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(*(float*)&a[i]) : "v"(*(float*)&b[i]), "v"(*(float*)&c[i])); \
    if (OP == 19) asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(a[i]) : "v"(b[i]));                                          \
    if (OP == 20) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(a[i]));                                                      \
    if (OP == 21) asm volatile("v_fma_f64 %0, %1, 2.0, %0" : "+v"(a[i]) : "v"(b[i]));                                         \
    if (OP == 22) asm volatile("v_fma_f64 %0, %1, %2, 1.0" : "=v"(a[i]) : "v"(b[i]), "v"(c[i]));                              \
    if (OP == 23) a[i] = b[i] / a[i];                                                                                         \
    if (OP == 24) asm volatile("v_fma_f64 %0, -%1, %2, 1.0" : "=v"(a[i]) : "v"(b[i]), "v"(c[i]));
            REP8(ONE)
#undef ONE
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    double s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int OP>
void run(const char* name) {
    double* out;
    long long* cyc;
    hipMalloc(&out, 512 * 1024 * 8);
    hipMalloc(&cyc, 8);
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int cfg = 0; cfg < 4; ++cfg) {
        const int threads = cfg == 0 ? 256 : cfg == 1 ? 512 : 1024, blocks = cfg == 3 ? 512 : 256;
        k_rate<OP><<<blocks, threads>>>(out, cyc, iters, 1.0);
        hipEventRecord(e0);
        k_rate<OP><<<blocks, threads>>>(out, cyc, iters, 1.0);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        long long c;
        hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double per = (double)c / ((double)iters * 32);
        const double winst = (double)blocks * (threads / 64) * iters * 32;  // wave-instructions
        printf("%-16s %2d waves/SIMD: %6.2f ticks/instr/wave; wall %7.3f ms -> %6.2f ns per wave-instr per SIMD = %5.2f cycles at 2.4 GHz\n",
               name, blocks / 256 * threads / 256, per, ms, ms * 1e6 / (winst / 1024.0), ms * 1e6 / (winst / 1024.0) * 2.4);
    }
    hipFree(out);
    hipFree(cyc);
}

int main() {
    run<0>("v_fma_f64");
    run<1>("v_add_f64");
    run<2>("v_mul_f64");
    run<3>("v_max_f64");
    run<4>("v_min_f64");
    run<5>("v_rcp_f64");
    run<6>("v_div_scale_f64");
    run<7>("v_div_fmas_f64");
    run<8>("v_div_fixup_f64");
    run<9>("v_mov_b32_dpp");
    run<10>("v_add_u32");
    run<11>("v_cndmask_b32");
    run<12>("v_rcp_f32");
    run<13>("v_cvt_f32_f64");
    run<14>("v_cvt_f64_f32");
    run<15>("v_pk_fma_f32");
    run<16>("v_ldexp_f64");
    run<17>("v_cmp_lt_f64");
    run<18>("v_fma_f32");
    run<19>("fma a=b*b+a");
    run<20>("fma a=a*a+a");
    run<21>("fma a=b*2+a");
    run<22>("fma a=b*c+1");
    run<23>("IEEE a=b/a");
    run<24>("fma a=-b*c+1");
    return 0;
}
