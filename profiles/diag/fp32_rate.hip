// fp32_rate.hip -- issue cost of the single-precision vector instructions the fp32 saturation sweep is made of, and what the
// VOP3 `clamp` output modifier does to the products the sweep forms (gfx950).
//   hipcc -O3 --offload-arch=gfx950 fp32_rate.hip -o fp32_rate && ./fp32_rate
// Part 1: every kernel runs ITER trips of 32 independent instructions of one kind in every wave of a 512-thread workgroup (2 waves per
// SIMD, one workgroup per CU): wall time -> ns and cycles per wave-instruction per SIMD.
// Part 2: clamp(a * b) for the operand classes the sweep meets: is it max(a * b, +0) bit for bit (also for -0, denormals, and
// products above 1, which the sweep never forms)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

template <int OP>
__global__ void k_rate(float* out, int iters, float seed) {
    float a[8], b[8], c[8];
    float2 pa[8], pb[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = seed + 0.001f * (threadIdx.x + i);
        b[i] = 1.0f + 0.01f * i + 1e-3f * threadIdx.x;
        c[i] = 0.5f + 0.02f * i;
        pa[i] = make_float2(a[i], b[i]);
        pb[i] = make_float2(c[i], a[i]);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 2) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]));
                if (OP == 3) asm volatile("v_mul_f32_e64 %0, -%0, %1 clamp" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 4) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
                if (OP == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pa[i]) : "v"(pb[i]));
                if (OP == 6) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pa[i]) : "v"(pb[i]));
                if (OP == 7) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 8) asm volatile("v_mov_b32_dpp %0, %0 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]));
                if (OP == 9) asm volatile("v_sub_f32 %0, 1.0, %0" : "+v"(a[i]));
                if (OP == 10) asm volatile("v_add_f32_dpp %0, %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 11) asm volatile("v_add_f64 %0, %0, %1" : "+v"(*(double*)&pa[i]) : "v"(*(double*)&pb[i]));
            }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + pa[i].x + pa[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
void run(const char* name, float* out) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int threads = 256; threads <= 1024; threads *= 2) {
        k_rate<OP><<<256, threads>>>(out, iters, 1.0f);
        hipEventRecord(e0);
        k_rate<OP><<<256, threads>>>(out, iters, 1.0f);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double winst = 256.0 * (threads / 64) * iters * 32;  // wave-instructions
        const double ns = ms * 1e6 / (winst / 1024.0);
        printf("%-28s %d waves/SIMD: wall %7.3f ms -> %5.2f ns per wave-instruction per SIMD = %5.2f cycles at 2.4 GHz\n", name, threads / 256, ms, ns, ns * 2.4);
    }
}

__global__ void k_clamp(const float* a, const float* b, float* r_clamp, float* r_negclamp, float* r_max, int n) {
    const int i = threadIdx.x;
    if (i >= n) return;
    float x = a[i], y = b[i], c, nc;
    asm volatile("v_mul_f32_e64 %0, %1, %2 clamp" : "=v"(c) : "v"(x), "v"(y));
    asm volatile("v_mul_f32_e64 %0, -%1, %2 clamp" : "=v"(nc) : "v"(x), "v"(y));
    r_clamp[i] = c;
    r_negclamp[i] = nc;
    float prod;
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(prod) : "v"(x), "v"(y));
    float z = 0.0f, m;
    asm volatile("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(prod), "v"(z));
    r_max[i] = m;
}

int main() {
    float* out;
    hipMalloc(&out, 1024 * 256 * 4);
    run<0>("v_add_f32", out);
    run<1>("v_mul_f32", out);
    run<2>("v_fma_f32", out);
    run<3>("v_mul_f32 -a, b clamp", out);
    run<4>("v_rcp_f32", out);
    run<5>("v_pk_mul_f32", out);
    run<6>("v_pk_add_f32", out);
    run<7>("v_max_f32", out);
    run<8>("v_mov_b32_dpp", out);
    run<9>("v_sub_f32 1.0 - a", out);
    run<10>("v_add_f32_dpp", out);
    run<11>("v_add_f64", out);

    const float den = 1e-40f;  // denormal
    const float av[] = {1e-5f, -1e-5f, 0.0f, -0.0f, 1e-5f, -1e-5f, den, -den, 1e-30f, -1e-30f, 3.0f, -3.0f, 0.0f, -0.0f, 1e-5f};
    const float bv[] = {0.5f, 0.5f, 0.5f, 0.5f, 0.0f, 0.0f, 0.5f, 0.5f, 1e-12f, 1e-12f, 0.5f, 0.5f, 0.0f, 0.0f, 1.0f};
    const int n = sizeof(av) / sizeof(float);
    float *da, *db, *rc, *rn, *rm;
    hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&rc, n * 4); hipMalloc(&rn, n * 4); hipMalloc(&rm, n * 4);
    hipMemcpy(da, av, n * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, bv, n * 4, hipMemcpyHostToDevice);
    k_clamp<<<1, 64>>>(da, db, rc, rn, rm, n);
    float hc[32], hn[32], hm[32];
    hipMemcpy(hc, rc, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hn, rn, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hm, rm, n * 4, hipMemcpyDeviceToHost);
    printf("\n%14s %14s | %-22s %-22s %-22s\n", "a", "b", "clamp(a*b)", "clamp(-a*b)", "max(a*b, +0)");
    for (int i = 0; i < n; ++i) {
        unsigned uc, un, um;
        memcpy(&uc, &hc[i], 4); memcpy(&un, &hn[i], 4); memcpy(&um, &hm[i], 4);
        printf("%14.6e %14.6e | %12.5e %08x  %12.5e %08x  %12.5e %08x%s\n", av[i], bv[i], hc[i], uc, hn[i], un, hm[i], um,
               uc == um ? "" : "   clamp != max");
    }
    return 0;
}
