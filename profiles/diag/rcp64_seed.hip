// rcp64_seed.hip -- how accurate is gfx950's v_rcp_f64?  max |1 - d r0| over d in [0.5, 2): the top 31 mantissa bits swept, the low bits from a hash,
// and after one quadratic / one cubic refinement.      hipcc -O3 --offload-arch=gfx950 -ffp-contract=off rcp64_seed.hip -o rcp64_seed && ./rcp64_seed
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>

__global__ void k(double* out) {
    const unsigned long long gid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x, stride = (unsigned long long)gridDim.x * blockDim.x;
    double m0 = 0.0, m1 = 0.0, m2 = 0.0;
    for (unsigned long long b = gid; b < (1ull << 32); b += stride) {
        unsigned long long h = b * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29;
        const unsigned long long mant = ((b & 0x7fffffffull) << 21) | (h & 0x1fffffull);
        const unsigned long long bits = ((b >> 31) ? 0x3FF0000000000000ull : 0x3FE0000000000000ull) | mant;  // [1, 2) or [0.5, 1)
        const double d = __longlong_as_double((long long)bits);
        const double r0 = __builtin_amdgcn_rcp(d);
        const double e0 = __builtin_fma(-d, r0, 1.0);
        const double r1 = __builtin_fma(r0, e0, r0);                       // quadratic
        const double e1 = __builtin_fma(-d, r1, 1.0);
        const double r2 = __builtin_fma(r0, __builtin_fma(e0, e0, e0), r0);  // cubic
        const double e2 = __builtin_fma(-d, r2, 1.0);
        m0 = fmax(m0, fabs(e0)); m1 = fmax(m1, fabs(e1)); m2 = fmax(m2, fabs(e2));
    }
    for (int off = 32; off > 0; off >>= 1) { m0 = fmax(m0, __shfl_xor(m0, off)); m1 = fmax(m1, __shfl_xor(m1, off)); m2 = fmax(m2, __shfl_xor(m2, off)); }
    if ((threadIdx.x & 63) == 0) {
        atomicMax((unsigned long long*)out, (unsigned long long)__double_as_longlong(m0));
        atomicMax((unsigned long long*)out + 1, (unsigned long long)__double_as_longlong(m1));
        atomicMax((unsigned long long*)out + 2, (unsigned long long)__double_as_longlong(m2));
    }
}
int main() {
    double* d; (void)hipMalloc(&d, 24); (void)hipMemset(d, 0, 24);
    k<<<2048, 256>>>(d);
    double h[3]; if (hipMemcpy(h, d, 24, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    printf("v_rcp_f64 on 2^32 operands in [0.5, 2): max |1 - d r| seed %.3e = 2^%.2f; after a quadratic step %.3e = 2^%.2f; after a cubic step %.3e = 2^%.2f\n",
           h[0], log2(h[0]), h[1], log2(h[1]), h[2], log2(h[2]));
}
