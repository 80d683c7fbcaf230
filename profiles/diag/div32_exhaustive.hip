// div32_exhaustive.hip -- the float32 fractional flow of the upstream fluid, fw(s) = s^2 / (s^2 + (1 - s)^2), is a function of ONE float: every
// shorter instruction sequence for its division can be checked against the IEEE quotient on ALL 2^32 bit patterns of s (gfx950's v_rcp_f32
// included: the check is of this hardware's seed).  Candidates, against `mw / d` as the compiler divides (v_div_scale .. v_div_fixup):
//   A  rcp, Newton step, q = n r, two residual corrections      (8 instructions: fracflow.h before this check)
//   B  rcp, Newton step, q = n r, ONE residual correction       (6)
//   C  rcp,              q = n r, two residual corrections      (6)
//   D  rcp,              q = n r, one residual correction       (4)
//   E  rcp, Newton step, q = n r                                (4)
//   F, G, H   D with the denominator formed with one rounding less (an fma): not the specified arithmetic -- do the results differ?
// Per candidate: the number of s whose result differs from the IEEE quotient in any bit (NaN == NaN), over all floats and over |s| <= 2,
// the first few such s, and the smallest |s| that differs at all.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off div32_exhaustive.hip -o div32_exhaustive && ./div32_exhaustive
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>

constexpr int NC = 8;

__device__ __forceinline__ float rcp(float d) { return __builtin_amdgcn_rcpf(d); }
__device__ __forceinline__ float cand(int c, float n, float d, float s, float o) {
    if (c == 5) d = __builtin_fmaf(o, o, n);                        // F: d with one rounding less, then D
    if (c == 6) d = __builtin_fmaf(s, s, o * o);                    // G: the other way round, then D
    if (c == 7) d = __builtin_fmaf(s + s, s - 1.0f, 1.0f);          // H: d = 2 s (s - 1) + 1, then D
    float r = rcp(d), e, q;
    if (c == 0 || c == 1 || c == 4) {
        e = __builtin_fmaf(-d, r, 1.0f);
        r = __builtin_fmaf(e, r, r);
    }
    q = n * r;
    if (c == 4) return q;
    e = __builtin_fmaf(-d, q, n);
    q = __builtin_fmaf(e, r, q);
    if (c == 1 || c == 3 || c >= 5) return q;
    e = __builtin_fmaf(-d, q, n);
    return __builtin_fmaf(e, r, q);
}

__global__ void k_check(unsigned long long* counts, unsigned* firsts) {
    const unsigned long long gid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long bad[NC] = {}, bad2[NC] = {};
    for (unsigned long long b = gid; b < (1ull << 32); b += stride) {
        const float s = __uint_as_float((unsigned)b);
        const float mw = s * s, o = 1.0f - s, mo = o * o, d = mw + mo;
        const float ref = mw / d;
        const bool in2 = fabsf(s) <= 2.0f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float v = cand(c, mw, d, s, o);
            const bool same = __float_as_uint(v) == __float_as_uint(ref) || (v != v && ref != ref);
            if (!same) {
                ++bad[c];
                if (s == s) atomicMin(&firsts[NC * 16 + c], __float_as_uint(fabsf(s)));  // smallest |s| that differs (positive floats order as their bits)
                if (in2) {
                    if (bad2[c] == 0 && atomicAdd(&firsts[c * 16], 1u) < 15u) {
                        const unsigned k = atomicAdd(&firsts[c * 16 + 15], 1u);
                        if (k < 14) firsts[c * 16 + 1 + k] = (unsigned)b;
                    }
                    ++bad2[c];
                }
            }
        }
    }
    for (int c = 0; c < NC; ++c) {
        if (bad[c]) atomicAdd(&counts[c], bad[c]);
        if (bad2[c]) atomicAdd(&counts[NC + c], bad2[c]);
    }
}

int main() {
    unsigned long long* counts;
    unsigned* firsts;
    hipMalloc(&counts, 2 * NC * 8);
    hipMalloc(&firsts, (NC * 16 + NC) * 4);
    hipMemset(counts, 0, 2 * NC * 8);
    hipMemset(firsts, 0, NC * 16 * 4);
    hipMemset(firsts + NC * 16, 0x7f, NC * 4);
    k_check<<<256 * 8, 256>>>(counts, firsts);
    unsigned long long h[2 * NC];
    unsigned f[NC * 16 + NC];
    if (hipMemcpy(h, counts, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) { printf("failed\n"); return 1; }
    hipMemcpy(f, firsts, sizeof f, hipMemcpyDeviceToHost);
    const char* names[NC] = {"A  rcp + Newton + 2 corrections (8 instr)", "B  rcp + Newton + 1 correction  (6 instr)", "C  rcp + 2 corrections          (6 instr)",
                             "D  rcp + 1 correction           (4 instr)", "E  rcp + Newton, no correction  (4 instr)",
                             "F  d = fma(o, o, s s), then D             ", "G  d = fma(s, s, o o), then D             ", "H  d = fma(2 s, s - 1, 1), then D         "};
    printf("fw(s) = s^2 / (s^2 + (1 - s)^2) in float32, all 2^32 bit patterns of s, against the IEEE quotient (compiler's division):\n");
    for (int c = 0; c < NC; ++c) {
        printf("  %s: %llu of 4294967296 differ; with |s| <= 2: %llu", names[c], h[c], h[NC + c]);
        const unsigned n = f[c * 16 + 15] < 14 ? f[c * 16 + 15] : 14;
        if (n) printf("   e.g. s =");
        for (unsigned k = 0; k < n && k < 4; ++k) { float s; memcpy(&s, &f[c * 16 + 1 + k], 4); printf(" %.9g", s); }
        float smin;
        memcpy(&smin, &f[NC * 16 + c], 4);
        printf(";  smallest |s| that differs: %.9g\n", smin);
    }
    return 0;
}
