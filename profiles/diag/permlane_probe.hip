// Probe of v_permlane16_swap / v_permlane32_swap (gfx950) and of the row broadcast built from them:
//   hipcc --offload-arch=gfx950 -O2 permlane_probe.hip -o permlane_probe && ./permlane_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int S>
__device__ __forceinline__ unsigned row_bcast32(unsigned v) {  // lanes 16 S .. 16 S + 15 to every row of 16 lanes
    auto p = __builtin_amdgcn_permlane16_swap(v, v, false, false);   // p[0] = {r0, r0, r2, r2}, p[1] = {r1, r1, r3, r3}
    const unsigned x = (S & 1) ? p[1] : p[0];
    auto q = __builtin_amdgcn_permlane32_swap(x, x, false, false);   // q[0] = {x.r0, x.r1, x.r0, x.r1}, q[1] = {x.r2, x.r3, x.r2, x.r3}
    return (S & 2) ? q[1] : q[0];
}

__global__ void k(unsigned* out) {
    const unsigned l = threadIdx.x;
    auto p = __builtin_amdgcn_permlane16_swap(l, 100 + l, false, false);
    out[l] = p[0];
    out[64 + l] = p[1];
    auto q = __builtin_amdgcn_permlane32_swap(l, 100 + l, false, false);
    out[128 + l] = q[0];
    out[192 + l] = q[1];
    out[256 + l] = row_bcast32<0>(l);
    out[320 + l] = row_bcast32<1>(l);
    out[384 + l] = row_bcast32<2>(l);
    out[448 + l] = row_bcast32<3>(l);
}

int main() {
    unsigned *d, h[512];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[8] = {"permlane16_swap(l, 100+l)[0]", "permlane16_swap(l, 100+l)[1]", "permlane32_swap(l, 100+l)[0]", "permlane32_swap(l, 100+l)[1]",
                            "row_bcast<0>", "row_bcast<1>", "row_bcast<2>", "row_bcast<3>"};
    int ok = 1;
    for (int a = 0; a < 8; ++a) {
        printf("%-30s", names[a]);
        for (int l = 0; l < 64; l += 1) if ((l & 15) == 0 || (l & 15) == 15) printf(" [%2d]=%3u", l, h[64 * a + l]);
        printf("\n");
        if (a >= 4) for (int l = 0; l < 64; ++l) ok &= h[64 * a + l] == (unsigned)(16 * (a - 4) + (l & 15));
    }
    printf("row broadcast %s\n", ok ? "OK" : "WRONG");
    return ok ? 0 : 1;
}
