// clamp_f64.hip -- what the VOP3 `clamp` output modifier does to the fp64 products of the saturation sweep (gfx950): is
// clamp(a * b) == max(a * b, +0) bit for bit for products below 1 (zeros of both signs, denormal products, tiny normals)?
//   hipcc -O3 --offload-arch=gfx950 clamp_f64.hip -o clamp_f64 && ./clamp_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
#include <random>

__global__ void k_clamp(const double* a, const double* b, double* rc, double* rn, double* rm, double* rmn, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x = a[i], y = b[i], c, nc, prod, nprod, m, mn, z = 0.0;
    asm volatile("v_mul_f64 %0, %1, %2 clamp" : "=v"(c) : "v"(x), "v"(y));
    asm volatile("v_mul_f64 %0, -%1, %2 clamp" : "=v"(nc) : "v"(x), "v"(y));
    asm volatile("v_mul_f64 %0, %1, %2" : "=v"(prod) : "v"(x), "v"(y));
    asm volatile("v_mul_f64 %0, -%1, %2" : "=v"(nprod) : "v"(x), "v"(y));
    asm volatile("v_max_f64 %0, %1, %2" : "=v"(m) : "v"(prod), "v"(z));
    asm volatile("v_max_f64 %0, %1, %2" : "=v"(mn) : "v"(nprod), "v"(z));
    rc[i] = c; rn[i] = nc; rm[i] = m; rmn[i] = mn;
}

int main() {
    std::vector<double> a, b;
    const double den = 4.9406564584124654e-324, tiny = 2.2250738585072014e-308;
    const double sp[] = {0.0, -0.0, den, -den, 3 * den, tiny, -tiny, tiny * (1 + 1e-15), 1e-300, -1e-300, 1e-160, -1e-160, 1e-5, -1e-5, 0.3, -0.3, 0.999, 1e-320, -1e-320};
    const double fw[] = {0.0, den, 1e-320, tiny, 1e-300, 1e-160, 1e-17, 1e-3, 0.5, 0.999999, 1.0};
    for (double x : sp) for (double y : fw) { a.push_back(x); b.push_back(y); }
    std::mt19937_64 g(1);
    std::uniform_real_distribution<double> u(-1, 1), e(-1070, 0);
    for (int i = 0; i < 1 << 20; ++i) { a.push_back(u(g) * 0.34 * std::exp2(e(g) * (i & 1))); b.push_back(std::fabs(u(g)) * std::exp2(e(g) * ((i >> 1) & 1))); }
    const int n = (int)a.size();
    double *da, *db, *rc, *rn, *rm, *rmn;
    hipMalloc(&da, n * 8); hipMalloc(&db, n * 8); hipMalloc(&rc, n * 8); hipMalloc(&rn, n * 8); hipMalloc(&rm, n * 8); hipMalloc(&rmn, n * 8);
    hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
    k_clamp<<<(n + 255) / 256, 256>>>(da, db, rc, rn, rm, rmn, n);
    std::vector<double> hc(n), hn(n), hm(n), hmn(n);
    hipMemcpy(hc.data(), rc, n * 8, hipMemcpyDeviceToHost); hipMemcpy(hn.data(), rn, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(hm.data(), rm, n * 8, hipMemcpyDeviceToHost); hipMemcpy(hmn.data(), rmn, n * 8, hipMemcpyDeviceToHost);
    long bad = 0, denormal_products = 0, zeros = 0;
    for (int i = 0; i < n; ++i) {
        if (memcmp(&hc[i], &hm[i], 8) || memcmp(&hn[i], &hmn[i], 8)) {
            if (bad++ < 10) printf("MISMATCH a = %.17g b = %.17g: clamp %.17g max %.17g | clamp(-ab) %.17g max %.17g\n", a[i], b[i], hc[i], hm[i], hn[i], hmn[i]);
        }
        if (hm[i] != 0 && std::fabs(hm[i]) < tiny) ++denormal_products;
        if (hm[i] == 0) ++zeros;
    }
    printf("%d operand pairs (|a| < 0.34, 0 <= b <= 1; %ld denormal positive products, %ld zero results): clamp(a b) and clamp(-a b) differ from max(+-a b, +0) in %ld cases\n",
           n, denormal_products, zeros, bad);
    return bad != 0;
}
