// Accuracy of v_rcp_f64 on gfx950 and of one quadratic / cubic Newton step behind it (fast_rcp in avt_lm.hip uses the cubic step): run through gpurun,
//   hipcc -O2 --offload-arch=gfx950 -o tools/ubench/rcp_accuracy.bin tools/ubench/rcp_accuracy.hip && gpurun -- ./tools/ubench/rcp_accuracy.bin
// measured (round 6): v_rcp_f64 4.6e-08 | + quadratic step 2.2e-15 | + cubic step 1.1e-16 (2^-52 = 2.2e-16): the quadratic step would save one of the four
// dependent operations per pivot and leave ten units in the last place - kept cubic.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double* x, double* r0, double* r1, double* r2, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    double d = x[i];
    double a = __builtin_amdgcn_rcp(d);
    r0[i] = a;
    double e = fma(-d, a, 1.0);
    r1[i] = fma(a, e, a);                 // quadratic
    double t2 = fma(e, e, e);
    r2[i] = fma(a, t2, a);                // cubic (the library's fast_rcp)
}
int main() {
    const int n = 1 << 20; std::vector<double> x(n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; double u = (s >> 11) * (1.0 / 9007199254740992.0); x[i] = ldexp(1.0 + u, (int)(s % 61) - 30); }
    double *dx, *d0, *d1, *d2; hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, d0, d1, d2, n);
    std::vector<double> a(n), b(n), c(n);
    hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost);
    double m0 = 0, m1 = 0, m2 = 0;
    for (int i = 0; i < n; ++i) { long double t = 1.0L / (long double)x[i]; m0 = fmax(m0, fabs((double)(((long double)a[i] - t) / t))); m1 = fmax(m1, fabs((double)(((long double)b[i] - t) / t))); m2 = fmax(m2, fabs((double)(((long double)c[i] - t) / t))); }
    printf("max relative error: v_rcp_f64 %.3e | + quadratic step %.3e | + cubic step %.3e  (2^-52 = %.3e)\n", m0, m1, m2, ldexp(1.0, -52));
    return 0;
}
