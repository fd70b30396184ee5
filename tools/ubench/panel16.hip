// micro-benchmark: in-wave LDL^T of a (R x 16) panel, one row (two for R > 64) per lane, pivot row broadcast by v_readlane.
// Candidate building block for a 16-pivot-per-round k_solve; prints shader clocks per panel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
__device__ __forceinline__ double fast_rcp(double d) {
    const double r0 = __builtin_amdgcn_rcp(d);
    const double e = fma(-d, r0, 1.0);
    const double t2 = fma(e, e, e);
    return fma(r0, t2, r0);
}
__device__ __forceinline__ long long tick(double dep) { long long t; asm volatile("s_nop 4\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "v"(dep) : "memory"); return t; }
__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
// A: R x 16 row-major (rows 0..15 = the diagonal block, symmetric input; only the lower part is used)
__global__ __launch_bounds__(64) void kern(const double* __restrict__ A, double* __restrict__ W, double* __restrict__ rinv, long long* cyc, int R, int reps) {
    const int t = threadIdx.x;
    long long total = 0;
    double a0[16], a1[16], r[16];
    for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
        for (int c = 0; c < 16; ++c) { a0[c] = (t < R) ? A[(size_t)t * 16 + c] : 0.0; a1[c] = (t + 64 < R) ? A[(size_t)(t + 64) * 16 + c] : 0.0; }
        __syncthreads();
        double dep0 = 0.0;
#pragma unroll
        for (int c = 0; c < 16; ++c) dep0 += a0[c] + a1[c];
        const long long c0 = tick(dep0);
#pragma unroll
        for (int c = 0; c < 16; ++c) { asm volatile("" : "+v"(a0[c])); asm volatile("" : "+v"(a1[c])); }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            // pivot row j lives in lane j: d_j and w_c = a(c,j) for c > j are held by lanes c (column j of rows c)
            const double dj = readlane_f64(a0[j], j);
            const double rj = fast_rcp(dj);
            r[j] = rj;
            const double l0 = a0[j] * rj, l1 = a1[j] * rj;          // L(row, j) for my rows (rows <= j: unused)
#pragma unroll
            for (int c = j + 1; c < 16; ++c) {
                const double wc = readlane_f64(a0[j], c);            // W(c, j) = a(c, j)
                a0[c] = fma(-l0, wc, a0[c]);
                a1[c] = fma(-l1, wc, a1[c]);
            }
        }
        double dep1 = 0.0;
#pragma unroll
        for (int c = 0; c < 16; ++c) dep1 += a0[c] + a1[c] + r[c];
        total += tick(dep1) - c0;
        __syncthreads();
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) { if (t < R) W[(size_t)t * 16 + c] = a0[c]; if (t + 64 < R) W[(size_t)(t + 64) * 16 + c] = a1[c]; }
    if (t < 16) { double v = 0; 
#pragma unroll
        for (int c = 0; c < 16; ++c) if (c == t) v = r[c];
        rinv[t] = v; }
    if (t == 0) cyc[0] = total;
}
int main() {
    const int R = 86;
    std::vector<double> M((size_t)200 * R), A((size_t)R * 16);
    srand(1);
    for (auto& v : M) v = rand() / (double)RAND_MAX - 0.5;
    std::vector<double> full((size_t)R * R);
    for (int i = 0; i < R; ++i) for (int j = 0; j < R; ++j) { double s = 0; for (int k = 0; k < 200; ++k) s += M[(size_t)k * R + i] * M[(size_t)k * R + j]; full[(size_t)i * R + j] = s + (i == j ? 1.0 : 0.0); }
    for (int i = 0; i < R; ++i) for (int c = 0; c < 16; ++c) A[(size_t)i * 16 + c] = full[(size_t)i * R + c];
    // host reference: W(i,j) = L(i,j) d_j for the first 16 columns
    std::vector<double> ref(A), d(16);
    for (int j = 0; j < 16; ++j) {
        d[j] = ref[(size_t)j * 16 + j];
        for (int c = j + 1; c < 16; ++c) for (int i = c; i < R; ++i) ref[(size_t)i * 16 + c] -= ref[(size_t)i * 16 + j] / d[j] * ref[(size_t)c * 16 + j];
    }
    double *dA, *dW, *dr; long long* dc;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dW, A.size() * 8); hipMalloc(&dr, 128); hipMalloc(&dc, 64);
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    const int reps = 20;
    long long best = 1ll << 60, h;
    for (int it = 0; it < 6; ++it) { hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, dA, dW, dr, dc, R, reps); hipMemcpy(&h, dc, 8, hipMemcpyDeviceToHost); if (h < best) best = h; }
    std::vector<double> W(A.size()); hipMemcpy(W.data(), dW, W.size() * 8, hipMemcpyDeviceToHost);
    double err = 0;
    for (int i = 0; i < R; ++i) for (int c = 0; c <= (i < 16 ? i : 15); ++c) err = fmax(err, fabs(W[(size_t)i * 16 + c] - ref[(size_t)i * 16 + c]));
    printf("in-wave 16-pivot panel of %d rows: %.0f clk (%.0f per pivot), max |W - ref| = %.3g\n", R, best / (double)reps, best / (double)reps / 16, err);
    return 0;
}
