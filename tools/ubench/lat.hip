// micro-latency probes on gfx950: one wave, clock64() deltas
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ long long tick(double dep) { long long t; asm volatile("s_nop 4\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "v"(dep) : "memory"); return t; }
__device__ __forceinline__ long long ticki(int dep) { long long t; asm volatile("s_nop 4\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "v"(dep) : "memory"); return t; }
__device__ __forceinline__ long long tickf(float dep) { long long t; asm volatile("s_nop 4\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "v"(dep) : "memory"); return t; }
__global__ void k(double* out, long long* t, double a0, double b0) {
    __shared__ double lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = a0 + i;
    __syncthreads();
    double a = a0 + threadIdx.x, b = b0;
    long long c0 = tick(a);
#pragma unroll
    for (int i = 0; i < 256; ++i) { a = fma(a, b, b); asm volatile("" : "+v"(a)); }   // dependent fma chain
    long long c1 = tick(a);
    double x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, x6 = a + 6, x7 = a + 7;
    long long c2 = tick(x7);
#pragma unroll
    for (int i = 0; i < 32; ++i) { x0 = fma(x0, b, b); x1 = fma(x1, b, b); x2 = fma(x2, b, b); x3 = fma(x3, b, b); x4 = fma(x4, b, b); x5 = fma(x5, b, b); x6 = fma(x6, b, b); x7 = fma(x7, b, b); asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)); }
    long long c3 = tick(x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7);
    double r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    long long c4 = tick(r);
#pragma unroll
    for (int i = 0; i < 64; ++i) r = __builtin_amdgcn_rcp(r) + 1.0;   // rcp + add chain
    long long c5 = tick(r);
    int idx = ((int)r) & 1023;
    long long c6 = ticki(idx);
#pragma unroll
    for (int i = 0; i < 64; ++i) idx = ((int)lds[idx & 1023]) & 1023;   // dependent LDS read chain (+cvt)
    long long c7 = ticki(idx);
    float f = (float)r;
    long long c8 = tickf(f);
#pragma unroll
    for (int i = 0; i < 256; ++i) { f = fmaf(f, 1.0001f, 0.5f); asm volatile("" : "+v"(f)); }
    long long c9 = tickf(f);
    out[threadIdx.x] = r + idx + f;
    if (threadIdx.x == 0) { t[0] = c1 - c0; t[1] = c3 - c2; t[2] = c5 - c4; t[3] = c7 - c6; t[4] = c9 - c8; }
}
int main() {
    double* out; long long* t; hipMalloc(&out, 64 * 8); hipMalloc(&t, 64);
    long long h[5];
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, t, 1.0000001, 0.999999);
        hipMemcpy(h, t, 40, hipMemcpyDeviceToHost);
        printf("dep fma64: %.1f clk/op | 8-way indep fma64: %.1f clk/op | rcp64+add: %.1f clk/pair | LDS dep read(+cvt): %.1f clk | dep fma32: %.1f clk/op\n",
               h[0] / 256.0, h[1] / 256.0, h[2] / 64.0, h[3] / 64.0, h[4] / 256.0);
    }
    return 0;
}
