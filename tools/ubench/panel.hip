// micro-benchmark of k_solve's panel-column sequence on gfx950: one wave, s_memtime deltas (shader clocks)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ long long tick(double dep) { long long t; asm volatile("s_nop 4\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "v"(dep) : "memory"); return t; }
__device__ __forceinline__ double fast_rcp(double d) {
    const double r0 = __builtin_amdgcn_rcp(d);
    const double e = fma(-d, r0, 1.0);
    const double t2 = fma(e, e, e);
    return fma(r0, t2, r0);
}
__global__ void k(double* out, long long* tt, double a0) {
    __shared__ __attribute__((aligned(16))) double lds[4096];
    const int t = threadIdx.x;
    for (int i = t; i < 4096; i += 64) lds[i] = (i % 5 == 0) ? 4.0 + 0.001 * i : 0.01 * (i % 7) * a0;
    __syncthreads();
    double acc = 0.0;
    long long T[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int rep = 0; rep < 16; ++rep) {
        const d2v* Dq = (const d2v*)(lds + 18 * rep);
        long long c0 = tick(acc);
        const d2v q0 = Dq[0], q2 = Dq[2], q4 = Dq[4], q5 = Dq[5], q6 = Dq[6], q7 = Dq[7];
        double D00 = q0.x + 8.0, D10 = q2.x, D11 = q2.y + 8.0, D20 = q4.x, D21 = q4.y, D22 = q5.x + 8.0, D30 = q6.x, D31 = q6.y, D32 = q7.x, D33 = q7.y + 8.0;
        long long c1 = tick(D00 + D11 + D22 + D33 + D10 + D20 + D21 + D30 + D31 + D32);
        const double r0 = fast_rcp(D00);
        const double l10 = D10 * r0, l20 = D20 * r0, l30 = D30 * r0;
        D11 = fma(-l10, D10, D11); D21 = fma(-l20, D10, D21); D31 = fma(-l30, D10, D31);
        D22 = fma(-l20, D20, D22); D32 = fma(-l30, D20, D32); D33 = fma(-l30, D30, D33);
        const double r1 = fast_rcp(D11);
        const double l21 = D21 * r1, l31 = D31 * r1;
        D22 = fma(-l21, D21, D22); D32 = fma(-l31, D21, D32); D33 = fma(-l31, D31, D33);
        const double r2 = fast_rcp(D22);
        const double l32 = D32 * r2;
        D33 = fma(-l32, D32, D33);
        const double r3 = fast_rcp(D33);
        long long c2 = tick(r3);
        double a4[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) a4[r][c] = a0 * (r + 1) + c + t;
        d2v* Wo = (d2v*)(lds + 1024 + (size_t)(t % 22) * 18);
        d2v* Lo = (d2v*)(lds + 2048 + (size_t)(t % 22) * 18);
        long long c3 = tick(a4[3][3]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double w0 = a4[r][0];
            const double w1 = fma(-w0, l10, a4[r][1]);
            const double w2 = fma(-w1, l21, fma(-w0, l20, a4[r][2]));
            const double w3 = fma(-w2, l32, fma(-w1, l31, fma(-w0, l30, a4[r][3])));
            Wo[2 * r] = (d2v){w0, w1}; Wo[2 * r + 1] = (d2v){w2, w3};
            Lo[2 * r] = (d2v){w0 * r0, w1 * r1}; Lo[2 * r + 1] = (d2v){w2 * r2, w3 * r3};
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        long long c4 = tick(r3);
        // dependent rcp chain alone
        double z = r3 + 3.0;
        long long c5 = tick(z);
#pragma unroll
        for (int i = 0; i < 4; ++i) z = __builtin_amdgcn_rcp(z) + 2.0;
        long long c6 = tick(z);
        acc += z + lds[1024 + t];
        T[0] += c1 - c0; T[1] += c2 - c1; T[2] += c4 - c3; T[3] += c6 - c5;
    }
    out[t] = acc;
    if (t == 0) for (int i = 0; i < 4; ++i) tt[i] = T[i];
}
int main() {
    double* out; long long* t; hipMalloc(&out, 64 * 8); hipMalloc(&t, 64);
    long long h[4];
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, t, 1.0000001);
        hipMemcpy(h, t, 32, hipMemcpyDeviceToHost);
        printf("per round: 6 x ds_read_b128 %.0f clk | 4x4 LDL^T chain %.0f clk | W/L (30 fma + 16 mul) + 16 ds_write_b128 + wait %.0f clk | 4 x (rcp64 + add) %.0f clk\n",
               h[0] / 16.0, h[1] / 16.0, h[2] / 16.0, h[3] / 16.0);
    }
    return 0;
}
