// launch-boundary probes on gfx950 (hipcc --offload-arch=gfx950 -O2 launch.hip -o launch):
//   1. cost of a dependent kernel boundary inside a hipGraph as a function of the kernarg size (8 B pointer vs a
//      ~1 KB by-value struct, which is what DeviceModel + FrameBuffers amount to) and of the grid size;
//   2. whether hipEventRecord captured into a graph yields usable hipEventElapsedTime after a replay;
//   3. issue rate of v_mfma_f64_16x16x4_f64 (one wave, independent accumulators / one dependent accumulator).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

struct Big { double v[120]; int i[16]; };   // 1024 bytes by value

__global__ void k_small(double* x) { if (threadIdx.x == 0 && blockIdx.x == 0) x[0] += 1.0; }
__global__ void k_big(Big b, double* x) { if (threadIdx.x == 0 && blockIdx.x == 0) x[0] += b.v[3]; }
__global__ void k_ptr(const Big* __restrict__ b, double* x) { if (threadIdx.x == 0 && blockIdx.x == 0) x[0] += b->v[3]; }

typedef double v4 __attribute__((ext_vector_type(4)));
__global__ void k_mfma(double* out, long long* t) {
    v4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    const double f = 1.0 + threadIdx.x * 1e-9;
    long long c0 = clock64();
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(f, f, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(f, f, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(f, f, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(f, f, a3, 0, 0, 0);
    }
    asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
    long long c1 = clock64();
#pragma unroll
    for (int i = 0; i < 128; ++i) a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(f, f, a0, 0, 0, 0);
    asm volatile("" : "+v"(a0));
    long long c2 = clock64();
    out[threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
    if (threadIdx.x == 0) { t[0] = c1 - c0; t[1] = c2 - c1; }
}

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("ERR %s: %s\n", #e, hipGetErrorString(_e)); } } while (0)

template <class F>
double time_graph(hipStream_t s, int nk, F enqueue, int reps = 200) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < nk; ++i) enqueue();
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(a, s));
    for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(b, s));
    CK(hipStreamSynchronize(s));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return ms * 1e3 / reps / nk;     // us per kernel
}

int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    double* x; CK(hipMalloc(&x, 4096)); CK(hipMemset(x, 0, 4096));
    Big hb{}; hb.v[3] = 1.0;
    Big* db; CK(hipMalloc(&db, sizeof(Big))); CK(hipMemcpy(db, &hb, sizeof(Big), hipMemcpyHostToDevice));
    for (int grid : {1, 128, 768}) {
        const double ts = time_graph(s, 40, [&] { hipLaunchKernelGGL(k_small, dim3(grid), dim3(256), 0, s, x); });
        const double tb = time_graph(s, 40, [&] { hipLaunchKernelGGL(k_big, dim3(grid), dim3(256), 0, s, hb, x); });
        const double tp = time_graph(s, 40, [&] { hipLaunchKernelGGL(k_ptr, dim3(grid), dim3(256), 0, s, db, x); });
        printf("graph, %4d workgroups: us per dependent kernel: 8 B kernarg %.2f | 1 KB by-value kernarg %.2f | pointer to the same struct %.2f\n", grid, ts, tb, tp);
    }
    {   // eager (no graph), same chain
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int big = 0; big < 2; ++big) {
            for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_small, dim3(1), dim3(256), 0, s, x);
            CK(hipStreamSynchronize(s));
            CK(hipEventRecord(a, s));
            for (int i = 0; i < 2000; ++i) { if (big) hipLaunchKernelGGL(k_big, dim3(1), dim3(256), 0, s, hb, x); else hipLaunchKernelGGL(k_small, dim3(1), dim3(256), 0, s, x); }
            CK(hipEventRecord(b, s));
            CK(hipStreamSynchronize(s));
            float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
            printf("eager, 1 workgroup, %s kernarg: %.2f us per kernel\n", big ? "1 KB" : "8 B", ms * 1e3 / 2000);
        }
    }
    {   // events captured into a graph
        hipEvent_t e[3]; for (auto& v : e) CK(hipEventCreate(&v));
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        hipError_t r0 = hipEventRecord(e[0], s);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_small, dim3(128), dim3(256), 0, s, x);
        hipError_t r1 = hipEventRecord(e[1], s);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_small, dim3(128), dim3(256), 0, s, x);
        hipError_t r2 = hipEventRecord(e[2], s);
        CK(hipStreamEndCapture(s, &g));
        printf("event capture: record rc = %d %d %d\n", (int)r0, (int)r1, (int)r2);
        hipError_t ri = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        printf("event capture: instantiate rc = %d (%s)\n", (int)ri, hipGetErrorString(ri));
        if (ri == hipSuccess) {
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
                float m01 = -1, m12 = -1;
                hipError_t q0 = hipEventElapsedTime(&m01, e[0], e[1]), q1 = hipEventElapsedTime(&m12, e[1], e[2]);
                printf("event capture: replay %d: elapsed 10 kernels %.2f us (rc %d), 20 kernels %.2f us (rc %d)\n", rep, m01 * 1e3, (int)q0, m12 * 1e3, (int)q1);
            }
        }
    }
    {
        double* out; long long* t; CK(hipMalloc(&out, 64 * 8)); CK(hipMalloc(&t, 64));
        long long h[2];
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, s, out, t);
            CK(hipStreamSynchronize(s));
            CK(hipMemcpy(h, t, 16, hipMemcpyDeviceToHost));
            printf("v_mfma_f64_16x16x4_f64, one wave: %.1f clk per instruction (4 independent accumulators), %.1f clk (one dependent accumulator)\n", h[0] / 256.0, h[1] / 128.0);
        }
    }
    return 0;
}
