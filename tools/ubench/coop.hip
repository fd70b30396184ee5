// coop.hip - what a persistent (one launch for all GN iterations) few-frames loop could cost on gfx950:
//   1. is hipLaunchCooperativeKernel accepted inside stream capture (hipGraph)?  what does a cooperative launch cost when
//      launched eagerly back to back on an in-order stream, against an ordinary launch and against a graph kernel node?
//   2. the floor of one GN iteration as a ring of three in-launch hand-overs (128 "evaluation" workgroups -> 84 "strip"
//      workgroups -> 1 "solver" -> the evaluation workgroups again) with agent-scope write-through stores, agent-scope loads
//      and one counter per hop, no work in between: microseconds per iteration, against two dependent graph kernel nodes.
// hipcc --offload-arch=gfx950 -O2 coop.hip -o coop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("ERR %s: %s\n", #e, hipGetErrorString(_e)); } } while (0)

__global__ void k_tiny(double* x) { if (threadIdx.x == 0 && blockIdx.x == 0) x[0] += 1.0; }

__device__ __forceinline__ unsigned ld_ctr(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_agent(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// roles by blockIdx.x: 0 solver, 1..NS strips, then NE evaluation workgroups.  payload: every producer stores 256 doubles
// (one per thread), every consumer reads 256 doubles of some producer.
template <int SL>
__global__ __launch_bounds__(256) void k_ring(int NS, int NE, int iters, unsigned* ctr /* [3]: state_seq, ev_ctr, st_ctr */, double* buf, long long* t_out) {
    const int b = blockIdx.x, t = threadIdx.x;
    double* xbuf = buf;                          // solver -> evaluation: 2048 doubles
    double* ebuf = buf + 4096;                   // evaluation -> strips: NE x 256
    double* sbuf = ebuf + (size_t)NE * 256;      // strips -> solver: NS x 256
    double acc = 0.0;
    const long long t0 = wall_clock64();
    for (int it = 1; it <= iters; ++it) {
        if (b == 0) {                            // solver: wait for the strips, read, publish the state
            if (t == 0) while (ld_ctr(ctr + 2) < (unsigned)(it * NS)) __builtin_amdgcn_s_sleep(SL);
            __syncthreads();
            for (int k = 0; k < 4; ++k) acc += ld_agent(sbuf + (size_t)((t + k) % NS) * 256 + t);
            for (int k = 0; k < 8; ++k) st_agent(xbuf + k * 256 + t, acc + k);
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
            if (t == 0) __hip_atomic_store(ctr + 0, (unsigned)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (b <= NS) {                    // strip: wait for the evaluation workgroups, read a few, publish
            if (t == 0) while (ld_ctr(ctr + 1) < (unsigned)(it * NE)) __builtin_amdgcn_s_sleep(SL);
            __syncthreads();
            for (int k = 0; k < 4; ++k) acc += ld_agent(ebuf + (size_t)((b + 7 * k) % NE) * 256 + t);
            st_agent(sbuf + (size_t)(b - 1) * 256 + t, acc);
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
            if (t == 0) __hip_atomic_fetch_add(ctr + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {                                 // evaluation: wait for the state of the previous iteration, read, publish
            if (t == 0) while (ld_ctr(ctr + 0) < (unsigned)(it - 1)) __builtin_amdgcn_s_sleep(SL);
            __syncthreads();
            for (int k = 0; k < 8; ++k) acc += ld_agent(xbuf + k * 256 + t);
            st_agent(ebuf + (size_t)(b - 1 - NS) * 256 + t, acc);
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
            if (t == 0) __hip_atomic_fetch_add(ctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (b == 0 && t == 0) t_out[0] = wall_clock64() - t0;
    if (acc == 1.2345) buf[0] = acc;
}

int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    double* x; CK(hipMalloc(&x, 1 << 22)); CK(hipMemset(x, 0, 1 << 22));
    unsigned* ctr; CK(hipMalloc(&ctr, 64)); long long* tt; CK(hipMalloc(&tt, 64));
    int coop = 0; CK(hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, 0));
    printf("cooperative launch supported: %d\n", coop);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float ms;
    // ---- 1a. eager ordinary launches back to back
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(a, s));
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, x);
        CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s)); CK(hipEventElapsedTime(&ms, a, b));
    }
    printf("ordinary launch, eager, in-order stream: %.2f us per kernel\n", ms * 1e3 / 200);
    // ---- 1b. eager cooperative launches
    void* args[] = {&x};
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(a, s));
        for (int i = 0; i < 200; ++i) CK(hipLaunchCooperativeKernel((const void*)k_tiny, dim3(1), dim3(64), args, 0, s));
        CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s)); CK(hipEventElapsedTime(&ms, a, b));
    }
    printf("cooperative launch, eager: %.2f us per kernel\n", ms * 1e3 / 200);
    // ---- 1c. cooperative launch inside a graph capture
    {
        hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        hipError_t e1 = hipLaunchCooperativeKernel((const void*)k_tiny, dim3(1), dim3(64), args, 0, s);
        hipError_t e2 = hipLaunchCooperativeKernel((const void*)k_tiny, dim3(1), dim3(64), args, 0, s);
        hipError_t e3 = hipStreamEndCapture(s, &g);
        printf("capture of cooperative launches: launch %s / %s, end capture %s\n", hipGetErrorString(e1), hipGetErrorString(e2), hipGetErrorString(e3));
        (void)hipGetLastError();
        if (e3 == hipSuccess && g) {
            hipError_t e4 = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            printf("instantiate: %s\n", hipGetErrorString(e4));
            if (e4 == hipSuccess) {
                for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, s));
                CK(hipStreamSynchronize(s));
                CK(hipEventRecord(a, s));
                for (int i = 0; i < 100; ++i) CK(hipGraphLaunch(ge, s));
                CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s)); CK(hipEventElapsedTime(&ms, a, b));
                printf("graph of 2 cooperative nodes: %.2f us per kernel\n", ms * 1e3 / 200);
            }
        }
    }
    // ---- 1d. ordinary graph node chain (reference)
    {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, x);
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(a, s));
        for (int i = 0; i < 100; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s)); CK(hipEventElapsedTime(&ms, a, b));
        printf("graph of 20 ordinary nodes: %.2f us per kernel\n", ms * 1e3 / 2000);
    }
    // ---- 2. the ring (polling interval: s_sleep 1 / 8 / 32 = 64 / 512 / 2048 clocks)
    for (int sl : {1, 8, 32}) for (int NE : {64, 128}) for (int NS : {84, 168}) {
        if (1 + NS + NE > 256) continue;
        const int iters = 200;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemsetAsync(ctr, 0, 64, s));
            int ns = NS, ne = NE, it = iters; double* xb = x;
            void* ra[] = {&ns, &ne, &it, &ctr, &xb, &tt};
            const void* fn = sl == 1 ? (const void*)k_ring<1> : sl == 8 ? (const void*)k_ring<8> : (const void*)k_ring<32>;
            CK(hipEventRecord(a, s));
            CK(hipLaunchCooperativeKernel(fn, dim3(1 + NS + NE), dim3(256), ra, 0, s));
            CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s)); CK(hipEventElapsedTime(&ms, a, b));
        }
        long long ticks; CK(hipMemcpy(&ticks, tt, 8, hipMemcpyDeviceToHost));
        printf("ring s_sleep(%d) NE=%d NS=%d: %.2f us per iteration (in-kernel clock %.2f)\n", sl, NE, NS, ms * 1e3 / iters, ticks / 100.0 / iters);
    }
    return 0;
}
