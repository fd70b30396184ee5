// micro-benchmark: latency of a producer -> consumer hand-off between two workgroups of ONE launch through global memory
// (release: stores, __threadfence(), atomic increment; acquire: bounded spin on the counter, __threadfence(), loads).
// Blocks 0 and B ping-pong `n` times; prints wall nanoseconds per one-way hop.  Every spin is bounded: a lost hand-off
// ends the kernel with an error count instead of hanging the device.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ bool wait_ge(unsigned* ctr, unsigned want) {
    if (threadIdx.x == 0) {
        int spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want && spins < (1 << 20)) { __builtin_amdgcn_s_sleep(1); ++spins; }
    }
    __syncthreads();
    __threadfence();
    return __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want;
}
__device__ __forceinline__ void signal(unsigned* ctr) {
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// variant without cache-wide fences: payload moved by agent-scope relaxed atomics (write-through stores, cache-bypassing
// loads), ordering by the wave's own memory counters only
__device__ __forceinline__ bool wait_ge_nf(unsigned* ctr, unsigned want) {
    if (threadIdx.x == 0) {
        int spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want && spins < (1 << 20)) { __builtin_amdgcn_s_sleep(1); ++spins; }
    }
    __syncthreads();
    return __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want;
}
__device__ __forceinline__ void signal_nf(unsigned* ctr) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(double* p, double v) { __hip_atomic_store((unsigned long long*)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_agent(const double* p) { return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }

__global__ __launch_bounds__(256) void pingpong_nf(unsigned* ctr, double* data, long long* out, int n, int other) {
    const int role = blockIdx.x == 0 ? 0 : (blockIdx.x == other ? 1 : -1);
    if (role < 0) return;
    int errors = 0;
    const long long w0 = wall_clock64();
    double acc = 0.0;
    for (int i = 0; i < n; ++i) {
        if (role == 0) {
            st_agent(data + threadIdx.x, (double)(i + 1));
            signal_nf(ctr);
            if (!wait_ge_nf(ctr + 32, (unsigned)(i + 1))) ++errors;
            acc += ld_agent(data + 256 + threadIdx.x);
        } else {
            if (!wait_ge_nf(ctr, (unsigned)(i + 1))) ++errors;
            const double v = ld_agent(data + threadIdx.x);
            if (v != (double)(i + 1)) ++errors;
            st_agent(data + 256 + threadIdx.x, v + 0.5);
            signal_nf(ctr + 32);
        }
    }
    if (threadIdx.x == 0) { out[2 * role] = wall_clock64() - w0; out[2 * role + 1] = errors; }
    if (acc == -1.0) out[7] = 1;
}

__global__ __launch_bounds__(256) void pingpong(unsigned* ctr, double* data, long long* out, int n, int other) {
    const int role = blockIdx.x == 0 ? 0 : (blockIdx.x == other ? 1 : -1);
    if (role < 0) return;
    int errors = 0;
    const long long w0 = wall_clock64();
    double acc = 0.0;
    for (int i = 0; i < n; ++i) {
        if (role == 0) {
            data[threadIdx.x] = (double)(i + 1);              // payload the other side must see
            signal(ctr);                                      // ctr[0] = i + 1
            if (!wait_ge(ctr + 32, (unsigned)(i + 1))) ++errors;
            acc += data[256 + threadIdx.x];
        } else {
            if (!wait_ge(ctr, (unsigned)(i + 1))) ++errors;
            const double v = data[threadIdx.x];
            if (v != (double)(i + 1)) ++errors;
            data[256 + threadIdx.x] = v + 0.5;
            signal(ctr + 32);
        }
    }
    if (threadIdx.x == 0) { out[2 * role] = wall_clock64() - w0; out[2 * role + 1] = errors; }
    if (acc == -1.0) out[7] = 1;
}

int main() {
    unsigned* ctr; double* data; long long* out;
    hipMalloc(&ctr, 256); hipMalloc(&data, 512 * 8); hipMalloc(&out, 64);
    for (int other : {1, 8, 37, 200}) {
        hipMemset(ctr, 0, 256); hipMemset(out, 0, 64);
        const int n = 200;
        hipLaunchKernelGGL(pingpong, dim3(256), dim3(256), 0, 0, ctr, data, out, n, other);
        long long h[8]; hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
        printf("blocks 0 <-> %3d: %.0f ns per one-way hop (wall clock 100 MHz), errors %lld %lld\n", other, h[0] * 10.0 / (2.0 * n), h[1], h[3]);
        hipMemset(ctr, 0, 256); hipMemset(out, 0, 64);
        hipLaunchKernelGGL(pingpong_nf, dim3(256), dim3(256), 0, 0, ctr, data, out, n, other);
        hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
        printf("   no cache-wide fences: %.0f ns per one-way hop, errors %lld %lld\n", h[0] * 10.0 / (2.0 * n), h[1], h[3]);
    }
    return 0;
}
