// micro-check: v_mov_b32_dpp row_newbcast:n on gfx950 broadcasts lane n of every row of 16 lanes to the whole row
// (used by k_eval's row builder to hand the carried points from the four lanes that computed them to the point's 16 lanes).
// Build: hipcc -O3 --offload-arch=gfx950 -o dpp_bcast dpp_bcast.hip ; prints "ok" or the first mismatch.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N>
__device__ double row_bcast(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x150 + N, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x150 + N, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__global__ void k(double* out, const double* in) {
    const double v = in[threadIdx.x];
    out[threadIdx.x] = row_bcast<2>(v);
    out[64 + threadIdx.x] = row_bcast<15>(v);
    // under a partial EXEC mask: only odd lanes execute; the source lane (2, even) is switched off
    double w = -1.0;
    if (threadIdx.x & 1) w = row_bcast<2>(v);
    out[128 + threadIdx.x] = w;
}
int main() {
    double h[64], o[192], *di, *dout;
    for (int i = 0; i < 64; ++i) h[i] = 1000.0 + i * 1.25;
    hipMalloc(&di, sizeof h); hipMalloc(&dout, sizeof o);
    hipMemcpy(di, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dout, di);
    hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64 && !bad; ++i) {
        if (o[i] != h[(i & ~15) + 2]) { printf("row_newbcast:2 lane %d got %g want %g\n", i, o[i], h[(i & ~15) + 2]); bad = 1; }
        if (o[64 + i] != h[(i & ~15) + 15]) { printf("row_newbcast:15 lane %d got %g\n", i, o[64 + i]); bad = 1; }
    }
    printf("partial EXEC (source lane off): lane 1 got %g, lane 17 got %g (source values %g, %g)\n", o[129], o[128 + 17], h[2], h[18]);
    printf(bad ? "MISMATCH\n" : "ok\n");
    return bad;
}
