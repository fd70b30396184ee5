// avt_bucket.h - the two passes of the data bucketing as device functions: they ride in the trailing workgroups of launches
// they do not depend on (k_lbs, k_visibility / k_compact) and exist as thin kernels of their own for the stand-alone paths.
#pragma once
#include "avt_device.h"

// =================================================================================================
// Data bucketing by body-part label (the data-side counterpart of AvatarOptimizer.cpp:1274-1293): a
// two-pass counting sort over many workgroups.  Pass 1 histograms labels (LDS atomics, then one global atomic
// per (workgroup, part)); pass 2 reserves a range per (workgroup, part) and scatters.  The order of points
// INSIDE a part bucket is not deterministic, and nothing downstream depends on it: the nearest neighbour of a
// point does not depend on its neighbours, the correspondence sums are order-independent integer atomics and
// every floating-point reduction over data points runs in original index order.
// =================================================================================================
#define BUCKET_TILE 2048

__device__ __forceinline__ void bucket_count_block(const DeviceModel& dm, const FrameBuffers& fb, int f, int bx) {
    const int t = threadIdx.x, np = dm.d.num_parts;
    const int N = fb.ctl[f].N;
    const int s0 = bx * BUCKET_TILE;
    if (s0 >= N) return;
    __shared__ int hist[AVT_MAX_PARTS + 1];
    if (t <= np) hist[t] = 0;
    __syncthreads();
    const int* lab = fb.labels_raw + (size_t)f * fb.max_points;
#pragma unroll
    for (int u = 0; u < BUCKET_TILE / 256; ++u) {
        const int i = s0 + u * 256 + t;
        if (i < N) {
            int q = lab[i];
            if (q < 0 || q >= np) q = np;
            atomicAdd(&hist[q], 1);
        }
    }
    __syncthreads();
    if (t <= np && hist[t]) atomicAdd(fb.part_cnt + (size_t)f * 2 * (AVT_MAX_PARTS + 1) + t, hist[t]);
}

__device__ __forceinline__ void bucket_scatter_block(const DeviceModel& dm, const FrameBuffers& fb, int f, int bx) {
    const int t = threadIdx.x, np = dm.d.num_parts;
    AvtFrameCtl& ctl = fb.ctl[f];
    const int N = ctl.N;
    const int s0 = bx * BUCKET_TILE;
    const size_t base = (size_t)f * fb.max_points;
    __shared__ int hist[AVT_MAX_PARTS + 1], poff[AVT_MAX_PARTS + 2], bbase[AVT_MAX_PARTS + 1];
    int* pcnt = fb.part_cnt + (size_t)f * 2 * (AVT_MAX_PARTS + 1);
    int* cursor = pcnt + (AVT_MAX_PARTS + 1);
    if (t <= np) hist[t] = 0;
    if (t == 0) {
        int acc = 0;
        for (int q = 0; q <= np; ++q) { poff[q] = acc; acc += pcnt[q]; }
        poff[np + 1] = acc;
    }
    __syncthreads();
    if (bx == 0) {
        if (t <= np) fb.part_off[(size_t)f * (np + 1) + t] = poff[t];
        if (t < 3 && N > 0) ctl.centre[t] = fb.data_raw[3 * base + t];
    }
    if (s0 >= N) return;
    const int* lab = fb.labels_raw + base;
    int qs[BUCKET_TILE / 256];
#pragma unroll
    for (int u = 0; u < BUCKET_TILE / 256; ++u) {
        const int i = s0 + u * 256 + t;
        int q = -1;
        if (i < N) {
            q = lab[i];
            if (q < 0 || q >= np) q = np;
            atomicAdd(&hist[q], 1);
        }
        qs[u] = q;
    }
    __syncthreads();
    if (t <= np) {
        bbase[t] = hist[t] ? poff[t] + atomicAdd(cursor + t, hist[t]) : 0;
        hist[t] = 0;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < BUCKET_TILE / 256; ++u) {
        const int i = s0 + u * 256 + t;
        const int q = qs[u];
        if (q < 0) continue;
        const int pos = bbase[q] + atomicAdd(&hist[q], 1);
        fb.dx[base + pos] = fb.data_raw[3 * (base + i)];
        fb.dy[base + pos] = fb.data_raw[3 * (base + i) + 1];
        fb.dz[base + pos] = fb.data_raw[3 * (base + i) + 2];
        fb.dorig[base + pos] = i;
        if (q == np) fb.corr[base + i] = -1;
    }
}

