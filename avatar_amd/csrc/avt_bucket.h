// avt_bucket.h - the two passes of the data bucketing as device functions: they ride in the trailing workgroups of launches
// they do not depend on (k_lbs, k_visibility / k_compact) and exist as thin kernels of their own for the stand-alone paths.
#pragma once
#include "avt_device.h"

// =================================================================================================
// Data bucketing by body-part label (the data-side counterpart of AvatarOptimizer.cpp:1274-1293): a
// two-pass counting sort over many workgroups.  Pass 1 histograms labels (LDS atomics, then one global atomic
// per (workgroup, part) plus the tile's own histogram); pass 2 scatters STABLY: a point's position inside its part's bucket is
// the number of points of that part in front of it in the frame (earlier tiles: the tile histograms of pass 1; inside the tile:
// ranks by ballot, pass by pass and wave by wave).  Nothing downstream needs the order for correctness - the nearest neighbour of
// a point does not depend on its neighbours, the correspondence sums are integer atomics, every floating-point reduction over
// data points runs in original index order - but the order is the original pixel order (row-major), so 64 consecutive bucketed
// queries of a part are a thin slab in y: that is what the slab scan of k_nn_part prunes with (avt_nn.hip; until round 3 the
// scatter took its ranges and slots from atomics, and a wave's queries came from anywhere inside a 2048-point tile:
// tools/nn_slab_sim.py, 75 % of the candidates evaluated against 53 %).
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
    if (t <= np) {
        fb.tile_hist[((size_t)f * fb.bucket_tiles + bx) * (AVT_MAX_PARTS + 1) + t] = hist[t];     // every tile below N writes its row
        if (hist[t]) atomicAdd(fb.part_cnt + (size_t)f * 2 * (AVT_MAX_PARTS + 1) + t, hist[t]);
    }
}

// STABLE: pixel order inside a part (frame batches: the slab scan of k_nn_part wants it); otherwise slots come from atomics (few frames:
// k_nn_vis scans every candidate anyway, and the scatter rides in k_visibility on the frame's dependency chain: 8.2 against 12.9 us)
template <bool STABLE>
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
    if constexpr (STABLE) {
        const int* lab = fb.labels_raw + base;
        // ranks inside the tile: slot (pass u, wave w) counts its points per part; a point's rank = points of its part in earlier
        // slots + lanes of its part below it in its own wave (one ballot per distinct label of the wave: neighbouring pixels share labels)
        __shared__ int slot_cnt[(BUCKET_TILE / 64) * (AVT_MAX_PARTS + 1)];
        for (int e = t; e < (BUCKET_TILE / 64) * (np + 1); e += 256) slot_cnt[e] = 0;
        if (t <= np) {       // points of the part in the tiles in front of this one (pass 1 wrote the tile histograms)
            int acc = poff[t];
            const int* th = fb.tile_hist + (size_t)f * fb.bucket_tiles * (AVT_MAX_PARTS + 1) + t;
            for (int b = 0; b < bx; ++b) acc += th[(size_t)b * (AVT_MAX_PARTS + 1)];
            bbase[t] = acc;
        }
        __syncthreads();
        int qs[BUCKET_TILE / 256], rk[BUCKET_TILE / 256];
        const int w = t >> 6, lane = t & 63;
    #pragma unroll
        for (int u = 0; u < BUCKET_TILE / 256; ++u) {
            const int i = s0 + u * 256 + t;
            int q = -1;
            if (i < N) {
                q = lab[i];
                if (q < 0 || q >= np) q = np;
            }
            qs[u] = q;
            int rank = 0;
            unsigned long long todo = __ballot(q >= 0);
            while (todo) {                                   // wave-uniform: one round per distinct label present in the wave
                const int src = __ffsll((long long)todo) - 1;
                const int qq = __shfl(q, src, 64);
                const unsigned long long m = __ballot(q == qq);
                if (q == qq) rank = __popcll(m & ((1ull << lane) - 1ull));
                if (lane == src) slot_cnt[(u * 4 + w) * (np + 1) + qq] = __popcll(m);
                todo &= ~m;
            }
            rk[u] = rank;
        }
        __syncthreads();
        if (t <= np) {       // exclusive prefix over the 32 slots, in (pass, wave) = index order
            int run = 0;
            for (int sl = 0; sl < BUCKET_TILE / 64; ++sl) { const int c = slot_cnt[sl * (np + 1) + t]; slot_cnt[sl * (np + 1) + t] = run; run += c; }
        }
        __syncthreads();
    #pragma unroll
        for (int u = 0; u < BUCKET_TILE / 256; ++u) {
            const int i = s0 + u * 256 + t;
            const int q = qs[u];
            if (q < 0) continue;
            const int pos = bbase[q] + slot_cnt[(u * 4 + w) * (np + 1) + q] + rk[u];
            fb.dx[base + pos] = fb.data_raw[3 * (base + i)];
            fb.dy[base + pos] = fb.data_raw[3 * (base + i) + 1];
            fb.dz[base + pos] = fb.data_raw[3 * (base + i) + 2];
            fb.dorig[base + pos] = i;
            if (q == np) fb.corr[base + i] = -1;
        }
        return;
    }
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
