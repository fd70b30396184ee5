// avt_rtree.hip — per-pixel forest inference on gfx950: RTree::predictBest(depth, ...)
// (RTree.cpp:3184-3262) with the grid up-scaling of upscaleGrid (:70-99) folded in.
//
// One lane per pixel of the interval grid inside the region of interest.  The walk is a chain of dependent, data-
// dependent loads (node -> two depth probes -> next node), so the kernel is bound by L2 / MALL latency, not by
// arithmetic; the tree (32 B per node) and the probed part of the image stay cache-resident, and thousands of
// independent walks per CU hide the latency.  Arithmetic is the reference's float32 sequence exactly (this file is
// built with -ffp-contract=off): u / depth per component, round-half-away (std::round), int32 cast, bounds against the
// REGION OF INTEREST, zero depth -> BACKGROUND_DEPTH (RTree.cpp:325), zu - zv < thresh -> left child.
#include "avt_rtree.h"

#define RT_BACKGROUND_DEPTH 20.f

__global__ __launch_bounds__(256) void k_rtree_predict(const RtNodeDev* __restrict__ nodes, const float* __restrict__ depth,
                                                       unsigned char* __restrict__ labels, int rows, int cols, int interval, int tlx, int tly,
                                                       int brx, int bry, int gcols, int grows, int fill) {
    const int img = blockIdx.z;
    const int gc = blockIdx.x * 16 + (threadIdx.x & 15), gr = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (gc >= gcols || gr >= grows) return;
    // r = (row += interval): the reference's row counter is pre-incremented, the first row of the region is skipped
    const int r = tly + interval * (gr + 1), c = tlx + interval * gc;
    const float* d = depth + (size_t)img * rows * cols;
    unsigned char* out = labels + (size_t)img * rows * cols;
    const float sample = d[(size_t)r * cols + c];
    unsigned char lab = 255;
    if (sample != 0.f) {
        int nodeid = 0;
        const float4* nv = (const float4*)nodes;
        for (;;) {
            const float4 a = nv[2 * nodeid], b = nv[2 * nodeid + 1];
            const int lnode = __float_as_int(b.y);
            if (__float_as_int(b.w)) { lab = (unsigned char)lnode; break; }
            const int ux = (int)roundf(__fdiv_rn(a.x, sample)) + c, uy = (int)roundf(__fdiv_rn(a.y, sample)) + r;
            const int vx = (int)roundf(__fdiv_rn(a.z, sample)) + c, vy = (int)roundf(__fdiv_rn(a.w, sample)) + r;
            float zu = RT_BACKGROUND_DEPTH, zv = RT_BACKGROUND_DEPTH;
            if (!(ux < tlx || uy < tly || ux > brx || uy > bry)) { zu = d[(size_t)uy * cols + ux]; if (zu == 0.0f) zu = RT_BACKGROUND_DEPTH; }
            if (!(vx < tlx || vy < tly || vx > brx || vy > bry)) { zv = d[(size_t)vy * cols + vx]; if (zv == 0.0f) zv = RT_BACKGROUND_DEPTH; }
            nodeid = (zu - zv < b.x) ? lnode : __float_as_int(b.z);
        }
    }
    if (fill && interval > 1) {       // upscaleGrid: the cell [r, r+interval) x [c, c+interval), rows <= bot_right.y, width clamped
        for (int rr = r; rr < r + interval && rr <= bry; ++rr)
            for (int cc = c; cc < c + interval && cc < cols; ++cc) out[(size_t)rr * cols + cc] = lab;
    } else if (lab != 255) {
        out[(size_t)r * cols + c] = lab;
    }
}

// RTree::predict(depth) (RTree.cpp:3156-3182): every pixel, probes bounded by the image, the whole leaf distribution out
__global__ __launch_bounds__(256) void k_rtree_predict_dist(const RtNodeDev* __restrict__ nodes, const float* __restrict__ leaf_data,
                                                            const float* __restrict__ depth, float* __restrict__ out, int rows, int cols, int num_parts) {
    const int c = blockIdx.x * 16 + (threadIdx.x & 15), r = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (c >= cols || r >= rows) return;
    const float sample = depth[(size_t)r * cols + c];
    const size_t plane = (size_t)rows * cols, o = (size_t)r * cols + c;
    if (!(sample > 0.f)) {
        for (int i = 0; i < num_parts; ++i) out[(size_t)i * plane + o] = 0.f;
        return;
    }
    int nodeid = 0;
    const float4* nv = (const float4*)nodes;
    int leaf;
    for (;;) {
        const float4 a = nv[2 * nodeid], b = nv[2 * nodeid + 1];
        if (__float_as_int(b.w)) { leaf = __float_as_int(b.z); break; }
        const int ux = (int)roundf(__fdiv_rn(a.x, sample)) + c, uy = (int)roundf(__fdiv_rn(a.y, sample)) + r;
        const int vx = (int)roundf(__fdiv_rn(a.z, sample)) + c, vy = (int)roundf(__fdiv_rn(a.w, sample)) + r;
        float zu = RT_BACKGROUND_DEPTH, zv = RT_BACKGROUND_DEPTH;
        if (!(ux < 0 || uy < 0 || ux >= cols || uy >= rows)) { zu = depth[(size_t)uy * cols + ux]; if (zu == 0.0f) zu = RT_BACKGROUND_DEPTH; }
        if (!(vx < 0 || vy < 0 || vx >= cols || vy >= rows)) { zv = depth[(size_t)vy * cols + vx]; if (zv == 0.0f) zv = RT_BACKGROUND_DEPTH; }
        nodeid = (zu - zv < b.x) ? __float_as_int(b.y) : __float_as_int(b.z);
    }
    const float* d = leaf_data + (size_t)leaf * num_parts;
    for (int i = 0; i < num_parts; ++i) out[(size_t)i * plane + o] = d[i];
}

int avt_rtree_launch_predict_dist(avt_rtree* rt, int rows, int cols, float* d_out) {
    dim3 grid((cols + 15) / 16, (rows + 15) / 16);
    hipLaunchKernelGGL(k_rtree_predict_dist, grid, dim3(256), 0, rt->stream, rt->d_nodes, rt->d_leaf, rt->d_depth, d_out, rows, cols, rt->num_parts);
    return hipGetLastError() != hipSuccess;
}

int avt_rtree_launch_predict(avt_rtree* rt, int n_images, int rows, int cols, int interval, int tlx, int tly, int brx, int bry, int fill) {
    const size_t npix = (size_t)n_images * rows * cols;
    if (hipMemsetAsync(rt->d_labels, 255, npix, rt->stream) != hipSuccess) return 1;
    const int grows = (bry - tly) / interval;                 // rows tly + interval, tly + 2 interval, ... <= bry
    const int gcols = (brx - tlx) / interval + 1;             // cols tlx, tlx + interval, ... <= brx
    if (grows <= 0 || gcols <= 0) return 0;
    dim3 grid((gcols + 15) / 16, (grows + 15) / 16, n_images);
    hipLaunchKernelGGL(k_rtree_predict, grid, dim3(256), 0, rt->stream, rt->d_nodes, rt->d_depth, rt->d_labels, rows, cols, interval, tlx, tly,
                       brx, bry, gcols, grows, fill);
    return hipGetLastError() != hipSuccess;
}
