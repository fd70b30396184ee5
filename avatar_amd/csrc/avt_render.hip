// avt_render.hip — synthetic depth-frame generator on the GPU: depth + body-part render of posed
// avatars and back-projection into (data_cloud, data_part_labels), written straight into the context's resident frame
// buffers so that batched benchmarks need no host rasteriser.
//
// Behavioural counterpart of AvatarRenderer::renderDepth / renderPartMask (AvatarRenderer.cpp:72-101, :174-202), the
// projection of AvatarRenderer.cpp:11-24, CameraIntrin::to3D (Calibration.cpp:68-74, float arithmetic) and the y flip of
// optim.cpp:116-119.  Like the host generator (synth_render.cpp) it resolves visibility with a z-buffer instead of the
// reference's painter's algorithm; the two generators are bit-identical to each other (tests/test_gpu_render.py):
// this translation unit is built with -ffp-contract=off and uses the same float expressions, and depth ties go to the
// lowest face index (64-bit atomicMin on (depth bits, face id)) exactly like the host's in-order strict '<' test.
#include "avt_device.h"

__device__ __forceinline__ void project(const double* p, double fx, double fy, double cx, double cy, float* px, float* py) {
    *px = (float)(p[0] * fx / p[2] + cx);
    *py = (float)(-p[1] * fy / p[2] + cy);
}

__global__ __launch_bounds__(256) void k_raster_clear(unsigned long long* zkey, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) zkey[i] = 0xFFFFFFFFFFFFFFFFull;
}

// one lane per face
__global__ __launch_bounds__(256) void k_raster(DeviceModel dm, FrameBuffers fb, unsigned long long* zkey, double fx, double fy, double cx,
                                                double cy, int width, int height) {
    const int F = dm.d.F, V = dm.d.V;
    const int f = blockIdx.y + fb.f0;
    const int face = blockIdx.x * 256 + threadIdx.x;
    if (face >= F) return;
    const int ia = dm.mesh[face], ib = dm.mesh[(size_t)F + face], ic = dm.mesh[2 * (size_t)F + face];
    const double* cl = fb.cloud + (size_t)f * 3 * V;
    const double* a = cl + 3 * ia; const double* b = cl + 3 * ib; const double* c = cl + 3 * ic;
    const double ab0 = b[0] - a[0], ab1 = b[1] - a[1], ab2 = b[2] - a[2], ac0 = c[0] - a[0], ac1 = c[1] - a[1], ac2 = c[2] - a[2];
    const double n0 = ab1 * ac2 - ab2 * ac1, n1 = ab2 * ac0 - ab0 * ac2, n2 = ab0 * ac1 - ab1 * ac0;
    const double nn = sqrt(n0 * n0 + n1 * n1 + n2 * n2);
    if (!(nn > 0.0) || fabs(n2 / nn) < 0.1) return;             // edge-on faces carry no depth
    if (a[2] <= 0.0 || b[2] <= 0.0 || c[2] <= 0.0) return;
    float ax, ay, bx, by, cxx, cyy;
    project(a, fx, fy, cx, cy, &ax, &ay); project(b, fx, fy, cx, cy, &bx, &by); project(c, fx, fy, cx, cy, &cxx, &cyy);
    const float denom = (by - cyy) * (ax - cxx) + (cxx - bx) * (ay - cyy);
    if (denom == 0.0f) return;
    const float inv = 1.0f / denom;
    const int x0 = max(0, (int)floorf(fminf(ax, fminf(bx, cxx))));
    const int x1 = min(width - 1, (int)ceilf(fmaxf(ax, fmaxf(bx, cxx))));
    const int y0 = max(0, (int)floorf(fminf(ay, fminf(by, cyy))));
    const int y1 = min(height - 1, (int)ceilf(fmaxf(ay, fmaxf(by, cyy))));
    const float az = (float)a[2], bz = (float)b[2], cz = (float)c[2];
    unsigned long long* zk = zkey + (size_t)(f - fb.f0) * width * height;
    for (int r = y0; r <= y1; ++r)
        for (int col = x0; col <= x1; ++col) {
            const float w1 = ((by - cyy) * (col - cxx) + (cxx - bx) * (r - cyy)) * inv;
            const float w2 = ((cyy - ay) * (col - cxx) + (ax - cxx) * (r - cyy)) * inv;
            const float w3 = 1.0f - w1 - w2;
            if (w1 < 0.0f || w2 < 0.0f || w3 < 0.0f) continue;
            const float z = w1 * az + w2 * bz + w3 * cz;
            if (!(z > 0.0f)) continue;
            const unsigned long long key = ((unsigned long long)__float_as_uint(z) << 32) | (unsigned)face;
            atomicMin(zk + (size_t)r * width + col, key);
        }
}

// per pixel: winning face -> part label of its nearest projected vertex; count foreground pixels per 256-pixel block
__global__ __launch_bounds__(256) void k_raster_label(DeviceModel dm, FrameBuffers fb, const unsigned long long* zkey, const int* vertex_part,
                                                      unsigned char* label, int* block_count, double fx, double fy, double cx, double cy,
                                                      int width, int height) {
    const int F = dm.d.F, V = dm.d.V;
    const int fl = blockIdx.y, f = fl + fb.f0;
    const size_t npix = (size_t)width * height;
    const size_t o = (size_t)blockIdx.x * 256 + threadIdx.x;
    int fg = 0;
    if (o < npix) {
        const unsigned long long key = zkey[(size_t)fl * npix + o];
        unsigned char lab = 255;
        if (key != 0xFFFFFFFFFFFFFFFFull) {
            const int face = (int)(key & 0xFFFFFFFFull);
            const int r = (int)(o / width), col = (int)(o % width);
            const int ia = dm.mesh[face], ib = dm.mesh[(size_t)F + face], ic = dm.mesh[2 * (size_t)F + face];
            const double* cl = fb.cloud + (size_t)f * 3 * V;
            float ax, ay, bx, by, cxx, cyy;
            project(cl + 3 * ia, fx, fy, cx, cy, &ax, &ay); project(cl + 3 * ib, fx, fy, cx, cy, &bx, &by);
            project(cl + 3 * ic, fx, fy, cx, cy, &cxx, &cyy);
            const float da = (ax - col) * (ax - col) + (ay - r) * (ay - r);
            const float db = (bx - col) * (bx - col) + (by - r) * (by - r);
            const float dc = (cxx - col) * (cxx - col) + (cyy - r) * (cyy - r);
            lab = (unsigned char)((da < db && da < dc) ? vertex_part[ia] : (db < dc ? vertex_part[ib] : vertex_part[ic]));
            fg = 1;
        }
        label[(size_t)fl * npix + o] = lab;
    }
    const unsigned long long bal = __ballot(fg);
    __shared__ int s_c[4];
    if (lane_id() == 0) s_c[wave_id()] = __popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) block_count[(size_t)fl * gridDim.x + blockIdx.x] = s_c[0] + s_c[1] + s_c[2] + s_c[3];
}

// exclusive scan of the per-block counts, one workgroup per frame; also publishes N
__global__ __launch_bounds__(1024) void k_raster_scan(FrameBuffers fb, int* block_count, int nblocks) {
    const int fl = blockIdx.x, t = threadIdx.x;
    int* bc = block_count + (size_t)fl * nblocks;
    __shared__ int s_w[16];
    __shared__ int s_run;
    if (t == 0) s_run = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nblocks; b0 += 1024) {
        const int i = b0 + t;
        const int v = (i < nblocks) ? bc[i] : 0;
        const int incl = wave_incl_scan(v);
        if (lane_id() == 63) s_w[wave_id()] = incl;
        __syncthreads();
        int off = s_run;
        for (int w = 0; w < wave_id(); ++w) off += s_w[w];
        if (i < nblocks) bc[i] = off + incl - v;
        __syncthreads();
        if (t == 1023) s_run = off + incl;
        __syncthreads();
    }
    if (t == 0) {
        const int N = min(s_run, fb.max_points);
        fb.ctl[fl + fb.f0].N = N;
        fb.ctl[fl + fb.f0].T = s_run;       // total foreground pixels (reported even when truncated to max_points)
    }
}

// ordered back-projection of the foreground pixels (row-major pixel order, as the host loop at optim.cpp:100-120)
__global__ __launch_bounds__(256) void k_raster_emit(FrameBuffers fb, const unsigned long long* zkey, const unsigned char* label,
                                                     const int* block_off, float ffx, float ffy, float fcx, float fcy, int width, int height) {
    const int fl = blockIdx.y, f = fl + fb.f0;
    const size_t npix = (size_t)width * height;
    const size_t o = (size_t)blockIdx.x * 256 + threadIdx.x;
    bool fg = false;
    unsigned long long key = 0;
    if (o < npix) { key = zkey[(size_t)fl * npix + o]; fg = key != 0xFFFFFFFFFFFFFFFFull; }
    const unsigned long long bal = __ballot(fg);
    __shared__ int s_c[4];
    if (lane_id() == 0) s_c[wave_id()] = __popcll(bal);
    __syncthreads();
    if (!fg) return;
    int pos = block_off[(size_t)fl * gridDim.x + blockIdx.x] + __popcll(bal & ((1ull << lane_id()) - 1ull));
    for (int w = 0; w < wave_id(); ++w) pos += s_c[w];
    if (pos >= fb.max_points) return;
    const int r = (int)(o / width), col = (int)(o % width);
    const float depth = __uint_as_float((unsigned)(key >> 32));
    const float X = ((float)col - fcx) * depth / ffx;       // CameraIntrin::to3D (Calibration.cpp:68-74)
    const float Y = ((float)r - fcy) * depth / ffy;
    double* d = fb.data_raw + 3 * ((size_t)f * fb.max_points + pos);
    d[0] = (double)X; d[1] = -(double)Y; d[2] = (double)depth;  // y negated (optim.cpp:116-119)
    fb.labels_raw[(size_t)f * fb.max_points + pos] = (int)label[(size_t)fl * npix + o];
}

int avt_render_enqueue(avt_ctx* c, int nframes, const int* d_vertex_part, unsigned long long* d_zkey, unsigned char* d_label, int* d_block,
                       double fx, double fy, double cx, double cy, int width, int height) {
    const size_t npix = (size_t)width * height;
    const int nb = (int)((npix + 255) / 256);
    hipStream_t s = c->cur_stream;
    hipLaunchKernelGGL(k_raster_clear, dim3((unsigned)((npix * nframes + 255) / 256)), dim3(256), 0, s, d_zkey, npix * nframes);
    hipLaunchKernelGGL(k_raster, dim3((c->dm.d.F + 255) / 256, nframes), dim3(256), 0, s, c->dm, c->fb, d_zkey, fx, fy, cx, cy, width, height);
    hipLaunchKernelGGL(k_raster_label, dim3(nb, nframes), dim3(256), 0, s, c->dm, c->fb, d_zkey, d_vertex_part, d_label, d_block, fx, fy, cx, cy,
                       width, height);
    hipLaunchKernelGGL(k_raster_scan, dim3(nframes), dim3(1024), 0, s, c->fb, d_block, nb);
    hipLaunchKernelGGL(k_raster_emit, dim3(nb, nframes), dim3(256), 0, s, c->fb, d_zkey, d_label, d_block, (float)fx, (float)fy, (float)cx,
                       (float)cy, width, height);
    return hipGetLastError() != hipSuccess;
}
