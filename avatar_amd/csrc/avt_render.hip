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

// =====================================================================================================================
// Painter's-order mode: the REFERENCE's renderer, pixel for pixel (oracle/render_oracle.cpp restates the same lines).
//   * faces painted in order of decreasing mean depth, later faces overwrite earlier ones, no depth test
//     (AvatarRenderer.cpp:39-70, :72-101, :174-202)
//   * renderDepth: scanline fill over rows, barycentric depth from the FLOORED first / CEILED last vertex
//     (paintTriangleBary, AvatarHelpers.cpp:61-139); edge-on faces (|n_z| < 0.1) paint 0 with the end-EXCLUSIVE row fill of
//     paintTriangleSingleColor (AvatarHelpers.cpp:247-303)
//   * renderPartMask: scanline fill over COLUMNS, label of the nearest projected vertex by int-truncated squared distances
//     (paintPartsTriangleNN, AvatarHelpers.cpp:153-245); edge-on faces paint 255
// A pixel's final value is the value painted by the LAST covering face of the painter's order, and what a face paints at a
// pixel is a pure function of (face, row, column).  So instead of painting 13 776 faces one after the other, every face
// marks its coverage with atomicMax(order position << 32 | face) - one key image per output, because the two fills cover
// different pixel sets - and a resolve pass evaluates the winner's value with the reference's float / double expression
// order (this translation unit is built with -ffp-contract=off).  Equal sort keys: the reference's std::sort leaves their order
// unspecified; here ties go by ascending face id (a stable sort) - tests assert the fixtures are insensitive to it.
// =====================================================================================================================

// float / double -> int as the reference's x86-64 build converts them (cvttss2si / cvttsd2si: NaN and out-of-range values
// give INT_MIN; the GPU's own conversion saturates instead)
__device__ __forceinline__ int cvt_x86(float v) { return (v >= -2147483648.0f && v < 2147483648.0f) ? (int)v : (int)0x80000000; }
__device__ __forceinline__ int cvt_x86(double v) { return (v >= -2147483648.0 && v < 2147483648.0) ? (int)v : (int)0x80000000; }

struct PaintTri { float ax, ay, bx, by, cx, cy; int ia, ib, ic; };   // sorted vertices (a floored / c ceiled by the caller), sorted slots

// std::sort of three (double(coordinate), slot) pairs, lexicographic like std::pair's operator<
__device__ __forceinline__ void sort3(const float k[3], int o[3]) {
    o[0] = 0; o[1] = 1; o[2] = 2;
    auto less = [&](int x, int y) { return (double)k[x] < (double)k[y] || (!((double)k[y] < (double)k[x]) && x < y); };
    if (less(o[1], o[0])) { const int t = o[0]; o[0] = o[1]; o[1] = t; }
    if (less(o[2], o[1])) { const int t = o[1]; o[1] = o[2]; o[2] = t; }
    if (less(o[1], o[0])) { const int t = o[0]; o[0] = o[1]; o[1] = t; }
}

__device__ __forceinline__ void face_projection(const DeviceModel& dm, const double* cl, int face, double fx, double fy, double cx, double cy,
                                                float px[3], float py[3], int vid[3]) {
    const int F = dm.d.F;
    vid[0] = dm.mesh[face]; vid[1] = dm.mesh[(size_t)F + face]; vid[2] = dm.mesh[2 * (size_t)F + face];
    for (int k = 0; k < 3; ++k) project(cl + 3 * (size_t)vid[k], fx, fy, cx, cy, &px[k], &py[k]);
}

// sort key (AvatarRenderer.cpp:62-66: doubles summed, divided by 3.f, stored as float) and the edge-on flag
// (AvatarRenderer.cpp:88-90 with Eigen 3.3's normalized(): a zero vector is returned unchanged, so a degenerate face is edge-on)
__global__ __launch_bounds__(256) void k_paint_keys(DeviceModel dm, FrameBuffers fb, float* fkey, unsigned char* fedge) {
    const int F = dm.d.F, V = dm.d.V;
    const int fl = blockIdx.y, f = fl + fb.f0;
    const int face = blockIdx.x * 256 + threadIdx.x;
    if (face >= F) return;
    const int ia = dm.mesh[face], ib = dm.mesh[(size_t)F + face], ic = dm.mesh[2 * (size_t)F + face];
    const double* cl = fb.cloud + (size_t)f * 3 * V;
    const double* a = cl + 3 * (size_t)ia; const double* b = cl + 3 * (size_t)ib; const double* c = cl + 3 * (size_t)ic;
    fkey[(size_t)fl * F + face] = (float)((a[2] + b[2] + c[2]) / 3.f);
    const double ab0 = b[0] - a[0], ab1 = b[1] - a[1], ab2 = b[2] - a[2], ac0 = c[0] - a[0], ac1 = c[1] - a[1], ac2 = c[2] - a[2];
    const double n0 = ab1 * ac2 - ab2 * ac1, n1 = ab2 * ac0 - ab0 * ac2, n2 = ab0 * ac1 - ab1 * ac0;
    const double z = n0 * n0 + n1 * n1 + n2 * n2;
    const double nz = z > 0.0 ? n2 / sqrt(z) : n2;
    fedge[(size_t)fl * F + face] = fabs(nz) < 0.1 ? 1 : 0;
}

// position of every face in the painter's order = number of faces painted before it: those with a larger key, and those
// with an equal key and a smaller face id.  All lanes read the same key at the same time (scalar loads), 13 776 steps.
__global__ __launch_bounds__(256) void k_paint_rank(int F, const float* __restrict__ fkey, int* __restrict__ frank) {
    const int fl = blockIdx.y;
    const int face = blockIdx.x * 256 + threadIdx.x;
    const float* key = fkey + (size_t)fl * F;
    const float mine = face < F ? key[face] : 0.f;
    int before = 0;
    for (int g = 0; g < F; ++g) {
        const float k = key[g];
        before += (k > mine || (k == mine && g < face)) ? 1 : 0;
    }
    if (face < F) frank[(size_t)fl * F + face] = before;
}

// coverage of the row fills (paintTriangleBary when bary, paintTriangleSingleColor otherwise): calls mark(row, col)
template <bool BARY, class Mark>
__device__ __forceinline__ void cover_rows(const float px[3], const float py[3], int W, int H, Mark mark) {
    int o[3];
    sort3(py, o);
    float ax = px[o[0]], ay = floorf(py[o[0]]), bx = px[o[1]], by = py[o[1]], cx = px[o[2]], cy = ceilf(py[o[2]]);
    if (ay == cy) return;
    const int minyi = max(cvt_x86(ay), 0), maxyi = min(cvt_x86(cy), H - 1), midyi = cvt_x86(floorf(by));
    for (int half = 0; half < 2; ++half) {
        if (half == 0 ? !(ay != by) : !(by != cy)) continue;
        int i0, i1;
        if (half == 0) { i0 = minyi; i1 = min(midyi, H - 1); }
        else { i0 = max(midyi, 0) + (BARY ? (ay != by ? 1 : 0) : 1); i1 = maxyi; }
        if (BARY) {
            float mhi = (cx - ax) / (cy - ay), bhi = ax - ay * mhi, mlo, blo;
            if (half == 0) { mlo = (bx - ax) / (by - ay); blo = ax - ay * mlo; }
            else { mlo = (cx - bx) / (cy - by); blo = bx - by * mlo; }
            if (half == 0 ? bx > cx : bx > ax) { float t = mlo; mlo = mhi; mhi = t; t = blo; blo = bhi; bhi = t; }
            for (int i = i0; i <= i1; ++i) {
                const int minxi = max(cvt_x86(floorf(mlo * (float)i + blo)), 0), maxxi = min(cvt_x86(ceilf(mhi * (float)i + bhi)), W - 1);
                for (int j = minxi; j <= maxxi; ++j) mark(i, j);
            }
        } else {
            double mhi = (double)((cx - ax) / (cy - ay)), bhi = (double)ax - (double)ay * mhi, mlo, blo;
            if (half == 0) { mlo = (double)((bx - ax) / (by - ay)); blo = (double)ax - (double)ay * mlo; }
            else { mlo = (double)((cx - bx) / (cy - by)); blo = (double)bx - (double)by * mlo; }
            if (half == 0 ? bx > cx : bx > ax) { double t = mlo; mlo = mhi; mhi = t; t = blo; blo = bhi; bhi = t; }
            for (int i = i0; i <= i1; ++i) {
                const int minxi = max(cvt_x86(floor(mlo * (double)i + blo)), 0), maxxi = min(cvt_x86(ceil(mhi * (double)i + bhi)), W - 1);
                for (int j = minxi; j < maxxi; ++j) mark(i, j);          // std::fill(ptr + minxi, ptr + maxxi): end exclusive
            }
        }
    }
}

// coverage of the column fill of paintPartsTriangleNN: calls mark(row, col)
template <class Mark>
__device__ __forceinline__ void cover_cols(const float px[3], const float py[3], int W, int H, Mark mark) {
    int o[3];
    sort3(px, o);
    float ax = floorf(px[o[0]]), ay = py[o[0]], bx = px[o[1]], by = py[o[1]], cx = ceilf(px[o[2]]), cy = py[o[2]];
    if (ax == cx) return;
    const int minxi = max(cvt_x86(ax), 0), maxxi = min(cvt_x86(cx), W - 1), midxi = cvt_x86(floorf(bx));
    for (int half = 0; half < 2; ++half) {
        if (half == 0 ? !(ax != bx) : !(bx != cx)) continue;
        const int i0 = half == 0 ? minxi : max(midxi, 0) + 1, i1 = half == 0 ? min(midxi, W - 1) : maxxi;
        double mhi = (double)((cy - ay) / (cx - ax)), bhi = (double)ay - (double)ax * mhi, mlo, blo;
        if (half == 0) { mlo = (double)((by - ay) / (bx - ax)); blo = (double)ay - (double)ax * mlo; }
        else { mlo = (double)((cy - by) / (cx - bx)); blo = (double)by - (double)bx * mlo; }
        if (half == 0 ? by > cy : by > ay) { double t = mlo; mlo = mhi; mhi = t; t = blo; blo = bhi; bhi = t; }
        for (int i = i0; i <= i1; ++i) {
            const int minyi = max(cvt_x86(floor(mlo * (double)i + blo)), 0), maxyi = min(cvt_x86(ceil(mhi * (double)i + bhi)), H - 1);
            for (int j = minyi; j <= maxyi; ++j) mark(j, i);
        }
    }
}

// one lane per face: marks its coverage in both key images
__global__ __launch_bounds__(256) void k_paint_cover(DeviceModel dm, FrameBuffers fb, const int* __restrict__ frank, const unsigned char* __restrict__ fedge,
                                                     unsigned long long* dkey, unsigned long long* mkey, double fx, double fy, double cx,
                                                     double cy, int width, int height) {
    const int F = dm.d.F, V = dm.d.V;
    const int fl = blockIdx.y, f = fl + fb.f0;
    const int face = blockIdx.x * 256 + threadIdx.x;
    if (face >= F) return;
    float px[3], py[3]; int vid[3];
    face_projection(dm, fb.cloud + (size_t)f * 3 * V, face, fx, fy, cx, cy, px, py, vid);
    const size_t npix = (size_t)width * height;
    const unsigned long long key = ((unsigned long long)(unsigned)(frank[(size_t)fl * F + face] + 1) << 32) | (unsigned)face;
    unsigned long long* dk = dkey + (size_t)fl * npix;
    unsigned long long* mk = mkey + (size_t)fl * npix;
    const bool eo = fedge[(size_t)fl * F + face] != 0;
    auto mark_d = [&](int r, int c) { atomicMax(dk + (size_t)r * width + c, key); };
    auto mark_m = [&](int r, int c) { atomicMax(mk + (size_t)r * width + c, key); };
    if (eo) {
        cover_rows<false>(px, py, width, height, mark_d);
        cover_rows<false>(px, py, width, height, mark_m);
    } else {
        cover_rows<true>(px, py, width, height, mark_d);
        cover_cols(px, py, width, height, mark_m);
    }
}

// per pixel: what the winning faces painted.  Writes the two reference images (float depth, 0 = background; uint8 part mask,
// 255 = background), rewrites the depth key into the z-buffer generator's format (depth bits << 32, all ones = background) so
// that k_raster_scan / k_raster_emit back-project it unchanged, and counts the pixels with depth > 0 per 256-pixel block.
__global__ __launch_bounds__(256) void k_paint_resolve(DeviceModel dm, FrameBuffers fb, unsigned long long* dkey, const unsigned long long* mkey,
                                                       const unsigned char* __restrict__ fedge, const int* vertex_part, float* depth_img,
                                                       unsigned char* label, int* block_count, double fx, double fy, double cx, double cy,
                                                       int width, int height) {
    const int F = dm.d.F, V = dm.d.V;
    const int fl = blockIdx.y, f = fl + fb.f0;
    const size_t npix = (size_t)width * height;
    const size_t o = (size_t)blockIdx.x * 256 + threadIdx.x;
    const double* cl = fb.cloud + (size_t)f * 3 * V;
    int fg = 0;
    if (o < npix) {
        const int i = (int)(o / width), j = (int)(o % width);          // row, column
        float depth = 0.f;
        const unsigned long long kd = dkey[(size_t)fl * npix + o];
        if (kd != 0ull) {
            const int face = (int)(kd & 0xFFFFFFFFull);
            if (!fedge[(size_t)fl * F + face]) {
                float px[3], py[3]; int vid[3], s[3];
                face_projection(dm, cl, face, fx, fy, cx, cy, px, py, vid);
                sort3(py, s);
                const float ax = px[s[0]], ay = floorf(py[s[0]]), bx = px[s[1]], by = py[s[1]], cxx = px[s[2]], cyy = ceilf(py[s[2]]);
                const float az = (float)cl[3 * (size_t)vid[s[0]] + 2], bz = (float)cl[3 * (size_t)vid[s[1]] + 2], cz = (float)cl[3 * (size_t)vid[s[2]] + 2];
                const float denom = 1.0f / ((bx - cxx) * (ay - cyy) + (cyy - by) * (ax - cxx));
                const float w1v = (bx - cxx) * ((float)i - cyy), w2v = (cxx - ax) * ((float)i - cyy);
                const float w1 = (w1v + (cyy - by) * ((float)j - cxx)) * denom, w2 = (w2v + (ay - cyy) * ((float)j - cxx)) * denom;
                const float v = w1 * az + w2 * bz + (1.f - w1 - w2) * cz;
                const float lo = (v < 0.0f) ? 0.0f : v;                // std::max(v, 0.0f): NaN stays NaN
                depth = (255.0f < lo) ? 255.0f : lo;                   // std::min(., maxz = 255)
            }
        }
        unsigned char lab = 255;
        const unsigned long long km = mkey[(size_t)fl * npix + o];
        if (km != 0ull) {
            const int face = (int)(km & 0xFFFFFFFFull);
            if (!fedge[(size_t)fl * F + face]) {
                float px[3], py[3]; int vid[3], s[3];
                face_projection(dm, cl, face, fx, fy, cx, cy, px, py, vid);
                sort3(px, s);
                const float ax = floorf(px[s[0]]), ay = py[s[0]], bx = px[s[1]], by = py[s[1]], cxx = ceilf(px[s[2]]), cyy = py[s[2]];
                const int dista = cvt_x86((ax - (float)j) * (ax - (float)j) + (ay - (float)i) * (ay - (float)i));
                const int distb = cvt_x86((bx - (float)j) * (bx - (float)j) + (by - (float)i) * (by - (float)i));
                const int distc = cvt_x86((cxx - (float)j) * (cxx - (float)j) + (cyy - (float)i) * (cyy - (float)i));
                lab = (unsigned char)((dista < distb && dista < distc) ? vertex_part[vid[s[0]]] : (distb < distc ? vertex_part[vid[s[1]]] : vertex_part[vid[s[2]]]));
            }
        }
        depth_img[(size_t)fl * npix + o] = depth;
        label[(size_t)fl * npix + o] = lab;
        fg = !(depth <= 0.0f);                                         // optim.cpp:110: `if (ptr[c] <= 0.0) continue;`
        dkey[(size_t)fl * npix + o] = fg ? ((unsigned long long)__float_as_uint(depth) << 32) : 0xFFFFFFFFFFFFFFFFull;
    }
    const unsigned long long bal = __ballot(fg);
    __shared__ int s_c[4];
    if (lane_id() == 0) s_c[wave_id()] = __popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) block_count[(size_t)fl * gridDim.x + blockIdx.x] = s_c[0] + s_c[1] + s_c[2] + s_c[3];
}

__global__ __launch_bounds__(256) void k_paint_clear(unsigned long long* a, unsigned long long* b, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { a[i] = 0ull; b[i] = 0ull; }
}

int avt_paint_enqueue(avt_ctx* c, int nframes, const int* d_vertex_part, unsigned long long* d_dkey, unsigned long long* d_mkey, float* d_fkey,
                      int* d_frank, unsigned char* d_fedge, float* d_depth, unsigned char* d_label, int* d_block, double fx, double fy, double cx,
                      double cy, int width, int height) {
    const size_t npix = (size_t)width * height;
    const int nb = (int)((npix + 255) / 256), F = c->dm.d.F, fb = (F + 255) / 256;
    hipStream_t s = c->cur_stream;
    // the reference's intrinsics are floats (Calibration.h:13); the projection promotes them to double
    fx = (double)(float)fx; fy = (double)(float)fy; cx = (double)(float)cx; cy = (double)(float)cy;
    hipLaunchKernelGGL(k_paint_clear, dim3((unsigned)((npix * nframes + 255) / 256)), dim3(256), 0, s, d_dkey, d_mkey, npix * nframes);
    hipLaunchKernelGGL(k_paint_keys, dim3(fb, nframes), dim3(256), 0, s, c->dm, c->fb, d_fkey, d_fedge);
    hipLaunchKernelGGL(k_paint_rank, dim3(fb, nframes), dim3(256), 0, s, F, d_fkey, d_frank);
    hipLaunchKernelGGL(k_paint_cover, dim3(fb, nframes), dim3(256), 0, s, c->dm, c->fb, d_frank, d_fedge, d_dkey, d_mkey, fx, fy, cx, cy, width, height);
    hipLaunchKernelGGL(k_paint_resolve, dim3(nb, nframes), dim3(256), 0, s, c->dm, c->fb, d_dkey, d_mkey, d_fedge, d_vertex_part, d_depth, d_label,
                       d_block, fx, fy, cx, cy, width, height);
    hipLaunchKernelGGL(k_raster_scan, dim3(nframes), dim3(1024), 0, s, c->fb, d_block, nb);
    hipLaunchKernelGGL(k_raster_emit, dim3(nb, nframes), dim3(256), 0, s, c->fb, d_dkey, d_label, d_block, (float)fx, (float)fy, (float)cx,
                       (float)cy, width, height);
    return hipGetLastError() != hipSuccess;
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
