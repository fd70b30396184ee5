// synth_render.cpp — host-side synthetic depth-frame generator (workload harness, not on the hot path).
//
// Produces the (data_cloud, data_part_labels) pair that AvatarOptimizer::optimize() consumes, the way the
// reference's synthetic tools do: render the posed avatar's depth and part mask (AvatarRenderer.cpp:72-101,
// :174-202), back-project every foreground pixel with the pinhole model (Calibration.cpp:68-74, float
// arithmetic) and negate y (optim.cpp:116-119, demo.cpp:245).
//
// Deliberate simplification (documented in DESIGN.md): the reference paints depth-sorted triangles
// back-to-front (painter's algorithm, AvatarHelpers.cpp:61-139); this generator uses a z-buffer with the
// same pixel sampling (integer pixel centres), the same screen-space linear depth interpolation, the same
// edge-on rejection (|n_z| < 0.1 of the unit normal) and the same nearest-projected-vertex part rule
// (AvatarHelpers.cpp:170-209).  For a non-self-intersecting surface both give the visible surface.
// Built with g++ into libavt_synth.so; no GPU, no oracle dependency.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

static void rasterise(int V, int F, const double* cloud, const int* mesh, const int* vertex_part, double fx, double fy, double cx,
                      double cy, int width, int height, std::vector<float>& zbuf, std::vector<int>& lab);

// the same render as avt_synth_render_cloud, returned as images the way the reference's perception front-end hands
// them to the tracker loop (demo.cpp:215-250): xyz map (H x W x 3 float32, camera coordinates, y NOT yet negated)
// and body-part mask (H x W uint8, 255 = background).  Returns the number of foreground pixels.
extern "C" int avt_synth_render_images(int V, int F, const double* cloud, const int* mesh, const int* vertex_part, double fx,
                                       double fy, double cx, double cy, int width, int height, float* xyz_out,
                                       unsigned char* mask_out) {
    std::vector<float> zbuf;
    std::vector<int> lab;
    rasterise(V, F, cloud, mesh, vertex_part, fx, fy, cx, cy, width, height, zbuf, lab);
    int count = 0;
    const float ffx = (float)fx, ffy = (float)fy, fcx = (float)cx, fcy = (float)cy;
    for (int r = 0; r < height; ++r)
        for (int col = 0; col < width; ++col) {
            const size_t o = (size_t)r * width + col;
            float* q = xyz_out + 3 * o;
            if (lab[o] < 0) { q[0] = q[1] = q[2] = 0.f; mask_out[o] = 255; continue; }
            const float depth = zbuf[o];
            q[0] = ((float)col - fcx) * depth / ffx;
            q[1] = ((float)r - fcy) * depth / ffy;
            q[2] = depth;
            mask_out[o] = (unsigned char)lab[o];
            ++count;
        }
    return count;
}

static void rasterise(int V, int F, const double* cloud, const int* mesh, const int* vertex_part, double fx, double fy, double cx,
                      double cy, int width, int height, std::vector<float>& zbuf, std::vector<int>& lab) {
    std::vector<float> px(V), py(V);
    for (int i = 0; i < V; ++i) {  // AvatarRenderer.cpp:11-24 (y flipped on projection)
        const double* p = cloud + 3 * i;
        px[i] = (float)(p[0] * fx / p[2] + cx);
        py[i] = (float)(-p[1] * fy / p[2] + cy);
    }
    zbuf.assign((size_t)width * height, std::numeric_limits<float>::infinity());
    lab.assign((size_t)width * height, -1);
    for (int f = 0; f < F; ++f) {
        const int ia = mesh[3 * f], ib = mesh[3 * f + 1], ic = mesh[3 * f + 2];
        const double* a = cloud + 3 * ia; const double* b = cloud + 3 * ib; const double* c = cloud + 3 * ic;
        const double ab[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, ac[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
        const double n[3] = {ab[1] * ac[2] - ab[2] * ac[1], ab[2] * ac[0] - ab[0] * ac[2], ab[0] * ac[1] - ab[1] * ac[0]};
        const double nn = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        if (!(nn > 0.0) || std::fabs(n[2] / nn) < 0.1) continue;  // edge-on faces carry no depth
        if (a[2] <= 0.0 || b[2] <= 0.0 || c[2] <= 0.0) continue;
        const float ax = px[ia], ay = py[ia], bx = px[ib], by = py[ib], cxx = px[ic], cyy = py[ic];
        const float denom = (by - cyy) * (ax - cxx) + (cxx - bx) * (ay - cyy);
        if (denom == 0.0f) continue;
        const float inv = 1.0f / denom;
        const int x0 = std::max(0, (int)std::floor(std::min(ax, std::min(bx, cxx))));
        const int x1 = std::min(width - 1, (int)std::ceil(std::max(ax, std::max(bx, cxx))));
        const int y0 = std::max(0, (int)std::floor(std::min(ay, std::min(by, cyy))));
        const int y1 = std::min(height - 1, (int)std::ceil(std::max(ay, std::max(by, cyy))));
        const float az = (float)a[2], bz = (float)b[2], cz = (float)c[2];
        for (int r = y0; r <= y1; ++r) {
            for (int col = x0; col <= x1; ++col) {
                const float w1 = ((by - cyy) * (col - cxx) + (cxx - bx) * (r - cyy)) * inv;
                const float w2 = ((cyy - ay) * (col - cxx) + (ax - cxx) * (r - cyy)) * inv;
                const float w3 = 1.0f - w1 - w2;
                if (w1 < 0.0f || w2 < 0.0f || w3 < 0.0f) continue;
                const float z = w1 * az + w2 * bz + w3 * cz;
                const size_t o = (size_t)r * width + col;
                if (z > 0.0f && z < zbuf[o]) {
                    zbuf[o] = z;
                    const float da = (ax - col) * (ax - col) + (ay - r) * (ay - r);
                    const float db = (bx - col) * (bx - col) + (by - r) * (by - r);
                    const float dc = (cxx - col) * (cxx - col) + (cyy - r) * (cyy - r);
                    lab[o] = (da < db && da < dc) ? vertex_part[ia] : (db < dc ? vertex_part[ib] : vertex_part[ic]);
                }
            }
        }
    }
}

extern "C" int avt_synth_render_cloud(int V, int F, const double* cloud /*3xV*/, const int* mesh /*3xF*/,
                                      const int* vertex_part /*V*/, double fx, double fy, double cx, double cy,
                                      int width, int height, int capacity, double* out_xyz /*3 x capacity*/,
                                      int* out_labels) {
    std::vector<float> zbuf;
    std::vector<int> lab;
    rasterise(V, F, cloud, mesh, vertex_part, fx, fy, cx, cy, width, height, zbuf, lab);
    int count = 0;
    const float ffx = (float)fx, ffy = (float)fy, fcx = (float)cx, fcy = (float)cy;
    for (int r = 0; r < height; ++r)
        for (int col = 0; col < width; ++col) {
            const size_t o = (size_t)r * width + col;
            if (lab[o] < 0) continue;
            if (count < capacity) {
                const float depth = zbuf[o];
                const float X = ((float)col - fcx) * depth / ffx;  // CameraIntrin::to3D (Calibration.cpp:68-74)
                const float Y = ((float)r - fcy) * depth / ffy;
                out_xyz[3 * (size_t)count] = (double)X;
                out_xyz[3 * (size_t)count + 1] = -(double)Y;        // y negated (optim.cpp:116-119)
                out_xyz[3 * (size_t)count + 2] = (double)depth;
                out_labels[count] = lab[o];
            }
            ++count;
        }
    return count;
}
