// render_oracle.cpp — CPU restatement of the reference's synthetic-frame renderer.
//
// TEST INFRASTRUCTURE ONLY (see avatar_oracle.cpp header): nothing under avatar_amd/ links or calls this file.
// What it restates, in its own scalar types (no OpenCV / Eigen), citing the reference lines it follows:
//   * vertex projection to float pixel coordinates                          AvatarRenderer.cpp:11-24
//   * faces sorted by DECREASING mean depth (painter's order, std::sort)    AvatarRenderer.cpp:39-69
//   * renderDepth: edge-on faces (|n_z| < 0.1) paint 0, the others a scanline fill with screen-space barycentric depth
//     clamped to [0, 255]                                                   AvatarRenderer.cpp:72-101, AvatarHelpers.cpp:61-139
//   * renderPartMask: edge-on faces paint 255, the others a COLUMN-major scanline fill labelled by the nearest of the
//     three projected vertices, distances truncated to int                  AvatarRenderer.cpp:174-202, AvatarHelpers.cpp:153-245
//   * paintTriangleSingleColor fills [minx, maxx) - the last pixel of a row is NOT painted (std::fill end exclusive)
//                                                                           AvatarHelpers.cpp:247-303
//   * back-projection of every pixel with depth > 0 (float arithmetic), y negated
//                                                                           Calibration.cpp:68-74, optim.cpp:104-120
// Quirks kept on purpose: the first vertex' coordinate is floored and the last one's ceiled BEFORE the edge slopes and the
// nearest-vertex distances are computed; the second half of the part fill always starts one column after the middle
// vertex; later faces simply overwrite earlier ones (no depth test).
// PARITY STATUS: the reference cannot be built here (OpenCV) and holds no rendered fixtures, so this file is pinned only
// by hand-computed small cases (tests/test_render_oracle_cpu.py).  The product's generator has two modes: AVT_RENDER_PAINTER must
// reproduce this file's images bit for bit (tests/test_gpu_render.py), the z-buffer mode is a different visibility algorithm by
// design and is compared statistically.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <utility>
#include <vector>

namespace {

struct P2 { float x, y; };

void project_points(int V, const double* cloud, float fx, float fy, float cx, float cy, std::vector<P2>& out) {
    out.resize(V);
    for (int i = 0; i < V; ++i) {
        const double* p = cloud + 3 * (size_t)i;
        out[i].x = (float)(p[0] * fx / p[2] + cx);          // double arithmetic, stored as float (cv::Point2f)
        out[i].y = (float)(-p[1] * fy / p[2] + cy);
    }
}

// (mean depth, face) sorted by decreasing depth with std::sort, like the reference
void ordered_faces(int F, const int* mesh, const double* cloud, std::vector<std::pair<float, int>>& out, bool stable = false) {
    out.resize(F);
    for (int f = 0; f < F; ++f) {
        const int* m = mesh + 3 * (size_t)f;
        out[f].first = (float)((cloud[3 * (size_t)m[0] + 2] + cloud[3 * (size_t)m[1] + 2] + cloud[3 * (size_t)m[2] + 2]) / 3.f);
        out[f].second = f;
    }
    auto comp = [](const std::pair<float, int>& a, const std::pair<float, int>& b) { return a.first > b.first; };
    // the reference calls std::sort, which leaves the order of equal keys unspecified; `stable` orders them by face id (what
    // the HIP generator does) so that a test can assert a fixture does not depend on it
    if (stable) std::stable_sort(out.begin(), out.end(), comp); else std::sort(out.begin(), out.end(), comp);
}

bool edge_on(const double* a, const double* b, const double* c) {
    const double ab[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, ac[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
    const double n[3] = {ab[1] * ac[2] - ab[2] * ac[1], ab[2] * ac[0] - ab[0] * ac[2], ab[0] * ac[1] - ab[1] * ac[0]};
    // Eigen 3.3 MatrixBase::normalized(): z = squaredNorm(); z > 0 ? n / sqrt(z) : n  (a zero vector is returned unchanged, so a
    // degenerate face counts as edge-on)
    const double z = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
    const double nz = z > 0.0 ? n[2] / std::sqrt(z) : n[2];
    return std::fabs(nz) < 0.1;
}

// AvatarHelpers.cpp:61-139
void paint_bary(std::vector<float>& img, int W, int H, const std::vector<P2>& pr, const int* face, const float* zvec, float maxz) {
    std::pair<double, int> yf[3] = {{pr[face[0]].y, 0}, {pr[face[1]].y, 1}, {pr[face[2]].y, 2}};
    std::sort(yf, yf + 3);
    P2 a = pr[face[yf[0].second]], b = pr[face[yf[1].second]], c = pr[face[yf[2].second]];
    a.y = std::floor(a.y);
    c.y = std::ceil(c.y);
    if (a.y == c.y) return;
    const int minyi = std::max<int>((int)a.y, 0), maxyi = std::min<int>((int)c.y, H - 1), midyi = (int)std::floor(b.y);
    const float az = zvec[yf[0].second], bz = zvec[yf[1].second], cz = zvec[yf[2].second];
    const float denom = 1.0f / ((b.x - c.x) * (a.y - c.y) + (c.y - b.y) * (a.x - c.x));
    auto row = [&](int i, float mlo, float blo, float mhi, float bhi) {
        const int minxi = std::max<int>((int)std::floor(mlo * i + blo), 0), maxxi = std::min<int>((int)std::ceil(mhi * i + bhi), W - 1);
        if (minxi > maxxi) return;
        const float w1v = (b.x - c.x) * (i - c.y), w2v = (c.x - a.x) * (i - c.y);
        float* ptr = img.data() + (size_t)i * W;
        for (int j = minxi; j <= maxxi; ++j) {
            const float w1 = (w1v + (c.y - b.y) * (j - c.x)) * denom, w2 = (w2v + (a.y - c.y) * (j - c.x)) * denom;
            ptr[j] = std::min(std::max(w1 * az + w2 * bz + (1.f - w1 - w2) * cz, 0.0f), maxz);
        }
    };
    if (a.y != b.y) {
        float mhi = (c.x - a.x) / (c.y - a.y), bhi = a.x - a.y * mhi, mlo = (b.x - a.x) / (b.y - a.y), blo = a.x - a.y * mlo;
        if (b.x > c.x) { std::swap(mlo, mhi); std::swap(blo, bhi); }
        for (int i = minyi; i <= std::min(midyi, H - 1); ++i) row(i, mlo, blo, mhi, bhi);
    }
    if (b.y != c.y) {
        float mhi = (c.x - a.x) / (c.y - a.y), bhi = a.x - a.y * mhi, mlo = (c.x - b.x) / (c.y - b.y), blo = b.x - b.y * mlo;
        if (b.x > a.x) { std::swap(mlo, mhi); std::swap(blo, bhi); }
        for (int i = std::max(midyi, 0) + (a.y != b.y); i <= maxyi; ++i) row(i, mlo, blo, mhi, bhi);
    }
}

// AvatarHelpers.cpp:247-303 (row-major scan, end-exclusive fill)
template <class T>
void paint_single(std::vector<T>& img, int W, int H, const std::vector<P2>& pr, const int* face, T color) {
    std::pair<double, int> yf[3] = {{pr[face[0]].y, 0}, {pr[face[1]].y, 1}, {pr[face[2]].y, 2}};
    std::sort(yf, yf + 3);
    P2 a = pr[face[yf[0].second]], b = pr[face[yf[1].second]], c = pr[face[yf[2].second]];
    a.y = std::floor(a.y);
    c.y = std::ceil(c.y);
    if (a.y == c.y) return;
    const int minyi = std::max<int>((int)a.y, 0), maxyi = std::min<int>((int)c.y, H - 1), midyi = (int)std::floor(b.y);
    auto row = [&](int i, double mlo, double blo, double mhi, double bhi) {
        const int minxi = std::max<int>((int)std::floor(mlo * i + blo), 0), maxxi = std::min<int>((int)std::ceil(mhi * i + bhi), W - 1);
        if (minxi > maxxi) return;
        T* ptr = img.data() + (size_t)i * W;
        std::fill(ptr + minxi, ptr + maxxi, color);
    };
    if (a.y != b.y) {
        double mhi = (c.x - a.x) / (c.y - a.y), bhi = a.x - a.y * mhi, mlo = (b.x - a.x) / (b.y - a.y), blo = a.x - a.y * mlo;
        if (b.x > c.x) { std::swap(mlo, mhi); std::swap(blo, bhi); }
        for (int i = minyi; i <= std::min(midyi, H - 1); ++i) row(i, mlo, blo, mhi, bhi);
    }
    if (b.y != c.y) {
        double mhi = (c.x - a.x) / (c.y - a.y), bhi = a.x - a.y * mhi, mlo = (c.x - b.x) / (c.y - b.y), blo = b.x - b.y * mlo;
        if (b.x > a.x) { std::swap(mlo, mhi); std::swap(blo, bhi); }
        for (int i = std::max(midyi, 0) + 1; i <= maxyi; ++i) row(i, mlo, blo, mhi, bhi);
    }
}

// AvatarHelpers.cpp:153-245 (column-major scan; int-truncated squared distances to the floored / ceiled vertices)
void paint_parts(std::vector<std::uint8_t>& img, int W, int H, const std::vector<P2>& pr, const int* face, const int* vertex_part) {
    std::pair<double, int> xf[3] = {{pr[face[0]].x, 0}, {pr[face[1]].x, 1}, {pr[face[2]].x, 2}};
    std::sort(xf, xf + 3);
    P2 a = pr[face[xf[0].second]], b = pr[face[xf[1].second]], c = pr[face[xf[2].second]];
    a.x = std::floor(a.x);
    c.x = std::ceil(c.x);
    if (a.x == c.x) return;
    const int pa = vertex_part[face[xf[0].second]], pb = vertex_part[face[xf[1].second]], pc = vertex_part[face[xf[2].second]];
    const int minxi = std::max<int>((int)a.x, 0), maxxi = std::min<int>((int)c.x, W - 1), midxi = (int)std::floor(b.x);
    auto col = [&](int i, double mlo, double blo, double mhi, double bhi) {
        const int minyi = std::max<int>((int)std::floor(mlo * i + blo), 0), maxyi = std::min<int>((int)std::ceil(mhi * i + bhi), H - 1);
        if (minyi > maxyi) return;
        for (int j = minyi; j <= maxyi; ++j) {
            const int dista = (int)((a.x - i) * (a.x - i) + (a.y - j) * (a.y - j));
            const int distb = (int)((b.x - i) * (b.x - i) + (b.y - j) * (b.y - j));
            const int distc = (int)((c.x - i) * (c.x - i) + (c.y - j) * (c.y - j));
            img[(size_t)j * W + i] = (std::uint8_t)((dista < distb && dista < distc) ? pa : (distb < distc ? pb : pc));
        }
    };
    if (a.x != b.x) {
        double mhi = (c.y - a.y) / (c.x - a.x), bhi = a.y - a.x * mhi, mlo = (b.y - a.y) / (b.x - a.x), blo = a.y - a.x * mlo;
        if (b.y > c.y) { std::swap(mlo, mhi); std::swap(blo, bhi); }
        for (int i = minxi; i <= std::min(midxi, W - 1); ++i) col(i, mlo, blo, mhi, bhi);
    }
    if (b.x != c.x) {
        double mhi = (c.y - a.y) / (c.x - a.x), bhi = a.y - a.x * mhi, mlo = (c.y - b.y) / (c.x - b.x), blo = b.y - b.x * mlo;
        if (b.y > a.y) { std::swap(mlo, mhi); std::swap(blo, bhi); }
        for (int i = std::max(midxi, 0) + 1; i <= maxxi; ++i) col(i, mlo, blo, mhi, bhi);
    }
}

}  // namespace

extern "C" {

// AvatarRenderer::renderDepth (float32 H x W, 0 = background) and renderPartMask (uint8 H x W, 255 = background) of one
// posed avatar.  vertex_part[v] = part_map[assignedJoints[v][0].second].  Either output may be NULL.
int orc_render_ex(int V, int F, const double* cloud, const int* mesh, const int* vertex_part, float fx, float fy, float cx, float cy,
                  int W, int H, float* depth_out, std::uint8_t* mask_out, int stable_sort, int* tied_keys_out) {
    std::vector<P2> pr;
    project_points(V, cloud, fx, fy, cx, cy, pr);
    std::vector<std::pair<float, int>> faces;
    ordered_faces(F, mesh, cloud, faces, stable_sort != 0);
    if (tied_keys_out) {
        int t = 0;
        for (int k = 1; k < F; ++k) t += faces[k].first == faces[k - 1].first;
        *tied_keys_out = t;
    }
    std::vector<float> depth;
    std::vector<std::uint8_t> mask;
    if (depth_out) depth.assign((size_t)W * H, 0.f);
    if (mask_out) mask.assign((size_t)W * H, 255);
    for (int k = 0; k < F; ++k) {
        const int* fc = mesh + 3 * (size_t)faces[k].second;
        const double* a = cloud + 3 * (size_t)fc[0]; const double* b = cloud + 3 * (size_t)fc[1]; const double* c = cloud + 3 * (size_t)fc[2];
        const bool eo = edge_on(a, b, c);
        if (depth_out) {
            if (eo) paint_single<float>(depth, W, H, pr, fc, 0.f);
            else { const float zv[3] = {(float)a[2], (float)b[2], (float)c[2]}; paint_bary(depth, W, H, pr, fc, zv, 255.0f); }
        }
        if (mask_out) {
            if (eo) paint_single<std::uint8_t>(mask, W, H, pr, fc, (std::uint8_t)255);
            else paint_parts(mask, W, H, pr, fc, vertex_part);
        }
    }
    if (depth_out) std::copy(depth.begin(), depth.end(), depth_out);
    if (mask_out) std::copy(mask.begin(), mask.end(), mask_out);
    return 0;
}

int orc_render(int V, int F, const double* cloud, const int* mesh, const int* vertex_part, float fx, float fy, float cx, float cy,
               int W, int H, float* depth_out, std::uint8_t* mask_out) {
    return orc_render_ex(V, F, cloud, mesh, vertex_part, fx, fy, cx, cy, W, H, depth_out, mask_out, 0, nullptr);
}

// optim.cpp:104-120: every pixel with depth > 0 back-projected by CameraIntrin::to3D (float), y negated; labels from the
// part mask at the same pixel (255 where the two renders disagree about coverage).  Returns the number of points.
int orc_backproject(int W, int H, const float* depth, const std::uint8_t* mask, float fx, float fy, float cx, float cy, int capacity,
                    double* xyz_out, int* labels_out) {
    int n = 0;
    for (int r = 0; r < H; ++r)
        for (int c = 0; c < W; ++c) {
            const float d = depth[(size_t)r * W + c];
            if (d <= 0.0) continue;
            if (n < capacity) {
                xyz_out[3 * (size_t)n] = (double)(((float)c - cx) * d / fx);
                xyz_out[3 * (size_t)n + 1] = -(double)(((float)r - cy) * d / fy);
                xyz_out[3 * (size_t)n + 2] = (double)d;
                if (labels_out) labels_out[n] = mask ? (int)mask[(size_t)r * W + c] : 0;
            }
            ++n;
        }
    return n;
}

}  // extern "C"
