// ark/Types.h — minimal column-major containers standing in for the Eigen / OpenCV types of the reference's
// public signatures (Eigen and OpenCV are not dependencies of this engine).  Layouts are identical to the Eigen
// types they replace, so `.data()` can be handed to code written against the reference:
//   CloudType  <-> Eigen::Matrix<double, 3, Dynamic>  (Avatar.h:19)      column i at data() + 3*i
//   VectorXd   <-> Eigen::VectorXd,  VectorXi <-> Eigen::VectorXi
//   Matrix3d   <-> Eigen::Matrix3d (column-major),  Quaterniond <-> Eigen::Quaterniond coeffs (x,y,z,w)
#pragma once
#include <array>
#include <cmath>
#include <cstddef>
#include <limits>
#include <random>
#include <vector>

namespace ark {

using VectorXd = std::vector<double>;
using VectorXi = std::vector<int>;

struct Vector3d {
    double v[3] = {0, 0, 0};
    double& operator()(int i) { return v[i]; }
    double operator()(int i) const { return v[i]; }
    double* data() { return v; }
    const double* data() const { return v; }
};

struct Matrix3d {  // column-major
    double m[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    double& operator()(int r, int c) { return m[3 * c + r]; }
    double operator()(int r, int c) const { return m[3 * c + r]; }
    void setIdentity() { *this = Matrix3d(); }
    double* data() { return m; }
    const double* data() const { return m; }
    static Matrix3d AngleAxis(double angle, double ax, double ay, double az) {  // Rodrigues, unit axis
        Matrix3d R;
        const double c = std::cos(angle), s = std::sin(angle), t = 1 - c;
        R(0, 0) = c + ax * ax * t;      R(0, 1) = ax * ay * t - az * s; R(0, 2) = ax * az * t + ay * s;
        R(1, 0) = ay * ax * t + az * s; R(1, 1) = c + ay * ay * t;      R(1, 2) = ay * az * t - ax * s;
        R(2, 0) = az * ax * t - ay * s; R(2, 1) = az * ay * t + ax * s; R(2, 2) = c + az * az * t;
        return R;
    }
};

inline Matrix3d operator*(const Matrix3d& a, const Matrix3d& b) {
    Matrix3d o;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) o(r, c) = a(r, 0) * b(0, c) + a(r, 1) * b(1, c) + a(r, 2) * b(2, c);
    return o;
}
inline Matrix3d transpose(const Matrix3d& a) {
    Matrix3d o;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) o(r, c) = a(c, r);
    return o;
}
inline Vector3d operator-(const Vector3d& a, const Vector3d& b) { Vector3d o; for (int i = 0; i < 3; ++i) o(i) = a(i) - b(i); return o; }
inline double norm(const Vector3d& a) { return std::sqrt(a(0) * a(0) + a(1) * a(1) + a(2) * a(2)); }

struct Quaterniond {  // coefficient order (x, y, z, w) as Eigen stores it
    double c[4] = {0, 0, 0, 1};
    double* coeffsData() { return c; }
    const double* coeffsData() const { return c; }
    double x() const { return c[0]; }
    double y() const { return c[1]; }
    double z() const { return c[2]; }
    double w() const { return c[3]; }
};

template <int ROWS>
struct MatrixNX {  // ROWS x N, column-major
    std::vector<double> a;
    size_t cols() const { return a.size() / ROWS; }
    size_t rows() const { return ROWS; }
    size_t size() const { return a.size(); }
    void resize(size_t /*rows*/, size_t n) { a.assign(ROWS * n, 0.0); }
    double& operator()(int r, size_t c) { return a[ROWS * c + r]; }
    double operator()(int r, size_t c) const { return a[ROWS * c + r]; }
    double* col(size_t c) { return a.data() + ROWS * c; }
    const double* col(size_t c) const { return a.data() + ROWS * c; }
    double* data() { return a.data(); }
    const double* data() const { return a.data(); }
};
using CloudType = MatrixNX<3>;

struct MeshType {  // 3 x F int, column-major
    std::vector<int> a;
    size_t cols() const { return a.size() / 3; }
    int operator()(int r, size_t c) const { return a[3 * c + r]; }
    const int* data() const { return a.data(); }
};

struct Size {  // cv::Size stand-in
    int width = 0, height = 0;
    Size() {}
    Size(int w, int h) : width(w), height(h) {}
};

struct CameraIntrin {  // Calibration.h:11-77: API-surface type only, does not influence optimize() (AvatarOptimizer.cpp:1271)
    float fx = 606.438f, fy = 606.351f, cx = 637.294f, cy = 366.992f;
    float k[6] = {0, 0, 0, 0, 0, 0}, p[2] = {0, 0};
};

// Eigen 3.3 closed forms used by optimize() to move between rotation matrices and quaternions
// (AvatarOptimizer.cpp:1250-1254: Matrix3 -> Quaternion -> AngleAxis -> Quaternion; :1494-1496 back).
/** Eigen's AngleAxis::fromRotationMatrix (through the quaternion of the matrix): angle in [0, pi], unit axis ((1,0,0) for the identity). */
inline void rotationToAngleAxis(const Matrix3d& m, double& angle, Vector3d& axis) {
    double q[4];
    double t = m(0, 0) + m(1, 1) + m(2, 2);
    if (t > 0.0) {
        t = std::sqrt(t + 1.0); q[3] = 0.5 * t; t = 0.5 / t;
        q[0] = (m(2, 1) - m(1, 2)) * t; q[1] = (m(0, 2) - m(2, 0)) * t; q[2] = (m(1, 0) - m(0, 1)) * t;
    } else {
        int i = 0;
        if (m(1, 1) > m(0, 0)) i = 1;
        if (m(2, 2) > m(i, i)) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(m(i, i) - m(j, j) - m(k, k) + 1.0); q[i] = 0.5 * t; t = 0.5 / t;
        q[3] = (m(k, j) - m(j, k)) * t; q[j] = (m(j, i) + m(i, j)) * t; q[k] = (m(k, i) + m(i, k)) * t;
    }
    double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    if (n < std::numeric_limits<double>::epsilon()) {
        const double mx = std::fmax(std::fabs(q[0]), std::fmax(std::fabs(q[1]), std::fabs(q[2])));
        if (mx > 0.0) { const double a = q[0] / mx, b = q[1] / mx, c = q[2] / mx; n = mx * std::sqrt(a * a + b * b + c * c); }
        else n = 0.0;
    }
    angle = 0.0; axis(0) = 1.0; axis(1) = 0.0; axis(2) = 0.0;
    if (n != 0.0) {
        angle = 2.0 * std::atan2(n, std::fabs(q[3]));
        if (q[3] < 0) n = -n;
        axis(0) = q[0] / n; axis(1) = q[1] / n; axis(2) = q[2] / n;
    }
}

/** Eigen's Quaternion::FromTwoVectors(a, b).toRotationMatrix(): the rotation that takes the direction of a to the direction of b about their
 *  common normal.  (Opposite directions: Eigen takes the axis from an SVD; any unit vector normal to a serves - here the one built from a's
 *  smallest component.) */
inline Matrix3d rotationFromTwoVectors(const Vector3d& a, const Vector3d& b) {
    const double na = norm(a), nb = norm(b);
    Vector3d v0, v1;
    for (int i = 0; i < 3; ++i) { v0(i) = a(i) / na; v1(i) = b(i) / nb; }
    const double c = v0(0) * v1(0) + v0(1) * v1(1) + v0(2) * v1(2);
    Quaterniond q;
    if (c < -1.0 + 1e-12) {
        int s = 0;
        if (std::fabs(v0(1)) < std::fabs(v0(s))) s = 1;
        if (std::fabs(v0(2)) < std::fabs(v0(s))) s = 2;
        Vector3d e; e(s) = 1.0;
        Vector3d ax;      // e x v0, normalised
        ax(0) = e(1) * v0(2) - e(2) * v0(1); ax(1) = e(2) * v0(0) - e(0) * v0(2); ax(2) = e(0) * v0(1) - e(1) * v0(0);
        const double n = norm(ax), w2 = (1.0 + c) * 0.5, sw = std::sqrt(std::fmax(0.0, 1.0 - w2));
        q.c[3] = std::sqrt(std::fmax(0.0, w2)); q.c[0] = ax(0) / n * sw; q.c[1] = ax(1) / n * sw; q.c[2] = ax(2) / n * sw;
    } else {
        const double s = std::sqrt((1.0 + c) * 2.0), invs = 1.0 / s;
        q.c[0] = (v0(1) * v1(2) - v0(2) * v1(1)) * invs; q.c[1] = (v0(2) * v1(0) - v0(0) * v1(2)) * invs; q.c[2] = (v0(0) * v1(1) - v0(1) * v1(0)) * invs;
        q.c[3] = s * 0.5;
    }
    // (declared below)
    const double tx = 2 * q.c[0], ty = 2 * q.c[1], tz = 2 * q.c[2];
    const double twx = tx * q.c[3], twy = ty * q.c[3], twz = tz * q.c[3], txx = tx * q.c[0], txy = ty * q.c[0], txz = tz * q.c[0];
    const double tyy = ty * q.c[1], tyz = tz * q.c[1], tzz = tz * q.c[2];
    Matrix3d R;
    R(0, 0) = 1 - (tyy + tzz); R(0, 1) = txy - twz;       R(0, 2) = txz + twy;
    R(1, 0) = txy + twz;       R(1, 1) = 1 - (txx + tzz); R(1, 2) = tyz - twx;
    R(2, 0) = txz - twy;       R(2, 1) = tyz + twx;       R(2, 2) = 1 - (txx + tyy);
    return R;
}

/** The reference's random_util (Util.h:264-277, Util.cpp:312-332): float distributions over std::mt19937; the two-argument forms draw from a
 *  generator of their own per thread, seeded from std::random_device - reseed() (not in the reference) makes them reproducible for tests. */
namespace random_util {
inline std::mt19937& threadGenerator(int which) { thread_local static std::mt19937 rg[2] = {std::mt19937(std::random_device{}()), std::mt19937(std::random_device{}())}; return rg[which]; }
inline float uniform(std::mt19937& rg, float min_inc = 0.f, float max_exc = 1.f) { std::uniform_real_distribution<float> d(min_inc, max_exc); return d(rg); }
inline float randn(std::mt19937& rg, float mean = 0.f, float variance = 1.f) { std::normal_distribution<float> d(mean, variance); return d(rg); }      // (the reference passes its "variance" as the distribution's sigma)
inline float uniform(float min_inc = 0.f, float max_exc = 1.f) { return uniform(threadGenerator(0), min_inc, max_exc); }
inline float randn(float mean = 0.f, float variance = 1.f) { return randn(threadGenerator(1), mean, variance); }
inline void reseed(unsigned seed) { threadGenerator(0).seed(seed); threadGenerator(1).seed(seed + 1u); }
}  // namespace random_util

inline Quaterniond rotationToQuaternion(const Matrix3d& m) {
    double q[4];
    double t = m(0, 0) + m(1, 1) + m(2, 2);
    if (t > 0.0) {
        t = std::sqrt(t + 1.0); q[3] = 0.5 * t; t = 0.5 / t;
        q[0] = (m(2, 1) - m(1, 2)) * t; q[1] = (m(0, 2) - m(2, 0)) * t; q[2] = (m(1, 0) - m(0, 1)) * t;
    } else {
        int i = 0;
        if (m(1, 1) > m(0, 0)) i = 1;
        if (m(2, 2) > m(i, i)) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(m(i, i) - m(j, j) - m(k, k) + 1.0); q[i] = 0.5 * t; t = 0.5 / t;
        q[3] = (m(k, j) - m(j, k)) * t; q[j] = (m(j, i) + m(i, j)) * t; q[k] = (m(k, i) + m(i, k)) * t;
    }
    double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    if (n < std::numeric_limits<double>::epsilon()) {
        const double mx = std::fmax(std::fabs(q[0]), std::fmax(std::fabs(q[1]), std::fabs(q[2])));
        if (mx > 0.0) { const double a = q[0] / mx, b = q[1] / mx, c = q[2] / mx; n = mx * std::sqrt(a * a + b * b + c * c); }
        else n = 0.0;
    }
    double angle = 0.0, ax[3] = {1, 0, 0};
    if (n != 0.0) {
        angle = 2.0 * std::atan2(n, std::fabs(q[3]));
        if (q[3] < 0) n = -n;
        ax[0] = q[0] / n; ax[1] = q[1] / n; ax[2] = q[2] / n;
    }
    Quaterniond out;
    const double ha = 0.5 * angle, s = std::sin(ha);
    out.c[3] = std::cos(ha); out.c[0] = s * ax[0]; out.c[1] = s * ax[1]; out.c[2] = s * ax[2];
    return out;
}

inline Matrix3d quaternionToRotation(const Quaterniond& qq) {
    const double* q = qq.c;
    const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    Matrix3d R;
    R(0, 0) = 1 - (tyy + tzz); R(0, 1) = txy - twz;       R(0, 2) = txz + twy;
    R(1, 0) = txy + twz;       R(1, 1) = 1 - (txx + tzz); R(1, 2) = tyz - twx;
    R(2, 0) = txz - twy;       R(2, 1) = tyz + twx;       R(2, 2) = 1 - (txx + tyy);
    return R;
}

}  // namespace ark
