// avt_device.h — small device helpers shared by the kernels (gfx950, wave64).
#pragma once
#include "avt_internal.h"

#define AVT_INF (__builtin_inf())

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// inclusive wave (64-lane) prefix sum of ints via cross-lane shuffles
__device__ __forceinline__ int wave_incl_scan(int v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(v, d, 64);
        if (lane_id() >= d) v += t;
    }
    return v;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// Eigen-style quaternion (x,y,z,w) -> row-major 3x3 (Quaternion::toRotationMatrix closed form)
__device__ __forceinline__ void quat_to_rot(const double* q, double* R) {
    const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

// Forward kinematics shared by Avatar::update (Avatar.cpp:41-64) and PrepareForEvaluation
// (AvatarOptimizer.cpp:303-315): world rotation Rw[j] = Rw[parent]*rot[j], origin o[j] = o[parent] +
// Rw[parent]*(jp[j]-jp[parent]), root at p.  All arrays live in LDS; must be called by every thread of the
// block (contains barriers).  rot[J][9], jp[J][3] are inputs; lvl[J+1] is LDS scratch.  Joints are processed
// one tree level per barrier (SMPL: 9 levels instead of 24 sequential joints).
__device__ __forceinline__ void fk_chain(int J, const int* __restrict__ parent, const double* rot, const double* jp,
                                         const double* p, double* Rw, double* o, int* lvl) {
    const int t = threadIdx.x;
    if (t == 0) {
        int mx = 0;
        lvl[0] = 0;
        for (int j = 1; j < J; ++j) { lvl[j] = lvl[parent[j]] + 1; mx = max(mx, lvl[j]); }
        lvl[J] = mx;
    }
    __syncthreads();
    const int nl = lvl[J];
    for (int L = 0; L <= nl; ++L) {
        for (int idx = t; idx < 12 * J; idx += blockDim.x) {
            const int j = idx / 12, e = idx % 12;
            if (lvl[j] != L) continue;
            if (j == 0) {
                if (e < 9) Rw[e] = rot[e];
                else o[e - 9] = p[e - 9];
            } else {
                const int pa = parent[j];
                const double* Rp = Rw + 9 * pa;
                if (e < 9) {
                    const int r = e / 3, c = e % 3;
                    Rw[9 * j + e] = Rp[3 * r] * rot[9 * j + c] + Rp[3 * r + 1] * rot[9 * j + 3 + c] + Rp[3 * r + 2] * rot[9 * j + 6 + c];
                } else {
                    const int r = e - 9;
                    const double d0 = jp[3 * j] - jp[3 * pa], d1 = jp[3 * j + 1] - jp[3 * pa + 1], d2 = jp[3 * j + 2] - jp[3 * pa + 2];
                    o[3 * j + r] = o[3 * pa + r] + (Rp[3 * r] * d0 + Rp[3 * r + 1] * d1 + Rp[3 * r + 2] * d2);
                }
            }
        }
        __syncthreads();
    }
}
