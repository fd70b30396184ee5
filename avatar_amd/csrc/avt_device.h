// avt_device.h — small device helpers shared by the kernels (gfx950, wave64).
#pragma once
#include <cstddef>
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
// block (contains barriers).  rot[J][9], jp[J][3] are inputs; lvl[J+1] holds the joints' tree depths and the deepest level
// (staged by the caller from DeviceModel::jlevel / AvtDims::nlevels).  Joints are processed one tree level per barrier
// (SMPL: 9 levels instead of 24 sequential joints).
__device__ __forceinline__ void fk_chain(int J, const int* __restrict__ parent, const double* rot, const double* jp,
                                         const double* p, double* Rw, double* o, int* lvl) {
    const int t = threadIdx.x;
    // lvl[0..J-1] = tree depth of every joint (host-computed), lvl[J] = deepest level: staged by the caller before its last barrier
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

// The objective of a point from its parts - 1/2 sum c|r|^2 + the constant of the data term, the pose prior's best component, the shape prior - formed
// by ONE sequence of individually rounded operations wherever an accept test needs it (k_solve, the folded test of reduce_spec_cost, the closing
// test in k_lbs: avt_decide.h).  The three places used to spell the same expression out separately and relied on the compiler contracting
// multiplies and adds the same way in three inlined contexts for their "bit for bit" agreement; nothing here can be contracted.
__device__ __forceinline__ double lm_objective_data(double sum_cr2, double cost_const) { return __dadd_rn(__dmul_rn(0.5, sum_cr2), cost_const); }
__device__ __forceinline__ double lm_objective_add_pose(double cost, double sbp, double best_score) { return __dadd_rn(cost, __dmul_rn(__dmul_rn(__dmul_rn(0.5, sbp), sbp), best_score)); }
__device__ __forceinline__ double lm_shape_term_add(double acc, double w_k, double sbs) { const double r = __dmul_rn(w_k, sbs); return __dadd_rn(acc, __dmul_rn(r, r)); }
__device__ __forceinline__ double lm_objective_add_shape(double cost, double shape_acc) { return __dadd_rn(cost, __dmul_rn(0.5, shape_acc)); }

// (cur_slot, try_valid) of a frame's control block as one 8-byte load.  The kernels of a Gauss-Newton iteration need the first for the address of
// everything they read next; the second says whether there is a trial point at all: AVT_TRY_DONE = the frame met the stopping rule
// (avt_options::function_tolerance, k_solve) and the launches left in this ICP iteration have nothing to do for it.
static_assert(offsetof(AvtFrameCtl, cur_slot) % 8 == 0 && offsetof(AvtFrameCtl, try_valid) == offsetof(AvtFrameCtl, cur_slot) + 4, "frame_slot_state reads both as one int2");
__device__ __forceinline__ int2 frame_slot_state(const FrameBuffers& fb, int f) { return *(const int2*)&fb.ctl[f].cur_slot; }

// Grids of shape (workgroups of a frame, frames) whose workgroups share per-frame data.  The dispatcher deals workgroups to the eight
// XCDs round-robin in linear order (block b -> XCD b % 8: observed, not promised), so a frame's workgroups land on all of them and the
// frame's data is fetched into eight L2s.  The remap gives every XCD a contiguous range of the (frame, block) space instead - whole
// frames, up to one split at each end - and is bijective for every grid size (q, r: the guide's variant for nwg % 8 != 0).
// A speed choice only: nothing depends on where a workgroup runs.  Returns the frame index RELATIVE to the launch (add fb.f0).
__device__ __forceinline__ void xcd_frame_block(const FrameBuffers& fb, int& bx, int& fy) {
    bx = blockIdx.x; fy = blockIdx.y;
    if (!fb.xcd_frames || gridDim.y < 8) return;
    const unsigned nb = gridDim.x, nwg = nb * gridDim.y, lin = blockIdx.x + nb * blockIdx.y;
    const unsigned q = nwg >> 3, r = nwg & 7, x = lin & 7;
    const unsigned vid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (lin >> 3);
    fy = (int)(vid / nb); bx = (int)(vid - (unsigned)fy * nb);
}

// the same for grids (frames, 1) - one workgroup per frame -: the frame this workgroup takes, relative to the launch, so that it runs on
// the XCD the frame's workgroups of the (blocks, frames) grids ran on (whole frames per XCD when the launch has a multiple of 8 frames)
__device__ __forceinline__ int xcd_frame_1d(const FrameBuffers& fb) {
    const unsigned F = gridDim.x, lin = blockIdx.x;
    if (!fb.xcd_frames || F < 8 || gridDim.y != 1) return (int)lin;
    const unsigned q = F >> 3, r = F & 7, x = lin & 7;
    return (int)((x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (lin >> 3));
}

// 0.5*sum_i |d_i - dbar_m(i)|^2: the part of the data cost that does not depend on the parameters once the
// correspondences are fixed.  Deterministic two-level reduction in original data order (block partials,
// summed in a fixed order by the solve kernel).
__device__ __forceinline__ void cost_const_block(const DeviceModel& dm, const FrameBuffers& fb, int f, int blk) {
    const int t = threadIdx.x, V = dm.d.V;
    const AvtFrameCtl& ctl = fb.ctl[f];
    const int N = ctl.N;
    const size_t base = (size_t)f * fb.max_points;
    const int i = blk * 256 + t;             // ORIGINAL data index: the reduction order is fixed
    double acc = 0.0;
    if (i < N) {
        const int m = fb.corr[base + i];
        if (m >= 0) {
            const int c = fb.cnt[(size_t)f * V + m];
            const long long* fs = fb.fsum + (size_t)f * 3 * V;
            const double mx = ctl.centre[0] + ((double)fs[m] / AVT_FIX_SCALE) / (double)c;
            const double my = ctl.centre[1] + ((double)fs[(size_t)V + m] / AVT_FIX_SCALE) / (double)c;
            const double mz = ctl.centre[2] + ((double)fs[2 * (size_t)V + m] / AVT_FIX_SCALE) / (double)c;
            const double* dp = fb.data_raw + 3 * (base + i);
            const double ex = dp[0] - mx, ey = dp[1] - my, ez = dp[2] - mz;
            acc = ex * ex + ey * ey + ez * ez;
        }
    }
    __shared__ double s_part[4];
    acc = wave_sum(acc);
    if (lane_id() == 0) s_part[wave_id()] = acc;
    __syncthreads();
    if (t == 0) fb.const_part[(size_t)f * fb.const_blocks + blk] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);   // every block of the grid writes
}

