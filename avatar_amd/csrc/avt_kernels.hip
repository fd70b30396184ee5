// avt_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the fitting path, part 1:
// linear-blend skinning, back-face visibility, data bucketing by body part, correspondence finalisation.
// (nearest neighbour: avt_nn.hip; residual/Jacobian/J^T J: avt_eval.hip; reduce + LM solve: avt_lm.hip)
#include "avt_device.h"
#include "avt_prep.h"
#include "avt_decide.h"
#include "avt_bucket.h"

__global__ __launch_bounds__(256) void k_bucket_count(DeviceModel dm, FrameBuffers fb) { bucket_count_block(dm, fb, blockIdx.y + fb.f0, blockIdx.x); }

__global__ __launch_bounds__(256) void k_bucket_scatter(DeviceModel dm, FrameBuffers fb) { bucket_scatter_block<true>(dm, fb, blockIdx.y + fb.f0, blockIdx.x); }

// =================================================================================================
// Avatar::update()  (Avatar.cpp:22-75)
// grid (ceil(V/256), nframes), block 256.  Joint matrices are staged in LDS (<= 32 joints x 24 doubles),
// the per-vertex loads are SoA and fully coalesced: 3(K+1) shape planes + 4 (weight, joint) pairs in,
// 3 doubles out => ~336 B/vertex algorithmic traffic.
// =================================================================================================
// Result record of a frame (include/avt_shard.h: the current state x = (p, q, w), AVT_SHARD_STAT_DOUBLES statistics): written by the calling workgroup
__device__ __forceinline__ void pack_result_row(const FrameBuffers& fb, int f, int xsize, int t, int nth) {
    const AvtFrameCtl& ctl = fb.ctl[f];
    const double* x = fb.x + ((size_t)f * 2 + ctl.cur_slot) * xsize;
    double* o = fb.results + (size_t)f * (xsize + 8);
    for (int e = t; e < xsize; e += nth) o[e] = x[e];
    if (t == 0) {
        double* s = o + xsize;
        s[0] = ctl.cost_initial; s[1] = ctl.cost_cur; s[2] = ctl.lambda; s[3] = (double)ctl.T; s[4] = (double)ctl.M;
        s[5] = (double)ctl.gn_iterations; s[6] = (double)ctl.accepted;
        s[7] = (double)fb.fault[f];      // sticky device fault bits of the frame (0 = its result is valid)
    }
}

__global__ __launch_bounds__(256) void k_lbs(DeviceModel dm, FrameBuffers fb, const double* __restrict__ w_in,
                                             const double* __restrict__ p_in, const double* __restrict__ R_in,
                                             int from_state, int vis_init, int nlbs, int with_init, int decide, int write_pc, int pack) {
    const AvtDims d = dm.d;
    const int J = d.J, K = d.K, V = d.V;
    int wx, fy;
    xcd_frame_block(fb, wx, fy);      // (frame batches: a frame's workgroups on one XCD, the one its later kernels run on)
    const int f = fy + fb.f0, t = threadIdx.x;
    // decide (from_state == 2 only): the accept test of the last trial point happens here, in every wave (avt_decide.h)
    // (its requests go out first; the skeleton tables of both slots follow in the same round trip)
    LastDecisionInputs dec_in;
    if (decide) dec_in = lm_last_load(dm, fb, f);
    // trailing workgroups of the grid: work that depends on nothing this kernel computes and would otherwise be a launch of its
    // own on the dependency chain - (few frames) the trial point of the ICP iteration that follows (prep_init_block), (first
    // launch of optimize()) the label histogram of the data points
    extern __shared__ __attribute__((aligned(16))) char lbs_dyn[];
    if (wx >= nlbs) {
        const int bx = wx - nlbs;
        if (with_init && bx == 0) prep_init_block(dm, fb, f, lbs_dyn, decide ? lm_last_decide(dm, fb, f, dec_in, false) : fb.ctl[f].cur_slot);
        else bucket_count_block(dm, fb, f, bx - (with_init ? 1 : 0));
        return;
    }
    __shared__ double s_rot[AVT_MAX_JOINTS * 9], s_Rw[AVT_MAX_JOINTS * 9], s_o[AVT_MAX_JOINTS * 3], s_jp[AVT_MAX_JOINTS * 3];
    __shared__ double s_T[AVT_MAX_JOINTS * 12];  // jointTrans, column-major 3x4 per joint (Avatar.h:215)
    __shared__ double s_w[AVT_MAX_SHAPE], s_p[3];
    __shared__ int s_parent[AVT_MAX_JOINTS], s_lvl[AVT_MAX_JOINTS + 1];

    if (from_state == 2) {
        // from the skeleton tables k_solve made for the current point (its world rotations, joint origins and rest joints are
        // what the forward kinematics below would recompute): no chain of tree levels on the way to the vertices
        if (decide) {      // both slots' tables were requested together with the inputs of the decision: one round trip
            const double* p0 = fb.prep + ((size_t)f * 2) * d.prep_size;
            const double* p1 = p0 + d.prep_size;
            constexpr int NR = (9 * AVT_MAX_JOINTS + 255) / 256;
            double ra[NR], rb[NR], oa = 0.0, ob = 0.0, ja = 0.0, jb = 0.0, ka = 0.0, kb = 0.0, wa = 0.0, wb = 0.0;
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int e = min(t + 256 * i, 9 * J - 1);
                ra[i] = p0[prep_off_Rw(d) + e]; rb[i] = p1[prep_off_Rw(d) + e];
            }
            {
                const int e = min(t, 3 * J - 1), k = min(t, K - 1);
                oa = p0[prep_off_o(d) + e]; ob = p1[prep_off_o(d) + e];
                ja = p0[prep_off_Jh(d) + e]; jb = p1[prep_off_Jh(d) + e];
                ka = p0[prep_off_off(d) + e % 3]; kb = p1[prep_off_off(d) + e % 3];
                wa = p0[prep_off_w(d) + k]; wb = p1[prep_off_w(d) + k];
            }
            const int dec_slot = lm_last_decide(dm, fb, f, dec_in, wx == 0 && t == 0);
#pragma unroll
            for (int i = 0; i < NR; ++i) if (t + 256 * i < 9 * J) s_Rw[t + 256 * i] = dec_slot ? rb[i] : ra[i];
            if (t < 3 * J) { s_o[t] = dec_slot ? ob : oa; s_jp[t] = dec_slot ? jb + kb : ja + ka; }
            if (t < K) s_w[t] = dec_slot ? wb : wa;
        } else {
            const double* pp = fb.prep + ((size_t)f * 2 + fb.ctl[f].cur_slot) * d.prep_size;
            for (int e = t; e < 9 * J; e += 256) s_Rw[e] = pp[prep_off_Rw(d) + e];
            if (t < 3 * J) { s_o[t] = pp[prep_off_o(d) + t]; s_jp[t] = pp[prep_off_Jh(d) + t] + pp[prep_off_off(d) + t % 3]; }
            if (t < K) s_w[t] = pp[prep_off_w(d) + t];
        }
        __syncthreads();
    } else {
        const double* xs = nullptr;
        if (from_state) xs = fb.x + ((size_t)f * 2 + fb.ctl[f].cur_slot) * d.xsize;
        if (t < K) s_w[t] = from_state ? xs[3 + 4 * J + t] : w_in[(size_t)f * K + t];
        if (t < 3) s_p[t] = from_state ? xs[t] : p_in[(size_t)f * 3 + t];
        if (t < J) { s_parent[t] = dm.parent[t]; s_lvl[t] = dm.jlevel[t]; }
        if (t == 0) s_lvl[J] = d.nlevels - 1;
        if (from_state) {
            if (t < J) quat_to_rot(xs + 3 + 4 * t, s_rot + 9 * t);
        } else {
            for (int e = t; e < 9 * J; e += 256) {  // R is column-major per joint -> row-major in LDS
                const int j = e / 9, r = (e % 9) / 3, c = e % 3;
                s_rot[e] = R_in[(size_t)f * 9 * J + 9 * j + 3 * c + r];
            }
        }
        __syncthreads();
        // jointPos = initialJointPos + jointShapeReg * w   (Avatar.cpp:31-36)
        if (t < 3 * J) {
            double s = 0.0;
            for (int k = 0; k < K; ++k) s += dm.jsr[(size_t)t * K + k] * s_w[k];
            s_jp[t] = dm.jsr_base[t] + s;
        }
        __syncthreads();
        fk_chain(J, s_parent, s_rot, s_jp, s_p, s_Rw, s_o, s_lvl);
    }
    // jointPos_i <- t_i ; t_i -= R_i * jPosInit   (Avatar.cpp:59-64)
    if (t < 3 * J) {
        const int j = t / 3, r = t % 3;
        const double tr = s_o[t] - (s_Rw[9 * j + 3 * r] * s_jp[3 * j] + s_Rw[9 * j + 3 * r + 1] * s_jp[3 * j + 1] +
                                    s_Rw[9 * j + 3 * r + 2] * s_jp[3 * j + 2]);
        s_T[12 * j + 9 + r] = tr;
    }
    for (int e = t; e < 9 * J; e += 256) {
        const int j = e / 9, r = (e % 9) / 3, c = e % 3;
        s_T[12 * j + 3 * c + r] = s_Rw[e];
    }
    __syncthreads();
    if (wx == 0) {
        if (t < 3 * J) fb.jointpos[(size_t)f * 3 * J + t] = s_o[t];
        for (int e = t; e < 12 * J; e += 256) fb.jointtrans[(size_t)f * 12 * J + e] = s_T[e];
        // the launch that closes optimize(): the frame's result record (what k_pack_results wrote in a 4.9 us launch of its own behind every
        // optimize() whose results are gathered or copied back).  The accept test of the last trial point - its writer is thread 0 of this
        // workgroup - lies two barriers back: the control block is final.
        if (pack) pack_result_row(fb, f, d.xsize, t, 256);
    }
    const int v = wx * 256 + t;
    if (v >= V) return;
    // shapedCloud = keyClouds * w + baseCloud  (Avatar.cpp:26)
    double sx = 0.0, sy = 0.0, sz = 0.0;
    // (SMPL's ten keys as a compile-time trip count: the loop unrolls and its 30 plane values are requested together - with the count in a
    // register every turn waited for its own three loads, ten memory round trips one after the other; same operations in the same order)
    auto shape_sum = [&](auto kc) __attribute__((always_inline)) {
        constexpr int KC = decltype(kc)::value;
        const int Kn = KC ? KC : K;
#pragma unroll
        for (int k = 0; k < Kn; ++k) {
            const double wk = s_w[k];
            sx += dm.shape_planes[((size_t)k * 3 + 0) * V + v] * wk;
            sy += dm.shape_planes[((size_t)k * 3 + 1) * V + v] * wk;
            sz += dm.shape_planes[((size_t)k * 3 + 2) * V + v] * wk;
        }
    };
    if (K == 10) shape_sum(std::integral_constant<int, 10>{}); else shape_sum(std::integral_constant<int, 0>{});
    sx += dm.shape_planes[((size_t)K * 3 + 0) * V + v];
    sy += dm.shape_planes[((size_t)K * 3 + 1) * V + v];
    sz += dm.shape_planes[((size_t)K * 3 + 2) * V + v];
    // pointTrans = jointTrans * weights (sparse column, CSC order)  (Avatar.cpp:69)
    double pt[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) pt[e] = 0.0;
    double lw[4];
    int lj[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) { lw[a] = dm.lbs_w[(size_t)a * V + v]; lj[a] = dm.lbs_j[(size_t)a * V + v]; }      // (the joint index with the weight, not behind the test of it)
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const double wt = lw[a];
        if (wt != 0.0) {
            const double* T = s_T + 12 * lj[a];
#pragma unroll
            for (int e = 0; e < 12; ++e) pt[e] += T[e] * wt;
        }
    }
    // cloud.col(i) = pointTrans_i * [shaped_i; 1]  (Avatar.cpp:70-73)
    const double cx = pt[0] * sx + pt[3] * sy + pt[6] * sz + pt[9];
    const double cy = pt[1] * sx + pt[4] * sy + pt[7] * sz + pt[10];
    const double cz = pt[2] * sx + pt[5] * sy + pt[8] * sz + pt[11];
    double* cl = fb.cloud + (size_t)f * 3 * V + 3 * (size_t)v;
    cl[0] = cx; cl[1] = cy; cl[2] = cz;
    if (vis_init >= 0) {
        // this launch also resets the per-vertex bookkeeping of the next ICP iteration (saves four memset nodes):
        // visibility flags (0, or 1 when occlusion is off), correspondence counts and fixed-point sums
        fb.visible[(size_t)f * V + v] = (unsigned char)vis_init;
        fb.cnt[(size_t)f * V + v] = 0;
        long long* fs = fb.fsum + (size_t)f * 3 * V;
        fs[v] = 0; fs[(size_t)V + v] = 0; fs[2 * (size_t)V + v] = 0;
    }
    if (dm.part_pos && write_pc) {     // the part-sorted copy k_nn_vis scans (few frames; frame batches gather from the cloud in k_compact)
        const int pp = dm.part_pos[v];
        if (vis_init >= 0) fb.vis_sorted[(size_t)f * V + pp] = (unsigned char)vis_init;
        fb.pcx[(size_t)f * V + pp] = cx;
        fb.pcy[(size_t)f * V + pp] = cy;
        fb.pcz[(size_t)f * V + pp] = cz;
    }
}

// =================================================================================================
// k_lbs_multi<FT>: Avatar::update() of FT frames per workgroup (frame batches).  The skinning of a vertex reads its 3 (K + 1) shape-plane
// values and four (weight, joint) pairs - 312 bytes that are the same for every frame - and k_lbs re-read them from L2 once per frame: at
// 256 frames per launch that is 550 MB through the L2s per launch, what the kernel waits for.  Here a thread loads
// them once and applies them to FT frames (FT sets of joint matrices in LDS): same operations in the same order per frame, hence the same
// bits as k_lbs.  grid (ceil(V / 256) + FT nb, ceil(frames / FT)); from_state 1 (forward kinematics from the current state: the first
// launch of optimize()) or 2 (from the skeleton tables k_solve made for the current point); the trailing nb workgroups per frame are the
// label histogram of the data bucketing (first launch).  No accept test and no trial-point workgroup here: the shapes that have them
// (few frames) keep k_lbs.
// =================================================================================================
template <int FT>
__device__ __forceinline__ void fk_chain_multi(int J, const int* __restrict__ parent, const double* rot, const double* jp, const double* p, double* Rw,
                                               double* o, const int* lvl) {
    const int t = threadIdx.x;
    const int nl = lvl[J];
    for (int L = 0; L <= nl; ++L) {
        for (int idx = t; idx < FT * 12 * J; idx += blockDim.x) {
            const int fi = idx / (12 * J), rem = idx - fi * 12 * J, j = rem / 12, e = rem % 12;
            if (lvl[j] != L) continue;
            const double* rotf = rot + (size_t)fi * 9 * J; const double* jpf = jp + (size_t)fi * 3 * J;
            double* Rwf = Rw + (size_t)fi * 9 * J; double* of = o + (size_t)fi * 3 * J;
            if (j == 0) {
                if (e < 9) Rwf[e] = rotf[e];
                else of[e - 9] = p[3 * fi + e - 9];
            } else {
                const int pa = parent[j];
                const double* Rp = Rwf + 9 * pa;
                if (e < 9) {
                    const int r = e / 3, c = e % 3;
                    Rwf[9 * j + e] = Rp[3 * r] * rotf[9 * j + c] + Rp[3 * r + 1] * rotf[9 * j + 3 + c] + Rp[3 * r + 2] * rotf[9 * j + 6 + c];
                } else {
                    const int r = e - 9;
                    const double d0 = jpf[3 * j] - jpf[3 * pa], d1 = jpf[3 * j + 1] - jpf[3 * pa + 1], d2 = jpf[3 * j + 2] - jpf[3 * pa + 2];
                    of[3 * j + r] = of[3 * pa + r] + (Rp[3 * r] * d0 + Rp[3 * r + 1] * d1 + Rp[3 * r + 2] * d2);
                }
            }
        }
        __syncthreads();
    }
}

template <int FT>
__global__ __launch_bounds__(256) void k_lbs_multi(DeviceModel dm, FrameBuffers fb, int from_state, int vis_init, int nlbs, int nb, int nframes, int write_pc, int pack) {
    const AvtDims d = dm.d;
    const int J = d.J, K = d.K, V = d.V;
    int wx, fy;
    xcd_frame_block(fb, wx, fy);
    const int t = threadIdx.x, y0 = fy * FT;
    if (wx >= nlbs) {      // trailing workgroups: the label histogram of one 2048-point tile of one of my frames
        const int bx = wx - nlbs, fi = bx / nb;
        if (y0 + fi < nframes) bucket_count_block(dm, fb, fb.f0 + y0 + fi, bx - fi * nb);
        return;
    }
    extern __shared__ __attribute__((aligned(16))) char lbs_dyn[];
    // per frame: rot [9 J] | Rw [9 J] | o [3 J] | jp [3 J] | T [12 J] - frame-major inside every array - then w [FT][16], p [FT][3]
    double* s_rot = (double*)lbs_dyn;
    double* s_Rw = s_rot + (size_t)FT * 9 * J;
    double* s_o = s_Rw + (size_t)FT * 9 * J;
    double* s_jp = s_o + (size_t)FT * 3 * J;
    double* s_T = s_jp + (size_t)FT * 3 * J;
    double* s_w = s_T + (size_t)FT * 12 * J;
    double* s_p = s_w + FT * AVT_MAX_SHAPE;
    __shared__ int s_parent[AVT_MAX_JOINTS], s_lvl[AVT_MAX_JOINTS + 1];
    bool on[FT];
#pragma unroll
    for (int i = 0; i < FT; ++i) on[i] = y0 + i < nframes;
    if (from_state == 2) {
#pragma unroll
        for (int i = 0; i < FT; ++i) {
            const int f = fb.f0 + min(y0 + i, nframes - 1);
            const double* pp = fb.prep + ((size_t)f * 2 + fb.ctl[f].cur_slot) * d.prep_size;
            for (int e = t; e < 9 * J; e += 256) s_Rw[(size_t)i * 9 * J + e] = pp[prep_off_Rw(d) + e];
            if (t < 3 * J) { s_o[(size_t)i * 3 * J + t] = pp[prep_off_o(d) + t]; s_jp[(size_t)i * 3 * J + t] = pp[prep_off_Jh(d) + t] + pp[prep_off_off(d) + t % 3]; }
            if (t < K) s_w[i * AVT_MAX_SHAPE + t] = pp[prep_off_w(d) + t];
        }
        __syncthreads();
    } else {
        if (t < J) { s_parent[t] = dm.parent[t]; s_lvl[t] = dm.jlevel[t]; }
        if (t == 0) s_lvl[J] = d.nlevels - 1;
#pragma unroll
        for (int i = 0; i < FT; ++i) {
            const int f = fb.f0 + min(y0 + i, nframes - 1);
            const double* xs = fb.x + ((size_t)f * 2 + fb.ctl[f].cur_slot) * d.xsize;
            if (t < K) s_w[i * AVT_MAX_SHAPE + t] = xs[3 + 4 * J + t];
            if (t < 3) s_p[3 * i + t] = xs[t];
            if (t < J) quat_to_rot(xs + 3 + 4 * t, s_rot + (size_t)i * 9 * J + 9 * t);
        }
        __syncthreads();
        // jointPos = initialJointPos + jointShapeReg * w   (Avatar.cpp:31-36)
        for (int e = t; e < FT * 3 * J; e += 256) {
            const int i = e / (3 * J), tt = e - i * 3 * J;
            double sacc = 0.0;
            for (int k = 0; k < K; ++k) sacc += dm.jsr[(size_t)tt * K + k] * s_w[i * AVT_MAX_SHAPE + k];
            s_jp[e] = dm.jsr_base[tt] + sacc;
        }
        __syncthreads();
        fk_chain_multi<FT>(J, s_parent, s_rot, s_jp, s_p, s_Rw, s_o, s_lvl);
    }
    // jointPos_i <- t_i ; t_i -= R_i * jPosInit   (Avatar.cpp:59-64)
    for (int e = t; e < FT * 3 * J; e += 256) {
        const int i = e / (3 * J), tt = e - i * 3 * J, j = tt / 3, r = tt % 3;
        const double* Rwf = s_Rw + (size_t)i * 9 * J; const double* jpf = s_jp + (size_t)i * 3 * J;
        const double tr = s_o[e] - (Rwf[9 * j + 3 * r] * jpf[3 * j] + Rwf[9 * j + 3 * r + 1] * jpf[3 * j + 1] + Rwf[9 * j + 3 * r + 2] * jpf[3 * j + 2]);
        s_T[(size_t)i * 12 * J + 12 * j + 9 + r] = tr;
    }
    for (int e = t; e < FT * 9 * J; e += 256) {
        const int i = e / (9 * J), ee = e - i * 9 * J, j = ee / 9, r = (ee % 9) / 3, c = ee % 3;
        s_T[(size_t)i * 12 * J + 12 * j + 3 * c + r] = s_Rw[e];
    }
    __syncthreads();
    if (wx == 0) {
#pragma unroll
        for (int i = 0; i < FT; ++i) {
            if (!on[i]) continue;
            const int f = fb.f0 + y0 + i;
            if (t < 3 * J) fb.jointpos[(size_t)f * 3 * J + t] = s_o[(size_t)i * 3 * J + t];
            for (int e = t; e < 12 * J; e += 256) fb.jointtrans[(size_t)f * 12 * J + e] = s_T[(size_t)i * 12 * J + e];
            if (pack) pack_result_row(fb, f, d.xsize, t, 256);      // (the closing launch of optimize(): the frame's result record, see k_lbs)
        }
    }
    const int v = wx * 256 + t;
    if (v >= V) return;
    // shapedCloud = keyClouds * w + baseCloud  (Avatar.cpp:26): every plane value is loaded once and used by all FT frames
    double sx[FT], sy[FT], sz[FT];
#pragma unroll
    for (int i = 0; i < FT; ++i) { sx[i] = 0.0; sy[i] = 0.0; sz[i] = 0.0; }
    for (int k = 0; k < K; ++k) {
        const double p0 = dm.shape_planes[((size_t)k * 3 + 0) * V + v], p1 = dm.shape_planes[((size_t)k * 3 + 1) * V + v], p2 = dm.shape_planes[((size_t)k * 3 + 2) * V + v];
#pragma unroll
        for (int i = 0; i < FT; ++i) {
            const double wk = s_w[i * AVT_MAX_SHAPE + k];
            sx[i] += p0 * wk; sy[i] += p1 * wk; sz[i] += p2 * wk;
        }
    }
    {
        const double b0 = dm.shape_planes[((size_t)K * 3 + 0) * V + v], b1 = dm.shape_planes[((size_t)K * 3 + 1) * V + v], b2 = dm.shape_planes[((size_t)K * 3 + 2) * V + v];
#pragma unroll
        for (int i = 0; i < FT; ++i) { sx[i] += b0; sy[i] += b1; sz[i] += b2; }
    }
    double wt[4];
    int wj[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) { wt[a] = dm.lbs_w[(size_t)a * V + v]; wj[a] = dm.lbs_j[(size_t)a * V + v]; }
    const int pp = (dm.part_pos && write_pc) ? dm.part_pos[v] : -1;
#pragma unroll
    for (int i = 0; i < FT; ++i) {
        if (!on[i]) continue;
        const int f = fb.f0 + y0 + i;
        // pointTrans = jointTrans * weights (sparse column, CSC order)  (Avatar.cpp:69)
        double pt[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) pt[e] = 0.0;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            if (wt[a] != 0.0) {
                const double* T = s_T + (size_t)i * 12 * J + 12 * wj[a];
#pragma unroll
                for (int e = 0; e < 12; ++e) pt[e] += T[e] * wt[a];
            }
        }
        // cloud.col(i) = pointTrans_i * [shaped_i; 1]  (Avatar.cpp:70-73)
        const double cx = pt[0] * sx[i] + pt[3] * sy[i] + pt[6] * sz[i] + pt[9];
        const double cy = pt[1] * sx[i] + pt[4] * sy[i] + pt[7] * sz[i] + pt[10];
        const double cz = pt[2] * sx[i] + pt[5] * sy[i] + pt[8] * sz[i] + pt[11];
        double* cl = fb.cloud + (size_t)f * 3 * V + 3 * (size_t)v;
        cl[0] = cx; cl[1] = cy; cl[2] = cz;
        if (vis_init >= 0) {
            fb.visible[(size_t)f * V + v] = (unsigned char)vis_init;
            fb.cnt[(size_t)f * V + v] = 0;
            long long* fs = fb.fsum + (size_t)f * 3 * V;
            fs[v] = 0; fs[(size_t)V + v] = 0; fs[2 * (size_t)V + v] = 0;
        }
        if (pp >= 0) {
            if (vis_init >= 0) fb.vis_sorted[(size_t)f * V + pp] = (unsigned char)vis_init;
            fb.pcx[(size_t)f * V + pp] = cx;
            fb.pcy[(size_t)f * V + pp] = cy;
            fb.pcz[(size_t)f * V + pp] = cz;
        }
    }
}

static size_t lbs_multi_lds(const AvtDims& d, int FT) { return sizeof(double) * ((size_t)FT * (36 * d.J + AVT_MAX_SHAPE + 3) + 2) + 16; }

// with_bucket_count: also histogram the data labels (first half of launch_bucket) in trailing workgroups;
// with_init: also set up the trial point of the next ICP iteration (one trailing workgroup per frame)
void launch_lbs(avt_ctx* c, int nframes, const double*, const double* w, const double* p, const double* R, int from_state, int vis_init,
                bool with_bucket_count, bool with_init, bool decide, bool write_pc, bool pack) {
    const int nlbs = (c->dm.d.V + 255) / 256;
    const int nb = with_bucket_count ? std::max(1, (c->launch_maxN + BUCKET_TILE - 1) / BUCKET_TILE) : 0;
    // frame batches: several frames per workgroup (k_lbs_multi) - the shapes without an accept test or a trial-point workgroup in the launch
    const int FT = (from_state >= 1 && !with_init && !decide && c->tun.lbs_frames != 1) ? (c->tun.lbs_frames > 1 ? c->tun.lbs_frames : (nframes >= 128 ? 4 : (nframes >= 64 ? 2 : 1))) : 1;
    if (FT > 1) {
        const dim3 g2(nlbs + FT * nb, (nframes + FT - 1) / FT);
        if (FT == 4) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_lbs_multi<4>), g2, dim3(256), lbs_multi_lds(c->dm.d, 4), c->cur_stream, c->dm, c->fb, from_state, vis_init, nlbs, std::max(nb, 1), nframes, write_pc ? 1 : 0, pack ? 1 : 0);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_lbs_multi<2>), g2, dim3(256), lbs_multi_lds(c->dm.d, 2), c->cur_stream, c->dm, c->fb, from_state, vis_init, nlbs, std::max(nb, 1), nframes, write_pc ? 1 : 0, pack ? 1 : 0);
        return;
    }
    dim3 grid(nlbs + (with_init ? 1 : 0) + nb, nframes);
    const size_t lds = with_init ? prep_init_lds_bytes(c->dm.d) : 0;
    hipLaunchKernelGGL(k_lbs, grid, dim3(256), lds, c->cur_stream, c->dm, c->fb, w, p, R, from_state, vis_init, nlbs, with_init ? 1 : 0,
                       decide && from_state == 2 ? 1 : 0, write_pc ? 1 : 0, pack ? 1 : 0);
}

// the trial-point workgroup's scratch must fit beside k_lbs's static LDS (large skeletons fall back to the k_solve INIT launch)
bool avt_lbs_can_init(const AvtDims& d) { return prep_init_lds_bytes(d) <= 96 * 1024; }

__global__ void k_visibility_frame(DeviceModel dm, FrameBuffers fb);
int avt_lbs_set_attributes() {
    return hipFuncSetAttribute((const void*)k_lbs, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess ||
           hipFuncSetAttribute((const void*)k_lbs_multi<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess ||
           hipFuncSetAttribute((const void*)k_lbs_multi<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess ||
           hipFuncSetAttribute((const void*)k_visibility_frame, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess;
}

// =================================================================================================
// back-face visibility (AvatarOptimizer.cpp:1349-1367): a face whose ((p2-p1)x(p1-p3)).z > 1e-4 marks its
// three vertices visible.  `visible` is cleared (or set, when occlusion is off) by a memset node first.
// =================================================================================================
__global__ __launch_bounds__(256) void k_visibility(DeviceModel dm, FrameBuffers fb, int nvis) {
    const int F = dm.d.F, V = dm.d.V;
    const int f = blockIdx.y + fb.f0;
    // trailing workgroups (first ICP iteration only): the scatter pass of the data bucketing (its histogram rode in k_lbs)
    if ((int)blockIdx.x >= nvis) { bucket_scatter_block<false>(dm, fb, f, (int)blockIdx.x - nvis); return; }   // few frames: the fast unordered scatter
    const int face = blockIdx.x * 256 + threadIdx.x;
    if (face >= F) return;
    const int i1 = dm.mesh[face], i2 = dm.mesh[(size_t)F + face], i3 = dm.mesh[2 * (size_t)F + face];
    const double* cl = fb.cloud + (size_t)f * 3 * V;
    const double p1x = cl[3 * i1], p1y = cl[3 * i1 + 1];
    const double p2x = cl[3 * i2], p2y = cl[3 * i2 + 1];
    const double p3x = cl[3 * i3], p3y = cl[3 * i3 + 1];
    const double ax = p2x - p1x, ay = p2y - p1y, bx = p1x - p3x, by = p1y - p3y;
    const double z = __dsub_rn(__dmul_rn(ax, by), __dmul_rn(ay, bx));  // no FMA: same rounding as the CPU expression
    if (z > 1e-4) {
        unsigned char* vis = fb.visible + (size_t)f * V;
        vis[i1] = 1; vis[i2] = 1; vis[i3] = 1;
        if (dm.part_pos) {      // the same flags in part-sorted order: what the fused compaction of k_nn_vis reads
            unsigned char* vs = fb.vis_sorted + (size_t)f * V;
            vs[dm.part_pos[i1]] = 1; vs[dm.part_pos[i2]] = 1; vs[dm.part_pos[i3]] = 1;
        }
    }
}

// k_visibility_frame: the same test with ONE workgroup of 1024 threads per frame.  The x, y of the frame's cloud are read once,
// coalesced, into LDS (16 V bytes: 110 KB for SMPL); the faces gather from there and set byte flags in LDS; the flags leave
// as two contiguous byte rows (vertex order and part-sorted order).  No scattered global stores, no cleared flags needed
// beforehand (k_visibility: 6 x 8-byte gathers and up to 6 scattered byte stores per face, 226 us per 256 frames).
// grid (frames).  Frame batches only (a lone frame is faster spread over 54 workgroups); the scatter pass of the data bucketing,
// which rides in k_visibility's grid, rides in k_compact's here (a 1024-thread workgroup with 117 KB of LDS is no place for it).
__global__ __launch_bounds__(1024) void k_visibility_frame(DeviceModel dm, FrameBuffers fb) {
    const int F = dm.d.F, V = dm.d.V;
    const int f = xcd_frame_1d(fb) + fb.f0, t = threadIdx.x;
    extern __shared__ __attribute__((aligned(16))) char vis_dyn[];
    double2* s_xy = (double2*)vis_dyn;
    unsigned char* s_flag = (unsigned char*)(s_xy + V);
    const double* cl = fb.cloud + (size_t)f * 3 * V;
    for (int v = t; v < V; v += 1024) s_xy[v] = make_double2(cl[3 * v], cl[3 * v + 1]);
    for (int w = t; w < (V + 3) / 4; w += 1024) ((unsigned*)s_flag)[w] = 0u;
    __syncthreads();
    for (int face = t; face < F; face += 1024) {
        const int i1 = dm.mesh[face], i2 = dm.mesh[(size_t)F + face], i3 = dm.mesh[2 * (size_t)F + face];
        const double2 p1 = s_xy[i1], p2 = s_xy[i2], p3 = s_xy[i3];
        const double ax = p2.x - p1.x, ay = p2.y - p1.y, bx = p1.x - p3.x, by = p1.y - p3.y;
        const double z = __dsub_rn(__dmul_rn(ax, by), __dmul_rn(ay, bx));  // no FMA: same rounding as the CPU expression
        if (z > 1e-4) { s_flag[i1] = 1; s_flag[i2] = 1; s_flag[i3] = 1; }
    }
    __syncthreads();
    unsigned char* vis = fb.visible + (size_t)f * V;
    for (int v = t; v < V; v += 1024) vis[v] = s_flag[v];
    if (dm.part_vertices) {
        unsigned char* vs = fb.vis_sorted + (size_t)f * V;
        for (int pos = t; pos < V; pos += 1024) vs[pos] = s_flag[dm.part_vertices[pos]];
    }
}

size_t avt_visibility_frame_lds(const AvtDims& d) { return (size_t)d.V * 16 + (((size_t)d.V + 3) & ~(size_t)3); }

// with_bucket_scatter: also run the scatter pass of the data bucketing (second half of launch_bucket) in trailing workgroups
void launch_visibility(avt_ctx* c, int nframes, int enable, bool with_bucket_scatter) {
    const int V = c->dm.d.V;
    if (!c->lbs_cleared) (void)hipMemsetAsync(c->fb.visible + (size_t)c->fb.f0 * V, enable ? 0 : 1, (size_t)nframes * V, c->cur_stream);
    const int nb = with_bucket_scatter ? std::max(1, (c->launch_maxN + BUCKET_TILE - 1) / BUCKET_TILE) : 0;
    c->scatter_in_compact = false;
    if (enable && c->lbs_cleared && c->vis_frame_min > 0 && nframes >= c->vis_frame_min && !avt_nn_few(c, nframes)) {
        // inside optimize(), frame batches: one workgroup per frame; k_compact follows and takes the scatter workgroups
        hipLaunchKernelGGL(k_visibility_frame, dim3(nframes), dim3(1024), avt_visibility_frame_lds(c->dm.d), c->cur_stream, c->dm, c->fb);
        c->scatter_in_compact = with_bucket_scatter;
    } else if (enable) {
        const int nvis = (c->dm.d.F + 255) / 256;
        hipLaunchKernelGGL(k_visibility, dim3(nvis + nb, nframes), dim3(256), 0, c->cur_stream, c->dm, c->fb, nvis);
    } else if (nb) {
        hipLaunchKernelGGL(k_bucket_scatter, dim3(nb, nframes), dim3(256), 0, c->cur_stream, c->dm, c->fb);
    }
}

// Invariant: the label histogram / scatter cursors (part_cnt) are all zero between API calls.  optimize() restores it in
// k_finalize (one memset node less per call); the stand-alone avt_nn path clears after itself (clear_after).
// avt_state_reset: start state -> working state, both buffers in one launch
__global__ __launch_bounds__(256) void k_state_reset(FrameBuffers fb, int nx, int nctl, int nframes) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < nframes) fb.fault[i] = 0;
    if (i < nx) fb.x[i] = fb.x_start[i];
    const int* src = (const int*)fb.ctl_start;
    int* dst = (int*)fb.ctl;
    if (i < nctl) dst[i] = src[i];
}

void launch_state_reset(avt_ctx* c, int nframes) {
    const int nx = nframes * 2 * c->dm.d.xsize, nctl = nframes * (int)(sizeof(AvtFrameCtl) / sizeof(int));
    hipLaunchKernelGGL(k_state_reset, dim3((std::max(nx, nctl) + 255) / 256), dim3(256), 0, c->stream, c->fb, nx, nctl, nframes);
}

void launch_bucket(avt_ctx* c, int nframes, bool clear_after) {
    const int maxN = c->launch_maxN;
    const int nb = std::max(1, (maxN + BUCKET_TILE - 1) / BUCKET_TILE);
    hipLaunchKernelGGL(k_bucket_count, dim3(nb, nframes), dim3(256), 0, c->cur_stream, c->dm, c->fb);
    hipLaunchKernelGGL(k_bucket_scatter, dim3(nb, nframes), dim3(256), 0, c->cur_stream, c->dm, c->fb);
    if (clear_after)
        (void)hipMemsetAsync(c->fb.part_cnt + (size_t)c->fb.f0 * 2 * (AVT_MAX_PARTS + 1), 0, (size_t)nframes * 2 * (AVT_MAX_PARTS + 1) * sizeof(int), c->cur_stream);
}

// =================================================================================================
// Correspondence finalisation: turns the per-vertex (count, fixed-point sum) accumulated by the NN kernel
// into the compacted list of matched model points (the `caches` of AvatarOptimizer.cpp:1419-1431; sqrt(count) and
// the mean data point are formed where the records are gathered, k_records) and the prior weight rescale
// (AvatarOptimizer.cpp:1457-1458).
// One workgroup of 1024 threads per frame; the compaction keeps the model's vertex order (vorder: by the set of
// tiles a vertex's rows touch, then ascending id), so that batches of 16 matched points share their live tiles.
// =================================================================================================
__global__ __launch_bounds__(1024) void k_finalize(DeviceModel dm, FrameBuffers fb, int first_icp) {
    const double beta_pose = fb.params->beta_pose, beta_shape = fb.params->beta_shape, lambda0 = fb.params->lambda0, nu0 = fb.params->lm_up;
    const int f = xcd_frame_1d(fb) + fb.f0, t = threadIdx.x, V = dm.d.V;
    AvtFrameCtl& ctl = fb.ctl[f];
    if (t < 2 * (AVT_MAX_PARTS + 1)) fb.part_cnt[(size_t)f * 2 * (AVT_MAX_PARTS + 1) + t] = 0;   // bucketing is over: restore the invariant
    if (t == 0) { fb.ride_ctr[f] = 0; fb.spec[f].n = 0; fb.spec[f].next = 0; fb.spec[f].ahead = 0; }
    if (t == 0 && first_icp) fb.fault[f] = 0;      // a fault belongs to the optimize() call that raised it: one nobody downloaded must not mark this call's fit   // a new ICP iteration: no reduction has ridden yet, no speculative step exists
    __shared__ int s_wave_m[16], s_wave_t[16];
    const int chunk = (V + 1023) / 1024;
    const int lo = min(V, t * chunk), hi = min(V, lo + chunk);
    const int* cnt = fb.cnt + (size_t)f * V;
    // positions in the model's tile-set order (DeviceModel::vorder); up to 8 per thread are kept in registers so that the
    // gathers of both passes are in flight together
    constexpr int CH = 8;
    int m = 0, tt = 0, vs[CH], cs[CH];
    const bool small = chunk <= CH;
    if (small) {
#pragma unroll
        for (int u = 0; u < CH; ++u) vs[u] = dm.vorder[min(lo + u, V - 1)];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            cs[u] = (lo + u < hi) ? cnt[vs[u]] : 0;
            m += (cs[u] > 0);
            tt += cs[u];
        }
    } else {
        for (int i = lo; i < hi; ++i) {
            const int c = cnt[dm.vorder[i]];
            m += (c > 0);
            tt += c;
        }
    }
    const int im = wave_incl_scan(m), it = wave_incl_scan(tt);
    if (lane_id() == 63) { s_wave_m[wave_id()] = im; s_wave_t[wave_id()] = it; }
    __syncthreads();
    int mbase = 0, total_m = 0, total_t = 0;
    for (int w = 0; w < 16; ++w) {
        if (w < wave_id()) mbase += s_wave_m[w];
        total_m += s_wave_m[w];
        total_t += s_wave_t[w];
    }
    int pos = mbase + im - m;
    if (small) {
#pragma unroll
        for (int u = 0; u < CH; ++u)
            if (cs[u] > 0) fb.matched[(size_t)f * V + pos++] = vs[u];
    } else {
        for (int i = lo; i < hi; ++i) {
            const int v = dm.vorder[i];
            if (cnt[v] > 0) fb.matched[(size_t)f * V + pos++] = v;
        }
    }
    if (t == 0) {
        ctl.M = total_m;
        ctl.T = total_t;
        ctl.sbp = beta_pose * sqrt((double)total_t) / 15.0;
        ctl.sbs = beta_shape * sqrt((double)total_t) / 15.0;
        if (first_icp) {
            ctl.lambda = lambda0;
            ctl.nu = nu0; ctl.pred = 0.0;      // (read by the gain-ratio schedule only)
            ctl.gn_iterations = 0;
            ctl.accepted = 0;
        }
    }
}

void launch_finalize(avt_ctx* c, int nframes) {
    hipLaunchKernelGGL(k_finalize, dim3(nframes), dim3(1024), 0, c->cur_stream, c->dm, c->fb, c->ran_icp_iters == 0 ? 1 : 0);
    c->fb.const_used = (c->launch_maxN + 255) / 256;      // cost-constant workgroups ride in k_records' grid (avt_eval.hip)
}

// =================================================================================================
// Result record of every resident frame for the batch split's all-gather (include/avt_shard.h): the current state
// x = (p, q, w) followed by AVT_SHARD_STAT_DOUBLES statistics, `stride` doubles per frame.  grid (nframes), block 128.
// =================================================================================================
__global__ __launch_bounds__(128) void k_pack_results(FrameBuffers fb, double* __restrict__ out, int xsize, int stride) {
    // (out == fb.results, stride == xsize + 8: the one layout there is; the closing k_lbs launch of optimize() writes the same rows itself - this kernel
    // remains for results asked for without an optimize() in front, and for a send block that is not fb.results)
    const int f = blockIdx.x, t = threadIdx.x;
    (void)stride;
    FrameBuffers fo = fb;
    fo.results = out;
    pack_result_row(fo, f, xsize, t, 128);
}

void launch_pack_results(avt_ctx* c, int nframes, double* out, int stride) {
    hipLaunchKernelGGL(k_pack_results, dim3(nframes), dim3(128), 0, c->stream, c->fb, out, c->dm.d.xsize, stride);
}
