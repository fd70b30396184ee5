// avt_eval.hip — data term of one Gauss-Newton / LM evaluation (gfx950, wave64):
//   k_records : once per ICP iteration, gathers everything an evaluation needs about each matched model point
//               (shape planes, mean data point, sqrt(count), assigned joints/weights, ancestor words) into
//               contiguous per-wave records, so that k_eval streams them with two 16-byte loads per lane, plus one
//               word per batch of 16 points with the 16-column tiles (and tile pairs) its rows touch;
//   k_eval    : residual + analytic Jacobian rows per matched model point (AvatarCostFunctorCache::updateData,
//               AvatarOptimizer.cpp:505-582) staged in LDS and contracted on the fp64 matrix cores
//               (v_mfma_f64_16x16x4_f64) into per-workgroup partial tiles of [J | r]^T W [J | r] - block-sparse: the
//               columns are grouped into tiles by branch of the kinematic tree (build_tile_layout, avt_model.cpp) and
//               only the tile pairs a batch touches are multiplied; extra workgroups of the same grid evaluate the GMM
//               pose prior (avt_prior.h), one component each.
//   (k_reduce / k_solve: avt_lm.hip.)
//
// Algebra used (exact, only the summation order differs from the reference's per-residual-block form):
// all residual blocks matched to model point m share one Jacobian block J_m (AvatarOptimizer.cpp:1445-1449),
// so with c_m = #matches and dbar_m their mean data point,
//   J^T J = sum_m c_m J_m^T J_m,  J^T r = sum_m c_m J_m^T (x_m - dbar_m),
//   sum_i |x_m - d_i|^2 = c_m |x_m - dbar_m|^2 + sum_i |d_i - dbar_m|^2   (2nd term: cost_const_block, avt_device.h).
// Each matched point therefore contributes 3 rows sqrt(c_m) [J_m | x_m - dbar_m] to an augmented matrix
// A (3M x (P+1)); A^T A holds H, g and the data cost at once.
//
// Rotation block in closed form.  The reference builds, per ancestor joint j of a point, the 3x3 block
//   R(-1,parent j) * dRot(q_j, v_j) * localJacobian(q_j)                      (AvatarOptimizer.cpp:524-566)
// with v_j = sum_{k under j} a_k (R(j,k)(x^-J^_k) + t(j,k)).  For a unit quaternion, dRot(q,v)*localJacobian(q)
// is the derivative of R(dq*q) v at dq = (delta,1), i.e. -2 [R(q) v]_x, and R(-1,parent j) [y]_x =
// [R(-1,parent j) y]_x R(-1,parent j).  With l_j = R(-1,j) v_j = sum_{k under j} a_k x_k - (sum a_k) o_j (x_k the
// point carried by assigned joint k, o_j the world origin of joint j) the block is
//   B_j = -2 [l_j]_x R(-1,parent j),
// which needs neither q_j nor the (J+1)^2 relative-transform tables of :303-315.  It equals the reference's
// expression up to rounding (|q_j| = 1 is maintained by the retraction; tests compare against the oracle,
// which evaluates the reference's formula literally).
#include <algorithm>

#include "avt_device.h"
#include "avt_prior.h"

typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef double d2v __attribute__((ext_vector_type(2)));

#ifdef AVT_TIMING
#define EPROBE(k) do { const long long _n = clock64(); tacc[k] += _n - tlast; tlast = _n; } while (0)
#else
#define EPROBE(k) do {} while (0)
#endif

// LDS operations of one wave execute in order; this only stops the compiler from moving accesses across
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// -------------------------------------------------------------------------------------------------
// Matched-point records.  Frame f, batch b (16 matched points), wave quad w (4 points) -> rec_quad doubles:
//   double field F of point p4 at [F*4 + p4], F = 0..3K+2: shape plane k*3+c (k = K: base cloud);
//   3K+3..3K+5: mean data point; 3K+6: sqrt(count); 3K+7..3K+10: assigned weights;
//   then 20 int fields at int index [I*4 + p4]: I = 0..3 assigned joints, 4..19 ancestor words
//   joint | mask << 8 | (parent joint + 1) << 16 | storage column of the joint << 24 (mask bit a: assigned joint a lies
//   under the joint; 0 = no ancestor).
// Points past M in the last batch are written as zeros (sqrt(count) = 0 silences their rows).
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_records(DeviceModel dm, FrameBuffers fb) {
    const AvtDims d = dm.d;
    const int f = blockIdx.y + fb.f0, b = blockIdx.x, t = threadIdx.x, V = d.V, K = d.K;
    if (b >= d.nb_max) {   // trailing workgroups of the grid: the constant part of the data cost (one launch less)
        cost_const_block(dm, fb, f, b - d.nb_max);
        return;
    }
    const int M = fb.ctl[f].M;
    if (b * AVT_EVAL_PTS >= M) return;
    const int wv = t >> 6, ln = t & 63;
    const int ND = 3 * K + 11, RQ = d.rec_quad, RV = RQ >> 2;
    double* R = fb.rec + (((size_t)f * d.nb_max + b) * 4 + wv) * RQ;
    // everything but the mean data point and sqrt(count) is a property of the vertex: copied from its contiguous static record
    // (DeviceModel::vrec; lanes of one point read consecutive doubles)
    for (int e = ln; e < ND * 4; e += 64) {
        const int field = e >> 2, pos = b * AVT_EVAL_PTS + wv * 4 + (e & 3);
        double v = 0.0;
        if (pos < M) {
            const int m = fb.matched[(size_t)f * V + pos];
            if (field >= 3 * (K + 1) && field < 3 * K + 6) {   // mean data point of the vertex (AvatarOptimizer.cpp:1419-1431), from the NN kernel's fixed-point sums
                const int k = field - 3 * (K + 1);
                v = fb.ctl[f].centre[k] + ((double)fb.fsum[((size_t)f * 3 + k) * V + m] / AVT_FIX_SCALE) / (double)fb.cnt[(size_t)f * V + m];
            } else if (field == 3 * K + 6) v = sqrt((double)fb.cnt[(size_t)f * V + m]);
            else v = dm.vrec[(size_t)m * RV + field];
        }
        R[e] = v;
    }
    int* RI = (int*)(R + ND * 4);
    for (int e = ln; e < 80; e += 64) {
        const int ifield = e >> 2, pos = b * AVT_EVAL_PTS + wv * 4 + (e & 3);
        int v = 0;
        if (pos < M) {
            const int m = fb.matched[(size_t)f * V + pos];
            v = ((const int*)(dm.vrec + (size_t)m * RV + ND))[ifield];
        }
        RI[e] = v;
    }
    if (t < 32) {   // tiles this batch's rows touch: J^T J skips the other tile pairs (k_eval)
        const int pos = b * AVT_EVAL_PTS + t;
        int bm = (t < 16 && pos < M) ? (int)dm.vmask[fb.matched[(size_t)f * V + pos]] : 0;
#pragma unroll
        for (int s = 16; s >= 1; s >>= 1) bm |= __shfl_xor(bm, s, 64);
        // word (NT <= 8): bits 24..31 live tiles; bits 0..23 live tile pairs in upper-triangle order (lane p = pair p; used when
        // NT <= 6).  NT > 8: the live tiles in the low 16 bits.
        int p = t, ti = 0;
        while (ti < d.NT && p >= d.NT - ti) { p -= d.NT - ti; ++ti; }
        const bool live = t < 24 && ti < d.NT && ((bm >> ti) & (bm >> (ti + p)) & 1);
        const unsigned long long bal = __ballot(live);
        if (t == 0) fb.bmask[(size_t)f * d.nb_max + b] = d.NT > 8 ? (bm & 0xffff) : ((int)(bal & 0xffffffu) | (bm << 24));
    }
}

void launch_records(avt_ctx* c, int nframes) {
    hipLaunchKernelGGL(k_records, dim3(c->dm.d.nb_max + c->fb.const_used, nframes), dim3(256), 0, c->cur_stream, c->dm, c->fb);
}

// What a wave needs to turn the records of its 4 points into rows of the tile: the skeleton tables staged in LDS
struct EvalTables {
    const double *Rw, *oo, *Jh, *Gm, *ww, *off, *ident;
};

// One wave, 4 points x 16 slots (after stage_records put the wave's records in LDS): my 12 rows of the tile zeroed, shaped rest position,
// the <= 4 carried points x_k, then one lane per (point, ancestor) writes the rotation block; shape block, residual
// column and translation block follow.  Everything is wave-local (LDS operations of one wave execute in order), so the
// caller needs no workgroup barrier around it.  qw = the wave's point quad (rows 12*qw..12*qw+11 of the tile), R = the
// wave's record region, s_xhat / s_xk / s_T = scratch of the 16 points this wave's workgroup (or producer group) builds.
template <int NPF>
__device__ __forceinline__ void stage_records(double* __restrict__ Rrec, int RQ2, int ln, const d2v (&pf)[NPF]) {
    d2v* R2 = (d2v*)Rrec;
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
        const int idx = ln + 64 * i;
        if (idx < RQ2) R2[idx] = pf[i];
    }
}

// COST: only the residual column is built (the last evaluation of an ICP iteration is followed by no solve: the accept test
// needs sum c|r|^2 of the trial point and nothing else; same instructions for that column, hence the same bits).
template <int CJ, int CK, int MT, bool COST>
__device__ __forceinline__ void build_rows(const AvtDims& d, const EvalTables& T, double* __restrict__ s_Jt, const double* __restrict__ Rrec,
                                           double* __restrict__ s_xhat, double* __restrict__ s_xk, double* __restrict__ s_T, int qw, int ln,
                                           unsigned long long zmask) {
    constexpr bool FIXED = CJ != 0;
    constexpr int RS = AVT_EVAL_RS;
    const int J = FIXED ? CJ : d.J, K = FIXED ? CK : d.K, P = 3 + 3 * J + K, NC = P + 1;
    const int ND = 3 * K + 11;
    const int p4 = ln >> 4, slot = ln & 15, pi = qw * 4 + p4;
    const double *Rw = T.Rw, *oo = T.oo, *Jh = T.Jh, *Gm = T.Gm, *ww = T.ww, *off = T.off;
    if (ln < 60) {   // zero my wave's 12 rows of every column of the batch's live tiles: 5 columns x 12 rows per pass (the stride is odd: 8-byte stores)
        double* z = s_Jt + (size_t)(ln / 12) * RS + qw * 12 + (ln % 12);
#pragma unroll
        for (int pass = 0; pass < (16 * MT + 4) / 5; ++pass)
            if ((zmask >> pass) & 1) {          // wave-uniform: some column of this pass lies in a live tile
                if (5 * pass + ln / 12 < NC) z[pass * 5 * RS] = 0.0;
            }
    }
    wave_sync();
    const double* R = Rrec;
    const int* RI = (const int*)(R + ND * 4);
    if (slot < 3) {   // shaped rest position, root-subtracted (CalcShape, :249-272)
        double a = 0.0;
#pragma unroll
        for (int k = 0; k < (FIXED ? CK : AVT_MAX_SHAPE); ++k)
            if (k < K) a += R[(3 * k + slot) * 4 + p4] * ww[k];
        s_xhat[pi * 3 + slot] = (a + R[(3 * K + slot) * 4 + p4]) - off[slot];
    }
    wave_sync();
    const double sc = R[(3 * K + 6) * 4 + p4];
    double aw[4];
    int aj[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) { aw[a] = R[(3 * K + 7 + a) * 4 + p4]; aj[a] = RI[a * 4 + p4]; }
    // x_k = R(-1,k)(x^ - J^_k) + t(-1,k) for the <=4 assigned joints (:508-514); blended rotation T = sum a_k Rw_k
    if (slot < 4) {
        const int k = aj[slot];
        const double* Rk = Rw + 9 * k;
        const double e0 = s_xhat[pi * 3] - Jh[3 * k], e1 = s_xhat[pi * 3 + 1] - Jh[3 * k + 1], e2 = s_xhat[pi * 3 + 2] - Jh[3 * k + 2];
        double* xk = s_xk + (pi * 4 + slot) * 3;
        xk[0] = (Rk[0] * e0 + Rk[1] * e1 + Rk[2] * e2) + oo[3 * k];
        xk[1] = (Rk[3] * e0 + Rk[4] * e1 + Rk[5] * e2) + oo[3 * k + 1];
        xk[2] = (Rk[6] * e0 + Rk[7] * e1 + Rk[8] * e2) + oo[3 * k + 2];
    } else if (!COST && slot < 13) {
        const int e9 = slot - 4;
        s_T[pi * 9 + e9] = ((aw[0] * Rw[9 * aj[0] + e9] + aw[1] * Rw[9 * aj[1] + e9]) + aw[2] * Rw[9 * aj[2] + e9]) + aw[3] * Rw[9 * aj[3] + e9];
    }
    wave_sync();
    const double* xk = s_xk + pi * 12;
    const int aword = COST ? 0 : RI[(4 + slot) * 4 + p4];
    if (aword != 0) {   // rotation block of ancestor j: -2 sqrt(c) [l_j]_x R(-1,parent j)
        const int j = aword & 0xff;
        const unsigned mask = ((unsigned)aword >> 8) & 0xffu;
        // every LDS operand is in registers before the first store to the tile: the compiler cannot prove that the tile and
        // the tables do not alias and would otherwise re-load the tables after every store (one LDS round trip each)
        const int pj = (aword >> 16) & 0xff;
        const double* Rp = pj ? Rw + 9 * (pj - 1) : T.ident;
        double rp[9], xkr[12];
#pragma unroll
        for (int e = 0; e < 9; ++e) rp[e] = Rp[e];
#pragma unroll
        for (int e = 0; e < 12; ++e) xkr[e] = xk[e];
        const double oj0 = oo[3 * j], oj1 = oo[3 * j + 1], oj2 = oo[3 * j + 2];
        double X0 = 0.0, X1 = 0.0, X2 = 0.0, cj = 0.0;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const double wa = (mask & (1u << a)) ? aw[a] : 0.0;      // select, not a branch (adding +0 products leaves the sums unchanged)
            X0 += wa * xkr[3 * a]; X1 += wa * xkr[3 * a + 1]; X2 += wa * xkr[3 * a + 2];
            cj += wa;
        }
        const double m2 = -2.0 * sc;
        const double L0 = m2 * (X0 - cj * oj0), L1 = m2 * (X1 - cj * oj1), L2 = m2 * (X2 - cj * oj2);
        double* o0 = s_Jt + (size_t)((unsigned)aword >> 24) * RS + pi * 3;      // the joint's three storage columns
        double ov[9];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            ov[3 * c] = L1 * rp[6 + c] - L2 * rp[3 + c];
            ov[3 * c + 1] = L2 * rp[c] - L0 * rp[6 + c];
            ov[3 * c + 2] = L0 * rp[3 + c] - L1 * rp[c];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) { o0[c * RS] = ov[3 * c]; o0[c * RS + 1] = ov[3 * c + 1]; o0[c * RS + 2] = ov[3 * c + 2]; }
    }
    // shape block (:568-580): (sum a_k Rw_k) D_k + sum a_k G_k; residual column; identity translation block.
    // All values first, all stores afterwards (see the rotation block).
    constexpr int NSH = (3 * (FIXED ? CK : AVT_MAX_SHAPE) + 15) / 16;
    double shv[NSH];
#pragma unroll
    for (int it = 0; it < NSH; ++it) {
        const int e = slot + 16 * it;
        shv[it] = 0.0;
        if (!COST && e < 3 * K) {
            const int r = e / K, k = e - r * K;
            const double* Tr = s_T + pi * 9 + 3 * r;
            const double gs = ((aw[0] * Gm[aj[0] * 3 * K + e] + aw[1] * Gm[aj[1] * 3 * K + e]) + aw[2] * Gm[aj[2] * 3 * K + e]) + aw[3] * Gm[aj[3] * 3 * K + e];
            const double a = (Tr[0] * R[(3 * k) * 4 + p4] + Tr[1] * R[(3 * k + 1) * 4 + p4] + Tr[2] * R[(3 * k + 2) * 4 + p4]) + gs;
            shv[it] = sc * a;
        }
    }
    double resv = 0.0;
    if (slot < 3) {          // residual column: sqrt(c) (x_m - dbar_m)
        double xm = 0.0;
#pragma unroll
        for (int a = 0; a < 4; ++a) xm += aw[a] * xk[3 * a + slot];
        resv = sc * (xm - R[(3 * K + 3 + slot) * 4 + p4]);
    }
#pragma unroll
    for (int it = 0; it < NSH; ++it) {
        const int e = slot + 16 * it;
        if (!COST && e < 3 * K) {
            const int r = e / K, k = e - r * K;
            s_Jt[(size_t)(d.col_shape + k) * RS + pi * 3 + r] = shv[it];
        }
    }
    if (slot < 3) s_Jt[(size_t)d.col_res * RS + pi * 3 + slot] = resv;
    else if (!COST && slot < 6) {   // identity root-translation block (:476-481)
        const int r = slot - 3;
        s_Jt[(size_t)(d.col_tr + r) * RS + pi * 3 + r] = sc;
    }
}

// upper-triangular tile pairs of the 6x6 tile grid, in the order k_reduce / k_solve decode them
__device__ constexpr int PAIR6_TI[21] = {0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 5};
__device__ constexpr int PAIR6_TJ[21] = {0, 1, 2, 3, 4, 5, 1, 2, 3, 4, 5, 2, 3, 4, 5, 3, 4, 5, 4, 5, 5};

// SMPL shape (6 column tiles, 21 tile pairs, 12 k-steps per batch = 252 matrix instructions when every tile is live):
// wave W owns pairs W, W+4, .., W+16 and k-steps 3W..3W+2 of the last pair (5,5), 63 instructions per wave and batch.
// A and B operands are the same kind of fragment (lane l: tile column (l&15) -> its storage column, row k0 + (l>>4)), so
// a pair costs 24 LDS reads (12 on the diagonal).  pm = the batch's live tile pairs (bit p): one wave-uniform scalar
// branch per pair (matrix instructions ignore EXEC), the 12 k-steps of a live pair are straight-line code.
template <int W>
__device__ __forceinline__ void mfma_batch6(const double* const (&base)[6], int pm, v4f64 (&acc)[6]) {
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int p = W + 4 * i;
        if ((pm >> p) & 1) {                 // one scalar branch per live pair, straight-line code inside
            constexpr int NK = AVT_EVAL_ROWS / 4;
            const int ti = PAIR6_TI[p], tj = PAIR6_TJ[p];
            double fa[NK], fbv[NK];
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                fa[ks] = base[ti][4 * ks];
                fbv[ks] = (ti != tj) ? base[tj][4 * ks] : fa[ks];
            }
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[ks], fbv[ks], acc[i], 0, 0, 0);
        }
    }
    if ((pm >> 20) & 1) {
#pragma unroll
        for (int ks = 3 * W; ks < 3 * W + 3; ++ks) {
            const double f = base[5][4 * ks];
            acc[5] = __builtin_amdgcn_mfma_f64_16x16x4f64(f, f, acc[5], 0, 0, 0);
        }
    }
}

// =================================================================================================
// k_eval.  1-D grid of nframes*G + nframes*ncomps workgroups of 256 threads = 4 waves (1-D on purpose: the hardware
// deals workgroups to shader engines by linear id, and a 2-D grid whose x extent is a multiple of 16 would send all
// the short-lived prior workgroups to one half of the chip and all the evaluation workgroups to the other).
// Workgroup g of frame f walks batches g, g+G, ... of
// the frame's matched-point records (16 points = 48 rows of the augmented matrix per batch; wave w owns points
// 4w..4w+3, lane = (point, slot) with 16 slots per point).
//   Per batch, wave-local (no workgroup barrier): records registers -> LDS, next batch's records requested,
//     own 12 rows of the tile zeroed, shaped rest position, the <= 4 carried points x_k, then one lane per
//     (point, ancestor) writes B_j scaled by sqrt(c_m); shape block, residual column, translation block.
//   The tile is stored transposed [column][row] with an odd row stride of 49 doubles (MFMA operand fetch = 16
//     columns x 4 rows per ds_read_b64 / two k-steps per ds_read2_b64, conflict-free either way).
//   MFMA phase between two workgroup barriers: the upper-triangular 16x16 output tiles accumulate in registers
//     over all batches of the workgroup and leave as NPAIR partial tiles (reduced in fixed order by k_reduce, which
//     also maps tile coordinates back to parameter indices); tile pairs the batch does not touch are skipped.
// CJ/CK != 0: dimensions fixed at compile time (SMPL: 24 joints, 10 shape keys); 0: taken from the model.
// =================================================================================================
// MT = column tiles the instantiation is sized for (accumulators per wave, zeroing passes): 6 for the SMPL shape, 8 for
// generic skeletons of up to 128 columns, AVT_MAX_TILES (11) up to 176 columns (SMPL-H)
// COST: the evaluation behind which no solve follows - residual column and tile pair (res, res) only (see build_rows)
template <int CJ, int CK, int MT, bool COST>
__global__ __launch_bounds__(256, (CJ != 0) ? 3 : (MT > 8 ? 1 : 2)) void k_eval(DeviceModel dm, FrameBuffers fb, int nframes) {
    constexpr bool FIXED = CJ != 0;
    const AvtDims d = dm.d;
    const int J = FIXED ? CJ : d.J, K = FIXED ? CK : d.K, P = 3 + 3 * J + K;
    const int NT = FIXED ? (3 + 3 * CJ + CK + 16) / 16 : d.NT, NPAIR = NT * (NT + 1) / 2;
    static_assert(!FIXED || (3 + 3 * CJ + CK + 16) / 16 == 6, "fixed-shape path is written for 6 column tiles");
    constexpr int RS = AVT_EVAL_RS;
    constexpr int MAXPW = FIXED ? 6 : (MT * (MT + 1) / 2 + 3) / 4;
    const int G = fb.G, t = threadIdx.x;
    const int id = blockIdx.x;
    if (id >= nframes * G) {   // trailing workgroups: pose prior of the trial point, one (frame, GMM component) each
        const int id2 = id - nframes * G, fp = id2 / d.ncomps + fb.f0;
        extern __shared__ __attribute__((aligned(16))) char smem_prior[];
        prior_component(dm, fb, fp, id2 % d.ncomps, 1 - fb.ctl[fp].cur_slot, (double*)smem_prior);
        return;
    }
    const int f = id / G + fb.f0, g = id % G;
    const AvtFrameCtl& ctl = fb.ctl[f];
    const int wv = t >> 6, ln = t & 63, p4 = ln >> 4, slot = ln & 15, pi = wv * 4 + p4;
    const int ND = 3 * K + 11, RQ = 12 * K + 84, RQ2 = RQ / 2;
    constexpr int NPF = FIXED ? (12 * CK + 84 + 127) / 128 : (12 * AVT_MAX_SHAPE + 84 + 127) / 128;

    // the first batch's records do not depend on anything else this kernel loads: request them first
    const d2v* recf = (const d2v*)(fb.rec + (size_t)f * d.nb_max * 4 * RQ);
    d2v pf[NPF];
#pragma unroll
    for (int i = 0; i < NPF; ++i) pf[i] = (d2v){0.0, 0.0};
    int bm_next = 0;                                               // live tiles / pairs word of the batch being prefetched
    auto prefetch = [&](int b) {
        bm_next = fb.bmask[(size_t)f * d.nb_max + b];
        const d2v* src = recf + ((size_t)b * 4 + wv) * RQ2;
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int idx = ln + 64 * i;
            pf[i] = (idx < RQ2) ? __builtin_nontemporal_load(src + idx) : (d2v){0.0, 0.0};
        }
    };
    // a workgroup takes chunks of S consecutive batches (consecutive matched points share their live tiles), chunk g, g+G, ...
    const int S = AVT_EVAL_CHUNK(G);
    auto next_batch = [&](int b) { const int b1 = b + 1; return (b1 % S) ? b1 : b1 - S + G * S; };
    if (g * S < d.nb_max) prefetch(g * S);
    const int M = ctl.M;
    const int try_slot = 1 - ctl.cur_slot;
    const int nb = (M + AVT_EVAL_PTS - 1) / AVT_EVAL_PTS;

    // LDS: the skeleton tables of the trial point (prep block without the quaternions), the transposed tile with one
    // extra all-zero column that stands in for the padding columns P+1.. of the last column tile, the records of the
    // 4 waves, per-point scratch.  53.2 KB for SMPL: three workgroups per CU.
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int NC = P + 1;                                         // real columns; column NC is the zero column
    const int npre = 15 * J + 3 * J * K, nprep = (npre + K + 3 + 1) & ~1;
    double* s_prep = (double*)smem;                               // Rw o Jh G | w off
    double* s_Jt = s_prep + nprep;                                // [NC + 1][RS]
    double* s_rec = s_Jt + AVT_EVAL_TILE(NC + 1);                 // [4 waves][RQ]
    double* s_xhat = s_rec + 4 * RQ;                              // [16][3]
    double* s_xk = s_xhat + 48;                                   // [16][4][3]
    double* s_T = s_xk + 192;                                     // [16][9]  blended rotation per point
    double* s_ident = s_T + 144;                                  // [9]  R(-1,parent of the root) = I

    const double* prep = fb.prep + ((size_t)f * 2 + try_slot) * d.prep_size;
    for (int e = t; e < npre + K + 3; e += 256) s_prep[e] = prep[e < npre ? e : e + 4 * J];
    if (t < 9) s_ident[t] = (t == 0 || t == 4 || t == 8) ? 1.0 : 0.0;
    if (t < RS) s_Jt[(size_t)NC * RS + t] = 0.0;
    // MFMA operand fragments: lane l reads storage column tile_col[tile*16 + (l&15)] (padding -> the zero column), rows k0 + (l>>4)
    const double* fbase[6];
    if constexpr (FIXED) {
#pragma unroll
        for (int ti = 0; ti < 6; ++ti) fbase[ti] = s_Jt + (size_t)dm.tile_col[ti * 16 + (ln & 15)] * RS + (ln >> 4);
    }
    const double* Rw = s_prep;                                    // prep_off_Rw = 0
    const double* oo = s_prep + 9 * J;
    const double* Jh = s_prep + 12 * J;
    const double* Gm = s_prep + 15 * J;
    const double* ww = s_prep + npre;
    const double* off = ww + K;
    const EvalTables tabs{Rw, oo, Jh, Gm, ww, off, s_ident};

    // generic shapes: static round-robin deal of whole tile pairs to the waves
    int pr_ti[MAXPW], pr_tj[MAXPW];
    const double *pr_a[MAXPW], *pr_b[MAXPW];
    if constexpr (!FIXED) {
#pragma unroll
        for (int i = 0; i < MAXPW; ++i) {
            int p = wv + 4 * i, ti = 0;
            pr_a[i] = s_Jt; pr_b[i] = s_Jt;
            if (p < NPAIR) {
                while (p >= NT - ti) { p -= NT - ti; ++ti; }
                pr_ti[i] = ti; pr_tj[i] = ti + p;
                pr_a[i] = s_Jt + (size_t)dm.tile_col[pr_ti[i] * 16 + (ln & 15)] * RS + (ln >> 4);
                pr_b[i] = s_Jt + (size_t)dm.tile_col[pr_tj[i] * 16 + (ln & 15)] * RS + (ln >> 4);
            } else { pr_ti[i] = -1; pr_tj[i] = -1; }
        }
    }
    v4f64 acc[MAXPW];
#pragma unroll
    for (int i = 0; i < MAXPW; ++i) acc[i] = (v4f64){0.0, 0.0, 0.0, 0.0};

#ifdef AVT_TIMING
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long tlast = clock64(); const long long wall0 = wall_clock64();
#endif
    // tile pairs this workgroup accumulated into: only those partial tiles are written, k_reduce reads the mask
    unsigned long long wm = FIXED ? 0ull : ~0ull;     // generic shapes write every pair (k_reduce treats pairs >= 64 as written)
    for (int b = g * S; b < nb; b = next_batch(b)) {
        __syncthreads();  // previous batch's MFMA reads are done (also covers the prep staging on the first pass)
        EPROBE(0);
        // ---- wave-local from here to the next barrier ------------------------------------------------------
        stage_records<NPF>(s_rec + (size_t)wv * RQ, RQ2, ln, pf);
        const int bw = __builtin_amdgcn_readfirstlane(bm_next);   // live tiles / tile pairs of this batch (k_records)
        // (COST: only the tile of the residual column and its diagonal pair)
        const int tm = COST ? (1 << d.res_tile) : (NT > 8 ? (bw & 0xffff) : (int)((unsigned)bw >> 24)), pm = COST ? (1 << d.res_pair) : (bw & 0xffffff);
        if (next_batch(b) < nb) prefetch(next_batch(b));
        // zeroing passes (5 consecutive storage columns each) that touch a live tile (AvtDims::tile_zpass, avt_model.cpp)
        unsigned long long zmask = 0ull;
#pragma unroll
        for (int ti = 0; ti < MT; ++ti)
            if (ti < NT && ((tm >> ti) & 1)) zmask |= d.tile_zpass[ti];
        build_rows<CJ, CK, MT, COST>(d, tabs, s_Jt, s_rec + (size_t)wv * RQ, s_xhat, s_xk, s_T, wv, ln, zmask);
        EPROBE(3);
        __syncthreads();
        EPROBE(4);
        // MFMA phase: 12 k-steps of 4 rows
        wm |= (unsigned long long)pm;
        if constexpr (FIXED) {
            switch (wv) {
                case 0: mfma_batch6<0>(fbase, pm, acc); break;
                case 1: mfma_batch6<1>(fbase, pm, acc); break;
                case 2: mfma_batch6<2>(fbase, pm, acc); break;
                default: mfma_batch6<3>(fbase, pm, acc); break;
            }
            EPROBE(1);
        } else {
#pragma unroll 1
            for (int k0 = 0; k0 < AVT_EVAL_ROWS; k0 += 4) {
#pragma unroll
                for (int i = 0; i < MAXPW; ++i) {
                    if (pr_ti[i] >= 0 && ((tm >> pr_ti[i]) & (tm >> pr_tj[i]) & 1)) {
                        acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(pr_a[i][k0], pr_b[i][k0], acc[i], 0, 0, 0);
                    }
                }
            }
        }
    }
#ifdef AVT_TIMING
    EPROBE(5);
    if (ln == 0 && f == fb.f0 && g == 0) { for (int k = 0; k < 8; ++k) fb.trace[(size_t)f * 64 + 16 + 8 * wv + k] = (double)tacc[k]; if (wv == 0) fb.trace[(size_t)f * 64 + 56] = (double)(wall_clock64() - wall0); }
#endif
#ifdef AVT_TIMELINE
    if (t == 0 && g < 8) {   // where and when this workgroup ran (tools/eval_block_timeline.py, -DAVT_TIMELINE builds)
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        double* tr = fb.trace + (size_t)f * 64 + 12 + 3 * g;
        tr[0] = (double)wall0; tr[1] = (double)wall_clock64(); tr[2] = (double)((xcc & 0xf) * 65536 + (hw & 0xffff));
    }
#endif
    if (FIXED && ((wm >> 20) & 1)) {   // the four waves' shares of pair (5,5) are summed in wave order by wave 0 (wm is workgroup-uniform)
        __syncthreads();
        if (wv > 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) s_Jt[(wv - 1) * 256 + r * 64 + ln] = acc[5][r];
        }
        __syncthreads();
        if (wv == 0) {
#pragma unroll
            for (int w = 0; w < 3; ++w)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[5][r] += s_Jt[w * 256 + r * 64 + ln];
        }
    }
    // partial tiles out: element (row = (ln>>4) + 4*reg, col = ln&15) of pair p at [p][reg*64 + ln]
    double* part = fb.partial + (((size_t)f * G + g) * NPAIR) * 256;
#pragma unroll
    for (int i = 0; i < MAXPW; ++i) {
        const int p = wv + 4 * i;
        if (p < NPAIR && (p >= 64 || ((wm >> (p & 63)) & 1))) {   // untouched pairs stay unwritten: k_reduce reads the mask
#pragma unroll
            for (int r = 0; r < 4; ++r) part[(size_t)p * 256 + r * 64 + ln] = acc[i][r];
        }
    }
    if (t == 0) fb.wmask[(size_t)f * G + g] = wm;
}

static bool eval_fixed_shape(const AvtDims& d) { return d.J == 24 && d.K == 10; }

static size_t eval_lds_bytes(const AvtDims& d) {
    const size_t nprep = ((size_t)15 * d.J + 3 * d.J * d.K + d.K + 3 + 1) & ~(size_t)1;
    return sizeof(double) * (nprep + (size_t)AVT_EVAL_TILE(d.P + 2) + 4 * (size_t)d.rec_quad + 48 + 192 + 144 + 10);
}

// cost_only: the last evaluation of an ICP iteration (launch_reduce(.., decide = true) follows)
void launch_eval(avt_ctx* c, int nframes, bool cost_only) {
    const AvtDims& d = c->dm.d;
    dim3 grid((unsigned)nframes * (c->fb.G + std::max(0, d.ncomps)));
    const size_t lds = eval_lds_bytes(d);
#define AVT_EVAL_LAUNCH(...) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_eval<__VA_ARGS__>), grid, dim3(256), lds, c->cur_stream, c->dm, c->fb, nframes)
    if (eval_fixed_shape(d)) { if (cost_only) AVT_EVAL_LAUNCH(24, 10, 6, true); else AVT_EVAL_LAUNCH(24, 10, 6, false); }
    else if (d.NT <= 8) { if (cost_only) AVT_EVAL_LAUNCH(0, 0, 8, true); else AVT_EVAL_LAUNCH(0, 0, 8, false); }
    else { if (cost_only) AVT_EVAL_LAUNCH(0, 0, AVT_MAX_TILES, true); else AVT_EVAL_LAUNCH(0, 0, AVT_MAX_TILES, false); }
#undef AVT_EVAL_LAUNCH
}

void avt_eval_report_occupancy(const AvtDims& d) {
    int nb = -1;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_eval<24, 10, 6, false>, 256, eval_lds_bytes(d));
    fprintf(stderr, "[avt] k_eval<24,10>: dynamic LDS %zu B, occupancy query -> %d blocks/CU (%s)\n", eval_lds_bytes(d), nb, hipGetErrorString(e));
}

int avt_eval_set_attributes() {
    // the fixed-shape kernel needs < 64 KB of dynamic LDS: leave its attribute alone (raising the cap costs residency);
    // the generic shape may need more.
    return hipFuncSetAttribute((const void*)k_eval<0, 0, 8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess ||
           hipFuncSetAttribute((const void*)k_eval<0, 0, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess ||
           hipFuncSetAttribute((const void*)k_eval<0, 0, AVT_MAX_TILES, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512) != hipSuccess ||
           hipFuncSetAttribute((const void*)k_eval<0, 0, AVT_MAX_TILES, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512) != hipSuccess;
}
