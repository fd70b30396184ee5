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

// lane N of every row of 16 lanes (= the 16 slots of one point) -> all lanes of that row: a DPP operand modifier
// (v_mov_b32_dpp row_newbcast, tools/ubench/dpp_bcast.hip), no LDS round trip.  Must run with the source lanes switched on.
template <int N>
__device__ __forceinline__ double row_bcast(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x150 + N, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x150 + N, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

template <int N>
__device__ __forceinline__ int row_bcast(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x150 + N, 0xf, 0xf, false); }
// lane l <- lane l - SH of its row (0.0 where that leaves the row)
template <int SH>
__device__ __forceinline__ double row_shr(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x110 + SH, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x110 + SH, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// What a wave needs to turn the records of its 4 points into rows of the tile: the skeleton tables staged in LDS
struct EvalTables {
    const double *Rw, *oo, *Jh, *Gm, *ww, *off, *ident;
};

// One wave, 4 points x 16 slots (after stage_records put the wave's records in LDS): my 12 rows of the live tiles zeroed, then
// every sum over a point's data is spread over the point's 16 lanes (= one DPP row) and what a lane computed reaches the
// others by DPP row broadcasts / row shifts - registers, no LDS round trip: the per-point scalars, the shaped rest position
// (lane k: shape key k, row scan), the blended rotation (lane e9), the <= 4 carried points x_k (lane 3a + c); then one lane
// per (point, ancestor) writes the rotation block, lane k the three rows of shape key k, lanes 0..5 the residual column and
// the translation block.  The only LDS traffic left is the records, the skeleton tables and the tile itself.  Everything is
// wave-local (LDS operations of one wave execute in order), so the caller needs no workgroup barrier around it.
// qw = the wave's point quad (rows 12*qw..12*qw+11 of the tile), Rrec = the wave's record region.
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
                                           int qw, int ln, unsigned long long zmask) {
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
    // ---- per-point scalars: one LDS read per lane, handed to the point's 16 lanes by DPP broadcasts ----------------------
    double pv = 0.0;
    int piv = 0;
    if (slot < 5) pv = R[(3 * K + 6 + slot) * 4 + p4];           // sqrt(count), the 4 weights
    else if (slot < 9) piv = RI[(slot - 5) * 4 + p4];            // the 4 assigned joints
    const double sc = row_bcast<0>(pv);
    const double aw[4] = {row_bcast<1>(pv), row_bcast<2>(pv), row_bcast<3>(pv), row_bcast<4>(pv)};
    const int aj[4] = {row_bcast<5>(piv), row_bcast<6>(piv), row_bcast<7>(piv), row_bcast<8>(piv)};
    // ---- shaped rest position, root-subtracted (CalcShape, :249-272): lane k holds shape key k's term of each coordinate,
    // the 16 lanes are summed by a row scan (fixed order), lane c < 3 adds the base cloud and subtracts the root offset
    double t0 = 0.0, t1 = 0.0, t2 = 0.0;
    if (slot < K) {
        const double wk = ww[slot];
        t0 = R[(3 * slot) * 4 + p4] * wk; t1 = R[(3 * slot + 1) * 4 + p4] * wk; t2 = R[(3 * slot + 2) * 4 + p4] * wk;
    }
    t0 += row_shr<8>(t0); t1 += row_shr<8>(t1); t2 += row_shr<8>(t2);
    t0 += row_shr<4>(t0); t1 += row_shr<4>(t1); t2 += row_shr<4>(t2);
    t0 += row_shr<2>(t0); t1 += row_shr<2>(t1); t2 += row_shr<2>(t2);
    t0 += row_shr<1>(t0); t1 += row_shr<1>(t1); t2 += row_shr<1>(t2);
    const double s0 = row_bcast<15>(t0), s1 = row_bcast<15>(t1), s2 = row_bcast<15>(t2);
    double xh = 0.0;
    if (slot < 3) xh = ((slot == 0 ? s0 : (slot == 1 ? s1 : s2)) + R[(3 * K + slot) * 4 + p4]) - off[slot];
    const double xh0 = row_bcast<0>(xh), xh1 = row_bcast<1>(xh), xh2 = row_bcast<2>(xh);
    if constexpr (!COST) {
        // ---- shape block (:568-580): (sum a_k Rw_k) D_k + sum a_k G_k.  Lane e9 < 9 blends entry e9 of the rotation, all
        // nine entries reach every lane by broadcast, lane k < K writes the three rows of shape key k
        double tb = 0.0;
        if (slot < 9) tb = ((aw[0] * Rw[9 * aj[0] + slot] + aw[1] * Rw[9 * aj[1] + slot]) + aw[2] * Rw[9 * aj[2] + slot]) + aw[3] * Rw[9 * aj[3] + slot];
        const double Tb[9] = {row_bcast<0>(tb), row_bcast<1>(tb), row_bcast<2>(tb), row_bcast<3>(tb), row_bcast<4>(tb),
                              row_bcast<5>(tb), row_bcast<6>(tb), row_bcast<7>(tb), row_bcast<8>(tb)};
        if (slot < K) {
            const double D0 = R[(3 * slot) * 4 + p4], D1 = R[(3 * slot + 1) * 4 + p4], D2 = R[(3 * slot + 2) * 4 + p4];
            double shv[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int e = r * K + slot;
                const double gs = ((aw[0] * Gm[aj[0] * 3 * K + e] + aw[1] * Gm[aj[1] * 3 * K + e]) + aw[2] * Gm[aj[2] * 3 * K + e]) + aw[3] * Gm[aj[3] * 3 * K + e];
                shv[r] = sc * ((Tb[3 * r] * D0 + Tb[3 * r + 1] * D1 + Tb[3 * r + 2] * D2) + gs);
            }
            double* o = s_Jt + (size_t)(d.col_shape + slot) * RS + pi * 3;
            o[0] = shv[0]; o[1] = shv[1]; o[2] = shv[2];
        }
    }
    // ---- x_k = R(-1,k)(x^ - J^_k) + t(-1,k) for the <=4 assigned joints (:508-514): lane 3a + c computes coordinate c of
    // carried point a; the twelve values reach every lane of the point by broadcast
    double xkv = 0.0;
    if (slot < 12) {
        const int a = slot / 3, c = slot - 3 * a;
        const int k = a == 0 ? aj[0] : (a == 1 ? aj[1] : (a == 2 ? aj[2] : aj[3]));
        const double* Rk = Rw + 9 * k + 3 * c;
        const double e0 = xh0 - Jh[3 * k], e1 = xh1 - Jh[3 * k + 1], e2 = xh2 - Jh[3 * k + 2];
        xkv = (Rk[0] * e0 + Rk[1] * e1 + Rk[2] * e2) + oo[3 * k + c];
    }
    const double xkr[12] = {row_bcast<0>(xkv), row_bcast<1>(xkv), row_bcast<2>(xkv), row_bcast<3>(xkv), row_bcast<4>(xkv), row_bcast<5>(xkv),
                            row_bcast<6>(xkv), row_bcast<7>(xkv), row_bcast<8>(xkv), row_bcast<9>(xkv), row_bcast<10>(xkv), row_bcast<11>(xkv)};
    const int aword = COST ? 0 : RI[(4 + slot) * 4 + p4];
    if (aword != 0) {   // rotation block of ancestor j: -2 sqrt(c) [l_j]_x R(-1,parent j)
        const int j = aword & 0xff;
        const unsigned mask = ((unsigned)aword >> 8) & 0xffu;
        // every LDS operand is in registers before the first store to the tile: the compiler cannot prove that the tile and
        // the tables do not alias and would otherwise re-load the tables after every store (one LDS round trip each)
        const int pj = (aword >> 16) & 0xff;
        const double* Rp = pj ? Rw + 9 * (pj - 1) : T.ident;
        double rp[9];
#pragma unroll
        for (int e = 0; e < 9; ++e) rp[e] = Rp[e];
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
    // residual column sqrt(c) (x_m - dbar_m); identity root-translation block (:476-481)
    if (slot < 3) {
        double xm = 0.0;
#pragma unroll
        for (int a = 0; a < 4; ++a) xm += aw[a] * (slot == 0 ? xkr[3 * a] : (slot == 1 ? xkr[3 * a + 1] : xkr[3 * a + 2]));
        s_Jt[(size_t)d.col_res * RS + pi * 3 + slot] = sc * (xm - R[(3 * K + 3 + slot) * 4 + p4]);
    } else if (!COST && slot < 6) {
        const int r = slot - 3;
        s_Jt[(size_t)(d.col_tr + r) * RS + pi * 3 + r] = sc;
    }
}

// SMPL shape (6 column tiles, 21 tile pairs, 12 k-steps per batch = 252 matrix instructions when every tile is live): a wave
// owns the five pairs the host dealt it (AvtDims::pair_deal: even expected loads, avt_model.cpp) and k-steps 3W..3W+2 of the
// split pair.  A and B operands are the same kind of fragment (lane l: tile column (l&15) -> its storage column, row
// k0 + (l>>4)), so a pair costs 24 LDS reads (12 on the diagonal).  pm = the batch's live tile pairs (bit p): one wave-uniform
// scalar branch per pair (matrix instructions ignore EXEC), the 12 k-steps of a live pair are straight-line code.
template <bool DIAG>
__device__ __forceinline__ void mfma_pair(const double* __restrict__ a, const double* __restrict__ b, v4f64& acc) {
    constexpr int NK = AVT_EVAL_ROWS / 4;
    double fa[NK], fbv[NK];
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
        fa[ks] = a[4 * ks];
        fbv[ks] = DIAG ? fa[ks] : b[4 * ks];
    }
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[ks], fbv[ks], acc, 0, 0, 0);
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
// generic skeletons of up to 128 columns, AVT_MAX_TILES (12) up to 192 columns (SMPL-H, SMPL-X)
// Generic skeletons with more than 8 column tiles (SMPL-H, SMPL-X: 11 / 12): tile pairs are dealt to the waves by tile ROW -
// rows in ascending order (longest first) to waves 0 1 2 3 3 2 1 0 0 1 .. - so that a wave reads the fragment of every column
// tile it needs ONCE per k-step (<= NT reads) and uses it as the A operand of its rows and the B operand of their columns,
// instead of two reads per matrix instruction (the fragment reads, 40 per k-step and wave, bounded the phase: 6.4 k clocks
// per batch).  NT is a template argument there, so tiles and accumulator slots are compile-time.
__host__ __device__ constexpr int rd_wave(int r) { return ((r >> 2) & 1) ? 3 - (r & 3) : (r & 3); }
__host__ __device__ constexpr int rd_slots(int NT, int W) { int n = 0; for (int r = 0; r < NT; ++r) if (rd_wave(r) == W) n += NT - r; return n; }
__host__ __device__ constexpr int rd_maxslots(int NT) { int m = 0; for (int w = 0; w < 4; ++w) m = rd_slots(NT, w) > m ? rd_slots(NT, w) : m; return m; }
__host__ __device__ constexpr int rd_pair(int NT, int ti, int tj) { return ti * NT - (ti * (ti - 1)) / 2 + (tj - ti); }

// COST: the evaluation behind which no solve follows - residual column and tile pair (res, res) only (see build_rows)
// SPEC: the body as a spec-cost workgroup runs it (COST = true there): the trial path and the speculative steps' path are two
// instantiations behind ONE scalar branch on the block index in k_eval below - with a run-time flag inside one body the trial path paid
// 0.5 us per launch for carrying the other (same-box A/B of the two builds, tools/kt_lib_ab.sh)
template <int CJ, int CK, int MT, bool COST, bool SPEC>
__device__ __forceinline__ void eval_body(const DeviceModel& dm, const FrameBuffers& fb, int nframes) {
    constexpr bool FIXED = CJ != 0;
    const AvtDims d = dm.d;
    const int J = FIXED ? CJ : d.J, K = FIXED ? CK : d.K, P = 3 + 3 * J + K;
    constexpr bool ROWDEAL = !FIXED && MT > 8;            // NT == MT exactly (launch_eval picks the instantiation)
    const int NT = FIXED ? (3 + 3 * CJ + CK + 16) / 16 : (ROWDEAL ? MT : d.NT), NPAIR = NT * (NT + 1) / 2;
    static_assert(!FIXED || (3 + 3 * CJ + CK + 16) / 16 == 6, "fixed-shape path is written for 6 column tiles");
    constexpr int RS = AVT_EVAL_RS;
    constexpr int MAXPW = FIXED ? 6 : (ROWDEAL ? rd_maxslots(MT) : (MT * (MT + 1) / 2 + 3) / 4);
    const int G = fb.G, t = threadIdx.x;
    int id = blockIdx.x;
    // Spec-cost workgroups (riding shapes, behind the evaluation and prior workgroups; the full evaluation's grid only): the COST of the
    // trial point of every speculative LM step that is still in the queue (AvtSpecCtl: made by the solve launch in front with lambda up,
    // lambda up^2 ..) - residual column and one tile pair, exactly the instructions of the cost-only evaluation - plus its pose-prior
    // scores.  The solve launch behind then knows, when it rejects the trial point, which of the queued steps would be rejected as well,
    // takes those accept tests at once and installs the first step that passes (avt_lm.hip): a run of rejections costs ONE launch pair.
    int spec_s = 0;
    const int nspecwg = nframes * fb.nspec_cost * (G + d.ncomps);      // the spec-cost workgroups come FIRST in the grid (k_eval below): dispatched first, they run beside the trial point's instead of behind them
    if constexpr (SPEC) {
        const int per = G + d.ncomps, ns = fb.nspec_cost, id2 = id;      // ns slots per frame: slot j = the j-th step still in the queue
        const int fl = id2 / (ns * per), rem = id2 - fl * ns * per, slot = rem / per, r = rem - slot * per, fsp = fl + fb.f0;
        const AvtSpecCtl& sp = fb.spec[fsp];
        const int s = sp.next + slot;
        // (a launch past the iteration budget - every accept test of the ICP iteration but the last has been taken - evaluates nothing)
        if (fsp >= fb.spec_frames || s >= sp.n || s >= AVT_MAX_SPEC || !sp.valid[min(s, AVT_MAX_SPEC - 1)] || (fb.seq >= 2 && fb.seq + sp.ahead > fb.max_iters)) return;
        if (r >= G) {
            extern __shared__ __attribute__((aligned(16))) char smem_prior[];
            prior_component_at<256>(dm, fb.x_spec + ((size_t)fsp * AVT_MAX_SPEC + s) * d.xsize,
                                    fb.prior_spec + (((size_t)fsp * AVT_MAX_SPEC + s) * AVT_MAX_COMPS + (r - G)) * AVT_PRIOR_STRIDE, r - G, (double*)smem_prior);
            return;
        }
        spec_s = s;
        id = fl * G + r;
    } else if ((id -= nspecwg) >= nframes * G) {   // trailing workgroups: pose prior of the trial point, one (frame, GMM component) each
        const int id2 = id - nframes * G, fp = id2 / d.ncomps + fb.f0;
        if (!COST && FIXED && fb.nspec_cost > 0 && fp < fb.spec_frames && fb.seq >= 2 && fb.seq + fb.spec[fp].ahead > fb.max_iters) return;      // past the iteration budget
        const int2 cs = *(const int2*)&fb.ctl[fp].cur_slot;      // (cur_slot, try_valid): one load
        if (cs.y == AVT_TRY_DONE) return;                         // the frame met the stopping rule (avt_options::function_tolerance): no trial point
        extern __shared__ __attribute__((aligned(16))) char smem_prior[];
        // (fixed-shape instantiation = SMPL's joint and key counts; a model with a prior has 3 (J - 1) prior dimensions (avt_model_create checks it):
        // as constants they unroll the prior's loops over its rows - its passes become one memory round trip instead of one each)
        if constexpr (FIXED && CJ == 24 && CK == 10) { __builtin_assume(dm.d.ndims == 69); __builtin_assume(dm.d.J == 24); __builtin_assume(dm.d.xsize == 109); }
        prior_component(dm, fb, fp, id2 % d.ncomps, 1 - cs.x, (double*)smem_prior);
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
    // Which batches a workgroup takes.  Few frames (G >= 64, one or two batches per workgroup): batch g, g+G - known without
    // the frame's batch count, so the first records are requested before anything else arrives.  Frame batches (G < 64): the
    // CONTIGUOUS range k_solve INIT dealt it by estimated cost (eval_ranges: the kernel lasts as long as its busiest
    // workgroup; with strided chunks the busiest workgroup of a 150-batch frame at G = 24 had 8 batches against a mean of
    // 6.25); consecutive matched points share their live tiles, so fewer partial tiles are written as well.
    const bool strided = G >= 64;
    if (strided && g < d.nb_max) prefetch(g);
    const int M = ctl.M;
    const int try_slot = 1 - ctl.cur_slot;
    const int try_state = ctl.try_valid;      // (requested with the rest of the control block, tested behind the staging below like the budget word)
    if (!COST && g == G - 1 && t < 64) {      // what the solver roles of the k_solve launch behind this one decide on (AvtSolveSnap); the last workgroup of a frame has the fewest batches
        static_assert(sizeof(AvtFrameCtl) == 160 && sizeof(AvtSpecCtl) == 96, "snapshot copy below");
        double* sn = (double*)(fb.snap + f);
        if (t < 20) sn[t] = ((const double*)&ctl)[t];
        else if (t < 32) sn[t] = ((const double*)(fb.spec + f))[t - 20];
        else if (t < 32 + K) fb.snap[f].xw[t - 32] = fb.x[((size_t)f * 2 + try_slot) * d.xsize + 3 + 4 * J + (t - 32)];
    }
    // Past the iteration budget (folded accept tests, avt_lm.hip): the solve launch behind takes no test and needs no system.  The word is
    // REQUESTED here, in the round trip of the control block, and TESTED behind the staging of the skeleton tables below - where the
    // workgroup waits for memory anyway: a test right here would put a round trip of its own in front of every evaluation (+0.6 us per
    // launch, measured), for the sake of the few idle launches at the end of an ICP iteration.
    const bool budget_watch = !COST && FIXED && fb.nspec_cost > 0 && f < fb.spec_frames && fb.seq >= 2;
    const int ahead_now = budget_watch ? fb.spec[f].ahead : 0;
    const int nb = (M + AVT_EVAL_PTS - 1) / AVT_EVAL_PTS;
    const int* er = fb.erange + (size_t)f * AVT_ERANGE + g;       // cost-balanced contiguous ranges (eval_ranges, avt_lm.hip)
    const int b_first = strided ? g : er[0], b_end = strided ? nb : er[1];
    const int b_step = strided ? G : 1;
    if (!strided && b_first < b_end) prefetch(b_first);

    // LDS: the skeleton tables of the trial point (prep block without the quaternions), the transposed tile with one
    // extra all-zero column that stands in for the padding columns P+1.. of the last column tile, the records of the
    // 4 waves, per-point scratch.  53.2 KB for SMPL: three workgroups per CU.
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int NC = P + 1;                                         // real columns; column NC is the zero column
    const int npre = 15 * J + 3 * J * K, nprep = (npre + K + 3 + 1) & ~1;
    double* s_prep = (double*)smem;                               // Rw o Jh G | w off
    double* s_Jt = s_prep + nprep;                                // [NC + 1][RS]
    double* s_rec = s_Jt + AVT_EVAL_TILE(NC + 1);                 // [4 waves][RQ]
    double* s_ident = s_rec + 4 * RQ;                             // [9]  R(-1,parent of the root) = I

    const double* prep = SPEC ? fb.prep_spec + ((size_t)f * AVT_MAX_SPEC + spec_s) * d.prep_size : fb.prep + ((size_t)f * 2 + try_slot) * d.prep_size;
    // The skeleton tables are REQUESTED, then the two "nothing to do" tests are taken, then the tables are stored: the tests' inputs were requested in
    // front of the tables and arrive in front of them, so an idle workgroup leaves a round trip earlier and a working one waits for nothing it would
    // not have waited for (round 6; the six-tile shape - other skeletons keep the loop and test behind it).
    constexpr int NST = FIXED ? (15 * CJ + 3 * CJ * CK + CK + 3 + 255) / 256 : 0;
    double stg[NST ? NST : 1];
    if constexpr (NST > 0) {
#pragma unroll
        for (int i = 0; i < NST; ++i) { const int e = t + 256 * i; stg[i] = e < npre + K + 3 ? prep[e < npre ? e : e + 4 * J] : 0.0; }
    } else {
        for (int e = t; e < npre + K + 3; e += 256) s_prep[e] = prep[e < npre ? e : e + 4 * J];
    }
    if (budget_watch && fb.seq + ahead_now > fb.max_iters) return;      // (workgroup-uniform; nothing has been written yet)
    // the frame met the stopping rule in this ICP iteration (avt_options::function_tolerance, k_solve): there is no trial point to evaluate
    // (the snapshot above says so to the solver roles of the launch behind)
    if (try_state == AVT_TRY_DONE) return;
    if constexpr (NST > 0) {
#pragma unroll
        for (int i = 0; i < NST; ++i) { const int e = t + 256 * i; if (e < npre + K + 3) s_prep[e] = stg[i]; }
    }
    if (t < 9) s_ident[t] = (t == 0 || t == 4 || t == 8) ? 1.0 : 0.0;
    if (t < RS) s_Jt[(size_t)NC * RS + t] = 0.0;
    // MFMA operand fragments: lane l reads storage column tile_col[tile*16 + (l&15)] (padding -> the zero column), rows k0 + (l>>4)
    // six-tile shape: my wave's five dealt pairs (pair index = bit of the batch word, operand fragments, diagonal or not) and my
    // k-steps of the split pair
    int dp_bit[6];
    bool dp_diag[6];
    const double *dp_a[6], *dp_b[6];
    if constexpr (FIXED) {
        const int wvs = __builtin_amdgcn_readfirstlane(wv);
        const unsigned mydeal = wvs == 0 ? d.pair_deal[0] : (wvs == 1 ? d.pair_deal[1] : (wvs == 2 ? d.pair_deal[2] : d.pair_deal[3]));
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            dp_bit[i] = i < 5 ? (int)((mydeal >> (5 * i)) & 31u) : d.pair_split;
            dp_diag[i] = (mydeal >> (25 + i)) & 1u;
            const int k0 = i < 5 ? 0 : 12 * wvs;     // the split pair: rows 12 W .. 12 W + 11
            // storage columns of my fragments, per wave and dealt pair, from the host (twelve independent loads)
            dp_a[i] = s_Jt + (size_t)dm.deal_col[((wvs * 6 + i) * 2 + 0) * 16 + (ln & 15)] * RS + (ln >> 4) + k0;
            dp_b[i] = s_Jt + (size_t)dm.deal_col[((wvs * 6 + i) * 2 + 1) * 16 + (ln & 15)] * RS + (ln >> 4) + k0;
        }
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
    const double* colp[ROWDEAL ? MT : 1];                         // row deal: my lane's fragment address of every column tile
    if constexpr (ROWDEAL) {
#pragma unroll
        for (int tl = 0; tl < MT; ++tl) colp[tl] = s_Jt + (size_t)dm.tile_col[tl * 16 + (ln & 15)] * RS + (ln >> 4);
    }
    if constexpr (!FIXED && !ROWDEAL) {
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
    for (int b = b_first; b < b_end; b += b_step) {
        __syncthreads();  // previous batch's MFMA reads are done (also covers the prep staging on the first pass)
        EPROBE(0);
        // ---- wave-local from here to the next barrier ------------------------------------------------------
        stage_records<NPF>(s_rec + (size_t)wv * RQ, RQ2, ln, pf);
#ifdef AVT_TIMING
        wave_sync(); __builtin_amdgcn_s_waitcnt(0); EPROBE(2);      // records-wait: the prefetched records have arrived and sit in LDS
#endif
        const int bw = __builtin_amdgcn_readfirstlane(bm_next);   // live tiles / tile pairs of this batch (k_records)
        // (COST: only the tile of the residual column and its diagonal pair)
        const int tm = COST ? (1 << d.res_tile) : (NT > 8 ? (bw & 0xffff) : (int)((unsigned)bw >> 24)), pm = COST ? (1 << d.res_pair) : (bw & 0xffffff);
        if (b + b_step < b_end) prefetch(b + b_step);
        // zeroing passes (5 consecutive storage columns each) that touch a live tile (AvtDims::tile_zpass, avt_model.cpp)
        unsigned long long zmask = 0ull;
#pragma unroll
        for (int ti = 0; ti < MT; ++ti)
            if (ti < NT && ((tm >> ti) & 1)) zmask |= d.tile_zpass[ti];
        build_rows<CJ, CK, MT, COST>(d, tabs, s_Jt, s_rec + (size_t)wv * RQ, wv, ln, zmask);
        EPROBE(3);
        __syncthreads();
        EPROBE(4);
        // MFMA phase: 12 k-steps of 4 rows
        wm |= (unsigned long long)pm;
        if constexpr (FIXED) {
#pragma unroll
            for (int i = 0; i < 5; ++i)
                if ((pm >> dp_bit[i]) & 1) {             // one scalar branch per live pair, straight-line code inside
                    if (dp_diag[i]) mfma_pair<true>(dp_a[i], dp_a[i], acc[i]);
                    else mfma_pair<false>(dp_a[i], dp_b[i], acc[i]);
                }
            if ((pm >> dp_bit[5]) & 1) {
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) {
                    const double fa = dp_a[5][4 * ks], fbv = dp_b[5][4 * ks];
                    acc[5] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, fbv, acc[5], 0, 0, 0);
                }
            }
            EPROBE(1);
        } else if constexpr (ROWDEAL) {
            const double* zc = s_Jt + (size_t)NC * RS + (ln >> 4);           // the zero column, rows k0 + (l >> 4): what a dead tile reads
            const double* cq[MT];
#pragma unroll
            for (int tl = 0; tl < MT; ++tl) cq[tl] = ((tm >> tl) & 1) ? colp[tl] : zc;
            auto role = [&](auto wc) {
                constexpr int W = decltype(wc)::value;
#pragma unroll 1
                for (int k0 = 0; k0 < AVT_EVAL_ROWS; k0 += 4) {
                    double fr[MT];
#pragma unroll
                    for (int tl = 0; tl < MT; ++tl) fr[tl] = tl >= W ? cq[tl][k0] : 0.0;      // (my smallest row is W)
                    int slot = 0;
#pragma unroll
                    for (int r = 0; r < MT; ++r) {
                        if (rd_wave(r) != W) continue;
                        if ((tm >> r) & 1) {                     // wave-uniform: a dead row tile has no live pair
#pragma unroll
                            for (int tj = r; tj < MT; ++tj)
                                acc[slot + tj - r] = __builtin_amdgcn_mfma_f64_16x16x4f64(fr[r], fr[tj], acc[slot + tj - r], 0, 0, 0);
                        }
                        slot += MT - r;
                    }
                }
            };
            switch (__builtin_amdgcn_readfirstlane(wv)) {
                case 0: role(std::integral_constant<int, 0>{}); break;
                case 1: role(std::integral_constant<int, 1>{}); break;
                case 2: role(std::integral_constant<int, 2>{}); break;
                default: role(std::integral_constant<int, 3>{}); break;
            }
        } else {
            // generic shapes: the wave's pairs in chunks of CH; a chunk without a live pair is skipped for the whole batch (one
            // wave-uniform branch), inside a live chunk the dead pairs read the all-zero column, so a k-step is branch-free:
            // the fragment reads of all live chunks go out together, then the matrix instructions (independent accumulators)
            // - instead of a branch, two reads and a wait in front of every single matrix instruction (12.6 k -> ... clocks
            // per batch of the 11-tile shape).
#ifndef AVT_EVAL_CH
#define AVT_EVAL_CH 5
#endif
            constexpr int CH = AVT_EVAL_CH, NCH = (MAXPW + CH - 1) / CH;
            const double* zc = s_Jt + (size_t)NC * RS + (ln >> 4);           // the zero column, rows k0 + (l >> 4)
            const double *qa[MAXPW], *qb[MAXPW];
            bool chl[NCH];
#pragma unroll
            for (int c = 0; c < NCH; ++c) chl[c] = false;
#pragma unroll
            for (int i = 0; i < MAXPW; ++i) {
                const bool live = pr_ti[i] >= 0 && ((tm >> pr_ti[i]) & (tm >> pr_tj[i]) & 1);       // wave-uniform
                qa[i] = live ? pr_a[i] : zc; qb[i] = live ? pr_b[i] : zc;
                chl[i / CH] = chl[i / CH] || live;
            }
#pragma unroll 1
            for (int k0 = 0; k0 < AVT_EVAL_ROWS; k0 += 4) {
                double fa[MAXPW], fbv[MAXPW];
#pragma unroll
                for (int c = 0; c < NCH; ++c)
                    if (chl[c]) {
#pragma unroll
                        for (int i = c * CH; i < (c + 1) * CH && i < MAXPW; ++i) { fa[i] = qa[i][k0]; fbv[i] = qb[i][k0]; }
                    }
#pragma unroll
                for (int c = 0; c < NCH; ++c)
                    if (chl[c]) {
#pragma unroll
                        for (int i = c * CH; i < (c + 1) * CH && i < MAXPW; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[i], fbv[i], acc[i], 0, 0, 0);
                    }
            }
        }
    }
#ifdef AVT_TIMING
    EPROBE(5);
    if (!COST && ln == 0 && f == fb.f0 && g == 0) { for (int k = 0; k < 8; ++k) fb.trace[(size_t)f * 64 + 16 + 8 * wv + k] = (double)tacc[k]; if (wv == 0) fb.trace[(size_t)f * 64 + 56] = (double)(wall_clock64() - wall0); }
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
    if (FIXED && ((wm >> d.pair_split) & 1)) {   // the four waves' shares of the split pair are summed in wave order by wave 0 (wm is workgroup-uniform)
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
    // (a spec-cost workgroup writes its one tile - pair res_pair - into the spec step's own block: the base is shifted so that the store below lands there)
    double* part = SPEC ? fb.partial_spec + (((size_t)f * AVT_MAX_SPEC + spec_s) * AVT_G_MAX + g) * 256 - (size_t)d.res_pair * 256
                           : fb.partial + (((size_t)f * G + g) * NPAIR) * 256;
    if constexpr (ROWDEAL) {
        auto out_role = [&](auto wc) {
            constexpr int W = decltype(wc)::value;
            int slot = 0;
#pragma unroll
            for (int r = 0; r < MT; ++r) {
                if (rd_wave(r) != W) continue;
#pragma unroll
                for (int tj = r; tj < MT; ++tj) {
                    const int p = rd_pair(MT, r, tj);
#pragma unroll
                    for (int q = 0; q < 4; ++q) part[(size_t)p * 256 + q * 64 + ln] = acc[slot + tj - r][q];
                }
                slot += MT - r;
            }
        };
        switch (__builtin_amdgcn_readfirstlane(wv)) {
            case 0: out_role(std::integral_constant<int, 0>{}); break;
            case 1: out_role(std::integral_constant<int, 1>{}); break;
            case 2: out_role(std::integral_constant<int, 2>{}); break;
            default: out_role(std::integral_constant<int, 3>{}); break;
        }
    }
#pragma unroll
    for (int i = 0; i < (ROWDEAL ? 0 : MAXPW); ++i) {
        int p = wv + 4 * i;
        if constexpr (FIXED) p = (i < 5 || wv == 0) ? dp_bit[i] : NPAIR;     // my dealt pairs; wave 0 also writes the split pair
        if (p < NPAIR && (p >= 64 || ((wm >> (p & 63)) & 1))) {   // untouched pairs stay unwritten: k_reduce reads the mask
#pragma unroll
            for (int r = 0; r < 4; ++r) part[(size_t)p * 256 + r * 64 + ln] = acc[i][r];
        }
    }
    if (t == 0) {
        if (SPEC) fb.wmask_spec[((size_t)f * AVT_MAX_SPEC + spec_s) * AVT_G_MAX + g] = wm;
        else fb.wmask[(size_t)f * G + g] = wm;
    }
}

// COST: the evaluation behind which no solve follows - residual column and tile pair (res, res) only (see build_rows)
template <int CJ, int CK, int MT, bool COST>
__global__ __launch_bounds__(256, (CJ != 0) ? 3 : 1) void k_eval(DeviceModel dm, FrameBuffers fb, int nframes) {
    if constexpr (!COST && CJ != 0) {      // the spec-cost workgroups of the riding shapes sit behind the evaluation and prior workgroups (launch_eval)
        if ((int)blockIdx.x < nframes * fb.nspec_cost * (fb.G + dm.d.ncomps)) { eval_body<CJ, CK, MT, true, true>(dm, fb, nframes); return; }
    }
    eval_body<CJ, CK, MT, COST, false>(dm, fb, nframes);
}

static bool eval_fixed_shape(const AvtDims& d) { return d.J == 24 && d.K == 10; }

static size_t eval_lds_bytes(const AvtDims& d) {
    const size_t nprep = ((size_t)15 * d.J + 3 * d.J * d.K + d.K + 3 + 1) & ~(size_t)1;
    return sizeof(double) * (nprep + (size_t)AVT_EVAL_TILE(d.P + 2) + 4 * (size_t)d.rec_quad + 10);
}

// cost_only: the last evaluation of an ICP iteration (launch_reduce(.., decide = true) follows)
void launch_eval(avt_ctx* c, int nframes, bool cost_only, int next_seq) {
    const AvtDims& d = c->dm.d;
    // riding shapes with speculative solver workgroups: the cost of every queued speculative step beside the trial point (k_eval)
    const int nspec = (!cost_only && next_seq >= 1 && eval_fixed_shape(d)) ? avt_solve_nspec(c, nframes) : 0;      // (how many of the queued steps: avt_tuning.spec_cost)
    c->fb.nspec_cost = nspec; c->fb.seq = next_seq;
    dim3 grid((unsigned)nframes * (c->fb.G + std::max(0, d.ncomps)) * (1 + nspec));
    const size_t lds = eval_lds_bytes(d);
#define AVT_EVAL_LAUNCH(...) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_eval<__VA_ARGS__>), grid, dim3(256), lds, c->cur_stream, c->dm, c->fb, nframes)
    if (eval_fixed_shape(d)) { if (cost_only) AVT_EVAL_LAUNCH(24, 10, 6, true); else AVT_EVAL_LAUNCH(24, 10, 6, false); }
    else if (d.NT <= 8) { if (cost_only) AVT_EVAL_LAUNCH(0, 0, 8, true); else AVT_EVAL_LAUNCH(0, 0, 8, false); }
    else {
        switch (d.NT) {       // row-dealt shapes: NT is a template argument
            case 9: if (cost_only) AVT_EVAL_LAUNCH(0, 0, 9, true); else AVT_EVAL_LAUNCH(0, 0, 9, false); break;
            case 10: if (cost_only) AVT_EVAL_LAUNCH(0, 0, 10, true); else AVT_EVAL_LAUNCH(0, 0, 10, false); break;
            case 11: if (cost_only) AVT_EVAL_LAUNCH(0, 0, 11, true); else AVT_EVAL_LAUNCH(0, 0, 11, false); break;
            default: if (cost_only) AVT_EVAL_LAUNCH(0, 0, 12, true); else AVT_EVAL_LAUNCH(0, 0, 12, false); break;
        }
    }
#undef AVT_EVAL_LAUNCH
}

void avt_eval_report_occupancy(const AvtDims& d) {
    int nb = -1;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_eval<24, 10, 6, false>, 256, eval_lds_bytes(d));
    fprintf(stderr, "[avt] k_eval<24,10>: dynamic LDS %zu B, occupancy query -> %d blocks/CU (%s)\n", eval_lds_bytes(d), nb, hipGetErrorString(e));
}

template <int NTX>
static int eval_big_attr() {
    static_assert(NTX <= AVT_MAX_TILES, "row-dealt evaluation shapes up to AVT_MAX_TILES column tiles");
    return hipFuncSetAttribute((const void*)k_eval<0, 0, NTX, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512) != hipSuccess ||
           hipFuncSetAttribute((const void*)k_eval<0, 0, NTX, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512) != hipSuccess;
}

int avt_eval_set_attributes() {
    // the fixed-shape kernel needs < 64 KB of dynamic LDS: leave its attribute alone (raising the cap costs residency);
    // the generic shape may need more.
    return hipFuncSetAttribute((const void*)k_eval<0, 0, 8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess ||
           hipFuncSetAttribute((const void*)k_eval<0, 0, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess ||
           eval_big_attr<9>() || eval_big_attr<10>() || eval_big_attr<11>() || eval_big_attr<12>();
}
