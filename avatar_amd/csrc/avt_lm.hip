// avt_lm.hip — the Gauss-Newton / Levenberg-Marquardt step kernels (gfx950, wave64):
//   k_reduce : fixed-order reduction of k_eval's partial MFMA tiles into the dense symmetric data-term system
//              [J|r]^T W [J|r] of the trial point;
//   k_solve  : one workgroup per frame — LM accept/reject, prior assembly (AvatarOptimizer.cpp:647-726,
//              :1457-1458), damped LDL^T solve held in registers, quaternion retraction
//              (FakeQuaternionParameterization::Plus, :123-143) and the skeleton tables of the next trial point
//              (PrepareForEvaluation, :283-325).  The whole inner loop runs without host synchronisation.
//
// Everything in k_solve is latency-bound (an 85-long pivot chain), so it is organised around the dependency
// chain: one global round trip for all inputs, no divide / sqrt on the chain (v_rcp_f64 + cubic Newton), one barrier
// per 4 pivots, every lane keeps its own copy of its column's diagonal block instead of reading a published one,
// back-substitution four unknowns at a time with cross-lane v_readlane, the skeleton pass from LDS only, and no global
// load inside any sequential loop.  -DAVT_TIMING adds s_memtime probes (tools/kernel_timing_probe.py).
#include <algorithm>

#include <type_traits>

#include <cstring>
#include "avt_device.h"

#ifdef AVT_TIMING
#define TPROBE(i) do { if (threadIdx.x == 0 && blockIdx.y == 0) fb.trace[(size_t)(blockIdx.x + fb.f0) * 64 + 40 + (i)] = (double)clock64(); } while (0)
// the start of the launch is kept in a register and written with probe 1: the launches that install a step or only decide return in front of
// probe 1, and their start must not overwrite that of the last full solve
#define TPROBE_START() const long long tprobe_start = clock64()
#define TPROBE_FIRST() do { if (threadIdx.x == 0 && blockIdx.y == 0) { double* tp_ = fb.trace + (size_t)(blockIdx.x + fb.f0) * 64 + 40; tp_[0] = (double)tprobe_start; tp_[1] = (double)clock64(); } } while (0)
#else
#define TPROBE(i) do {} while (0)
#define TPROBE_START() do {} while (0)
#define TPROBE_FIRST() do {} while (0)
#endif

#include "avt_prep.h"

// =================================================================================================
// k_reduce<Q, DECIDE>.  grid (NPAIR, nframes), block Q x 256 tile elements.  Q = 1 is the shape of frame batches (few partials
// per pair, many workgroups); few frames (G >= 64) use k_reduce_strip below (Q = 4, one workgroup of four quarters per pair, was
// that shape until late round 2 and is kept as the reference form of it).
//   Hraw[f][try][r][c] = sum_g partial[f][g][pair][e] in a fixed order (deterministic): quarter q sums its
//   contiguous range of workgroups g in ascending order with up to 32 loads in flight (the kernel is a chain of
//   L2 round trips, so what counts is how few rounds it takes), the quarter sums are added in order 0..Q-1.
//   The tile is written to both triangles of the dense (HS x HS) symmetric block; row/column P carry J^T r and
//   sum c|r|^2.  (The GMM pose prior of the trial point is evaluated by extra workgroups of k_eval, avt_prior.h.)
// =================================================================================================
template <int Q>
__global__ __launch_bounds__(256 * Q) void k_reduce(DeviceModel dm, FrameBuffers fb) {
    __builtin_amdgcn_s_setprio(3);
    const AvtDims d = dm.d;
    const int f = blockIdx.y + fb.f0, t = threadIdx.x & 255, q = threadIdx.x >> 8, NPAIR = d.NPAIR, NT = d.NT, P = d.P, HS = d.HS;
    const int G = fb.G, glo = (G * q) / Q, ghi = (G * (q + 1)) / Q;
    const int pair = (int)blockIdx.x;
    const double* part = fb.partial + ((size_t)f * G * NPAIR + pair) * 256 + t;
    const size_t st = (size_t)NPAIR * 256;
    // which of my workgroups g wrote this pair (k_eval skips the tile pairs its batches never touch): lane l asks for
    // g = glo + l, the ballot makes the answer wave-uniform (ghi - glo <= 64 in both launch shapes).  Q = 1 (frame batches)
    // asks first and skips the loads; Q = 4 (few frames, one L2 round trip long) requests mask and tiles together and
    // discards what was never written - there the point is the producer's shorter store tail, not the loads.
    const int lane = threadIdx.x & 63;
    const unsigned long long wmine = (glo + lane < ghi) ? fb.wmask[(size_t)f * G + glo + lane] : 0ull;
    unsigned long long live = ~0ull;
    // (the mask has 64 bits; only skeletons with more than 64 tile pairs - evaluated by the generic kernel, which writes
    // every pair - have pairs beyond it)
    const bool masked = pair < 64;
    if constexpr (Q == 1) live = masked ? __ballot((int)((wmine >> (pair & 63)) & 1ull)) : ~0ull;
    double a = 0.0;
    int g = glo;
#define AVT_REDUCE_ROUND(NLD)                                                                        \
    for (; g + NLD <= ghi; g += NLD) {                                                                \
        double v[NLD];                                                                               \
        _Pragma("unroll") for (int u = 0; u < NLD; ++u) v[u] = ((live >> (g + u - glo)) & 1ull) ? __builtin_nontemporal_load(part + (size_t)(g + u) * st) : 0.0; \
        _Pragma("unroll") for (int u = 0; u < NLD; ++u) a += v[u];                                    \
    }
    if constexpr (Q > 1) {      // G/4 <= 32 tiles per quarter: all in flight together with the mask, selected afterwards
        double v[32];
        const int ng = ghi - glo;
#pragma unroll
        for (int u = 0; u < 32; ++u) v[u] = (u < ng) ? __builtin_nontemporal_load(part + (size_t)(glo + u) * st) : 0.0;
        const unsigned long long wrote = masked ? __ballot((int)((wmine >> (pair & 63)) & 1ull)) : ~0ull;
#pragma unroll
        for (int u = 0; u < 32; ++u) a += ((wrote >> u) & 1ull) ? v[u] : 0.0;
        g = ghi;
    }
    AVT_REDUCE_ROUND(16) AVT_REDUCE_ROUND(8) AVT_REDUCE_ROUND(4) AVT_REDUCE_ROUND(1)
#undef AVT_REDUCE_ROUND
    const int try_slot = 1 - fb.ctl[f].cur_slot;
    __shared__ double s_q[Q > 1 ? Q - 1 : 1][256];
    if constexpr (Q > 1) {
        if (q > 0) s_q[q - 1][t] = a;
        __syncthreads();
    }
    if (q == 0) {
        if constexpr (Q > 1) {
#pragma unroll
            for (int i = 0; i < Q - 1; ++i) a += s_q[i][t];
        }
        int p = pair, ti = 0;
        while (p >= NT - ti) { p -= NT - ti; ++ti; }
        const int tj = ti + p;
        // tile coordinates -> parameter indices (the evaluation tile has its own column order, avt_model.cpp)
        const int r = dm.tile_param[ti * 16 + ((t >> 4) & 3) + 4 * (t >> 6)], c = dm.tile_param[tj * 16 + (t & 15)];
        if (r >= 0 && c >= 0) {
            double* H = fb.Hraw + ((size_t)f * 2 + try_slot) * HS * HS;
            H[(size_t)r * HS + c] = a;
            if (ti != tj) H[(size_t)c * HS + r] = a;
        }
    }
}

// k_reduce_strip<NS>: the few-frames shape (G >= 64) of the reduction.  One workgroup per tile pair pulls G x 2 KB =
// 256 KB of partial tiles through one CU (k_reduce<4>: 7.3 us per launch on one frame); here a pair is NS workgroups on NS CUs,
// each reducing a strip of 256 / NS tile elements (contiguous in every partial tile): 1024 threads = EL elements x NSL slices of
// the G workgroups, <= 8 loads per thread in flight together with the written-masks, the slice sums added in fixed order
// through LDS.  grid (NS NPAIR, frames).  4.7 us per launch on one frame.
template <int NS>
__global__ __launch_bounds__(1024) void k_reduce_strip(DeviceModel dm, FrameBuffers fb) {
    __builtin_amdgcn_s_setprio(3);
    constexpr int EL = 256 / NS, NSL = 1024 / EL, NLD = (AVT_G_MAX + NSL - 1) / NSL;     // elements per strip, slices, loads per thread (G <= AVT_G_MAX)
    const AvtDims d = dm.d;
    const int f = blockIdx.y + fb.f0, el = threadIdx.x % EL, slice = threadIdx.x / EL, strip = blockIdx.x % NS;
    const int NPAIR = d.NPAIR, NT = d.NT, P = d.P, HS = d.HS;
    const int pair = (int)(blockIdx.x / NS);
    const int e = strip * EL + el;                       // element of the tile: (row = (e >> 4 & 3) + 4 (e >> 6), col = e & 15)
    const int G = fb.G, glo = (G * slice) / NSL, ghi = (G * (slice + 1)) / NSL;
    const double* part = fb.partial + ((size_t)f * G * NPAIR + pair) * 256 + e;
    const size_t st = (size_t)NPAIR * 256;
    const unsigned long long* wm = fb.wmask + (size_t)f * G;
    double v[NLD];
    unsigned long long m[NLD];
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
        const int g = min(glo + u, G - 1);
        v[u] = __builtin_nontemporal_load(part + (size_t)g * st);
        m[u] = wm[g];
    }
    double a = 0.0;
#pragma unroll
    for (int u = 0; u < NLD; ++u) a += (glo + u < ghi && (pair >= 64 || ((m[u] >> (pair & 63)) & 1ull))) ? v[u] : 0.0;
    __shared__ double s_q[NSL - 1][EL];
    if (slice > 0) s_q[slice - 1][el] = a;
    __syncthreads();
    if (slice != 0) return;
#pragma unroll
    for (int i = 0; i < NSL - 1; ++i) a += s_q[i][el];
    int p = pair, ti = 0;
    while (p >= NT - ti) { p -= NT - ti; ++ti; }
    const int tj = ti + p;
    const int r = dm.tile_param[ti * 16 + ((e >> 4) & 3) + 4 * (e >> 6)], c = dm.tile_param[tj * 16 + (e & 15)];
    if (r >= 0 && c >= 0) {
        double* H = fb.Hraw + ((size_t)f * 2 + (1 - fb.ctl[f].cur_slot)) * HS * HS;
        H[(size_t)r * HS + c] = a;
        if (ti != tj) H[(size_t)c * HS + r] = a;
    }
}

typedef double d2v __attribute__((ext_vector_type(2)));

// Frame batches: the matched-point batches of a frame are dealt to its G evaluation workgroups as CONTIGUOUS ranges of equal
// estimated cost (a batch costs a fixed part for the row builder plus one unit per live tile pair of its matrix phase: hip
// batches touch ten pairs, torso batches three), so that the workgroups of a frame finish together - k_eval lasts as long as
// its busiest workgroup.  erange[g] = the first batch whose cost prefix reaches g/G of the frame's total.  Deterministic.
// s_pre: AVT_ERANGE_CAP + 1 ints of LDS scratch.
#define AVT_ERANGE_CAP 1024                          // batches a frame can have here (V <= 16384); beyond: equal counts
template <int NTH>
__device__ __forceinline__ void eval_ranges(const AvtDims& d, const FrameBuffers& fb, int f, int t, int* __restrict__ s_pre) {
    constexpr int CAP = AVT_ERANGE_CAP;
    __shared__ int s_wsum[NTH / 64];
    const int nb = (fb.ctl[f].M + AVT_EVAL_PTS - 1) / AVT_EVAL_PTS, G = fb.G;
    int* out = fb.erange + (size_t)f * AVT_ERANGE;
    if (nb > CAP) {
        for (int g = t; g <= G; g += NTH) out[g] = (int)(((long long)g * nb) / G);
        return;
    }
    // inclusive prefix sums of the batch costs: PER = ceil(CAP / NTH) consecutive batches per thread, wave scan, wave totals
    constexpr int PER = (CAP + NTH - 1) / NTH;
    int loc[PER], sum = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int b = t * PER + i;
        int c = 0;
        if (b < nb) {
            const int w = fb.bmask[(size_t)f * d.nb_max + b];
            const int tiles = __popc(w & 0xffff);
            c = 16 + (d.NT > 8 ? tiles * (tiles + 1) / 2 : __popc(w & 0xffffff));
        }
        sum += c; loc[i] = sum;
    }
    int incl = sum;
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) { const int v = __shfl_up(incl, sft, 64); if ((t & 63) >= sft) incl += v; }
    if ((t & 63) == 63) s_wsum[t >> 6] = incl;
    __syncthreads();
    int base = incl - sum;
    for (int w = 0; w < (t >> 6); ++w) base += s_wsum[w];
#pragma unroll
    for (int i = 0; i < PER; ++i) { const int b = t * PER + i; if (b < CAP) s_pre[b + 1] = base + loc[i]; }
    if (t == 0) s_pre[0] = 0;
    __syncthreads();
    const long long total = s_pre[min(nb, CAP)];
    for (int g = t; g <= G; g += NTH) {
        const long long want = (total * g) / G;            // first batch b with prefix(b) = s_pre[b] >= want
        int lo = 0, hi = nb;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_pre[mid] >= want) hi = mid; else lo = mid + 1; }
        out[g] = g == G ? nb : lo;
    }
}

// reciprocal off the slow path: v_rcp_f64 (~2^-26 relative) + one cubic Newton step (error e^3)
__device__ __forceinline__ double fast_rcp(double d) {
    const double r0 = __builtin_amdgcn_rcp(d);
    const double e = fma(-d, r0, 1.0);
    const double t2 = fma(e, e, e);
    return fma(r0, t2, r0);
}

// entry k of one of AvtSpecCtl's arrays by compares: an index the compiler does not know puts the whole snapshot of the structure into
// scratch memory, and every field the decision reads - next, n, valid[k] one after the other - becomes a dependent scratch round trip on
// the install path of every riding k_solve launch (104 bytes of scratch, three dependent loads in the builds up to round 5's first)
template <typename T>
__device__ __forceinline__ T spec_pick(int k, T a0, T a1, T a2, T a3) {      // (k past the end: the last entry, like the clamped index it replaces)
    return k >= 3 ? a3 : (k == 2 ? a2 : (k == 1 ? a1 : a0));
}

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// Back substitution L^T delta = y by one wave, four unknowns per step.  The factorisation left W = L diag(d) and the
// reciprocal pivots r; row P of W holds L^-1 rhs, so y_l = W(P,l) r_l and
//   delta_i = r_i ( W(P,i) - sum_{k>i} W(k,i) delta_k ).
// Lane l keeps W(P,l), W(P,l+64) and the running sums acc_l = sum_{k>i} W(k,l) delta_k in registers.  Per step the
// four values W(P,.) - acc of the block cross the wave by v_readlane, the 4x4 triangular system is solved by every lane
// alike, and each lane adds the block's contribution to its sums.
// Block (pivot block kb, row block bi >= kb) of W: square storage [kb][bi] (every block addressable, the never-written ones
// zeroed: the SMPL shape) or packed lower triangle (systems above 88 columns: the square does not fit the LDS).
template <bool TRI>
__device__ __forceinline__ int wblk(int kb, int bi, int NB) {
    return TRI ? kb * NB - (kb * (kb - 1)) / 2 + (bi - kb) : kb * NB + bi;
}

// Round 6: the 4x4 triangular solves are folded INTO THE FACTOR before the chain starts.  With M_kb the upper triangular matrix of a
// step (d = M u: d3 = r3 u3, d2 = r2 (u2 - w32 d3) .., a function of the diagonal block alone), a step's update of the running
// sums, acc_l += sum_k W(4kb+k, l) d_k, is acc_l += sum_j G_l[j] u_j with G = W^T M - four products per lane that do not wait for d.
// backsub_fold (every thread of the workgroup, one 4x4 block each) replaces W's block (column block lb, row block kb > lb) by
// -G in place ([column in block][j]: a lane's four coefficients of a step are 32 contiguous bytes), keeps row P of W (the
// forward-substituted right-hand side, which lives in one of the overwritten block rows) in s_wp, and leaves the M's in s_M.
// backsub_chain (wave 0) then carries v_l = W(P,l) - acc_l: per step 8 v_readlane + 5 operations on the chain instead of ~40
// instructions, the unknowns themselves (delta = M u) are formed for all blocks at once behind the last step.
// Item i of the fold: block (column block lb, row block kb > lb) for i < nblk, the matrix M of block i - nblk behind them.  Consecutive
// items walk DOWN a block column (same lb, consecutive kb): their blocks are 144 bytes apart, 16 lanes cover the 64 banks with their 16-byte
// accesses (along a block row, 3168 bytes apart, the same accesses collide eight-fold).  Packed lb | kb << 8; a function of the thread index and
// the system's size alone, so k_solve computes its first item at kernel start, off the chain (45 instructions with the square root).
__device__ __forceinline__ int backsub_fold_item(int i, int NB) {
    const int nblk = (NB * (NB - 1)) >> 1;
    if (i >= nblk) return (i - nblk) | ((i - nblk) << 8);
    const int j = nblk - 1 - i;
    int k2 = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)j)) * 0.5f);
    k2 -= ((k2 * (k2 - 1)) >> 1) > j ? 1 : 0;
    k2 += (((k2 + 1) * k2) >> 1) <= j ? 1 : 0;
    return (NB - 1 - k2) | ((NB - 1 - (j - ((k2 * (k2 - 1)) >> 1))) << 8);
}

// one item of the fold: block (lb, kb) -> -G in place, or (is_M) the matrix N = -M of block kb into s_M
__device__ __forceinline__ void backsub_fold_block(double* __restrict__ Lblk, const double* __restrict__ s_R, int NB, int Pb, int Pr, int lb, int kb, bool is_M,
                                                   double* __restrict__ s_M, double* __restrict__ s_wp) {
    const d2v* Wd = (const d2v*)(Lblk + ((size_t)kb * NB + kb) * 18);
    d2v* blk = (d2v*)(Lblk + ((size_t)lb * NB + kb) * 18);
    const d2v a = Wd[2], bq = Wd[4], c = Wd[6], e = Wd[7];
    const d2v* Rq = (const d2v*)(s_R + 4 * kb);
    const d2v ra = Rq[0], rb = Rq[1];
    d2v w[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) w[q] = blk[q];                  // W[k][c] = w[2 k + c / 2].(c % 2)   (is_M: the diagonal block once more)
    // N = -M: d = M u with d3 = r3 u3, d2 = r2 (u2 - w32 d3), d1 = r1 (u1 - w21 d2 - w31 d3), d0 = r0 (u0 - w10 d1 - w20 d2 - w30 d3)
    const double w10 = a.x, w20 = bq.x, w21 = bq.y, w30 = c.x, w31 = c.y, w32 = e.x;
    const double r0 = ra.x, r1 = ra.y, r2 = rb.x, r3 = rb.y;
    const double N33 = -r3;
    const double N22 = -r2, N23 = -r2 * (w32 * N33);
    const double N11 = -r1, N12 = -r1 * (w21 * N22), N13 = -r1 * fma(w21, N23, w31 * N33);
    const double N00 = -r0, N01 = -r0 * (w10 * N11), N02 = -r0 * fma(w10, N12, w20 * N22), N03 = -r0 * fma(w10, N13, fma(w20, N23, w30 * N33));
    if (kb == Pb) {      // row P of W, columns 4 lb ..
        d2v* o = (d2v*)(s_wp + 4 * lb);
        o[0] = Pr == 0 ? w[0] : (Pr == 1 ? w[2] : (Pr == 2 ? w[4] : w[6]));      // (by compares: a run-time index would put w into scratch memory)
        o[1] = Pr == 0 ? w[1] : (Pr == 1 ? w[3] : (Pr == 2 ? w[5] : w[7]));
    }
    if (is_M) {          // N itself, rows padded to four entries: lane l of the chain reads row l & 3 of block l >> 2
        d2v* o = (d2v*)(s_M + 16 * kb);
        o[0] = (d2v){N00, N01}; o[1] = (d2v){N02, N03}; o[2] = (d2v){0.0, N11}; o[3] = (d2v){N12, N13};
        o[4] = (d2v){0.0, 0.0}; o[5] = (d2v){N22, N23}; o[6] = (d2v){0.0, 0.0}; o[7] = (d2v){0.0, N33};
        return;
    }
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
        const double W0 = (cc & 1) ? w[cc >> 1].y : w[cc >> 1].x, W1 = (cc & 1) ? w[2 + (cc >> 1)].y : w[2 + (cc >> 1)].x;
        const double W2 = (cc & 1) ? w[4 + (cc >> 1)].y : w[4 + (cc >> 1)].x, W3 = (cc & 1) ? w[6 + (cc >> 1)].y : w[6 + (cc >> 1)].x;
        const double G0 = W0 * N00;
        const double G1 = fma(W0, N01, W1 * N11);
        const double G2 = fma(W0, N02, fma(W1, N12, W2 * N22));
        const double G3 = fma(W0, N03, W1 * N13) + fma(W2, N23, W3 * N33);
        blk[2 * cc] = (d2v){G0, G1}; blk[2 * cc + 1] = (d2v){G2, G3};
    }
}

// the whole fold behind the factorisation (systems of any size): every thread one item, the first one computed at kernel start
template <int NTH>
__device__ __forceinline__ void backsub_fold(double* __restrict__ Lblk, const double* __restrict__ s_R, int NB, int P, int t, int item0,
                                             double* __restrict__ s_M, double* __restrict__ s_wp) {
    const int nblk = (NB * (NB - 1)) >> 1;
    for (int i = t; i < nblk + NB; i += NTH) {
        const int item = i == t ? item0 : backsub_fold_item(i, NB);
        backsub_fold_block(Lblk, s_R, NB, P >> 2, P & 3, item & 0xff, item >> 8, i >= nblk, s_M, s_wp);
    }
}

template <int CTRL>
__device__ __forceinline__ double quad_dpp(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

template <int NBC>
__device__ __forceinline__ void backsub_chain(const double* __restrict__ Lblk, int NBrt, int P, int t,
                                              double* s_delta, const double* __restrict__ s_M, const double* s_wp) {      // (s_wp may be s_delta: read first, written last)
    const int NB = NBC > 0 ? NBC : NBrt;
    double v0 = (t < P) ? s_wp[t] : 0.0;                 // W(P,l) - acc_l, columns l = t and t + 64
    double v1 = (t + 64 < P) ? s_wp[t + 64] : 0.0;
    // row l & 3 of -M of my columns' blocks, for the unknowns themselves behind the chain (requested now, read there)
    const d2v* Nq0 = (const d2v*)(s_M + 16 * (t >> 2) + 4 * (t & 3));
    const d2v* Nq1 = (const d2v*)(s_M + 16 * (((t + 64) >> 2) < NB ? ((t + 64) >> 2) : 0) + 4 * (t & 3));
    const d2v n0a = Nq0[0], n0b = Nq0[1], n1a = Nq1[0], n1b = Nq1[1];
    // my column's coefficients of step kb: block [t >> 2][kb], entries (t & 3) * 4 + j.  Blocks above the diagonal were zeroed at kernel
    // start (W(k,l) = 0 for l >= k); the diagonal blocks still hold W - what a lane of the step's own block adds to its v is never read
    // again (its unknown was taken at the head of the step).
    const double* col0 = Lblk + (size_t)(t >> 2) * NB * 18 + (t & 3) * 4;
    const double* col1 = Lblk + (size_t)((t + 64) >> 2) * NB * 18 + (t & 3) * 4;
    constexpr int DEPTH = NBC > 0 ? 3 : 1;                            // runtime NB: no unrolling, no ring
    d2v g0[DEPTH][2], g1[DEPTH][2];
    auto fetch = [&](int kb, int sl) {
        const d2v* q0 = (const d2v*)(col0 + (size_t)kb * 18);
        g0[sl][0] = q0[0]; g0[sl][1] = q0[1];
        if (4 * kb > 64) { const d2v* q1 = (const d2v*)(col1 + (size_t)kb * 18); g1[sl][0] = q1[0]; g1[sl][1] = q1[1]; }
        else { g1[sl][0] = (d2v){0.0, 0.0}; g1[sl][1] = (d2v){0.0, 0.0}; }
    };
#pragma unroll
    for (int i = 0; i < DEPTH; ++i)
        if (NB - 1 - i >= 0) fetch(NB - 1 - i, (NB - 1 - i) % DEPTH);
    double uf0 = 0.0, uf1 = 0.0;                                       // my unknowns' u, taken at their steps
#pragma unroll
    for (int kb = NB - 1; kb >= 0; --kb) {
        const int base = 4 * kb, sl = kb % DEPTH;
        const d2v ga = g0[sl][0], gb = g0[sl][1], ha = g1[sl][0], hb = g1[sl][1];
        if (kb - DEPTH >= 0) fetch(kb - DEPTH, sl);
        const double u = (base >= 64) ? v1 : v0;
        const bool mine = (t >> 2) == (kb & 15);
        if (base >= 64) uf1 = mine ? u : uf1; else uf0 = mine ? u : uf0;
        const double u0 = readlane_f64(u, base & 63), u1 = readlane_f64(u, (base + 1) & 63);
        const double u2 = readlane_f64(u, (base + 2) & 63), u3 = readlane_f64(u, (base + 3) & 63);
        v0 = fma(ga.x, u0, fma(gb.x, u2, v0)) + fma(ga.y, u1, gb.y * u3);
        if (NBC == 0 || 4 * kb > 64) v1 = fma(ha.x, u0, fma(hb.x, u2, v1)) + fma(ha.y, u1, hb.y * u3);
    }
    // delta_l = sum_j M[l & 3][j] u_(quad's j): the quad's four u's by quad_perm broadcasts, no LDS round trip
    {
        const double q0 = quad_dpp<0x00>(uf0), q1 = quad_dpp<0x55>(uf0), q2 = quad_dpp<0xAA>(uf0), q3 = quad_dpp<0xFF>(uf0);
        const double dl = fma(n0a.x, q0, n0a.y * q1) + fma(n0b.x, q2, n0b.y * q3);
        s_delta[t] = -dl;
        const double p0 = quad_dpp<0x00>(uf1), p1 = quad_dpp<0x55>(uf1), p2 = quad_dpp<0xAA>(uf1), p3 = quad_dpp<0xFF>(uf1);
        const double dh = fma(n1a.x, p0, n1a.y * p1) + fma(n1b.x, p2, n1b.y * p3);
        if (t + 64 < 4 * NB) s_delta[t + 64] = -dh;
    }
}

// The same back substitution for the packed-triangle storage and up to 192 unknowns (three columns per lane of the wave).
// W(k,l) = 0 for l >= k by a select instead of by stored zeros.  Everything a step reads from the factor - the diagonal block,
// the reciprocal pivots, this lane's three column blocks - has an address that does not depend on the unknowns, so it is
// requested one step ahead; a step is then readlanes, the 4-unknown chain and 12 FMAs (61 k -> 40 k clocks for 43 steps
// on the 52-joint model, tools/big_model_phase_probe.py; what is left is the instruction issue of one wave, ~100 per step).
__device__ __forceinline__ void backsub_tri(const double* __restrict__ Lblk, const double* __restrict__ s_R, int NB, int P, int t,
                                            double* __restrict__ s_delta) {
    auto blockp = [&](int kb, int bi) { return Lblk + (size_t)wblk<true>(kb, bi, NB) * 18; };
    double wp[3], acc[3] = {0.0, 0.0, 0.0};
    int cbk[3], li[3];
    bool live[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int l = t + 64 * c;
        live[c] = l < P; cbk[c] = min(l, P - 1) >> 2; li[c] = l & 3;
        wp[c] = live[c] ? blockp(l >> 2, P >> 2)[(P & 3) * 4 + (l & 3)] : 0.0;
    }
    struct Step { d2v a, bq, cq, e, r01, r23; double bp[3][4]; };
    auto fetch = [&](int kb) {
        Step s;
        const d2v* Wd = (const d2v*)blockp(kb, kb);
        s.a = Wd[2]; s.bq = Wd[4]; s.cq = Wd[6]; s.e = Wd[7];
        const d2v* Rq = (const d2v*)(s_R + 4 * kb);
        s.r01 = Rq[0]; s.r23 = Rq[1];
#pragma unroll
        for (int c = 0; c < 3; ++c) {      // my column l lives in pivot block cb; rows of block kb matter if kb >= cb (kb == cb: strictly lower part)
            const double* bp = blockp(cbk[c], max(kb, cbk[c])) + li[c];
            s.bp[c][0] = bp[0]; s.bp[c][1] = bp[4]; s.bp[c][2] = bp[8]; s.bp[c][3] = bp[12];
        }
        return s;
    };
    // (two named buffers swapping roles instead of the copy below cost more than the copy: the kernel runs at 128 registers per lane
    // and the second buffer spilled - 66 k clocks against 40 k)
    Step nx = fetch(NB - 1);
    for (int kb = NB - 1; kb >= 0; --kb) {
        const Step cu = nx;
        if (kb > 0) nx = fetch(kb - 1);
        const int base = 4 * kb;
        const double w10 = cu.a.x, w20 = cu.bq.x, w21 = cu.bq.y, w30 = cu.cq.x, w31 = cu.cq.y, w32 = cu.e.x;
        const double r0 = cu.r01.x, r1 = cu.r01.y, r2 = cu.r23.x, r3 = cu.r23.y;
        const int own = base >> 6;                                  // which of my columns holds unknown `base`
        const double u = (own == 0 ? wp[0] - acc[0] : (own == 1 ? wp[1] - acc[1] : wp[2] - acc[2]));
        const double u0 = readlane_f64(u, base & 63), u1 = readlane_f64(u, (base + 1) & 63);
        const double u2 = readlane_f64(u, (base + 2) & 63), u3 = readlane_f64(u, (base + 3) & 63);
        const double d3 = r3 * u3;
        const double d2 = r2 * fma(-w32, d3, u2);
        const double d1 = r1 * fma(-w21, d2, fma(-w31, d3, u1));
        const double d0 = r0 * fma(-w10, d1, fma(-w20, d2, fma(-w30, d3, u0)));
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            // (rows of my own pivot block would only change sums whose unknowns this step has just finished; lanes past P are never read)
            const double full = fma(cu.bp[c][0], d0, cu.bp[c][1] * d1) + fma(cu.bp[c][2], d2, cu.bp[c][3] * d3);
            acc[c] += kb > cbk[c] ? full : 0.0;
        }
        if (t == 0) { d2v* o = (d2v*)(s_delta + base); o[0] = (d2v){d0, d1}; o[1] = (d2v){d2, d3}; }
    }
}

// -------------------------------------------------------------------------------------------------
// LDL^T with the trailing matrix in MFMA accumulators (the 256-thread solve, systems of up to 95 columns; measured in
// isolation in tools/ubench/ldlt_mfma.hip: 1620 clocks per 4-pivot round against 1900 for the register-blocked rounds).
// The lower triangle of the bordered system lives as 21 tiles of 16x16 in the accumulators of the 4 waves: wave w holds tile
// row rA = 5 - w (columns 0..rA) and, for w >= 2, tile row rB = w - 2 (columns 0..rB).  Two barriers per round:
//   phase 1 (threads 0..95, one matrix row each): factor the 4x4 diagonal block of the published panel, W row = raw L^-T,
//            stored in the layout the back substitution reads (Lblk[kb][row block][18]);
//   phase 2 (all waves): A = W(:,k) and B = -W(:,k)/d_k fragments straight from Lblk / s_R, one v_mfma_f64_16x16x4_f64 per
//            live tile (next panel's block column first), publish the next panel's 4 columns from the accumulators.
// -------------------------------------------------------------------------------------------------
typedef double v4f64 __attribute__((ext_vector_type(4)));
#ifdef AVT_TIMING
__device__ long long g_mf_phase[8];     // waves 0 and 3, lane 0, workgroup 0: clocks at barrier A, row phase, barrier B, matrix phase (summed over rounds and launches)
extern "C" void avt_debug_mf_phases(long long* out8, int reset) {
    if (out8) (void)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_mf_phase), sizeof(long long) * 8);
    if (reset) { long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_mf_phase), z, sizeof z); }
}
#ifdef AVT_TIMING_ROUNDS     // (the per-round probes are global read-modify-writes on the factorisation's critical path: they double what they measure, so they are a build of their own)
#define MFP(i) do { if ((t & 63) == 0 && (W == 0 || W == 3) && blockIdx.x == 0) { const long long _n = clock64(); g_mf_phase[(W == 0 ? 0 : 4) + (i)] += _n - _tl; _tl = _n; } } while (0)
#else
#define MFP(i) do {} while (0)
#endif
#else
#define MFP(i) do {} while (0)
#endif
#define MF_PB_STRIDE 4
#define MF_PB_DOUBLES (96 * MF_PB_STRIDE)

template <int W, int CBN>
__device__ __forceinline__ void mf_publish(const v4f64 (&accA)[6], const v4f64 (&accB)[2], int jq, double* __restrict__ PB, int ln) {
    constexpr int rA = 5 - W, rB = W - 2;
    const int k = ln >> 4, c16 = ln & 15;
    if ((c16 >> 2) == jq) {
        if constexpr (CBN <= rA) {
#pragma unroll
            for (int v = 0; v < 4; ++v) PB[(16 * rA + 4 * v + k) * MF_PB_STRIDE + (ln & 3)] = accA[CBN][v];
        }
        if constexpr (CBN < 2 && CBN <= rB) {
#pragma unroll
            for (int v = 0; v < 4; ++v) PB[(16 * rB + 4 * v + k) * MF_PB_STRIDE + (ln & 3)] = accB[CBN < 2 ? CBN : 0][v];
        }
    }
}

// one round = 4 pivots (columns 16 CB + 4 jq ..); CBN = block column of the next panel = first block column still live
template <int W, int CB, int CBN>
__device__ __forceinline__ bool mf_round(v4f64 (&accA)[6], v4f64 (&accB)[2], int jq, double* __restrict__ PB,
                                           double* __restrict__ Lblk, double* __restrict__ s_R, int* __restrict__ s_fail, int P, int NB, int NR, int t) {
    constexpr int rA = 5 - W, rB = W - 2;
    const int ln = t & 63, k = ln >> 4, c16 = ln & 15, kb = 4 * CB + jq;
#ifdef AVT_TIMING
    long long _tl = clock64();
#endif
    __syncthreads();                                         // the panel of this round is published
    MFP(0);
    if (W < 2 && t < 96) {
        const d2v* PB2 = (const d2v*)PB;
        const d2v q0 = PB2[(4 * kb) * 2], q1 = PB2[(4 * kb + 1) * 2], q2a = PB2[(4 * kb + 2) * 2], q2b = PB2[(4 * kb + 2) * 2 + 1];
        const d2v q3a = PB2[(4 * kb + 3) * 2], q3b = PB2[(4 * kb + 3) * 2 + 1];
        const d2v s01 = PB2[t * 2], s23 = PB2[t * 2 + 1];
        const double D00 = q0.x, D10 = q1.x;
        double D11 = q1.y, D20 = q2a.x, D21 = q2a.y, D22 = q2b.x, D30 = q3a.x, D31 = q3a.y, D32 = q3b.x, D33 = q3b.y;
        const double r0 = fast_rcp(D00), l10 = D10 * r0, l20 = D20 * r0, l30 = D30 * r0;
        D11 = fma(-l10, D10, D11); D21 = fma(-l20, D10, D21); D31 = fma(-l30, D10, D31);
        D22 = fma(-l20, D20, D22); D32 = fma(-l30, D20, D32); D33 = fma(-l30, D30, D33);
        const double r1 = fast_rcp(D11), l21 = D21 * r1, l31 = D31 * r1;
        D22 = fma(-l21, D21, D22); D32 = fma(-l31, D21, D32); D33 = fma(-l31, D31, D33);
        const double r2 = fast_rcp(D22), l32 = D32 * r2;
        const double w0 = s01.x, w1 = fma(-w0, l10, s01.y), w2 = fma(-w1, l21, fma(-w0, l20, s23.x));
        const double w3 = fma(-w2, l32, fma(-w1, l31, fma(-w0, l30, s23.y)));
        const int rr = t - 4 * kb;     // row inside the trailing part; the diagonal block keeps its strictly lower part
        if (rr >= 0 && (t >> 2) < NB) {     // (rows past the matrix have no block)
            d2v* Wo = (d2v*)(Lblk + ((size_t)kb * NB + (t >> 2)) * 18 + (t & 3) * 4);
            // (the four rows of the diagonal block leave what they computed in its upper triangle too: nobody reads it as part of the factor - the
            // matrix phase and the back substitution's chain only add it to rows and columns that are already final, the fold reads the strictly
            // lower part by name - and zeroing it was four compares and eight selects per row and round)
            Wo[0] = (d2v){w0, w1};
            Wo[1] = (d2v){w2, w3};
        }
    }
    // The reciprocal pivots (what the matrix phase and the back substitution read) and the verdict on the pivots are the business of ONE lane of
    // wave 2, which has no rows: on thread 0 they were ~25 instructions - the fourth reciprocal among them, which no row needs - behind its
    // row's stores, on the wave every barrier of the factorisation waits for.  Same panel, same operations, same bits as the row threads'.
    if (W == 2 && t == 128) {
        const d2v* PB2 = (const d2v*)PB;
        const d2v q0 = PB2[(4 * kb) * 2], q1 = PB2[(4 * kb + 1) * 2], q2a = PB2[(4 * kb + 2) * 2], q2b = PB2[(4 * kb + 2) * 2 + 1];
        const d2v q3a = PB2[(4 * kb + 3) * 2], q3b = PB2[(4 * kb + 3) * 2 + 1];
        const double D00 = q0.x, D10 = q1.x;
        double D11 = q1.y, D20 = q2a.x, D21 = q2a.y, D22 = q2b.x, D30 = q3a.x, D31 = q3a.y, D32 = q3b.x, D33 = q3b.y;
        const double r0 = fast_rcp(D00), l10 = D10 * r0, l20 = D20 * r0, l30 = D30 * r0;
        D11 = fma(-l10, D10, D11); D21 = fma(-l20, D10, D21); D31 = fma(-l30, D10, D31);
        D22 = fma(-l20, D20, D22); D32 = fma(-l30, D20, D32); D33 = fma(-l30, D30, D33);
        const double r1 = fast_rcp(D11), l21 = D21 * r1, l31 = D31 * r1;
        D22 = fma(-l21, D21, D22); D32 = fma(-l31, D21, D32); D33 = fma(-l31, D31, D33);
        const double r2 = fast_rcp(D22), l32 = D32 * r2;
        D33 = fma(-l32, D32, D33);
        const double r3 = fast_rcp(D33);
        const bool real1 = 4 * kb + 1 < P, real2 = 4 * kb + 2 < P, real3 = 4 * kb + 3 < P;
        const bool bad = !(D00 > 0.0) | (real1 & !(D11 > 0.0)) | (real2 & !(D22 > 0.0)) | (real3 & !(D33 > 0.0));
        d2v* Ro = (d2v*)(s_R + 4 * kb);
        Ro[0] = (d2v){r0, real1 ? r1 : 0.0}; Ro[1] = (d2v){real2 ? r2 : 0.0, real3 ? r3 : 0.0};
        if (bad) *s_fail = 1;
    }
    MFP(1);
    __syncthreads();                                         // W rows, reciprocal pivots and the failure flag are visible
    MFP(2);
    // ---- fragments: lane (c16, k) holds row 16 b + c16, pivot k of the round
    const double* Wk = Lblk + (size_t)kb * NB * 18 + (c16 >> 2) * 18 + (c16 & 3) * 4 + k;
    double fw[6];
#pragma unroll
    for (int c = CBN; c <= rA; ++c) fw[c] = (4 * c + (c16 >> 2) < NB) ? Wk[c * 4 * 18] : 0.0;     // (rows past the matrix: zero)
    constexpr bool useB = rB >= 0 && CBN <= rB;
    double fAB = 0.0;
    if constexpr (useB) fAB = (4 * (rB > 0 ? rB : 0) + (c16 >> 2) < NB) ? Wk[(rB > 0 ? rB : 0) * 4 * 18] : 0.0;
    const double rk = s_R[4 * kb + k];
    // (a refused pivot - *s_fail, set by the row phase - is looked at ONCE, behind the last round: read here, the flag's LDS round trip and a branch sat
    // in front of every round's fragment loads; a factorisation that goes on past a refused pivot computes garbage from fixed addresses, nothing else)
    if (kb + 1 >= NR) return true;
    if constexpr (CBN <= rA) {
        double fb[6];
#pragma unroll
        for (int c = CBN; c <= rA; ++c) fb[c] = fw[c] * -rk;
        const double fA = fw[rA];
        accA[CBN] = __builtin_amdgcn_mfma_f64_16x16x4f64(fA, fb[CBN], accA[CBN], 0, 0, 0);
        if constexpr (useB) {
#pragma unroll
            for (int c = CBN; c <= rB; ++c) accB[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(fAB, fb[c], accB[c], 0, 0, 0);
        }
#pragma unroll
        for (int c = CBN + 1; c <= rA; ++c) accA[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(fA, fb[c], accA[c], 0, 0, 0);
    }
    mf_publish<W, CBN>(accA, accB, (jq + 1) & 3, PB, ln);
    MFP(3);
    return true;
}

template <int W, int CB>
__device__ __forceinline__ bool mf_block_column(v4f64 (&accA)[6], v4f64 (&accB)[2], double* __restrict__ PB,
                                                  double* __restrict__ Lblk, double* __restrict__ s_R, int* s_fail, int P, int NB, int NR, int t) {
#pragma unroll
    for (int jq = 0; jq < 3; ++jq) {      // (unrolled: as a loop the accumulators are copied from one turn's registers to the next's, 8 to 32 v_accvgpr_mov per round)
        if (4 * CB + jq >= NR) return true;
        if (!mf_round<W, CB, CB>(accA, accB, jq, PB, Lblk, s_R, s_fail, P, NB, NR, t)) return false;
    }
    if (4 * CB + 3 >= NR) return true;
    return mf_round<W, CB, (CB < 5 ? CB + 1 : 5)>(accA, accB, 3, PB, Lblk, s_R, s_fail, P, NB, NR, t);
}

template <int W>
__device__ __forceinline__ bool mf_rounds(v4f64 (&accA)[6], v4f64 (&accB)[2], double* __restrict__ PB,
                                            double* __restrict__ Lblk, double* __restrict__ s_R, int* s_fail, int P, int NB, int t) {
    const int NR = (P + 3) >> 2;          // rounds: pivots 0..P-1
    mf_publish<W, 0>(accA, accB, 0, PB, t & 63);
    return mf_block_column<W, 0>(accA, accB, PB, Lblk, s_R, s_fail, P, NB, NR, t) && mf_block_column<W, 1>(accA, accB, PB, Lblk, s_R, s_fail, P, NB, NR, t) &&
           mf_block_column<W, 2>(accA, accB, PB, Lblk, s_R, s_fail, P, NB, NR, t) && mf_block_column<W, 3>(accA, accB, PB, Lblk, s_R, s_fail, P, NB, NR, t) &&
           mf_block_column<W, 4>(accA, accB, PB, Lblk, s_R, s_fail, P, NB, NR, t) && mf_block_column<W, 5>(accA, accB, PB, Lblk, s_R, s_fail, P, NB, NR, t);
}


// -------------------------------------------------------------------------------------------------
// The same factorisation for the 1024-thread shape (systems of 89..180 columns): a T x T tile grid, T <= 12, 16 waves.  Tile
// tau = rb (rb + 1) / 2 + cb of the lower triangle belongs to wave tau mod 16, as its accumulator slot tau / 16 (at most
// MFG_SLOTS = 5); which tiles a slot holds is wave-uniform run-time data, the slots themselves are compile-time registers.
// Rounds as above: row phase (threads 0 .. 16 T - 1, one matrix row each), matrix phase (one matrix instruction per live
// slot), publish.  The factor is the packed lower triangle (wblk<true>): blocks above the diagonal do not exist, so fragment
// rows above the panel read as zero by a select.
// -------------------------------------------------------------------------------------------------
#define MFG_SLOTS 5
#define MFG_PB_DOUBLES (192 * MF_PB_STRIDE)
struct MfgSlots { int rb[MFG_SLOTS], cb[MFG_SLOTS]; };      // rb = -1: empty slot

__device__ __forceinline__ MfgSlots mfg_slots(int wv, int T) {
    MfgSlots S;
#pragma unroll
    for (int s = 0; s < MFG_SLOTS; ++s) {
        int tau = wv + 16 * s, rb = 0;
        if (tau < T * (T + 1) / 2) {
            while (tau > rb) { tau -= rb + 1; ++rb; }
            S.rb[s] = rb; S.cb[s] = tau;
        } else { S.rb[s] = -1; S.cb[s] = -1; }
    }
    return S;
}

__device__ __forceinline__ bool mfg_rounds(v4f64 (&acc)[MFG_SLOTS], const MfgSlots& S, double* __restrict__ PB, double* __restrict__ Lblk,
                                           double* __restrict__ s_R, int* __restrict__ s_fail, int P, int NB, int T, int t) {
    const int NR = (P + 3) >> 2;          // rounds: pivots 0..P-1
    const int ln = t & 63, k = ln >> 4, c16 = ln & 15;
    // the panel of a round: columns 4 jq .. 4 jq + 3 of the tiles in tile column cbn, rows 16 rb + 4 v + k
    auto publish = [&](int cbn, int jq) {
        if ((c16 >> 2) == jq) {
#pragma unroll
            for (int s = 0; s < MFG_SLOTS; ++s)
                if (S.cb[s] == cbn) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) PB[(16 * S.rb[s] + 4 * v + k) * MF_PB_STRIDE + (ln & 3)] = acc[s][v];
                }
        }
    };
    publish(0, 0);
#pragma unroll 1
    for (int kb = 0; kb < NR; ++kb) {
        const int cb = kb >> 2, jq = kb & 3;
        const int cbn = jq == 3 ? cb + 1 : cb;               // tile column of the next panel = first tile column still live
        __syncthreads();                                     // the panel of this round is published
        if (t < 16 * T) {
            const d2v* PB2 = (const d2v*)PB;
            const d2v q0 = PB2[(4 * kb) * 2], q1 = PB2[(4 * kb + 1) * 2], q2a = PB2[(4 * kb + 2) * 2], q2b = PB2[(4 * kb + 2) * 2 + 1];
            const d2v q3a = PB2[(4 * kb + 3) * 2], q3b = PB2[(4 * kb + 3) * 2 + 1];
            const d2v s01 = PB2[t * 2], s23 = PB2[t * 2 + 1];
            const double D00 = q0.x, D10 = q1.x;
            double D11 = q1.y, D20 = q2a.x, D21 = q2a.y, D22 = q2b.x, D30 = q3a.x, D31 = q3a.y, D32 = q3b.x, D33 = q3b.y;
            const double r0 = fast_rcp(D00), l10 = D10 * r0, l20 = D20 * r0, l30 = D30 * r0;
            D11 = fma(-l10, D10, D11); D21 = fma(-l20, D10, D21); D31 = fma(-l30, D10, D31);
            D22 = fma(-l20, D20, D22); D32 = fma(-l30, D20, D32); D33 = fma(-l30, D30, D33);
            const double r1 = fast_rcp(D11), l21 = D21 * r1, l31 = D31 * r1;
            D22 = fma(-l21, D21, D22); D32 = fma(-l31, D21, D32); D33 = fma(-l31, D31, D33);
            const double r2 = fast_rcp(D22), l32 = D32 * r2;
            D33 = fma(-l32, D32, D33);
            const double r3 = fast_rcp(D33);
            const bool real1 = 4 * kb + 1 < P, real2 = 4 * kb + 2 < P, real3 = 4 * kb + 3 < P;
            const bool bad = !(D00 > 0.0) | (real1 & !(D11 > 0.0)) | (real2 & !(D22 > 0.0)) | (real3 & !(D33 > 0.0));
            const double w0 = s01.x, w1 = fma(-w0, l10, s01.y), w2 = fma(-w1, l21, fma(-w0, l20, s23.x));
            const double w3 = fma(-w2, l32, fma(-w1, l31, fma(-w0, l30, s23.y)));
            const int rr = t - 4 * kb;     // row inside the trailing part; the diagonal block keeps its strictly lower part
            if (rr >= 0 && (t >> 2) < NB) {
                d2v* Wo = (d2v*)(Lblk + (size_t)wblk<true>(kb, t >> 2, NB) * 18 + (t & 3) * 4);
                Wo[0] = (d2v){rr < 1 ? 0.0 : w0, rr < 2 ? 0.0 : w1};
                Wo[1] = (d2v){rr < 3 ? 0.0 : w2, rr < 4 ? 0.0 : w3};
            }
            if (t == 0) {
                d2v* Ro = (d2v*)(s_R + 4 * kb);
                Ro[0] = (d2v){r0, real1 ? r1 : 0.0}; Ro[1] = (d2v){real2 ? r2 : 0.0, real3 ? r3 : 0.0};
                if (bad) *s_fail = 1;
            }
        }
        __syncthreads();                                     // W rows, reciprocal pivots and the failure flag are visible
        if (*s_fail) return false;
        if (kb + 1 >= NR) return true;
        // fragments: lane (c16, k) holds row 16 b + c16, pivot k of the round; A = W(:,k), B = -W(:,k) / d_k
        const double nrk = -s_R[4 * kb + k];
        auto frag = [&](int r) {
            const int rowblk = 4 * r + (c16 >> 2);
            const bool there = rowblk >= kb && rowblk < NB;          // (above the panel / past the matrix: zero)
            return there ? Lblk[(size_t)wblk<true>(kb, there ? rowblk : kb, NB) * 18 + (c16 & 3) * 4 + k] : 0.0;
        };
#pragma unroll
        for (int s = 0; s < MFG_SLOTS; ++s)
            if (S.cb[s] >= cbn)                                      // wave-uniform; empty slots have cb = -1
                acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(frag(S.rb[s]), frag(S.cb[s]) * nrk, acc[s], 0, 0, 0);
        publish(cbn, (jq + 1) & 3);
    }
    return true;
}

// -------------------------------------------------------------------------------------------------
// Few frames: the reduction rides in k_solve's own launch (RIDE).  Workgroups 1 .. 4 NPAIR of a frame are the reduction - four
// 256-thread workgroups per tile pair, each a strip of 64 tile elements, its four waves each summing a quarter of the G partial
// tiles in ascending order (<= 32 loads per lane in flight), the quarter sums added in order through LDS - and hand the system
// to the frame's solver (workgroup 0) INSIDE the launch: the entries leave as agent-scope write-through stores, the wave waits
// for its own stores, one agent-scope atomic increment per workgroup; the solver spins (bounded) on the count and reads its
// entries with agent-scope loads (no cache-wide fence on either side: tools/ubench/hop.hip, 0.8-1.1 us per hand-over against a
// kernel of its own at >= 4.6 us plus a 1.7 us boundary).  Every kernel of a one-frame chain pays ~4.6 us before its first
// and after its last instruction whatever it does, so a GN iteration is two launches instead of three.
// Deadlock-free for the shapes that use it (<= 6 frames): the waiting workgroups are one per frame, the producers wait for nobody.
// -------------------------------------------------------------------------------------------------
__device__ __forceinline__ void st_agent(double* p, double v) {
    __hip_atomic_store((unsigned long long*)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_agent(const double* p) {
    return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// The accept test of the FIRST queued speculative step, taken ahead (one workgroup per frame in the riding k_solve launch, beside the
// strips).  k_eval's spec-cost workgroups left one partial tile per evaluation workgroup: the element that holds sum c |r|^2 is summed
// over the workgroups by the operations reduce_ride_block below performs for that element of the trial point's system - same slices, same
// order, same written-mask rule - and the step's objective is formed from it term by term as k_solve forms the trial point's (cost
// constant, GMM score of the minimising component in ascending order with strict '<', shape prior), so that the test gives what the test
// on the fully evaluated point would give, bit for bit.  Published (agent scope, read by every solver role of this launch behind its
// wait): 1.0 = the step FAILS the test against the current point's cost AND may be folded - it exists, is valid, and the solve launch
// takes test number seq - 1 + ahead with one more still leaving the LAST test of the ICP iteration (number max_iters) to the k_lbs
// launch; 0.0 otherwise.  Inputs: the snapshot k_eval left (what every solver role decides on).  Counts itself in like the strips.
// layout of FrameBuffers::ride_ctr: the workgroups that have delivered since k_finalize cleared the word | the launch the fold verdict is for | idle
#define AVT_RIDE_COUNT_MASK 0xffffu
#define AVT_RIDE_SEQ_SHIFT 16
#define AVT_RIDE_SEQ_MASK 0x7fu
#define AVT_RIDE_IDLE 0x80000000u
template <int STRIPS>
__device__ __forceinline__ void reduce_spec_cost(const DeviceModel& dm, const FrameBuffers& fb, int f, int slot, double* s_q) {
    constexpr int EL = 256 / STRIPS, NSL = 256 / EL, NLD = AVT_G_MAX / NSL;
    const AvtDims& d = dm.d;
    const int t = threadIdx.x, el = t % EL, slice = t / EL;
    const bool have = f < fb.spec_frames;
    const AvtSolveSnap& sn = fb.snap[f];
    if (sn.ctl.try_valid == AVT_TRY_DONE) return;      // the frame met the stopping rule: no solver role of this launch waits for this count
    const int spn = sn.sp.next, s = min(spn + slot, AVT_MAX_SPEC - 1);      // slot 0 = the first step still in the queue when k_eval ran
    const int pair = d.res_pair, strip = d.res_elem / EL;
    const int e = strip * EL + el;
    const int G = fb.G, glo = (G * slice) / NSL, ghi = (G * (slice + 1)) / NSL, ng = ghi - glo;
    const size_t blk = ((size_t)(have ? f : 0) * AVT_MAX_SPEC + s) * AVT_G_MAX;
    const double* part = fb.partial_spec + blk * 256 + e;
    double v[NLD];
#pragma unroll
    for (int u = 0; u < NLD; ++u) v[u] = __builtin_nontemporal_load(part + (size_t)min(glo + u, G - 1) * 256);
    const unsigned long long wmine = el < ng ? fb.wmask_spec[blk + glo + el] : 0ull;
    // (the test's other inputs, requested in the same round trip: they depend on the snapshot only)
    const double cost_cur = sn.ctl.cost_cur, cost_const = sn.ctl.cost_const, sbp = sn.ctl.sbp, sbs = sn.ctl.sbs;
    const int try_valid = sn.ctl.try_valid, sp_n = sn.sp.n, valid = sn.sp.valid[s], ahead = sn.sp.ahead;
    const double* ps = fb.prior_spec + ((size_t)(have ? f : 0) * AVT_MAX_SPEC + s) * AVT_MAX_COMPS * AVT_PRIOR_STRIDE;
    const double* xk = fb.x_spec + ((size_t)f * AVT_MAX_SPEC + s) * d.xsize + 3 + 4 * d.J;
    const double pv = t < d.ncomps ? ps[(size_t)t * AVT_PRIOR_STRIDE] : 0.0;                 // lane c: score of component c
    const double xw = (t >= 64 && t - 64 < d.K) ? xk[t - 64] : 0.0;                           // lane 64 + k: shape coefficient k
    unsigned long long wrote = pair < 64 ? __ballot((int)((wmine >> (pair & 63)) & 1ull)) : ~0ull;
    if (EL == 32) wrote >>= 32 * (slice & 1);
    double a = 0.0;
#pragma unroll
    for (int u = 0; u < NLD; ++u) a += (u < ng && ((wrote >> u) & 1ull)) ? v[u] : 0.0;
    __shared__ double s_pv[AVT_MAX_COMPS], s_xw[AVT_MAX_SHAPE];
    if (t < AVT_MAX_COMPS) s_pv[t] = pv;
    if (t >= 64 && t < 64 + AVT_MAX_SHAPE) s_xw[t - 64] = xw;
    if (slice > 0) s_q[(slice - 1) * EL + el] = a;
    __syncthreads();
    if (slice == 0) {
#pragma unroll
        for (int i = 0; i < NSL - 1; ++i) a += s_q[i * EL + el];
        if (e == d.res_elem) {
            double ck = lm_objective_data(a, cost_const);      // (avt_device.h: the one spelling of the objective)
            if (sbp > 0.0 && d.ncomps > 0) {
                double bs = 1.7976931348623157e308;
                for (int c = 0; c < d.ncomps; ++c) if (s_pv[c] < bs) bs = s_pv[c];
                ck = lm_objective_add_pose(ck, sbp, bs);
            }
            if (sbs > 0.0) {
                double sa = 0.0;
#pragma unroll
                for (int kk = 0; kk < AVT_MAX_SHAPE; ++kk) if (kk < d.K) sa = lm_shape_term_add(sa, s_xw[kk], sbs);
                ck = lm_objective_add_shape(ck, sa);
            }
            const bool may = have && slot == 0 && try_valid == 1 && spn < sp_n && spn < AVT_MAX_SPEC && valid != 0 && fb.seq >= 2 &&
                             (fb.seq - 1 + ahead) + 1 <= fb.max_iters - 1;
            // The verdicts travel IN THE RIDE COUNTER (AVT_RIDE_* below), the one word of this launch every solver role reads anyway
            // (its poll): a load of their own for them taxed every k_solve launch 0.25-0.5 us, a branch on the snapshot's `ahead` 0.75 us
            // (same-box A/B, the feature off).  Set before this workgroup counts itself in, by the same thread: final when the count is.
            unsigned bits = 0u;
            if (have && fb.seq >= 2 && fb.seq + ahead > fb.max_iters) bits |= AVT_RIDE_IDLE;                      // sticky for the rest of the ICP iteration
            if (may && !(ck < cost_cur)) bits |= ((unsigned)fb.seq & AVT_RIDE_SEQ_MASK) << AVT_RIDE_SEQ_SHIFT;     // "the first queued step fails: take its test now" - tagged with the launch it is for
            __hip_atomic_fetch_and(fb.ride_ctr + f, ~(AVT_RIDE_SEQ_MASK << AVT_RIDE_SEQ_SHIFT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (bits) __hip_atomic_fetch_or(fb.ride_ctr + f, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(fb.ride_ctr + f, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <int STRIPS>
__device__ __forceinline__ void reduce_ride_block(const DeviceModel& dm, const FrameBuffers& fb, int f, int bx, double* s_q /* 256 doubles of the launch's dynamic LDS */) {
    constexpr int EL = 256 / STRIPS, NSL = 256 / EL, NLD = AVT_G_MAX / NSL;      // elements per strip, slices (of EL lanes), partial tiles per slice
    static_assert(EL == 64 || EL == 32, "a slice is a wave or half a wave");
    const AvtDims& d = dm.d;
    const int t = threadIdx.x, el = t % EL, slice = t / EL;
    const int NPAIR = d.NPAIR, NT = d.NT, HS = d.HS;
    const int pair = bx / STRIPS, strip = bx % STRIPS;
    const int e = strip * EL + el;                         // element of the tile: (row = (e >> 4 & 3) + 4 (e >> 6), col = e & 15)
    const int G = fb.G, glo = (G * slice) / NSL, ghi = (G * (slice + 1)) / NSL, ng = ghi - glo;       // G <= AVT_G_MAX: ng <= NLD
    const double* part = fb.partial + ((size_t)f * G * NPAIR + pair) * 256 + e;
    const size_t st = (size_t)NPAIR * 256;
    // (the snapshot: the solver of this launch rewrites the live block.  Requested FIRST, so that it arrives first: a frame that met the stopping
    // rule has nothing to reduce, and the launch is as long as its strips - the solver roles of such a frame have left without waiting for them)
    const int2 slot_state = *(const int2*)&fb.snap[f].ctl.cur_slot;
    double v[NLD];
#pragma unroll
    for (int u = 0; u < NLD; ++u) v[u] = __builtin_nontemporal_load(part + (size_t)min(glo + u, G - 1) * st);
    const unsigned long long wmine = el < ng ? fb.wmask[(size_t)f * G + glo + el] : 0ull;
    if (slot_state.y == AVT_TRY_DONE) return;
    const int try_slot = 1 - slot_state.x;
    int p = pair, ti = 0;
    while (p >= NT - ti) { p -= NT - ti; ++ti; }
    const int tj = ti + p;
    const int r = dm.tile_param[ti * 16 + ((e >> 4) & 3) + 4 * (e >> 6)], c = dm.tile_param[tj * 16 + (e & 15)];
    // a workgroup without batches in this pair wrote nothing: its tile is stale memory
    unsigned long long wrote = pair < 64 ? __ballot((int)((wmine >> (pair & 63)) & 1ull)) : ~0ull;
    if (EL == 32) wrote >>= 32 * (slice & 1);              // two slices per wave: my half of the ballot
    double a = 0.0;
#pragma unroll
    for (int u = 0; u < NLD; ++u) a += (u < ng && ((wrote >> u) & 1ull)) ? v[u] : 0.0;
    if (slice > 0) s_q[(slice - 1) * EL + el] = a;
    __syncthreads();
    if (slice == 0) {
#pragma unroll
        for (int i = 0; i < NSL - 1; ++i) a += s_q[i * EL + el];
        if (r >= 0 && c >= 0) {
            double* H = fb.Hraw + ((size_t)f * 2 + try_slot) * HS * HS;
            st_agent(H + (size_t)r * HS + c, a);
            if (ti != tj) st_agent(H + (size_t)c * HS + r, a);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);                      // my stores have been acknowledged
        if (t == 0) __hip_atomic_fetch_add(fb.ride_ctr + f, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// =================================================================================================
// k_solve<NTH, TRI>.  grid (nframes), block NTH.  <256, false>: systems of up to 88 columns (SMPL: 86), one 4x4 block per
// lane of 256 threads, the factor in a square LDS array whose unused blocks read as zeros.  <1024, true>: up to 180 columns
// (SMPL-H: 170): the same algorithm on 16 waves, the factor as a packed lower triangle (990 blocks = 139 KB), the skeleton
// scratch overlaid on it once the back substitution is done.
// =================================================================================================
// MODE (SOLVE_INIT / FIRST / NORMAL) is a template parameter so that the three roles are three symbols in a kernel trace
// (their durations differ six-fold) and the short ones do not carry the factorisation's code.
// SM: the model has SMPL's dimensions (24 joints, 10 shape keys, a 69-dimensional pose prior: solve_dims_smpl on the host) and the kernel
// is compiled for them - every count below that follows from them is a literal, so the index arithmetic of the system's loads, of its
// assembly and of the skeleton pass folds away (the loads of the system alone were ~27 instructions and two branches per entry pair with the
// row stride in a register: ~4.5 k clocks between the hand-over and the last request, one wave per SIMD issuing ~one instruction per 5 clocks)
__host__ __device__ inline void solve_dims_smpl_set(AvtDims& d) {
    d.J = 24; d.K = 10; d.P = 85; d.HS = 88; d.xsize = 109; d.NT = 6; d.NPAIR = 21; d.ndims = 69;
    d.prep_size = ((19 * 24 + 3 * 24 * 10 + 10 + 3) + 7) & ~7;
}
template <int NTH, bool TRI, int MODE, int RIDE = 0, bool SM = false>      // RIDE: 0, or the riding reduction's strips per tile pair
__global__ __launch_bounds__(NTH) void k_solve(DeviceModel dm_arg, FrameBuffers fb) {
    constexpr int mode = MODE;
    static_assert(!RIDE || (NTH == 256 && !TRI && MODE != SOLVE_INIT), "the riding reduction exists for the 256-thread solves");
    static_assert(!SM || (NTH == 256 && !TRI), "SMPL is a 256-thread solve");
    DeviceModel dm_local = dm_arg;
    if constexpr (SM) solve_dims_smpl_set(dm_local.d);
    const DeviceModel& dm = dm_local;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // RIDE: grid (frames, 1 + nspec + RIDE NPAIR): y = 0 the solver, y = 1 .. nspec the speculative solvers (the same system with
    // lambda up, lambda up^2 ..: the steps a run of rejected trial points will ask for), the rest the reduction in front of them
    // (behind the strips, when k_eval evaluated them: one workgroup per speculative step that reduces the step's cost, reduce_spec_cost)
    const int role = RIDE ? (int)blockIdx.y : 0;
    if constexpr (RIDE) {
        if (role > fb.nspec + RIDE * dm.d.NPAIR) { reduce_spec_cost<RIDE>(dm, fb, blockIdx.x + fb.f0, role - 1 - fb.nspec - RIDE * dm.d.NPAIR, (double*)smem); return; }
        if (role > fb.nspec) { reduce_ride_block<RIDE>(dm, fb, blockIdx.x + fb.f0, role - 1 - fb.nspec, (double*)smem); return; }
    }
    TPROBE_START();      // (the probes' time zero is the solver role's first instruction: its staging is part of what they measure)
    __builtin_amdgcn_s_setprio(3);   // one dependency chain per frame: issue ahead of the evaluation waves of another frame group on this CU
    const AvtDims d = dm.d;
    const int J = d.J, K = d.K, P = d.P, HS = d.HS;
    const int f = (RIDE ? (int)blockIdx.x : xcd_frame_1d(fb)) + fb.f0, t = threadIdx.x;      // (frame batches: on the XCD the frame's assembly ran on)
    AvtFrameCtl& ctl = fb.ctl[f];
    const int NBk = HS >> 2;                                // 4-row blocks covering rows 0..P (22 for SMPL)
    // W = L diag(d) of the factorisation H = L diag(d) L^T, block layout [pivot block kb][row block bi][18]: a 4x4
    // block is 16 doubles + 2 of padding (144 B), so lanes reading different blocks spread over the LDS banks; row P
    // carries L^-1 rhs
    // (256-thread shape: the small arrays every round of the factorisation touches - panel, reciprocal pivots - lie IN FRONT of the 70 KB factor:
    // below 64 KB an LDS address is an immediate offset of the instruction, above it three instructions that build it in a register)
    const size_t nblk = TRI ? (size_t)NBk * (NBk + 1) / 2 : (size_t)NBk * NBk;
    const size_t small_doubles = (size_t)(TRI ? MFG_PB_DOUBLES : MF_PB_DOUBLES) + ((max(HS, 4 * J) + 3) & ~1) + HS + 2 + (TRI ? 0 : 2 * HS);
    double* Lblk = TRI ? (double*)smem : (double*)smem + small_doubles;
    double* s_PB = TRI ? Lblk + nblk * 18 : (double*)smem;  // [96 or 192][4] the four panel columns of a round
    double* s_W = s_PB + (TRI ? MFG_PB_DOUBLES : MF_PB_DOUBLES);         // [max(HS, 4J) + 2]  reciprocal pivots, later the new quaternions
    double* s_delta = s_W + ((max(HS, 4 * J) + 3) & ~1);    // [HS]
    // [2][HS] gradient and diagonal of the undamped system (predicted decrease of the step, gain-ratio schedule); the 1024-thread
    // shape has no LDS left for it (P = 178: 158.5 of 159.5 KB) and keeps it in the frame's global scratch - written and read by
    // this workgroup only, a barrier in between
    // (SOLVE_DECIDE - the accept test alone, moment form - touches nothing of the above: its launch asks for the two state slots only)
    double* s_gD = s_delta + HS + 2;                        // (256-thread shape) [2][HS]: g | D, left there by the assembly (sys_tile)
    double* s_x = MODE == SOLVE_DECIDE ? (double*)smem : (TRI ? s_gD : Lblk + nblk * 18);      // [2][xsize] both state slots
    const PrepLayout L = prep_layout(J, K, d.xsize);
    // skeleton scratch: behind the factor (SMPL shape: staged at kernel start, hidden behind the factorisation) or ON it
    // (triangular shape: the factor is dead once the back substitution is done)
    double* B = TRI ? Lblk : s_x + ((2 * d.xsize + 1) & ~1);
    int2* s_items = (int2*)(B + L.ndoubles);
    int* s_level = (int*)(s_items + L.nitems);
    __shared__ int s_failf[2];
    const int xs = d.xsize;
    double* x0 = fb.x + ((size_t)f * 2) * xs;
    double* prep0 = fb.prep + ((size_t)f * 2) * d.prep_size;

    // everything that does not depend on the LM decision is requested now: both state slots and the skeleton constants
    // (256-thread shape: this thread's work items of the skeleton pass, every tree level - constants of the model, far in front of their use; the
    // 1024-thread shape stages the items behind the back substitution and reads them there)
    // 256-thread solves: all of it is requested into registers here and stored to LDS further down, behind the requests for the system's entries -
    // one round trip for everything instead of one per staging loop (avt_prep.h, prep_stage_request)
    // (not the riding shapes: their solver roles wait for the reduction's hand-over anyway, the staging loops' round trips hide in that wait, and
    // with the two-phase code beside them they measured 0.503 against 0.492 ms per step)
    constexpr bool TWO_PHASE = !TRI && !RIDE && MODE != SOLVE_DECIDE && MODE != SOLVE_INIT;
    PrepItems prep_items;
    const bool items_global = !TRI && MODE != SOLVE_DECIDE && MODE != SOLVE_INIT && d.fk_reg != 0;      // this thread's work items of the skeleton pass: requested now, from the model's (level, thread) table
    if (items_global) prep_items = prep_preload_items_global(dm);
    PrepStaged staged;
    double staged_x[2] = {0.0, 0.0};
    bool two_phase = false;
    if constexpr (TWO_PHASE) {
        two_phase = d.fk_reg != 0 && prep_stage_fits(d, 0) && 2 * xs <= 2 * NTH;      // (workgroup-uniform; SMPL-sized skeletons)
        if (two_phase) {
            staged_x[0] = t < 2 * xs ? x0[t] : 0.0; staged_x[1] = t + NTH < 2 * xs ? x0[t + NTH] : 0.0;
            staged = prep_stage_request<false>(dm, L);
        }
    }
    auto stage_store = [&]() {      // the LDS half of the two-phase staging: called once, in front of the barrier that publishes the state slots
        if constexpr (TWO_PHASE) {
            if (two_phase) {
                if (t < 2 * xs) s_x[t] = staged_x[0];
                if (t + NTH < 2 * xs) s_x[t + NTH] = staged_x[1];
                prep_stage_store<false>(d, L, staged, B, s_items, s_level);
            }
        }
    };
    if (!two_phase) {
        for (int e = t; e < 2 * xs; e += NTH) s_x[e] = x0[e];
        if ((!TRI || mode == SOLVE_INIT) && mode != SOLVE_DECIDE) prep_stage_constants<NTH>(dm, L, B, s_items, s_level);
    }
    if (!TRI && mode != SOLVE_INIT && mode != SOLVE_DECIDE) {   // the never-written blocks of the factor must read as zeros (back substitution)
        d2v* z = (d2v*)Lblk;
        for (int e = t; e < NBk * NBk * 9; e += NTH) z[e] = (d2v){0.0, 0.0};
    }

    if constexpr (MODE != SOLVE_DECIDE && MODE != SOLVE_INIT) TPROBE(18);
    if (mode == SOLVE_INIT) {
        // trial point := current point (frame batches; few frames: prep_init_block in k_lbs's grid does this).  The constant part
        // of the data cost is summed by the FIRST solve.
        const int cur = ctl.cur_slot, tr = 1 - cur;
        __syncthreads();
        if (t == 0) ctl.try_valid = 1;
        // frame batches: which batches each evaluation workgroup takes (scratch: the end of the factor's area, unused here)
        if (fb.G < 64) eval_ranges<NTH>(d, fb, f, t, (int*)(Lblk + nblk * 18) - (AVT_ERANGE_CAP + 2));
        const double* xc = s_x + cur * xs;
        for (int e = t; e < xs; e += NTH) x0[(size_t)tr * xs + e] = xc[e];
        prep_set_state(d, L, B, xc + 3, xc + 3 + 4 * J, xc);
        __syncthreads();
        prep_run<NTH>(dm, L, B, s_items, s_level, xc + 3, prep0 + (size_t)tr * d.prep_size, prep_preload_items<NTH>(d, s_items, s_level));
        return;
    }
    // ---- a. one round trip for everything the LM decision and the system need: the control block, the objective
    // terms of BOTH state slots and (256-thread shape) this lane's entries of BOTH data-term matrices (the slot is chosen
    // below).  The system is the bordered (P+1)x(P+1) matrix [[H + lambda diag H, .],[-g^T, .]] (row P carries the rhs so
    // D^-1 L^-1 (-g) falls out of the factorisation as row P of the unit-lower factor).
    const int NB = HS >> 2;
    const int fold_item0 = TRI ? 0 : backsub_fold_item(t, NB);      // this thread's block of the back substitution's fold (index arithmetic, off the chain)
    const double* H0 = fb.Hraw + ((size_t)f * 2) * HS * HS;
    // 256-thread shape: the system goes straight into the MFMA accumulator layout - wave w owns tile row rA = 5 - w (tile
    // slots 0..5) and, for w >= 2, tile row rB = w - 2 (slots 6, 7); lane (g4 = l >> 4, c16 = l & 15) holds rows 4v + g4 of
    // column c16 of a tile.  Raw data-term entries of BOTH slots are requested now (indices clamped, selected later).
    const int mf_wv = t >> 6, mf_g4 = (t >> 4) & 3, mf_c16 = t & 15, mf_rA = 5 - mf_wv, mf_rB = mf_wv - 2;
    // (six tile slots per wave: slot s <= rA is tile (rA, s), the slots behind are tiles (rB, 0..rB))
    double mraw[1][TRI ? 1 : 6][4];      // (one slot: the riding shapes request the trial slot's system in front of the decision, the batch shapes the chosen slot's behind it)
    // RIDE: the snapshot every solver role decides on (AvtSolveSnap) is requested now, before the wait for the reduction
    AvtFrameCtl snap_ctl;
    // (the speculative-step queue field by field, never as a structure: an entry chosen by an index the compiler does not know would put a copy of
    // the structure into scratch memory - 104 bytes and three dependent scratch round trips on the install path of every riding launch in the
    // round-4 build: next, n, valid[k] -; spec_pick chooses among values)
    static_assert(AVT_MAX_SPEC == 4, "the queue's entries are named one by one below");
    int sq_next = 0, sq_n = 0, sq_ahead = 0, sq_v0 = 0, sq_v1 = 0, sq_v2 = 0, sq_v3 = 0;
    double sq_l0 = 0.0, sq_l1 = 0.0, sq_l2 = 0.0, sq_l3 = 0.0, sq_p0 = 0.0, sq_p1 = 0.0, sq_p2 = 0.0, sq_p3 = 0.0;
    double snap_xw[AVT_MAX_SHAPE];
    unsigned fault_at_start = 0;      // (requested with everything else: a load of its own in front of the wait would be a round trip on the chain)
    if constexpr (RIDE) {
        fault_at_start = fb.fault[f];
        snap_ctl = fb.snap[f].ctl;
        const AvtSpecCtl& gq = fb.snap[f].sp;
        // (wave-uniform values, made scalars at once and never put into an array: the compiler folds a choice among array elements back into
        // a choice among addresses; the kernel has scalar registers to spare, no vector ones)
        auto uni_i = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
        auto uni_d = [](double v) { return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v))); };
        sq_next = uni_i(gq.next); sq_n = uni_i(gq.n); sq_ahead = uni_i(gq.ahead);
        sq_v0 = uni_i(gq.valid[0]); sq_v1 = uni_i(gq.valid[1]); sq_v2 = uni_i(gq.valid[2]); sq_v3 = uni_i(gq.valid[3]);
        sq_l0 = uni_d(gq.lambda[0]); sq_l1 = uni_d(gq.lambda[1]); sq_l2 = uni_d(gq.lambda[2]); sq_l3 = uni_d(gq.lambda[3]);
        sq_p0 = uni_d(gq.pred[0]); sq_p1 = uni_d(gq.pred[1]); sq_p2 = uni_d(gq.pred[2]); sq_p3 = uni_d(gq.pred[3]);
#pragma unroll
        for (int k = 0; k < AVT_MAX_SHAPE; ++k) snap_xw[k] = k < K ? fb.snap[f].xw[k] : 0.0;
        // The frame met the stopping rule in an earlier launch of this ICP iteration: nothing to decide, nothing to wait for (its strips leave as
        // early).  The snapshot has just been waited for by the scalar reads above, so this branch costs the other launches nothing.
        if (MODE == SOLVE_NORMAL && snap_ctl.try_valid == AVT_TRY_DONE) return;
    }
    // Past the iteration budget: earlier launches of this ICP iteration took accept tests ahead of their launches (folded rejections,
    // below), every test but the last - which belongs to the k_lbs launch - has been taken, and the trial point is waiting for that one.
    // Such a launch is idle: its solver roles do not wait for the reduction and leave behind the barrier below.  (A `return` right here -
    // a branch on the freshly loaded snapshot in front of everything else - cost every install-type launch 0.25 us: same-box A/B.)
    bool idle = false;
    unsigned ride_word = 0u;
    // (solver, RIDE) the speculative step the accept test may ask for in a moment: its 10 KB are requested now, while the reduction
    // is still on its way, so that a rejection only has to store them
    // (256-thread shape: 3 + 3J + K <= 87 and K <= 16 bound the prep block by 1496 doubles and the state by 115)
    constexpr int SPN = RIDE ? (1496 + 120 + NTH - 1) / NTH : 1;
    double sp_pre[SPN];
    if constexpr (RIDE) {
        const int k = min(sq_next, AVT_MAX_SPEC - 1), ncopy = xs + d.prep_size;
        const double* xsrc = fb.x_spec + ((size_t)f * AVT_MAX_SPEC + k) * xs;
        const double* psrc = fb.prep_spec + ((size_t)f * AVT_MAX_SPEC + k) * d.prep_size;
#pragma unroll
        for (int i = 0; i < SPN; ++i) {
            const int e = t + i * NTH;
            sp_pre[i] = (role == 0 && e < ncopy) ? (e < xs ? xsrc[e] : psrc[e - xs]) : 0.0;
        }
    }
    if constexpr (RIDE) stage_store();      // (riding shapes: the solver roles have time to spare in front of the reduction's hand-over)
    double prt[AVT_MAX_COMPS];      // (riding shapes) the prior's score of every component at the trial slot: the previous launch made them
    if constexpr (RIDE) {
#pragma unroll
        for (int c = 0; c < AVT_MAX_COMPS; ++c)
            prt[c] = (c < d.ncomps) ? fb.prior[(((size_t)f * 2 + (1 - snap_ctl.cur_slot)) * AVT_MAX_COMPS + c) * AVT_PRIOR_STRIDE] : 0.0;
    }
    if constexpr (RIDE) {      // the reduction workgroups of this launch have all delivered their strips of the trial point's system
        if (t == 0) {
            // (the count runs on from launch to launch of an ICP iteration - k_finalize clears it -: several workgroups wait on it)
            // The launch is an ordinary one: its reduction workgroups never wait for anything and at most 1 + AVT_MAX_SPEC workgroups
            // per frame wait here, so they always get to run - but WHEN is the dispatcher's business (another stream, another
            // process, a profiler serialising dispatch).  The wait is therefore bounded by wall-clock time (2 s by default), and a
            // role that gives up says so: the frame's fault word makes the host calls that return its result fail
            // (download_state, avt_shard_gather_download) instead of handing out a fit made from a half-reduced system.
            const unsigned want = (unsigned)(fb.seq * (RIDE * d.NPAIR + fb.nspec_cost));      // (< 2^16: AVT_RIDE_COUNT_MASK)
            const long long t0 = wall_clock64();
            TPROBE(10);
            int spins_ = 0;
            // (a frame that already carries a fault of this call fails fast: every later launch would wait the full time again)
            const long long limit = fault_at_start ? 0 : fb.ride_timeout;
            bool there;
            unsigned word;
            while (!(there = ((word = __hip_atomic_load(fb.ride_ctr + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & AVT_RIDE_COUNT_MASK) >= want) &&
                   wall_clock64() - t0 < limit)
                { __builtin_amdgcn_s_sleep(1); ++spins_; }
            TPROBE(11);
#ifdef AVT_TIMING
            if (blockIdx.y == 0) fb.trace[(size_t)(blockIdx.x + fb.f0) * 64 + 52] = (double)spins_;
#endif
            if (!there) atomicOr(fb.fault + f, AVT_FAULT_RIDE_TIMEOUT);
            s_failf[1] = (int)word;      // the spec-cost workgroup's verdicts ride in the word (reduce_spec_cost): handed to the other threads across the barrier
        }
        __syncthreads();
        // (behind the second barrier instead - after the requests for the system's entries - this costs an install-type launch 0.2 instead of 0.6 us,
        // but an idle launch 7.6 instead of 4.8 us: measured both ways, the frames whose rejections come in runs gain more from cheap idle launches)
        ride_word = (unsigned)s_failf[1];
        // (or the frame met the stopping rule in an earlier launch of this ICP iteration: AVT_TRY_DONE in the snapshot, which arrived long ago)
        idle = MODE == SOLVE_NORMAL && ((ride_word & AVT_RIDE_IDLE) != 0 || snap_ctl.try_valid == AVT_TRY_DONE);      // past the iteration budget: nothing to decide, nothing to solve
        if (idle) return;
    }
    auto hload = [&](const double* q) { if constexpr (RIDE) return ld_agent(q); else return *q; };
    // Riding shapes know the slots from the snapshot (it arrived in front of the wait) and request the TRIAL slot's system alone: an accepted
    // trial point (87 % of them under the default step rule) is the point the system is needed at; a rejected one is followed by an installed
    // speculative step (no system at all) or - the queue empty - by a solve of the current slot's system, requested then (a round trip more on
    // that rare path, 24 instead of 48 requests per lane on every other).  The batch shapes learn the slots in this very round trip: both.
    auto prior_entries_early = [&](int rb, int cb, bool up, const double* PrC, const double* priC, int nn, double (&prvs)[4], double& gqc) __attribute__((always_inline)) {
        const int pcc = min(max(16 * cb + mf_c16 - 6, 0), max(nn - 1, 0));
        gqc = up ? priC[2 + pcc] : 0.0;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int prc = min(max(16 * rb + 4 * v + mf_g4 - 6, 0), max(nn - 1, 0));
            prvs[v] = up ? PrC[(size_t)prc * nn + pcc] : 0.0;
        }
    };
    int ride_slot = RIDE ? 1 - snap_ctl.cur_slot : 0;
    // ... and the pose prior's entries this lane adds to its tiles (precision matrix and gradient of ONE component at ONE slot, 24 + 6 values):
    // riding shapes request them here, with the system - for the component the trial point sits in, whose scores arrived in front of the wait -
    // instead of in a second round trip behind the decision (~2 k clocks of every full pass); the decision picking another point or component
    // requests them again.  Batch shapes: behind the decision, as before (they learn slot and scores in this round trip).
    double prv_pre[TRI ? 1 : 6][4], gq_pre[TRI ? 1 : 6];
    int prior_comp = -2, prior_slot = -1;
    auto load_prior = [&](int comp_sel, int slot_sel, double sbp_sel) __attribute__((always_inline)) {
        if constexpr (!TRI && MODE != SOLVE_DECIDE) {
            prior_comp = comp_sel; prior_slot = slot_sel;
            const bool up = sbp_sel > 0.0 && d.ncomps > 0 && comp_sel >= 0;
            const int nn = d.ndims, cs = comp_sel >= 0 ? comp_sel : 0;
            const double* PrC = dm.prior_prec + (size_t)cs * nn * nn;
            const double* priC = fb.prior + (((size_t)f * 2 + slot_sel) * AVT_MAX_COMPS + cs) * AVT_PRIOR_STRIDE;
            auto role_fn = [&](auto role_c) __attribute__((always_inline)) {
                constexpr int W = decltype(role_c)::value, rA = 5 - W, rB = W - 2;
#pragma unroll
                for (int ti = 0; ti < 6; ++ti) {
                    const bool first = ti <= rA;
                    const int rb = first ? rA : (rB > 0 ? rB : 0), cb = first ? ti : ti - rA - 1;
                    const bool own = first || (rB >= 0 && cb <= rB);
                    if (own) prior_entries_early(rb, cb, up, PrC, priC, nn, prv_pre[ti], gq_pre[ti]);
                    else { gq_pre[ti] = 0.0; prv_pre[ti][0] = prv_pre[ti][1] = prv_pre[ti][2] = prv_pre[ti][3] = 0.0; }
                }
            };
            switch (mf_wv) {
                case 0: role_fn(std::integral_constant<int, 0>{}); break;
                case 1: role_fn(std::integral_constant<int, 1>{}); break;
                case 2: role_fn(std::integral_constant<int, 2>{}); break;
                default: role_fn(std::integral_constant<int, 3>{}); break;
            }
        }
    };
    const double* Hl = H0 + (size_t)(4 * 0 + mf_g4) * HS + mf_c16;
    auto load_system = [&]() __attribute__((always_inline)) {
    if constexpr (!TRI && MODE != SOLVE_DECIDE) {
        // (one copy per wave role: tile rows and columns are compile-time constants there, an entry's address is one add)
        auto load_role = [&](auto role) {
            constexpr int W = decltype(role)::value, rA = 5 - W, rB = W - 2;
#pragma unroll
            for (int ti = 0; ti < 6; ++ti) {
                constexpr bool hasB = rB >= 0;
                const bool first = ti <= rA;
                const int rb = first ? rA : (hasB ? rB : 0), cb = first ? ti : ti - rA - 1;
                const bool own = first || (hasB && cb <= rB);
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    // rows / columns past the matrix (tile row 5, tile column 5) are clamped: their values are replaced below
                    const int row = 16 * rb + 4 * v, col = 16 * cb;
                    const bool inb = row + 3 < HS && col + 15 < HS;
                    const size_t off = inb ? (size_t)row * HS + col
                                           : (size_t)(min(row + mf_g4, HS - 1) - mf_g4) * HS + (min(col + mf_c16, HS - 1) - mf_c16);
                    mraw[0][ti][v] = own ? hload(Hl + ride_slot * ((size_t)HS * HS) + off) : 0.0;      // (one slot, see above)
                }
            }
        };
        switch (mf_wv) {
            case 0: load_role(std::integral_constant<int, 0>{}); break;
            case 1: load_role(std::integral_constant<int, 1>{}); break;
            case 2: load_role(std::integral_constant<int, 2>{}); break;
            default: load_role(std::integral_constant<int, 3>{}); break;
        }
    }
    };
    // (batch shapes, late round 6: they used to request BOTH slots' systems here - 98 requests, 3.1 k clocks of issue alone - and the prior's
    // entries in a second round trip behind the decision.  The decision needs the control block, the scores and the two corner entries only: those
    // go out here with the staging; the chosen slot's system and the prior's entries go out together behind the decision - one slot, one round trip)
    if constexpr (MODE != SOLVE_DECIDE) TPROBE(16);
    if constexpr (RIDE) load_system();
    if constexpr (MODE != SOLVE_DECIDE) TPROBE(17);
    // prior score of every component: strict '<' in ascending component order (GaussianMixture.cpp:103)
    double best[2] = {1.7976931348623157e308, 1.7976931348623157e308};
    int bcomp[2] = {-1, -1};
    if constexpr (RIDE) {
        if (d.ncomps > 0) {
#pragma unroll
            for (int c = 0; c < AVT_MAX_COMPS; ++c)
                if (c < d.ncomps && prt[c] < best[0]) { best[0] = prt[c]; bcomp[0] = c; }
            best[1] = best[0]; bcomp[1] = bcomp[0];      // (the trial slot's: the current slot's component is the control block's)
        }
        load_prior(snap_ctl.sbp > 0.0 && d.ncomps > 0 ? bcomp[0] : -1, ride_slot, snap_ctl.sbp);
    }
    double hpp0, hpp1;
    if constexpr (RIDE) { hpp0 = hpp1 = hload(H0 + ride_slot * ((size_t)HS * HS) + (size_t)P * HS + P); }
    else { hpp0 = hload(H0 + (size_t)P * HS + P); hpp1 = hload(H0 + (size_t)HS * HS + (size_t)P * HS + P); }

    if constexpr (MODE != SOLVE_DECIDE) TPROBE(13);
    const double lm_up = fb.params->lm_up, lm_down = fb.params->lm_down, lm_min = fb.params->lm_min, lm_max = fb.params->lm_max;   // same round trip
    const bool gain = fb.params->lm_policy != 0.0;          // gain-ratio damping schedule (avt_options::lm_policy)
    const double ftol = fb.params->ftol;                    // the stopping rule (avt_options::function_tolerance; 0 = off)
    // RIDE: every solver role decides on the snapshot the evaluation launch made (AvtSolveSnap): the solver rewrites the live
    // control block further down, and a speculative workgroup may start late
    const AvtFrameCtl& cin = RIDE ? snap_ctl : ctl;
    const int cur0 = cin.cur_slot, try_valid = cin.try_valid, comp_cur0 = cin.comp_cur;
    const double sbp = cin.sbp, sbs = cin.sbs, cost_cur0 = cin.cost_cur;
    double cost_const = cin.cost_const;
    double lambda = cin.lambda, nu = cin.nu;
    const double pred0 = cin.pred;
    // (the iteration counters too: read where they are bumped - behind the first stores to the control block - they were a memory round trip of
    // their own on wave 0, between the decision and the system's assembly)
    const int gn_iterations0 = cin.gn_iterations, accepted0 = cin.accepted;
    __shared__ double s_cc;
    if (mode == SOLVE_FIRST && t < 64) {   // the constant part of the data cost (k_records' trailing workgroups), once per ICP iteration
        double a = 0.0;
        for (int e = t; e < fb.const_used; e += 64) a += fb.const_part[(size_t)f * fb.const_blocks + e];
        a = wave_sum(a);
        if (t == 0) s_cc = 0.5 * a;
    }
    // (batch shapes) prior score of every component at both slots
    if (!RIDE && d.ncomps > 0) {
        double pr[2][AVT_MAX_COMPS];
#pragma unroll
        for (int sl = 0; sl < 2; ++sl)
#pragma unroll
            for (int c = 0; c < AVT_MAX_COMPS; ++c)
                pr[sl][c] = (c < d.ncomps) ? fb.prior[(((size_t)f * 2 + sl) * AVT_MAX_COMPS + c) * AVT_PRIOR_STRIDE] : 0.0;
#pragma unroll
        for (int sl = 0; sl < 2; ++sl)
#pragma unroll
            for (int c = 0; c < AVT_MAX_COMPS; ++c)
                if (c < d.ncomps && pr[sl][c] < best[sl]) { best[sl] = pr[sl][c]; bcomp[sl] = c; }
    }
    if constexpr (!RIDE) stage_store();      // (behind the requests for the system's entries: one round trip for everything)
    __syncthreads();   // the staged state slots are visible; every lane has read the control block
    if constexpr (MODE != SOLVE_DECIDE) TPROBE(14);
    // the frame met the stopping rule in an earlier launch of this ICP iteration: no trial point, no test, no iteration (the riding shapes left above)
    if (!RIDE && mode != SOLVE_FIRST && try_valid == AVT_TRY_DONE) return;

    if (mode == SOLVE_FIRST) cost_const = s_cc;
    const int try0 = 1 - cur0;
    const double* xt = s_x + try0 * xs;
    double cost = lm_objective_data(try0 ? hpp1 : hpp0, cost_const);      // (avt_device.h: the one spelling of the objective)
    int comp_try = -1;
    if (sbp > 0.0 && d.ncomps > 0) {
        comp_try = try0 ? bcomp[1] : bcomp[0];
        cost = lm_objective_add_pose(cost, sbp, try0 ? best[1] : best[0]);
    }
    if (sbs > 0.0) {
        double a = 0.0;
        if constexpr (RIDE) {
#pragma unroll
            for (int k = 0; k < AVT_MAX_SHAPE; ++k) if (k < K) a = lm_shape_term_add(a, snap_xw[k], sbs);
        } else {
            for (int k = 0; k < K; ++k) a = lm_shape_term_add(a, xt[3 + 4 * J + k], sbs);
        }
        cost = lm_objective_add_shape(cost, a);
    }
    int cur = cur0;
    bool accepted = false;
    // The reference's stopping rule (options.function_tolerance, AvatarOptimizer.cpp:1333; Ceres' line-search minimiser: |cost change| <=
    // function_tolerance x the cost the step started from): an ACCEPTED step that small ends the frame's iterations of this ICP iteration.
    // Every solver role takes the test on the same numbers, so the speculative ones leave with the solver.
    bool converged = false;
    if (mode == SOLVE_FIRST) {
        accepted = true;
        cur = try0;
    } else if (try_valid == 1) {
        if (cost < cost_cur0) {
            accepted = true; cur = try0;
            converged = ftol > 0.0 && (cost_cur0 - cost) <= ftol * cost_cur0;
            if (gain) {      // rho = actual / predicted decrease; lambda *= max(lm_down, 1 - (2 rho - 1)^3), the rejection factor starts over at lm_up
                const double u = 2.0 * ((cost_cur0 - cost) / pred0) - 1.0;
                lambda = fmin(fmax(lambda * fmax(lm_down, 1.0 - u * u * u), lm_min), lm_max);
                nu = lm_up;
            } else lambda = fmax(lambda * lm_down, lm_min);
        } else if (gain) { lambda = fmin(lambda * nu, lm_max); nu *= 2.0; }
        else lambda = fmin(lambda * lm_up, lm_max);
    }
    const double cost_cur = accepted ? cost : cost_cur0;
    const int comp = accepted ? comp_try : comp_cur0;
    if constexpr (MODE != SOLVE_DECIDE) TPROBE(15);
    // Speculative steps (RIDE shapes): a rejected trial point is followed by a solve of the SAME system with lambda up - which
    // a speculative workgroup of the last full solve launch has already made.  The solver then installs it (trial state and
    // skeleton tables copied into the trial slot) instead of factoring, and the speculative workgroups of this launch go home.
    AvtSpecCtl& sp = fb.spec[f];
    const int sp_next = RIDE ? sq_next : 0, sp_n = RIDE ? sq_n : 0;
    const bool rejected = mode != SOLVE_FIRST && try_valid == 1 && !accepted;
    // Folded rejections (RIDE shapes whose k_eval evaluated the queued speculative steps' costs, fb.nspec_cost): the trial point has just
    // been rejected, so the next trial point is the first queued step - made from the SAME system with the lambda this rejection leaves.
    // Its cost is known; if it fails the accept test as well, that test (the next GN iteration's) is taken here and now, lambda advances
    // as it would have, and the step after it is looked at - until one passes (it is installed below, evaluated in full by the next
    // k_eval launch and accepted by the next solve) or the queue ends (a full solve with the lambda reached).  Same tests on the same
    // numbers in the same order as one launch pair per rejection: nothing changes but the number of launches that do something.
    // The budget: this launch takes test number seq - 1 + ahead; the LAST test of the ICP iteration is left to the k_lbs launch.
    // Folded rejection (riding shapes whose k_eval evaluated the cost of the first queued speculative step, FrameBuffers::nspec_cost = 1).
    // The trial point has just been rejected, so the next trial point is that step - made from the SAME system with the lambda this
    // rejection leaves.  Its cost is known: if it fails the accept test as well, that test (the next GN iteration's) is taken here and
    // now, lambda advances as it would have, and the step behind it is the one to install (or, the queue at its end, a full solve with
    // the lambda reached).  Same tests on the same numbers in the same order as one launch pair per rejection: nothing changes but the
    // number of launches that do something.
    // The test itself is NOT taken here: this launch's spec-cost workgroup (reduce_spec_cost, another CU, beside the strips) forms the
    // step's objective and publishes one word - "fails, and the iteration budget allows taking its test now" - which arrived with the
    // system's corner entries.  Everything tried inside the solver's own instruction stream taxed every launch, the feature off: a loop
    // over the queue behind `if (rejected)` 0.45 us, the same as a function 10 us (a call gives the kernel a private stack), its
    // inputs requested unconditionally with selects instead of branches 2.1 us (same-box A/B of the builds, tools/kt_lib_ab.sh).
    int nfold = 0;
    if constexpr (RIDE) {
        const bool fold = rejected && fb.nspec_cost > 0 && ((ride_word >> AVT_RIDE_SEQ_SHIFT) & AVT_RIDE_SEQ_MASK) == ((unsigned)fb.seq & AVT_RIDE_SEQ_MASK);
        nfold = fold ? 1 : 0;
        const double lam_adv = gain ? fmin(lambda * nu, lm_max) : fmin(lambda * lm_up, lm_max);
        lambda = fold ? lam_adv : lambda;
        nu = (fold && gain) ? nu * 2.0 : nu;
    }
    const int sp_use = sp_next + nfold;
    const bool use_spec = RIDE && rejected && sp_use < sp_n && sp_use < AVT_MAX_SPEC && spec_pick(sp_use, sq_v0, sq_v1, sq_v2, sq_v3) != 0;
    if (RIDE && role > 0 && use_spec) return;
    if (RIDE && role > 0) {      // my damping: what `role` rejections in a row would make of the solver's
        for (int i = 0; i < role; ++i) {
            if (gain) { lambda = fmin(lambda * nu, lm_max); nu *= 2.0; }
            else lambda = fmin(lambda * lm_up, lm_max);
        }
    }
    if (t == 0 && role == 0) {
        ctl.cur_slot = cur;
        ctl.cost_cur = cost_cur;
        ctl.comp_cur = comp;
        int it = gn_iterations0;
        if (mode == SOLVE_FIRST) { ctl.cost_initial = cost; ctl.cost_const = cost_const; }
        else {
            for (int i = 0; i < nfold; ++i) { it += 1; if (it < 40) fb.trace[(size_t)f * 64 + it] = cost_cur; }      // the folded tests: rejections, the objective stays
            it += 1; ctl.gn_iterations = it; if (accepted) ctl.accepted = accepted0 + 1;
        }
        if (it < 40) fb.trace[(size_t)f * 64 + it] = cost_cur;
    }
    if (converged) {      // (never in SOLVE_FIRST) the accepted point stays; nothing follows it in this ICP iteration
        if (t == 0 && role == 0) {
            ctl.lambda = lambda; ctl.nu = nu; ctl.try_valid = AVT_TRY_DONE;
            ctl.dec_cur_slot = cur; ctl.dec_try_valid = AVT_TRY_DONE; ctl.dec_cost_cur = cost_cur; ctl.dec_lambda = lambda; ctl.dec_nu = nu;
            if (RIDE) { sp.next = 0; sp.n = 0; }      // no speculative steps: k_eval's spec-cost workgroups find the queue empty
        }
        return;
    }
    if constexpr (MODE == SOLVE_DECIDE) {
        // The accept test of the LAST trial point of an ICP iteration in the moment form (no solve follows it): the accept / reject update of
        // lambda and nothing else - what lm_last_decide (avt_decide.h) does for the row form and the oracle does for both (a full
        // SOLVE_NORMAL pass here also multiplied lambda when its unused factorisation was refused, and cost a factorisation per frame).
        if (t == 0) { ctl.lambda = lambda; ctl.nu = nu; }
        return;
    }
    if (RIDE && use_spec) {      // (role 0) the step exists: install it as the new trial point
        const int k = sp_use, tr = 1 - cur;
        const double* xsrc = fb.x_spec + ((size_t)f * AVT_MAX_SPEC + k) * xs;
        const double* psrc = fb.prep_spec + ((size_t)f * AVT_MAX_SPEC + k) * d.prep_size;
        if (nfold > 0) {      // (not the step that was requested at kernel start)
#pragma unroll
            for (int i = 0; i < SPN; ++i) {
                const int e = t + i * NTH;
                sp_pre[i] = e < xs + d.prep_size ? (e < xs ? xsrc[e] : psrc[e - xs]) : 0.0;
            }
        }
#pragma unroll
        for (int i = 0; i < SPN; ++i) {
            const int e = t + i * NTH;
            if (e < xs) x0[(size_t)tr * xs + e] = sp_pre[i];
            else if (e < xs + d.prep_size) prep0[(size_t)tr * d.prep_size + (e - xs)] = sp_pre[i];
        }
        if (t == 0) {
            const double lam = spec_pick(k, sq_l0, sq_l1, sq_l2, sq_l3), prd = spec_pick(k, sq_p0, sq_p1, sq_p2, sq_p3);
            sp.next = k + 1; sp.ahead = sq_ahead + nfold;
            ctl.lambda = lam; ctl.try_valid = 1;
            ctl.dec_cur_slot = cur; ctl.dec_try_valid = 1; ctl.dec_cost_cur = cost_cur; ctl.dec_lambda = lam;
            // (the accept test above already took this rejection into lambda / nu: the installed step was made with exactly that lambda)
            ctl.pred = prd; ctl.nu = nu; ctl.dec_pred = prd; ctl.dec_nu = nu;
        }
        return;
    }
    if constexpr (RIDE) {
        if (cur != ride_slot) {      // (workgroup-uniform) a rejected trial point and no speculative step to install: the current slot's system after all
            ride_slot = cur;
            load_system();
        }
    } else {
        ride_slot = cur;
        load_system();
    }
    if constexpr (!TRI && MODE != SOLVE_DECIDE) {
        const int comp_need = (sbp > 0.0 && d.ncomps > 0) ? comp : -1;
        if (!RIDE || prior_comp != comp_need || prior_slot != cur) load_prior(comp_need, cur, sbp);      // (workgroup-uniform)
    }
    TPROBE_FIRST();

    // ---- b. the damped system of the current point, straight into registers (second, short round trip: the
    // precision block and gradient of the chosen GMM component, read-only data) ------------------------------
    const double* xc = s_x + cur * xs;
    const double* pri = fb.prior + (((size_t)f * 2 + cur) * AVT_MAX_COMPS + (comp >= 0 ? comp : 0)) * AVT_PRIOR_STRIDE;
    const bool use_pose = sbp > 0.0 && d.ncomps > 0 && comp >= 0;
    const int n = d.ndims;
    const double sc = 0.707106781186548 * sbp;          // literal constant (AvatarOptimizer.cpp:684)
    const double sc2 = sc * sc;
    const double gs = sc * sbp * 0.7071067811865476;    // J^T r = sc*sbp*sqrt(1/2) * Prec (x - mu)
    const double* Pr = dm.prior_prec + (size_t)(comp >= 0 ? comp : 0) * n * n;
    bool fail = false;
    double* s_R = s_W;                                      // [HS] reciprocal pivots
    // ---- b. the damped system of the current point in MFMA accumulator layout: tile (rb, cb), lane (g4, c16) holds rows
    // 16 rb + 4 v + g4 (v = 0..3) of column 16 cb + c16.  raw(v) = the data-term entry; the priors come from a second, short
    // round trip (precision entries and gradient of the chosen GMM component, read-only data).  All loads unconditional on
    // clamped indices, the conditions applied as selects: loads in flight together, no branches.
    // (dgn: the UNDAMPED diagonal entry this lane holds in the tile, if it holds one - what the predicted decrease of the gain-ratio schedule needs beside
    // the gradient, which is the negated row P of the tile.  Returned in a register: a store to LDS in here makes the compiler re-request the prior's
    // tables after it, tile by tile - 11 k clocks, measured)
    auto sys_tile = [&](int rb, int cb, bool own, const double (&raw)[4], double& dgn, const double (&prvs)[4], double gqc) __attribute__((always_inline)) {
        const int col = 16 * cb + mf_c16, pc = col - 6, sk = col - (3 + 3 * J);
        const int skc = min(max(sk, 0), max(K - 1, 0));
        const double xqc = xc[3 + 4 * J + skc];
        const bool in_pose_c = use_pose && pc >= 0 && pc < n;
        const bool shape_c = sbs > 0.0 && sk >= 0 && col < P;
        v4f64 out;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int row = 16 * rb + 4 * v + mf_g4, pr_ = row - 6;
            const double prv = prvs[v];
            const double v0 = raw[v];
            // rows < P: H + priors, diagonal damped
            double vh = v0;
            vh += (in_pose_c && pr_ >= 0 && pr_ < n) ? sc2 * prv : 0.0;
            if (row == col) { vh += shape_c ? sbs * sbs : 0.0; dgn = vh; vh += lambda * vh; }
            // row P: -(J^T r) including the priors
            double vg = v0;
            vg += in_pose_c ? gs * gqc : 0.0;
            vg += shape_c ? sbs * (xqc * sbs) : 0.0;
            const bool inside = row <= P && col < P;
            const double val = inside ? (row < P ? vh : -vg) : ((row == col) ? 1.0 : 0.0);
            out[v] = own ? val : 0.0;
        }
        return out;
    };
    if constexpr (!TRI) {
        v4f64 tile[6];
        double dgn[6];
        // One copy per wave role (as the loads above): a tile's block row and column are compile-time constants there, and with SMPL's dimensions
        // literal as well (SM) nearly every select of sys_tile folds - which rows and columns the priors touch, where the diagonal and row P lie,
        // what is padding.  Written for run-time tile coordinates the assembly was ~70 instructions per entry, 8.6 k clocks of every full pass.
        auto asm_role = [&](auto role_c) __attribute__((always_inline)) {
            constexpr int W = decltype(role_c)::value, rA = 5 - W, rB = W - 2;
#pragma unroll
            for (int ti = 0; ti < 6; ++ti) {
                const bool first = ti <= rA;
                const int rb = first ? rA : (rB > 0 ? rB : 0), cb = first ? ti : ti - rA - 1;
                const bool own = first || (rB >= 0 && cb <= rB);
                double raw[4];
#pragma unroll
                for (int v = 0; v < 4; ++v) raw[v] = mraw[0][TRI ? 0 : ti][v];
                dgn[ti] = 0.0;
                if (own) tile[ti] = sys_tile(rb, cb, true, raw, dgn[ti], prv_pre[ti], gq_pre[ti]);
                else tile[ti] = (v4f64){0.0, 0.0, 0.0, 0.0};
            }
            if (gain) {      // g and D for the predicted decrease (wave 1, behind the back substitution): every entry has exactly one owner
#pragma unroll
                for (int ti = 0; ti < 6; ++ti) {
                    const bool first = ti <= rA;
                    const int rb = first ? rA : (rB > 0 ? rB : 0), cb = first ? ti : ti - rA - 1;
                    const bool own = first || (rB >= 0 && cb <= rB);
                    const int col = 16 * cb + mf_c16, v = (mf_c16 - mf_g4) >> 2;      // the diagonal of a diagonal tile: row 4 v + g4 = column c16
                    if (own && rb == cb && ((mf_c16 - mf_g4) & 3) == 0 && v >= 0 && col < P) s_gD[HS + col] = dgn[ti];
                    const int vP = (P & 15) >> 2;      // row P = 16 (P >> 4) + 4 vP + (P & 3) holds -g
                    const double mg = vP == 0 ? tile[ti][0] : (vP == 1 ? tile[ti][1] : (vP == 2 ? tile[ti][2] : tile[ti][3]));
                    if (own && rb == (P >> 4) && mf_g4 == (P & 3) && col < P) s_gD[col] = -mg;
                }
            }
        };
        switch (mf_wv) {
            case 0: asm_role(std::integral_constant<int, 0>{}); break;
            case 1: asm_role(std::integral_constant<int, 1>{}); break;
            case 2: asm_role(std::integral_constant<int, 2>{}); break;
            default: asm_role(std::integral_constant<int, 3>{}); break;
        }
        TPROBE(2);
        // ---- c. LDL^T, four pivots and two barriers per round, the trailing matrix in the accumulators (mf_rounds) ----
        if (t == 0) s_failf[0] = 0;
        bool okf;
        const v4f64 z4 = {0.0, 0.0, 0.0, 0.0};
        switch (mf_wv) {
            case 0: { v4f64 accA[6] = {tile[0], tile[1], tile[2], tile[3], tile[4], tile[5]}, accB[2] = {z4, z4};
                      okf = mf_rounds<0>(accA, accB, s_PB, Lblk, s_R, &s_failf[0], P, NB, t); break; }
            case 1: { v4f64 accA[6] = {tile[0], tile[1], tile[2], tile[3], tile[4], z4}, accB[2] = {z4, z4};
                      okf = mf_rounds<1>(accA, accB, s_PB, Lblk, s_R, &s_failf[0], P, NB, t); break; }
            case 2: { v4f64 accA[6] = {tile[0], tile[1], tile[2], tile[3], z4, z4}, accB[2] = {tile[4], z4};
                      okf = mf_rounds<2>(accA, accB, s_PB, Lblk, s_R, &s_failf[0], P, NB, t); break; }
            default: { v4f64 accA[6] = {tile[0], tile[1], tile[2], z4, z4, z4}, accB[2] = {tile[3], tile[4]};
                       okf = mf_rounds<3>(accA, accB, s_PB, Lblk, s_R, &s_failf[0], P, NB, t); break; }
        }
        fail = !okf;
    } else {
        // 1024-thread shape: 16 waves on a T x T tile grid, run-time tile slots (mfg_rounds); the entries of the chosen slot only,
        // requested after the decision (128 registers per lane)
        const int T = (P + 16) >> 4;
        const MfgSlots S = mfg_slots(__builtin_amdgcn_readfirstlane(t >> 6), T);
        const double* Hc = H0 + (size_t)cur * HS * HS;
        v4f64 acc[MFG_SLOTS];
#pragma unroll
        for (int sl = 0; sl < MFG_SLOTS; ++sl) {
            const int rb = max(S.rb[sl], 0), cb = max(S.cb[sl], 0);
            double raw[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) raw[v] = Hc[(size_t)min(16 * rb + 4 * v + mf_g4, HS - 1) * HS + min(16 * cb + mf_c16, HS - 1)];
            double dgn_unused = 0.0, prvs[4], gqc;
            prior_entries_early(rb, cb, use_pose, Pr, pri, n, prvs, gqc);
            acc[sl] = sys_tile(rb, cb, S.rb[sl] >= 0, raw, dgn_unused, prvs, gqc);
        }
        TPROBE(2);
        if (t == 0) s_failf[0] = 0;
        fail = !mfg_rounds(acc, S, s_PB, Lblk, s_R, &s_failf[0], P, NB, T, t);
    }
    __syncthreads();
    if constexpr (!TRI) fail = s_failf[0] != 0;      // (256-thread shape: the rounds do not stop at a refused pivot, see mf_round)
    TPROBE(3);
    const bool ok = !fail;
    const int ntry = 1 - cur;
    // where the new trial point goes: the trial slot, or (speculative solver) its own slot until a rejection asks for it
    double* xn = role == 0 ? x0 + (size_t)ntry * xs : fb.x_spec + ((size_t)f * AVT_MAX_SPEC + (role - 1)) * xs;
    double* prep_out = role == 0 ? prep0 + (size_t)ntry * d.prep_size : fb.prep_spec + ((size_t)f * AVT_MAX_SPEC + (role - 1)) * d.prep_size;
    double* s_qnew = s_W;                                   // [4J] quaternions of the new trial point (s_W is free again)
    double pred_new = 0.0;                                  // (wave 0)
    if (ok) {
        // ---- back substitution by wave 0 (the other waves wait at the barrier)
        if constexpr (!TRI) {      // (all waves) the steps' 4x4 solves folded into the factor; the M's take the panel buffer, row P of W waits in s_delta
            backsub_fold<NTH>(Lblk, s_R, NB, P, t, fold_item0, s_PB, s_delta);
            TPROBE(8);
            __syncthreads();
            TPROBE(9);
        }
        if (t < 64) {
            if constexpr (TRI) backsub_tri(Lblk, s_R, NB, P, t, s_delta);
            else if (NB == 22) backsub_chain<22>(Lblk, NB, P, t, s_delta, s_PB, s_delta);
            else backsub_chain<0>(Lblk, NB, P, t, s_delta, s_PB, s_delta);
            if (TRI && gain) {
                // (1024-thread shape: no LDS left for g and D)
                // Predicted decrease of the quadratic model, 1/2 delta^T (lambda D delta - g), by the same wave (fixed butterfly).  g and the
                // undamped diagonal D are formed again from the reduced system and the priors, entry by entry as sys_tile forms them
                // (same operations, same bits) - only this schedule pays for them, the fixed-factor one never sees this block.
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const double* Hc = H0 + (size_t)cur * HS * HS;
                double a = 0.0;
                for (int i = t; i < P; i += 64) {
                    const int pc = i - 6, sk = i - (3 + 3 * J);
                    const bool in_pose = use_pose && pc >= 0 && pc < n, shape_c = sbs > 0.0 && sk >= 0;
                    double Dii = hload(Hc + (size_t)i * HS + i), gi = hload(Hc + (size_t)P * HS + i);
                    Dii += in_pose ? sc2 * Pr[(size_t)pc * n + pc] : 0.0;
                    Dii += shape_c ? sbs * sbs : 0.0;
                    gi += in_pose ? gs * pri[2 + pc] : 0.0;
                    gi += shape_c ? sbs * (xc[3 + 4 * J + (sk >= 0 ? sk : 0)] * sbs) : 0.0;
                    const double dl = s_delta[i];
                    a += dl * (lambda * Dii * dl - gi);
                }
                pred_new = 0.5 * wave_sum(a);
            }
        }
        __syncthreads();
        if constexpr (TRI) {   // the factor is dead: the skeleton constants take its place
            prep_stage_constants<NTH>(dm, L, B, s_items, s_level);
            __syncthreads();
        }
        TPROBE(4);
        if constexpr (!TRI) {
            // Behind the back substitution's barrier three things run side by side on different waves: the retraction (wave 0, below), the
            // predicted decrease of the quadratic model (wave 1) and the joint positions of the new shape (waves 2 and 3) - round 6: the
            // last two sat on wave 0's path before and behind the retraction (2.7 k + 1.4 k clocks of the 84 k-clock pass).
            if (gain && t >= 64 && t < 128) {
                // 1/2 delta^T (lambda D delta - g), fixed butterfly.  g and the undamped diagonal D are what the assembly (sys_tile) formed
                // and left in LDS: the same values the lone wave used to rebuild from the reduced system with 170 global loads.
                double a = 0.0;
                for (int i = t - 64; i < P; i += 64) {
                    const double dl = s_delta[i];
                    a += dl * (lambda * s_gD[HS + i] * dl - s_gD[i]);
                }
                pred_new = 0.5 * wave_sum(a);
            }
            // (the shape coefficients of the new point: formed again here by the operation the retraction stores them with)
            if (t >= 128) prep_joint_positions(d, L, B, s_level, t - 128, [&](int k) { return xc[3 + 4 * J + k] + s_delta[3 + 3 * J + k]; });
        }
        // retraction (FakeQuaternionParameterization::Plus, :123-143): the new trial point goes to global memory for
        // the kernels that follow and straight into the skeleton scratch
        // (position and shape coefficients: lanes of the last wave in the 256-thread shape - on wave 0 they were two more divergent sections, each with
        // its own LDS round trip, in front of the quaternions)
        { const int tp = TRI ? t : t - 224; if (tp >= 0 && tp < 3) { const double v = xc[tp] + s_delta[tp]; xn[tp] = v; B[L.dv + tp] = v; } }
        { const int tw = TRI ? t : t - 232; if (tw >= 0 && tw < K) { const double v = xc[3 + 4 * J + tw] + s_delta[3 + 3 * J + tw]; xn[3 + 4 * J + tw] = v; B[L.w + tw] = v; } }
        if (t < J) {
            const double* dl = s_delta + 3 + 3 * t;
            const double* q = xc + 3 + 4 * t;
            // delta q = (sin|d| / |d| d, cos|d|): both factors are even power series in |d|, so up to |d| = 1/2 (a rotation of one radian in ONE
            // step) they are two interleaved Horner chains in z = |d|^2 - no square root, no division, no argument reduction (truncation
            // < 1e-21, rounding ~1 ulp; the library calls were 2.9 k clocks of the pass).  Larger steps take the library's functions.
            const double z = dl[0] * dl[0] + dl[1] * dl[1] + dl[2] * dl[2];
            double qo[4];
            if (z > 0.0) {
                // sin x / x = sum (-1)^k z^k / (2k+1)!,  cos x = sum (-1)^k z^k / (2k)!,  k = 0 .. 8
                double sdd = fma(z, 1.0 / 355687428096000.0, -1.0 / 1307674368000.0), a3 = fma(z, 1.0 / 20922789888000.0, -1.0 / 87178291200.0);
                sdd = fma(sdd, z, 1.0 / 6227020800.0);  a3 = fma(a3, z, 1.0 / 479001600.0);
                sdd = fma(sdd, z, -1.0 / 39916800.0);   a3 = fma(a3, z, -1.0 / 3628800.0);
                sdd = fma(sdd, z, 1.0 / 362880.0);      a3 = fma(a3, z, 1.0 / 40320.0);
                sdd = fma(sdd, z, -1.0 / 5040.0);       a3 = fma(a3, z, -1.0 / 720.0);
                sdd = fma(sdd, z, 1.0 / 120.0);         a3 = fma(a3, z, 1.0 / 24.0);
                sdd = fma(sdd, z, -1.0 / 6.0);          a3 = fma(a3, z, -0.5);
                sdd = fma(sdd, z, 1.0);                 a3 = fma(a3, z, 1.0);
                if (!(z < 0.25)) { const double nd = sqrt(z); sdd = sin(nd) / nd; a3 = cos(nd); }
                const double a0 = sdd * dl[0], a1 = sdd * dl[1], a2 = sdd * dl[2];
                qo[3] = a3 * q[3] - a0 * q[0] - a1 * q[1] - a2 * q[2];
                qo[0] = a3 * q[0] + a0 * q[3] + a1 * q[2] - a2 * q[1];
                qo[1] = a3 * q[1] + a1 * q[3] + a2 * q[0] - a0 * q[2];
                qo[2] = a3 * q[2] + a2 * q[3] + a0 * q[1] - a1 * q[0];
            } else {
                qo[0] = q[0]; qo[1] = q[1]; qo[2] = q[2]; qo[3] = q[3];
            }
            quat_to_rot(qo, B + L.rot + 9 * t);
            double* qn = xn + 3 + 4 * t;
            double* qs = s_qnew + 4 * t;
#pragma unroll
            for (int e = 0; e < 4; ++e) { qn[e] = qo[e]; qs[e] = qo[e]; }
        }
    } else {
        for (int e = t; e < xs; e += NTH) xn[e] = xc[e];
        for (int e = t; e < 4 * J; e += NTH) s_qnew[e] = xc[3 + e];
        if constexpr (TRI) { __syncthreads(); prep_stage_constants<NTH>(dm, L, B, s_items, s_level); __syncthreads(); }
        prep_set_state(d, L, B, xc + 3, xc + 3 + 4 * J, xc);
        if constexpr (!TRI) { if (t >= 128) prep_joint_positions(d, L, B, s_level, t - 128, [&](int k) { return xc[3 + 4 * J + k]; }); }
        if (gain) { lambda = fmin(lambda * nu, lm_max); nu *= 2.0; }
        else lambda = fmin(lambda * lm_up, lm_max);
    }
    if (t == 0 && role == 0) {
        ctl.lambda = lambda; ctl.try_valid = ok ? 1 : 0;
        // what the accept test of the trial point just made reads if no further solve follows (avt_decide.h)
        ctl.dec_cur_slot = cur; ctl.dec_try_valid = ok ? 1 : 0; ctl.dec_cost_cur = cost_cur; ctl.dec_lambda = lambda;
        ctl.nu = nu; ctl.dec_nu = nu;
        if (RIDE) { sp.next = 0; sp.n = fb.nspec; sp.ahead = (mode == SOLVE_FIRST ? 0 : sq_ahead) + nfold; }          // the speculative workgroups of this launch are making steps 0 .. nspec - 1
    }
    if (RIDE && t == 0 && role > 0) { sp.valid[role - 1] = ok ? 1 : 0; sp.lambda[role - 1] = lambda; }
    // (the predicted decrease is with the wave that formed it: wave 1 in the 256-thread shape, wave 0 in the 1024-thread one; 0 when the factorisation was refused)
    if (t == (TRI ? 0 : 64)) {
        if (role == 0) { ctl.pred = pred_new; ctl.dec_pred = pred_new; }
        else if (RIDE) sp.pred[role - 1] = pred_new;
    }
    if (RIDE && role > 0 && !ok) return;                     // (a refused speculative factorisation: the slot stays invalid)
    __syncthreads();
    TPROBE(5);
    // ---- d. skeleton tables of the new trial point ----------------------------------------------------
    if (!items_global) prep_items = prep_preload_items<NTH>(d, s_items, s_level);      // (staged items: the 1024-thread shape, skeletons without the per-thread table)
    prep_run<NTH, !TRI>(dm, L, B, s_items, s_level, s_qnew, prep_out, prep_items);      // (256-thread shape: the joint positions are made - waves 2 and 3, beside the retraction)
    TPROBE(6);
#ifdef AVT_TIMING
    __syncthreads();
    if (t == 0) { const double* pp = prep0 + (size_t)ntry * d.prep_size; fb.trace[(size_t)f * 64 + 62] = pp[d.prep_size - 1]; fb.trace[(size_t)f * 64 + 63] = pp[d.prep_size - 2]; }
#endif
}

static bool solve_big(const AvtDims& d) { return d.HS / 4 > 22; }      // more than 253 blocks: the 1024-thread triangular shape

// Matrix instructions of ONE factorisation (mf_rounds / mfg_rounds): every 4-pivot round but the last updates the lower-triangle tiles
// from the next panel's tile column on (rank-4 update, one v_mfma_f64_16x16x4_f64 per tile)
long long avt_solve_mfma_count(const AvtDims& d) {
    const int NR = (d.P + 3) >> 2, T = solve_big(d) ? (d.P + 16) >> 4 : 6;
    long long n = 0;
    for (int kb = 0; kb + 1 < NR; ++kb) {
        const int cb = kb >> 2, cbn = (kb & 3) == 3 ? std::min(cb + 1, T - 1) : cb;
        for (int c = cbn; c < T; ++c) n += T - c;
    }
    return n;
}

static size_t solve_lds_bytes(const AvtDims& d) {
    const int HS = d.HS, NB = HS / 4;
    const PrepLayout L = prep_layout(d.J, d.K, d.xsize);
    const size_t nblk = solve_big(d) ? (size_t)NB * (NB + 1) / 2 : (size_t)NB * NB;
    const size_t prep_bytes = sizeof(double) * (size_t)L.ndoubles + sizeof(int) * (2 * (size_t)L.nitems + 2 * AVT_MAX_JOINTS + 4);
    const size_t fixed = sizeof(double) * (((std::max(HS, 4 * d.J) + 3) & ~1) + HS + 2 + (solve_big(d) ? 0 : 2 * HS) + ((2 * d.xsize + 1) & ~1));
    const size_t factor = sizeof(double) * nblk * 18;
    return (solve_big(d) ? std::max(factor, prep_bytes) + sizeof(double) * MFG_PB_DOUBLES + fixed
                         : factor + sizeof(double) * MF_PB_DOUBLES + fixed + prep_bytes) + 64;
}

// (the accept test behind the LAST evaluation of an ICP iteration needs no reduction launch: avt_decide.h, inside k_lbs)
void launch_reduce(avt_ctx* c, int nframes) {
    const AvtDims& d = c->dm.d;
    if (c->fb.G >= 64) {      // few frames: several workgroups per pair (k_reduce_strip)
        // measured, one frame: 2 / 4 / 8 / 16 strips 0.582 / 0.556 / 0.543 / 0.560 ms per step; eight frames: 0.817 / 0.778 / 0.800 / 0.915
        if (nframes == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_reduce_strip<8>), dim3(8 * d.NPAIR, nframes), dim3(1024), 0, c->cur_stream, c->dm, c->fb);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_reduce_strip<4>), dim3(4 * d.NPAIR, nframes), dim3(1024), 0, c->cur_stream, c->dm, c->fb);
    } else {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_reduce<1>), dim3(d.NPAIR, nframes), dim3(256), 0, c->cur_stream, c->dm, c->fb);
    }
}

// the model has the dimensions k_solve<.., SM = true> is compiled for (solve_dims_smpl_set)
static bool solve_dims_smpl(const AvtDims& d) {
    AvtDims e = d;
    solve_dims_smpl_set(e);
    return d.ncomps > 0 && d.J == e.J && d.K == e.K && d.P == e.P && d.HS == e.HS && d.xsize == e.xsize && d.NT == e.NT && d.NPAIR == e.NPAIR &&
           d.ndims == e.ndims && d.prep_size == e.prep_size;
}

template <int NTH, bool TRI>
static void launch_solve_shape(avt_ctx* c, int nframes, int mode, size_t lds) {
    if constexpr (NTH == 256 && !TRI) {
        if (c->tun.literal_dims && solve_dims_smpl(c->dm.d) && (mode == SOLVE_FIRST || mode == SOLVE_NORMAL)) {
            if (mode == SOLVE_FIRST) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_solve<256, false, SOLVE_FIRST, 0, true>), dim3(nframes), dim3(256), lds, c->cur_stream, c->dm, c->fb);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_solve<256, false, SOLVE_NORMAL, 0, true>), dim3(nframes), dim3(256), lds, c->cur_stream, c->dm, c->fb);
            return;
        }
    }
    switch (mode) {
        case SOLVE_INIT: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_solve<NTH, TRI, SOLVE_INIT>), dim3(nframes), dim3(NTH), lds, c->cur_stream, c->dm, c->fb); break;
        case SOLVE_FIRST: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_solve<NTH, TRI, SOLVE_FIRST>), dim3(nframes), dim3(NTH), lds, c->cur_stream, c->dm, c->fb); break;
        case SOLVE_DECIDE: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_solve<NTH, TRI, SOLVE_DECIDE>), dim3(nframes), dim3(NTH), sizeof(double) * 2 * c->dm.d.xsize + 64, c->cur_stream, c->dm, c->fb); break;
        default: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_solve<NTH, TRI, SOLVE_NORMAL>), dim3(nframes), dim3(NTH), lds, c->cur_stream, c->dm, c->fb); break;
    }
}

// few frames, 256-thread solves: the reduction of the trial point's system rides in the solve's launch (no launch_reduce in front)
// (while every workgroup of the launch - 1 + 4 NPAIR per frame, one per CU with the solver's LDS request - is resident at once:
// three SMPL frames on 256 CUs; six frames in two rounds measured 0.751 against 0.689 ms with the reduction as its own launch)
// speculative solver workgroups per frame (AVT_NSPEC, default 2) next to the solver, as many as leave the grid resident
// Residency is a matter of speed, not of correctness (see the wait in k_solve).  AVT_RIDE_SIZING=groups sizes the shapes by the
// frame groups that run side by side (two / three frames: one group each) instead of by the one launch.
static int ride_concurrency(const avt_ctx* c) { return c->tun.ride_sizing_groups ? std::max(1, c->concurrent_groups) : 1; }
// how many queued speculative steps have their cost evaluated beside the trial point (k_eval's spec-cost workgroups exist in the six-tile shape)
static int ride_spec_cost(const avt_ctx* c) { return (c->dm.d.J == 24 && c->dm.d.K == 10 && c->tun.spec_cost > 0) ? 1 : 0; }      // (one step ahead: two and four measured the same over twelve frames and lose on frames that accept)
static int ride_nspec(const avt_ctx* c, int nframes, int strips) {
    // over twelve frames, one frame each: 0 / 2 / 3 / 4 speculative workgroups 0.5045 / 0.4305 / 0.4217 / 0.4166 ms (rejections come
    // in runs of up to five).  Frame batches gain nothing from them - a launch lasts as long as its slowest frame, and with four
    // frames or more some frame always needs a full solve (4 / 8 / 16 / 64 frames: 0.639 / 0.712 / 0.786 / 1.247 ms without,
    // 0.643 / 0.712 / 0.793 / 1.317 ms with four speculative workgroups per frame) - so only the riding shapes have them.
    int want = std::max(0, std::min(AVT_MAX_SPEC, c->tun.nspec));
    while (want > 0 && ride_concurrency(c) * nframes * (1 + want + std::min(want, ride_spec_cost(c)) + strips * c->dm.d.NPAIR) > c->num_cus) --want;
    return want;
}
static int ride_strips(const avt_ctx* c, int nframes) {      // 8 strips per pair while the whole grid is resident (one SMPL frame), else 4, else none
    if (c->fb.G < 64 || solve_big(c->dm.d) || !c->tun.ride || c->fb.use_moments) return 0;      // (moment form: k_assemble writes the system, nothing to reduce)
    const int want = c->tun.ride_strips ? c->tun.ride_strips : 8;      // the two instantiated shapes (avt_ctx_set_tuning admits 4 and 8 only)
    for (int s = want; s >= 4; s -= 4) if (ride_concurrency(c) * nframes * (1 + s * c->dm.d.NPAIR) <= c->num_cus) return s;
    return 0;
}
bool avt_solve_rides(const avt_ctx* c, int nframes) { return ride_strips(c, nframes) != 0; }
int avt_solve_nspec(const avt_ctx* c, int nframes) {
    const int rs = ride_strips(c, nframes);
    return rs ? std::min(ride_nspec(c, nframes, rs), ride_spec_cost(c)) : 0;
}

void launch_solve(avt_ctx* c, int nframes, int mode, int seq) {
    const AvtDims& d = c->dm.d;
    const int rs = (mode != SOLVE_INIT && mode != SOLVE_DECIDE) ? ride_strips(c, nframes) : 0;
    c->fb.nspec = 0; c->fb.nspec_cost = 0; c->fb.seq = seq;
    if (rs) {
        c->fb.nspec = ride_nspec(c, nframes, rs);
        c->fb.nspec_cost = std::min(c->fb.nspec, ride_spec_cost(c));
        const dim3 grid(nframes, 1 + c->fb.nspec + rs * d.NPAIR + c->fb.nspec_cost);
#define AVT_RIDE(M, S, SMD) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_solve<256, false, M, S, SMD>), grid, dim3(256), solve_lds_bytes(d), c->cur_stream, c->dm, c->fb)
        if (c->tun.literal_dims && solve_dims_smpl(d)) {
            if (mode == SOLVE_FIRST) { if (rs == 8) AVT_RIDE(SOLVE_FIRST, 8, true); else AVT_RIDE(SOLVE_FIRST, 4, true); }
            else { if (rs == 8) AVT_RIDE(SOLVE_NORMAL, 8, true); else AVT_RIDE(SOLVE_NORMAL, 4, true); }
        } else {
            if (mode == SOLVE_FIRST) { if (rs == 8) AVT_RIDE(SOLVE_FIRST, 8, false); else AVT_RIDE(SOLVE_FIRST, 4, false); }
            else { if (rs == 8) AVT_RIDE(SOLVE_NORMAL, 8, false); else AVT_RIDE(SOLVE_NORMAL, 4, false); }
        }
#undef AVT_RIDE
        return;
    }
    if (solve_big(d)) launch_solve_shape<1024, true>(c, nframes, mode, solve_lds_bytes(d));
    else launch_solve_shape<256, false>(c, nframes, mode, solve_lds_bytes(d));
}

template <int NTH, bool TRI>
static int solve_attr() {
    const int cap = 160 * 1024 - 512;
    return hipFuncSetAttribute((const void*)k_solve<NTH, TRI, SOLVE_INIT>, hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess ||
           hipFuncSetAttribute((const void*)k_solve<NTH, TRI, SOLVE_FIRST>, hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess ||
           hipFuncSetAttribute((const void*)k_solve<NTH, TRI, SOLVE_NORMAL>, hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess;
}

int avt_solve_set_attributes() {
    const int cap = 160 * 1024 - 512;
    return solve_attr<256, false>() || solve_attr<1024, true>() ||
           hipFuncSetAttribute((const void*)k_solve<256, false, SOLVE_FIRST, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess ||
           hipFuncSetAttribute((const void*)k_solve<256, false, SOLVE_NORMAL, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess ||
           hipFuncSetAttribute((const void*)k_solve<256, false, SOLVE_FIRST, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess ||
           hipFuncSetAttribute((const void*)k_solve<256, false, SOLVE_NORMAL, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess ||
           hipFuncSetAttribute((const void*)k_solve<256, false, SOLVE_FIRST, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess ||
           hipFuncSetAttribute((const void*)k_solve<256, false, SOLVE_NORMAL, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess ||
           hipFuncSetAttribute((const void*)k_solve<256, false, SOLVE_FIRST, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess ||
           hipFuncSetAttribute((const void*)k_solve<256, false, SOLVE_NORMAL, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess ||
           hipFuncSetAttribute((const void*)k_solve<256, false, SOLVE_FIRST, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess ||
           hipFuncSetAttribute((const void*)k_solve<256, false, SOLVE_NORMAL, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess;
}
