// avt_lm.hip — the Gauss-Newton / Levenberg-Marquardt step kernels (gfx950, wave64):
//   k_reduce : fixed-order reduction of k_eval's partial MFMA tiles into the dense symmetric data-term system
//              [J|r]^T W [J|r] of the trial point;
//   k_solve  : one workgroup per frame — LM accept/reject, prior assembly (AvatarOptimizer.cpp:647-726,
//              :1457-1458), damped LDL^T solve held in registers, quaternion retraction
//              (FakeQuaternionParameterization::Plus, :123-143) and the skeleton tables of the next trial point
//              (PrepareForEvaluation, :283-325).  The whole inner loop runs without host synchronisation.
//
// Everything in k_solve is latency-bound (an 85-long pivot chain), so it is organised around the dependency
// chain: no divide / sqrt on the chain (v_rcp_f64 + cubic Newton), one barrier per 4 pivots, the diagonal 4x4
// block factored redundantly by every lane instead of being published, back-substitution by cross-lane
// v_readlane instead of LDS round trips, and no global load inside any sequential loop.
#include "avt_device.h"

#ifdef AVT_TIMING
#define TPROBE(i) do { if (threadIdx.x == 0) fb.trace[(size_t)(blockIdx.x + fb.f0) * 64 + 40 + (i)] = (double)clock64(); } while (0)
#else
#define TPROBE(i) do {} while (0)
#endif

// -------------------------------------------------------------------------------------------------
// skeleton tables of a state x=(p,q,w) -> prep block in global memory.  Called by all 256 threads.
// -------------------------------------------------------------------------------------------------
struct PrepScratch {
    double rot[AVT_MAX_JOINTS * 9], Rw[AVT_MAX_JOINTS * 9], o[AVT_MAX_JOINTS * 3], jp[AVT_MAX_JOINTS * 3];
    double H[AVT_MAX_JOINTS * 3 * AVT_MAX_SHAPE], Sp[AVT_MAX_JOINTS * 3 * AVT_MAX_SHAPE];
    double w[AVT_MAX_SHAPE], p[3];
    int parent[AVT_MAX_JOINTS], level[AVT_MAX_JOINTS + 2];
    unsigned short items[AVT_MAX_JOINTS * (12 + 3 * AVT_MAX_SHAPE)];
};

__device__ void compute_prep(const DeviceModel& dm, const double* __restrict__ x, double* __restrict__ prep, PrepScratch& s) {
    const AvtDims d = dm.d;
    const int J = d.J, K = d.K, t = threadIdx.x;
    const double* q = x + 3;
    const double* w = x + 3 + 4 * J;
    // everything that comes from global memory is requested up front, in one round trip
    if (t < J) { s.parent[t] = dm.parent[t]; quat_to_rot(q + 4 * t, s.rot + 9 * t); }
    if (t < 3) s.p[t] = x[t];
    if (t < K) s.w[t] = w[t];
    if (t <= d.nlevels) s.level[t] = dm.fk_level_off[t];
    for (int e = t; e < 3 * J * K; e += 256) s.Sp[e] = dm.Sp[e];
    // CalcShape (AvatarOptimizer.cpp:249-281): jointPosInit = base + jointShapeReg*w
    if (t < 3 * J) {
        double a = 0.0;
        for (int k = 0; k < K; ++k) a += dm.jsr[(size_t)t * K + k] * w[k];
        s.jp[t] = dm.jsr_base[t] + a;
    }
    // per-level work items (joint, entry), at most one per lane and level for SMPL
    const int per = 12 + 3 * K;
    const int nitems = J * per;
    for (int e = t; e < nitems; e += 256) s.items[e] = (unsigned short)dm.fk_items[e];
    __syncthreads();
#ifdef AVT_TIMING
    if (threadIdx.x == 0) prep[d.prep_size - 1] = (double)clock64();
#endif
    // one tree level per barrier: world rotation/origin (:303-315) and H[j] = R(-1,parent j) Sp[j] + H[parent j]
    // (:318-324) of every joint of the level in parallel
    for (int L = 0; L < d.nlevels; ++L) {
        const int lo = s.level[L], hi = s.level[L + 1];
        for (int idx = lo + t; idx < hi; idx += 256) {
            const int item = s.items[idx];
            const int j = item >> 8, e = item & 0xff;
            const int pa = s.parent[j];
            if (e < 12) {
                if (j == 0) {
                    if (e < 9) s.Rw[e] = s.rot[e];
                    else s.o[e - 9] = s.p[e - 9];
                } else {
                    const double* Rp = s.Rw + 9 * pa;
                    if (e < 9) {
                        const int r = e / 3, c = e % 3;
                        s.Rw[9 * j + e] = Rp[3 * r] * s.rot[9 * j + c] + Rp[3 * r + 1] * s.rot[9 * j + 3 + c] + Rp[3 * r + 2] * s.rot[9 * j + 6 + c];
                    } else {
                        const int r = e - 9;
                        const double d0 = s.jp[3 * j] - s.jp[3 * pa], d1 = s.jp[3 * j + 1] - s.jp[3 * pa + 1], d2 = s.jp[3 * j + 2] - s.jp[3 * pa + 2];
                        s.o[3 * j + r] = s.o[3 * pa + r] + (Rp[3 * r] * d0 + Rp[3 * r + 1] * d1 + Rp[3 * r + 2] * d2);
                    }
                }
            } else {
                const int e2 = e - 12, r = e2 / K, k = e2 - r * K;
                double v = 0.0;
                if (j > 0) {
                    const double* Rp = s.Rw + 9 * pa;
                    const double* Sp = s.Sp + j * 3 * K;
                    v = (Rp[3 * r] * Sp[k] + Rp[3 * r + 1] * Sp[K + k] + Rp[3 * r + 2] * Sp[2 * K + k]) + s.H[pa * 3 * K + e2];
                }
                s.H[j * 3 * K + e2] = v;
            }
        }
        __syncthreads();
    }
#ifdef AVT_TIMING
    if (threadIdx.x == 0) prep[d.prep_size - 2] = (double)clock64();
#endif
    const double off0 = s.jp[0], off1 = s.jp[1], off2 = s.jp[2];
    for (int e = t; e < 9 * J; e += 256) prep[prep_off_Rw(d) + e] = s.Rw[e];
    for (int e = t; e < 3 * J; e += 256) {
        prep[prep_off_o(d) + e] = s.o[e];
        const int c = e % 3;
        prep[prep_off_Jh(d) + e] = s.jp[e] - (c == 0 ? off0 : (c == 1 ? off1 : off2));   // root at origin (:270-272)
    }
    for (int e = t; e < 3 * J * K; e += 256) {  // G[j] = H[j] - Rw[j]*S[j]  (shape block of :568-580)
        const int j = e / (3 * K), r = (e / K) % 3, k = e % K;
        const double* Rj = s.Rw + 9 * j;
        const double* S = dm.S + (size_t)j * 3 * K;
        prep[prep_off_G(d) + e] = s.H[e] - (Rj[3 * r] * S[k] + Rj[3 * r + 1] * S[K + k] + Rj[3 * r + 2] * S[2 * K + k]);
    }
    for (int e = t; e < 4 * J; e += 256) prep[prep_off_q(d) + e] = q[e];
    if (t < K) prep[prep_off_w(d) + t] = s.w[t];
    if (t < 3) prep[prep_off_off(d) + t] = (t == 0 ? off0 : (t == 1 ? off1 : off2));
}

// =================================================================================================
// k_reduce.  grid (NPAIR, nframes), block 256.
//   Hraw[f][try][r][c] = sum_g partial[f][g][pair][e], g ascending (deterministic); the tile is written to both
//   triangles of the dense (HS x HS) symmetric block; row/column P carry J^T r and sum c|r|^2.
//   (The GMM pose prior of the trial point is evaluated by extra workgroups of k_eval, avt_prior.h.)
// =================================================================================================
__global__ __launch_bounds__(256) void k_reduce(DeviceModel dm, FrameBuffers fb) {
    const AvtDims d = dm.d;
    const int f = blockIdx.y + fb.f0, t = threadIdx.x, NPAIR = d.NPAIR, NT = d.NT, P = d.P, HS = d.HS;
    const int try_slot = 1 - fb.ctl[f].cur_slot;
    {
        int p = blockIdx.x, ti = 0;
        while (p >= NT - ti) { p -= NT - ti; ++ti; }
        const int tj = ti + p;
        const double* part = fb.partial + ((size_t)f * fb.G * NPAIR + blockIdx.x) * 256 + t;
        // 16 independent loads in flight per lane, summed in ascending g (same order as a plain loop)
        const size_t st = (size_t)NPAIR * 256;
        double a = 0.0;
        int g = 0;
        for (; g + 16 <= fb.G; g += 16) {
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = __builtin_nontemporal_load(part + (size_t)(g + u) * st);
#pragma unroll
            for (int u = 0; u < 16; ++u) a += v[u];
        }
        for (; g < fb.G; ++g) a += part[(size_t)g * st];
        const int r = ti * 16 + ((t >> 4) & 3) + 4 * (t >> 6), c = tj * 16 + (t & 15);
        if (r <= P && c <= P) {
            double* H = fb.Hraw + ((size_t)f * 2 + try_slot) * HS * HS;
            H[(size_t)r * HS + c] = a;
            if (ti != tj) H[(size_t)c * HS + r] = a;
        }
    }
}

typedef double d2 __attribute__((ext_vector_type(2)));

// reciprocal off the slow path: v_rcp_f64 (~2^-26 relative) + one cubic Newton step (error e^3)
__device__ __forceinline__ double fast_rcp(double d) {
    const double r0 = __builtin_amdgcn_rcp(d);
    const double e = fma(-d, r0, 1.0);
    const double t2 = fma(e, e, e);
    return fma(r0, t2, r0);
}

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// Back substitution L^T delta = w for a compile-time size, fully unrolled: lane indices of the v_readlane broadcasts
// and all LDS offsets are immediates, so the factor rows are fetched far ahead of the 85-step dependency chain.
template <int PP>
__device__ __forceinline__ void backsub_unrolled(const double* __restrict__ Lblk, int NB, int t, double* __restrict__ s_delta) {
    // element (i, l) of the unit-lower factor lives at Lblk[((l>>2)*NB + (i>>2))*18 + (i&3)*4 + (l&3)]: the lane-dependent
    // part (column l = t or t+64) is a base pointer, the row-dependent part a compile-time offset
    const double* col0 = Lblk + ((size_t)(t >> 2) * NB) * 18 + (t & 3);
    const double* col1 = Lblk + ((size_t)((t + 64) >> 2) * NB) * 18 + (t & 3);
    constexpr int PO = (PP >> 2) * 18 + (PP & 3) * 4;
    const double w0 = (t < PP) ? col0[PO] : 0.0;
    const double w1 = (t + 64 < PP) ? col1[PO] : 0.0;
    double acc0 = 0.0, acc1 = 0.0, dl0 = 0.0, dl1 = 0.0;
#pragma unroll
    for (int i = PP - 1; i >= 0; --i) {
        const int io = (i >> 2) * 18 + (i & 3) * 4;
        const double c0 = (t < i) ? col0[io] : 0.0;
        double di;
        if (i >= 64) {
            const double c1 = (t + 64 < i) ? col1[io] : 0.0;
            di = readlane_f64(w1, i - 64) - readlane_f64(acc1, i - 64);
            if (t == i - 64) dl1 = di;
            acc1 = fma(c1, di, acc1);
        } else {
            di = readlane_f64(w0, i) - readlane_f64(acc0, i);
            if (t == i) dl0 = di;
        }
        acc0 = fma(c0, di, acc0);
    }
    if (t < PP) s_delta[t] = dl0;
    if (t + 64 < PP) s_delta[t + 64] = dl1;
}

// =================================================================================================
// k_solve.  grid (nframes), block 256.
// =================================================================================================
__global__ __launch_bounds__(256) void k_solve(DeviceModel dm, FrameBuffers fb, int mode, double lm_up, double lm_down,
                                               double lm_min, double lm_max) {
    const AvtDims d = dm.d;
    const int J = d.J, K = d.K, P = d.P, HS = d.HS;
    const int f = blockIdx.x + fb.f0, t = threadIdx.x;
    AvtFrameCtl& ctl = fb.ctl[f];
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int NBk = HS >> 2;                                // 4-row blocks covering rows 0..P (22 for SMPL)
    // unit-lower factor, block layout [pivot block kb][row block bi][18]: a 4x4 block is 16 doubles + 2 of padding
    // (144 B), so lanes reading different blocks spread over the LDS banks; row P carries D^-1 L^-1 rhs
    double* Lblk = (double*)smem;
    double* s_W = Lblk + (size_t)NBk * NBk * 18;            // [NB][18]  W = A_panel Ld^-T of the current pivot block
    double* s_D = s_W + (size_t)NBk * 18;                   // [18]      updated diagonal block of the next pivot block
    double* s_delta = s_D + 18;                             // [HS]
    PrepScratch* ps = (PrepScratch*)(s_delta + HS + 2);
    __shared__ int s_fail;
    const int xs = d.xsize;
    double* x0 = fb.x + ((size_t)f * 2) * xs;

    if (mode == SOLVE_INIT) {
        // trial point := current point; sum the constant part of the data cost
        const int cur = ctl.cur_slot, tr = 1 - cur;
        for (int e = t; e < xs; e += 256) x0[(size_t)tr * xs + e] = x0[(size_t)cur * xs + e];
        double a = 0.0;
        if (t < 64) {
            for (int e = t; e < fb.const_used; e += 64) a += fb.const_part[(size_t)f * fb.const_blocks + e];
            a = wave_sum(a);
        }
        __syncthreads();
        if (t == 0) { ctl.cost_const = 0.5 * a; ctl.try_valid = 1; }
        compute_prep(dm, x0 + (size_t)tr * xs, fb.prep + ((size_t)f * 2 + tr) * d.prep_size, *ps);
        return;
    }
    TPROBE(0);
    // ---- a. objective of the trial point + LM decision (uniform work, done redundantly by every lane) -----
    const int cur0 = ctl.cur_slot, try0 = 1 - cur0;
    const int try_valid = ctl.try_valid, comp_cur0 = ctl.comp_cur;
    const double sbp = ctl.sbp, sbs = ctl.sbs, cost_cur0 = ctl.cost_cur, cost_const = ctl.cost_const;
    double lambda = ctl.lambda;
    const double* Htry = fb.Hraw + ((size_t)f * 2 + try0) * HS * HS;
    const double* xt = x0 + (size_t)try0 * xs;
    double cost = 0.5 * Htry[(size_t)P * HS + P] + cost_const;
    int comp_try = -1;
    if (sbp > 0.0 && d.ncomps > 0) {
        // best component: strict '<' in ascending component order (GaussianMixture.cpp:103)
        double best = 1.7976931348623157e308;
        for (int c = 0; c < d.ncomps; ++c) {
            const double pr = fb.prior[(((size_t)f * 2 + try0) * AVT_MAX_COMPS + c) * AVT_PRIOR_STRIDE];
            if (pr < best) { best = pr; comp_try = c; }
        }
        cost += 0.5 * sbp * sbp * best;
    }
    if (sbs > 0.0) {
        double a = 0.0;
        for (int k = 0; k < K; ++k) { const double r = xt[3 + 4 * J + k] * sbs; a += r * r; }
        cost += 0.5 * a;
    }
    int cur = cur0;
    bool accepted = false;
    if (mode == SOLVE_FIRST) {
        accepted = true;
        cur = try0;
    } else if (try_valid) {
        if (cost < cost_cur0) { accepted = true; cur = try0; lambda = fmax(lambda * lm_down, lm_min); }
        else lambda = fmin(lambda * lm_up, lm_max);
    }
    const double cost_cur = accepted ? cost : cost_cur0;
    const int comp = accepted ? comp_try : comp_cur0;
    __syncthreads();   // every lane has read the control block before lane 0 rewrites it
    if (t == 0) {
        ctl.cur_slot = cur;
        ctl.cost_cur = cost_cur;
        ctl.comp_cur = comp;
        int it = ctl.gn_iterations;
        if (mode == SOLVE_FIRST) ctl.cost_initial = cost;
        else { it += 1; ctl.gn_iterations = it; if (accepted) ctl.accepted += 1; }
        if (it < 40) fb.trace[(size_t)f * 64 + it] = cost_cur;
        if (mode == SOLVE_LAST) ctl.lambda = lambda;
    }
    if (mode == SOLVE_LAST) return;
    TPROBE(1);

    // ---- b. the damped system of the current point, straight into registers --------------------------------
    // Thread t owns the 4x4 block (bi >= bj) of the bordered (P+1)x(P+1) matrix [[H + lambda diag H, .],[-g^T, .]]
    // (row P carries the rhs so D^-1 L^-1 (-g) falls out of the factorisation as row P of the unit-lower factor).
    const int NB = HS >> 2;
    int bi = -1, bj = -1;
    if (t < NB * (NB + 1) / 2) {
        int r0 = 0, rem = t;
        while (rem > r0) { rem -= r0 + 1; ++r0; }
        bi = r0; bj = rem;
    }
    const double* Hc = fb.Hraw + ((size_t)f * 2 + cur) * HS * HS;
    const double* xc = x0 + (size_t)cur * xs;
    const double* pri = fb.prior + (((size_t)f * 2 + cur) * AVT_MAX_COMPS + (comp >= 0 ? comp : 0)) * AVT_PRIOR_STRIDE;
    const bool use_pose = sbp > 0.0 && d.ncomps > 0 && comp >= 0;
    const int n = d.ndims;
    const double sc = 0.707106781186548 * sbp;          // literal constant (AvatarOptimizer.cpp:684)
    const double sc2 = sc * sc;
    const double gs = sc * sbp * 0.7071067811865476;    // J^T r = sc*sbp*sqrt(1/2) * Prec (x - mu)
    const double* Pr = dm.prior_prec + (size_t)(comp >= 0 ? comp : 0) * n * n;
    double a4[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int row = 4 * bi + r, col = 4 * bj + c;
            double v = (row == col) ? 1.0 : 0.0;
            if (bi >= 0 && row <= P && col < P) {
                v = Hc[(size_t)row * HS + col];
                const int pc = col - 6, sk = col - (3 + 3 * J);
                if (row < P) {
                    const int pr_ = row - 6;
                    if (use_pose && pr_ >= 0 && pr_ < n && pc >= 0 && pc < n) v += sc2 * Pr[(size_t)pr_ * n + pc];
                    if (row == col) {
                        if (sbs > 0.0 && sk >= 0) v += sbs * sbs;
                        v += lambda * v;
                    }
                } else {  // rhs row: -(J^T r) including the priors
                    if (use_pose && pc >= 0 && pc < n) v += gs * pri[2 + pc];
                    if (sbs > 0.0 && sk >= 0) v += sbs * (xc[3 + 4 * J + sk] * sbs);
                    v = -v;
                }
            }
            a4[r][c] = v;
        }
    TPROBE(2);

    // ---- c. register-blocked LDL^T, four pivots per round, two barriers per round -----------------------------------
    //  (1) the lanes owning the pivot block column (bj == kb) read the updated diagonal block, factor it
    //      (D = Ld diag(d) Ld^T), solve their own 4x4 block W = A Ld^-T, L = W diag(d)^-1 and publish W and L;
    //  (2) every trailing lane (bj > kb) reads W of its row block and L of its column block: A -= W L^T;
    //      the owner of the next diagonal block publishes it.
    // Nothing is recomputed: per round a trailing lane issues 16 LDS reads and 64 FMAs.
    typedef double d2v __attribute__((ext_vector_type(2)));
    if (t == 0) s_fail = 0;
    if (bi == 0 && bj == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) s_D[r * 4 + c] = a4[r][c];
    }
#ifdef AVT_TIMING
    long long lacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long llast = clock64();
#define LPROBE(k) do { const long long _n = clock64(); lacc[k] += _n - llast; llast = _n; } while (0)
#else
#define LPROBE(k) do {} while (0)
#endif
    bool fail = false;
    for (int kb = 0; kb < NB; ++kb) {
        __syncthreads();                                    // B1: diagonal block kb is visible
        LPROBE(0);
        if (bj == kb) {
            const d2v* Dq = (const d2v*)s_D;
            const d2v q0 = Dq[0], q2 = Dq[2], q4 = Dq[4], q5 = Dq[5], q6 = Dq[6], q7 = Dq[7];
            const double D00 = q0.x, D10 = q2.x;
#ifdef AVT_TIMING
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); LPROBE(5);
#endif
            double D11 = q2.y, D20 = q4.x, D21 = q4.y, D22 = q5.x, D30 = q6.x, D31 = q6.y, D32 = q7.x, D33 = q7.y;
            const bool real1 = 4 * kb + 1 < P, real2 = 4 * kb + 2 < P, real3 = 4 * kb + 3 < P;   // 4*kb < P always
            const double P0 = D00;
            const double r0 = fast_rcp(D00);
            const double l10 = D10 * r0, l20 = D20 * r0, l30 = D30 * r0;
            D11 = fma(-l10, D10, D11); D21 = fma(-l20, D10, D21); D31 = fma(-l30, D10, D31);
            D22 = fma(-l20, D20, D22); D32 = fma(-l30, D20, D32); D33 = fma(-l30, D30, D33);
            const double P1 = D11;
            const double r1 = fast_rcp(D11);
            const double l21 = D21 * r1, l31 = D31 * r1;
            D22 = fma(-l21, D21, D22); D32 = fma(-l31, D21, D32); D33 = fma(-l31, D31, D33);
            const double P2 = D22;
            const double r2 = fast_rcp(D22);
            const double l32 = D32 * r2;
            D33 = fma(-l32, D32, D33);
            const double P3 = D33;
            const double r3 = fast_rcp(D33);
            const int bad = (int)!(P0 > 0.0) | ((int)real1 & (int)!(P1 > 0.0)) | ((int)real2 & (int)!(P2 > 0.0)) | ((int)real3 & (int)!(P3 > 0.0));
            if (bad) s_fail = 1;
#ifdef AVT_TIMING
            { double keep = r3; asm volatile("" : "+v"(keep)); LPROBE(6); }
#endif
            d2v* Wo = (d2v*)(s_W + (size_t)bi * 18);
            d2v* Lo = (d2v*)(Lblk + ((size_t)kb * NB + bi) * 18);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double w0 = a4[r][0];
                const double w1 = fma(-w0, l10, a4[r][1]);
                const double w2 = fma(-w1, l21, fma(-w0, l20, a4[r][2]));
                const double w3 = fma(-w2, l32, fma(-w1, l31, fma(-w0, l30, a4[r][3])));
                Wo[2 * r] = (d2v){w0, w1}; Wo[2 * r + 1] = (d2v){w2, w3};
                Lo[2 * r] = (d2v){w0 * r0, w1 * r1}; Lo[2 * r + 1] = (d2v){w2 * r2, w3 * r3};
            }
#ifdef AVT_TIMING
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); LPROBE(7);
#endif
        }
        LPROBE(1);
        __syncthreads();                                    // B2: W and L of pivot block kb are visible
        LPROBE(2);
        if (s_fail) { fail = true; break; }
        if (bj > kb) {
            const d2v* Wi = (const d2v*)(s_W + (size_t)bi * 18);
            const d2v* Lj = (const d2v*)(Lblk + ((size_t)kb * NB + bj) * 18);
            d2v wv[4][2], lv[4][2];
#pragma unroll
            for (int r = 0; r < 4; ++r) { wv[r][0] = Wi[2 * r]; wv[r][1] = Wi[2 * r + 1]; lv[r][0] = Lj[2 * r]; lv[r][1] = Lj[2 * r + 1]; }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    double v = a4[r][cc];
                    v = fma(-wv[r][0].x, lv[cc][0].x, v);
                    v = fma(-wv[r][0].y, lv[cc][0].y, v);
                    v = fma(-wv[r][1].x, lv[cc][1].x, v);
                    v = fma(-wv[r][1].y, lv[cc][1].y, v);
                    a4[r][cc] = v;
                }
            LPROBE(3);
            if (bi == kb + 1 && bj == kb + 1) {             // publish the next diagonal block
                d2v* Do = (d2v*)s_D;
#pragma unroll
                for (int r = 0; r < 4; ++r) { Do[2 * r] = (d2v){a4[r][0], a4[r][1]}; Do[2 * r + 1] = (d2v){a4[r][2], a4[r][3]}; }
            }
        }
    }
#ifdef AVT_TIMING
    if (t == 251) { for (int k = 0; k < 5; ++k) fb.trace[(size_t)f * 64 + 56 + k] = (double)lacc[k]; for (int k = 5; k < 8; ++k) fb.trace[(size_t)f * 64 + 32 + k] = (double)lacc[k]; }
#endif
    __syncthreads();
    TPROBE(3);
    const bool ok = !fail;
    const int ntry = 1 - cur;
    double* xn = x0 + (size_t)ntry * xs;
    if (ok) {
        // ---- back substitution L^T delta = w (w = row P of Lf) by wave 0.  Lane l keeps w[l], w[l+64] and the
        // running sums acc[l] = sum_{k>i} L[k][l] delta_k in registers; values cross lanes by v_readlane.
        if (t < 64) {
            if (P == 85) backsub_unrolled<85>(Lblk, NB, t, s_delta);
            else {
                auto Lat = [&](int i, int l) { return Lblk[((size_t)(l >> 2) * NB + (i >> 2)) * 18 + (i & 3) * 4 + (l & 3)]; };
                const double w0 = (t < P) ? Lat(P, t) : 0.0;
                const double w1 = (t + 64 < P) ? Lat(P, t + 64) : 0.0;
                double acc0 = 0.0, acc1 = 0.0, dl0 = 0.0, dl1 = 0.0;
                for (int i = P - 1; i >= 0; --i) {
                    const double c0 = (t < i) ? Lat(i, t) : 0.0;
                    const double c1 = (t + 64 < i) ? Lat(i, t + 64) : 0.0;
                    double di;
                    if (i < 64) { di = readlane_f64(w0, i) - readlane_f64(acc0, i); if (t == i) dl0 = di; }
                    else { di = readlane_f64(w1, i - 64) - readlane_f64(acc1, i - 64); if (t == i - 64) dl1 = di; }
                    acc0 = fma(c0, di, acc0);
                    acc1 = fma(c1, di, acc1);
                }
                if (t < P) s_delta[t] = dl0;
                if (t + 64 < P) s_delta[t + 64] = dl1;
            }
        }
        __syncthreads();
        TPROBE(4);
        // retraction (FakeQuaternionParameterization::Plus, :123-143)
        if (t < 3) xn[t] = xc[t] + s_delta[t];
        if (t < K) xn[3 + 4 * J + t] = xc[3 + 4 * J + t] + s_delta[3 + 3 * J + t];
        if (t < J) {
            const double* dl = s_delta + 3 + 3 * t;
            const double* q = xc + 3 + 4 * t;
            const double nd = sqrt(dl[0] * dl[0] + dl[1] * dl[1] + dl[2] * dl[2]);
            double* qo = xn + 3 + 4 * t;
            if (nd > 0.0) {
                const double sdd = sin(nd) / nd;
                const double a0 = sdd * dl[0], a1 = sdd * dl[1], a2 = sdd * dl[2], a3 = cos(nd);
                qo[3] = a3 * q[3] - a0 * q[0] - a1 * q[1] - a2 * q[2];
                qo[0] = a3 * q[0] + a0 * q[3] + a1 * q[2] - a2 * q[1];
                qo[1] = a3 * q[1] + a1 * q[3] + a2 * q[0] - a0 * q[2];
                qo[2] = a3 * q[2] + a2 * q[3] + a0 * q[1] - a1 * q[0];
            } else {
                qo[0] = q[0]; qo[1] = q[1]; qo[2] = q[2]; qo[3] = q[3];
            }
        }
    } else {
        for (int e = t; e < xs; e += 256) xn[e] = xc[e];
        lambda = fmin(lambda * lm_up, lm_max);
    }
    if (t == 0) { ctl.lambda = lambda; ctl.try_valid = ok ? 1 : 0; }
    __syncthreads();
    __threadfence_block();
    TPROBE(5);
    // ---- d. skeleton tables of the new trial point ----------------------------------------------------
    compute_prep(dm, xn, fb.prep + ((size_t)f * 2 + ntry) * d.prep_size, *ps);
    TPROBE(6);
#ifdef AVT_TIMING
    __syncthreads();
    if (t == 0) { const double* pp = fb.prep + ((size_t)f * 2 + ntry) * d.prep_size; fb.trace[(size_t)f * 64 + 62] = pp[d.prep_size - 1]; fb.trace[(size_t)f * 64 + 63] = pp[d.prep_size - 2]; }
#endif
}

static size_t solve_lds_bytes(const AvtDims& d) {
    const int HS = d.HS, NB = HS / 4;
    return sizeof(double) * ((size_t)NB * NB * 18 + (size_t)NB * 18 + 18 + HS + 2) + sizeof(PrepScratch) + 64;
}

void launch_reduce(avt_ctx* c, int nframes) {
    const AvtDims& d = c->dm.d;
    hipLaunchKernelGGL(k_reduce, dim3(d.NPAIR, nframes), dim3(256), 0, c->cur_stream, c->dm, c->fb);
}

void launch_solve(avt_ctx* c, int nframes, int mode, const avt_options* o) {
    const AvtDims& d = c->dm.d;
    hipLaunchKernelGGL(k_solve, dim3(nframes), dim3(256), solve_lds_bytes(d), c->cur_stream, c->dm, c->fb, mode, o->lm_up, o->lm_down,
                       o->lm_lambda_min, o->lm_lambda_max);
}

int avt_solve_set_attributes() {
    return hipFuncSetAttribute((const void*)k_solve, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512) != hipSuccess;
}
