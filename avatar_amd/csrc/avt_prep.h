// avt_prep.h - the skeleton pass shared by k_solve (avt_lm.hip) and the initial-point workgroup that rides in k_lbs's grid
// (avt_kernels.hip).  Device code only.
#pragma once
#include "avt_device.h"

// -------------------------------------------------------------------------------------------------
// Skeleton pass: state x=(p,q,w) -> prep block (PrepareForEvaluation, AvatarOptimizer.cpp:283-325), LDS only.
//   B      : double scratch laid out by prep_layout(J,K) (avt_internal.h);
//   items  : host-built work items of the level-parallel pass, two 32-bit words each:
//            word0 = rp | v << 14 | stride_code << 28,  word1 = add | out << 14   (offsets into B)
//            B[out] = (B[rp] B[v] + B[rp+1] B[v+st] + B[rp+2] B[v+2st]) + B[add]
//            which is a world-rotation entry (R(-1,pa) rot_j), a world-origin entry (o_pa + R(-1,pa)(J_j - J_pa)) or a
//            shape-table entry (H_pa + R(-1,pa) Sp_j) depending on the offsets (:303-324); the root uses the identity.
// prep_stage_constants() runs at kernel start (its global loads hide behind the LM decision and the factorisation),
// prep_set_state() installs the state, prep_run() does the pass and writes the prep block.
// -------------------------------------------------------------------------------------------------
template <int NTH>
__device__ __forceinline__ void prep_stage_constants(const DeviceModel& dm, const PrepLayout& L, double* __restrict__ B,
                                                     int2* __restrict__ items, int* __restrict__ level) {
    const AvtDims& d = dm.d;
    const int J = d.J, K = d.K, t = threadIdx.x;
    for (int e = t; e < 3 * J * K; e += NTH) {
        B[L.Sp + e] = dm.Sp[e];
        B[L.S + e] = dm.S[e];
        B[L.jsr + e] = dm.jsr[e];
    }
    if (t < 3 * J) B[L.jsrb + t] = dm.jsr_base[t];
    if (t < 9) B[L.ident + t] = (t == 0 || t == 4 || t == 8) ? 1.0 : 0.0;
    if (t < 3) B[L.zero + t] = 0.0;
    const int2* gi = (const int2*)dm.fk_items;
    for (int e = t; e < L.nitems; e += NTH) items[e] = gi[e];
    if (t <= d.nlevels) level[t] = dm.fk_level_off[t];
    if (t < J) level[AVT_MAX_JOINTS + 2 + t] = dm.parent[t];     // parents follow the level offsets
}

// The same staging in two phases for the 256-thread kernels, whose prologue is a dependency chain: every load is REQUESTED first, into registers
// (prep_stage_request), and stored to LDS later (prep_stage_store), with whatever else the caller wants to request in between.  Written as loops of
// "LDS[e] = global[e]" each array is a round trip of its own - load, wait, store, next loop - because the scheduler does not move a load across the
// loop that waits for the previous one: six dependent round trips (9 k clocks) in front of the first instruction that needs any of the data
// (round 6, found in the ISA).  Covers 3 J K <= 6 x 256 and (work items from LDS, skeletons without the per-thread table) J (12 + 3 K) <= 8 x 256.
struct PrepStaged { double sp[6], s[6], jsr[6], jsrb; int lvl, par; int2 it[8]; };
__device__ __forceinline__ bool prep_stage_fits(const AvtDims& d, int nitems) { return 3 * d.J * d.K <= 6 * 256 && 3 * d.J <= 256 && nitems <= 8 * 256; }
template <bool ITEMS>
__device__ __forceinline__ PrepStaged prep_stage_request(const DeviceModel& dm, const PrepLayout& L) {
    const AvtDims& d = dm.d;
    const int J = d.J, K = d.K, t = threadIdx.x, n = 3 * J * K;
    PrepStaged st;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int e = t + 256 * i;
        const bool on = e < n;
        st.sp[i] = on ? dm.Sp[e] : 0.0; st.s[i] = on ? dm.S[e] : 0.0; st.jsr[i] = on ? dm.jsr[e] : 0.0;
    }
    st.jsrb = t < 3 * J ? dm.jsr_base[t] : 0.0;
    st.lvl = t <= d.nlevels ? dm.fk_level_off[t] : 0;
    st.par = t < J ? dm.parent[t] : 0;
    if constexpr (ITEMS) {
        const int2* gi = (const int2*)dm.fk_items;
#pragma unroll
        for (int i = 0; i < 8; ++i) { const int e = t + 256 * i; st.it[i] = e < L.nitems ? gi[e] : make_int2(0, 0); }
    }
    return st;
}
template <bool ITEMS>
__device__ __forceinline__ void prep_stage_store(const AvtDims& d, const PrepLayout& L, const PrepStaged& st, double* __restrict__ B, int2* __restrict__ items, int* __restrict__ level) {
    const int J = d.J, K = d.K, t = threadIdx.x, n = 3 * J * K;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int e = t + 256 * i;
        if (e < n) { B[L.Sp + e] = st.sp[i]; B[L.S + e] = st.s[i]; B[L.jsr + e] = st.jsr[i]; }
    }
    if (t < 3 * J) B[L.jsrb + t] = st.jsrb;
    if (t < 9) B[L.ident + t] = (t == 0 || t == 4 || t == 8) ? 1.0 : 0.0;
    if (t < 3) B[L.zero + t] = 0.0;
    if (t <= d.nlevels) level[t] = st.lvl;
    if (t < J) level[AVT_MAX_JOINTS + 2 + t] = st.par;
    if constexpr (ITEMS) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { const int e = t + 256 * i; if (e < L.nitems) items[e] = st.it[i]; }
    }
}

// local rotations, shape parameters and root position of the state into the scratch (q, w, p: LDS or registers' source)
__device__ __forceinline__ void prep_set_state(const AvtDims& d, const PrepLayout& L, double* __restrict__ B, const double* q,
                                               const double* w, const double* p) {
    const int t = threadIdx.x;
    if (t < d.J) quat_to_rot(q + 4 * t, B + L.rot + 9 * t);
    if (t < d.K) B[L.w + t] = w[t];
    if (t < 3) B[L.dv + t] = p[t];          // the root's "offset from its parent" is the global position
}

// CalcShape (AvatarOptimizer.cpp:249-281): jointPosInit = base + jointShapeReg*w, and each joint's offset from its parent (the
// parent's position is recomputed by the same lane: same operations, same bits, no barrier).  Lane tl in [0, 3J) takes coordinate tl;
// wk(k) = shape coefficient k of the state (a functor: the scratch entry, or - k_solve - the sum the retraction stores there, formed
// again by the same operation so that this pass need not wait for that store).
template <typename WK>
__device__ __forceinline__ void prep_joint_positions(const AvtDims& d, const PrepLayout& L, double* __restrict__ B, const int* __restrict__ level, int tl, WK wk) {
    const int J = d.J, K = d.K;
    auto run = [&](const int KK) {    // KK: constant for SMPL, so both dot products unroll and their LDS reads are in flight together
        if (tl >= 0 && tl < 3 * J) {
            const int j = tl / 3, c = tl - 3 * j;
            const int tp = j > 0 ? 3 * level[AVT_MAX_JOINTS + 2 + j] + c : tl;
            double a = 0.0, ap = 0.0;
            for (int k = 0; k < KK; ++k) {
                const double w = wk(k);
                a += B[L.jsr + tl * KK + k] * w;
                ap += B[L.jsr + tp * KK + k] * w;
            }
            const double mine = B[L.jsrb + tl] + a;
            B[L.jp + tl] = mine;
            if (j > 0) B[L.dv + tl] = mine - (B[L.jsrb + tp] + ap);
        }
    };
    if (K == 10) run(10); else run(K);
}

// The work items of prep_run's level loop are constants of the model (staged by prep_stage_constants): a thread's item of every level is read
// ONCE, all levels together and as early as the caller likes (k_solve: at kernel start, in front of the factorisation), instead of one dependent
// LDS round trip per level in front of the operand reads (round 6: level loop 4.8 k -> 3.8 k clocks; read inside prep_run the two round trips
// of this function were 1.3 k clocks of their own on the chain).
struct PrepItems { int2 v[AVT_PREP_LEVELS_REG]; bool reg; };
// ... for a 256-thread workgroup straight from the model's (level, thread) table in global memory: independent loads that can be the first thing a
// kernel requests (no staging, no barrier, no dependent address)
__device__ __forceinline__ PrepItems prep_preload_items_global(const DeviceModel& dm) {
    PrepItems pi;
    pi.reg = dm.d.fk_reg != 0;
    const int2* src = (const int2*)dm.fk_titems + threadIdx.x;
#pragma unroll
    for (int lv = 0; lv < AVT_PREP_LEVELS_REG; ++lv) pi.v[lv] = src[(size_t)lv * AVT_PREP_TITEM_THREADS];
    return pi;
}
template <int NTH>
__device__ __forceinline__ PrepItems prep_preload_items(const AvtDims& d, const int2* __restrict__ items, const int* __restrict__ level) {
    PrepItems pi;
    pi.reg = d.nlevels <= AVT_PREP_LEVELS_REG;
    const int t = threadIdx.x;
#pragma unroll
    for (int lv = 0; lv < AVT_PREP_LEVELS_REG; ++lv) {
        const int lo = level[min(lv, d.nlevels)], hi = level[min(lv + 1, d.nlevels)];
        pi.reg = pi.reg && hi - lo <= NTH;
        pi.v[lv] = (lo + t < hi) ? items[lo + t] : make_int2(-1, -1);
    }
    return pi;
}

// callers: a barrier separates prep_set_state() from prep_run().  JPDONE: the caller ran prep_joint_positions() in front of that barrier.
template <int NTH, bool JPDONE = false>
__device__ __forceinline__ void prep_run(const DeviceModel& dm, const PrepLayout& L, double* __restrict__ B, const int2* __restrict__ items,
                         const int* __restrict__ level, const double* __restrict__ q, double* __restrict__ prep, const PrepItems& pi) {
    const AvtDims d = dm.d;
    const int J = d.J, K = d.K, t = threadIdx.x;
    if constexpr (!JPDONE) {
        prep_joint_positions(d, L, B, level, t, [&](int k) { return B[L.w + k]; });
        __syncthreads();
    }
#ifdef AVT_TIMING
    if (threadIdx.x == 0) prep[d.prep_size - 1] = (double)clock64();
#endif
    // one tree level per barrier
    auto do_item = [&](const int2 it) {
        const int rp = it.x & 0x3fff, v = (it.x >> 14) & 0x3fff, scode = (unsigned)it.x >> 28;
        const int st = scode == 3 ? K : (scode == 2 ? 3 : scode);
        const int add = it.y & 0x3fff, out = (unsigned)it.y >> 14;
        B[out] = (B[rp] * B[v] + B[rp + 1] * B[v + st] + B[rp + 2] * B[v + 2 * st]) + B[add];
    };
    if (pi.reg) {
#pragma unroll
        for (int lv = 0; lv < AVT_PREP_LEVELS_REG; ++lv) {
            if (lv < d.nlevels) {      // (workgroup-uniform)
                if (pi.v[lv].x != -1) do_item(pi.v[lv]);
                __syncthreads();
            }
        }
    } else {
        for (int lv = 0; lv < d.nlevels; ++lv) {
            const int lo = level[lv], hi = level[lv + 1];
            for (int idx = lo + t; idx < hi; idx += NTH) do_item(items[idx]);
            __syncthreads();
        }
    }
#ifdef AVT_TIMING
    if (threadIdx.x == 0) prep[d.prep_size - 2] = (double)clock64();
#endif
    const double off0 = B[L.jp], off1 = B[L.jp + 1], off2 = B[L.jp + 2];
    for (int e = t; e < 9 * J; e += NTH) prep[prep_off_Rw(d) + e] = B[L.Rw + e];
    for (int e = t; e < 3 * J; e += NTH) {
        prep[prep_off_o(d) + e] = B[L.o + e];
        const int c = e % 3;
        prep[prep_off_Jh(d) + e] = B[L.jp + e] - (c == 0 ? off0 : (c == 1 ? off1 : off2));   // root at origin (:270-272)
    }
    // G[j] = H[j] - Rw[j]*S[j]  (shape block of :568-580); the index split divides by K: constant for SMPL
    auto g_table = [&](const int KK) {
        for (int e = t; e < 3 * J * KK; e += NTH) {
            const int j = e / (3 * KK), r = (e / KK) % 3, k = e % KK;
            const double* Rj = B + L.Rw + 9 * j;
            const double* S = B + L.S + j * 3 * KK;
            prep[prep_off_G(d) + e] = B[L.H + e] - (Rj[3 * r] * S[k] + Rj[3 * r + 1] * S[KK + k] + Rj[3 * r + 2] * S[2 * KK + k]);
        }
    };
    if (K == 10) g_table(10); else g_table(K);
    for (int e = t; e < 4 * J; e += NTH) prep[prep_off_q(d) + e] = q[e];
    if (t < K) prep[prep_off_w(d) + t] = B[L.w + t];
    if (t < 3) prep[prep_off_off(d) + t] = (t == 0 ? off0 : (t == 1 ? off1 : off2));
}


// LDS the initial-point workgroup needs behind its 16-byte aligned base: the current state, the skeleton scratch, the work items
inline size_t prep_init_lds_bytes(const AvtDims& d) {
    const PrepLayout L = prep_layout(d.J, d.K, d.xsize);
    return sizeof(double) * (((size_t)d.xsize + 1) & ~(size_t)1) + sizeof(double) * (size_t)L.ndoubles + sizeof(int) * (2 * (size_t)L.nitems + 2 * AVT_MAX_JOINTS + 4) + 64;
}

// Trial point := current point, with its skeleton tables: what the first evaluation of an ICP iteration runs on
// (the state part of k_solve<.., SOLVE_INIT>; it depends on the state alone, so for few frames it rides in the grid of the
// k_lbs launch in front of the ICP iteration instead of being a launch of its own behind k_records).  256 threads.
__device__ __forceinline__ void prep_init_block(const DeviceModel& dm, const FrameBuffers& fb, int f, char* smem, int cur) {
    const AvtDims& d = dm.d;
    const int J = d.J, K = d.K, xs = d.xsize, t = threadIdx.x;
    const PrepLayout L = prep_layout(J, K, xs);
    double* s_x = (double*)smem;
    double* B = s_x + ((xs + 1) & ~1);
    int2* s_items = (int2*)(B + L.ndoubles);
    int* s_level = (int*)(s_items + L.nitems);
    AvtFrameCtl& ctl = fb.ctl[f];
    const int tr = 1 - cur;
    double* x0 = fb.x + ((size_t)f * 2) * xs;
    // (two-phase staging, round 6: the state, the constants and this thread's work items of every tree level are requested together - one round trip
    // instead of eight - and stored behind it; skeletons the register forms do not cover stage the old way)
    const bool fast = d.fk_reg != 0 && prep_stage_fits(d, 0) && xs <= 256;
    PrepItems items;
    if (fast) {
        items = prep_preload_items_global(dm);
        const double xv = t < xs ? x0[(size_t)cur * xs + t] : 0.0;
        const PrepStaged st = prep_stage_request<false>(dm, L);
        if (t < xs) s_x[t] = xv;
        prep_stage_store<false>(d, L, st, B, s_items, s_level);
    } else {
        for (int e = t; e < xs; e += 256) s_x[e] = x0[(size_t)cur * xs + e];
        prep_stage_constants<256>(dm, L, B, s_items, s_level);
    }
    __syncthreads();
    if (t == 0) ctl.try_valid = 1;
    for (int e = t; e < xs; e += 256) x0[(size_t)tr * xs + e] = s_x[e];
    prep_set_state(d, L, B, s_x + 3, s_x + 3 + 4 * J, s_x);
    if (!fast) items = prep_preload_items<256>(d, s_items, s_level);
    __syncthreads();
    prep_run<256>(dm, L, B, s_items, s_level, s_x + 3, fb.prep + ((size_t)f * 2 + tr) * d.prep_size, items);
}
