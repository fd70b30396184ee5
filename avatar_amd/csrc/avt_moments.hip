// avt_moments.hip — the sufficient-statistics ("moment") form of the ICP data term (gfx950, wave64).
//
// AvatarCostFunctorCache::updateData (AvatarOptimizer.cpp:505-582) evaluates, per matched model point m and per Gauss-Newton
// iteration, the point and its Jacobian blocks.  Every one of those rows is LINEAR in the state-independent vector
//     psi_m = [ base_m | key_0m .. key_(K-1)m | 1 ]          (3 (K+1) + 1 numbers, index (K+1) i + s, s = 0: base)
// with coefficients that depend on the state alone:
//     x_mk  = R_k Phi_m omega + tau_k,   omega = [1; w],  tau_k = o_k - R_k J_k(omega)              (:507-514)
//     rotation column (j, c) = 2 [R_par(j) e_c] x  sum_{k under j} a_mk (x_mk - o_j)                (:529-566 in closed form)
//     shape column s         = sum_k a_mk (R_k Phi_m e_s + eta_ks),  eta_ks = H_k e_s - R_k S_k e_s  (:568-580)
//     residual               = sum_k a_mk x_mk - dbar_m                                             (:636-637)
// so J^T J, J^T r and the cost are exact contractions of
//     T_kk' = sum_m c_m a_mk a_mk' psi_m psi_m^T      one symmetric matrix per pair (k <= k') of joints assigned to a common vertex
//     D_k   = sum_m a_mk psi_m (sum_i (d_i - centre))^T
// which change only with the correspondences: k_moments accumulates them ONCE per ICP iteration (the only dense contraction left,
// on the fp64 matrix cores), and a Gauss-Newton iteration contracts them with the state (~0.45 M multiply-adds against ~20 M for
// rebuilding and contracting the Jacobian rows; tools/moment_proto2.py is the executable specification and checks it against
// the oracle's literal per-block formulas: H 4e-15, g 4e-13, cost 1e-12 relative).
//
//   k_moments    grid (np + cost-constant blocks, frames): workgroup p < np accumulates T_p (packed upper triangle) and, for a
//                diagonal pair, D_k; the others sum |d_i - centre|^2 over the matched data points;
//   k_pairpass   grid (ceil(np / 4) [+ GMM components], frames), 64 threads: one 16-lane group per joint pair contracts its T with the
//                trial state (phase A); up to 256 frames per launch trailing workgroups evaluate the pose prior there, one component
//                each - above that the prior is a launch of its own (k_prior);
//   k_assemble   grid (1, frames): turns the pair results into the dense system of the trial point in Hraw (the layout k_reduce
//                produces: full symmetric, row / column P = J^T r, [P][P] = sum c |r|^2).
#include <algorithm>
#include <type_traits>

#include "avt_device.h"
#include "avt_prior.h"

typedef double v4f64 __attribute__((ext_vector_type(4)));

#ifdef AVT_TIMING      // in-kernel phase probes of the assembly (tools/moment_phase_probe.py): shader clocks into the frame's debug trace
#define MPROBE(i) do { if (threadIdx.x == 0) fb.trace[(size_t)f * 64 + 40 + (i)] = (double)clock64(); } while (0)
#else
#define MPROBE(i) do {} while (0)
#endif

// packed upper triangle (row-major, diagonal included) of a symmetric n x n matrix: element (a <= b) at tri_off(a, n) + b - a
__host__ __device__ inline int tri_off(int a, int n) { return a * n - (a * (a - 1)) / 2; }
__host__ __device__ inline int tri_size(int n) { return n * (n + 1) / 2; }
__host__ __device__ inline int mom_tstride(int n) { return (tri_size(n) + 1) & ~1; }      // doubles between the packed T blocks of consecutive pairs (16-byte aligned blocks)

// =================================================================================================
// k_moments<NTP>.  One 256-thread workgroup per (frame, unordered pair).  The pair's static vertex list is compacted to the matched
// vertices (order kept), four of them per matrix instruction: lane (r = l & 15, v = l >> 4) holds psi_v[16 t + r] for the NTP
// 16-row tiles of psi - straight from the psi table, no LDS staging -; A = weight x psi, B = psi, one accumulator tile per upper
// tile pair, plus NTP tiles for D_k of a diagonal pair.  The accumulator tiles are DEALT to the four waves (tile g to wave g mod 4)
// and every wave walks all rounds: no cross-wave sum, no barrier behind the compaction, a third of the registers - the kernel's
// speed is how many workgroups a CU holds (16 accumulator registers per tile; all nine in every wave: 176 registers, two waves per
// SIMD).  MOM_UN rounds' fragments are requested together: a round is an L2 round trip.
// =================================================================================================
#ifdef AVT_TIMING
#define KPROBE(i) do { if (threadIdx.x == 0 && bx == 20) fb.trace[(size_t)f * 64 + 50 + (i)] = (double)clock64(); } while (0)
#else
#define KPROBE(i) do {} while (0)
#endif
#define MOM_SEG 512           // static list entries compacted per pass
#define MOM_EPT (MOM_SEG / 256)     // ... per thread
#define MOM_UN 2              // rounds whose fragments a wave requests together

// the rounds of one compacted list segment for wave WV: its tiles are g = WV, WV + 4, .. of [upper tile pairs | D tiles].
// The segment is padded to a multiple of 4 MOM_UN entries with weight-zero entries (no masks in the loop), an entry is the byte offset of its
// psi row, its weight and - for a diagonal pair - a_mk (sum_i d_i - c centre), all in LDS: the only global loads here are the psi fragments.
#ifndef MOM_PSI_LDS
#define MOM_PSI_LDS 0         // 1: the psi rows of a group of entries are fetched ONCE per workgroup (a quarter by every wave) and handed round through LDS; 0: every wave gathers its own fragments (the shipped form)
#endif
#define MOM_GROUP_ROUNDS (4 * MOM_UN)      // rounds per shared group: one step of MOM_UN rounds fetched by each of the four waves
#define MOM_PSI_ROW 48                     // doubles of a psi row in the shared buffer (16 NTP, NTP = 3: the SMPL shape)
// rounds [r_begin, r_end) of the current group with the psi fragments read from the shared buffer psi_s[entry in group][48]
template <int NTP, int WV, bool diag>
__device__ __forceinline__ void moments_consume(const double* __restrict__ psi_s, const double* __restrict__ s_w, const double* __restrict__ s_bd, int r_begin, int r_end, int ln,
                                                v4f64 (&acc)[(NTP * (NTP + 1) / 2 + NTP + 3) / 4]) {
    constexpr int NTPAIR = NTP * (NTP + 1) / 2, NSLOT = (NTPAIR + NTP + 3) / 4;
    constexpr int firstD = ((NTPAIR - WV + 3) / 4) * 4 + WV;
    constexpr bool hasD = firstD < NTPAIR + NTP;
    const int r16 = ln & 15, kk = ln >> 4;
    const double* bdrow = s_bd + (r16 < 3 ? r16 : 0) * (MOM_SEG + 4 * MOM_UN);
    const int rb = __builtin_amdgcn_readfirstlane(r_begin), re = __builtin_amdgcn_readfirstlane(r_end);
    for (int r0 = rb; r0 < re; r0 += MOM_UN) {
        double fr[MOM_UN][NTP], wgt[MOM_UN], bd[MOM_UN];
#pragma unroll
        for (int u = 0; u < MOM_UN; ++u) {
            const int idx = 4 * (r0 + u) + kk;
            const double* ps = psi_s + (size_t)(4 * (r0 - rb + u) + kk) * MOM_PSI_ROW + r16;
            wgt[u] = s_w[idx];
#pragma unroll
            for (int q = 0; q < NTP; ++q) fr[u][q] = ps[16 * q];
            bd[u] = 0.0;
            if (hasD && diag) { const double b = bdrow[idx]; bd[u] = r16 < 3 ? b : 0.0; }
        }
#pragma unroll
        for (int u = 0; u < MOM_UN; ++u) {
#pragma unroll
            for (int sl = 0; sl < NSLOT; ++sl) {
                const int g = WV + 4 * sl;
                if (g < NTPAIR) {
                    int ti = 0, pp = g;
#pragma unroll
                    for (int i = 0; i < NTP; ++i) if (pp >= NTP - ti && ti == i) { pp -= NTP - ti; ++ti; }
                    const int tj = ti + pp;
                    acc[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(fr[u][ti] * wgt[u], fr[u][tj], acc[sl], 0, 0, 0);
                } else if (g < NTPAIR + NTP) {
                    if (diag) acc[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(fr[u][g - NTPAIR], bd[u], acc[sl], 0, 0, 0);
                }
            }
        }
    }
}

template <int NTP, int WV, bool diag>
__device__ __forceinline__ void moments_rounds(const DeviceModel& dm, const unsigned* __restrict__ s_off, const double* __restrict__ s_w,
                                               const double* __restrict__ s_bd, int mseg, int ln,
                                               v4f64 (&acc)[(NTP * (NTP + 1) / 2 + NTP + 3) / 4]) {
    constexpr int NTPAIR = NTP * (NTP + 1) / 2, NSLOT = (NTPAIR + NTP + 3) / 4;
    constexpr int firstD = ((NTPAIR - WV + 3) / 4) * 4 + WV;                    // the wave's first tile index >= NTPAIR
    constexpr bool hasD = firstD < NTPAIR + NTP;
    const int r16 = ln & 15, kk = ln >> 4;
    const int nr = __builtin_amdgcn_readfirstlane((mseg + 4 * MOM_UN - 1) / (4 * MOM_UN) * MOM_UN);      // (a scalar: a loop the compiler takes for divergent keeps the accumulators in vector registers and copies them around every matrix instruction)
    const char* psi = (const char*)(dm.mom_psi + r16);
    const double* bdrow = s_bd + (r16 < 3 ? r16 : 0) * (MOM_SEG + 4 * MOM_UN);
    for (int r0 = 0; r0 < nr; r0 += MOM_UN) {
        double fr[MOM_UN][NTP], wgt[MOM_UN], bd[MOM_UN];
#pragma unroll
        for (int u = 0; u < MOM_UN; ++u) {
            const int idx = 4 * (r0 + u) + kk;
            const double* ps = (const double*)(psi + s_off[idx]);
            wgt[u] = s_w[idx];
#pragma unroll
            for (int q = 0; q < NTP; ++q) fr[u][q] = ps[16 * q];
            bd[u] = 0.0;
            if (hasD && diag) { const double b = bdrow[idx]; bd[u] = r16 < 3 ? b : 0.0; }      // B = a_mk (sum_i d_i - c centre), columns 0..2
        }
#pragma unroll
        for (int u = 0; u < MOM_UN; ++u) {
#pragma unroll
            for (int sl = 0; sl < NSLOT; ++sl) {
                const int g = WV + 4 * sl;          // compile-time after unrolling
                if (g < NTPAIR) {
                    int ti = 0, pp = g;
#pragma unroll
                    for (int i = 0; i < NTP; ++i) if (pp >= NTP - ti && ti == i) { pp -= NTP - ti; ++ti; }
                    const int tj = ti + pp;
                    acc[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(fr[u][ti] * wgt[u], fr[u][tj], acc[sl], 0, 0, 0);
                } else if (g < NTPAIR + NTP) {
                    if (diag) acc[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(fr[u][g - NTPAIR], bd[u], acc[sl], 0, 0, 0);
                }
            }
        }
    }
}

template <int NTP, int WV>
__device__ __forceinline__ void moments_store(const FrameBuffers& fb, const AvtDims& d, int f, int p, int k, bool diag, int ln,
                                              const v4f64 (&acc)[(NTP * (NTP + 1) / 2 + NTP + 3) / 4]) {
    constexpr int NTPAIR = NTP * (NTP + 1) / 2, NSLOT = (NTPAIR + NTP + 3) / 4;
    const int NPSI = d.mom_npsi, r16 = ln & 15, kk = ln >> 4;
    // accumulator element v of lane (c16 = l & 15, g4 = l >> 4): row 4 v + g4, column c16 of the tile; the upper triangle is kept
    double* T = fb.mom_T + ((size_t)f * d.mom_np + p) * mom_tstride(NPSI);
    double* D = fb.mom_D + ((size_t)f * d.J + k) * NPSI * 3;
#pragma unroll
    for (int sl = 0; sl < NSLOT; ++sl) {
        const int g = WV + 4 * sl;
        if (g < NTPAIR) {
            int ti = 0, pp = g;
#pragma unroll
            for (int i = 0; i < NTP; ++i) if (pp >= NTP - ti && ti == i) { pp -= NTP - ti; ++ti; }
            const int tj = ti + pp;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int a = 16 * ti + 4 * v + kk, b = 16 * tj + r16;
                if (a <= b && b < NPSI) T[tri_off(a, NPSI) + b - a] = acc[sl][v];
            }
        } else if (g < NTPAIR + NTP) {
            if (diag) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int a = 16 * (g - NTPAIR) + 4 * v + kk;
                    if (a < NPSI && r16 < 3) D[(size_t)a * 3 + r16] = acc[sl][v];
                }
            }
        }
    }
}

template <int NTP>
__global__ __launch_bounds__(256, 4) void k_moments(DeviceModel dm, FrameBuffers fb) {
    const AvtDims& d = dm.d;
    int bx, fy;
    xcd_frame_block(fb, bx, fy);      // a frame's workgroups share its counts and sums (cnt, fsum: 193 KB per SMPL frame, gathered by every pair)
    const int f = fy + fb.f0, t = threadIdx.x, V = d.V, NP = d.mom_np;
    __shared__ unsigned s_off[MOM_SEG + 4 * MOM_UN];
    __shared__ double s_w[MOM_SEG + 4 * MOM_UN], s_bd[3 * (MOM_SEG + 4 * MOM_UN)];
    __shared__ int s_wcnt[4];
    __shared__ double s_red[4];
    __shared__ __attribute__((aligned(16))) double s_psi[(MOM_PSI_LDS && NTP == 3) ? 4 * MOM_GROUP_ROUNDS * MOM_PSI_ROW : 2];      // [32 entries of a group][48]
    const int* cnt = fb.cnt + (size_t)f * V;
    const long long* fs = fb.fsum + (size_t)f * 3 * V;
    if (bx >= NP) {
        // sum over the matched data points of |d_i - centre|^2, 2048 points per workgroup in data order (fixed order): with
        //   sum_i |x_m - d_i|^2 = c_m |x_m|^2 - 2 x_m . sum_i d_i + sum_i |d_i|^2       (all relative to the frame centre)
        // it is the part of the data cost that does not depend on the state; k_solve FIRST adds the workgroups' parts up
        const int blk = bx - NP, N = fb.ctl[f].N;
        const size_t base = (size_t)f * fb.max_points;
        const double c0 = fb.ctl[f].centre[0], c1 = fb.ctl[f].centre[1], c2 = fb.ctl[f].centre[2];
        double a = 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = blk * 2048 + u * 256 + t;
            if (i < N && fb.corr[base + i] >= 0) {
                const double* dp = fb.data_raw + 3 * (base + i);
                const double ex = dp[0] - c0, ey = dp[1] - c1, ez = dp[2] - c2;
                a += ex * ex + ey * ey + ez * ez;
            }
        }
        a = wave_sum(a);
        if ((t & 63) == 0) s_red[t >> 6] = a;
        __syncthreads();
        if (t == 0) fb.const_part[(size_t)f * fb.const_blocks + blk] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
        return;
    }
    KPROBE(0);
    const int p = bx, k = dm.mom_pair[2 * p], k2 = dm.mom_pair[2 * p + 1];
    const bool diag = k == k2;
    const int lo = dm.mom_lstart[p], n = dm.mom_lstart[p + 1] - lo;
    constexpr int NSLOT = (NTP * (NTP + 1) / 2 + NTP + 3) / 4;
    v4f64 acc[NSLOT];
    const v4f64 z4 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) acc[i] = z4;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6), ln = t & 63;      // (a scalar: the switch over the wave's share below must not look divergent)
    for (int base = 0; base < n; base += MOM_SEG) {
        // ---- compaction of entries base + MOM_EPT t .. base + MOM_EPT (t + 1) - 1 (order kept)
        int vv[MOM_EPT], cc[MOM_EPT], mine = 0;
        double wa[MOM_EPT], wb[MOM_EPT];
#pragma unroll
        for (int u = 0; u < MOM_EPT; ++u) {
            const int e = min(base + MOM_EPT * t + u, n - 1);
            vv[u] = dm.mom_lv[lo + e];
            wa[u] = dm.mom_lw[2 * (size_t)(lo + e)]; wb[u] = dm.mom_lw[2 * (size_t)(lo + e) + 1];
        }
#pragma unroll
        for (int u = 0; u < MOM_EPT; ++u) {
            const int e = base + MOM_EPT * t + u;
            cc[u] = e < n ? cnt[vv[u]] : 0;
            mine += cc[u] > 0;
        }
        KPROBE(1);
        const int incl = wave_incl_scan(mine);
        if (ln == 63) s_wcnt[wv] = incl;
        __syncthreads();
        int pos = incl - mine, mseg = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { if (w < wv) pos += s_wcnt[w]; mseg += s_wcnt[w]; }
        long long fv[MOM_EPT][3];
        if (diag) {
#pragma unroll
            for (int u = 0; u < MOM_EPT; ++u)
#pragma unroll
                for (int i = 0; i < 3; ++i) fv[u][i] = cc[u] > 0 ? fs[(size_t)i * V + vv[u]] : 0;
        }
#pragma unroll
        for (int u = 0; u < MOM_EPT; ++u)
            if (cc[u] > 0) {
                s_off[pos] = (unsigned)vv[u] * (unsigned)(16 * NTP * sizeof(double)); s_w[pos] = (double)cc[u] * wa[u] * wb[u];
                if (diag) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) s_bd[i * (MOM_SEG + 4 * MOM_UN) + pos] = wa[u] * ((double)fv[u][i] / AVT_FIX_SCALE);
                }
                ++pos;
            }
        if (t < 4 * MOM_UN && mseg + t < ((mseg + 4 * MOM_UN - 1) / (4 * MOM_UN)) * (4 * MOM_UN)) {      // weight-zero entries up to a whole number of trips
            s_off[mseg + t] = 0; s_w[mseg + t] = 0.0;
            s_bd[mseg + t] = 0.0; s_bd[MOM_SEG + 4 * MOM_UN + mseg + t] = 0.0; s_bd[2 * (MOM_SEG + 4 * MOM_UN) + mseg + t] = 0.0;
        }
        __syncthreads();
        KPROBE(2);
        if constexpr (MOM_PSI_LDS && NTP == 3) {
            // psi shared through LDS: a group = 4 MOM_UN rounds = 32 entries; wave w fetches the rows of entries 8 w .. 8 w + 7 of the group (lane: entry
            // l >> 3, 48-byte chunk l & 7: three 16-byte loads), the NEXT group's rows while this one is contracted; two barriers per group
            typedef double d2 __attribute__((ext_vector_type(2)));
            const int nr = (mseg + 4 * MOM_UN - 1) / (4 * MOM_UN) * MOM_UN;      // rounds of this segment (a whole number of MOM_UN trips)
            d2 pf[3];
            auto fetch = [&](int g0) {      // rounds g0 .. g0 + MOM_GROUP_ROUNDS - 1: my step's 8 entries
                const int idx = min(4 * g0 + 8 * wv + (ln >> 3), 4 * nr - 1);
                const d2* row = (const d2*)((const char*)dm.mom_psi + s_off[idx]) + 3 * (ln & 7);
                pf[0] = row[0]; pf[1] = row[1]; pf[2] = row[2];
            };
            fetch(0);
            for (int g0 = 0; g0 < nr; g0 += MOM_GROUP_ROUNDS) {
                if (g0 > 0) __syncthreads();      // the previous group has been contracted by every wave
                { d2* dst = (d2*)(s_psi + (size_t)(8 * wv + (ln >> 3)) * MOM_PSI_ROW) + 3 * (ln & 7); dst[0] = pf[0]; dst[1] = pf[1]; dst[2] = pf[2]; }
                __syncthreads();
                if (g0 + MOM_GROUP_ROUNDS < nr) fetch(g0 + MOM_GROUP_ROUNDS);
                const int r_end = min(nr, g0 + MOM_GROUP_ROUNDS);
                switch (wv) {
                    case 0: { if (diag) moments_consume<NTP, 0, true>(s_psi, s_w, s_bd, g0, r_end, ln, acc); else moments_consume<NTP, 0, false>(s_psi, s_w, s_bd, g0, r_end, ln, acc); } break;
                    case 1: { if (diag) moments_consume<NTP, 1, true>(s_psi, s_w, s_bd, g0, r_end, ln, acc); else moments_consume<NTP, 1, false>(s_psi, s_w, s_bd, g0, r_end, ln, acc); } break;
                    case 2: { if (diag) moments_consume<NTP, 2, true>(s_psi, s_w, s_bd, g0, r_end, ln, acc); else moments_consume<NTP, 2, false>(s_psi, s_w, s_bd, g0, r_end, ln, acc); } break;
                    default: { if (diag) moments_consume<NTP, 3, true>(s_psi, s_w, s_bd, g0, r_end, ln, acc); else moments_consume<NTP, 3, false>(s_psi, s_w, s_bd, g0, r_end, ln, acc); } break;
                }
            }
        } else
        switch (wv) {
            case 0: { if (diag) moments_rounds<NTP, 0, true>(dm, s_off, s_w, s_bd, mseg, ln, acc); else moments_rounds<NTP, 0, false>(dm, s_off, s_w, s_bd, mseg, ln, acc); } break;
            case 1: { if (diag) moments_rounds<NTP, 1, true>(dm, s_off, s_w, s_bd, mseg, ln, acc); else moments_rounds<NTP, 1, false>(dm, s_off, s_w, s_bd, mseg, ln, acc); } break;
            case 2: { if (diag) moments_rounds<NTP, 2, true>(dm, s_off, s_w, s_bd, mseg, ln, acc); else moments_rounds<NTP, 2, false>(dm, s_off, s_w, s_bd, mseg, ln, acc); } break;
            default: { if (diag) moments_rounds<NTP, 3, true>(dm, s_off, s_w, s_bd, mseg, ln, acc); else moments_rounds<NTP, 3, false>(dm, s_off, s_w, s_bd, mseg, ln, acc); } break;
        }
        KPROBE(3);
        if (base + MOM_SEG < n) __syncthreads();      // the list is rewritten by the next pass
    }
    KPROBE(4);
    switch (wv) {
        case 0: moments_store<NTP, 0>(fb, d, f, p, k, diag, ln, acc); break;
        case 1: moments_store<NTP, 1>(fb, d, f, p, k, diag, ln, acc); break;
        case 2: moments_store<NTP, 2>(fb, d, f, p, k, diag, ln, acc); break;
        default: moments_store<NTP, 3>(fb, d, f, p, k, diag, ln, acc); break;
    }
    KPROBE(6);
}

void launch_moments(avt_ctx* c, int nframes) {
    const AvtDims& d = c->dm.d;
    const dim3 grid(d.mom_np + c->fb.const_used, nframes);
    switch (d.mom_ntp) {
        case 1: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_moments<1>), grid, dim3(256), 0, c->cur_stream, c->dm, c->fb); break;
        case 2: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_moments<2>), grid, dim3(256), 0, c->cur_stream, c->dm, c->fb); break;
        case 3: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_moments<3>), grid, dim3(256), 0, c->cur_stream, c->dm, c->fb); break;
        default: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_moments<4>), grid, dim3(256), 0, c->cur_stream, c->dm, c->fb); break;
    }
}

// =================================================================================================
// Skeleton tables of the assembly (what every coefficient above is made of), in LDS, from a prep block (avt_internal.h).
// =================================================================================================
struct MomSkel {
    const double* Rw;     // [J][9] world rotations, row-major
    const double* oc;     // [J][3] world joint origins minus the frame centre
    const double* tau;    // [J][3] o_k - R_k J_k(omega) - centre
    const double* eta;    // [J][3][K] H_k - R_k S_k
    const double* om;     // [K + 1] omega = [1; w]
    const int* parent;    // [J]
};
__host__ __device__ inline int mom_skel_doubles(const AvtDims& d) { return ((15 * d.J + 3 * d.J * d.K + d.K + 1) + 1) & ~1; }

template <int NTH>
__device__ __forceinline__ void mom_skel_from_prep(const AvtDims& d, const double* __restrict__ prep, const double* centre, double* __restrict__ sk_mem,
                                                   int* __restrict__ s_parent, const int* __restrict__ parent_g, MomSkel& sk) {
    const int J = d.J, K = d.K, t = threadIdx.x;
    double* Rw = sk_mem;                 // [9 J]
    double* oc = Rw + 9 * J;             // [3 J]
    double* tau = oc + 3 * J;            // [3 J]
    double* eta = tau + 3 * J;           // [3 J K]
    double* om = eta + 3 * J * K;        // [K + 1]
    for (int e = t; e < 9 * J; e += NTH) Rw[e] = prep[prep_off_Rw(d) + e];
    for (int e = t; e < 3 * J * K; e += NTH) eta[e] = prep[prep_off_G(d) + e];
    for (int e = t; e < 3 * J; e += NTH) {
        const int j = e / 3, r = e - 3 * j;
        const double* Rj = prep + prep_off_Rw(d) + 9 * j;
        const double* Jh = prep + prep_off_Jh(d) + 3 * j;
        const double* off = prep + prep_off_off(d);
        const double o = prep[prep_off_o(d) + e] - centre[r];
        oc[e] = o;
        tau[e] = o - (Rj[3 * r] * (Jh[0] + off[0]) + Rj[3 * r + 1] * (Jh[1] + off[1]) + Rj[3 * r + 2] * (Jh[2] + off[2]));
    }
    if (t <= K) om[t] = t == 0 ? 1.0 : prep[prep_off_w(d) + t - 1];
    if (t < J) s_parent[t] = parent_g[t];
    sk.Rw = Rw; sk.oc = oc; sk.tau = tau; sk.eta = eta; sk.om = om; sk.parent = s_parent;
}

// per-frame scratch between k_pairpass and k_assemble (FrameBuffers::mom_rec), in doubles:
//   X16 [2 np + 1][16]   per ordered pair (op = 2 p: k -> k', 2 p + 1: k' -> k): W (9, row-major), Va, Vb, t0; the last row stays zero
//   REC [2 np][K][6]     per (ordered pair, shape key): axial(Y), U
//   Z   [nwg][K K + K]   per pair-pass workgroup: its pairs' shape-shape columns and sum tr(Y)
#ifndef MOM_PP_PAIRS
#define MOM_PP_PAIRS 4        // pairs (16-lane groups) per pair-pass workgroup
#endif
__host__ __device__ inline int mom_nwg(const AvtDims& d) { return (d.mom_np + MOM_PP_PAIRS - 1) / MOM_PP_PAIRS; }      // pair-pass workgroups per frame
__host__ __device__ inline size_t mom_off_rec(const AvtDims& d) { return (size_t)(2 * d.mom_np + 1) * 16; }
__host__ __device__ inline size_t mom_off_z(const AvtDims& d) { return mom_off_rec(d) + (size_t)2 * d.mom_np * d.K * 6; }
__host__ __device__ inline size_t mom_frame_scratch(const AvtDims& d) { return (mom_off_z(d) + (size_t)mom_nwg(d) * (d.K * d.K + d.K) + 7) & ~(size_t)7; }

// =================================================================================================
// k_pairpass<KC>.  grid (ceil(np / MOM_PP_PAIRS) [+ GMM components], frames), block 16 MOM_PP_PAIRS = 64: one 16-lane group per unordered
// pair (k <= k'), lane = s' (0 .. K).  tools/moment_proto2.py::assemble "phase A":
//   the pair's packed T and the two joints' tables are staged in the group's LDS slice (coalesced 16-byte loads, wave-local hand-over);
//   Q[i][i'] = sum_s om_s T[(i,s),(i',s')],  zz[s] = sum_ii' (R_k^T R_k')[i][i'] T[(i,s),(i',s')]   - 9 (K + 1) + 4 reads of the packed T,
//   P2, p1 by DPP row sums over the group; lanes 1 .. K write the per-(ordered pair, shape key) records, lane 0 - through the same
//   instructions - X16 of both orders; the lanes' shape-shape columns and sum tr(Y) are added over the workgroup's pairs in group order
//   through LDS.  Trailing workgroups (launch_assemble decides): the GMM pose prior of the trial point, one component each.
// =================================================================================================
// sum over the 16 lanes of a DPP row, every lane ending with the same bits (the two operands of each add are the same pair of numbers in
// both lanes): quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror - register moves, no LDS permutes
template <int CTRL>
__device__ __forceinline__ double mom_dpp(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double mom_row_sum(double v) {
    v += mom_dpp<0xB1>(v);
    v += mom_dpp<0x4E>(v);
    v += mom_dpp<0x141>(v);
    v += mom_dpp<0x140>(v);
    return v;
}

#ifdef AVT_TIMING
#define PPROBE(i) do { if (threadIdx.x == 0 && bx == 5) fb.trace[(size_t)f * 64 + 24 + (i)] = (double)clock64(); } while (0)
#else
#define PPROBE(i) do {} while (0)
#endif
#define AVT_SMPL_PROMISES(d) do { __builtin_assume((d).J == 24); __builtin_assume((d).K == 10); __builtin_assume((d).P == 85); __builtin_assume((d).HS == 88); \
                                   __builtin_assume((d).xsize == 109); __builtin_assume((d).ndims == 69); __builtin_assume((d).prep_size == 1192); } while (0)
template <int KC, bool SM = false>      // SM: the model has SMPL's dimensions (AVT_SMPL_PROMISES)
__global__ __launch_bounds__(16 * MOM_PP_PAIRS) void k_pairpass(DeviceModel dm, FrameBuffers fb) {
    // SM (the host launches this copy for models with SMPL's dimensions, launch_assemble): promises about the dimensions where the structure lies -
    // a private copy would live in scratch memory, its arrays are indexed at run time.  One wave per workgroup, six workgroups per CU: the kernel is
    // short of issue slots and its index arithmetic folds; the prior's workgroups in this grid are the launch's longest and gain most (their loops over
    // the 69 prior dimensions).  k_prior carries the same promises: the two homes of the prior must give the same bits.
    if constexpr (SM) AVT_SMPL_PROMISES(dm.d);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NTH = 16 * MOM_PP_PAIRS;
    const AvtDims& d = dm.d;
    int bx, fy;
    xcd_frame_block(fb, bx, fy);      // the frame's k_assemble_parts and k_solve workgroups run on the same XCD (what this launch writes is what they read)
    const int f = fy + fb.f0, t = threadIdx.x;
    const int J = d.J, K = KC ? KC : d.K, S1 = K + 1, NP = d.mom_np, NPSI = KC ? 3 * (KC + 1) + 1 : d.mom_npsi;
    const int TS = mom_tstride(NPSI), JS = 15 + 3 * K, GS = (2 * JS + S1 + 1) & ~1;      // doubles of a pair's T block / of one joint's tables / of a group's slice
    const int2 slot_state = frame_slot_state(fb, f);
    if (slot_state.y == AVT_TRY_DONE) return;      // the frame met the stopping rule in this ICP iteration: no trial point (the prior workgroups in this grid too)
    const int try_slot = 1 - slot_state.x;
    double* scr = fb.mom_rec + (size_t)f * mom_frame_scratch(d);
    double* X16 = scr;
    double* REC = scr + mom_off_rec(d);
    if (bx >= mom_nwg(d)) {      // trailing workgroups: the GMM pose prior of the trial point, one component each (avt_prior.h)
        prior_component<NTH>(dm, fb, f, bx - mom_nwg(d), try_slot, (double*)smem);
        return;
    }
    const int gid = t >> 4, sl = t & 15;
    PPROBE(0);
    const int p = bx * MOM_PP_PAIRS + gid;
    const bool pair_on = p < NP, lane_on = sl < S1;
    const int pc = pair_on ? p : 0;
    const int k = dm.mom_pair[2 * pc], k2 = dm.mom_pair[2 * pc + 1];
    // ---- staging: the pair's packed T (TS doubles, one contiguous block: the whole workgroup streams MOM_PP_PAIRS consecutive blocks) and
    // the two joints' tables go through the group's own LDS; a group lives inside one wave, so the hand-over is wave-local
    double* tl = (double*)smem + (size_t)gid * TS;                                   // [TS]
    double* gsl = (double*)smem + (size_t)MOM_PP_PAIRS * TS + (size_t)gid * GS;      // [GS]: joint k | joint k' (Rw 9, o 3, Jh 3, eta 3 K) | omega
    double* ZR = (double*)smem + (size_t)MOM_PP_PAIRS * (TS + GS);                  // [MOM_PP_PAIRS][K K + K]
    {
        typedef double d2 __attribute__((ext_vector_type(2)));
        const d2* Tg = (const d2*)(fb.mom_T + ((size_t)f * NP + pc) * TS);
        const int nv = TS >> 1;
        constexpr int NLD = KC == 10 ? 19 : 0;                                      // 596 / 2 / 16 rounded up (SMPL); other K: the loop below
        const double* prep = fb.prep + ((size_t)f * 2 + try_slot) * d.prep_size;
        auto jsrc = [&](int j, int e) -> const double* {                              // element e of joint j's tables in the prep block
            return e < 9 ? prep + prep_off_Rw(d) + 9 * j + e : (e < 12 ? prep + prep_off_o(d) + 3 * j + (e - 9) : (e < 15 ? prep + prep_off_Jh(d) + 3 * j + (e - 12) : prep + prep_off_G(d) + (size_t)3 * K * j + (e - 15)));
        };
        if (NLD) {
            d2 v[NLD ? NLD : 1];
#pragma unroll
            for (int u = 0; u < NLD; ++u) { const int i = 16 * u + sl; v[u] = i < nv ? Tg[i] : (d2){0.0, 0.0}; }
            for (int e = sl; e < 2 * JS; e += 16) gsl[e] = *jsrc(e < JS ? k : k2, e < JS ? e : e - JS);
            if (lane_on) gsl[2 * JS + sl] = sl == 0 ? 1.0 : prep[prep_off_w(d) + sl - 1];
#pragma unroll
            for (int u = 0; u < NLD; ++u) { const int i = 16 * u + sl; if (i < nv) ((d2*)tl)[i] = v[u]; }
        } else {
            for (int e = sl; e < 2 * JS; e += 16) gsl[e] = *jsrc(e < JS ? k : k2, e < JS ? e : e - JS);
            if (lane_on) gsl[2 * JS + sl] = sl == 0 ? 1.0 : prep[prep_off_w(d) + sl - 1];
            for (int i = sl; i < nv; i += 16) ((d2*)tl)[i] = Tg[i];
        }
    }
    const double* offp = fb.prep + ((size_t)f * 2 + try_slot) * d.prep_size + prep_off_off(d);
    const double off0 = offp[0], off1 = offp[1], off2 = offp[2];
    const double cen0 = fb.ctl[f].centre[0], cen1 = fb.ctl[f].centre[1], cen2 = fb.ctl[f].centre[2];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    PPROBE(1);
    const double* RwA = gsl, *RwB = gsl + JS;
    const double* etaA = gsl + 15, *etaB = gsl + JS + 15;      // eta_k[r][s] at [r K + s]
    const double* omL = gsl + 2 * JS;
    // tau = o - centre - R (Jh + off): the joint's constant in x_mk = R_k Phi_m omega + tau_k, relative to the frame centre
    double tauA[3], tauB[3];
    {
        const double ja0 = gsl[12] + off0, ja1 = gsl[13] + off1, ja2 = gsl[14] + off2, jb0 = gsl[JS + 12] + off0, jb1 = gsl[JS + 13] + off1, jb2 = gsl[JS + 14] + off2;
        const double ca[3] = {cen0, cen1, cen2};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            tauA[r] = (gsl[9 + r] - ca[r]) - (RwA[3 * r] * ja0 + RwA[3 * r + 1] * ja1 + RwA[3 * r + 2] * ja2);
            tauB[r] = (gsl[JS + 9 + r] - ca[r]) - (RwB[3 * r] * jb0 + RwB[3 * r + 1] * jb1 + RwB[3 * r + 2] * jb2);
        }
    }
    const int col = lane_on ? sl : 0;
    // my three columns c = S1 i' + col of the packed triangle: element (r, c) at (r <= c ? rowoff(r) + c : coloff[i'] + r)
    int coloff[3], cidx[3];
#pragma unroll
    for (int i2 = 0; i2 < 3; ++i2) { cidx[i2] = S1 * i2 + col; coloff[i2] = tri_off(cidx[i2], NPSI) - cidx[i2]; }
    auto tload = [&](int r, int i2) { return tl[r <= cidx[i2] ? tri_off(r, NPSI) - r + cidx[i2] : coloff[i2] + r]; };
    double v0[9], tph[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int i2 = 0; i2 < 3; ++i2) v0[3 * i + i2] = tload(S1 * i, i2);
#pragma unroll
    for (int i2 = 0; i2 < 3; ++i2) tph[i2] = tl[coloff[i2] + NPSI - 1];
    const double t0 = tl[tri_size(NPSI) - 1];
    double zc[KC ? KC : AVT_MAX_SHAPE], yx = 0.0;
#pragma unroll
    for (int s = 0; s < (KC ? KC : AVT_MAX_SHAPE); ++s) zc[s] = 0.0;
    const double om_l = lane_on ? omL[sl] : 0.0;
    const double nu = k == k2 ? 0.5 : 1.0;
    double Q[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) Q[e] = v0[e];        // s = 0 (base): omega_0 = 1
    {
        double G[9];
        {   // R_k^T R_k' (the rotations themselves are fetched again behind the loop: they would only occupy registers in it)
            double A[9], B[9];
#pragma unroll
            for (int e = 0; e < 9; ++e) { A[e] = RwA[e]; B[e] = RwB[e]; }
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int i2 = 0; i2 < 3; ++i2) G[3 * i + i2] = A[i] * B[i2] + A[3 + i] * B[3 + i2] + A[6 + i] * B[6 + i2];
        }
        auto use = [&](int s, const double (&v)[9], double& zout) {
            const double oms = omL[s];
            double zz = 0.0;
#pragma unroll
            for (int e = 0; e < 9; ++e) { Q[e] = fma(oms, v[e], Q[e]); zz = fma(G[e], v[e], zz); }
            zout = nu * zz;
        };
        if (KC) {
            // the kernel is L2 latency: the 9 K loads go out in three batches, each requested as a whole before its first use
            constexpr int KK = KC ? KC : 3, B1 = (KK + 2) / 3, B2 = (KK - B1 + 1) / 2, B3 = KK - B1 - B2;
            auto batch = [&](auto nb, int first) {
                constexpr int NB = decltype(nb)::value;
                double vv[NB > 0 ? NB : 1][9];
#pragma unroll
                for (int u = 0; u < NB; ++u)
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int i2 = 0; i2 < 3; ++i2) vv[u][3 * i + i2] = tload(S1 * i + first + u, i2);
#pragma unroll
                for (int u = 0; u < NB; ++u) {
#pragma unroll
                    for (int q = 0; q < KK; ++q) if (q == first + u - 1) use(first + u, vv[u], zc[q]);
                }
            };
            batch(std::integral_constant<int, B1>{}, 1);
            batch(std::integral_constant<int, B2>{}, 1 + B1);
            batch(std::integral_constant<int, B3>{}, 1 + B1 + B2);
        } else {
            for (int s = 1; s < S1; ++s) {
                double v[9], z;
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int i2 = 0; i2 < 3; ++i2) v[3 * i + i2] = tload(S1 * i + s, i2);
                use(s, v, z);
#pragma unroll
                for (int q = 0; q < AVT_MAX_SHAPE; ++q) if (q == s - 1) zc[q] = z;
            }
        }
    }
    PPROBE(2);
    double Ra[9], Rb[9], ta[3], tb[3];
#pragma unroll
    for (int e = 0; e < 9; ++e) { Ra[e] = RwA[e]; Rb[e] = RwB[e]; }
#pragma unroll
    for (int e = 0; e < 3; ++e) { ta[e] = tauA[e]; tb[e] = tauB[e]; }
    // group sums over the lanes (fixed butterfly: every lane of the group ends with the same bits)
    double P2[9], p1[3];
#pragma unroll
    for (int e = 0; e < 9; ++e) P2[e] = om_l * Q[e];
#pragma unroll
    for (int e = 0; e < 3; ++e) p1[e] = om_l * tph[e];
#pragma unroll
    for (int e = 0; e < 9; ++e) P2[e] = mom_row_sum(P2[e]);
#pragma unroll
    for (int e = 0; e < 3; ++e) p1[e] = mom_row_sum(p1[e]);
    PPROBE(3);
    double Rap1[3], Rbp1[3], Va[3], Vb[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        Rap1[r] = Ra[3 * r] * p1[0] + Ra[3 * r + 1] * p1[1] + Ra[3 * r + 2] * p1[2];
        Rbp1[r] = Rb[3 * r] * p1[0] + Rb[3 * r + 1] * p1[1] + Rb[3 * r + 2] * p1[2];
        Va[r] = fma(t0, ta[r], Rap1[r]);
        Vb[r] = fma(t0, tb[r], Rbp1[r]);
    }
    if (pair_on && lane_on) {
        // Lanes 1 .. K: the records of shape key s = lane - 1,  Y = R_a Qs R_b^T + V_a eta_b,s^T + tau_a ylin^T,  U = ylin + t0 eta_b,s.
        // Lane 0 runs the SAME instructions on other operands and gets W_kk' = R_k P2 R_k'^T + Va tau_k'^T + tau_k (R_k' p1)^T  (W_k'k = W_kk'^T)
        // - a branch of its own would be executed in series with the records.
        const bool lane0 = sl == 0;
        const int s = lane0 ? 0 : sl - 1;
        const double* ea = etaA;                             // eta_k[r][s]
        const double* eb = etaB;
        double ya[3], yb[3];                                 // R_k tphi, R_k' tphi
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            ya[r] = Ra[3 * r] * tph[0] + Ra[3 * r + 1] * tph[1] + Ra[3 * r + 2] * tph[2];
            yb[r] = Rb[3 * r] * tph[0] + Rb[3 * r + 1] * tph[1] + Rb[3 * r + 2] * tph[2];
        }
        const double eas[3] = {ea[s], ea[K + s], ea[2 * K + s]}, ebs[3] = {eb[s], eb[K + s], eb[2 * K + s]};
        double Qs[9], yl0[3], eB0[3];
#pragma unroll
        for (int e = 0; e < 9; ++e) Qs[e] = lane0 ? P2[e] : Q[e];
#pragma unroll
        for (int e = 0; e < 3; ++e) { yl0[e] = lane0 ? Rbp1[e] : yb[e]; eB0[e] = lane0 ? tb[e] : ebs[e]; }
        auto sandwich = [&](const double (&RA)[9], const double (&RB)[9], const double (&QQ)[9], const double (&VA)[3], const double (&TA)[3],
                            const double (&ylin)[3], const double (&eB)[3], double (&Y)[9]) {
            double RQ[9];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) RQ[3 * r + c] = RA[3 * r] * QQ[c] + RA[3 * r + 1] * QQ[3 + c] + RA[3 * r + 2] * QQ[6 + c];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    Y[3 * r + c] = (RQ[3 * r] * RB[3 * c] + RQ[3 * r + 1] * RB[3 * c + 1] + RQ[3 * r + 2] * RB[3 * c + 2]) + VA[r] * eB[c] + TA[r] * ylin[c];
        };
        double Y[9];
        sandwich(Ra, Rb, Qs, Va, ta, yl0, eB0, Y);
        if (lane0) {
            double* x0 = X16 + (size_t)(2 * p) * 16;
#pragma unroll
            for (int e = 0; e < 9; ++e) x0[e] = Y[e];
#pragma unroll
            for (int e = 0; e < 3; ++e) { x0[9 + e] = Va[e]; x0[12 + e] = Vb[e]; }
            x0[15] = t0;
            double* x1 = x0 + 16;
            const bool two = k != k2;
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) x1[3 * r + c] = two ? Y[3 * c + r] : 0.0;
#pragma unroll
            for (int e = 0; e < 3; ++e) { x1[9 + e] = two ? Vb[e] : 0.0; x1[12 + e] = two ? Va[e] : 0.0; }
            x1[15] = two ? t0 : 0.0;
            if (p == 0) {      // the all-zero row the padded lists of k_assemble point at
                double* xz = X16 + (size_t)(2 * NP) * 16;
#pragma unroll
                for (int e = 0; e < 16; ++e) xz[e] = 0.0;
            }
        } else {
            double* rc = REC + ((size_t)(2 * p) * K + s) * 6;
            rc[0] = Y[5] - Y[7]; rc[1] = Y[6] - Y[2]; rc[2] = Y[1] - Y[3];
            rc[3] = fma(t0, ebs[0], yb[0]); rc[4] = fma(t0, ebs[1], yb[1]); rc[5] = fma(t0, ebs[2], yb[2]);
            yx += (Y[0] + Y[4]) + Y[8];
        }
        if (!lane0) {
            double* rc = REC + ((size_t)(2 * p + 1) * K + s) * 6;
            if (k != k2) {
                sandwich(Rb, Ra, Q, Vb, tb, ya, eas, Y);
                rc[0] = Y[5] - Y[7]; rc[1] = Y[6] - Y[2]; rc[2] = Y[1] - Y[3];
                rc[3] = fma(t0, eas[0], ya[0]); rc[4] = fma(t0, eas[1], ya[1]); rc[5] = fma(t0, eas[2], ya[2]);
                yx += (Y[0] + Y[4]) + Y[8];
            } else {
#pragma unroll
                for (int e = 0; e < 6; ++e) rc[e] = 0.0;
            }
            // column t = s of Z~': nu (zz + eta_k,s2 . yb + eta_k',s2 . ya + t0 eta_k,s2 . eta_k',t)   (nu zz is already in zc)
#pragma unroll
            for (int s2 = 0; s2 < (KC ? KC : AVT_MAX_SHAPE); ++s2) {
                if (s2 < K) {
                    const double e0 = ea[s2], e1 = ea[K + s2], e2 = ea[2 * K + s2];
                    const double f0 = eb[s2], f1 = eb[K + s2], f2 = eb[2 * K + s2];
                    const double add = (e0 * yb[0] + e1 * yb[1] + e2 * yb[2]) + (f0 * ya[0] + f1 * ya[1] + f2 * ya[2]) + t0 * (e0 * ebs[0] + e1 * ebs[1] + e2 * ebs[2]);
                    zc[s2] = fma(nu, add, zc[s2]);
                }
            }
        }
    }
    PPROBE(4);
    // the workgroup's shape-shape columns / traces, added in group order
    if (sl >= 1 && lane_on) {
        double* zr = ZR + (size_t)gid * (K * K + K);
#pragma unroll
        for (int s2 = 0; s2 < (KC ? KC : AVT_MAX_SHAPE); ++s2) if (s2 < K) zr[s2 * K + (sl - 1)] = pair_on ? zc[s2] : 0.0;
        zr[K * K + (sl - 1)] = pair_on ? yx : 0.0;
    }
    __syncthreads();
    double* Zg = scr + mom_off_z(d) + (size_t)bx * (K * K + K);
    for (int e = t; e < K * K + K; e += NTH) {
        double a = 0.0;
#pragma unroll
        for (int g = 0; g < MOM_PP_PAIRS; ++g) a += ZR[(size_t)g * (K * K + K) + e];
        Zg[e] = a;
    }
    PPROBE(5);
}

// the GMM pose prior of the trial point, one workgroup per (component, frame) (avt_prior.h).  A launch of its own: in the pair pass's
// grid every one of these small workgroups would hold that kernel's 52 KB of LDS, a third of a CU
template <bool SM = false>
__global__ __launch_bounds__(128) void k_prior(DeviceModel dm, FrameBuffers fb) {
    if constexpr (SM) AVT_SMPL_PROMISES(dm.d);
    __shared__ double s_scratch[5 * AVT_MAX_JOINTS];
    int bx, fy;
    xcd_frame_block(fb, bx, fy);
    const int f = fy + fb.f0;
    const int2 slot_state = frame_slot_state(fb, f);
    if (slot_state.y == AVT_TRY_DONE) return;      // the frame met the stopping rule in this ICP iteration: no trial point
    prior_component<128>(dm, fb, f, bx, 1 - slot_state.x, s_scratch);
}

static size_t pairpass_lds_bytes(const AvtDims& d) {
    const int TS = mom_tstride(d.mom_npsi), GS = (2 * (15 + 3 * d.K) + d.K + 2) & ~1;
    const size_t pairs = sizeof(double) * ((size_t)MOM_PP_PAIRS * (TS + GS) + (size_t)MOM_PP_PAIRS * (d.K * d.K + d.K)) + 64;
    return std::max(pairs, sizeof(double) * 6 * AVT_MAX_JOINTS + 64);      // (the prior's workgroups in this grid: avt_prior.h's scratch)
}

// =================================================================================================
// k_assemble<KC, NTH>.  grid (1, frames).  The dense system of the frame's trial point from the
// pair results (tools/moment_proto2.py::assemble, phases A' and B).  Everything here is latency, so every phase is many short
// independent items; list walks go four entries at a time (the lists are padded with an index of an all-zero row).
//   B-a  X16 -> LDS; data side (u_k, Dl_k, YF per joint); PR[k][s] = partner sums of the records; Z, sum tr W;
//   B-b  XD_k pieces, per-joint partner sums PK;   B-c  subtree sums TK, TR;   B-d  every block of H.
// =================================================================================================
struct MomTab { const unsigned short *opk_start, *opk, *sub_start, *sub, *bseg, *seg, *jj; };
__host__ __device__ inline int mom_tab_words(const AvtDims& d) { return d.mom_toff[7]; }
__host__ __device__ inline int mom_asm_doubles(const AvtDims& d) {
    const int J = d.J, K = d.K;
    const int rr = (d.mom_nseg + d.mom_nb2) * 16;      // rot-rot: segment sums + block sums
    const int prtr = 2 * (J + 1) * K * 6 > rr ? 2 * (J + 1) * K * 6 : rr;      // PR | TR, later the rot-rot sums
    return (2 * d.mom_np + 1) * 16 + 2 * (J + 1) * 16 + prtr + J * 9 + J * 4 + J * K + (K * K + K) + 2 * K + 8 + K * ((J + 3) / 4);
}

#define MOM_ASM_NTH 1024     // threads of the assembly workgroup: every phase is latency, more items in flight is what helps
#define MOM_MAXOPS 12      // ordered pairs per lever joint the unrolled partner sums cover in one go (longer lists loop)

template <int KC, int NTH>
__global__ __launch_bounds__(NTH) void k_assemble(DeviceModel dm, FrameBuffers fb) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const AvtDims& d = dm.d;
    const int f = blockIdx.y + fb.f0, t = threadIdx.x;
    const int2 slot_state = frame_slot_state(fb, f);
    if (slot_state.y == AVT_TRY_DONE) return;      // the frame met the stopping rule in this ICP iteration: no trial point, no system
    const int try_slot = 1 - slot_state.x;
    const int J = d.J, K = KC ? KC : d.K, S1 = K + 1, NP = d.mom_np, NPSI = d.mom_npsi, P = d.P, HS = d.HS;
    double* skm = (double*)smem;
    double* X16 = skm + mom_skel_doubles(d);    // [2 NP + 1][16]
    double* PK = X16 + (2 * NP + 1) * 16;       // [J + 1][16]  axial(sum W), sum Va, sum Vb, sum t0, axial(XD), Dl; row J: zeros
    double* TK = PK + (J + 1) * 16;             // [J + 1][16]  subtree sums of PK
    double* PR = TK + (J + 1) * 16;             // [J + 1][K][6]
    double* TR = PR + (J + 1) * K * 6;          // [J + 1][K][6]
    const int rrsz = (d.mom_nseg + d.mom_nb2) * 16;
    double* Uk = PR + (2 * (J + 1) * K * 6 > rrsz ? 2 * (J + 1) * K * 6 : rrsz);          // [J][3][3] sum_s om_s Dphi_k[i][s][c]
    double* XQ = Uk + J * 9;                    // [J][4] axial(XD_k), tr XD_k
    double* YFk = XQ + J * 4;                   // [J][K]
    double* ZS = YFk + J * K;                   // [K K + K]
    double* YFs = ZS + K * K + K;               // [K]
    double* misc = YFs + K;                     // [K + 8]
    double* YF4 = misc + K + 8;                 // [K][ceil(J / 4)] YF summed over four joints
    int* s_parent = (int*)(YF4 + K * ((J + 3) / 4));
    unsigned short* tabm = (unsigned short*)(s_parent + AVT_MAX_JOINTS);
    MomSkel sk;
    mom_skel_from_prep<NTH>(d, fb.prep + ((size_t)f * 2 + try_slot) * d.prep_size, fb.ctl[f].centre, skm, s_parent, dm.parent, sk);
    MomTab tb;
    {   // the index lists: one block of 16-bit words, copied eight bytes at a time
        const unsigned long long* src = (const unsigned long long*)dm.mom_tab16;
        unsigned long long* dst = (unsigned long long*)tabm;
        for (int e = t; e < d.mom_toff[7] / 4; e += NTH) dst[e] = src[e];
        tb.opk_start = tabm + d.mom_toff[0]; tb.opk = tabm + d.mom_toff[1]; tb.sub_start = tabm + d.mom_toff[2]; tb.sub = tabm + d.mom_toff[3];
        tb.bseg = tabm + d.mom_toff[4]; tb.seg = tabm + d.mom_toff[5]; tb.jj = tabm + d.mom_toff[6];
    }
    // (every phase is a handful of short items per thread; the loops of a phase start at different threads so that no thread queues one item
    // of each kind behind another)
    auto rot = [&](int k) { return (t + k * (NTH / 4)) & (NTH - 1); };
    const double* scr = fb.mom_rec + (size_t)f * mom_frame_scratch(d);
    const double* REC = scr + mom_off_rec(d);
    const double* Zg = scr + mom_off_z(d);
    const double* Df = fb.mom_D + (size_t)f * J * NPSI * 3;
    double* Hout = fb.Hraw + ((size_t)f * 2 + try_slot) * HS * HS;
    MPROBE(0);
    for (int e = t; e < (2 * NP + 1) * 16; e += NTH) X16[e] = scr[e];
    for (int e = t; e < 16; e += NTH) { PK[J * 16 + e] = 0.0; }
    for (int e = t; e < K * 6; e += NTH) PR[(size_t)J * K * 6 + e] = 0.0;
    for (int e = t; e < K * K + K; e += NTH) {           // the pair-pass workgroups' shape-shape columns / traces, in workgroup order
        double a = 0.0;
        for (int g0 = 0; g0 < mom_nwg(d); g0 += 8) {      // (eight partials requested together)
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = Zg[(size_t)min(g0 + u, mom_nwg(d) - 1) * (K * K + K) + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) a += g0 + u < mom_nwg(d) ? v[u] : 0.0;
        }
        ZS[e] = a;
    }
    __syncthreads();      // skeleton tables, index lists
    MPROBE(1);
    // ---------------- B-a
    for (int e = t; e < J * 9; e += NTH) {      // u_k[i][c] = sum_s om_s D_k[(i,s)][c]
        const int k = e / 9, ic = e - 9 * k, i = ic / 3, c = ic - 3 * i;
        const double* Dk = Df + (size_t)k * NPSI * 3 + (size_t)(S1 * i) * 3 + c;
        double a = 0.0;
        if (KC) {
            double v[KC + 1];
#pragma unroll
            for (int s = 0; s <= KC; ++s) v[s] = Dk[3 * s];
#pragma unroll
            for (int s = 0; s <= KC; ++s) a = fma(sk.om[s], v[s], a);
        } else {
            for (int s = 0; s < S1; ++s) a = fma(sk.om[s], Dk[3 * s], a);
        }
        Uk[e] = a;
    }
    for (int e = rot(1); e < J * 3; e += NTH) { const int k = e / 3, c = e - 3 * k; PK[16 * k + 13 + c] = Df[(size_t)k * NPSI * 3 + (size_t)(NPSI - 1) * 3 + c]; }
    for (int e = rot(2); e < J * K; e += NTH) {      // tr(R_k Dphi_k[s]) + eta_k,s . Dl_k
        const int k = e / K, s = e - k * K;
        const double* Dk = Df + (size_t)k * NPSI * 3;
        const double* Rk = sk.Rw + 9 * k;
        double v[12];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int i = 0; i < 3; ++i) v[3 * c + i] = Dk[(size_t)(S1 * i + s + 1) * 3 + c];
#pragma unroll
        for (int c = 0; c < 3; ++c) v[9 + c] = Dk[(size_t)(NPSI - 1) * 3 + c];
        double q = 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int i = 0; i < 3; ++i) q = fma(Rk[3 * c + i], v[3 * c + i], q);
#pragma unroll
        for (int c = 0; c < 3; ++c) q = fma(sk.eta[((size_t)k * 3 + c) * K + s], v[9 + c], q);
        YFk[e] = q;
    }
    for (int e = rot(3); e < J * K * 6; e += NTH) {  // PR[k][s][.] = sum over the ordered pairs with lever joint k; all loads of an item in flight
        const int k = e / (K * 6), r = e - k * K * 6;
        const int lo = tb.opk_start[k], hi = tb.opk_start[k + 1];
        double v[MOM_MAXOPS];
        int op[MOM_MAXOPS];
#pragma unroll
        for (int u = 0; u < MOM_MAXOPS; ++u) op[u] = tb.opk[max(min(lo + u, hi - 1), 0)];
#pragma unroll
        for (int u = 0; u < MOM_MAXOPS; ++u) v[u] = REC[(size_t)min(op[u], 2 * NP - 1) * K * 6 + r];      // (the padding index 2 NP has no record: masked below)
        double a = 0.0;
#pragma unroll
        for (int u = 0; u < MOM_MAXOPS; ++u) a += (lo + u < hi && op[u] < 2 * NP) ? v[u] : 0.0;
        for (int i = lo + MOM_MAXOPS; i < hi; ++i) if (tb.opk[i] < 2 * NP) a += REC[(size_t)tb.opk[i] * K * 6 + r];
        PR[e] = hi > lo ? a : 0.0;
    }
    __syncthreads();      // X16, Uk
    MPROBE(2);
    // ---------------- B-b
    for (int e = t; e < J * 10; e += NTH) {      // PK[k][0..9]: axial(sum W) 3, sum Va 3, sum Vb 3, sum t0; four pairs per trip (the list is padded with the zero row)
        const int k = e / 10, q = e - 10 * k;
        const int i0 = q == 0 ? 5 : (q == 1 ? 6 : (q == 2 ? 1 : 9 + (q - 3))), i1 = q == 0 ? 7 : (q == 1 ? 2 : 3);      // q < 3: x[i0] - x[i1]; else x[i0] (q = 9: x[15] = t0)
        double a = 0.0;
        for (int i = tb.opk_start[k]; i < tb.opk_start[k + 1]; i += 4) {
            const double* x0 = X16 + (size_t)tb.opk[i] * 16; const double* x1 = X16 + (size_t)tb.opk[i + 1] * 16;
            const double* x2 = X16 + (size_t)tb.opk[i + 2] * 16; const double* x3 = X16 + (size_t)tb.opk[i + 3] * 16;
            const double p0 = x0[i0], p1 = x1[i0], p2 = x2[i0], p3 = x3[i0];
            const double m0 = q < 3 ? x0[i1] : 0.0, m1 = q < 3 ? x1[i1] : 0.0, m2 = q < 3 ? x2[i1] : 0.0, m3 = q < 3 ? x3[i1] : 0.0;
            a += ((p0 - m0) + (p1 - m1)) + ((p2 - m2) + (p3 - m3));
        }
        PK[16 * k + q] = a;
    }
    for (int e = rot(1); e < J * 4; e += NTH) {       // XD_k = R_k u_k + tau_k Dl_k^T: its axial vector (PK[10..12]) and trace
        const int k = e >> 2, q = e & 3;
        const double* Rk = sk.Rw + 9 * k;
        const double* u = Uk + 9 * k;
        const double* dl = PK + 16 * k + 13;
        auto xd = [&](int r, int c) { return (Rk[3 * r] * u[c] + Rk[3 * r + 1] * u[3 + c] + Rk[3 * r + 2] * u[6 + c]) + sk.tau[3 * k + r] * dl[c]; };
        const double v = q == 0 ? xd(1, 2) - xd(2, 1) : (q == 1 ? xd(2, 0) - xd(0, 2) : (q == 2 ? xd(0, 1) - xd(1, 0) : (xd(0, 0) + xd(1, 1)) + xd(2, 2)));
        if (q < 3) PK[16 * k + 10 + q] = v;
        XQ[e] = v;
    }
    for (int e = rot(2); e < K * ((J + 3) / 4); e += NTH) {      // YF over four joints at a time (summed up in the next phase)
        const int g4 = (J + 3) / 4, s = e / g4, g = e - s * g4;
        const double v0 = YFk[min(4 * g, J - 1) * K + s], v1 = YFk[min(4 * g + 1, J - 1) * K + s], v2 = YFk[min(4 * g + 2, J - 1) * K + s], v3 = YFk[min(4 * g + 3, J - 1) * K + s];
        YF4[e] = ((4 * g < J ? v0 : 0.0) + (4 * g + 1 < J ? v1 : 0.0)) + ((4 * g + 2 < J ? v2 : 0.0) + (4 * g + 3 < J ? v3 : 0.0));
    }
    if (t >= NTH - 64) {      // sum tr W over the ordered pairs, fixed order (one wave)
        const int l = t - (NTH - 64);
        double a = 0.0;
        for (int op = l; op < 2 * NP; op += 64) { const double* x = X16 + (size_t)op * 16; a += (x[0] + x[4]) + x[8]; }
        a = wave_sum(a);
        if (l == 0) misc[1] = a;
    }
    __syncthreads();
    MPROBE(3);
    // ---------------- B-c: subtree sums, eight list entries at a time
    for (int e = t; e < J * 16; e += NTH) {
        const int j = e >> 4, q = e & 15;
        double a = 0.0;
        for (int i = tb.sub_start[j]; i < tb.sub_start[j + 1]; i += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = PK[16 * tb.sub[i + u] + q];
            a += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
        TK[e] = a;
    }
    for (int e = rot(1); e < J * K * 6; e += NTH) {
        const int j = e / (K * 6), r = e - j * K * 6;
        double a = 0.0;
        for (int i = tb.sub_start[j]; i < tb.sub_start[j + 1]; i += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = PR[(size_t)tb.sub[i + u] * K * 6 + r];
            a += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
        TR[e] = a;
    }
    if (t >= NTH - 64 && t - (NTH - 64) < K) { const int sq = t - (NTH - 64), g4 = (J + 3) / 4; double a = 0.0; for (int g = 0; g < g4; ++g) a += YF4[sq * g4 + g]; YFs[sq] = a; }
    if (t == NTH / 2) {
        double xf = 0.0;
        for (int k = 0; k < J; ++k) xf += XQ[4 * k + 3];
        misc[0] = misc[1] - 2.0 * xf;      // (+ sum |d_i - centre|^2: the constant part, k_moments)
    }
    __syncthreads();
    MPROBE(4);
    // ---------------- B-d: the blocks of H
    auto Hset = [&](int r, int c, double v) { Hout[(size_t)r * HS + c] = v; Hout[(size_t)c * HS + r] = v; };
    const int SH = 3 + 3 * J;
    if (t < 9) { const int r = t / 3, c = t - 3 * r; Hout[(size_t)r * HS + c] = r == c ? TK[9] : 0.0; }
    if (t < 3) Hset(t, P, TK[3 + t] - TK[13 + t]);
    if (t == 3) Hout[(size_t)P * HS + P] = misc[0];
    for (int e = t; e < J * 3; e += NTH) {      // rotation rows against translation and residual
        const int j = e / 3, c = e - 3 * j, pa = sk.parent[j];
        const double a0 = pa < 0 ? (c == 0 ? 1.0 : 0.0) : sk.Rw[9 * pa + c], a1 = pa < 0 ? (c == 1 ? 1.0 : 0.0) : sk.Rw[9 * pa + 3 + c],
                     a2 = pa < 0 ? (c == 2 ? 1.0 : 0.0) : sk.Rw[9 * pa + 6 + c];        // column c of R_par(j)
        const double* tk = TK + 16 * j;
        const double o0 = sk.oc[3 * j], o1 = sk.oc[3 * j + 1], o2 = sk.oc[3 * j + 2];
        const double l0 = tk[3] - tk[9] * o0, l1 = tk[4] - tk[9] * o1, l2 = tk[5] - tk[9] * o2;
        const int r = 3 + 3 * j + c;
        Hset(r, 0, 2.0 * (a1 * l2 - a2 * l1)); Hset(r, 1, 2.0 * (a2 * l0 - a0 * l2)); Hset(r, 2, 2.0 * (a0 * l1 - a1 * l0));
        // lr = (axial(sum W) - o x sum Vb) - (axial(XD) - o x Dl)
        const double b0 = tk[6] - tk[13], b1 = tk[7] - tk[14], b2 = tk[8] - tk[15];
        const double r0 = (tk[0] - tk[10]) - (o1 * b2 - o2 * b1), r1 = (tk[1] - tk[11]) - (o2 * b0 - o0 * b2), r2 = (tk[2] - tk[12]) - (o0 * b1 - o1 * b0);
        Hset(r, P, 2.0 * (a0 * r0 + a1 * r1 + a2 * r2));
    }
    for (int e = t; e < J * 3 * K; e += NTH) {  // rotation rows against the shape columns
        const int j = e / (3 * K), rem = e - j * 3 * K, c = rem / K, s = rem - c * K, pa = sk.parent[j];
        const double a0 = pa < 0 ? (c == 0 ? 1.0 : 0.0) : sk.Rw[9 * pa + c], a1 = pa < 0 ? (c == 1 ? 1.0 : 0.0) : sk.Rw[9 * pa + 3 + c],
                     a2 = pa < 0 ? (c == 2 ? 1.0 : 0.0) : sk.Rw[9 * pa + 6 + c];
        const double* tr = TR + ((size_t)j * K + s) * 6;
        const double o0 = sk.oc[3 * j], o1 = sk.oc[3 * j + 1], o2 = sk.oc[3 * j + 2];
        const double v0 = tr[0] - (o1 * tr[5] - o2 * tr[4]), v1 = tr[1] - (o2 * tr[3] - o0 * tr[5]), v2 = tr[2] - (o0 * tr[4] - o1 * tr[3]);
        Hset(3 + 3 * j + c, SH + s, 2.0 * (a0 * v0 + a1 * v1 + a2 * v2));
    }
    for (int e = t; e < K * 3; e += NTH) { const int s = e / 3, c = e - 3 * s; Hset(SH + s, c, TR[(size_t)s * 6 + 3 + c]); }      // root's subtree = every joint
    if (t < K) Hset(SH + t, P, ZS[K * K + t] - YFs[t]);
    for (int e = t; e < K * K; e += NTH) { const int s = e / K, u = e - s * K; Hout[(size_t)(SH + s) * HS + SH + u] = ZS[s * K + u] + ZS[u * K + s]; }
    MPROBE(5);
    // rot-rot.  Block (j <= j') = sum over the ordered pairs (k under j, k' under j') of X16; only the blocks with such pairs are
    // listed (left leg against right arm: none), the others are structural zeros.  Phase 1: the sums, one item per (block, entry
    // of X16), into the PR / TR area (free now); phase 2: one item per (block, matrix entry).
    __syncthreads();
    double* SEG = PR;                          // [nseg][16] sums of the 16-entry segments ...
    double* S16 = PR + d.mom_nseg * 16;        // [nb2][16]  ... and of the blocks (the PR | TR area is sized for both)
    for (int e = t; e < d.mom_nseg * 16; e += NTH) {
        const int sg = e >> 4, q = e & 15;
        const unsigned short* li = tb.seg + 16 * sg;
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = X16[(size_t)li[u] * 16 + q];
        SEG[e] = (((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]))) + (((v[8] + v[9]) + (v[10] + v[11])) + ((v[12] + v[13]) + (v[14] + v[15])));
    }
    for (int e = rot(2); e < d.mom_nz2 * 9; e += NTH) {      // the structural zeros
        const int b = e / 9, q = e - 9 * b, jj = dm.mom_z2_jj[b], j = jj & 0xff, jp = jj >> 8, r = q / 3, c = q - 3 * r;
        Hout[(size_t)(3 + 3 * j + r) * HS + 3 + 3 * jp + c] = 0.0;
        Hout[(size_t)(3 + 3 * jp + c) * HS + 3 + 3 * j + r] = 0.0;
    }
    __syncthreads();
    for (int e = t; e < d.mom_nb2 * 16; e += NTH) {      // a block's segments, in order
        const int b = e >> 4, q = e & 15;
        const int s0 = tb.bseg[b], s1 = tb.bseg[b + 1];
        double a = 0.0;
        for (int sg = s0; sg < s1; sg += 4) {
            const double v0 = SEG[sg * 16 + q], v1 = SEG[min(sg + 1, s1 - 1) * 16 + q], v2 = SEG[min(sg + 2, s1 - 1) * 16 + q], v3 = SEG[min(sg + 3, s1 - 1) * 16 + q];
            a += v0; a += sg + 1 < s1 ? v1 : 0.0; a += sg + 2 < s1 ? v2 : 0.0; a += sg + 3 < s1 ? v3 : 0.0;
        }
        S16[e] = a;
    }
    __syncthreads();
    for (int e = t; e < d.mom_nb2 * 9; e += NTH) {
        const int b = e / 9, q = e - 9 * b;
        const int jj = tb.jj[b], j = jj & 0xff, jp = jj >> 8;
        const double* S = S16 + b * 16;
        const double oj[3] = {sk.oc[3 * j], sk.oc[3 * j + 1], sk.oc[3 * j + 2]}, op[3] = {sk.oc[3 * jp], sk.oc[3 * jp + 1], sk.oc[3 * jp + 2]};
        // LL = S.W - S.Va o_j'^T - o_j S.Vb^T + S.t0 o_j o_j'^T   (sum over the block of V_k'k = sum of Vb)
        double LL[9];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) LL[3 * r + c] = S[3 * r + c] - S[9 + r] * op[c] - oj[r] * S[12 + c] + S[15] * oj[r] * op[c];
        const double trL = (LL[0] + LL[4]) + LL[8];
        const int pj = sk.parent[j], pp = sk.parent[jp];
        const int r = q / 3, c = q - 3 * r;
        // entry (r, c) of 4 (tr(LL) A^T B - A^T LL^T B),  A = R_par(j), B = R_par(j')
        double Ar[3], Bc[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            Ar[i] = pj < 0 ? (i == r ? 1.0 : 0.0) : sk.Rw[9 * pj + 3 * i + r];
            Bc[i] = pp < 0 ? (i == c ? 1.0 : 0.0) : sk.Rw[9 * pp + 3 * i + c];
        }
        const double atb = Ar[0] * Bc[0] + Ar[1] * Bc[1] + Ar[2] * Bc[2];
        double atl = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double ltb = LL[i] * Bc[0] + LL[3 + i] * Bc[1] + LL[6 + i] * Bc[2];      // (LL^T B)[i][c]
            atl = fma(Ar[i], ltb, atl);
        }
        const double v = 4.0 * (trL * atb - atl);
        if (j != jp || r <= c) {      // (a diagonal block: the upper triangle, mirrored - symmetric to the bit)
            Hout[(size_t)(3 + 3 * j + r) * HS + 3 + 3 * jp + c] = v;
            Hout[(size_t)(3 + 3 * jp + c) * HS + 3 + 3 * j + r] = v;
        }
    }
    MPROBE(6);
}

// =================================================================================================
// k_assemble_parts<KC>.  grid (MOM_ASM_ROLES, frames), 256 threads.  The same assembly as k_assemble, item for item and bit for bit, as
// SIX independent workgroups per frame instead of one 1024-thread workgroup walking six phases in a row (35 k clocks: staging 4.7 k,
// data side + partner sums 6 k, per-joint sums 5 k, subtree sums 4 k, blocks 3.5 k, rot-rot 12 k):
//   role 0            "core": the X16 / data side per-joint sums PK -> subtree sums TK -> translation block, rotation rows against translation
//                     and residual, shape rows against the residual, shape-shape; 
//   roles 1 .. NA     the shape keys [s0, s1): partner sums PR of the pair pass's records -> subtree sums TR -> rotation rows against those
//                     shape columns, shape rows against the translation;
//   roles NA+1 ..     the rot-rot blocks [AvtDims::mom_rsplit[r], mom_rsplit[r + 1]) - ranges of equal segment counts - and a share of the
//                     structural zeros.  They depend on the staged X16 alone, which is why they need not wait for the tree sums.
// No role reads what another writes: every one goes from the pair pass's scratch to its own entries of H.  The dependency chain of a GN
// iteration is the longest role (the core: five short phases) instead of the sum of all phases; a 256-thread workgroup with <= 46 KB of LDS
// finds a place on a CU another frame group's kernels are using, which the 1024-thread, 80 KB workgroup did not.
// =================================================================================================
#define MOM_ASM_NA 2
#define MOM_ASM_NR 3
#define MOM_ASM_ROLES (1 + MOM_ASM_NA + MOM_ASM_NR)
#define MOM_PARTS_NTH 256

// world rotations, joint origins relative to the frame centre and the parents: what the shape and rot-rot roles need of the skeleton
__device__ __forceinline__ void mom_skel_light(const AvtDims& d, const double* __restrict__ prep, const double* centre, double* __restrict__ Rw, double* __restrict__ oc,
                                               int* __restrict__ s_parent, const int* __restrict__ parent_g) {
    const int J = d.J, t = threadIdx.x;
    for (int e = t; e < 9 * J; e += MOM_PARTS_NTH) Rw[e] = prep[prep_off_Rw(d) + e];
    for (int e = t; e < 3 * J; e += MOM_PARTS_NTH) { const int j = e / 3, r = e - 3 * j; oc[e] = prep[prep_off_o(d) + e] - centre[r]; }
    if (t < J) s_parent[t] = parent_g[t];
}

template <int KC>
__device__ __forceinline__ void asm_role_core(const DeviceModel& dm, const FrameBuffers& fb, int f, char* smem) {
    constexpr int NTH = MOM_PARTS_NTH;
    const AvtDims& d = dm.d;
    const int2 slot_state = frame_slot_state(fb, f);
    if (slot_state.y == AVT_TRY_DONE) return;      // the frame met the stopping rule in this ICP iteration: no trial point, no system
    const int t = threadIdx.x, try_slot = 1 - slot_state.x;
    const int J = d.J, K = KC ? KC : d.K, S1 = K + 1, NP = d.mom_np, NPSI = d.mom_npsi, P = d.P, HS = d.HS;
    double* skm = (double*)smem;
    double* PK = skm + mom_skel_doubles(d);     // [J + 1][16]
    double* TK = PK + (J + 1) * 16;             // [J + 1][16]
    double* Uk = TK + (J + 1) * 16;             // [J][9]
    double* XQ = Uk + J * 9;                    // [J][4]
    double* YFk = XQ + J * 4;                   // [J][K]
    double* ZS = YFk + J * K;                   // [K K + K]
    double* YFs = ZS + K * K + K;               // [K]
    double* misc = YFs + K;                     // [K + 8]
    double* YF4 = misc + K + 8;                 // [K][ceil(J / 4)]
    int* s_parent = (int*)(YF4 + K * ((J + 3) / 4));
    unsigned short* tabm = (unsigned short*)(s_parent + AVT_MAX_JOINTS);      // words [0, mom_toff[4]): opk_start | opk | sub_start | sub
    MomSkel sk;
    mom_skel_from_prep<NTH>(d, fb.prep + ((size_t)f * 2 + try_slot) * d.prep_size, fb.ctl[f].centre, skm, s_parent, dm.parent, sk);
    MomTab tb;
    {
        const unsigned long long* src = (const unsigned long long*)dm.mom_tab16;
        unsigned long long* dst = (unsigned long long*)tabm;
        for (int e = t; e < d.mom_toff[4] / 4; e += NTH) dst[e] = src[e];
        tb.opk_start = tabm + d.mom_toff[0]; tb.opk = tabm + d.mom_toff[1]; tb.sub_start = tabm + d.mom_toff[2]; tb.sub = tabm + d.mom_toff[3];
        tb.bseg = tb.seg = tb.jj = nullptr;
    }
    auto rot = [&](int k) { return (t + k * (NTH / 4)) & (NTH - 1); };
    const double* scr = fb.mom_rec + (size_t)f * mom_frame_scratch(d);
    const double* X16 = scr;                    // [2 NP + 1][16] read where the pair pass left it (one 128-byte line per ordered pair, L2): staging it costs 21.6 KB of LDS per workgroup
    const double* Zg = scr + mom_off_z(d);
    const double* Df = fb.mom_D + (size_t)f * J * NPSI * 3;
    double* Hout = fb.Hraw + ((size_t)f * 2 + try_slot) * HS * HS;
    for (int e = t; e < 16; e += NTH) { PK[J * 16 + e] = 0.0; }
    for (int e = t; e < K * K + K; e += NTH) {           // the pair-pass workgroups' shape-shape columns / traces, in workgroup order
        double a = 0.0;
        for (int g0 = 0; g0 < mom_nwg(d); g0 += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = Zg[(size_t)min(g0 + u, mom_nwg(d) - 1) * (K * K + K) + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) a += g0 + u < mom_nwg(d) ? v[u] : 0.0;
        }
        ZS[e] = a;
    }
    // ---------------- B-a (the data side).  The D_k entries of this thread's first item are requested in front of the barrier (they depend on
    // nothing staged): one L2 round trip hidden behind the staging
    double uk_v[KC ? KC + 1 : 1];
    if (KC && t < J * 9) {
        const int k = t / 9, ic = t - 9 * k, i = ic / 3, c = ic - 3 * i;
        const double* Dk = Df + (size_t)k * NPSI * 3 + (size_t)(S1 * i) * 3 + c;
#pragma unroll
        for (int s = 0; s <= KC; ++s) uk_v[s] = Dk[3 * s];
    }
    __syncthreads();      // skeleton tables, index lists
    if (KC && t < J * 9) {      // u_k[i][c] = sum_s om_s D_k[(i,s)][c]
        double a = 0.0;
#pragma unroll
        for (int s = 0; s <= KC; ++s) a = fma(sk.om[s], uk_v[s], a);
        Uk[t] = a;
    }
    for (int e = KC ? t + NTH : t; e < J * 9; e += NTH) {
        const int k = e / 9, ic = e - 9 * k, i = ic / 3, c = ic - 3 * i;
        const double* Dk = Df + (size_t)k * NPSI * 3 + (size_t)(S1 * i) * 3 + c;
        double a = 0.0;
        for (int s = 0; s < S1; ++s) a = fma(sk.om[s], Dk[3 * s], a);
        Uk[e] = a;
    }
    for (int e = rot(1); e < J * 3; e += NTH) { const int k = e / 3, c = e - 3 * k; PK[16 * k + 13 + c] = Df[(size_t)k * NPSI * 3 + (size_t)(NPSI - 1) * 3 + c]; }
    for (int e = rot(2); e < J * K; e += NTH) {      // tr(R_k Dphi_k[s]) + eta_k,s . Dl_k
        const int k = e / K, s = e - k * K;
        const double* Dk = Df + (size_t)k * NPSI * 3;
        const double* Rk = sk.Rw + 9 * k;
        double v[12];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int i = 0; i < 3; ++i) v[3 * c + i] = Dk[(size_t)(S1 * i + s + 1) * 3 + c];
#pragma unroll
        for (int c = 0; c < 3; ++c) v[9 + c] = Dk[(size_t)(NPSI - 1) * 3 + c];
        double q = 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int i = 0; i < 3; ++i) q = fma(Rk[3 * c + i], v[3 * c + i], q);
#pragma unroll
        for (int c = 0; c < 3; ++c) q = fma(sk.eta[((size_t)k * 3 + c) * K + s], v[9 + c], q);
        YFk[e] = q;
    }
    // (B-b's partner sums of X16 read nothing B-a writes: same phase)
    for (int e = rot(3); e < J * 10; e += NTH) {      // PK[k][0..9]: axial(sum W) 3, sum Va 3, sum Vb 3, sum t0
        const int k = e / 10, q = e - 10 * k;
        const int i0 = q == 0 ? 5 : (q == 1 ? 6 : (q == 2 ? 1 : 9 + (q - 3))), i1 = q == 0 ? 7 : (q == 1 ? 2 : 3);
        double a = 0.0;
        for (int i = tb.opk_start[k]; i < tb.opk_start[k + 1]; i += 4) {
            const double* x0 = X16 + (size_t)tb.opk[i] * 16; const double* x1 = X16 + (size_t)tb.opk[i + 1] * 16;
            const double* x2 = X16 + (size_t)tb.opk[i + 2] * 16; const double* x3 = X16 + (size_t)tb.opk[i + 3] * 16;
            const double p0 = x0[i0], p1 = x1[i0], p2 = x2[i0], p3 = x3[i0];
            const double m0 = q < 3 ? x0[i1] : 0.0, m1 = q < 3 ? x1[i1] : 0.0, m2 = q < 3 ? x2[i1] : 0.0, m3 = q < 3 ? x3[i1] : 0.0;
            a += ((p0 - m0) + (p1 - m1)) + ((p2 - m2) + (p3 - m3));
        }
        PK[16 * k + q] = a;
    }
    if (t >= NTH - 64) {      // sum tr W over the ordered pairs, fixed order (one wave)
        const int l = t - (NTH - 64);
        double a = 0.0;
        for (int op = l; op < 2 * NP; op += 64) { const double* x = X16 + (size_t)op * 16; a += (x[0] + x[4]) + x[8]; }
        a = wave_sum(a);
        if (l == 0) misc[1] = a;
    }
    __syncthreads();      // Uk, Dl, YFk
    // ---------------- B-b
    for (int e = t; e < J * 4; e += NTH) {       // XD_k = R_k u_k + tau_k Dl_k^T: its axial vector (PK[10..12]) and trace
        const int k = e >> 2, q = e & 3;
        const double* Rk = sk.Rw + 9 * k;
        const double* u = Uk + 9 * k;
        const double* dl = PK + 16 * k + 13;
        auto xd = [&](int r, int c) { return (Rk[3 * r] * u[c] + Rk[3 * r + 1] * u[3 + c] + Rk[3 * r + 2] * u[6 + c]) + sk.tau[3 * k + r] * dl[c]; };
        const double v = q == 0 ? xd(1, 2) - xd(2, 1) : (q == 1 ? xd(2, 0) - xd(0, 2) : (q == 2 ? xd(0, 1) - xd(1, 0) : (xd(0, 0) + xd(1, 1)) + xd(2, 2)));
        if (q < 3) PK[16 * k + 10 + q] = v;
        XQ[e] = v;
    }
    for (int e = rot(2); e < K * ((J + 3) / 4); e += NTH) {      // YF over four joints at a time
        const int g4 = (J + 3) / 4, s = e / g4, g = e - s * g4;
        const double v0 = YFk[min(4 * g, J - 1) * K + s], v1 = YFk[min(4 * g + 1, J - 1) * K + s], v2 = YFk[min(4 * g + 2, J - 1) * K + s], v3 = YFk[min(4 * g + 3, J - 1) * K + s];
        YF4[e] = ((4 * g < J ? v0 : 0.0) + (4 * g + 1 < J ? v1 : 0.0)) + ((4 * g + 2 < J ? v2 : 0.0) + (4 * g + 3 < J ? v3 : 0.0));
    }
    __syncthreads();
    // ---------------- B-c: subtree sums, eight list entries at a time
    for (int e = t; e < J * 16; e += NTH) {
        const int j = e >> 4, q = e & 15;
        double a = 0.0;
        for (int i = tb.sub_start[j]; i < tb.sub_start[j + 1]; i += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = PK[16 * tb.sub[i + u] + q];
            a += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
        TK[e] = a;
    }
    if (t >= NTH - 64 && t - (NTH - 64) < K) { const int sq = t - (NTH - 64), g4 = (J + 3) / 4; double a = 0.0; for (int g = 0; g < g4; ++g) a += YF4[sq * g4 + g]; YFs[sq] = a; }
    if (t == NTH / 2) {
        double xf = 0.0;
        for (int k = 0; k < J; ++k) xf += XQ[4 * k + 3];
        misc[0] = misc[1] - 2.0 * xf;
    }
    __syncthreads();
    // ---------------- B-d: this role's blocks of H
    auto Hset = [&](int r, int c, double v) { Hout[(size_t)r * HS + c] = v; Hout[(size_t)c * HS + r] = v; };
    const int SH = 3 + 3 * J;
    if (t < 9) { const int r = t / 3, c = t - 3 * r; Hout[(size_t)r * HS + c] = r == c ? TK[9] : 0.0; }
    if (t < 3) Hset(t, P, TK[3 + t] - TK[13 + t]);
    if (t == 3) Hout[(size_t)P * HS + P] = misc[0];
    for (int e = rot(1); e < J * 3; e += NTH) {      // rotation rows against translation and residual
        const int j = e / 3, c = e - 3 * j, pa = sk.parent[j];
        const double a0 = pa < 0 ? (c == 0 ? 1.0 : 0.0) : sk.Rw[9 * pa + c], a1 = pa < 0 ? (c == 1 ? 1.0 : 0.0) : sk.Rw[9 * pa + 3 + c],
                     a2 = pa < 0 ? (c == 2 ? 1.0 : 0.0) : sk.Rw[9 * pa + 6 + c];
        const double* tk = TK + 16 * j;
        const double o0 = sk.oc[3 * j], o1 = sk.oc[3 * j + 1], o2 = sk.oc[3 * j + 2];
        const double l0 = tk[3] - tk[9] * o0, l1 = tk[4] - tk[9] * o1, l2 = tk[5] - tk[9] * o2;
        const int r = 3 + 3 * j + c;
        Hset(r, 0, 2.0 * (a1 * l2 - a2 * l1)); Hset(r, 1, 2.0 * (a2 * l0 - a0 * l2)); Hset(r, 2, 2.0 * (a0 * l1 - a1 * l0));
        const double b0 = tk[6] - tk[13], b1 = tk[7] - tk[14], b2 = tk[8] - tk[15];
        const double r0 = (tk[0] - tk[10]) - (o1 * b2 - o2 * b1), r1 = (tk[1] - tk[11]) - (o2 * b0 - o0 * b2), r2 = (tk[2] - tk[12]) - (o0 * b1 - o1 * b0);
        Hset(r, P, 2.0 * (a0 * r0 + a1 * r1 + a2 * r2));
    }
    if (t >= NTH - 64 && t - (NTH - 64) < K) { const int s = t - (NTH - 64); Hset(SH + s, P, ZS[K * K + s] - YFs[s]); }
    for (int e = rot(2); e < K * K; e += NTH) { const int s = e / K, u = e - s * K; Hout[(size_t)(SH + s) * HS + SH + u] = ZS[s * K + u] + ZS[u * K + s]; }
}

template <int KC>
__device__ __forceinline__ void asm_role_shape(const DeviceModel& dm, const FrameBuffers& fb, int f, int h, char* smem) {
    constexpr int NTH = MOM_PARTS_NTH;
    const AvtDims& d = dm.d;
    const int2 slot_state = frame_slot_state(fb, f);
    if (slot_state.y == AVT_TRY_DONE) return;      // the frame met the stopping rule in this ICP iteration: no trial point, no system
    const int t = threadIdx.x, try_slot = 1 - slot_state.x;
    const int J = d.J, K = KC ? KC : d.K, NP = d.mom_np, HS = d.HS;
    const int s0 = (h * K) / MOM_ASM_NA, s1 = ((h + 1) * K) / MOM_ASM_NA, Kh = s1 - s0;
    double* Rw = (double*)smem;                 // [J][9]
    double* oc = Rw + 9 * J;                    // [J][3]
    double* PR = oc + 3 * J + (J & 1);          // [J + 1][Kh][6]
    double* TR = PR + (J + 1) * Kh * 6;         // [J][Kh][6]
    int* s_parent = (int*)(TR + J * Kh * 6);
    unsigned short* tabm = (unsigned short*)(s_parent + AVT_MAX_JOINTS);
    mom_skel_light(d, fb.prep + ((size_t)f * 2 + try_slot) * d.prep_size, fb.ctl[f].centre, Rw, oc, s_parent, dm.parent);
    {
        const unsigned long long* src = (const unsigned long long*)dm.mom_tab16;
        unsigned long long* dst = (unsigned long long*)tabm;
        for (int e = t; e < d.mom_toff[4] / 4; e += NTH) dst[e] = src[e];
    }
    const unsigned short* opk_start = tabm + d.mom_toff[0], *opk = tabm + d.mom_toff[1], *sub_start = tabm + d.mom_toff[2], *sub = tabm + d.mom_toff[3];
    const double* scr = fb.mom_rec + (size_t)f * mom_frame_scratch(d);
    const double* REC = scr + mom_off_rec(d);
    double* Hout = fb.Hraw + ((size_t)f * 2 + try_slot) * HS * HS;
    for (int e = t; e < Kh * 6; e += NTH) PR[(size_t)J * Kh * 6 + e] = 0.0;      // the all-zero row the padded subtree lists point at
    __syncthreads();      // index lists
    for (int e = t; e < J * Kh * 6; e += NTH) {  // PR[k][s][.] = sum over the ordered pairs with lever joint k; all loads of an item in flight
        const int k = e / (Kh * 6), rl = e - k * Kh * 6, r = s0 * 6 + rl;      // r: offset inside a pair's [K][6] record block
        const int lo = opk_start[k], hi = opk_start[k + 1];
        double v[MOM_MAXOPS];
        int op[MOM_MAXOPS];
#pragma unroll
        for (int u = 0; u < MOM_MAXOPS; ++u) op[u] = opk[max(min(lo + u, hi - 1), 0)];
#pragma unroll
        for (int u = 0; u < MOM_MAXOPS; ++u) v[u] = REC[(size_t)min(op[u], 2 * NP - 1) * K * 6 + r];
        double a = 0.0;
#pragma unroll
        for (int u = 0; u < MOM_MAXOPS; ++u) a += (lo + u < hi && op[u] < 2 * NP) ? v[u] : 0.0;
        for (int i = lo + MOM_MAXOPS; i < hi; ++i) if (opk[i] < 2 * NP) a += REC[(size_t)opk[i] * K * 6 + r];
        PR[e] = hi > lo ? a : 0.0;
    }
    __syncthreads();
    for (int e = t; e < J * Kh * 6; e += NTH) {  // subtree sums, eight list entries at a time
        const int j = e / (Kh * 6), r = e - j * Kh * 6;
        double a = 0.0;
        for (int i = sub_start[j]; i < sub_start[j + 1]; i += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = PR[(size_t)sub[i + u] * Kh * 6 + r];
            a += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
        TR[e] = a;
    }
    __syncthreads();
    auto Hset = [&](int r, int c, double v) { Hout[(size_t)r * HS + c] = v; Hout[(size_t)c * HS + r] = v; };
    const int SH = 3 + 3 * J;
    for (int e = t; e < J * 3 * Kh; e += NTH) {  // rotation rows against this role's shape columns
        const int j = e / (3 * Kh), rem = e - j * 3 * Kh, c = rem / Kh, sl = rem - c * Kh, pa = s_parent[j];
        const double a0 = pa < 0 ? (c == 0 ? 1.0 : 0.0) : Rw[9 * pa + c], a1 = pa < 0 ? (c == 1 ? 1.0 : 0.0) : Rw[9 * pa + 3 + c],
                     a2 = pa < 0 ? (c == 2 ? 1.0 : 0.0) : Rw[9 * pa + 6 + c];
        const double* tr = TR + ((size_t)j * Kh + sl) * 6;
        const double o0 = oc[3 * j], o1 = oc[3 * j + 1], o2 = oc[3 * j + 2];
        const double v0 = tr[0] - (o1 * tr[5] - o2 * tr[4]), v1 = tr[1] - (o2 * tr[3] - o0 * tr[5]), v2 = tr[2] - (o0 * tr[4] - o1 * tr[3]);
        Hset(3 + 3 * j + c, SH + s0 + sl, 2.0 * (a0 * v0 + a1 * v1 + a2 * v2));
    }
    for (int e = t; e < Kh * 3; e += NTH) { const int sl = e / 3, c = e - 3 * sl; Hset(SH + s0 + sl, c, TR[(size_t)sl * 6 + 3 + c]); }      // root's subtree = every joint
}

__device__ __forceinline__ void asm_role_rotrot(const DeviceModel& dm, const FrameBuffers& fb, int f, int r, char* smem) {
    constexpr int NTH = MOM_PARTS_NTH;
    const AvtDims& d = dm.d;
    const int2 slot_state = frame_slot_state(fb, f);
    if (slot_state.y == AVT_TRY_DONE) return;      // the frame met the stopping rule in this ICP iteration: no trial point, no system
    const int t = threadIdx.x, try_slot = 1 - slot_state.x;
    const int J = d.J, NP = d.mom_np, HS = d.HS;
    const int b0 = d.mom_rsplit[r], b1 = d.mom_rsplit[r + 1];
    double* Rw = (double*)smem;                 // [J][9]
    double* oc = Rw + 9 * J;                    // [J][3]
    int* s_parent = (int*)(oc + 3 * J + (J & 1));
    unsigned short* tabm = (unsigned short*)(s_parent + AVT_MAX_JOINTS);      // words [mom_toff[4], mom_toff[7]): bseg | seg | jj
    const int w0 = d.mom_toff[4], nw = d.mom_toff[7] - w0;
    double* SEG = (double*)(tabm + ((nw + 3) & ~3));      // [own segments][16]
    const unsigned short* bseg = tabm + (d.mom_toff[4] - w0), *seg = tabm + (d.mom_toff[5] - w0), *jjl = tabm + (d.mom_toff[6] - w0);
    mom_skel_light(d, fb.prep + ((size_t)f * 2 + try_slot) * d.prep_size, fb.ctl[f].centre, Rw, oc, s_parent, dm.parent);
    {
        const unsigned long long* src = (const unsigned long long*)(dm.mom_tab16 + w0);      // (every list starts on a multiple of four words)
        unsigned long long* dst = (unsigned long long*)tabm;
        for (int e = t; e < nw / 4; e += NTH) dst[e] = src[e];
    }
    const double* X16 = fb.mom_rec + (size_t)f * mom_frame_scratch(d);      // [2 NP + 1][16] read in place: the 16 lanes of an item group share one 128-byte row
    double* Hout = fb.Hraw + ((size_t)f * 2 + try_slot) * HS * HS;
    {   // this role's share of the structural zeros (they depend on nothing)
        const int z0 = (int)(((long long)d.mom_nz2 * 9 * r) / MOM_ASM_NR), z1 = (int)(((long long)d.mom_nz2 * 9 * (r + 1)) / MOM_ASM_NR);
        for (int e = z0 + t; e < z1; e += NTH) {
            const int b = e / 9, q = e - 9 * b, jj = dm.mom_z2_jj[b], j = jj & 0xff, jp = jj >> 8, rr = q / 3, c = q - 3 * rr;
            Hout[(size_t)(3 + 3 * j + rr) * HS + 3 + 3 * jp + c] = 0.0;
            Hout[(size_t)(3 + 3 * jp + c) * HS + 3 + 3 * j + rr] = 0.0;
        }
    }
    __syncthreads();      // lists, skeleton
    const int sg0 = bseg[b0], sg1 = bseg[b1], nsg = sg1 - sg0, nb = b1 - b0;
    double* S16 = SEG + (size_t)nsg * 16;       // [own blocks][16]
    for (int e = t; e < nsg * 16; e += NTH) {
        const int sg = sg0 + (e >> 4), q = e & 15;
        const unsigned short* li = seg + 16 * sg;
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = X16[(size_t)li[u] * 16 + q];
        SEG[e] = (((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]))) + (((v[8] + v[9]) + (v[10] + v[11])) + ((v[12] + v[13]) + (v[14] + v[15])));
    }
    __syncthreads();
    for (int e = t; e < nb * 16; e += NTH) {      // a block's segments, in order
        const int b = b0 + (e >> 4), q = e & 15;
        const int s0 = bseg[b] - sg0, s1 = bseg[b + 1] - sg0;
        double a = 0.0;
        for (int sg = s0; sg < s1; sg += 4) {
            const double v0 = SEG[sg * 16 + q], v1 = SEG[min(sg + 1, s1 - 1) * 16 + q], v2 = SEG[min(sg + 2, s1 - 1) * 16 + q], v3 = SEG[min(sg + 3, s1 - 1) * 16 + q];
            a += v0; a += sg + 1 < s1 ? v1 : 0.0; a += sg + 2 < s1 ? v2 : 0.0; a += sg + 3 < s1 ? v3 : 0.0;
        }
        S16[e] = a;
    }
    __syncthreads();
    for (int e = t; e < nb * 9; e += NTH) {
        const int bl = e / 9, q = e - 9 * bl, b = b0 + bl;
        const int jj = jjl[b], j = jj & 0xff, jp = jj >> 8;
        const double* S = S16 + bl * 16;
        const double oj[3] = {oc[3 * j], oc[3 * j + 1], oc[3 * j + 2]}, op[3] = {oc[3 * jp], oc[3 * jp + 1], oc[3 * jp + 2]};
        double LL[9];
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
            for (int c = 0; c < 3; ++c) LL[3 * rr + c] = S[3 * rr + c] - S[9 + rr] * op[c] - oj[rr] * S[12 + c] + S[15] * oj[rr] * op[c];
        const double trL = (LL[0] + LL[4]) + LL[8];
        const int pj = s_parent[j], pp = s_parent[jp];
        const int rr = q / 3, c = q - 3 * rr;
        double Ar[3], Bc[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            Ar[i] = pj < 0 ? (i == rr ? 1.0 : 0.0) : Rw[9 * pj + 3 * i + rr];
            Bc[i] = pp < 0 ? (i == c ? 1.0 : 0.0) : Rw[9 * pp + 3 * i + c];
        }
        const double atb = Ar[0] * Bc[0] + Ar[1] * Bc[1] + Ar[2] * Bc[2];
        double atl = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double ltb = LL[i] * Bc[0] + LL[3 + i] * Bc[1] + LL[6 + i] * Bc[2];
            atl = fma(Ar[i], ltb, atl);
        }
        const double v = 4.0 * (trL * atb - atl);
        if (j != jp || rr <= c) {
            Hout[(size_t)(3 + 3 * j + rr) * HS + 3 + 3 * jp + c] = v;
            Hout[(size_t)(3 + 3 * jp + c) * HS + 3 + 3 * j + rr] = v;
        }
    }
}

template <int KC>
__global__ __launch_bounds__(MOM_PARTS_NTH) void k_assemble_parts(DeviceModel dm, FrameBuffers fb) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int role, fy;
    xcd_frame_block(fb, role, fy);
    const int f = fy + fb.f0;
    if (role == 0) asm_role_core<KC>(dm, fb, f, smem);
    else if (role <= MOM_ASM_NA) asm_role_shape<KC>(dm, fb, f, role - 1, smem);
    else asm_role_rotrot(dm, fb, f, role - 1 - MOM_ASM_NA, smem);
}

static size_t assemble_parts_lds_bytes(const AvtDims& d) {
    const int J = d.J, K = d.K, NP = d.mom_np;
    const size_t lists_a = sizeof(unsigned short) * (size_t)d.mom_toff[4], lists_r = sizeof(unsigned short) * (size_t)((d.mom_toff[7] - d.mom_toff[4] + 3) & ~3);
    const size_t core = sizeof(double) * ((size_t)mom_skel_doubles(d) + 2 * (J + 1) * 16 + J * 9 + J * 4 + J * K + (K * K + K) + K + (K + 8) + K * ((J + 3) / 4)) +
                        sizeof(int) * AVT_MAX_JOINTS + lists_a;
    const int Kh = (K + MOM_ASM_NA - 1) / MOM_ASM_NA;
    const size_t shape = sizeof(double) * ((size_t)12 * J + 1 + (2 * J + 1) * Kh * 6) + sizeof(int) * AVT_MAX_JOINTS + lists_a;
    const size_t rotrot = sizeof(double) * ((size_t)12 * J + 1 + d.mom_rr_doubles) + sizeof(int) * AVT_MAX_JOINTS + lists_r;
    return std::max(core, std::max(shape, rotrot)) + 64;
}

static size_t assemble_lds_bytes(const AvtDims& d) {
    return sizeof(double) * ((size_t)mom_skel_doubles(d) + mom_asm_doubles(d)) + sizeof(int) * AVT_MAX_JOINTS + sizeof(unsigned short) * (size_t)mom_tab_words(d) + 64;
}

void launch_assemble(avt_ctx* c, int nframes) {
    const AvtDims& d = c->dm.d;
    {
        // the pose prior: workgroups in the pair pass's grid up to 256 frames per launch (one launch and its boundary less on the chain: 64 frames
        // per GPU 1.25 -> 1.17 ms in round 4; 256 frames per launch 3.965 -> 3.903 ms per 512-frame step in round 5, after the pair pass's LDS went
        // from 52 to 26 KB - with 52 KB its 2 k small workgroups took the slots of the pair workgroups: 4.42 against 4.28 ms), a launch of its own above
#ifndef MOM_PRIOR_RIDE_MAX
#define MOM_PRIOR_RIDE_MAX 256
#endif
        const bool prior_rides = 16 * MOM_PP_PAIRS >= 64 && nframes <= MOM_PRIOR_RIDE_MAX;
        if (d.ncomps > 0 && !prior_rides) {
            if (c->tun.literal_dims && d.K == 10 && d.J == 24 && d.ndims == 69) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_prior<true>), dim3(d.ncomps, nframes), dim3(128), 0, c->cur_stream, c->dm, c->fb);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_prior<false>), dim3(d.ncomps, nframes), dim3(128), 0, c->cur_stream, c->dm, c->fb);
        }
        const dim3 grid(mom_nwg(d) + (prior_rides ? d.ncomps : 0), nframes);
        const size_t lds = pairpass_lds_bytes(d);
        if (c->tun.literal_dims && d.K == 10 && d.J == 24 && d.ncomps > 0 && d.ndims == 69) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pairpass<10, true>), grid, dim3(16 * MOM_PP_PAIRS), lds, c->cur_stream, c->dm, c->fb);
        else if (d.K == 10) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pairpass<10>), grid, dim3(16 * MOM_PP_PAIRS), lds, c->cur_stream, c->dm, c->fb);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pairpass<0>), grid, dim3(16 * MOM_PP_PAIRS), lds, c->cur_stream, c->dm, c->fb);
    }
    if (c->tun.asm_parts || assemble_lds_bytes(d) > avt_moments_lds_cap()) {      // six role workgroups of 256 threads per frame (the default)
        const dim3 grid(MOM_ASM_ROLES, nframes);
        const size_t lds = assemble_parts_lds_bytes(d);
        if (d.K == 10) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_assemble_parts<10>), grid, dim3(MOM_PARTS_NTH), lds, c->cur_stream, c->dm, c->fb);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_assemble_parts<0>), grid, dim3(MOM_PARTS_NTH), lds, c->cur_stream, c->dm, c->fb);
        return;
    }
    const dim3 grid(1, nframes);
    const size_t lds = assemble_lds_bytes(d);
    if (d.K == 10) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_assemble<10, MOM_ASM_NTH>), grid, dim3(MOM_ASM_NTH), lds, c->cur_stream, c->dm, c->fb);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_assemble<0, MOM_ASM_NTH>), grid, dim3(MOM_ASM_NTH), lds, c->cur_stream, c->dm, c->fb);
}

// Matrix instructions (v_mfma_f64_16x16x4_f64, 2048 flop each) one k_moments pass over a frame executes, from the kernel's own trip counts:
// per pair and 512-entry segment of its static list, ceil(matched / (4 MOM_UN)) MOM_UN rounds of one instruction per upper tile pair of psi
// (+ one per psi tile for D_k of a diagonal pair).  `cnt` = the frame's per-vertex correspondence counts (host copy).  bench.py prices
// the kernel's matrix-pipe utilisation on THIS number (profiles/: SQ_INSTS_VALU_MFMA_MOPS_F64 of the same launch agrees).
long long avt_moments_mfma_count(const avt_model* m, const int* cnt) {
    const AvtDims& d = m->d;
    const int NTP = d.mom_ntp, NTPAIR = NTP * (NTP + 1) / 2;
    long long n = 0;
    for (int p = 0; p < d.mom_np; ++p) {
        const bool diag = m->mom_pair[2 * p] == m->mom_pair[2 * p + 1];
        const int lo = m->mom_lstart[p], len = m->mom_lstart[p + 1] - lo;
        for (int base = 0; base < len; base += MOM_SEG) {
            int mseg = 0;
            for (int e = base; e < std::min(len, base + MOM_SEG); ++e) mseg += cnt[m->mom_lv[lo + e]] > 0;
            const long long rounds = (long long)((mseg + 4 * MOM_UN - 1) / (4 * MOM_UN)) * MOM_UN;
            n += rounds * (NTPAIR + (diag ? NTP : 0));
        }
    }
    return n;
}

size_t avt_moments_frame_scratch(const AvtDims& d) { return mom_frame_scratch(d); }
// both kernels of the GN loop must fit the dynamic-LDS cap set below: the assembly's request grows with the pair count and the rot-rot lists
// (SMPL: 80 KB), so a model with denser skinning can exceed it although K and P qualify - the context then keeps the row form
size_t avt_moments_lds_need(const AvtDims& d) { return std::max(assemble_parts_lds_bytes(d), pairpass_lds_bytes(d)); }      // (the one-workgroup assembly is only selectable where it fits)
size_t avt_moments_lds_cap() { return 160 * 1024 - 512; }
size_t avt_moments_T_doubles(const AvtDims& d) { return (size_t)d.mom_np * mom_tstride(d.mom_npsi); }

int avt_moments_set_attributes() {
    const int cap = (int)avt_moments_lds_cap();
    return hipFuncSetAttribute((const void*)k_assemble<10, MOM_ASM_NTH>, hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess ||
           hipFuncSetAttribute((const void*)k_assemble<0, MOM_ASM_NTH>, hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess ||
           hipFuncSetAttribute((const void*)k_assemble_parts<10>, hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess ||
           hipFuncSetAttribute((const void*)k_assemble_parts<0>, hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess ||
           hipFuncSetAttribute((const void*)k_pairpass<10>, hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess ||
           hipFuncSetAttribute((const void*)k_pairpass<0>, hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess ||
           hipFuncSetAttribute((const void*)k_pairpass<10, true>, hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess;
}
