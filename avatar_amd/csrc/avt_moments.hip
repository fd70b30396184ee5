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
// on the fp64 matrix cores), and a Gauss-Newton iteration contracts them with the state (mom_assemble: ~0.45 M multiply-adds
// against ~20 M for rebuilding and contracting the Jacobian rows; tools/moment_proto2.py is the executable specification and
// checks it against the oracle's literal per-block formulas: H 4e-15, g 4e-13, cost 1e-12 relative).
//
//   k_moments   grid (np + 1 + cost-constant blocks, frames): workgroup p < np accumulates T_p (and D_k for a diagonal pair);
//   k_assemble  grid (1 + GMM components, frames): workgroup 0 builds the dense system of the trial point into Hraw (the layout
//               k_reduce used to produce: full symmetric, row / column P = J^T r, [P][P] = sum c |r|^2), the others the pose prior.
#include <algorithm>

#include "avt_device.h"
#include "avt_prior.h"

typedef double v4f64 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double mld(const double* p) {      // scratch written earlier in the same kernel by another wave: past the L1
    return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void mst(double* p, double v) {
    __hip_atomic_store((unsigned long long*)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// =================================================================================================
// k_moments<NTP>.  One 256-thread workgroup per (frame, unordered pair).  The pair's static vertex list is compacted to the matched
// vertices (order kept), four of them per matrix instruction: lane (r = l & 15, v = l >> 4) holds psi_v[16 t + r] for the NTP
// 16-row tiles of psi - straight from the psi table, no LDS staging -; A = weight x psi, B = psi, one accumulator tile per upper
// tile pair.  The four waves split the list's rounds and their tiles are added in wave order (deterministic).
// =================================================================================================
#define MOM_SEG 1024          // static list entries compacted per pass (4 per thread)

template <int NTP>
__global__ __launch_bounds__(256) void k_moments(DeviceModel dm, FrameBuffers fb) {
    const AvtDims& d = dm.d;
    const int f = blockIdx.y + fb.f0, t = threadIdx.x, V = d.V, NP = d.mom_np, NPSI = d.mom_npsi;
    const int bx = blockIdx.x;
    if (bx > NP) { cost_const_block(dm, fb, f, bx - NP - 1); return; }
    __shared__ int s_v[MOM_SEG];
    __shared__ double s_w[MOM_SEG], s_a[MOM_SEG];
    __shared__ int s_wcnt[4];
    __shared__ double s_red[4];
    const int* cnt = fb.cnt + (size_t)f * V;
    const long long* fs = fb.fsum + (size_t)f * 3 * V;
    if (bx == NP) {      // sum_m |fsum_m|^2 / c_m over the matched vertices, fixed order
        const int M = fb.ctl[f].M;
        double a = 0.0;
        for (int e = t; e < M; e += 256) {
            const int m = fb.matched[(size_t)f * V + e];
            const double x = (double)fs[m] / AVT_FIX_SCALE, y = (double)fs[(size_t)V + m] / AVT_FIX_SCALE, z = (double)fs[2 * (size_t)V + m] / AVT_FIX_SCALE;
            a += (x * x + y * y + z * z) / (double)cnt[m];
        }
        a = wave_sum(a);
        if ((t & 63) == 0) s_red[t >> 6] = a;
        __syncthreads();
        if (t == 0) fb.mom_E[(size_t)f * 2] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
        return;
    }
    const int p = bx, k = dm.mom_pair[2 * p], k2 = dm.mom_pair[2 * p + 1];
    const bool diag = k == k2;
    const int lo = dm.mom_lstart[p], n = dm.mom_lstart[p + 1] - lo;
    constexpr int NTPAIR = NTP * (NTP + 1) / 2;
    constexpr int PW = 16 * NTP;
    v4f64 acc[NTPAIR], accD[NTP];
    const v4f64 z4 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < NTPAIR; ++i) acc[i] = z4;
#pragma unroll
    for (int i = 0; i < NTP; ++i) accD[i] = z4;
    const int wv = t >> 6, ln = t & 63, r16 = ln & 15, kk = ln >> 4;
    for (int base = 0; base < n; base += MOM_SEG) {
        // ---- compaction of entries base + 4 t .. base + 4 t + 3 (order kept)
        int vv[4], cc[4], mine = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = base + 4 * t + u;
            vv[u] = e < n ? dm.mom_lv[lo + e] : 0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = base + 4 * t + u;
            cc[u] = e < n ? cnt[vv[u]] : 0;
            mine += cc[u] > 0;
        }
        const int incl = wave_incl_scan(mine);
        if (ln == 63) s_wcnt[wv] = incl;
        __syncthreads();
        int pos = incl - mine, mseg = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { if (w < wv) pos += s_wcnt[w]; mseg += s_wcnt[w]; }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (cc[u] > 0) {
                const int e = base + 4 * t + u;
                const double wa = dm.mom_lw[2 * (size_t)(lo + e)], wb = dm.mom_lw[2 * (size_t)(lo + e) + 1];
                s_v[pos] = vv[u]; s_w[pos] = (double)cc[u] * wa * wb; s_a[pos] = wa;
                ++pos;
            }
        __syncthreads();
        // ---- rounds of four matched vertices: wave w takes rounds w, w + 4, ..
        const int nr = (mseg + 3) >> 2;
#pragma unroll 2
        for (int r = wv; r < nr; r += 4) {
            const int idx = 4 * r + kk;
            const bool on = idx < mseg;
            const int v = on ? s_v[idx] : 0;
            const double wgt = on ? s_w[idx] : 0.0;
            const double* ps = dm.mom_psi + (size_t)v * PW + r16;
            double fr[NTP], fw[NTP];
#pragma unroll
            for (int q = 0; q < NTP; ++q) fr[q] = ps[16 * q];
#pragma unroll
            for (int q = 0; q < NTP; ++q) { fr[q] = on ? fr[q] : 0.0; fw[q] = fr[q] * wgt; }
            int pi = 0;
#pragma unroll
            for (int ti = 0; ti < NTP; ++ti)
#pragma unroll
                for (int tj = ti; tj < NTP; ++tj) { acc[pi] = __builtin_amdgcn_mfma_f64_16x16x4f64(fw[ti], fr[tj], acc[pi], 0, 0, 0); ++pi; }
            if (diag) {      // D_k: B = a_mk (sum_i d_i - c centre) in columns 0..2
                double bd = 0.0;
                if (on && r16 < 3) bd = s_a[idx] * ((double)fs[(size_t)r16 * V + v] / AVT_FIX_SCALE);
#pragma unroll
                for (int q = 0; q < NTP; ++q) accD[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(fr[q], bd, accD[q], 0, 0, 0);
            }
        }
        __syncthreads();      // the list is rewritten by the next pass
    }
    // ---- the four waves' tiles, added in wave order through one buffer
    __shared__ double s_buf[(NTPAIR + NTP) * 256];
    for (int w = 1; w < 4; ++w) {
        if (wv == w) {
#pragma unroll
            for (int i = 0; i < NTPAIR; ++i)
#pragma unroll
                for (int v = 0; v < 4; ++v) s_buf[(i * 4 + v) * 64 + ln] = acc[i][v];
            if (diag) {
#pragma unroll
                for (int i = 0; i < NTP; ++i)
#pragma unroll
                    for (int v = 0; v < 4; ++v) s_buf[((NTPAIR + i) * 4 + v) * 64 + ln] = accD[i][v];
            }
        }
        __syncthreads();
        if (wv == 0) {
#pragma unroll
            for (int i = 0; i < NTPAIR; ++i)
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[i][v] += s_buf[(i * 4 + v) * 64 + ln];
            if (diag) {
#pragma unroll
                for (int i = 0; i < NTP; ++i)
#pragma unroll
                    for (int v = 0; v < 4; ++v) accD[i][v] += s_buf[((NTPAIR + i) * 4 + v) * 64 + ln];
            }
        }
        __syncthreads();
    }
    if (wv != 0) return;
    // accumulator element v of lane (c16 = l & 15, g4 = l >> 4): row 4 v + g4, column c16 of the tile
    double* T = fb.mom_T + ((size_t)f * NP + p) * NPSI * NPSI;
    int pi = 0;
#pragma unroll
    for (int ti = 0; ti < NTP; ++ti)
#pragma unroll
        for (int tj = ti; tj < NTP; ++tj) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int a = 16 * ti + 4 * v + kk, b = 16 * tj + r16;
                if (a < NPSI && b < NPSI) {
                    T[(size_t)a * NPSI + b] = acc[pi][v];
                    if (ti != tj) T[(size_t)b * NPSI + a] = acc[pi][v];
                }
            }
            ++pi;
        }
    if (diag) {
        double* D = fb.mom_D + ((size_t)f * d.J + k) * NPSI * 3;
#pragma unroll
        for (int q = 0; q < NTP; ++q)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int a = 16 * q + 4 * v + kk;
                if (a < NPSI && r16 < 3) D[(size_t)a * 3 + r16] = accD[q][v];
            }
    }
}

void launch_moments(avt_ctx* c, int nframes) {
    const AvtDims& d = c->dm.d;
    const dim3 grid(d.mom_np + 1 + c->fb.const_used, nframes);
    switch (d.mom_ntp) {
        case 1: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_moments<1>), grid, dim3(256), 0, c->cur_stream, c->dm, c->fb); break;
        case 2: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_moments<2>), grid, dim3(256), 0, c->cur_stream, c->dm, c->fb); break;
        case 3: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_moments<3>), grid, dim3(256), 0, c->cur_stream, c->dm, c->fb); break;
        default: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_moments<4>), grid, dim3(256), 0, c->cur_stream, c->dm, c->fb); break;
    }
}

// =================================================================================================
// mom_assemble<NTH>: the dense system of one state from the moments.  Called by every thread of a workgroup (barriers inside).
//   sk : the skeleton tables of the state in LDS (MomSkel below);
//   scr: LDS scratch, mom_scratch_doubles(d) doubles;
//   Hout: HS x HS block, full symmetric; row / column P = J^T r; [P][P] = sum_m c_m |x_m - dbar_m|^2.
// Phases (tools/moment_proto2.py::assemble, same names):
//   A   one 16-lane group per unordered pair, lane = s' (0 .. K): 9 (K + 1) + 4 loads of T, Q = omega-contraction, zz = G-contraction,
//       group butterfly for P2 / p1; lane 0 writes X16 of both orders, lanes 1 .. K the per-(ordered pair, shape key) records;
//       the shape-shape columns and sum tr(Y) stay in registers across the group's pairs;
//   A'  per joint: XD_k, Dl_k, and YF;
//   B0  per-joint partner sums, B1 subtree sums, B2 every block but rot-rot; rot-rot in three stages (M1, S2, blocks).
// =================================================================================================
struct MomSkel {
    const double* Rw;     // [J][9] world rotations, row-major
    const double* oc;     // [J][3] world joint origins minus the frame centre
    const double* tau;    // [J][3] o_k - R_k J_k(omega) - centre
    const double* eta;    // [J][3][K] H_k - R_k S_k
    const double* om;     // [K + 1] omega = [1; w]
    const int* parent;    // [J]
};

__host__ __device__ inline int mom_scratch_doubles(const AvtDims& d, int nth) {
    const int J = d.J, K = d.K, NG = nth / 16;
    return 2 * d.mom_np * 16 + d.mom_nm1 * 16 + 2 * J * 16 + 2 * J * K * 6 + J * 9 + NG * (K * K + K) + 2 * K + 8;
}

template <int NTH, int KC /* K if known at compile time, else 0 */>
__device__ __forceinline__ void mom_assemble(const DeviceModel& dm, const FrameBuffers& fb, int f, const MomSkel& sk, double* __restrict__ scr,
                                             double* __restrict__ Hout) {
    const AvtDims& d = dm.d;
    const int J = d.J, K = KC ? KC : d.K, S1 = K + 1, NP = d.mom_np, NPSI = d.mom_npsi, P = d.P, HS = d.HS, t = threadIdx.x;
    constexpr int NG = NTH / 16;
    double* X16 = scr;                          // [2 NP][16]  W (9, row-major), Va, Vb, t0
    double* M1 = X16 + 2 * NP * 16;             // [nm1][16]
    double* PK = M1 + d.mom_nm1 * 16;           // [J][16]  axial(sum W), sum Va, sum Vb, sum t0, axial(XD), Dl
    double* TK = PK + J * 16;                   // [J][16]  subtree sums of PK
    double* PR = TK + J * 16;                   // [J][K][6]
    double* TR = PR + J * K * 6;                // [J][K][6]
    double* XDt = TR + J * K * 6;               // [J][9]  (only the trace is needed beyond the axial vector: stored whole, small)
    double* ZR = XDt + J * 9;                   // [NG][K*K + K] the groups' shape-shape columns and sum tr(Y)
    double* YFs = ZR + NG * (K * K + K);        // [K]
    double* YXs = YFs + K;                      // [K]
    double* misc = YXs + K;                     // [8]
    double* REC = fb.mom_rec + (size_t)f * 2 * NP * K * 6;      // global scratch: [2 NP][K][6]
    const double* Tf = fb.mom_T + (size_t)f * NP * NPSI * NPSI;
    const double* Df = fb.mom_D + (size_t)f * J * NPSI * 3;

    // ---------------- phase A
    {
        const int gid = t >> 4, sl = t & 15;
        const bool lane_on = sl < S1;
        const double om_l = lane_on ? sk.om[sl] : 0.0;
        double zc[KC ? KC : AVT_MAX_SHAPE], yx = 0.0;
#pragma unroll
        for (int s = 0; s < (KC ? KC : AVT_MAX_SHAPE); ++s) zc[s] = 0.0;
        for (int p = gid; p < NP; p += NG) {
            const int k = dm.mom_pair[2 * p], k2 = dm.mom_pair[2 * p + 1];
            const double* Tp = Tf + (size_t)p * NPSI * NPSI;
            const double nu = k == k2 ? 0.5 : 1.0;
            double Ra[9], Rb[9], G[9], ta[3], tb[3];
#pragma unroll
            for (int e = 0; e < 9; ++e) { Ra[e] = sk.Rw[9 * k + e]; Rb[e] = sk.Rw[9 * k2 + e]; }
#pragma unroll
            for (int e = 0; e < 3; ++e) { ta[e] = sk.tau[3 * k + e]; tb[e] = sk.tau[3 * k2 + e]; }
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int i2 = 0; i2 < 3; ++i2) G[3 * i + i2] = Ra[i] * Rb[i2] + Ra[3 + i] * Rb[3 + i2] + Ra[6 + i] * Rb[6 + i2];     // R_k^T R_k'
            const int col = lane_on ? sl : 0;
            double Q[9];
#pragma unroll
            for (int e = 0; e < 9; ++e) Q[e] = 0.0;
            // s = 0 (base): only Q
            {
                double v[9];
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int i2 = 0; i2 < 3; ++i2) v[3 * i + i2] = Tp[(size_t)(S1 * i) * NPSI + S1 * i2 + col];
#pragma unroll
                for (int e = 0; e < 9; ++e) Q[e] += v[e];        // omega_0 = 1
            }
#pragma unroll
            for (int s = 1; s < (KC ? KC + 1 : 1); ++s) {
                double v[9];
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int i2 = 0; i2 < 3; ++i2) v[3 * i + i2] = Tp[(size_t)(S1 * i + s) * NPSI + S1 * i2 + col];
                const double oms = sk.om[s];
                double zz = 0.0;
#pragma unroll
                for (int e = 0; e < 9; ++e) { Q[e] = fma(oms, v[e], Q[e]); zz = fma(G[e], v[e], zz); }
                zc[s - 1] = fma(nu, zz, zc[s - 1]);
            }
            if (!KC) {
                for (int s = 1; s < S1; ++s) {
                    double v[9];
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int i2 = 0; i2 < 3; ++i2) v[3 * i + i2] = Tp[(size_t)(S1 * i + s) * NPSI + S1 * i2 + col];
                    const double oms = sk.om[s];
                    double zz = 0.0;
#pragma unroll
                    for (int e = 0; e < 9; ++e) { Q[e] = fma(oms, v[e], Q[e]); zz = fma(G[e], v[e], zz); }
#pragma unroll
                    for (int q = 0; q < AVT_MAX_SHAPE; ++q) if (q == s - 1) zc[q] = fma(nu, zz, zc[q]);
                }
            }
            double tph[3];
#pragma unroll
            for (int i2 = 0; i2 < 3; ++i2) tph[i2] = Tp[(size_t)(NPSI - 1) * NPSI + S1 * i2 + col];
            const double t0 = Tp[(size_t)NPSI * NPSI - 1];
            // group sums over the lanes (fixed butterfly: every lane of the group ends with the same bits)
            double P2[9], p1[3];
#pragma unroll
            for (int e = 0; e < 9; ++e) P2[e] = om_l * Q[e];
#pragma unroll
            for (int e = 0; e < 3; ++e) p1[e] = om_l * tph[e];
#pragma unroll
            for (int sft = 8; sft >= 1; sft >>= 1) {
#pragma unroll
                for (int e = 0; e < 9; ++e) P2[e] += __shfl_xor(P2[e], sft, 64);
#pragma unroll
                for (int e = 0; e < 3; ++e) p1[e] += __shfl_xor(p1[e], sft, 64);
            }
            double Rap1[3], Rbp1[3], Va[3], Vb[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                Rap1[r] = Ra[3 * r] * p1[0] + Ra[3 * r + 1] * p1[1] + Ra[3 * r + 2] * p1[2];
                Rbp1[r] = Rb[3 * r] * p1[0] + Rb[3 * r + 1] * p1[1] + Rb[3 * r + 2] * p1[2];
                Va[r] = fma(t0, ta[r], Rap1[r]);
                Vb[r] = fma(t0, tb[r], Rbp1[r]);
            }
            if (sl == 0) {
                // W_kk' = R_k P2 R_k'^T + Va tau_k'^T + tau_k (R_k' p1)^T;  W_k'k = W_kk'^T
                double RaP[9], W[9];
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) RaP[3 * r + c] = Ra[3 * r] * P2[c] + Ra[3 * r + 1] * P2[3 + c] + Ra[3 * r + 2] * P2[6 + c];
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        W[3 * r + c] = (RaP[3 * r] * Rb[3 * c] + RaP[3 * r + 1] * Rb[3 * c + 1] + RaP[3 * r + 2] * Rb[3 * c + 2]) + Va[r] * tb[c] + ta[r] * Rbp1[c];
                double* x0 = X16 + (size_t)(2 * p) * 16;
#pragma unroll
                for (int e = 0; e < 9; ++e) x0[e] = W[e];
#pragma unroll
                for (int e = 0; e < 3; ++e) { x0[9 + e] = Va[e]; x0[12 + e] = Vb[e]; }
                x0[15] = t0;
                double* x1 = x0 + 16;
                if (k != k2) {
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int c = 0; c < 3; ++c) x1[3 * r + c] = W[3 * c + r];
#pragma unroll
                    for (int e = 0; e < 3; ++e) { x1[9 + e] = Vb[e]; x1[12 + e] = Va[e]; }
                    x1[15] = t0;
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) x1[e] = 0.0;
                }
            } else if (lane_on) {
                const int s = sl - 1;
                const double* ea = sk.eta + (size_t)k * 3 * K;       // eta_k[r][s]
                const double* eb = sk.eta + (size_t)k2 * 3 * K;
                double ya[3], yb[3];                                 // R_k tphi, R_k' tphi
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    ya[r] = Ra[3 * r] * tph[0] + Ra[3 * r + 1] * tph[1] + Ra[3 * r + 2] * tph[2];
                    yb[r] = Rb[3 * r] * tph[0] + Rb[3 * r + 1] * tph[1] + Rb[3 * r + 2] * tph[2];
                }
                const double eas[3] = {ea[s], ea[K + s], ea[2 * K + s]}, ebs[3] = {eb[s], eb[K + s], eb[2 * K + s]};
                auto record = [&](const double (&RA)[9], const double (&RB)[9], const double (&VA)[3], const double (&TA)[3], const double (&ylin)[3],
                                  const double (&eB)[3], int op) {
                    // Y = R_a Qs R_b^T + V_a eta_b,s^T + tau_a ylin^T,  U = ylin + t0 eta_b,s
                    double RQ[9], Y[9];
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int c = 0; c < 3; ++c) RQ[3 * r + c] = RA[3 * r] * Q[c] + RA[3 * r + 1] * Q[3 + c] + RA[3 * r + 2] * Q[6 + c];
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int c = 0; c < 3; ++c)
                            Y[3 * r + c] = (RQ[3 * r] * RB[3 * c] + RQ[3 * r + 1] * RB[3 * c + 1] + RQ[3 * r + 2] * RB[3 * c + 2]) + VA[r] * eB[c] + TA[r] * ylin[c];
                    double* rc = REC + ((size_t)op * K + s) * 6;
                    mst(rc + 0, Y[5] - Y[7]); mst(rc + 1, Y[6] - Y[2]); mst(rc + 2, Y[1] - Y[3]);
                    mst(rc + 3, fma(t0, eB[0], ylin[0])); mst(rc + 4, fma(t0, eB[1], ylin[1])); mst(rc + 5, fma(t0, eB[2], ylin[2]));
                    yx += (Y[0] + Y[4]) + Y[8];
                };
                record(Ra, Rb, Va, ta, yb, ebs, 2 * p);
                if (k != k2) record(Rb, Ra, Vb, tb, ya, eas, 2 * p + 1);
                // column t = s of Z~': nu (zz + eta_k,s2 . yb + eta_k',s2 . ya + t0 eta_k,s2 . eta_k',t)   (zz is already in zc)
                const double ebt[3] = {ebs[0], ebs[1], ebs[2]};
#pragma unroll
                for (int s2 = 0; s2 < (KC ? KC : AVT_MAX_SHAPE); ++s2) {
                    if (s2 < K) {
                        const double e0 = ea[s2], e1 = ea[K + s2], e2 = ea[2 * K + s2];
                        const double f0 = eb[s2], f1 = eb[K + s2], f2 = eb[2 * K + s2];
                        const double add = (e0 * yb[0] + e1 * yb[1] + e2 * yb[2]) + (f0 * ya[0] + f1 * ya[1] + f2 * ya[2]) + t0 * (e0 * ebt[0] + e1 * ebt[1] + e2 * ebt[2]);
                        zc[s2] = fma(nu, add, zc[s2]);
                    }
                }
            }
        }
        if (sl >= 1 && lane_on) {
            double* zr = ZR + (size_t)gid * (K * K + K);
#pragma unroll
            for (int s2 = 0; s2 < (KC ? KC : AVT_MAX_SHAPE); ++s2) if (s2 < K) zr[s2 * K + (sl - 1)] = zc[s2];
            zr[K * K + (sl - 1)] = yx;
        }
    }
    // ---------------- phase A': data-side moments per joint
    for (int e = t; e < J * 3; e += NTH) {      // (k, c): column c of XD_k = R_k (sum_s om_s Dphi_k[.][s][c]) + tau_k Dl_k[c]
        const int k = e / 3, c = e - 3 * k;
        const double* Dk = Df + (size_t)k * NPSI * 3;
        double u[3] = {0.0, 0.0, 0.0};
        for (int i = 0; i < 3; ++i)
            for (int s = 0; s < S1; ++s) u[i] = fma(sk.om[s], Dk[(size_t)(S1 * i + s) * 3 + c], u[i]);
        const double dl = Dk[(size_t)(NPSI - 1) * 3 + c];
        const double* Rk = sk.Rw + 9 * k;
        for (int r = 0; r < 3; ++r) XDt[9 * k + 3 * r + c] = (Rk[3 * r] * u[0] + Rk[3 * r + 1] * u[1] + Rk[3 * r + 2] * u[2]) + sk.tau[3 * k + r] * dl;
        PK[16 * k + 13 + c] = dl;
    }
    if (t < K) {      // YF[s] = sum_k tr(R_k Dphi_k[s]) + eta_k,s . Dl_k
        const int s = t;
        double a = 0.0;
        for (int k = 0; k < J; ++k) {
            const double* Dk = Df + (size_t)k * NPSI * 3;
            const double* Rk = sk.Rw + 9 * k;
            double q = 0.0;
            for (int c = 0; c < 3; ++c)
                for (int i = 0; i < 3; ++i) q = fma(Rk[3 * c + i], Dk[(size_t)(S1 * i + s + 1) * 3 + c], q);
            for (int c = 0; c < 3; ++c) q = fma(sk.eta[((size_t)k * 3 + c) * K + s], Dk[(size_t)(NPSI - 1) * 3 + c], q);
            a += q;
        }
        YFs[s] = a;
    }
    __syncthreads();
    // ---------------- phase B0: per-joint partner sums
    for (int e = t; e < J * 10; e += NTH) {      // PK[k][0..9]: axial(sum W) 3, sum Va 3, sum Vb 3, sum t0
        const int k = e / 10, q = e - 10 * k;
        double a = 0.0;
        for (int i = dm.mom_opk_start[k]; i < dm.mom_opk_start[k + 1]; ++i) {
            const double* x = X16 + (size_t)dm.mom_opk[i] * 16;
            a += q == 0 ? x[5] - x[7] : (q == 1 ? x[6] - x[2] : (q == 2 ? x[1] - x[3] : x[9 + (q - 3)]));      // (q = 9: x[15] = t0)
        }
        PK[16 * k + q] = a;
    }
    for (int e = t; e < J * 3; e += NTH) {
        const int k = e / 3, c = e - 3 * k;
        const double* X = XDt + 9 * k;
        PK[16 * k + 10 + c] = c == 0 ? X[5] - X[7] : (c == 1 ? X[6] - X[2] : X[1] - X[3]);
    }
    for (int e = t; e < J * K * 6; e += NTH) {
        const int k = e / (K * 6), r = e - k * K * 6;
        double a = 0.0;
        for (int i = dm.mom_opk_start[k]; i < dm.mom_opk_start[k + 1]; ++i) a += mld(REC + (size_t)dm.mom_opk[i] * K * 6 + r);
        PR[e] = a;
    }
    for (int e = t; e < d.mom_nm1 * 16; e += NTH) {      // rot-rot stage 1
        const int id = e >> 4, q = e & 15;
        double a = 0.0;
        for (int i = dm.mom_m1_start[id]; i < dm.mom_m1_start[id + 1]; ++i) a += X16[(size_t)dm.mom_m1[i] * 16 + q];
        M1[e] = a;
    }
    for (int e = t; e < K * K + K; e += NTH) {           // the groups' shape-shape columns / traces, in group order
        double a = 0.0;
        for (int g = 0; g < NG; ++g) a += ZR[(size_t)g * (K * K + K) + e];
        if (e < K * K) ZR[e] = a; else YXs[e - K * K] = a;       // (group 0's slot is read by nobody else at index e before this write: one thread per e)
    }
    if (t == 0) {
        double xx = 0.0, xf = 0.0;
        for (int op = 0; op < 2 * NP; ++op) { const double* x = X16 + (size_t)op * 16; xx += (x[0] + x[4]) + x[8]; }
        for (int k = 0; k < J; ++k) xf += (XDt[9 * k] + XDt[9 * k + 4]) + XDt[9 * k + 8];
        misc[0] = xx - 2.0 * xf + fb.mom_E[(size_t)f * 2];
    }
    __syncthreads();
    // ---------------- phase B1: subtree sums
    for (int e = t; e < J * 16; e += NTH) {
        const int j = e >> 4, q = e & 15;
        double a = 0.0;
        for (int i = dm.mom_sub_start[j]; i < dm.mom_sub_start[j + 1]; ++i) a += PK[16 * dm.mom_sub[i] + q];
        TK[e] = a;
    }
    for (int e = t; e < J * K * 6; e += NTH) {
        const int j = e / (K * 6), r = e - j * K * 6;
        double a = 0.0;
        for (int i = dm.mom_sub_start[j]; i < dm.mom_sub_start[j + 1]; ++i) a += PR[(size_t)dm.mom_sub[i] * K * 6 + r];
        TR[e] = a;
    }
    __syncthreads();
    // ---------------- phase B2: every block but rot-rot
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
    if (t < K) Hset(SH + t, P, YXs[t] - YFs[t]);
    for (int e = t; e < K * K; e += NTH) { const int s = e / K, u = e - s * K; Hout[(size_t)(SH + s) * HS + SH + u] = ZR[s * K + u] + ZR[u * K + s]; }
    // ---------------- rot-rot: stage 2 + blocks, one thread per (j <= j')
    for (int b = t; b < d.mom_nb2; b += NTH) {
        const int jj = dm.mom_s2_jj[b], j = jj & 0xff, jp = jj >> 8;
        double S[16], Vt[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < 16; ++q) S[q] = 0.0;
        for (int i = dm.mom_s2_start[2 * b]; i < dm.mom_s2_start[2 * b + 1]; ++i) {
            const double* m1 = M1 + (size_t)dm.mom_s2[i] * 16;
#pragma unroll
            for (int q = 0; q < 16; ++q) S[q] += m1[q];
        }
        for (int i = dm.mom_s2_start[2 * b + 1]; i < dm.mom_s2_start[2 * b + 2]; ++i) {
            const double* m1 = M1 + (size_t)dm.mom_s2[i] * 16;
            Vt[0] += m1[9]; Vt[1] += m1[10]; Vt[2] += m1[11];
        }
        const double oj[3] = {sk.oc[3 * j], sk.oc[3 * j + 1], sk.oc[3 * j + 2]}, op[3] = {sk.oc[3 * jp], sk.oc[3 * jp + 1], sk.oc[3 * jp + 2]};
        double LL[9];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) LL[3 * r + c] = S[3 * r + c] - S[9 + r] * op[c] - oj[r] * Vt[c] + S[15] * oj[r] * op[c];
        const double trL = (LL[0] + LL[4]) + LL[8];
        double A[9], B[9];      // R_par(j), R_par(j')
        const int pj = sk.parent[j], pp = sk.parent[jp];
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            A[q] = pj < 0 ? ((q == 0 || q == 4 || q == 8) ? 1.0 : 0.0) : sk.Rw[9 * pj + q];
            B[q] = pp < 0 ? ((q == 0 || q == 4 || q == 8) ? 1.0 : 0.0) : sk.Rw[9 * pp + q];
        }
        // blk = 4 (tr(LL) A^T B - A^T LL^T B)
        double LtB[9];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) LtB[3 * r + c] = LL[r] * B[c] + LL[3 + r] * B[3 + c] + LL[6 + r] * B[6 + c];      // (LL^T B)[r][c]
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const double atb = A[r] * B[c] + A[3 + r] * B[3 + c] + A[6 + r] * B[6 + c];
                const double atl = A[r] * LtB[c] + A[3 + r] * LtB[3 + c] + A[6 + r] * LtB[6 + c];
                const double v = 4.0 * (trL * atb - atl);
                if (j != jp || r <= c) {      // (a diagonal block: the upper triangle, mirrored - symmetric to the bit)
                    Hout[(size_t)(3 + 3 * j + r) * HS + 3 + 3 * jp + c] = v;
                    Hout[(size_t)(3 + 3 * jp + c) * HS + 3 + 3 * j + r] = v;
                }
            }
    }
}

// skeleton tables of the assembly from a prep block (avt_internal.h) in global memory
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
__host__ __device__ inline int mom_skel_doubles(const AvtDims& d) { return ((15 * d.J + 3 * d.J * d.K + d.K + 1) + 1) & ~1; }

// =================================================================================================
// k_assemble.  grid (1 + GMM components, frames), block 256: workgroup 0 of a frame assembles the system of its TRIAL point
// (the slot 1 - cur_slot, like k_eval + k_reduce did), the others evaluate the pose prior there (avt_prior.h).
// =================================================================================================
template <int KC>
__global__ __launch_bounds__(256) void k_assemble(DeviceModel dm, FrameBuffers fb) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const AvtDims& d = dm.d;
    const int f = blockIdx.y + fb.f0;
    const int try_slot = 1 - fb.ctl[f].cur_slot;
    if (blockIdx.x > 0) { prior_component(dm, fb, f, blockIdx.x - 1, try_slot, (double*)smem); return; }
    double* skm = (double*)smem;
    double* scr = skm + mom_skel_doubles(d);
    int* s_parent = (int*)(scr + mom_scratch_doubles(d, 256));
    MomSkel sk;
    mom_skel_from_prep<256>(d, fb.prep + ((size_t)f * 2 + try_slot) * d.prep_size, fb.ctl[f].centre, skm, s_parent, dm.parent, sk);
    __syncthreads();
    mom_assemble<256, KC>(dm, fb, f, sk, scr, fb.Hraw + ((size_t)f * 2 + try_slot) * d.HS * d.HS);
}

static size_t assemble_lds_bytes(const AvtDims& d) {
    return sizeof(double) * ((size_t)mom_skel_doubles(d) + mom_scratch_doubles(d, 256)) + sizeof(int) * AVT_MAX_JOINTS + 64;
}

void launch_assemble(avt_ctx* c, int nframes) {
    const AvtDims& d = c->dm.d;
    const dim3 grid(1 + d.ncomps, nframes);
    const size_t lds = std::max(assemble_lds_bytes(d), sizeof(double) * 5 * AVT_MAX_JOINTS);
    if (d.K == 10) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_assemble<10>), grid, dim3(256), lds, c->cur_stream, c->dm, c->fb);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_assemble<0>), grid, dim3(256), lds, c->cur_stream, c->dm, c->fb);
}

int avt_moments_set_attributes() {
    const int cap = 160 * 1024 - 512;
    return hipFuncSetAttribute((const void*)k_assemble<10>, hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess ||
           hipFuncSetAttribute((const void*)k_assemble<0>, hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess;
}
