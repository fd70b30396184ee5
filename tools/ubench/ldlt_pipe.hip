// micro-benchmark: LDL^T of k_solve's bordered 86x86 system as a two-stage pipeline inside one workgroup, no barriers in the loop:
//   wave 0 (panel wave): one matrix row per lane (rows 64..95 as a second row of lanes 0..31).  Per round of 4 pivots it
//           reads its rows of the panel columns the matrix waves published (updated through round kb-2), applies the
//           update of round kb-1 itself (the four pivot rows' W values cross the wave by v_readlane), broadcasts the 4x4
//           diagonal block by v_readlane, factors it, forms its W rows and stores them (back-substitution layout);
//   waves 1..3 (matrix waves): the 21 lower-triangle tiles of the trailing matrix in MFMA accumulators (7 each); round
//           kb applies the rank-4 update with W(kb) (v_mfma_f64_16x16x4_f64), the block column of panel kb+2 first,
//           and publishes that panel's 4 columns.  They have a full round of slack, so the panel wave never waits.
//   Hand-offs are LDS flags (bounded polling): W rounds done / panels published per matrix wave.
// Measured on MI355X: exact (7e-16 against a host LDL^T) but 2640 clocks per round against 1900 for the shipped register-blocked
// scheme (ldlt.hip) and 1657 for the two-barrier MFMA variant (ldlt_mfma.hip): the panel wave alone issues ~350
// instructions per round (52 v_readlane, 72 FMAs, selects, stores) at one instruction per ~5-7 clocks, which is the whole
// budget.  NOT adopted.  (Two code-generation notes: flags polled through a volatile generic pointer become
// flat_load ... sc0 sc1, hundreds of clocks per poll - use LDS-typed atomics; and with __launch_bounds__(256) the
// accumulators are allocated as AGPRs and copied out around every matrix instruction - (256, 2) keeps them in VGPRs.)
// Build: hipcc -O3 --offload-arch=gfx950 -o ldlt_pipe ldlt_pipe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef double d2v __attribute__((ext_vector_type(2)));
typedef double v4f64 __attribute__((ext_vector_type(4)));
#define NBS 24
#define PB_STRIDE 4
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
// flags[0] = W rounds finished, flags[1..3] = panels published by matrix wave 1..3, flags[4] = failure / give up.
// LDS-typed pointers on purpose: a volatile generic pointer compiles to flat_load ... sc0 sc1 (hundreds of clocks per poll).
typedef __attribute__((address_space(3))) int lds_i32;
__device__ __forceinline__ int flag_load(lds_i32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void flag_store(lds_i32* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ bool wait_ge(lds_i32* f, int want, lds_i32* fail) {
    int spins = 0;
    while (flag_load(f) < want) {
        if (flag_load(fail) || ++spins > (1 << 15)) { flag_store(fail, 1); return false; }
        __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return true;
}

typedef __attribute__((address_space(3))) int4 lds_i32x4;
__device__ __forceinline__ bool panel_wave(const double* __restrict__ s_PB, double* __restrict__ Lblk, double* __restrict__ s_R, lds_i32* flags, int P, int ln) {
    const int NR = (P + 3) >> 2;
    double wp[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, lp[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};   // W / L rows of the previous round (row ln, row 64+ln)
    d2v x01, x23, y01, y23;                  // my rows of the panel about to be factored
    bool have = false;                       // ... already fetched during the previous round
    for (int kb = 0; kb < NR; ++kb) {
        if (!have) {
            if (!wait_ge(flags + 1, kb + 1, flags + 4) || !wait_ge(flags + 2, kb + 1, flags + 4) || !wait_ge(flags + 3, kb + 1, flags + 4)) return false;
            const d2v* pb = (const d2v*)(s_PB + (size_t)(kb & 1) * 96 * PB_STRIDE);
            x01 = pb[ln * 2]; x23 = pb[ln * 2 + 1]; y01 = pb[(64 + (ln & 31)) * 2]; y23 = pb[(64 + (ln & 31)) * 2 + 1];
        }
        // how far the matrix waves have published: asked now, looked at after the update below (next panel's prefetch)
        const int f1 = flag_load(flags + 1), f2 = flag_load(flags + 2), f3 = flag_load(flags + 3);
        double a[2][4] = {{x01.x, x01.y, x23.x, x23.y}, {y01.x, y01.y, y23.x, y23.y}};
        const int c0 = 4 * kb, cl = c0 & 63;
        const bool hi = c0 >= 64;                 // the four pivot rows live in the second row set (uniform)
        if (kb >= 1) {      // update of round kb-1: a(i, c) -= sum_p l_i[p] w_c[p]
            double ws[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) ws[p] = hi ? wp[1][p] : wp[0][p];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const double wc = readlane_f64(ws[p], cl + cc);
                    a[0][cc] = fma(-lp[0][p], wc, a[0][cc]);
                    a[1][cc] = fma(-lp[1][p], wc, a[1][cc]);
                }
        }
        // the diagonal block (lower triangle) from lanes cl..cl+3
        double as[4], Dm[4][4];
#pragma unroll
        for (int dd = 0; dd < 4; ++dd) as[dd] = hi ? a[1][dd] : a[0][dd];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
#pragma unroll
            for (int dd = 0; dd <= cc; ++dd) Dm[cc][dd] = readlane_f64(as[dd], cl + cc);
        // next panel already published?  then its rows travel while the pivot chain runs
        have = false;
        if (kb + 1 < NR && min(f1, min(f2, f3)) >= kb + 2) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const d2v* pb = (const d2v*)(s_PB + (size_t)((kb + 1) & 1) * 96 * PB_STRIDE);
            x01 = pb[ln * 2]; x23 = pb[ln * 2 + 1]; y01 = pb[(64 + (ln & 31)) * 2]; y23 = pb[(64 + (ln & 31)) * 2 + 1];
            have = true;
        }
        const double D00 = Dm[0][0], D10 = Dm[1][0];
        double D11 = Dm[1][1], D20 = Dm[2][0], D21 = Dm[2][1], D22 = Dm[2][2], D30 = Dm[3][0], D31 = Dm[3][1], D32 = Dm[3][2], D33 = Dm[3][3];
        const double r0 = fast_rcp(D00), l10 = D10 * r0, l20 = D20 * r0, l30 = D30 * r0;
        D11 = fma(-l10, D10, D11); D21 = fma(-l20, D10, D21); D31 = fma(-l30, D10, D31);
        D22 = fma(-l20, D20, D22); D32 = fma(-l30, D20, D32); D33 = fma(-l30, D30, D33);
        const double r1 = fast_rcp(D11), l21 = D21 * r1, l31 = D31 * r1;
        D22 = fma(-l21, D21, D22); D32 = fma(-l31, D21, D32); D33 = fma(-l31, D31, D33);
        const double r2 = fast_rcp(D22), l32 = D32 * r2;
        D33 = fma(-l32, D32, D33);
        const double r3 = fast_rcp(D33);
        const bool real1 = c0 + 1 < P, real2 = c0 + 2 < P, real3 = c0 + 3 < P;
        const bool bad = !(D00 > 0.0) | (real1 & !(D11 > 0.0)) | (real2 & !(D22 > 0.0)) | (real3 & !(D33 > 0.0));
        if (bad) { flag_store(flags + 4, 1); return false; }
        const double rr[4] = {r0, real1 ? r1 : 0.0, real2 ? r2 : 0.0, real3 ? r3 : 0.0};
        // W rows of my two matrix rows, stored for the matrix waves and the back substitution: rows above the panel keep
        // their zeros, the diagonal block keeps its strictly lower part
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            double w[4];
            w[0] = a[s2][0]; w[1] = fma(-w[0], l10, a[s2][1]); w[2] = fma(-w[1], l21, fma(-w[0], l20, a[s2][2]));
            w[3] = fma(-w[2], l32, fma(-w[1], l31, fma(-w[0], l30, a[s2][3])));
            const int row = 64 * s2 + ln, q = row - c0;
            if (q >= 0 && (s2 == 0 || ln < 32)) {
                d2v* Wo = (d2v*)(Lblk + ((size_t)kb * NBS + (row >> 2)) * 18 + (row & 3) * 4);
                Wo[0] = (d2v){q < 1 ? 0.0 : w[0], q < 2 ? 0.0 : w[1]};
                Wo[1] = (d2v){q < 3 ? 0.0 : w[2], q < 4 ? 0.0 : w[3]};
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) { wp[s2][p] = w[p]; lp[s2][p] = w[p] * rr[p]; }
        }
        if (ln == 0) { d2v* Ro = (d2v*)(s_R + 4 * kb); Ro[0] = (d2v){rr[0], rr[1]}; Ro[1] = (d2v){rr[2], rr[3]}; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (ln == 0) flag_store(flags, kb + 1);
    }
    return true;
}

// ---- matrix waves: everything about tile ownership is compile time (W = wave 1..3: tile rows rA = 6 - W and rB = W - 1)
template <int W, int C2>
__device__ __forceinline__ void publish_cols(const v4f64 (&accA)[6], const v4f64 (&accB)[3], int jq, double* __restrict__ PB, int ln) {
    constexpr int rA = 6 - W, rB = W - 1;
    const int k = ln >> 4, c16 = ln & 15;
    if ((c16 >> 2) != jq) return;
    if constexpr (C2 <= rA) {
#pragma unroll
        for (int v = 0; v < 4; ++v) PB[(16 * rA + 4 * v + k) * PB_STRIDE + (ln & 3)] = accA[C2][v];
    }
    if constexpr (C2 <= rB) {
#pragma unroll
        for (int v = 0; v < 4; ++v) PB[(16 * rB + 4 * v + k) * PB_STRIDE + (ln & 3)] = accB[C2 <= rB ? C2 : 0][v];
    }
}

// round kb: rank-4 update with W(kb) of the tiles in block columns >= CBN, the block column C2 of panel kb+2 first and published
template <int W, int CBN, int C2>
__device__ __forceinline__ bool matrix_round(v4f64 (&accA)[6], v4f64 (&accB)[3], int kb, double* __restrict__ s_PB, const double* __restrict__ Lblk,
                                             const double* __restrict__ s_R, lds_i32* flags, int ln) {
    constexpr int rA = 6 - W, rB = W - 1;
    const int k = ln >> 4, c16 = ln & 15;
    if (!wait_ge(flags, kb + 1, flags + 4)) return false;
    const double* Wk = Lblk + (size_t)kb * NBS * 18 + (c16 >> 2) * 18 + (c16 & 3) * 4 + k;
    const double nrk = -s_R[4 * kb + k];
    const double fA = Wk[rA * 4 * 18];
    double fB = 0.0;
    if constexpr (CBN <= rB) fB = Wk[rB * 4 * 18];
    double fb[6];
#pragma unroll
    for (int c = CBN; c <= rA; ++c) fb[c] = Wk[c * 4 * 18] * nrk;
    if constexpr (C2 <= rA) accA[C2] = __builtin_amdgcn_mfma_f64_16x16x4f64(fA, fb[C2], accA[C2], 0, 0, 0);
    if constexpr (C2 <= rB) accB[C2 <= rB ? C2 : 0] = __builtin_amdgcn_mfma_f64_16x16x4f64(fB, fb[C2], accB[C2 <= rB ? C2 : 0], 0, 0, 0);
    publish_cols<W, C2>(accA, accB, (kb + 2) & 3, s_PB + (size_t)(kb & 1) * 96 * PB_STRIDE, ln);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (ln == 0) flag_store(flags + W, kb + 3);
#pragma unroll
    for (int c = CBN; c <= rA; ++c)
        if (c != C2) accA[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(fA, fb[c], accA[c], 0, 0, 0);
#pragma unroll
    for (int c = CBN; c <= rB; ++c)
        if (c != C2) accB[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(fB, fb[c], accB[c], 0, 0, 0);
    return true;
}

// rounds 4 CB .. 4 CB + 3: (first live block column, block column of panel kb+2) = (CB, CB), (CB, CB), (CB, CB+1), (CB+1, CB+1)
template <int W, int CB>
__device__ __forceinline__ bool matrix_block_column(v4f64 (&accA)[6], v4f64 (&accB)[3], int nrounds, double* __restrict__ s_PB, const double* __restrict__ Lblk,
                                                    const double* __restrict__ s_R, lds_i32* flags, int ln) {
    constexpr int N1 = CB < 5 ? CB + 1 : 5;
#pragma unroll 1
    for (int jq = 0; jq < 2; ++jq) {
        if (4 * CB + jq >= nrounds) return true;
        if (!matrix_round<W, CB, CB>(accA, accB, 4 * CB + jq, s_PB, Lblk, s_R, flags, ln)) return false;
    }
    if (4 * CB + 2 >= nrounds) return true;
    if (!matrix_round<W, CB, N1>(accA, accB, 4 * CB + 2, s_PB, Lblk, s_R, flags, ln)) return false;
    if (4 * CB + 3 >= nrounds) return true;
    return matrix_round<W, N1, N1>(accA, accB, 4 * CB + 3, s_PB, Lblk, s_R, flags, ln);
}

template <int W>
__device__ bool matrix_wave(v4f64 (&accA)[6], v4f64 (&accB)[3], double* __restrict__ s_PB, const double* __restrict__ Lblk, const double* __restrict__ s_R,
                            lds_i32* flags, int P, int ln) {
    const int NR = (P + 3) >> 2, nrounds = NR - 2;      // the updates of the last two rounds are never read
    // panels 0 and 1 straight from the assembled matrix
    publish_cols<W, 0>(accA, accB, 0, s_PB, ln);
    publish_cols<W, 0>(accA, accB, 1, s_PB + 96 * PB_STRIDE, ln);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (ln == 0) flag_store(flags + W, 2);
    return matrix_block_column<W, 0>(accA, accB, nrounds, s_PB, Lblk, s_R, flags, ln) && matrix_block_column<W, 1>(accA, accB, nrounds, s_PB, Lblk, s_R, flags, ln) &&
           matrix_block_column<W, 2>(accA, accB, nrounds, s_PB, Lblk, s_R, flags, ln) && matrix_block_column<W, 3>(accA, accB, nrounds, s_PB, Lblk, s_R, flags, ln) &&
           matrix_block_column<W, 4>(accA, accB, nrounds, s_PB, Lblk, s_R, flags, ln) && matrix_block_column<W, 5>(accA, accB, nrounds, s_PB, Lblk, s_R, flags, ln);
}

__global__ __launch_bounds__(256, 2) void kern(const double* __restrict__ A, double* __restrict__ Lout, double* __restrict__ Rout, long long* cyc,
                                            int P, int HS, int reps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, NB = HS >> 2, wv = __builtin_amdgcn_readfirstlane(t >> 6), ln = t & 63;
    double* Lblk = (double*)smem;
    double* s_R = Lblk + (size_t)NB * NBS * 18;
    double* s_PB = s_R + 96;
    lds_i32* flags = (lds_i32*)(s_PB + 2 * 96 * PB_STRIDE);
    long long total = 0;
    bool ok = true;
    for (int rep = 0; rep < reps; ++rep) {
        for (int e = t; e < NB * NBS * 18; e += 256) Lblk[e] = 0.0;
        if (t < 8) flags[t] = 0;
        v4f64 accA[6], accB[3];
        const int rA = 6 - wv, rB = wv - 1;
        auto elem = [&](int rb, int cb, int v) {
            const int row = 16 * rb + 4 * v + (ln >> 4), col = 16 * cb + (ln & 15);
            double val = (row == col) ? 1.0 : 0.0;
            if (row <= P && col < P) val = A[(size_t)row * HS + col];
            return val;
        };
#pragma unroll
        for (int c = 0; c < 6; ++c)
#pragma unroll
            for (int v = 0; v < 4; ++v) accA[c][v] = (wv > 0 && c <= rA) ? elem(rA, c, v) : 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int v = 0; v < 4; ++v) accB[c][v] = (wv > 0 && c <= rB) ? elem(rB, c, v) : 0.0;
        __syncthreads();
        const long long c0 = clock64();
        if (wv == 0) ok = panel_wave(s_PB, Lblk, s_R, flags, P, ln);
        else if (wv == 1) matrix_wave<1>(accA, accB, s_PB, Lblk, s_R, flags, P, ln);
        else if (wv == 2) matrix_wave<2>(accA, accB, s_PB, Lblk, s_R, flags, P, ln);
        else matrix_wave<3>(accA, accB, s_PB, Lblk, s_R, flags, P, ln);
        __syncthreads();
        total += clock64() - c0;
        if (flag_load(flags + 4)) ok = false;
    }
    if (t == 0) { cyc[0] = total; cyc[1] = !ok; }
    for (int e = t; e < NB * NBS * 18; e += 256) Lout[e] = Lblk[e];
    if (t < HS) Rout[t] = s_R[t];
}

int main() {
    const int P = 85, HS = 88, NB = HS / 4;
    std::vector<double> M((size_t)200 * P), A((size_t)HS * HS, 0.0);
    srand(1);
    for (auto& v : M) v = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < P; ++i) for (int j = 0; j < P; ++j) { double s = 0; for (int k = 0; k < 200; ++k) s += M[(size_t)k * P + i] * M[(size_t)k * P + j]; A[(size_t)i * HS + j] = s + (i == j ? 1.0 : 0.0); }
    for (int j = 0; j < P; ++j) A[(size_t)P * HS + j] = rand() / (double)RAND_MAX - 0.5;
    // host LDL^T of the bordered system: W = L diag(d) for rows 0..P, pivots 0..P-1
    std::vector<double> Wr((size_t)(P + 1) * P, 0.0), dinv(P);
    {
        std::vector<double> S((size_t)(P + 1) * P);
        for (int i = 0; i <= P; ++i) for (int j = 0; j < P; ++j) S[(size_t)i * P + j] = A[(size_t)i * HS + j];
        for (int j = 0; j < P; ++j) {
            const double dj = S[(size_t)j * P + j]; dinv[j] = 1.0 / dj;
            for (int i = j; i <= P; ++i) Wr[(size_t)i * P + j] = S[(size_t)i * P + j];
            for (int i = j + 1; i <= P; ++i) { const double l = S[(size_t)i * P + j] / dj; for (int c = j + 1; c < P && c <= i; ++c) S[(size_t)i * P + c] -= l * S[(size_t)c * P + j]; }
        }
    }
    double *dA, *dL, *dR; long long* dc;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dL, (size_t)NB * NBS * 18 * 8); hipMalloc(&dR, HS * 8); hipMalloc(&dc, 64);
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    const size_t lds = sizeof(double) * ((size_t)NB * NBS * 18 + 96 + 2 * 96 * 4) + 64;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    const int reps = 20;
    long long h[2], best = 1ll << 60;
    for (int it = 0; it < 6; ++it) { hipLaunchKernelGGL(kern, dim3(1), dim3(256), lds, 0, dA, dL, dR, dc, P, HS, reps); hipMemcpy(h, dc, 16, hipMemcpyDeviceToHost); if (h[0] < best) best = h[0]; }
    std::vector<double> L((size_t)NB * NBS * 18), R(HS);
    hipMemcpy(L.data(), dL, L.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(R.data(), dR, HS * 8, hipMemcpyDeviceToHost);
    double worst = 0.0, worst_r = 0.0;
    for (int i = 0; i <= P; ++i) for (int j = 0; j < P && j < i; ++j) {
        const double got = L[((size_t)(j >> 2) * NBS + (i >> 2)) * 18 + (i & 3) * 4 + (j & 3)], ref = Wr[(size_t)i * P + j];
        worst = fmax(worst, fabs(got - ref) / (1.0 + fabs(ref)));
    }
    for (int j = 0; j < P; ++j) worst_r = fmax(worst_r, fabs(R[j] - dinv[j]) / fabs(dinv[j]));
    printf("pipelined LDLT: %.0f clk per factorisation (%.0f per round), fail=%lld, max dev W %.3g, 1/d %.3g\n", best / (double)reps, best / (double)reps / 22, h[1], worst, worst_r);
    return 0;
}
