// micro-benchmark: LDL^T of k_solve's bordered 86x86 system with the trailing matrix held in MFMA accumulators
// (v_mfma_f64_16x16x4_f64) instead of 4x4 register blocks updated by the vector ALU (ldlt.hip, the shipped scheme).
//   The lower triangle lives as 21 tiles of 16x16 in the accumulators of 4 waves; a round takes 4 pivots: the tile owners
//   publish the 4 panel columns to LDS, the 96 row lanes factor the 4x4 diagonal block and form their W rows, every wave
//   reads its A = W(:,k) and B = -W(:,k)/d_k fragments and issues one matrix instruction per live tile (rank-4 update,
//   next panel's block column first), two barriers per round.
// Measured on MI355X, clocks per 4-pivot round: this file 1657 (phase 1 ~850, phase 2 450-670 per wave); a one-barrier
// variant in which every wave factors the diagonal block and builds its fragments from the raw panel 1906; shipped
// register-blocked scheme 1897.  A lone wave issues one instruction per ~5 clocks and each round is a chain of
// LDS round trip -> 4 dependent reciprocals -> LDS round trip -> matrix instruction -> LDS round trip, so the matrix
// cores do not shorten the round much; NOT adopted (13 % of the factorisation = 2 us per k_solve for a second code path).
// Build: hipcc -O3 --offload-arch=gfx950 -o ldlt_mfma ldlt_mfma.hip ; prints shader clocks per factorisation and the
// largest deviation from a host LDL^T.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef double d2v __attribute__((ext_vector_type(2)));
typedef double v4f64 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double fast_rcp(double d) {
    const double r0 = __builtin_amdgcn_rcp(d);
    const double e = fma(-d, r0, 1.0);
    const double t2 = fma(e, e, e);
    return fma(r0, t2, r0);
}
// Tile ownership: wave w holds tile row rA = 5 - w (columns 0..rA) and, for w >= 2, tile row rB = w - 2 (columns 0..rB).
// Two barriers per round:
//   phase 1 (threads 0..95, one matrix row each): factor the 4x4 diagonal block of the published panel, W row = raw L^-T,
//            stored in the layout the back substitution reads (Lblk[kb][row block][18], NBS = 24 row blocks);
//   phase 2 (all waves): A = W(:,k) and B = -W(:,k)/d_k fragments straight from Lblk / s_R, one matrix instruction per live
//            tile (next panel's block column first), publish the next panel's 4 columns from the accumulators.
#define PB_STRIDE 4
#define NBS 24

template <int W, int CBN>
__device__ __forceinline__ void publish(const v4f64 (&accA)[6], const v4f64 (&accB)[2], int jq, double* __restrict__ PB, int ln) {
    constexpr int rA = 5 - W, rB = W - 2;
    const int k = ln >> 4, c16 = ln & 15;
    if ((c16 >> 2) == jq) {
        if constexpr (CBN <= rA) {
#pragma unroll
            for (int v = 0; v < 4; ++v) PB[(16 * rA + 4 * v + k) * PB_STRIDE + (ln & 3)] = accA[CBN][v];
        }
        if constexpr (CBN < 2 && CBN <= rB) {
#pragma unroll
            for (int v = 0; v < 4; ++v) PB[(16 * rB + 4 * v + k) * PB_STRIDE + (ln & 3)] = accB[CBN < 2 ? CBN : 0][v];
        }
    }
}

// one round = 4 pivots (columns 16 CB + 4 jq ..); CBN = block column of the next panel = first block column still live
template <int W, int CB, int CBN>
__device__ __forceinline__ bool ldlt_round(v4f64 (&accA)[6], v4f64 (&accB)[2], int jq, double* __restrict__ PB,
                                           double* __restrict__ Lblk, double* __restrict__ s_R, int* __restrict__ s_fail, int P, int NR, int t) {
    constexpr int rA = 5 - W, rB = W - 2;
    const int ln = t & 63, k = ln >> 4, c16 = ln & 15, kb = 4 * CB + jq;
    __syncthreads();                                         // the panel of this round is published
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
        D33 = fma(-l32, D32, D33);
        const double r3 = fast_rcp(D33);
        const bool real1 = 4 * kb + 1 < P, real2 = 4 * kb + 2 < P, real3 = 4 * kb + 3 < P;
        const bool bad = !(D00 > 0.0) | (real1 & !(D11 > 0.0)) | (real2 & !(D22 > 0.0)) | (real3 & !(D33 > 0.0));
        const double w0 = s01.x, w1 = fma(-w0, l10, s01.y), w2 = fma(-w1, l21, fma(-w0, l20, s23.x));
        const double w3 = fma(-w2, l32, fma(-w1, l31, fma(-w0, l30, s23.y)));
        const int rr = t - 4 * kb;     // row inside the trailing part; the diagonal block keeps its strictly lower part
        if (rr >= 0) {
            d2v* Wo = (d2v*)(Lblk + ((size_t)kb * NBS + (t >> 2)) * 18 + (t & 3) * 4);
            Wo[0] = (d2v){rr < 1 ? 0.0 : w0, rr < 2 ? 0.0 : w1};
            Wo[1] = (d2v){rr < 3 ? 0.0 : w2, rr < 4 ? 0.0 : w3};
        }
        if (t == 0) {
            d2v* Ro = (d2v*)(s_R + 4 * kb);
            Ro[0] = (d2v){r0, real1 ? r1 : 0.0}; Ro[1] = (d2v){real2 ? r2 : 0.0, real3 ? r3 : 0.0};
            if (bad) *s_fail = 1;
        }
    }
    __syncthreads();                                         // W rows, reciprocal pivots and the failure flag are visible
    // ---- fragments: lane (c16, k) holds row 16 b + c16, pivot k of the round
    const double* Wk = Lblk + (size_t)kb * NBS * 18 + (c16 >> 2) * 18 + (c16 & 3) * 4 + k;
    double fw[6];
#pragma unroll
    for (int c = CBN; c <= rA; ++c) fw[c] = Wk[c * 4 * 18];
    constexpr bool useB = rB >= 0 && CBN <= rB;
    double fAB = 0.0;
    if constexpr (useB) fAB = Wk[(rB > 0 ? rB : 0) * 4 * 18];
    const double rk = s_R[4 * kb + k];
    if (*s_fail) return false;
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
    publish<W, CBN>(accA, accB, (jq + 1) & 3, PB, ln);
    return true;
}

template <int W, int CB>
__device__ __forceinline__ bool ldlt_block_column(v4f64 (&accA)[6], v4f64 (&accB)[2], double* __restrict__ PB,
                                                  double* __restrict__ Lblk, double* __restrict__ s_R, int* s_fail, int P, int NR, int t) {
#pragma unroll 1
    for (int jq = 0; jq < 3; ++jq) {
        if (4 * CB + jq >= NR) return true;
        if (!ldlt_round<W, CB, CB>(accA, accB, jq, PB, Lblk, s_R, s_fail, P, NR, t)) return false;
    }
    if (4 * CB + 3 >= NR) return true;
    return ldlt_round<W, CB, (CB < 5 ? CB + 1 : 5)>(accA, accB, 3, PB, Lblk, s_R, s_fail, P, NR, t);
}

template <int W>
__device__ __forceinline__ bool ldlt_rounds(v4f64 (&accA)[6], v4f64 (&accB)[2], double* __restrict__ PB,
                                            double* __restrict__ Lblk, double* __restrict__ s_R, int* s_fail, int P, int t) {
    const int NR = (P + 3) >> 2;          // rounds: pivots 0..P-1
    publish<W, 0>(accA, accB, 0, PB, t & 63);
    return ldlt_block_column<W, 0>(accA, accB, PB, Lblk, s_R, s_fail, P, NR, t) && ldlt_block_column<W, 1>(accA, accB, PB, Lblk, s_R, s_fail, P, NR, t) &&
           ldlt_block_column<W, 2>(accA, accB, PB, Lblk, s_R, s_fail, P, NR, t) && ldlt_block_column<W, 3>(accA, accB, PB, Lblk, s_R, s_fail, P, NR, t) &&
           ldlt_block_column<W, 4>(accA, accB, PB, Lblk, s_R, s_fail, P, NR, t) && ldlt_block_column<W, 5>(accA, accB, PB, Lblk, s_R, s_fail, P, NR, t);
}

__global__ __launch_bounds__(256) void kern(const double* __restrict__ A, double* __restrict__ Lout, double* __restrict__ Rout, long long* cyc,
                                            int P, int HS, int reps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, NB = HS >> 2, wv = t >> 6, ln = t & 63;
    double* Lblk = (double*)smem;
    double* s_R = Lblk + (size_t)NB * NBS * 18;
    double* s_PB = s_R + 96;
    __shared__ int s_fail;
    long long total = 0;
    bool ok = true;
    for (int rep = 0; rep < reps; ++rep) {
        for (int e = t; e < NB * NBS * 18; e += 256) Lblk[e] = 0.0;
        if (t == 0) s_fail = 0;
        v4f64 accA[6], accB[2];
        const int rA = 5 - wv, rB = wv - 2;
        auto elem = [&](int rb, int cb, int v) {
            const int row = 16 * rb + 4 * v + (ln >> 4), col = 16 * cb + (ln & 15);
            double val = (row == col) ? 1.0 : 0.0;
            if (row <= P && col < P) val = A[(size_t)row * HS + col];
            return val;
        };
#pragma unroll
        for (int c = 0; c < 6; ++c)
#pragma unroll
            for (int v = 0; v < 4; ++v) accA[c][v] = (c <= rA) ? elem(rA, c, v) : 0.0;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int v = 0; v < 4; ++v) accB[c][v] = (c <= rB) ? elem(rB, c, v) : 0.0;
        __syncthreads();
        const long long c0 = clock64();
        switch (wv) {
            case 0: ok = ldlt_rounds<0>(accA, accB, s_PB, Lblk, s_R, &s_fail, P, t); break;
            case 1: ok = ldlt_rounds<1>(accA, accB, s_PB, Lblk, s_R, &s_fail, P, t); break;
            case 2: ok = ldlt_rounds<2>(accA, accB, s_PB, Lblk, s_R, &s_fail, P, t); break;
            default: ok = ldlt_rounds<3>(accA, accB, s_PB, Lblk, s_R, &s_fail, P, t); break;
        }
        __syncthreads();
        total += clock64() - c0;
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
    const size_t lds = sizeof(double) * ((size_t)NB * NBS * 18 + 96 + 2 * 96 * PB_STRIDE);
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
    printf("mfma LDLT: %.0f clk per factorisation (%.0f per round), fail=%lld, max dev W %.3g, 1/d %.3g\n", best / (double)reps, best / (double)reps / 22, h[1], worst, worst_r);
    return 0;
}
