// micro-benchmark: LDL^T of k_solve's bordered 86x86 system with the trailing matrix in MFMA accumulators (ldlt_mfma.hip) and
// EIGHT pivots per round: 11 rounds and 22 barriers instead of 22 and 44.  Per round: the owners publish the 8 panel columns,
// every row thread (0..95) factors the 8x8 diagonal block of the panel for itself (36 entries, 8 reciprocals in a chain) and
// forms its 8 entries of W = L diag(d); then two matrix instructions (rank 4 each) per live tile.
// Build: hipcc -O3 --offload-arch=gfx950 -o ldlt_mfma8 ldlt_mfma8.hip
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
__device__ long long g_ph[4];
#define PB_STRIDE 10       // 8 panel columns + 2 of padding: rows stay 16-byte aligned, row reads of neighbouring threads spread over the banks
#define NBS 24

// the panel of round (CBN, jq): columns 16 CBN + 8 jq .. + 7 of the tiles in block column CBN
template <int W, int CBN>
__device__ __forceinline__ void publish(const v4f64 (&accA)[6], const v4f64 (&accB)[2], int jq, double* __restrict__ PB, int ln) {
    constexpr int rA = 5 - W, rB = W - 2;
    const int k = ln >> 4, c16 = ln & 15;
    if ((c16 >> 3) == jq) {
        if constexpr (CBN <= rA) {
#pragma unroll
            for (int v = 0; v < 4; ++v) PB[(16 * rA + 4 * v + k) * PB_STRIDE + (ln & 7)] = accA[CBN][v];
        }
        if constexpr (CBN < 2 && CBN <= rB) {
#pragma unroll
            for (int v = 0; v < 4; ++v) PB[(16 * rB + 4 * v + k) * PB_STRIDE + (ln & 7)] = accB[CBN < 2 ? CBN : 0][v];
        }
    }
}

// one round = 8 pivots (columns 16 CB + 8 jq ..); CBN = block column of the next panel = first block column still live
template <int W, int CB, int CBN>
__device__ __forceinline__ bool ldlt_round(v4f64 (&accA)[6], v4f64 (&accB)[2], int jq, double* __restrict__ PB,
                                           double* __restrict__ Lblk, double* __restrict__ s_R, int* __restrict__ s_fail, int P, int NR, int t) {
    constexpr int rA = 5 - W, rB = W - 2;
    const int ln = t & 63, k = ln >> 4, c16 = ln & 15, kb = 4 * CB + 2 * jq, p0 = 4 * kb;    // first block column / pivot of the round
    const long long tq0 = clock64();
    __syncthreads();                                         // the panel of this round is published
    const long long tq1 = clock64();
    if (W < 2 && t < 96) {
        // my copy of the 8x8 diagonal block (lower triangle) and my row of the panel
        double D[8][8], s[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const d2v* q = (const d2v*)(PB + (p0 + i) * PB_STRIDE);
#pragma unroll
            for (int h = 0; h <= i / 2; ++h) { const d2v v = q[h]; D[i][2 * h] = v.x; D[i][2 * h + 1] = v.y; }
        }
        {
            const d2v* q = (const d2v*)(PB + t * PB_STRIDE);
#pragma unroll
            for (int h = 0; h < 4; ++h) { const d2v v = q[h]; s[2 * h] = v.x; s[2 * h + 1] = v.y; }
        }
        double r[8], l[8][8];
        bool bad = false;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool real = p0 + j < P;
            bad |= real & !(D[j][j] > 0.0);
            r[j] = fast_rcp(D[j][j]);
#pragma unroll
            for (int i = j + 1; i < 8; ++i) l[i][j] = D[i][j] * r[j];
#pragma unroll
            for (int i = j + 1; i < 8; ++i)
#pragma unroll
                for (int c = j + 1; c <= i; ++c) D[i][c] = fma(-l[i][j], D[c][j], D[i][c]);
        }
        double w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            double a = s[j];
#pragma unroll
            for (int c = 0; c < j; ++c) a = fma(-w[c], l[j][c], a);
            w[j] = a;
        }
        // W row into the two block columns of the round; a diagonal 4x4 block keeps its strictly lower part, blocks above the
        // diagonal are not written
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int rr = t - 4 * (kb + h);               // row inside the 4x4 diagonal block of block column kb + h (if 0..3)
            if (rr >= 0) {
                d2v* Wo = (d2v*)(Lblk + ((size_t)(kb + h) * NBS + (t >> 2)) * 18 + (t & 3) * 4);
                Wo[0] = (d2v){rr < 1 ? 0.0 : w[4 * h], rr < 2 ? 0.0 : w[4 * h + 1]};
                Wo[1] = (d2v){rr < 3 ? 0.0 : w[4 * h + 2], rr < 4 ? 0.0 : w[4 * h + 3]};
            }
        }
        if (t == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) s_R[p0 + j] = (p0 + j < P) ? r[j] : 0.0;
            if (bad) *s_fail = 1;
        }
    }
    const long long tq2 = clock64();
    __syncthreads();                                         // W rows, reciprocal pivots and the failure flag are visible
    const long long tq3 = clock64();
    if (t == 0) { g_ph[0] += tq1 - tq0; g_ph[1] += tq2 - tq1; g_ph[2] += tq3 - tq2; }
    if (*s_fail) return false;
    if (2 * (2 * CB + jq) + 2 >= NR) return true;            // no pivots left
    // ---- fragments: lane (c16, k) holds row 16 b + c16, pivots k and 4 + k of the round (two matrix k-steps)
    const double* Wk = Lblk + (size_t)kb * NBS * 18 + (c16 >> 2) * 18 + (c16 & 3) * 4 + k;
    const double rk0 = s_R[p0 + k], rk1 = s_R[p0 + 4 + k];
    if constexpr (CBN <= rA) {
        double fw0[6], fw1[6];
#pragma unroll
        for (int c = CBN; c <= rA; ++c) { fw0[c] = Wk[c * 4 * 18]; fw1[c] = Wk[NBS * 18 + c * 4 * 18]; }
        constexpr bool useB = rB >= 0 && CBN <= rB;
        double fB0 = 0.0, fB1 = 0.0;
        if constexpr (useB) { fB0 = Wk[(rB > 0 ? rB : 0) * 4 * 18]; fB1 = Wk[NBS * 18 + (rB > 0 ? rB : 0) * 4 * 18]; }
        double fb0[6], fb1[6];
#pragma unroll
        for (int c = CBN; c <= rA; ++c) { fb0[c] = fw0[c] * -rk0; fb1[c] = fw1[c] * -rk1; }
        // next panel's block column first
        accA[CBN] = __builtin_amdgcn_mfma_f64_16x16x4f64(fw0[rA], fb0[CBN], accA[CBN], 0, 0, 0);
        accA[CBN] = __builtin_amdgcn_mfma_f64_16x16x4f64(fw1[rA], fb1[CBN], accA[CBN], 0, 0, 0);
        if constexpr (useB) {
#pragma unroll
            for (int c = CBN; c <= rB; ++c) {
                accB[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(fB0, fb0[c], accB[c], 0, 0, 0);
                accB[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(fB1, fb1[c], accB[c], 0, 0, 0);
            }
        }
#pragma unroll
        for (int c = CBN + 1; c <= rA; ++c) {
            accA[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(fw0[rA], fb0[c], accA[c], 0, 0, 0);
            accA[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(fw1[rA], fb1[c], accA[c], 0, 0, 0);
        }
    }
    publish<W, CBN>(accA, accB, (jq + 1) & 1, PB, ln);
    if (t == 0) g_ph[3] += clock64() - tq3;
    return true;
}

template <int W, int CB>
__device__ __forceinline__ bool ldlt_block_column(v4f64 (&accA)[6], v4f64 (&accB)[2], double* __restrict__ PB,
                                                  double* __restrict__ Lblk, double* __restrict__ s_R, int* s_fail, int P, int NR, int t) {
    if (4 * CB >= NR) return true;
    if (!ldlt_round<W, CB, CB>(accA, accB, 0, PB, Lblk, s_R, s_fail, P, NR, t)) return false;
    if (4 * CB + 2 >= NR) return true;
    return ldlt_round<W, CB, (CB < 5 ? CB + 1 : 5)>(accA, accB, 1, PB, Lblk, s_R, s_fail, P, NR, t);
}

template <int W>
__device__ __forceinline__ bool ldlt_rounds(v4f64 (&accA)[6], v4f64 (&accB)[2], double* __restrict__ PB,
                                            double* __restrict__ Lblk, double* __restrict__ s_R, int* s_fail, int P, int t) {
    const int NR = (P + 3) >> 2;          // 4-pivot block columns with pivots: 0..NR-1
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
    long long ph[4]; (void)hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_ph), sizeof ph);
    printf("thread 0, clocks per factorisation: barrier A %.0f | phase 1 %.0f | barrier B %.0f | phase 2 + publish %.0f\n", ph[0] / 120.0, ph[1] / 120.0, ph[2] / 120.0, ph[3] / 120.0);
    printf("mfma 8-pivot LDLT: %.0f clk per factorisation (%.0f per 4 pivots), fail=%lld, max dev W %.3g, 1/d %.3g\n", best / (double)reps, best / (double)reps / 22, h[1], worst, worst_r);
    return 0;
}
