// micro-benchmark: LDL^T of k_solve's bordered 86x86 system in SIX panels of 16 pivots, factored IN PLACE in the MFMA
// accumulator registers with DPP row broadcasts - no v_readlane, no LDS traffic inside a panel.
//   The lower triangle lives as 21 tiles of 16x16 in the accumulators of 4 waves: lane (g = l >> 4, c = l & 15) of the owner
//   holds rows 4v + g (v = 0..3) of column c of a tile, i.e. a DPP row of 16 lanes = the 16 columns of the panel.  Per panel k:
//   (1) the owner of the diagonal tile publishes it (LDS, 2 KB); after a barrier every lane reads ITS column of it (16 values);
//   (2) 16 pivots, every wave alike and without communication: d_j = column j's entry of row j, broadcast inside the DPP row
//       (v_mov_b64_dpp row_newbcast:j); lane c > j scales its own entry of row j to -L(c,j) and applies
//       A(i,c) += A(i,j) * (-L(c,j)) to the rows it holds - its copy of the diagonal tile and its 4 rows of every
//       sub-diagonal tile of the panel - with ONE v_fmac_f64_dpp each (the broadcast of A(i,j) from lane j is the DPP operand);
//   (3) the tiles' W = L diag(d) and -W diag(1/d) go to LDS row-major, barrier, rank-16 update of the trailing tiles with 4
//       matrix instructions per tile; the W rows are copied into the block layout the back substitution reads.
//   Two barriers per 16 pivots; ~25 vector instructions per pivot and wave.
// Build: hipcc -O3 --offload-arch=gfx950 -o ldlt_dpp ldlt_dpp.hip ; prints shader clocks per factorisation and the largest
// deviation from a host LDL^T.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef double v4f64 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double fast_rcp(double d) {
    const double r0 = __builtin_amdgcn_rcp(d);
    const double e = fma(-d, r0, 1.0);
    const double t2 = fma(e, e, e);
    return fma(r0, t2, r0);
}
// lane J of every row of 16 lanes -> the whole row (all lanes must be switched on)
template <int J>
__device__ __forceinline__ double bcast(double v) {
    double r;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(J));
    return r;
}
// acc += (x of lane J of my row) * ns
template <int J>
__device__ __forceinline__ void fmac_bcast(double& acc, double x, double ns) {
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(ns), "n"(J));
}
#define PS 17          // row stride of the panel buffers (odd: conflict-free both for row-major stores and for fragment reads)

// Tile ownership: wave w holds tile row rA = 5 - w (columns 0..rA) and, for w >= 2, tile row rB = w - 2 (columns 0..rB).
template <int W, int K, int J>
__device__ __forceinline__ void pivot(double (&ad)[16], v4f64 (&accA)[6], v4f64 (&accB)[2], double& rmine, int& bad, int P, int c16) {
    constexpr int rA = 5 - W, rB = W - 2;
    const bool real = 16 * K + J < P;
    const double dj = bcast<J>(ad[J]);
    bad |= (int)(real && !(dj > 0.0));
    const double rj = real ? fast_rcp(dj) : 0.0;
    rmine = (c16 == J) ? rj : rmine;
    const double ns = (c16 > J) ? -(ad[J] * rj) : 0.0;             // -L(c, J) for my column c; columns <= J are final
    // the row that holds the next pivot first: its reciprocal chain can start while the other rows are still being updated
#pragma unroll
    for (int i = J + 1; i < 16; ++i) fmac_bcast<J>(ad[i], ad[i], ns);
    if constexpr (K < rA) {
#pragma unroll
        for (int v = 0; v < 4; ++v) { double x = accA[K][v]; fmac_bcast<J>(x, x, ns); accA[K][v] = x; }
    }
    if constexpr (rB >= 0 && K < rB) {
#pragma unroll
        for (int v = 0; v < 4; ++v) { double x = accB[K][v]; fmac_bcast<J>(x, x, ns); accB[K][v] = x; }
    }
}

template <int W, int K>
__device__ __forceinline__ void panel_step(v4f64 (&accA)[6], v4f64 (&accB)[2], double* __restrict__ DG, double* __restrict__ PB, double* __restrict__ WD,
                                           double* __restrict__ Lblk, double* __restrict__ s_R, int* __restrict__ s_fail, int P, int NB, int t) {
    constexpr int rA = 5 - W, rB = W - 2;
    if (16 * K >= P) return;
    const int ln = t & 63, g = ln >> 4, c16 = ln & 15;
    // (1) the diagonal tile (K, K): owned as accA[K] by the wave whose rA == K, as accB[K] by the wave whose rB == K
    if constexpr (rA == K) {
#pragma unroll
        for (int v = 0; v < 4; ++v) DG[(4 * v + g) * PS + c16] = accA[K][v];
    }
    if constexpr (rB == K) {
#pragma unroll
        for (int v = 0; v < 4; ++v) DG[(4 * v + g) * PS + c16] = accB[K][v];
    }
    __syncthreads();
    double ad[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) ad[j] = DG[j * PS + c16];
    // (2) sixteen pivots in registers
    double rmine = 0.0;
    int bad = 0;
    pivot<W, K, 0>(ad, accA, accB, rmine, bad, P, c16); pivot<W, K, 1>(ad, accA, accB, rmine, bad, P, c16);
    pivot<W, K, 2>(ad, accA, accB, rmine, bad, P, c16); pivot<W, K, 3>(ad, accA, accB, rmine, bad, P, c16);
    pivot<W, K, 4>(ad, accA, accB, rmine, bad, P, c16); pivot<W, K, 5>(ad, accA, accB, rmine, bad, P, c16);
    pivot<W, K, 6>(ad, accA, accB, rmine, bad, P, c16); pivot<W, K, 7>(ad, accA, accB, rmine, bad, P, c16);
    pivot<W, K, 8>(ad, accA, accB, rmine, bad, P, c16); pivot<W, K, 9>(ad, accA, accB, rmine, bad, P, c16);
    pivot<W, K, 10>(ad, accA, accB, rmine, bad, P, c16); pivot<W, K, 11>(ad, accA, accB, rmine, bad, P, c16);
    pivot<W, K, 12>(ad, accA, accB, rmine, bad, P, c16); pivot<W, K, 13>(ad, accA, accB, rmine, bad, P, c16);
    pivot<W, K, 14>(ad, accA, accB, rmine, bad, P, c16); pivot<W, K, 15>(ad, accA, accB, rmine, bad, P, c16);
    // (3) W and -W/d of my tiles, row-major (row = 16 (I - K) + 4 v + g); the diagonal tile's W by wave 0 (every wave has a copy)
    const bool realc = 16 * K + c16 < P;
    if constexpr (K < rA) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const double w = realc ? accA[K][v] : 0.0;
            PB[(16 * (rA - K) + 4 * v + g) * PS + c16] = w; WD[(16 * (rA - K) + 4 * v + g) * PS + c16] = -w * rmine;
        }
    }
    if constexpr (rB >= 0 && K < rB) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const double w = realc ? accB[K][v] : 0.0;
            PB[(16 * (rB - K) + 4 * v + g) * PS + c16] = w; WD[(16 * (rB - K) + 4 * v + g) * PS + c16] = -w * rmine;
        }
    }
    if (W == 0) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            double w = 0.0;
#pragma unroll
            for (int i = 0; i < 16; ++i) w = (i == 4 * v + g) ? ad[i] : w;
            PB[(4 * v + g) * PS + c16] = (realc && c16 < 4 * v + g) ? w : 0.0;     // strictly lower part of the diagonal tile
        }
        if (g == 0) s_R[16 * K + c16] = rmine;
        if (bad) *s_fail = 1;
    }
    __syncthreads();
    // rank-16 update of my trailing tiles
    const int off = c16 * PS + g;
    if constexpr (K < rA) {
        double fa[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) fa[s] = PB[16 * (rA - K) * PS + off + 4 * s];
#pragma unroll
        for (int c = K + 1; c <= rA; ++c) {
            double fb[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) fb[s] = WD[16 * (c - K) * PS + off + 4 * s];
#pragma unroll
            for (int s = 0; s < 4; ++s) accA[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[s], fb[s], accA[c], 0, 0, 0);
        }
    }
    if constexpr (rB >= 0 && K < rB) {
        double fa[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) fa[s] = PB[16 * (rB - K) * PS + off + 4 * s];
#pragma unroll
        for (int c = K + 1; c <= rB; ++c) {
            double fb[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) fb[s] = WD[16 * (c - K) * PS + off + 4 * s];
#pragma unroll
            for (int s = 0; s < 4; ++s) accB[c < 2 ? c : 0] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[s], fb[s], accB[c < 2 ? c : 0], 0, 0, 0);
        }
    }
    // W rows of the panel -> the block layout of the back substitution: block (kb, bi) at (kb * NB + bi) * 18, 4x4 row-major;
    // blocks above the diagonal are never written, the diagonal blocks keep their strictly lower part (PB already has it so)
    const int R = P + 1 - 16 * K;                      // rows 16K .. P
    for (int e = t; e < R * 16; e += 256) {
        const int r = e >> 4, c = e & 15, rg = 16 * K + r, cg = 16 * K + c;
        if ((cg >> 2) <= (rg >> 2) && (cg >> 2) < NB) Lblk[((size_t)(cg >> 2) * NB + (rg >> 2)) * 18 + (rg & 3) * 4 + (cg & 3)] = PB[r * PS + c];
    }
}

template <int W>
__device__ __forceinline__ void ldlt_panels(v4f64 (&accA)[6], v4f64 (&accB)[2], double* DG, double* PB, double* WD, double* Lblk, double* s_R, int* s_fail,
                                            int P, int NB, int t) {
    panel_step<W, 0>(accA, accB, DG, PB, WD, Lblk, s_R, s_fail, P, NB, t);
    panel_step<W, 1>(accA, accB, DG, PB, WD, Lblk, s_R, s_fail, P, NB, t);
    panel_step<W, 2>(accA, accB, DG, PB, WD, Lblk, s_R, s_fail, P, NB, t);
    panel_step<W, 3>(accA, accB, DG, PB, WD, Lblk, s_R, s_fail, P, NB, t);
    panel_step<W, 4>(accA, accB, DG, PB, WD, Lblk, s_R, s_fail, P, NB, t);
    panel_step<W, 5>(accA, accB, DG, PB, WD, Lblk, s_R, s_fail, P, NB, t);
}

__global__ __launch_bounds__(256) void kern(const double* __restrict__ A, double* __restrict__ Lout, double* __restrict__ Rout, long long* cyc,
                                            int P, int HS, int reps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, NB = HS >> 2, wv = t >> 6, ln = t & 63;
    double* Lblk = (double*)smem;
    double* s_R = Lblk + (size_t)NB * NB * 18;
    double* s_PB = s_R + 96;
    double* s_WD = s_PB + 96 * PS;
    double* s_DG = s_WD + 96 * PS;
    __shared__ int s_fail;
    long long total = 0;
    for (int rep = 0; rep < reps; ++rep) {
        for (int e = t; e < NB * NB * 18; e += 256) Lblk[e] = 0.0;
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
            case 0: ldlt_panels<0>(accA, accB, s_DG, s_PB, s_WD, Lblk, s_R, &s_fail, P, NB, t); break;
            case 1: ldlt_panels<1>(accA, accB, s_DG, s_PB, s_WD, Lblk, s_R, &s_fail, P, NB, t); break;
            case 2: ldlt_panels<2>(accA, accB, s_DG, s_PB, s_WD, Lblk, s_R, &s_fail, P, NB, t); break;
            default: ldlt_panels<3>(accA, accB, s_DG, s_PB, s_WD, Lblk, s_R, &s_fail, P, NB, t); break;
        }
        __syncthreads();
        total += clock64() - c0;
    }
    if (t == 0) { cyc[0] = total; cyc[1] = s_fail; }
    for (int e = t; e < NB * NB * 18; e += 256) Lout[e] = Lblk[e];
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
    (void)hipMalloc(&dA, A.size() * 8); (void)hipMalloc(&dL, (size_t)NB * NB * 18 * 8); (void)hipMalloc(&dR, HS * 8); (void)hipMalloc(&dc, 64);
    (void)hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    const size_t lds = sizeof(double) * ((size_t)NB * NB * 18 + 96 + 2 * 96 * PS + 16 * PS);
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    const int reps = 20;
    long long h[2], best = 1ll << 60;
    for (int it = 0; it < 6; ++it) { hipLaunchKernelGGL(kern, dim3(1), dim3(256), lds, 0, dA, dL, dR, dc, P, HS, reps); (void)hipMemcpy(h, dc, 16, hipMemcpyDeviceToHost); if (h[0] < best) best = h[0]; }
    std::vector<double> L((size_t)NB * NB * 18), R(HS);
    (void)hipMemcpy(L.data(), dL, L.size() * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(R.data(), dR, HS * 8, hipMemcpyDeviceToHost);
    double worst = 0.0, worst_r = 0.0;
    for (int i = 0; i <= P; ++i) for (int j = 0; j < P && j < i; ++j) {
        const double got = L[((size_t)(j >> 2) * NB + (i >> 2)) * 18 + (i & 3) * 4 + (j & 3)], ref = Wr[(size_t)i * P + j];
        worst = fmax(worst, fabs(got - ref) / (1.0 + fabs(ref)));
    }
    for (int j = 0; j < P; ++j) worst_r = fmax(worst_r, fabs(R[j] - dinv[j]) / fabs(dinv[j]));
    printf("dpp panel LDLT: %.0f clk per factorisation (%.0f per 16-pivot panel), fail=%lld, max dev W %.3g, 1/d %.3g\n", best / (double)reps, best / (double)reps / 6, h[1], worst, worst_r);
    return 0;
}
