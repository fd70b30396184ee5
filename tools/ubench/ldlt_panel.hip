// micro-benchmark: LDL^T of k_solve's bordered 86x86 system in SIX panels of 16 pivots instead of 22 rounds of 4.
//   The lower triangle lives as 21 tiles of 16x16 in the MFMA accumulators of 4 waves (as in ldlt_mfma.hip).  Per panel:
//   (1) the owners publish tile column k to an LDS panel buffer (row-major, odd stride);
//   (2) ONE wave per 48 sub-diagonal rows factors the panel inside the wave: lane = matrix row, the 16x16 diagonal block
//       sits in lanes 0..15 of every panel wave (each keeps its own copy), pivot row and 1/d cross the wave by v_readlane -
//       no LDS, no barrier for 16 pivots; it writes W = L diag(d) in the layout the back substitution reads (Lblk) and, for
//       the matrix phase, W and -W diag(1/d) row-major;
//   (3) every wave applies the rank-16 update to its tiles with 4 matrix instructions per tile.
//   Two barriers per 16 pivots instead of one per 4; the panel wave's ~500 instructions per panel replace 4 rounds of ~1900 clk.
// Build: hipcc -O3 --offload-arch=gfx950 -o ldlt_panel ldlt_panel.hip ; prints shader clocks per factorisation and the largest
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
__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ long long g_ph[8];
#define TP(k) do { if (W == 0) { const long long _n = clock64(); if (ln == 0) g_ph[k] += _n - tl; tl = _n; } } while (0)
#define PS 17          // row stride of the panel buffers (odd: conflict-free both for row reads and for fragment reads)

// Tile ownership: wave w holds tile row rA = 5 - w (columns 0..rA) and, for w >= 2, tile row rB = w - 2 (columns 0..rB).
template <int W, int K>
__device__ __forceinline__ void publish(const v4f64 (&accA)[6], const v4f64 (&accB)[2], double* __restrict__ PB, int ln) {
    constexpr int rA = 5 - W, rB = W - 2;
    const int r4 = ln >> 4, c16 = ln & 15;
    if constexpr (K <= rA) {
#pragma unroll
        for (int v = 0; v < 4; ++v) PB[(16 * (rA - K) + 4 * v + r4) * PS + c16] = accA[K][v];
    }
    if constexpr (K < 2 && K <= rB) {
#pragma unroll
        for (int v = 0; v < 4; ++v) PB[(16 * (rB - K) + 4 * v + r4) * PS + c16] = accB[K < 2 ? K : 0][v];
    }
}

// panel K: pivots 16K .. 16K+15 (real while < P), rows 16K .. P (P = the right-hand-side row) = R rows of PB
template <int K>
__device__ __forceinline__ void factor_panel(double* __restrict__ PB, double* __restrict__ WD, double* __restrict__ Lblk, double* __restrict__ s_R,
                                             int* __restrict__ s_fail, int P, int NB, int wv, int ln) {
    const int R = P + 1 - 16 * K;                        // rows of the panel
    if (16 + 48 * wv >= R && wv > 0) return;             // wave 0: rows 0..63; wave 1: diagonal copy + rows 64..111
    const int r = ln < 16 ? ln : 16 + 48 * wv + (ln - 16);   // my row inside the panel
    const bool have = r < R;
    double a[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = have ? PB[r * PS + c] : 0.0;
    double rinv[16];
    int bad = 0;
    long long tq = clock64();
    { double dep = 0; for (int c = 0; c < 16; ++c) dep += a[c]; asm volatile("" :: "v"(dep)); }
    { const long long n = clock64(); if (wv == 0 && ln == 0) g_ph[4] += n - tq; tq = n; }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const bool real = 16 * K + j < P;
        const double dj = real ? readlane_f64(a[j], j) : 1.0;      // padding pivots: nothing to eliminate, nothing that could overflow
        bad |= (int)(real && !(dj > 0.0));
        const double rj = fast_rcp(dj);
        rinv[j] = real ? rj : 0.0;
        const double l = a[j] * rj;
#pragma unroll
        for (int c = j + 1; c < 16; ++c) {
            const double wc = readlane_f64(a[j], c);     // W(c, j), held by lane c
            a[c] = fma(-l, wc, a[c]);
        }
    }
    { double dep = 0; for (int c = 0; c < 16; ++c) dep += a[c]; asm volatile("" :: "v"(dep)); }
    { const long long n = clock64(); if (wv == 0 && ln == 0) g_ph[5] += n - tq; tq = n; }
    if (ln == 0 && wv == 0 && bad) *s_fail = 1;
    // outputs.  Lblk: block (pivot block kb, row block bi) at (kb * NB + bi) * 18, 4x4 row-major; a diagonal block keeps its
    // strictly lower part, blocks above the diagonal are never written.  Diagonal rows are written by wave 0 only.
    if (have && (ln >= 16 || wv == 0)) {
        const int rg = 16 * K + r;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int cg = 16 * K + c;
            if ((cg >> 2) <= (rg >> 2) && (cg >> 2) < NB) Lblk[((size_t)(cg >> 2) * NB + (rg >> 2)) * 18 + (rg & 3) * 4 + (cg & 3)] = (cg < rg && cg < P) ? a[c] : 0.0;
        }
    }
    if (ln >= 16 && r < 96) {       // fragment rows of the matrix phase (rows past the matrix: zeros)
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const bool realc = have && 16 * K + c < P;
            PB[r * PS + c] = realc ? a[c] : 0.0;
            WD[r * PS + c] = realc ? -a[c] * rinv[c] : 0.0;
        }
    }
    if (wv == 0 && ln < 16) {
        double v = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) v = (ln == j) ? rinv[j] : v;
        s_R[16 * K + ln] = v;
    }
    __builtin_amdgcn_s_waitcnt(0);
    { const long long n = clock64(); if (wv == 0 && ln == 0) g_ph[6] += n - tq; tq = n; }
}

template <int W, int K>
__device__ __forceinline__ void update_tiles(v4f64 (&accA)[6], v4f64 (&accB)[2], const double* __restrict__ PB, const double* __restrict__ WD, int ln) {
    constexpr int rA = 5 - W, rB = W - 2;
    const int off = (ln & 15) * PS + (ln >> 4);
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
}

template <int W, int K>
__device__ __forceinline__ void panel_step(v4f64 (&accA)[6], v4f64 (&accB)[2], double* PB, double* WD, double* Lblk, double* s_R, int* s_fail,
                                           int P, int NB, int ln) {
    if (16 * K >= P) return;
    long long tl = clock64();
    publish<W, K>(accA, accB, PB, ln);
    __syncthreads();
    TP(0);
    if (W < 2) factor_panel<K>(PB, WD, Lblk, s_R, s_fail, P, NB, W, ln);
    TP(1);
    __syncthreads();
    TP(2);
    if (16 * (K + 1) < P + 1) update_tiles<W, K>(accA, accB, PB, WD, ln);
    TP(3);
}

template <int W>
__device__ __forceinline__ void ldlt_panels(v4f64 (&accA)[6], v4f64 (&accB)[2], double* PB, double* WD, double* Lblk, double* s_R, int* s_fail,
                                            int P, int NB, int ln) {
    panel_step<W, 0>(accA, accB, PB, WD, Lblk, s_R, s_fail, P, NB, ln);
    panel_step<W, 1>(accA, accB, PB, WD, Lblk, s_R, s_fail, P, NB, ln);
    panel_step<W, 2>(accA, accB, PB, WD, Lblk, s_R, s_fail, P, NB, ln);
    panel_step<W, 3>(accA, accB, PB, WD, Lblk, s_R, s_fail, P, NB, ln);
    panel_step<W, 4>(accA, accB, PB, WD, Lblk, s_R, s_fail, P, NB, ln);
    panel_step<W, 5>(accA, accB, PB, WD, Lblk, s_R, s_fail, P, NB, ln);
}

__global__ __launch_bounds__(256) void kern(const double* __restrict__ A, double* __restrict__ Lout, double* __restrict__ Rout, long long* cyc,
                                            int P, int HS, int reps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, NB = HS >> 2, wv = t >> 6, ln = t & 63;
    double* Lblk = (double*)smem;
    double* s_R = Lblk + (size_t)NB * NB * 18;
    double* s_PB = s_R + 96;
    double* s_WD = s_PB + 96 * PS;
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
            case 0: ldlt_panels<0>(accA, accB, s_PB, s_WD, Lblk, s_R, &s_fail, P, NB, ln); break;
            case 1: ldlt_panels<1>(accA, accB, s_PB, s_WD, Lblk, s_R, &s_fail, P, NB, ln); break;
            case 2: ldlt_panels<2>(accA, accB, s_PB, s_WD, Lblk, s_R, &s_fail, P, NB, ln); break;
            default: ldlt_panels<3>(accA, accB, s_PB, s_WD, Lblk, s_R, &s_fail, P, NB, ln); break;
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
    const size_t lds = sizeof(double) * ((size_t)NB * NB * 18 + 96 + 2 * 96 * PS);
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
    long long ph[8]; (void)hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_ph), sizeof ph);
    printf("wave 0 phases (clk per factorisation, all launches averaged): publish+barrier %.0f | panel %.0f | barrier %.0f | update %.0f || inside the panel: load %.0f | 16 pivots %.0f | stores %.0f\n", ph[0] / 120.0, ph[1] / 120.0, ph[2] / 120.0, ph[3] / 120.0, ph[4] / 120.0, ph[5] / 120.0, ph[6] / 120.0);
    printf("panel LDLT: %.0f clk per factorisation (%.0f per 16-pivot panel), fail=%lld, max dev W %.3g, 1/d %.3g\n", best / (double)reps, best / (double)reps / 6, h[1], worst, worst_r);
    return 0;
}
