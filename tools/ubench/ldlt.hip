// micro-benchmark of k_solve's register-blocked LDL^T variants on gfx950 (one workgroup of 256 threads):
//   V0: two barriers per round (panel phase, trailing phase);  V1: one barrier per round, look-ahead on the panel column;
//   V2/V3: V1 with only the barriers / only the rank-4 update; V5/V6: rank-4 update with only its LDS reads / only its FMAs
//   (timing only, wrong results).
// Build: hipcc -O3 --offload-arch=gfx950 -o ldlt ldlt.hip ; prints shader clocks for the 22-round factorisation of an 86x86
// bordered system and a checksum of the factor.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double fast_rcp(double d) {
    const double r0 = __builtin_amdgcn_rcp(d);
    const double e = fma(-d, r0, 1.0);
    const double t2 = fma(e, e, e);
    return fma(r0, t2, r0);
}
#define LPROBE(k) do {} while (0)
#define CSTAMP(k, d) do {} while (0)
#define TPROBE(k) do {} while (0)
template <int VAR>
__global__ __launch_bounds__(VAR == 10 ? 512 : 256) void kern(const double* __restrict__ A, double* __restrict__ Lout, long long* cyc, int P, int HS, int reps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = (VAR == 10) ? (threadIdx.x >> 1) : threadIdx.x, half = threadIdx.x & 1;   // V10: two lanes per 4x4 block
    const int NB = HS >> 2, NBk = NB;
    double* Lblk = (double*)smem;
    double* s_W = Lblk + (size_t)NBk * NBk * 18;
    double* s_D = s_W + 2 * (size_t)NBk * 18;
    __shared__ int s_failf[2];
    __shared__ int s_fail;
    int bi = -1, bj = -1;
    if (t < NB * (NB + 1) / 2) { int r0 = 0, rem = t; while (rem > r0) { rem -= r0 + 1; ++r0; } bi = r0; bj = rem; }
    long long total = 0;
    bool fail = false;
    for (int rep = 0; rep < reps; ++rep) {
        double a4[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int row = 4 * bi + r, col = 4 * bj + c;
                double v = (row == col) ? 1.0 : 0.0;
                if (bi >= 0 && row <= P && col < P) v = A[(size_t)row * HS + col];
                a4[r][c] = v;
            }
        __syncthreads();
        const long long c0 = clock64();
        if constexpr (VAR == 0 || VAR == 7 || VAR == 8) {
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
            fail = false;
            for (int kb = 0; kb < NB; ++kb) {
                __syncthreads();                                    // B1: diagonal block kb is visible
                if (bj == kb) {
                    const d2v* Dq = (const d2v*)s_D;
                    const d2v q0 = Dq[0], q2 = Dq[2], q4 = Dq[4], q5 = Dq[5], q6 = Dq[6], q7 = Dq[7];
                    const double D00 = q0.x, D10 = q2.x;
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
                    d2v* Wo = (d2v*)(s_W + (size_t)bi * 18);
                    d2v* Lo = (d2v*)(Lblk + ((size_t)kb * NB + bi) * 18);
        #pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double w0 = a4[r][0];
                        const double w1 = fma(-w0, l10, a4[r][1]);
                        const double w2 = fma(-w1, l21, fma(-w0, l20, a4[r][2]));
                        const double w3 = fma(-w2, l32, fma(-w1, l31, fma(-w0, l30, a4[r][3])));
                        if constexpr (VAR == 8) { Lo[2 * r] = (d2v){w0, w1}; Lo[2 * r + 1] = (d2v){w2, w3}; }
                        else {
                        Wo[2 * r] = (d2v){w0, w1}; Wo[2 * r + 1] = (d2v){w2, w3};
                        Lo[2 * r] = (d2v){w0 * r0, w1 * r1}; Lo[2 * r + 1] = (d2v){w2 * r2, w3 * r3};
                        }
                    }
                    if (VAR == 8 && bi == kb) { d2v* Ro = (d2v*)(s_W + 4 * kb); Ro[0] = (d2v){r0, r1}; Ro[1] = (d2v){r2, r3}; }
                }
                __syncthreads();                                    // B2: W and L of pivot block kb are visible
                if (s_fail) { fail = true; break; }
                if (bj > kb) {
                    const d2v* Wi = (VAR == 8) ? (const d2v*)(Lblk + ((size_t)kb * NB + bi) * 18) : (const d2v*)(s_W + (size_t)bi * 18);
                    const d2v* Lj = (const d2v*)(Lblk + ((size_t)kb * NB + bj) * 18);
                    d2v wv[4][2], lv[4][2];
        #pragma unroll
                    for (int r = 0; r < 4; ++r) { wv[r][0] = Wi[2 * r]; wv[r][1] = Wi[2 * r + 1]; lv[r][0] = Lj[2 * r]; lv[r][1] = Lj[2 * r + 1]; }
                    if constexpr (VAR == 8) {
                        const d2v* Rq = (const d2v*)(s_W + 4 * kb);
                        const d2v ra = Rq[0], rb = Rq[1];
        #pragma unroll
                        for (int r = 0; r < 4; ++r) { lv[r][0].x *= ra.x; lv[r][0].y *= ra.y; lv[r][1].x *= rb.x; lv[r][1].y *= rb.y; }
                    }
        #pragma unroll
                    for (int r = 0; r < 4; ++r)
        #pragma unroll
                        for (int cc = 0; cc < 4; ++cc) {
                            if constexpr (VAR == 7) continue;
                            double v = a4[r][cc];
                            v = fma(-wv[r][0].x, lv[cc][0].x, v);
                            v = fma(-wv[r][0].y, lv[cc][0].y, v);
                            v = fma(-wv[r][1].x, lv[cc][1].x, v);
                            v = fma(-wv[r][1].y, lv[cc][1].y, v);
                            a4[r][cc] = v;
                        }
                    if constexpr (VAR == 7) {   // k-major: 16 independent FMAs per step
        #pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4) {
        #pragma unroll
                            for (int r = 0; r < 4; ++r)
        #pragma unroll
                                for (int cc = 0; cc < 4; ++cc) {
                                    const double wq = (s4 == 0) ? wv[r][0].x : (s4 == 1) ? wv[r][0].y : (s4 == 2) ? wv[r][1].x : wv[r][1].y;
                                    const double lq = (s4 == 0) ? lv[cc][0].x : (s4 == 1) ? lv[cc][0].y : (s4 == 2) ? lv[cc][1].x : lv[cc][1].y;
                                    a4[r][cc] = fma(-wq, lq, a4[r][cc]);
                                }
                            asm volatile("" : "+v"(a4[0][0]), "+v"(a4[1][1]), "+v"(a4[2][2]), "+v"(a4[3][3]));
                        }
                    }
                    if (bi == kb + 1 && bj == kb + 1) {             // publish the next diagonal block
                        d2v* Do = (d2v*)s_D;
        #pragma unroll
                        for (int r = 0; r < 4; ++r) { Do[2 * r] = (d2v){a4[r][0], a4[r][1]}; Do[2 * r + 1] = (d2v){a4[r][2], a4[r][3]}; }
                    }
                }
            }


            __syncthreads();
        } else if constexpr (VAR == 10) {
            // V10 = V9 with two lanes per block: lane `half` owns rows 2*half, 2*half+1 of the block (half the W reads of its
            // row block, half the update FMAs, half the panel work); both keep the full diagonal-block copy.
            double* s_R = s_W;
            double dg[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int row = 4 * bj + r, col = 4 * bj + c;
                    double v = (row == col) ? 1.0 : 0.0;
                    if (bj >= 0 && row <= P && col < P) v = A[(size_t)row * HS + col];
                    dg[r][c] = v;
                }
            double a2[2][4];
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                for (int c = 0; c < 4; ++c) a2[rr][c] = half ? a4[2 + rr][c] : a4[rr][c];
            if (threadIdx.x < 2) s_failf[threadIdx.x] = 0;
            fail = false;
            __syncthreads();
            for (int kb = 0; kb < NB; ++kb) {
                if (kb > 0 && s_failf[(kb - 1) & 1]) { fail = true; break; }
                if (bj == kb) {
                    const double D00 = dg[0][0], D10 = dg[1][0];
                    double D11 = dg[1][1], D20 = dg[2][0], D21 = dg[2][1], D22 = dg[2][2], D30 = dg[3][0], D31 = dg[3][1], D32 = dg[3][2], D33 = dg[3][3];
                    const bool real1 = 4 * kb + 1 < P, real2 = 4 * kb + 2 < P, real3 = 4 * kb + 3 < P;
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
                    if (bad) s_failf[kb & 1] = 1;
                    d2v* Wo = (d2v*)(Lblk + ((size_t)kb * NB + bi) * 18) + 4 * half;
#pragma unroll
                    for (int rr = 0; rr < 2; ++rr) {
                        const double w0 = a2[rr][0];
                        const double w1 = fma(-w0, l10, a2[rr][1]);
                        const double w2 = fma(-w1, l21, fma(-w0, l20, a2[rr][2]));
                        const double w3 = fma(-w2, l32, fma(-w1, l31, fma(-w0, l30, a2[rr][3])));
                        Wo[2 * rr] = (d2v){w0, w1}; Wo[2 * rr + 1] = (d2v){w2, w3};
                    }
                    if (bi == kb && half == 0) { d2v* Ro = (d2v*)(s_R + 4 * kb); Ro[0] = (d2v){r0, r1}; Ro[1] = (d2v){r2, r3}; }
                }
                __syncthreads();
                if (bj > kb) {
                    const d2v* Wi = (const d2v*)(Lblk + ((size_t)kb * NB + bi) * 18) + 4 * half;
                    const d2v* Wj = (const d2v*)(Lblk + ((size_t)kb * NB + bj) * 18);
                    const d2v* Rq = (const d2v*)(s_R + 4 * kb);
                    d2v wv[2][2], wj[4][2], lv[4][2];
#pragma unroll
                    for (int rr = 0; rr < 2; ++rr) { wv[rr][0] = Wi[2 * rr]; wv[rr][1] = Wi[2 * rr + 1]; }
#pragma unroll
                    for (int r = 0; r < 4; ++r) { wj[r][0] = Wj[2 * r]; wj[r][1] = Wj[2 * r + 1]; }
                    const d2v ra = Rq[0], rb = Rq[1];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { lv[r][0].x = wj[r][0].x * ra.x; lv[r][0].y = wj[r][0].y * ra.y; lv[r][1].x = wj[r][1].x * rb.x; lv[r][1].y = wj[r][1].y * rb.y; }
#pragma unroll
                    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                        for (int cc = 0; cc < 4; ++cc) {
                            double v = a2[rr][cc];
                            v = fma(-wv[rr][0].x, lv[cc][0].x, v);
                            v = fma(-wv[rr][0].y, lv[cc][0].y, v);
                            v = fma(-wv[rr][1].x, lv[cc][1].x, v);
                            v = fma(-wv[rr][1].y, lv[cc][1].y, v);
                            a2[rr][cc] = v;
                        }
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int cc = 0; cc <= r; ++cc) {
                            double v = dg[r][cc];
                            v = fma(-wj[r][0].x, lv[cc][0].x, v);
                            v = fma(-wj[r][0].y, lv[cc][0].y, v);
                            v = fma(-wj[r][1].x, lv[cc][1].x, v);
                            v = fma(-wj[r][1].y, lv[cc][1].y, v);
                            dg[r][cc] = v;
                        }
                }
            }
            if (!fail && s_failf[(NB - 1) & 1]) fail = true;
        } else if constexpr (VAR == 9) {
            // V9: every lane also keeps the lower triangle of its column's diagonal block and applies every rank-4 update
            // to it; the panel column factors from registers: no diagonal publish, ONE barrier per round.
            double* s_R = s_W;
            double dg[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int row = 4 * bj + r, col = 4 * bj + c;
                    double v = (row == col) ? 1.0 : 0.0;
                    if (bj >= 0 && row <= P && col < P) v = A[(size_t)row * HS + col];
                    dg[r][c] = v;
                }
            if (t < 2) s_failf[t] = 0;
            fail = false;
            __syncthreads();
            for (int kb = 0; kb < NB; ++kb) {
                if (kb > 0 && s_failf[(kb - 1) & 1]) { fail = true; break; }
                if (bj == kb) {
                    const double D00 = dg[0][0], D10 = dg[1][0];
                    double D11 = dg[1][1], D20 = dg[2][0], D21 = dg[2][1], D22 = dg[2][2], D30 = dg[3][0], D31 = dg[3][1], D32 = dg[3][2], D33 = dg[3][3];
                    const bool real1 = 4 * kb + 1 < P, real2 = 4 * kb + 2 < P, real3 = 4 * kb + 3 < P;
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
                    if (bad) s_failf[kb & 1] = 1;
                    d2v* Wo = (d2v*)(Lblk + ((size_t)kb * NB + bi) * 18);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double w0 = a4[r][0];
                        const double w1 = fma(-w0, l10, a4[r][1]);
                        const double w2 = fma(-w1, l21, fma(-w0, l20, a4[r][2]));
                        const double w3 = fma(-w2, l32, fma(-w1, l31, fma(-w0, l30, a4[r][3])));
                        Wo[2 * r] = (d2v){w0, w1}; Wo[2 * r + 1] = (d2v){w2, w3};
                    }
                    if (bi == kb) { d2v* Ro = (d2v*)(s_R + 4 * kb); Ro[0] = (d2v){r0, r1}; Ro[1] = (d2v){r2, r3}; }
                }
                __syncthreads();
                if (bj > kb) {
                    const d2v* Wi = (const d2v*)(Lblk + ((size_t)kb * NB + bi) * 18);
                    const d2v* Wj = (const d2v*)(Lblk + ((size_t)kb * NB + bj) * 18);
                    const d2v* Rq = (const d2v*)(s_R + 4 * kb);
                    d2v wv[4][2], wj[4][2], lv[4][2];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { wv[r][0] = Wi[2 * r]; wv[r][1] = Wi[2 * r + 1]; wj[r][0] = Wj[2 * r]; wj[r][1] = Wj[2 * r + 1]; }
                    const d2v ra = Rq[0], rb = Rq[1];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { lv[r][0].x = wj[r][0].x * ra.x; lv[r][0].y = wj[r][0].y * ra.y; lv[r][1].x = wj[r][1].x * rb.x; lv[r][1].y = wj[r][1].y * rb.y; }
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
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int cc = 0; cc <= r; ++cc) {
                            double v = dg[r][cc];
                            v = fma(-wj[r][0].x, lv[cc][0].x, v);
                            v = fma(-wj[r][0].y, lv[cc][0].y, v);
                            v = fma(-wj[r][1].x, lv[cc][1].x, v);
                            v = fma(-wj[r][1].y, lv[cc][1].y, v);
                            dg[r][cc] = v;
                        }
                }
            }
            if (!fail && s_failf[(NB - 1) & 1]) fail = true;
        } else if constexpr (VAR == 11) {
            // V11 = V9 with the diagonal factorisation hoisted: every lane factors its diagonal copy right after updating it (same
            // straight-line block as the update of its own block, so the dependent rcp chain hides behind those FMAs).
            // V9: every lane also keeps the lower triangle of its column's diagonal block and applies every rank-4 update
            // to it; the panel column factors from registers: no diagonal publish, ONE barrier per round.
            double* s_R = s_W;
            double dg[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int row = 4 * bj + r, col = 4 * bj + c;
                    double v = (row == col) ? 1.0 : 0.0;
                    if (bj >= 0 && row <= P && col < P) v = A[(size_t)row * HS + col];
                    dg[r][c] = v;
                }
            double l10 = 0, l20 = 0, l30 = 0, l21 = 0, l31 = 0, l32 = 0, r0 = 0, r1 = 0, r2 = 0, r3 = 0, P0 = 1, P1 = 1, P2 = 1, P3 = 1;
#define FACTOR_DG() do { \
                const double D00 = dg[0][0], D10 = dg[1][0]; \
                double D11 = dg[1][1], D20 = dg[2][0], D21 = dg[2][1], D22 = dg[2][2], D30 = dg[3][0], D31 = dg[3][1], D32 = dg[3][2], D33 = dg[3][3]; \
                P0 = D00; r0 = fast_rcp(D00); l10 = D10 * r0; l20 = D20 * r0; l30 = D30 * r0; \
                D11 = fma(-l10, D10, D11); D21 = fma(-l20, D10, D21); D31 = fma(-l30, D10, D31); \
                D22 = fma(-l20, D20, D22); D32 = fma(-l30, D20, D32); D33 = fma(-l30, D30, D33); \
                P1 = D11; r1 = fast_rcp(D11); l21 = D21 * r1; l31 = D31 * r1; \
                D22 = fma(-l21, D21, D22); D32 = fma(-l31, D21, D32); D33 = fma(-l31, D31, D33); \
                P2 = D22; r2 = fast_rcp(D22); l32 = D32 * r2; D33 = fma(-l32, D32, D33); \
                P3 = D33; r3 = fast_rcp(D33); } while (0)
            FACTOR_DG();
            if (t < 2) s_failf[t] = 0;
            fail = false;
            __syncthreads();
            for (int kb = 0; kb < NB; ++kb) {
                if (kb > 0 && s_failf[(kb - 1) & 1]) { fail = true; break; }
                if (bj == kb) {
                    const bool real1 = 4 * kb + 1 < P, real2 = 4 * kb + 2 < P, real3 = 4 * kb + 3 < P;
                    const int bad = (int)!(P0 > 0.0) | ((int)real1 & (int)!(P1 > 0.0)) | ((int)real2 & (int)!(P2 > 0.0)) | ((int)real3 & (int)!(P3 > 0.0));
                    if (bad) s_failf[kb & 1] = 1;
                    d2v* Wo = (d2v*)(Lblk + ((size_t)kb * NB + bi) * 18);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double w0 = a4[r][0];
                        const double w1 = fma(-w0, l10, a4[r][1]);
                        const double w2 = fma(-w1, l21, fma(-w0, l20, a4[r][2]));
                        const double w3 = fma(-w2, l32, fma(-w1, l31, fma(-w0, l30, a4[r][3])));
                        Wo[2 * r] = (d2v){w0, w1}; Wo[2 * r + 1] = (d2v){w2, w3};
                    }
                    if (bi == kb) { d2v* Ro = (d2v*)(s_R + 4 * kb); Ro[0] = (d2v){r0, r1}; Ro[1] = (d2v){r2, r3}; }
                }
                __syncthreads();
                if (bj > kb) {
                    const d2v* Wi = (const d2v*)(Lblk + ((size_t)kb * NB + bi) * 18);
                    const d2v* Wj = (const d2v*)(Lblk + ((size_t)kb * NB + bj) * 18);
                    const d2v* Rq = (const d2v*)(s_R + 4 * kb);
                    d2v wv[4][2], wj[4][2], lv[4][2];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { wv[r][0] = Wi[2 * r]; wv[r][1] = Wi[2 * r + 1]; wj[r][0] = Wj[2 * r]; wj[r][1] = Wj[2 * r + 1]; }
                    const d2v ra = Rq[0], rb = Rq[1];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { lv[r][0].x = wj[r][0].x * ra.x; lv[r][0].y = wj[r][0].y * ra.y; lv[r][1].x = wj[r][1].x * rb.x; lv[r][1].y = wj[r][1].y * rb.y; }
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
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int cc = 0; cc <= r; ++cc) {
                            double v = dg[r][cc];
                            v = fma(-wj[r][0].x, lv[cc][0].x, v);
                            v = fma(-wj[r][0].y, lv[cc][0].y, v);
                            v = fma(-wj[r][1].x, lv[cc][1].x, v);
                            v = fma(-wj[r][1].y, lv[cc][1].y, v);
                            dg[r][cc] = v;
                        }
                    FACTOR_DG();
                }
            }
#undef FACTOR_DG
            if (!fail && s_failf[(NB - 1) & 1]) fail = true;
        } else if constexpr (VAR == 12) {
            // V12 = V11 with the column-block reads and the diagonal update / factorisation issued before the row-block reads and
            // the update of the own block.
            // V11 = V9 with the diagonal factorisation hoisted: every lane factors its diagonal copy right after updating it (same
            // straight-line block as the update of its own block, so the dependent rcp chain hides behind those FMAs).
            // V9: every lane also keeps the lower triangle of its column's diagonal block and applies every rank-4 update
            // to it; the panel column factors from registers: no diagonal publish, ONE barrier per round.
            double* s_R = s_W;
            double dg[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int row = 4 * bj + r, col = 4 * bj + c;
                    double v = (row == col) ? 1.0 : 0.0;
                    if (bj >= 0 && row <= P && col < P) v = A[(size_t)row * HS + col];
                    dg[r][c] = v;
                }
            double l10 = 0, l20 = 0, l30 = 0, l21 = 0, l31 = 0, l32 = 0, r0 = 0, r1 = 0, r2 = 0, r3 = 0, P0 = 1, P1 = 1, P2 = 1, P3 = 1;
#define FACTOR_DG() do { \
                const double D00 = dg[0][0], D10 = dg[1][0]; \
                double D11 = dg[1][1], D20 = dg[2][0], D21 = dg[2][1], D22 = dg[2][2], D30 = dg[3][0], D31 = dg[3][1], D32 = dg[3][2], D33 = dg[3][3]; \
                P0 = D00; r0 = fast_rcp(D00); l10 = D10 * r0; l20 = D20 * r0; l30 = D30 * r0; \
                D11 = fma(-l10, D10, D11); D21 = fma(-l20, D10, D21); D31 = fma(-l30, D10, D31); \
                D22 = fma(-l20, D20, D22); D32 = fma(-l30, D20, D32); D33 = fma(-l30, D30, D33); \
                P1 = D11; r1 = fast_rcp(D11); l21 = D21 * r1; l31 = D31 * r1; \
                D22 = fma(-l21, D21, D22); D32 = fma(-l31, D21, D32); D33 = fma(-l31, D31, D33); \
                P2 = D22; r2 = fast_rcp(D22); l32 = D32 * r2; D33 = fma(-l32, D32, D33); \
                P3 = D33; r3 = fast_rcp(D33); } while (0)
            FACTOR_DG();
            if (t < 2) s_failf[t] = 0;
            fail = false;
            __syncthreads();
            for (int kb = 0; kb < NB; ++kb) {
                if (kb > 0 && s_failf[(kb - 1) & 1]) { fail = true; break; }
                if (bj == kb) {
                    const bool real1 = 4 * kb + 1 < P, real2 = 4 * kb + 2 < P, real3 = 4 * kb + 3 < P;
                    const int bad = (int)!(P0 > 0.0) | ((int)real1 & (int)!(P1 > 0.0)) | ((int)real2 & (int)!(P2 > 0.0)) | ((int)real3 & (int)!(P3 > 0.0));
                    if (bad) s_failf[kb & 1] = 1;
                    d2v* Wo = (d2v*)(Lblk + ((size_t)kb * NB + bi) * 18);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double w0 = a4[r][0];
                        const double w1 = fma(-w0, l10, a4[r][1]);
                        const double w2 = fma(-w1, l21, fma(-w0, l20, a4[r][2]));
                        const double w3 = fma(-w2, l32, fma(-w1, l31, fma(-w0, l30, a4[r][3])));
                        Wo[2 * r] = (d2v){w0, w1}; Wo[2 * r + 1] = (d2v){w2, w3};
                    }
                    if (bi == kb) { d2v* Ro = (d2v*)(s_R + 4 * kb); Ro[0] = (d2v){r0, r1}; Ro[1] = (d2v){r2, r3}; }
                }
                __syncthreads();
                if (bj > kb) {
                    const d2v* Wi = (const d2v*)(Lblk + ((size_t)kb * NB + bi) * 18);
                    const d2v* Wj = (const d2v*)(Lblk + ((size_t)kb * NB + bj) * 18);
                    const d2v* Rq = (const d2v*)(s_R + 4 * kb);
                    d2v wv[4][2], wj[4][2], lv[4][2];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { wj[r][0] = Wj[2 * r]; wj[r][1] = Wj[2 * r + 1]; }
                    const d2v ra = Rq[0], rb = Rq[1];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { wv[r][0] = Wi[2 * r]; wv[r][1] = Wi[2 * r + 1]; }
#pragma unroll
                    for (int r = 0; r < 4; ++r) { lv[r][0].x = wj[r][0].x * ra.x; lv[r][0].y = wj[r][0].y * ra.y; lv[r][1].x = wj[r][1].x * rb.x; lv[r][1].y = wj[r][1].y * rb.y; }
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int cc = 0; cc <= r; ++cc) {
                            double v = dg[r][cc];
                            v = fma(-wj[r][0].x, lv[cc][0].x, v);
                            v = fma(-wj[r][0].y, lv[cc][0].y, v);
                            v = fma(-wj[r][1].x, lv[cc][1].x, v);
                            v = fma(-wj[r][1].y, lv[cc][1].y, v);
                            dg[r][cc] = v;
                        }
                    FACTOR_DG();
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
                }
            }
#undef FACTOR_DG
            if (!fail && s_failf[(NB - 1) & 1]) fail = true;
        } else if constexpr (VAR == 13) {
            // V13 = V12 with the update FMAs ordered k-major (16 independent FMAs per step).
            // V12 = V11 with the column-block reads and the diagonal update / factorisation issued before the row-block reads and
            // the update of the own block.
            // V11 = V9 with the diagonal factorisation hoisted: every lane factors its diagonal copy right after updating it (same
            // straight-line block as the update of its own block, so the dependent rcp chain hides behind those FMAs).
            // V9: every lane also keeps the lower triangle of its column's diagonal block and applies every rank-4 update
            // to it; the panel column factors from registers: no diagonal publish, ONE barrier per round.
            double* s_R = s_W;
            double dg[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int row = 4 * bj + r, col = 4 * bj + c;
                    double v = (row == col) ? 1.0 : 0.0;
                    if (bj >= 0 && row <= P && col < P) v = A[(size_t)row * HS + col];
                    dg[r][c] = v;
                }
            double l10 = 0, l20 = 0, l30 = 0, l21 = 0, l31 = 0, l32 = 0, r0 = 0, r1 = 0, r2 = 0, r3 = 0, P0 = 1, P1 = 1, P2 = 1, P3 = 1;
#define FACTOR_DG() do { \
                const double D00 = dg[0][0], D10 = dg[1][0]; \
                double D11 = dg[1][1], D20 = dg[2][0], D21 = dg[2][1], D22 = dg[2][2], D30 = dg[3][0], D31 = dg[3][1], D32 = dg[3][2], D33 = dg[3][3]; \
                P0 = D00; r0 = fast_rcp(D00); l10 = D10 * r0; l20 = D20 * r0; l30 = D30 * r0; \
                D11 = fma(-l10, D10, D11); D21 = fma(-l20, D10, D21); D31 = fma(-l30, D10, D31); \
                D22 = fma(-l20, D20, D22); D32 = fma(-l30, D20, D32); D33 = fma(-l30, D30, D33); \
                P1 = D11; r1 = fast_rcp(D11); l21 = D21 * r1; l31 = D31 * r1; \
                D22 = fma(-l21, D21, D22); D32 = fma(-l31, D21, D32); D33 = fma(-l31, D31, D33); \
                P2 = D22; r2 = fast_rcp(D22); l32 = D32 * r2; D33 = fma(-l32, D32, D33); \
                P3 = D33; r3 = fast_rcp(D33); } while (0)
            FACTOR_DG();
            if (t < 2) s_failf[t] = 0;
            fail = false;
            __syncthreads();
            for (int kb = 0; kb < NB; ++kb) {
                if (kb > 0 && s_failf[(kb - 1) & 1]) { fail = true; break; }
                if (bj == kb) {
                    const bool real1 = 4 * kb + 1 < P, real2 = 4 * kb + 2 < P, real3 = 4 * kb + 3 < P;
                    const int bad = (int)!(P0 > 0.0) | ((int)real1 & (int)!(P1 > 0.0)) | ((int)real2 & (int)!(P2 > 0.0)) | ((int)real3 & (int)!(P3 > 0.0));
                    if (bad) s_failf[kb & 1] = 1;
                    d2v* Wo = (d2v*)(Lblk + ((size_t)kb * NB + bi) * 18);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double w0 = a4[r][0];
                        const double w1 = fma(-w0, l10, a4[r][1]);
                        const double w2 = fma(-w1, l21, fma(-w0, l20, a4[r][2]));
                        const double w3 = fma(-w2, l32, fma(-w1, l31, fma(-w0, l30, a4[r][3])));
                        Wo[2 * r] = (d2v){w0, w1}; Wo[2 * r + 1] = (d2v){w2, w3};
                    }
                    if (bi == kb) { d2v* Ro = (d2v*)(s_R + 4 * kb); Ro[0] = (d2v){r0, r1}; Ro[1] = (d2v){r2, r3}; }
                }
                __syncthreads();
                if (bj > kb) {
                    const d2v* Wi = (const d2v*)(Lblk + ((size_t)kb * NB + bi) * 18);
                    const d2v* Wj = (const d2v*)(Lblk + ((size_t)kb * NB + bj) * 18);
                    const d2v* Rq = (const d2v*)(s_R + 4 * kb);
                    d2v wv[4][2], wj[4][2], lv[4][2];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { wj[r][0] = Wj[2 * r]; wj[r][1] = Wj[2 * r + 1]; }
                    const d2v ra = Rq[0], rb = Rq[1];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { wv[r][0] = Wi[2 * r]; wv[r][1] = Wi[2 * r + 1]; }
#pragma unroll
                    for (int r = 0; r < 4; ++r) { lv[r][0].x = wj[r][0].x * ra.x; lv[r][0].y = wj[r][0].y * ra.y; lv[r][1].x = wj[r][1].x * rb.x; lv[r][1].y = wj[r][1].y * rb.y; }
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int cc = 0; cc <= r; ++cc) {
                            double v = dg[r][cc];
                            v = fma(-wj[r][0].x, lv[cc][0].x, v);
                            v = fma(-wj[r][0].y, lv[cc][0].y, v);
                            v = fma(-wj[r][1].x, lv[cc][1].x, v);
                            v = fma(-wj[r][1].y, lv[cc][1].y, v);
                            dg[r][cc] = v;
                        }
                    FACTOR_DG();
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                            for (int cc = 0; cc < 4; ++cc) {
                                const double wq = (s4 == 0) ? wv[r][0].x : (s4 == 1) ? wv[r][0].y : (s4 == 2) ? wv[r][1].x : wv[r][1].y;
                                const double lq = (s4 == 0) ? lv[cc][0].x : (s4 == 1) ? lv[cc][0].y : (s4 == 2) ? lv[cc][1].x : lv[cc][1].y;
                                a4[r][cc] = fma(-wq, lq, a4[r][cc]);
                            }
                }
            }
#undef FACTOR_DG
            if (!fail && s_failf[(NB - 1) & 1]) fail = true;
        } else {
            // ---- c. register-blocked LDL^T, four pivots and ONE barrier per round (look-ahead on the panel column) ---------
            // Round kb, between two barriers:
            //  (1) every lane with bj >= kb applies the rank-4 update of pivot block kb-1 to its block, A -= W L^T, from
            //      the W of its row block and the L of its column block published in round kb-1 (16 LDS reads, 64 FMAs);
            //      the owner of diagonal block kb+1 then publishes it for the next round;
            //  (2) the lanes of panel column kb (bj == kb) rebuild the diagonal block themselves (the copy published one
            //      round earlier minus the same rank-4 update: identical operations in every lane, so all of them factor
            //      the same matrix), factor it (D = Ld diag(d) Ld^T), solve their own block W = A Ld^-T, L = W diag(d)^-1
            //      and publish W (double-buffered) and L.
            double* s_W2 = s_W;                                     // [2][NB][18]
            double* s_Dp = s_D;                                     // [2][18]
            if (t < 2) s_failf[t] = 0;
            if (bi == bj && bi >= 0 && bi < 2) {                    // diagonal blocks 0 and 1 have no update pending in their first round
                d2v* Do = (d2v*)(s_Dp + (size_t)bi * 18);
        #pragma unroll
                for (int r = 0; r < 4; ++r) { Do[2 * r] = (d2v){a4[r][0], a4[r][1]}; Do[2 * r + 1] = (d2v){a4[r][2], a4[r][3]}; }
            }
            fail = false;
            __syncthreads();
            for (int kb = 0; kb < NB; ++kb) {
                d2v lv[4][2];                                       // L(kb-1) of my column block, shared by (1) and (2)
                if (kb > 0) {
                    if (s_failf[(kb - 1) & 1]) { fail = true; break; }
                    if (bj >= kb && VAR != 4 && VAR != 2) {
                        if constexpr (VAR == 6) { for (int r = 0; r < 4; ++r) { lv[r][0] = (d2v){a4[r][0], a4[r][1]}; lv[r][1] = (d2v){a4[r][2], a4[r][3]}; } }
                        const d2v* Wi = (const d2v*)(s_W2 + ((size_t)((kb - 1) & 1) * NB + bi) * 18);
                        const d2v* Lj = (const d2v*)(Lblk + ((size_t)(kb - 1) * NB + bj) * 18);
                        d2v wv[4][2];
        #pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if constexpr (VAR == 6) { wv[r][0] = (d2v){a4[0][r], a4[1][r]}; wv[r][1] = (d2v){a4[2][r], a4[3][r]}; }
                            else { wv[r][0] = Wi[2 * r]; wv[r][1] = Wi[2 * r + 1]; lv[r][0] = Lj[2 * r]; lv[r][1] = Lj[2 * r + 1]; }
                        }
                        if constexpr (VAR == 5) {
        #pragma unroll
                            for (int r = 0; r < 4; ++r) { a4[r][0] += wv[r][0].x + lv[r][0].x; a4[r][1] += wv[r][0].y + lv[r][0].y; a4[r][2] += wv[r][1].x + lv[r][1].x; a4[r][3] += wv[r][1].y + lv[r][1].y; }
                        } else
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
                        if (bi == kb + 1 && bj == kb + 1) {         // publish the diagonal block of the next panel column
                            d2v* Do = (d2v*)(s_Dp + (size_t)((kb + 1) & 1) * 18);
        #pragma unroll
                            for (int r = 0; r < 4; ++r) { Do[2 * r] = (d2v){a4[r][0], a4[r][1]}; Do[2 * r + 1] = (d2v){a4[r][2], a4[r][3]}; }
                        }
                    }
                }
                if (bj == kb && VAR != 3 && VAR != 2 && VAR != 5 && VAR != 6) {
                    const d2v* Dq = (const d2v*)(s_Dp + (size_t)(kb & 1) * 18);
                    const d2v q0 = Dq[0], q2 = Dq[2], q4 = Dq[4], q5 = Dq[5], q6 = Dq[6], q7 = Dq[7];
                    double D00 = q0.x, D10 = q2.x, D11 = q2.y, D20 = q4.x, D21 = q4.y, D22 = q5.x, D30 = q6.x, D31 = q6.y, D32 = q7.x, D33 = q7.y;
                    if (kb > 0) {                                   // the rank-4 update of the diagonal block, lower triangle
                        const d2v* Wk = (const d2v*)(s_W2 + ((size_t)((kb - 1) & 1) * NB + kb) * 18);
                        d2v wk[4][2];
        #pragma unroll
                        for (int r = 0; r < 4; ++r) { wk[r][0] = Wk[2 * r]; wk[r][1] = Wk[2 * r + 1]; }
        #define DUPD(D, r, c) D = fma(-wk[r][1].y, lv[c][1].y, fma(-wk[r][1].x, lv[c][1].x, fma(-wk[r][0].y, lv[c][0].y, fma(-wk[r][0].x, lv[c][0].x, D))))
                        DUPD(D00, 0, 0); DUPD(D10, 1, 0); DUPD(D11, 1, 1); DUPD(D20, 2, 0); DUPD(D21, 2, 1); DUPD(D22, 2, 2);
                        DUPD(D30, 3, 0); DUPD(D31, 3, 1); DUPD(D32, 3, 2); DUPD(D33, 3, 3);
        #undef DUPD
                    }
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
                    if (bad) s_failf[kb & 1] = 1;
                    d2v* Wo = (d2v*)(s_W2 + ((size_t)(kb & 1) * NB + bi) * 18);
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
                }
                __syncthreads();
            }
            if (!fail && s_failf[(NB - 1) & 1]) fail = true;

        }
        __syncthreads();
        total += clock64() - c0;
    }
    if (threadIdx.x == 0) { cyc[0] = total; cyc[1] = fail; }
    for (int e = threadIdx.x; e < NB * NB * 18; e += blockDim.x) Lout[e] = Lblk[e];
}
template <int VAR> void run(const double* dA, double* dL, long long* dc, int P, int HS) {
    const int NB = HS / 4;
    const size_t lds = sizeof(double) * ((size_t)NB * NB * 18 + 2 * NB * 18 + 36 + 64);
    hipFuncSetAttribute((const void*)kern<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    const int reps = 20;
    long long h[2], best = 1ll << 60;
    for (int it = 0; it < 6; ++it) { hipLaunchKernelGGL(kern<VAR>, dim3(1), dim3(VAR == 10 ? 512 : 256), lds, 0, dA, dL, dc, P, HS, reps); hipMemcpy(h, dc, 16, hipMemcpyDeviceToHost); if (h[0] < best) best = h[0]; }
    h[0] = best;
    std::vector<double> L((size_t)NB * NB * 18); hipMemcpy(L.data(), dL, L.size() * 8, hipMemcpyDeviceToHost);
    double cs = 0; 
    for (int kb = 0; kb < NB; ++kb) for (int b = kb; b < NB; ++b) for (int e = 0; e < 16; ++e) cs += L[((size_t)kb * NB + b) * 18 + e] * (1 + (e % 3));
    printf("variant %d: %.0f clk per factorisation (%.0f per round), fail=%lld, checksum %.12g\n", VAR, h[0] / (double)reps, h[0] / (double)reps / NB, h[1], cs);
}
int main() {
    const int P = 85, HS = 88;
    std::vector<double> M((size_t)200 * P), A((size_t)HS * HS, 0.0);
    srand(1);
    for (auto& v : M) v = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < P; ++i) for (int j = 0; j < P; ++j) { double s = 0; for (int k = 0; k < 200; ++k) s += M[(size_t)k * P + i] * M[(size_t)k * P + j]; A[(size_t)i * HS + j] = s + (i == j ? 1.0 : 0.0); }
    for (int j = 0; j < P; ++j) A[(size_t)P * HS + j] = rand() / (double)RAND_MAX - 0.5;
    double *dA, *dL; long long* dc;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dL, (size_t)22 * 22 * 18 * 8); hipMalloc(&dc, 64);
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    run<0>(dA, dL, dc, P, HS); run<1>(dA, dL, dc, P, HS); run<2>(dA, dL, dc, P, HS); run<3>(dA, dL, dc, P, HS); run<5>(dA, dL, dc, P, HS); run<6>(dA, dL, dc, P, HS); run<7>(dA, dL, dc, P, HS); run<8>(dA, dL, dc, P, HS); run<9>(dA, dL, dc, P, HS); run<10>(dA, dL, dc, P, HS); run<11>(dA, dL, dc, P, HS); run<12>(dA, dL, dc, P, HS); run<13>(dA, dL, dc, P, HS);
    return 0;
}
