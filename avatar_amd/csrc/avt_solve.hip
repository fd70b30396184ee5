// avt_solve.hip — k_eval of the Gauss-Newton / LM inner loop (k_reduce / k_solve live in avt_lm.hip):
//   k_eval   : residual + analytic Jacobian rows per matched model point (AvatarCostFunctorCache::updateData,
//              AvatarOptimizer.cpp:505-582) staged in LDS and contracted on the fp64 matrix cores
//              (v_mfma_f64_16x16x4_f64) into per-workgroup partial tiles of [J | r]^T W [J | r];
//   k_reduce : fixed-order reduction of the partial tiles (+ one extra workgroup evaluating the GMM pose prior,
//              GaussianMixture::residual, GaussianMixture.cpp:95-114);
//   k_solve  : prior assembly (AvatarOptimizer.cpp:647-726, :1457-1458), LM accept/reject, damped Cholesky
//              solve, quaternion retraction (FakeQuaternionParameterization::Plus, :123-143) and the skeleton
//              tables of the next trial point (PrepareForEvaluation, :283-325) — one workgroup per frame,
//              the whole inner loop runs without host synchronisation.
//
// Algebra used (exact, only the summation order differs from the reference's per-residual-block form):
// all residual blocks matched to model point m share one Jacobian block J_m (AvatarOptimizer.cpp:1445-1449),
// so with c_m = #matches and dbar_m their mean data point,
//   J^T J = sum_m c_m J_m^T J_m,  J^T r = sum_m c_m J_m^T (x_m - dbar_m),
//   sum_i |x_m - d_i|^2 = c_m |x_m - dbar_m|^2 + sum_i |d_i - dbar_m|^2   (2nd term: k_cost_const).
// Each matched point therefore contributes 3 rows sqrt(c_m) [J_m | x_m - dbar_m] to an augmented matrix
// A (3M x (P+1)); A^T A holds H, g and the data cost at once.
//
// World-frame form of the per-ancestor vector (avoids the (J+1)^2 relative-transform tables of :303-315):
//   v_j = sum_{k under j} a_k (R(j,k)(x^-J^_k) + t(j,k)) = Rw_j^T (sum_{k under j} a_k x_k - (sum a_k) o_j),
// where x_k = Rw_k (x^ - J^_k) + o_k is the point carried by assigned joint k, Rw_j = R(-1,j), o_j = t(-1,j).
#include "avt_device.h"

typedef double v4f64 __attribute__((ext_vector_type(4)));

#ifdef AVT_TIMING
#define TPROBE(i) do { if (threadIdx.x == 0) fb.trace[(size_t)(blockIdx.x + fb.f0) * 64 + 40 + (i)] = (double)clock64(); } while (0)
#else
#define TPROBE(i) do {} while (0)
#endif

// upper-triangular tile pairs of the 6x6 tile grid, in the order k_reduce / k_solve decode them
__device__ constexpr int PAIR6_TI[21] = {0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 5};
__device__ constexpr int PAIR6_TJ[21] = {0, 1, 2, 3, 4, 5, 1, 2, 3, 4, 5, 2, 3, 4, 5, 3, 4, 5, 4, 5, 5};

// one batch (48 rows) of the [J|r]^T [J|r] contraction for wave W: pairs W, W+4, W+8, ... ; A and B operands are the
// same kind of fragment (lane l: column tile*16 + (l&15), row k0 + (l>>4)), so 6 LDS reads feed up to 6 MFMAs.
template <int W, int MAXPW>
__device__ __forceinline__ void mfma_batch6(const double* __restrict__ s_Jt, int ln, v4f64 (&acc)[MAXPW]) {
    const double* base = s_Jt + (size_t)(ln & 15) * AVT_EVAL_RS + (ln >> 4);
#pragma unroll
    for (int k0 = 0; k0 < AVT_EVAL_ROWS; k0 += 4) {
        double fr[6];
#pragma unroll
        for (int ti = 0; ti < 6; ++ti) fr[ti] = base[(size_t)ti * 16 * AVT_EVAL_RS + k0];
#pragma unroll
        for (int i = 0; i < MAXPW; ++i) {
            const int p = W + 4 * i;
            if (p < 21) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(fr[PAIR6_TI[p < 21 ? p : 0]], fr[PAIR6_TJ[p < 21 ? p : 0]], acc[i], 0, 0, 0);
        }
    }
}

// =================================================================================================
// k_eval.  grid (G, nframes), block 256 = 4 waves.  Workgroup g of frame f owns a contiguous slice of the
// frame's matched model points and walks it in batches of 16 points (48 rows of the augmented matrix).
//   VALU phase: thread (pi = t>>4, slot = t&15) — one lane per (point, ancestor slot) computes the 3x3 block
//     R(-1,parent j) * dRot(q_j, v_j) * localJacobian_j (AvatarOptimizer.cpp:524-566), scaled by sqrt(c_m), and
//     scatters it into the LDS tile, stored transposed [column][row] with row stride 50 doubles so that the
//     MFMA operand fetch (16 columns x 4 rows per instruction) is bank-conflict free for ds_read_b64.
//   MFMA phase: the NT(NT+1)/2 upper-triangular 16x16 output tiles are dealt round-robin to the 4 waves; each
//     wave accumulates its tiles over the 12 k-steps of the batch with v_mfma_f64_16x16x4_f64 (A and B operands
//     are the same kind of fragment: A[i][k] = Jt[ti*16+i][k], B[k][j] = Jt[tj*16+j][k]).
// =================================================================================================
template <int NT_MAX>
__global__ __launch_bounds__(256, 2) void k_eval(DeviceModel dm, FrameBuffers fb) {
    const AvtDims d = dm.d;
    const int J = d.J, K = d.K, V = d.V, P = d.P, NT = d.NT, NPAIR = d.NPAIR;
    const int f = blockIdx.y + fb.f0, g = blockIdx.x, G = gridDim.x, t = threadIdx.x;
    const AvtFrameCtl& ctl = fb.ctl[f];
    const int M = ctl.M;
    const int try_slot = 1 - ctl.cur_slot;
    constexpr int RS = AVT_EVAL_RS;
    constexpr int MAXPW = (NT_MAX * (NT_MAX + 1) / 2 + 3) / 4;
    const int NSH = 3 * (K + 1);

    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* s_prep = (double*)smem;                               // prep_size
    double* s_Jt = s_prep + d.prep_size;                          // [NT*16][RS]
    double* s_D = s_Jt + (size_t)NT * 16 * RS;                    // [16][NSH] shape planes of the batch points
    double* s_xhat = s_D + 16 * NSH;                              // [16][3]
    double* s_xk = s_xhat + 48;                                   // [16][4][3]
    double* s_aw = s_xk + 192;                                    // [16][4]
    double* s_T = s_aw + 64;                                      // [16][9]  blended rotation per point
    double* s_Gs = s_T + 144;                                     // [16][3K] blended shape table per point
    int* s_aj = (int*)(s_Gs + 16 * 3 * K);                        // [16][4]
    int* s_par = s_aj + 64;                                       // [J]

    const double* prep = fb.prep + ((size_t)f * 2 + try_slot) * d.prep_size;
    for (int e = t; e < d.prep_size; e += 256) s_prep[e] = prep[e];
    if (t < J) s_par[t] = dm.parent[t];
    const double* Rw = s_prep + prep_off_Rw(d);
    const double* oo = s_prep + prep_off_o(d);
    const double* Jh = s_prep + prep_off_Jh(d);
    const double* Gm = s_prep + prep_off_G(d);
    const double* qq = s_prep + prep_off_q(d);
    const double* ww = s_prep + prep_off_w(d);
    const double* off = s_prep + prep_off_off(d);

    // static deal of output tile pairs to waves: wave wv owns pairs wv, wv+4, wv+8, ...
    const int wv = wave_id(), ln = lane_id();
    int pr_ti[MAXPW], pr_tj[MAXPW];
#pragma unroll
    for (int i = 0; i < MAXPW; ++i) {
        int p = wv + 4 * i, ti = 0;
        if (p < NPAIR) {
            while (p >= NT - ti) { p -= NT - ti; ++ti; }
            pr_ti[i] = ti; pr_tj[i] = ti + p;
        } else { pr_ti[i] = -1; pr_tj[i] = -1; }
    }
    v4f64 acc[MAXPW];
#pragma unroll
    for (int i = 0; i < MAXPW; ++i) acc[i] = (v4f64){0.0, 0.0, 0.0, 0.0};

    const int lo = (int)(((long long)M * g) / G), hi = (int)(((long long)M * (g + 1)) / G);
    const int pi = t >> 4, slot = t & 15;
    const int* matched = fb.matched + (size_t)f * V;
    const double* mcnt = fb.mcnt + (size_t)f * V;
    const double* mdbar = fb.mdbar + (size_t)f * 3 * V;

#ifdef AVT_TIMING
    long long tacc[6] = {0, 0, 0, 0, 0, 0}; long long tlast = clock64();
#define EPROBE(k) do { const long long _n = clock64(); tacc[k] += _n - tlast; tlast = _n; } while (0)
#else
#define EPROBE(k) do {} while (0)
#endif
    // ---- software pipeline: the global loads of batch b+1 are issued while batch b is in its Jacobian / MFMA
    // phases; the matched-vertex id travels one batch further ahead (it addresses every other load).
    double pD[AVT_MAX_SHAPE + 1];      // slot < 3: component `slot` of the K key clouds + base cloud of my point
    double p_aw = 0.0, p_sc = 0.0, p_db = 0.0;
    int p_aj = 0, p_nanc = 0;
    unsigned p_aword = 0u;
    auto fetch = [&](int b0n, int mm) {
        const int idxn = b0n + pi;
        const bool vn = idxn < hi;
        const int m = vn ? mm : 0;
        if (slot < 3) {
#pragma unroll
            for (int k = 0; k <= AVT_MAX_SHAPE; ++k)
                if (k <= K) pD[k] = dm.shape_planes[((size_t)k * 3 + slot) * V + m];
            p_db = vn ? mdbar[(size_t)slot * V + idxn] : 0.0;
        }
        if (slot < 4) { p_aw = vn ? dm.asg_w[(size_t)slot * V + m] : 0.0; p_aj = dm.asg_j[(size_t)slot * V + m]; }
        p_nanc = vn ? (int)dm.anc_n[m] : 0;
        p_aword = (unsigned)dm.anc[(size_t)slot * V + m];
        p_sc = vn ? mcnt[idxn] : 0.0;
    };
    int m_nxt = (lo + AVT_EVAL_PTS + pi < hi) ? matched[lo + AVT_EVAL_PTS + pi] : 0;
    if (lo < hi) fetch(lo, (lo + pi < hi) ? matched[lo + pi] : 0);

    for (int b0 = lo; b0 < hi; b0 += AVT_EVAL_PTS) {
        const int idx = b0 + pi;
        const bool valid = idx < hi;
        // consume the prefetched registers of this batch
        const double sc = p_sc;
        const int nanc = p_nanc;
        const unsigned aword = (slot < nanc) ? p_aword : 0u;
        __syncthreads();  // previous batch's MFMA reads are done (also covers the prep load on the first pass)
        EPROBE(0);
        for (int e = t; e < NT * 16 * RS; e += 256) s_Jt[e] = 0.0;
        if (slot < 3) {
            // shaped rest position, root-subtracted (CalcShape, :249-272)
            double a = 0.0;
#pragma unroll
            for (int k = 0; k < AVT_MAX_SHAPE; ++k)
                if (k < K) { s_D[pi * NSH + 3 * k + slot] = pD[k]; a += pD[k] * ww[k]; }
            double bse = pD[0];
#pragma unroll
            for (int k = 1; k <= AVT_MAX_SHAPE; ++k)
                if (k == K) bse = pD[k];
            s_D[pi * NSH + 3 * K + slot] = bse;
            s_xhat[pi * 3 + slot] = (a + bse) - off[slot];
        }
        if (slot < 4) { s_aw[pi * 4 + slot] = p_aw; s_aj[pi * 4 + slot] = p_aj; }
        __syncthreads();
        EPROBE(1);
        const double* aw = s_aw + pi * 4;
        const int* aj = s_aj + pi * 4;
        // x_k = R(-1,k)(x^ - J^_k) + t(-1,k) for the <=4 assigned joints (:508-514); blended rotation T = sum a_k Rw_k
        // and blended shape table Gs = sum a_k G_k for the shape block (:568-580)
        if (slot < 4) {
            const int k = aj[slot];
            const double* R = Rw + 9 * k;
            const double e0 = s_xhat[pi * 3] - Jh[3 * k], e1 = s_xhat[pi * 3 + 1] - Jh[3 * k + 1], e2 = s_xhat[pi * 3 + 2] - Jh[3 * k + 2];
            double* xk = s_xk + (pi * 4 + slot) * 3;
            xk[0] = (R[0] * e0 + R[1] * e1 + R[2] * e2) + oo[3 * k];
            xk[1] = (R[3] * e0 + R[4] * e1 + R[5] * e2) + oo[3 * k + 1];
            xk[2] = (R[6] * e0 + R[7] * e1 + R[8] * e2) + oo[3 * k + 2];
        } else if (slot < 13) {
            const int e9 = slot - 4;
            s_T[pi * 9 + e9] = ((aw[0] * Rw[9 * aj[0] + e9] + aw[1] * Rw[9 * aj[1] + e9]) + aw[2] * Rw[9 * aj[2] + e9]) + aw[3] * Rw[9 * aj[3] + e9];
        }
        for (int e = slot; e < 3 * K; e += 16)
            s_Gs[pi * 3 * K + e] = ((aw[0] * Gm[aj[0] * 3 * K + e] + aw[1] * Gm[aj[1] * 3 * K + e]) + aw[2] * Gm[aj[2] * 3 * K + e]) + aw[3] * Gm[aj[3] * 3 * K + e];
        __syncthreads();
        EPROBE(2);
        // issue the next batch's loads now: they land while this batch does its Jacobian and MFMA phases
        const double db = p_db;
        {
            const int b0n = b0 + AVT_EVAL_PTS;
            const int mm = m_nxt;
            m_nxt = (b0n + AVT_EVAL_PTS + pi < hi) ? matched[b0n + AVT_EVAL_PTS + pi] : 0;
            if (b0n < hi) fetch(b0n, mm);
        }
        EPROBE(3);
        const double* xk = s_xk + pi * 12;
        if (slot < nanc) {
            const int j = aword & 0xff;
            const unsigned mask = aword >> 8;
            double X0 = 0.0, X1 = 0.0, X2 = 0.0, cj = 0.0;
#pragma unroll
            for (int a = 0; a < 4; ++a)
                if (mask & (1u << a)) {
                    X0 += aw[a] * xk[3 * a]; X1 += aw[a] * xk[3 * a + 1]; X2 += aw[a] * xk[3 * a + 2];
                    cj += aw[a];
                }
            const double l0 = X0 - cj * oo[3 * j], l1 = X1 - cj * oo[3 * j + 1], l2 = X2 - cj * oo[3 * j + 2];
            const double* Rj = Rw + 9 * j;
            const double v0 = Rj[0] * l0 + Rj[3] * l1 + Rj[6] * l2;   // Rw_j^T * l
            const double v1 = Rj[1] * l0 + Rj[4] * l1 + Rj[7] * l2;
            const double v2 = Rj[2] * l0 + Rj[5] * l1 + Rj[8] * l2;
            const double* q = qq + 4 * j;
            const double u0 = q[0] * 2, u1 = q[1] * 2, u2 = q[2] * 2, w2 = q[3] * 2;
            // quaternion-vector rotation pseudo-Jacobian d(q v q*)/d(x,y,z,w)  (:541-558)
            double dR[3][4];
            dR[0][0] = u1 * v1 + v2 * u2;
            dR[0][1] = w2 * v2 + u0 * v1 - 2 * u1 * v0;
            dR[0][2] = -w2 * v1 - 2 * v0 * u2 + u0 * v2;
            dR[0][3] = u1 * v2 - v1 * u2;
            dR[1][0] = -w2 * v2 - 2 * u0 * v1 + v0 * u1;
            dR[1][1] = v2 * u2 + u0 * v0;
            dR[1][2] = w2 * v0 + u1 * v2 - 2 * v1 * u2;
            dR[1][3] = v0 * u2 - u0 * v2;
            dR[2][0] = w2 * v1 + v0 * u2 - 2 * u0 * v2;
            dR[2][1] = -w2 * v0 - 2 * u1 * v2 + v1 * u2;
            dR[2][2] = u0 * v0 + v1 * u1;
            dR[2][3] = u0 * v1 - v0 * u1;
            // local parameterisation Jacobian at 0 (:293-299), rows (x,y,z,w) x 3
            const double L[4][3] = {{q[3], q[2], -q[1]}, {-q[2], q[3], q[0]}, {q[1], -q[0], q[3]}, {-q[0], -q[1], -q[2]}};
            double Rp[9];
            if (j == 0) { Rp[0] = 1; Rp[1] = 0; Rp[2] = 0; Rp[3] = 0; Rp[4] = 1; Rp[5] = 0; Rp[6] = 0; Rp[7] = 0; Rp[8] = 1; }
            else {
                const double* src = Rw + 9 * s_par[j];
#pragma unroll
                for (int e = 0; e < 9; ++e) Rp[e] = src[e];
            }
            double tmp[3][4];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) tmp[r][c] = Rp[3 * r] * dR[0][c] + Rp[3 * r + 1] * dR[1][c] + Rp[3 * r + 2] * dR[2][c];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const double bl = tmp[r][0] * L[0][c] + tmp[r][1] * L[1][c] + tmp[r][2] * L[2][c] + tmp[r][3] * L[3][c];
                    s_Jt[(size_t)(3 + 3 * j + c) * RS + pi * 3 + r] = sc * bl;
                }
        }
        // shape block (:568-580): (sum a_k Rw_k) D + sum a_k G_k
        for (int e = slot; e < 3 * K; e += 16) {
            const int r = e / K, k = e - r * K;
            const double* Tr = s_T + pi * 9 + 3 * r;
            const double* Dk = s_D + pi * NSH + 3 * k;
            const double a = (Tr[0] * Dk[0] + Tr[1] * Dk[1] + Tr[2] * Dk[2]) + s_Gs[pi * 3 * K + e];
            s_Jt[(size_t)(3 + 3 * J + k) * RS + pi * 3 + r] = sc * a;
        }
        if (slot < 3) {          // residual column: sqrt(c) (x_m - dbar_m)
            double xm = 0.0;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) xm += aw[s4] * xk[3 * s4 + slot];
            s_Jt[(size_t)P * RS + pi * 3 + slot] = sc * (xm - db);
        } else if (slot < 6) {   // identity root-translation block (:476-481)
            const int r = slot - 3;
            s_Jt[(size_t)r * RS + pi * 3 + r] = sc;
        }
        __syncthreads();
        EPROBE(4);
        // MFMA phase: 12 k-steps of 4 rows
        if constexpr (NT_MAX == 6) {
            // SMPL shape (P+1 <= 96): straight-line code per wave, fragments fetched once per k-step and shared by
            // the wave's tile pairs; no exec-mask branches between the matrix instructions.
            switch (wv) {
                case 0: mfma_batch6<0, MAXPW>(s_Jt, ln, acc); break;
                case 1: mfma_batch6<1, MAXPW>(s_Jt, ln, acc); break;
                case 2: mfma_batch6<2, MAXPW>(s_Jt, ln, acc); break;
                default: mfma_batch6<3, MAXPW>(s_Jt, ln, acc); break;
            }
        } else {
#pragma unroll 1
            for (int k0 = 0; k0 < AVT_EVAL_ROWS; k0 += 4) {
                const int rowoff = k0 + (ln >> 4);
#pragma unroll
                for (int i = 0; i < MAXPW; ++i) {
                    if (pr_ti[i] >= 0) {
                        const double a = s_Jt[(size_t)(pr_ti[i] * 16 + (ln & 15)) * RS + rowoff];
                        const double bq = s_Jt[(size_t)(pr_tj[i] * 16 + (ln & 15)) * RS + rowoff];
                        acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bq, acc[i], 0, 0, 0);
                    }
                }
            }
        }
    }
#ifdef AVT_TIMING
    EPROBE(5);
    if (t == 0 && f == 0 && g == 0) for (int k = 0; k < 6; ++k) fb.trace[48 + k] = (double)tacc[k];
#endif
    // partial tiles out: element (row = (ln>>4) + 4*reg, col = ln&15) of pair p at [p][reg*64 + ln]
    double* part = fb.partial + (((size_t)f * fb.G + g) * NPAIR) * 256;
#pragma unroll
    for (int i = 0; i < MAXPW; ++i) {
        const int p = wv + 4 * i;
        if (p < NPAIR) {
#pragma unroll
            for (int r = 0; r < 4; ++r) part[(size_t)p * 256 + r * 64 + ln] = acc[i][r];
        }
    }
}

static size_t eval_lds_bytes(const AvtDims& d) {
    const int NSH = 3 * (d.K + 1);
    return sizeof(double) * ((size_t)d.prep_size + (size_t)d.NT * 16 * AVT_EVAL_RS + 16 * NSH + 48 + 192 + 64 + 144 + 16 * 3 * d.K) +
           sizeof(int) * (64 + AVT_MAX_JOINTS);
}

void launch_eval(avt_ctx* c, int nframes) {
    const AvtDims& d = c->dm.d;
    dim3 grid(c->fb.G, nframes);
    const size_t lds = eval_lds_bytes(d);
    if (d.NT == 6) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_eval<6>), grid, dim3(256), lds, c->cur_stream, c->dm, c->fb);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_eval<AVT_MAX_TILES>), grid, dim3(256), lds, c->cur_stream, c->dm, c->fb);
}

void avt_eval_report_occupancy(const AvtDims& d) {
    int nb = -1;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_eval<6>, 256, eval_lds_bytes(d));
    fprintf(stderr, "[avt] k_eval<6>: dynamic LDS %zu B, occupancy query -> %d blocks/CU (%s)\n", eval_lds_bytes(d), nb, hipGetErrorString(e));
}

int avt_eval_set_attributes() {
    // k_eval<6> needs < 64 KB of dynamic LDS: leave its attribute alone (raising the cap costs residency);
    // the generic shape may need more.
    return hipFuncSetAttribute((const void*)k_eval<AVT_MAX_TILES>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess;
}
