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
#define TPROBE(i) do { if (threadIdx.x == 0) fb.trace[(size_t)blockIdx.x * 64 + 40 + (i)] = (double)clock64(); } while (0)
#else
#define TPROBE(i) do {} while (0)
#endif

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
__global__ __launch_bounds__(256) void k_eval(DeviceModel dm, FrameBuffers fb) {
    const AvtDims d = dm.d;
    const int J = d.J, K = d.K, V = d.V, P = d.P, NT = d.NT, NPAIR = d.NPAIR;
    const int f = blockIdx.y, g = blockIdx.x, G = gridDim.x, t = threadIdx.x;
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
    int* s_aj = (int*)(s_aw + 64);                                // [16][4]

    const double* prep = fb.prep + ((size_t)f * 2 + try_slot) * d.prep_size;
    for (int e = t; e < d.prep_size; e += 256) s_prep[e] = prep[e];
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

    for (int b0 = lo; b0 < hi; b0 += AVT_EVAL_PTS) {
        const int idx = b0 + pi;
        const bool valid = idx < hi;
        const int m = valid ? matched[idx] : 0;
        __syncthreads();  // previous batch's MFMA reads are done (also covers the prep load on the first pass)
        // clear the tile, fetch this batch's model data
        for (int e = t; e < NT * 16 * RS; e += 256) s_Jt[e] = 0.0;
        for (int e = slot; e < NSH; e += 16) s_D[pi * NSH + e] = valid ? dm.shape_planes[(size_t)e * V + m] : 0.0;
        if (slot < 4) {
            s_aw[pi * 4 + slot] = valid ? dm.asg_w[(size_t)slot * V + m] : 0.0;
            s_aj[pi * 4 + slot] = valid ? dm.asg_j[(size_t)slot * V + m] : 0;
        }
        const int nanc = valid ? (int)dm.anc_n[m] : 0;
        const unsigned aword = (slot < nanc) ? (unsigned)dm.anc[(size_t)slot * V + m] : 0u;
        const double sc = valid ? mcnt[idx] : 0.0;
        __syncthreads();
        // shaped rest position, root-subtracted (CalcShape, :249-272)
        if (slot < 3) {
            double a = 0.0;
            for (int k = 0; k < K; ++k) a += s_D[pi * NSH + 3 * k + slot] * ww[k];
            s_xhat[pi * 3 + slot] = (a + s_D[pi * NSH + 3 * K + slot]) - off[slot];
        }
        __syncthreads();
        // x_k = R(-1,k)(x^ - J^_k) + t(-1,k) for the <=4 assigned joints (:508-514)
        if (slot < 4) {
            const int k = s_aj[pi * 4 + slot];
            const double* R = Rw + 9 * k;
            const double e0 = s_xhat[pi * 3] - Jh[3 * k], e1 = s_xhat[pi * 3 + 1] - Jh[3 * k + 1], e2 = s_xhat[pi * 3 + 2] - Jh[3 * k + 2];
            double* xk = s_xk + (pi * 4 + slot) * 3;
            xk[0] = (R[0] * e0 + R[1] * e1 + R[2] * e2) + oo[3 * k];
            xk[1] = (R[3] * e0 + R[4] * e1 + R[5] * e2) + oo[3 * k + 1];
            xk[2] = (R[6] * e0 + R[7] * e1 + R[8] * e2) + oo[3 * k + 2];
        }
        __syncthreads();
        const double* aw = s_aw + pi * 4;
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
                const double* src = Rw + 9 * dm.parent[j];
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
        // shape block (:568-580), residual column and the identity root-translation block (:476-481)
        for (int e = slot; e < 3 * K + 6; e += 16) {
            if (e < 3 * K) {
                const int r = e / K, k = e % K;
                double a = 0.0;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const double wt = aw[s4];
                    if (wt != 0.0) {
                        const int jj = s_aj[pi * 4 + s4];
                        const double* R = Rw + 9 * jj;
                        const double* Dk = s_D + pi * NSH + 3 * k;
                        a += ((R[3 * r] * Dk[0] + R[3 * r + 1] * Dk[1] + R[3 * r + 2] * Dk[2]) + Gm[(jj * 3 + r) * K + k]) * wt;
                    }
                }
                s_Jt[(size_t)(3 + 3 * J + k) * RS + pi * 3 + r] = sc * a;
            } else if (e < 3 * K + 3) {
                const int r = e - 3 * K;
                double xm = 0.0;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) xm += aw[s4] * xk[3 * s4 + r];
                const double db = valid ? mdbar[(size_t)r * V + idx] : 0.0;
                s_Jt[(size_t)P * RS + pi * 3 + r] = sc * (xm - db);
            } else {
                const int r = e - 3 * K - 3;
                s_Jt[(size_t)r * RS + pi * 3 + r] = sc;
            }
        }
        __syncthreads();
        // MFMA phase: 12 k-steps of 4 rows
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
    return sizeof(double) * ((size_t)d.prep_size + (size_t)d.NT * 16 * AVT_EVAL_RS + 16 * NSH + 48 + 192 + 64) + sizeof(int) * 64;
}

void launch_eval(avt_ctx* c, int nframes) {
    const AvtDims& d = c->dm.d;
    dim3 grid(c->fb.G, nframes);
    const size_t lds = eval_lds_bytes(d);
    if (d.NT <= 6) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_eval<6>), grid, dim3(256), lds, c->stream, c->dm, c->fb);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_eval<AVT_MAX_TILES>), grid, dim3(256), lds, c->stream, c->dm, c->fb);
}

int avt_eval_set_attributes() {
    hipError_t e = hipFuncSetAttribute((const void*)k_eval<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
    if (e != hipSuccess) return 1;
    e = hipFuncSetAttribute((const void*)k_eval<AVT_MAX_TILES>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
    return e != hipSuccess;
}
