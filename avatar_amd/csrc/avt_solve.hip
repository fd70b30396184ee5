// avt_solve.hip — hand-written gfx950 kernels of the Gauss-Newton / LM inner loop:
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

// -------------------------------------------------------------------------------------------------
// skeleton tables of a state x=(p,q,w) -> prep block in global memory.  Called by all 256 threads.
// LDS scratch: s_rot[9J], s_Rw[9J], s_o[3J], s_jp[3J], s_H[J*3K], s_par[J]
// -------------------------------------------------------------------------------------------------
struct PrepScratch {
    double rot[AVT_MAX_JOINTS * 9], Rw[AVT_MAX_JOINTS * 9], o[AVT_MAX_JOINTS * 3], jp[AVT_MAX_JOINTS * 3];
    double H[AVT_MAX_JOINTS * 3 * AVT_MAX_SHAPE];
    double p[3];
    int parent[AVT_MAX_JOINTS];
};

__device__ void compute_prep(const DeviceModel& dm, const double* __restrict__ x, double* __restrict__ prep, PrepScratch& s) {
    const AvtDims d = dm.d;
    const int J = d.J, K = d.K, t = threadIdx.x;
    const double* q = x + 3;
    const double* w = x + 3 + 4 * J;
    if (t < J) { s.parent[t] = dm.parent[t]; quat_to_rot(q + 4 * t, s.rot + 9 * t); }
    if (t < 3) s.p[t] = x[t];
    // CalcShape (AvatarOptimizer.cpp:249-281): jointPosInit = base + jointShapeReg*w
    if (t < 3 * J) {
        double a = 0.0;
        for (int k = 0; k < K; ++k) a += dm.jsr[(size_t)t * K + k] * w[k];
        s.jp[t] = dm.jsr_base[t] + a;
    }
    __syncthreads();
    fk_chain(J, s.parent, s.rot, s.jp, s.p, s.Rw, s.o);
    // H[j] = R(-1,parent j) * Sp[j] + H[parent j]   (:318-324); thread e owns entry (r,k) of every joint
    if (t < 3 * K) {
        const int r = t / K, k = t % K;
        s.H[t] = 0.0;
        for (int j = 1; j < J; ++j) {
            const int pa = s.parent[j];
            const double* Rp = s.Rw + 9 * pa;
            const double* Sp = dm.Sp + (size_t)j * 3 * K;
            s.H[j * 3 * K + t] = (Rp[3 * r] * Sp[k] + Rp[3 * r + 1] * Sp[K + k] + Rp[3 * r + 2] * Sp[2 * K + k]) + s.H[pa * 3 * K + t];
        }
    }
    __syncthreads();
    const double off0 = s.jp[0], off1 = s.jp[1], off2 = s.jp[2];
    for (int e = t; e < 9 * J; e += 256) prep[prep_off_Rw(d) + e] = s.Rw[e];
    for (int e = t; e < 3 * J; e += 256) {
        prep[prep_off_o(d) + e] = s.o[e];
        const int c = e % 3;
        prep[prep_off_Jh(d) + e] = s.jp[e] - (c == 0 ? off0 : (c == 1 ? off1 : off2));   // root at origin (:270-272)
    }
    for (int e = t; e < 3 * J * K; e += 256) {  // G[j] = H[j] - Rw[j]*S[j]
        const int j = e / (3 * K), r = (e / K) % 3, k = e % K;
        const double* Rj = s.Rw + 9 * j;
        const double* S = dm.S + (size_t)j * 3 * K;
        prep[prep_off_G(d) + e] = s.H[e] - (Rj[3 * r] * S[k] + Rj[3 * r + 1] * S[K + k] + Rj[3 * r + 2] * S[2 * K + k]);
    }
    for (int e = t; e < 4 * J; e += 256) prep[prep_off_q(d) + e] = q[e];
    if (t < K) prep[prep_off_w(d) + t] = w[t];
    if (t < 3) prep[prep_off_off(d) + t] = (t == 0 ? off0 : (t == 1 ? off1 : off2));
    __syncthreads();
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

// =================================================================================================
// k_reduce.  grid (NPAIR + 1, nframes), block 256.
//   blocks 0..NPAIR-1: tiles[f][pair][e] = sum_g partial[f][g][pair][e], g ascending (deterministic).
//   block NPAIR: GMM pose prior at the trial point: smplParams (AvatarOptimizer.cpp:664-669), the best
//     component (GaussianMixture.cpp:95-114) and precision*(x-mean) for the gradient.  Output, per frame:
//     prior[0] = ||rho||^2 - consts_log of the chosen component, prior[1] = component, prior[2..2+n) = Prec*(x-mu).
// =================================================================================================
__global__ __launch_bounds__(256) void k_reduce(DeviceModel dm, FrameBuffers fb, double* __restrict__ prior_out) {
    const AvtDims d = dm.d;
    const int f = blockIdx.y, t = threadIdx.x, NPAIR = d.NPAIR;
    if ((int)blockIdx.x < NPAIR) {
        const int p = blockIdx.x;
        const double* part = fb.partial + ((size_t)f * fb.G * NPAIR + p) * 256 + t;
        double a = 0.0;
        for (int g = 0; g < fb.G; ++g) a += part[(size_t)g * NPAIR * 256];
        fb.tiles[((size_t)f * NPAIR + p) * 256 + t] = a;
        return;
    }
    const int n = d.ndims, C = d.ncomps, J = d.J;
    double* po = prior_out + (size_t)f * (2 + AVT_MAX_JOINTS * 3);
    if (C <= 0) { if (t == 0) { po[0] = 0.0; po[1] = -1.0; } return; }
    __shared__ double s_x[AVT_MAX_JOINTS * 3], s_q[AVT_MAX_JOINTS * 3 * 8], s_pr[8];
    const int try_slot = 1 - fb.ctl[f].cur_slot;
    const double* x = fb.x + ((size_t)f * 2 + try_slot) * d.xsize;
    if (t < J - 1) {  // Eigen AngleAxis(Quaternion): angle in [0,pi], axis sign follows w
        const double* q = x + 3 + 4 * (t + 1);
        double nrm = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
        if (nrm < 2.220446049250313e-16) {
            const double mx = fmax(fabs(q[0]), fmax(fabs(q[1]), fabs(q[2])));
            if (mx > 0.0) { const double a = q[0] / mx, b = q[1] / mx, c = q[2] / mx; nrm = mx * sqrt(a * a + b * b + c * c); }
            else nrm = 0.0;
        }
        double ang = 0.0, ax0 = 1.0, ax1 = 0.0, ax2 = 0.0;
        if (nrm != 0.0) {
            ang = 2.0 * atan2(nrm, fabs(q[3]));
            if (q[3] < 0) nrm = -nrm;
            ax0 = q[0] / nrm; ax1 = q[1] / nrm; ax2 = q[2] / nrm;
        }
        s_x[3 * t] = ax0 * ang; s_x[3 * t + 1] = ax1 * ang; s_x[3 * t + 2] = ax2 * ang;
    }
    __syncthreads();
    // y_c = Prec_c (x - mu_c) for every component, one thread per (component, row)
    for (int e = t; e < C * n; e += 256) {
        const int c = e / n, a = e % n;
        const double* Pr = dm.prior_prec + ((size_t)c * n + a) * n;
        const double* mu = dm.prior_mean + (size_t)c * n;
        double s = 0.0;
        for (int b = 0; b < n; ++b) s += Pr[b] * (s_x[b] - mu[b]);
        s_q[e] = s;
    }
    __syncthreads();
    if (t < C) {  // ||rho||^2 = 1/2 d^T Prec d  (rho = L^T d sqrt(1/2), Prec = L L^T)
        const double* mu = dm.prior_mean + (size_t)t * n;
        double s = 0.0;
        for (int a = 0; a < n; ++a) s += (s_x[a] - mu[a]) * s_q[t * n + a];
        s_pr[t] = 0.5 * s - dm.prior_clog[t];
    }
    __syncthreads();
    if (t == 0) {
        double best = 1.7976931348623157e308;
        int bc = 0;
        for (int c = 0; c < C; ++c)
            if (s_pr[c] < best) { best = s_pr[c]; bc = c; }
        po[0] = best; po[1] = (double)bc;
        s_pr[0] = (double)bc;
    }
    __syncthreads();
    const int bc = (int)s_pr[0];
    if (t < n) po[2 + t] = s_q[bc * n + t];
}

// =================================================================================================
// k_solve.  grid (nframes), block 256.
// =================================================================================================
__global__ __launch_bounds__(256) void k_solve(DeviceModel dm, FrameBuffers fb, const double* __restrict__ prior_in, int mode,
                                               double lm_up, double lm_down, double lm_min, double lm_max) {
    const AvtDims d = dm.d;
    const int J = d.J, K = d.K, P = d.P, NT = d.NT;
    const int f = blockIdx.x, t = threadIdx.x;
    AvtFrameCtl& ctl = fb.ctl[f];
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int LD = P + 2 + ((P & 1) ? 0 : 1);              // odd leading dimension: conflict-free column walks
    double* A = (double*)smem;                             // [(P+1)][LD]: rows 0..P-1 = H, row P = -g (rhs)
    double* s_delta = A + (size_t)(P + 1) * LD;            // [P]
    double* s_invd = s_delta + AVT_MAX_TILES * 16;         // [P]
    PrepScratch* ps = (PrepScratch*)(s_invd + AVT_MAX_TILES * 16);
    __shared__ int s_flag;
    __shared__ double s_red[4];
    const int xs = d.xsize;
    double* x0 = fb.x + ((size_t)f * 2) * xs;

    if (mode == SOLVE_INIT) {
        // trial point := current point; sum the constant part of the data cost
        const int cur = ctl.cur_slot, tr = 1 - cur;
        for (int e = t; e < xs; e += 256) x0[(size_t)tr * xs + e] = x0[(size_t)cur * xs + e];
        double a = 0.0;
        if (t < 64) {
            for (int e = t; e < fb.const_blocks; e += 64) a += fb.const_part[(size_t)f * fb.const_blocks + e];
            a = wave_sum(a);
        }
        __syncthreads();
        if (t == 0) { ctl.cost_const = 0.5 * a; ctl.try_valid = 1; }
        compute_prep(dm, x0 + (size_t)tr * xs, fb.prep + ((size_t)f * 2 + tr) * d.prep_size, *ps);
        return;
    }

    const int cur0 = ctl.cur_slot, try0 = 1 - cur0;
    const double* xt = x0 + (size_t)try0 * xs;
    // ---- a. finalise H, g, cost of the trial point: data tiles + priors ------------------------
    // unpack the upper-triangular tile pairs into the full symmetric matrix
    {
        int p = 0;
        for (int ti = 0; ti < NT; ++ti)
            for (int tj = ti; tj < NT; ++tj) {
                const double v = fb.tiles[((size_t)f * d.NPAIR + p) * 256 + t];
                const int r = ti * 16 + ((t >> 4) & 3) + 4 * (t >> 6), c = tj * 16 + (t & 15);
                if (r <= P && c <= P) {
                    if (r < P && c < P) { A[(size_t)r * LD + c] = v; A[(size_t)c * LD + r] = v; }
                    else if (c == P && r < P) A[(size_t)P * LD + r] = v;        // g (stored positive for now)
                    else if (r == P && c == P) A[(size_t)P * LD + P] = v;       // sum c_m |x_m - dbar_m|^2
                }
                ++p;
            }
    }
    __syncthreads();
    const double sbp = ctl.sbp, sbs = ctl.sbs;
    const double* pri = prior_in + (size_t)f * (2 + AVT_MAX_JOINTS * 3);
    double cost = 0.5 * A[(size_t)P * LD + P] + ctl.cost_const;
    int comp = -1;
    if (sbp > 0.0 && d.ncomps > 0 && (mode != SOLVE_INIT)) {
        const int n = d.ndims;
        comp = (int)pri[1];
        cost += 0.5 * sbp * sbp * pri[0];
        const double sc = 0.707106781186548 * sbp;          // literal constant (AvatarOptimizer.cpp:684)
        const double gs = sc * sbp * 0.7071067811865476;    // J^T r = sc*sbp*sqrt(1/2) * Prec (x - mu)
        const double* Pr = dm.prior_prec + (size_t)comp * n * n;
        for (int e = t; e < n * n; e += 256) {
            const int a = e / n, b = e % n;
            A[(size_t)(6 + a) * LD + 6 + b] += (sc * sc) * Pr[e];
        }
        if (t < n) A[(size_t)P * LD + 6 + t] += gs * pri[2 + t];
    }
    __syncthreads();
    if (sbs > 0.0) {
        const double* w = xt + 3 + 4 * J;
        double a = 0.0;
        if (t < K) {
            a = w[t] * sbs; a = a * a;
            A[(size_t)(3 + 3 * J + t) * LD + 3 + 3 * J + t] += sbs * sbs;
            A[(size_t)P * LD + 3 + 3 * J + t] += sbs * (w[t] * sbs);
        }
        if (t < 64) {
            a = wave_sum(a);
            if (t == 0) s_red[0] = a;
        }
        __syncthreads();
        cost += 0.5 * s_red[0];
    }
    __syncthreads();
    // keep the finalised system of this slot for a later rejected step
    double* Hf = fb.Hfin + ((size_t)f * 2 + try0) * (size_t)(P + 1) * P;
    for (int e = t; e < (P + 1) * P; e += 256) Hf[e] = A[(size_t)(e / P) * LD + (e % P)];

    // ---- b. LM decision ---------------------------------------------------------------------------
    double lambda = ctl.lambda;
    int cur = cur0;
    bool accepted = false;
    if (mode == SOLVE_FIRST) {
        accepted = true;
        cur = try0;
    } else {
        if (ctl.try_valid) {
            if (cost < ctl.cost_cur) { accepted = true; cur = try0; lambda = fmax(lambda * lm_down, lm_min); }
            else lambda = fmin(lambda * lm_up, lm_max);
        }
    }
    __syncthreads();
    if (!accepted && mode != SOLVE_LAST) {   // reload the current point's system
        const double* Hc = fb.Hfin + ((size_t)f * 2 + cur) * (size_t)(P + 1) * P;
        for (int e = t; e < (P + 1) * P; e += 256) A[(size_t)(e / P) * LD + (e % P)] = Hc[e];
    }
    const double cost_cur = accepted ? cost : ctl.cost_cur;
    __syncthreads();
    if (t == 0) {
        ctl.cur_slot = cur;
        ctl.cost_cur = cost_cur;
        if (accepted) ctl.comp_cur = comp;
        if (mode == SOLVE_FIRST) ctl.cost_initial = cost;
        else { ctl.gn_iterations += 1; if (accepted) ctl.accepted += 1; }
        const int it = ctl.gn_iterations;
        if (it < 64) fb.trace[(size_t)f * 64 + it] = cost_cur;
    }
    if (mode == SOLVE_LAST) { if (t == 0) ctl.lambda = lambda; return; }

    // ---- c. damped Cholesky solve of (H + lambda diag H) delta = -g --------------------------------
    // right-looking LL^T on the (P+1)x(P+1) bordered matrix: the extra row P carries the rhs, so forward
    // substitution falls out of the factorisation.
    for (int e = t; e < P; e += 256) {
        A[(size_t)e * LD + e] += lambda * A[(size_t)e * LD + e];
        A[(size_t)P * LD + e] = -A[(size_t)P * LD + e];
    }
    if (t == 0) s_flag = 1;
    __syncthreads();
    for (int k = 0; k < P; ++k) {
        const double piv = A[(size_t)k * LD + k];
        if (!(piv > 0.0)) { if (t == 0) s_flag = 0; break; }   // uniform: every thread reads the same value
        const double dk = sqrt(piv);
        __syncthreads();
        for (int i = k + 1 + t; i <= P; i += 256) A[(size_t)i * LD + k] = A[(size_t)i * LD + k] / dk;
        if (t == 0) { A[(size_t)k * LD + k] = dk; s_invd[k] = 1.0 / dk; }
        __syncthreads();
        // trailing update of the lower triangle (rows k+1..P, cols k+1..min(row,P-1))
        const int nrem = P - k;  // rows k+1..P
        for (int ii = (t >> 4); ii < nrem; ii += 16) {
            const int i = k + 1 + ii;
            const double lik = A[(size_t)i * LD + k];
            const int jmax = (i < P) ? i : P - 1;
            for (int j = k + 1 + (t & 15); j <= jmax; j += 16) A[(size_t)i * LD + j] -= lik * A[(size_t)j * LD + k];
        }
        __syncthreads();
    }
    __syncthreads();
    const bool ok = s_flag != 0;
    const int ntry = 1 - cur;
    double* xn = x0 + (size_t)ntry * xs;
    const double* xc = x0 + (size_t)cur * xs;
    if (ok) {
        // back substitution L^T delta = y by wave 0: lane l keeps y[l], y[l+64] in registers; L is read-only
        if (t < 64) {
            double y0 = (t < P) ? A[(size_t)P * LD + t] : 0.0;
            double y1 = (t + 64 < P) ? A[(size_t)P * LD + t + 64] : 0.0;
            for (int i = P - 1; i >= 0; --i) {
                const double yi = __shfl((i < 64) ? y0 : y1, i & 63, 64);
                const double di = yi * s_invd[i];
                if (t == (i & 63)) s_delta[i] = di;
                if (t < i) y0 -= A[(size_t)i * LD + t] * di;
                if (t + 64 < i) y1 -= A[(size_t)i * LD + t + 64] * di;
            }
        }
        __syncthreads();
        // retraction (FakeQuaternionParameterization::Plus, :123-143)
        if (t < 3) xn[t] = xc[t] + s_delta[t];
        if (t < K) xn[3 + 4 * J + t] = xc[3 + 4 * J + t] + s_delta[3 + 3 * J + t];
        if (t < J) {
            const double* dl = s_delta + 3 + 3 * t;
            const double* q = xc + 3 + 4 * t;
            const double nd = sqrt(dl[0] * dl[0] + dl[1] * dl[1] + dl[2] * dl[2]);
            double* qo = xn + 3 + 4 * t;
            if (nd > 0.0) {
                const double sdd = sin(nd) / nd;
                const double a0 = sdd * dl[0], a1 = sdd * dl[1], a2 = sdd * dl[2], a3 = cos(nd);
                qo[3] = a3 * q[3] - a0 * q[0] - a1 * q[1] - a2 * q[2];
                qo[0] = a3 * q[0] + a0 * q[3] + a1 * q[2] - a2 * q[1];
                qo[1] = a3 * q[1] + a1 * q[3] + a2 * q[0] - a0 * q[2];
                qo[2] = a3 * q[2] + a2 * q[3] + a0 * q[1] - a1 * q[0];
            } else {
                qo[0] = q[0]; qo[1] = q[1]; qo[2] = q[2]; qo[3] = q[3];
            }
        }
    } else {
        for (int e = t; e < xs; e += 256) xn[e] = xc[e];
        lambda = fmin(lambda * lm_up, lm_max);
    }
    if (t == 0) { ctl.lambda = lambda; ctl.try_valid = ok ? 1 : 0; }
    __syncthreads();
    __threadfence_block();
    // ---- d. skeleton tables of the new trial point ----------------------------------------------------
    compute_prep(dm, xn, fb.prep + ((size_t)f * 2 + ntry) * d.prep_size, *ps);
}

static size_t solve_lds_bytes(const AvtDims& d) {
    const int P = d.P;
    const int LD = P + 2 + ((P & 1) ? 0 : 1);
    return sizeof(double) * ((size_t)(P + 1) * LD + 2 * AVT_MAX_TILES * 16) + sizeof(PrepScratch) + 64;
}

static double* g_prior_buf(avt_ctx* c);

void launch_reduce(avt_ctx* c, int nframes) {
    const AvtDims& d = c->dm.d;
    hipLaunchKernelGGL(k_reduce, dim3(d.NPAIR + 1, nframes), dim3(256), 0, c->stream, c->dm, c->fb, g_prior_buf(c));
}

void launch_solve(avt_ctx* c, int nframes, int mode, const avt_options* o) {
    const AvtDims& d = c->dm.d;
    hipLaunchKernelGGL(k_solve, dim3(nframes), dim3(256), solve_lds_bytes(d), c->stream, c->dm, c->fb, g_prior_buf(c), mode,
                       o->lm_up, o->lm_down, o->lm_lambda_min, o->lm_lambda_max);
}

// the prior scratch buffer lives at the tail of the trace allocation (see avt_capi.cpp: trace has 64 doubles per
// frame followed by (2 + 3*AVT_MAX_JOINTS) doubles per frame of prior output)
static double* g_prior_buf(avt_ctx* c) { return c->fb.trace + (size_t)c->fb.max_frames * 64; }

int avt_solve_set_attributes() {
    // allow > 64 KB of dynamic LDS for the solve and eval kernels
    hipError_t e = hipFuncSetAttribute((const void*)k_solve, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
    if (e != hipSuccess) return 1;
    e = hipFuncSetAttribute((const void*)k_eval<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
    if (e != hipSuccess) return 1;
    e = hipFuncSetAttribute((const void*)k_eval<AVT_MAX_TILES>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
    return e != hipSuccess;
}
