// avt_prior.h — GMM pose prior at the trial point, one workgroup per mixture component (called from the extra
// workgroups of k_eval's grid so that it runs beside the data term): smplParams (AvatarOptimizer.cpp:664-669),
// score = ||rho_c||^2 - consts_log_c (GaussianMixture.cpp:95-114) and Prec_c (x - mu_c) for the gradient;
// k_solve picks the minimising component.
#pragma once
#include "avt_device.h"

// scratch: 2*ndims doubles of LDS
#define PRIOR_BATCH 18
// x: the state (p, q, w) the prior is evaluated at; po: where the component's score and Prec (x - mu) go
template <int NTH = 256>
__device__ __forceinline__ void prior_component_at(const DeviceModel& dm, const double* __restrict__ x, double* __restrict__ po, int c, double* scratch) {
    const AvtDims d = dm.d;
    const int t = threadIdx.x;
    const int n = d.ndims, J = d.J;
    double* s_x = scratch;
    double* s_q = scratch + AVT_MAX_JOINTS * 3;
    if (t < J - 1) {  // Eigen AngleAxis(Quaternion): angle in [0,pi], axis sign follows w
        const double* q = x + 3 + 4 * (t + 1);
        double nrm = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
        if (nrm < 2.220446049250313e-16) {
            const double mx = fmax(fabs(q[0]), fmax(fabs(q[1]), fabs(q[2])));
            if (mx > 0.0) { const double a = q[0] / mx, b = q[1] / mx, cc = q[2] / mx; nrm = mx * sqrt(a * a + b * b + cc * cc); }
            else nrm = 0.0;
        }
        double ang = 0.0, ax0 = 1.0, ax1 = 0.0, ax2 = 0.0;
        if (nrm != 0.0) {
            ang = 2.0 * atan2(nrm, fabs(q[3]));
            if (q[3] < 0) nrm = -nrm;
            ax0 = q[0] / nrm; ax1 = q[1] / nrm; ax2 = q[2] / nrm;
        }
        const double* mu = dm.prior_mean + (size_t)c * n;
        s_x[3 * t] = ax0 * ang - mu[3 * t]; s_x[3 * t + 1] = ax1 * ang - mu[3 * t + 1]; s_x[3 * t + 2] = ax2 * ang - mu[3 * t + 2];
    }
    __syncthreads();
    // y = Prec_c (x - mu_c): 4 lanes per row, NTH / 4 rows per pass
    for (int a0 = 0; a0 < n; a0 += NTH / 4) {
        const int a = a0 + (t >> 2), sub = t & 3;
        double sacc = 0.0;
        if (a < n) {      // (PRIOR_BATCH loads of the row in flight at a time - SMPL's 69 columns over four lanes are one batch -: the workgroup is one L2 round trip after another)
            const double* Pr = dm.prior_prec + ((size_t)c * n + a) * n;
            for (int b0 = sub; b0 < n; b0 += 4 * PRIOR_BATCH) {
                double v[PRIOR_BATCH];
#pragma unroll
                for (int u = 0; u < PRIOR_BATCH; ++u) v[u] = Pr[min(b0 + 4 * u, n - 1)];
#pragma unroll
                for (int u = 0; u < PRIOR_BATCH; ++u) sacc += b0 + 4 * u < n ? v[u] * s_x[b0 + 4 * u] : 0.0;
            }
        }
        sacc += __shfl_xor(sacc, 1, 64);
        sacc += __shfl_xor(sacc, 2, 64);
        if (a < n && sub == 0) { s_q[a] = sacc; po[2 + a] = sacc; }
    }
    __syncthreads();
    if (t < 64) {  // ||rho||^2 = 1/2 d^T Prec d  (rho = L^T d sqrt(1/2), Prec = L L^T)
        double sacc = 0.0;
        for (int a = t; a < n; a += 64) sacc += s_x[a] * s_q[a];
        sacc = wave_sum(sacc);
        if (t == 0) po[0] = 0.5 * sacc - dm.prior_clog[c];
    }
}

template <int NTH = 256>
__device__ __forceinline__ void prior_component(const DeviceModel& dm, const FrameBuffers& fb, int f, int c, int try_slot, double* scratch) {
    prior_component_at<NTH>(dm, fb.x + ((size_t)f * 2 + try_slot) * dm.d.xsize, fb.prior + (((size_t)f * 2 + try_slot) * AVT_MAX_COMPS + c) * AVT_PRIOR_STRIDE, c, scratch);
}
