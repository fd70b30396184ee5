// avt_decide.h - accept / reject of the LAST trial point of an ICP iteration (no further solve follows it): the same rule
// k_solve applies at the start of every other iteration (AvatarOptimizer.cpp:1486's accept test in the LM form of DESIGN
// section 4), taken inside the k_lbs launch that skins the accepted state instead of by a reduction launch in front of it.
// Every WAVE of that launch decides for itself - the inputs are a handful of values, requested in the same round trip as the
// skeleton tables - and the designated lane also writes the control block.  The inputs (AvtFrameCtl::dec_*, left by the last
// k_solve) are fields nobody writes during the launch, so all waves see the same values and take the same decision.
#pragma once
#include "avt_device.h"
#include <cstddef>

// The inputs of the decision: everything is requested by lm_last_load (no load depends on another one, none sits behind a
// branch), so that the caller can put further independent requests behind it before lm_last_decide waits for the answers.
// The per-component prior scores and the shape coefficients of BOTH slots are one vector load: lane 16 sl + c holds the score
// of component c at slot sl, lane 32 + 16 sl + k shape coefficient k of slot sl.
struct LastDecisionInputs {
    double v[AVT_G_MAX / 64];                  // this lane's partial sums of sum c|r|^2 of the trial point (workgroups lane, lane + 64, ..) ...
    unsigned long long m[AVT_G_MAX / 64];      // ... and the written-masks of the workgroups they come from
    double pv;                                 // this lane's prior score / shape coefficient (above)
    unsigned cw;                               // lane l < 40: 32-bit word l of the frame's control block; lanes 40..59: of AvtRunParams
};
static_assert(AVT_MAX_COMPS == 16 && AVT_MAX_SHAPE == 16, "lane layout of LastDecisionInputs::pv");
static_assert(sizeof(AvtFrameCtl) == 160 && sizeof(AvtRunParams) == 80, "lane layout of LastDecisionInputs::cw");

__device__ __forceinline__ LastDecisionInputs lm_last_load(const DeviceModel& dm, const FrameBuffers& fb, int f) {
    const AvtDims& d = dm.d;
    const int J = d.J, K = d.K, xs = d.xsize, G = fb.G, lane = threadIdx.x & 63, pair = d.res_pair;
    LastDecisionInputs in;
    // sum c|r|^2 of the trial point: the partial sums of the cost-only k_eval launch, lane order then a fixed butterfly
    const double* part = fb.partial + ((size_t)f * G * d.NPAIR + pair) * 256 + d.res_elem;
    const size_t st = (size_t)d.NPAIR * 256;
    const unsigned long long* wm = fb.wmask + (size_t)f * G;
#pragma unroll
    for (int u = 0; u < AVT_G_MAX / 64; ++u) {                             // G <= AVT_G_MAX (choose_G)
        const int g = min(lane + 64 * u, G - 1);
        in.v[u] = part[(size_t)g * st];
        in.m[u] = wm[g];
    }
    const int sl = (lane >> 4) & 1, c = lane & 15;
    const bool is_prior = lane < 32, there = is_prior ? c < d.ncomps : c < K;
    const double* src = is_prior ? fb.prior + (((size_t)f * 2 + sl) * AVT_MAX_COMPS + (there ? c : 0)) * AVT_PRIOR_STRIDE
                                 : fb.x + ((size_t)f * 2 + sl) * xs + 3 + 4 * J + (there ? c : 0);
    in.pv = *src;
    const unsigned* cws = lane < 40 ? (const unsigned*)(fb.ctl + f) + lane : (const unsigned*)fb.params + min(lane - 40, 19);
    in.cw = *cws;
    return in;
}

__device__ __forceinline__ double decide_readlane(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}

// Returns the slot of the current point after the decision.  All 64 lanes of the wave must be active.
__device__ __forceinline__ int lm_last_decide(const DeviceModel& dm, const FrameBuffers& fb, int f, const LastDecisionInputs& in, bool writer) {
    const AvtDims& d = dm.d;
    AvtFrameCtl& ctl = fb.ctl[f];
    const int lane = threadIdx.x & 63, G = fb.G, pair = d.res_pair;
    auto cword = [&](int w) { return __builtin_amdgcn_readlane((int)in.cw, w); };
    auto cdbl = [&](int w) { return __hiloint2double(cword(w + 1), cword(w)); };
#define AVT_CTL_W(field) ((int)(offsetof(AvtFrameCtl, field) / 4))
#define AVT_PAR_W(field) (40 + (int)(offsetof(AvtRunParams, field) / 4))
    const double sbp = cdbl(AVT_CTL_W(sbp)), sbs = cdbl(AVT_CTL_W(sbs)), cost_cur0 = cdbl(AVT_CTL_W(dec_cost_cur)), cost_const = cdbl(AVT_CTL_W(cost_const));
    double lambda = cdbl(AVT_CTL_W(dec_lambda));
    const double pred0 = cdbl(AVT_CTL_W(dec_pred)), nu0 = cdbl(AVT_CTL_W(dec_nu));
    const int cur0 = cword(AVT_CTL_W(dec_cur_slot)), try_valid = cword(AVT_CTL_W(dec_try_valid));
    const double lm_up = cdbl(AVT_PAR_W(lm_up)), lm_down = cdbl(AVT_PAR_W(lm_down)), lm_min = cdbl(AVT_PAR_W(lm_min)), lm_max = cdbl(AVT_PAR_W(lm_max));
    const bool gain = cdbl(AVT_PAR_W(lm_policy)) != 0.0;
#undef AVT_CTL_W
#undef AVT_PAR_W
    // a workgroup without batches wrote nothing: its tile is stale memory
    double a = 0.0;
#pragma unroll
    for (int u = 0; u < AVT_G_MAX / 64; ++u) {
        const bool on = lane + 64 * u < G && (pair >= 64 || ((in.m[u] >> (pair & 63)) & 1ull));
        a += on ? in.v[u] : 0.0;
    }
    const bool there = lane < 32 ? (lane & 15) < d.ncomps : (lane & 15) < d.K;
    const double pv = there ? in.pv : (lane < 32 ? 1.7976931348623157e308 : 0.0);
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) a += __shfl_xor(a, sft, 64);     // a + b == b + a: every lane ends with the same bits
    const int try0 = 1 - cur0;
    double cost = lm_objective_data(a, cost_const);      // (avt_device.h: the one spelling of the objective)
    int comp_try = -1;
    {   // strict '<' in ascending component order (GaussianMixture.cpp:103); absent components read as the largest double
        double best = 1.7976931348623157e308;
        int bc = -1;
#pragma unroll
        for (int c = 0; c < AVT_MAX_COMPS; ++c) {
            const double s0 = decide_readlane(pv, c), s1 = decide_readlane(pv, 16 + c);
            const double v = try0 ? s1 : s0;
            if (v < best) { best = v; bc = c; }
        }
        if (sbp > 0.0 && d.ncomps > 0) { comp_try = bc; cost = lm_objective_add_pose(cost, sbp, best); }
    }
    if (sbs > 0.0) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < AVT_MAX_SHAPE; ++k) {      // absent coefficients read as zero
            const double x0 = decide_readlane(pv, 32 + k), x1 = decide_readlane(pv, 48 + k);
            s = lm_shape_term_add(s, try0 ? x1 : x0, sbs);
        }
        cost = lm_objective_add_shape(cost, s);
    }
    // the frame met the stopping rule earlier in this ICP iteration (avt_options::function_tolerance, k_solve): there is no trial point and no test
    if (try_valid == AVT_TRY_DONE) return cur0;
    bool accepted = false;
    double nu = nu0;
    if (try_valid) {      // the same rule as k_solve's (avt_lm.hip)
        if (cost < cost_cur0) {
            accepted = true;
            if (gain) {
                const double u = 2.0 * ((cost_cur0 - cost) / pred0) - 1.0;
                lambda = fmin(fmax(lambda * fmax(lm_down, 1.0 - u * u * u), lm_min), lm_max);
                nu = lm_up;
            } else lambda = fmax(lambda * lm_down, lm_min);
        } else if (gain) { lambda = fmin(lambda * nu, lm_max); nu *= 2.0; }
        else lambda = fmin(lambda * lm_up, lm_max);
    }
    if (writer) {
        const double cost_cur = accepted ? cost : cost_cur0;
        if (accepted) { ctl.cur_slot = try0; ctl.cost_cur = cost; ctl.comp_cur = comp_try; ctl.accepted += 1; }
        const int it = ctl.gn_iterations + 1;
        ctl.gn_iterations = it;
        if (it < 40) fb.trace[(size_t)f * 64 + it] = cost_cur;
        ctl.lambda = lambda; ctl.nu = nu;
    }
    return accepted ? try0 : cur0;
}
