"""More GPU parity cases: full-size / dense frames, other part maps and knobs, ragged batches, determinism and
size-independent properties (translation equivariance, monotone objective)."""
import numpy as np
import pytest

from avatar_amd import synth
from avatar_amd.capi import Options

pytestmark = pytest.mark.gpu


def _start(fr):
    from avatar_amd import api
    w0, p0, R0 = fr["start"]
    return p0, api.rot_to_quat(R0), w0


def _check(ctx, ref, p, q, w, n, tol=1e-6):
    assert np.array_equal(ctx.correspondences(0, n), ref["corr"])
    assert np.abs(ctx.cloud(0) - ref["cloud"]).max() < tol
    assert np.abs(p[0] - ref["p"]).max() < tol and np.abs(w[0] - ref["w"]).max() < 10 * tol
    dq = np.minimum(np.abs(q[0] - ref["q"]).max(1), np.abs(q[0] + ref["q"]).max(1))
    assert dq.max() < tol


def test_dense_120k_frame_matches_oracle(smpl, omodel, gmodel):
    """BASELINE configs[4]: the 2560x1440 render (~150k points)."""
    from avatar_amd import api
    fr = synth.make_frame(smpl, 0, dense=True)
    assert len(fr["labels"]) > 100000
    pm = synth.identity_part_map()
    p0, q0, w0 = _start(fr)
    opt = Options.demo()
    ctx = api.Context(gmodel, 24, pm, len(fr["labels"]), 1)
    p, q, w, st = ctx.optimize_batch([fr["data"]], [fr["labels"]], opt, p0[None], q0[None], w0[None])
    ref = omodel.optimize(pm, 24, fr["data"], fr["labels"], opt, p0, q0, w0, aggregate=1)
    _check(ctx, ref, p, q, w, len(fr["labels"]))
    assert st[0].num_correspondences == ref["stats"].num_correspondences


def test_coarse_part_map_and_reference_default_betas(smpl, omodel, gmodel):
    from avatar_amd import api
    coarse = np.array([0, 1, 2, 0, 1, 2, 0, 1, 2, 0, 1, 2, 3, 3, 3, 3, 4, 5, 4, 5, 4, 5, 4, 5], np.int32)
    fr = synth.make_frame(smpl, 13, part_map=coarse)
    p0, q0, w0 = _start(fr)
    opt = Options.reference_defaults()          # betaPose 0.1, betaShape 1.0 (AvatarOptimizer.h:28)
    opt.icp_iters = 2
    ctx = api.Context(gmodel, 6, coarse, len(fr["labels"]), 1)
    p, q, w, st = ctx.optimize_batch([fr["data"]], [fr["labels"]], opt, p0[None], q0[None], w0[None])
    ref = omodel.optimize(coarse, 6, fr["data"], fr["labels"], opt, p0, q0, w0, aggregate=1)
    _check(ctx, ref, p, q, w, len(fr["labels"]))


def test_model_without_pose_prior(smpl):
    from avatar_amd import api
    from oracle import oracle as orc
    m2 = {k: v for k, v in smpl.items() if not k.startswith("prior_")}
    gm, om = api.AvatarModel(m2), orc.OracleModel(m2)
    fr = synth.make_frame(smpl, 14)
    pm = synth.identity_part_map()
    p0, q0, w0 = _start(fr)
    opt = Options.demo(max_iters_per_icp=6)
    ctx = api.Context(gm, 24, pm, len(fr["labels"]), 1)
    p, q, w, st = ctx.optimize_batch([fr["data"]], [fr["labels"]], opt, p0[None], q0[None], w0[None])
    ref = om.optimize(pm, 24, fr["data"], fr["labels"], opt, p0, q0, w0, aggregate=1)
    _check(ctx, ref, p, q, w, len(fr["labels"]))


def test_model_with_six_shape_keys_runs_the_runtime_dimension_kernels(smpl):
    """K = 6 (P = 81): k_eval<0,0>, the runtime-sized skeleton pass and LDL^T / back substitution (21 pivot blocks)."""
    from avatar_amd import api
    from oracle import oracle as orc
    m2 = dict(smpl)
    m2["shapedirs"] = np.ascontiguousarray(smpl["shapedirs"][:, :, :6])
    gm, om = api.AvatarModel(m2), orc.OracleModel(m2)
    fr = synth.make_frame(smpl, 17)
    pm = synth.identity_part_map()
    w0, p0, R0 = fr["start"]
    w0 = np.ascontiguousarray(w0[:6]); q0 = api.rot_to_quat(R0)
    opt = Options.demo()
    ctx = api.Context(gm, 24, pm, len(fr["labels"]), 1)
    p, q, w, st = ctx.optimize_batch([fr["data"]], [fr["labels"]], opt, p0[None], q0[None], w0[None])
    ref = om.optimize(pm, 24, fr["data"], fr["labels"], opt, p0, q0, w0, aggregate=1)
    _check(ctx, ref, p, q, w, len(fr["labels"]))
    assert st[0].gn_iterations == ref["stats"].gn_iterations


def test_model_with_twelve_shape_keys_keeps_the_plain_column_order(smpl):
    """K = 12 (P = 87, the largest system this build accepts): translation + shape keys + root no longer fit one 16-column
    tile, so build_tile_layout falls back to storage column = parameter index with every tile live; k_eval<0,0>."""
    import ctypes
    from avatar_amd import api, capi
    from oracle import oracle as orc
    rng = np.random.default_rng(5)
    m2 = dict(smpl)
    extra = 0.01 * rng.standard_normal(smpl["shapedirs"].shape[:2] + (2,))
    m2["shapedirs"] = np.ascontiguousarray(np.concatenate([smpl["shapedirs"], extra], axis=2))
    gm, om = api.AvatarModel(m2), orc.OracleModel(m2)
    lib = capi.load_library()
    nt = ctypes.c_int(); tp = np.zeros(16 * 11, np.int32)
    assert lib.avt_model_tile_layout(gm.h, ctypes.byref(nt), capi.iptr(tp), None, None) == 0
    assert nt.value == 6 and np.array_equal(tp[:88], np.arange(88)) and np.all(tp[88:96] == -1)
    fr = synth.make_frame(smpl, 18)
    pm = synth.identity_part_map()
    w0, p0, R0 = fr["start"]
    w0 = np.ascontiguousarray(np.concatenate([w0, [0.3, -0.2]])); q0 = api.rot_to_quat(R0)
    opt = Options.demo()
    ctx = api.Context(gm, 24, pm, len(fr["labels"]), 1)
    p, q, w, st = ctx.optimize_batch([fr["data"]], [fr["labels"]], opt, p0[None], q0[None], w0[None])
    ref = om.optimize(pm, 24, fr["data"], fr["labels"], opt, p0, q0, w0, aggregate=1)
    _check(ctx, ref, p, q, w, len(fr["labels"]))
    assert st[0].gn_iterations == ref["stats"].gn_iterations


def test_normal_equations_in_parameter_order(smpl, omodel, gmodel):
    """avt_get_normal_equations after a fit: the data term J^T J, J^T r the block-sparse contraction leaves (tile order inside
    k_eval, parameter order on the ABI) against the oracle's dense per-block accumulation at the same point and
    correspondences (AvatarOptimizer.cpp:473-503, :609-644)."""
    from avatar_amd import api
    pm = synth.identity_part_map()
    fr = synth.make_frame(smpl, 19)
    p0, q0, w0 = _start(fr)
    opt = Options.demo(max_iters_per_icp=5)
    n = len(fr["labels"])
    for frames in (1, 16):       # one frame: every tile written (k_reduce<4>); a batch: chunked batches, masked partial tiles (k_reduce<1>)
        ctx = api.Context(gmodel, 24, pm, n, frames)
        p, q, w, st = ctx.optimize_batch([fr["data"]] * frames, [fr["labels"]] * frames, opt, np.repeat(p0[None], frames, 0),
                                         np.repeat(q0[None], frames, 0), np.repeat(w0[None], frames, 0))
        f = frames - 1
        H, g, cost = ctx.normal_equations(f)
        corr = ctx.correspondences(f, n)
        oc, og, oH, _ = omodel.evaluate(p[f], q[f], w[f], corr, fr["data"], 0.0, 0.0, aggregate=0)
        scale = np.abs(oH).max()
        assert np.abs(H - oH).max() < 1e-9 * scale, np.abs(H - oH).max() / scale
        assert np.abs(H - H.T).max() == 0.0
        assert np.abs(g - og).max() < 1e-9 * max(1.0, np.abs(og).max())
        # structural zeros the tile grouping relies on: a left-leg joint and a right-arm joint never share a model point
        blk = H[3 + 3 * 4: 6 + 3 * 4, 3 + 3 * 19: 6 + 3 * 19]
        assert np.all(blk == 0.0) and np.all(oH[3 + 3 * 4: 6 + 3 * 4, 3 + 3 * 19: 6 + 3 * 19] == 0.0)


def test_ragged_batch_with_empty_frame_and_determinism(smpl, omodel, gmodel):
    from avatar_amd import api
    pm = synth.identity_part_map()
    frs = [synth.make_frame(smpl, s) for s in (15, 16)]
    datas = [frs[0]["data"], np.zeros((0, 3)), frs[1]["data"][:777]]
    labs = [frs[0]["labels"], np.zeros(0, np.int32), frs[1]["labels"][:777]]
    starts = [_start(frs[0]), _start(frs[0]), _start(frs[1])]
    opt = Options.demo(max_iters_per_icp=5)
    ctx = api.Context(gmodel, 24, pm, 60000, 3)
    args = (datas, labs, opt, np.array([s[0] for s in starts]), np.array([s[1] for s in starts]), np.array([s[2] for s in starts]))
    p1, q1, w1, st1 = ctx.optimize_batch(*args)
    p2, q2, w2, st2 = ctx.optimize_batch(*args)
    # bit-wise reproducible run to run (fixed-order reductions, integer atomics)
    assert np.array_equal(p1, p2) and np.array_equal(q1, q2) and np.array_equal(w1, w2)
    # the empty frame is left untouched
    assert st1[1].num_correspondences == 0 and np.array_equal(p1[1], starts[1][0]) and np.array_equal(w1[1], starts[1][2])
    for f in (0, 2):
        ref = omodel.optimize(pm, 24, datas[f], labs[f], opt, *starts[f], aggregate=1)
        assert np.abs(p1[f] - ref["p"]).max() < 1e-6 and np.abs(w1[f] - ref["w"]).max() < 1e-5


def test_refused_factorisation_leaves_the_state_untouched(smpl, omodel, gmodel):
    """No priors and three data points: most joints have no matched point, their rows of H are zero, H + lambda diag H is singular and
    every factorisation is refused - on the device (whose rounds run on past the refused pivot and look at the flag once, behind the
    last one: avt_lm.hip, mf_round) as in the oracle.  The state comes back bit for bit, nothing is NaN, and the other frames of the
    batch do what the oracle does with them."""
    from avatar_amd import api
    pm = synth.identity_part_map()
    fr = synth.make_frame(smpl, 9)
    start = _start(fr)
    opt = Options.demo(max_iters_per_icp=4)
    opt.beta_pose = 0.0; opt.beta_shape = 0.0
    for frames in (1, 5):
        ctx = api.Context(gmodel, 24, pm, 60000, frames)
        datas = [fr["data"][:3]] + [fr["data"]] * (frames - 1)
        labs = [fr["labels"][:3]] + [fr["labels"]] * (frames - 1)
        p0 = np.repeat(start[0][None], frames, 0); q0 = np.repeat(start[1][None], frames, 0); w0 = np.repeat(start[2][None], frames, 0)
        p, q, w, st = ctx.optimize_batch(datas, labs, opt, p0, q0, w0)
        ref = omodel.optimize(pm, 24, datas[0], labs[0], opt, *start, aggregate=1)
        assert ref["stats"].accepted_steps == 0 and st[0].accepted_steps == 0
        assert np.array_equal(p[0], start[0]) and np.array_equal(q[0], start[1]) and np.array_equal(w[0], start[2])
        assert np.isfinite(p).all() and np.isfinite(q).all() and np.isfinite(w).all()
        if frames > 1:      # (the full frames of the batch: whatever the oracle does with them without priors, step or refuse)
            ref1 = omodel.optimize(pm, 24, datas[1], labs[1], opt, *start, aggregate=1)
            assert st[1].accepted_steps == ref1["stats"].accepted_steps and np.abs(p[1] - ref1["p"]).max() < 1e-6


def test_literal_dimension_kernels_agree_with_the_runtime_dimension_copies(smpl, gmodel):
    """avt_tuning.literal_dims: a model with SMPL's dimensions runs the copies of k_solve / k_pairpass / k_prior that are compiled for them
    (every count a literal); with the knob at 0 it runs the run-time-dimension copies every other model runs.  Same source: the fits agree to
    rounding - one frame (the riding shape), five frames (row form, batch shape), forty frames (moment form, two groups)."""
    from avatar_amd import api
    pm = synth.identity_part_map()
    frs = [synth.make_frame(smpl, 60 + s) for s in range(4)]
    opt = Options.demo(max_iters_per_icp=6)
    for frames in (1, 5, 40):
        pick = [i % len(frs) for i in range(frames)]
        starts = [_start(frs[i]) for i in pick]
        args = ([frs[i]["data"] for i in pick], [frs[i]["labels"] for i in pick], opt, np.array([s[0] for s in starts]), np.array([s[1] for s in starts]),
                np.array([s[2] for s in starts]))
        res = []
        for lit in (1, 0):
            ctx = api.Context(gmodel, 24, pm, 60000, frames)
            ctx.set_tuning(literal_dims=lit)
            assert ctx.tuning().literal_dims == lit
            res.append(ctx.optimize_batch(*args))
            del ctx
        (p1, q1, w1, st1), (p0, q0, w0, st0) = res
        assert np.abs(p1 - p0).max() < 1e-9 and np.abs(q1 - q0).max() < 1e-9 and np.abs(w1 - w0).max() < 1e-9, frames
        assert [s.accepted_steps for s in st1] == [s.accepted_steps for s in st0] and [s.gn_iterations for s in st1] == [s.gn_iterations for s in st0]
        assert all(abs(a.final_cost - b.final_cost) <= 1e-9 * abs(b.final_cost) for a, b in zip(st1, st0))


def test_translation_equivariance_and_monotone_cost(smpl, gmodel):
    """Size-independent properties at full size: shifting the data and the start by t shifts the fit by t (the model
    enters only through p + R(...)); the LM objective never increases."""
    from avatar_amd import api
    pm = synth.identity_part_map()
    fr = synth.make_frame(smpl, 17)
    p0, q0, w0 = _start(fr)
    opt = Options.demo()
    ctx = api.Context(gmodel, 24, pm, len(fr["labels"]), 1)
    p, q, w, st = ctx.optimize_batch([fr["data"]], [fr["labels"]], opt, p0[None], q0[None], w0[None])
    assert st[0].final_cost <= st[0].initial_cost
    t = np.array([0.25, -0.125, 0.5])          # exactly representable shift
    pt, qt, wt, stt = ctx.optimize_batch([fr["data"] + t], [fr["labels"]], opt, (p0 + t)[None], q0[None], w0[None])
    assert np.abs(pt[0] - t - p[0]).max() < 1e-7 and np.abs(wt[0] - w[0]).max() < 1e-6
    assert np.abs(qt[0] - q[0]).max() < 1e-7


def test_large_batch_two_frame_groups_matches_oracle(smpl, omodel, gmodel):
    """70 frames: the two-stream frame-group pipeline (uneven 35/35 split boundary checked), spot-checked against the oracle."""
    from avatar_amd import api
    F = 70
    frs = [synth.make_frame(smpl, 100 + s) for s in range(F)]
    pm = synth.identity_part_map()
    ctx = api.Context(gmodel, 24, pm, 60000, F)
    p0 = np.array([f["start"][1] for f in frs]); q0 = np.array([api.rot_to_quat(f["start"][2]) for f in frs]); w0 = np.array([f["start"][0] for f in frs])
    opt = Options.demo()
    p, q, w, st = ctx.optimize_batch([f["data"] for f in frs], [f["labels"] for f in frs], opt, p0, q0, w0)
    for i in (0, 34, 35, 69):
        ref = omodel.optimize(pm, 24, frs[i]["data"], frs[i]["labels"], opt, p0[i], q0[i], w0[i], aggregate=1)
        assert np.array_equal(ctx.correspondences(i, len(frs[i]["labels"])), ref["corr"])
        assert np.abs(ctx.cloud(i) - ref["cloud"]).max() < 1e-6
        assert st[i].gn_iterations == ref["stats"].gn_iterations and st[i].accepted_steps == ref["stats"].accepted_steps


def test_512_frames_per_gpu_rendered_on_the_gpu(smpl, omodel, gmodel):
    """BASELINE configs[3] / the saturation point DESIGN quotes: 512 independent ~30k-point frames resident on one GPU
    (the per-GPU share of configs[3] is 64; 512 is where the frames-per-GPU curve flattens).  Frames come from the GPU
    generator, exactly as bench.py makes them.  Checks: every frame finite and improved; bit-wise reproducible run to run;
    frames at both ends of both frame groups against the oracle (correspondences bit-exact, vertices 1e-6); a frame's fit
    does not depend on its neighbours in the batch (the same frame alone agrees to 1e-9: only the summation order of the
    partial tiles differs with the launch shape)."""
    from avatar_amd import api
    F = 512
    pm = synth.identity_part_map()
    gts = [synth.sample_ground_truth(smpl, 2000 + f) for f in range(F)]
    starts = [synth.perturb_start(*gts[f], 2000 + f) for f in range(F)]
    ctx = api.Context(gmodel, 24, pm, 65536, F, device=0)
    npts = ctx.render_frames(np.array([g[0] for g in gts]), np.array([g[1] for g in gts]), np.array([g[2] for g in gts]))
    assert npts.min() > 5000 and npts.max() <= 65536
    p0 = np.array([s[1] for s in starts]); w0 = np.array([s[0] for s in starts])
    q0 = api.rot_to_quat(np.array([s[2] for s in starts]).reshape(-1, 3, 3)).reshape(F, 24, 4)
    ctx.state_upload(p0, q0, w0)
    opt = Options.demo()
    assert ctx.launch_shape()[:2] == (2, 256)
    ctx.optimize_resident(opt)
    p, q, w, st = ctx.state_download()
    ctx.state_reset(); ctx.optimize_resident(opt)
    p2, q2, w2, _ = ctx.state_download()
    assert np.array_equal(p, p2) and np.array_equal(q, q2) and np.array_equal(w, w2)
    assert np.isfinite(p).all() and np.isfinite(q).all() and np.isfinite(w).all()
    # (Options.demo() carries the reference's stopping rule, function_tolerance = 1e-4: a frame may end its iterations early)
    assert all(1 <= s.gn_iterations <= 10 and s.final_cost <= s.initial_cost and s.num_correspondences > 0 for s in st)
    assert np.abs(np.linalg.norm(q, axis=2) - 1.0).max() < 1e-9
    for i in (0, 255, 256, 511):
        d, l = ctx.frame_download(i)
        assert len(l) == npts[i]
        ref = omodel.optimize(pm, 24, d, l, opt, p0[i], q0[i], w0[i], aggregate=1)
        assert np.array_equal(ctx.correspondences(i, len(l)), ref["corr"])
        assert np.abs(ctx.cloud(i) - ref["cloud"]).max() < 1e-6
        assert st[i].accepted_steps == ref["stats"].accepted_steps
        assert np.abs(p[i] - ref["p"]).max() < 1e-7 and np.abs(q[i] - ref["q"]).max() < 1e-7
    one = api.Context(gmodel, 24, pm, 65536, 1, device=0)
    for i in (3, 300):
        d, l = ctx.frame_download(i)
        pa, qa, wa, _ = one.optimize_batch([d], [l], opt, p0[i][None], q0[i][None], w0[i][None])
        assert np.abs(pa[0] - p[i]).max() < 1e-9 and np.abs(qa[0] - q[i]).max() < 1e-9 and np.abs(wa[0] - w[i]).max() < 1e-8


@pytest.mark.gpu
def test_every_launch_shape_gives_the_single_frame_answer(smpl, gmodel):
    """optimize() picks its launch shape by frame count: the few-frames shape (up to 6 frames: strided batches, k_reduce_strip,
    the trial point set up in k_lbs's grid), the batch shape on one frame group (7..31) and two frame groups (32 on).  The same
    frame must come out the same (to summation order) from all of them, and identical frames identically inside a batch."""
    from avatar_amd import api
    pm = synth.identity_part_map()
    fr = synth.make_frame(smpl, 21)
    data, labels = fr["data"][::3], fr["labels"][::3]
    p0, q0, w0 = _start(fr)
    opt = Options.demo(max_iters_per_icp=4, icp_iters=2)
    ctx1 = api.Context(gmodel, 24, pm, len(labels), 1)
    pr, qr, wr, sr = ctx1.optimize_batch([data], [labels], opt, p0[None], q0[None], w0[None])
    for frames in (2, 3, 6, 7, 12, 31, 33, 45):
        ctx = api.Context(gmodel, 24, pm, len(labels), frames)
        p, q, w, st = ctx.optimize_batch([data] * frames, [labels] * frames, opt, np.repeat(p0[None], frames, 0),
                                         np.repeat(q0[None], frames, 0), np.repeat(w0[None], frames, 0))
        groups, nfg, G = ctx.launch_shape()
        # (two and three frames run one frame per group: each keeps its own pace through the speculative steps of DESIGN section 4)
        assert (G >= 64) == (frames <= 6) and groups == (frames if frames <= 3 else (2 if frames >= 44 else 1)), (frames, groups, nfg, G)
        assert np.abs(p - pr).max() < 1e-9 and np.abs(q - qr).max() < 1e-9 and np.abs(w - wr).max() < 1e-8, frames
        assert all(s.gn_iterations == sr[0].gn_iterations and s.accepted_steps == sr[0].accepted_steps for s in st)
        same_group = range(nfg)          # frames of one group run the same launches: bit-identical results
        assert all(np.array_equal(p[f], p[0]) and np.array_equal(q[f], q[0]) and np.array_equal(w[f], w[0]) for f in same_group)


@pytest.mark.gpu
def test_in_launch_hand_over_is_reproducible(smpl, gmodel):
    """Up to three frames the reduced system reaches the solver INSIDE the k_solve launch (agent-scope stores, a counter, a
    bounded spin): a lost, early or torn hand-over would change bits.  Every repeat of the same call must download the same
    bytes, and the launch shape with the reduction as its own launch (four frames) must agree to summation order."""
    from avatar_amd import api
    pm = synth.identity_part_map()
    frs = [synth.make_frame(smpl, 40 + s) for s in range(4)]
    opt = Options.demo()
    res = {}
    for F in (1, 3, 4):
        ctx = api.Context(gmodel, 24, pm, 60000, F)
        ctx.frames_upload([f["data"] for f in frs[:F]], [f["labels"] for f in frs[:F]])
        ctx.state_upload(np.array([f["start"][1] for f in frs[:F]]), np.array([api.rot_to_quat(f["start"][2]) for f in frs[:F]]),
                         np.array([f["start"][0] for f in frs[:F]]))
        ctx.state_reset(); ctx.optimize_resident(opt)
        ref = ctx.state_download()
        for _ in range(150):
            ctx.state_reset(); ctx.optimize_resident(opt)
            p, q, w, _st = ctx.state_download()
            assert np.array_equal(p, ref[0]) and np.array_equal(q, ref[1]) and np.array_equal(w, ref[2])
        res[F] = ref
    for F in (3, 4):
        assert np.abs(res[F][0][0] - res[1][0][0]).max() < 1e-9 and np.abs(res[F][1][0] - res[1][1][0]).max() < 1e-9


@pytest.mark.gpu
def test_speculative_steps_do_not_change_a_bit(smpl, gmodel):
    """One and two frames: the solve launch factors the steps a run of rejections will ask for beside the one needed now, and a
    rejection installs the step that is already there.  With the speculative workgroups switched off
    (avt_tuning.nspec = 0) every step is factored when it is asked for: same bytes."""
    from avatar_amd import api
    pm = synth.identity_part_map()
    outs = {}
    for nspec in (0, 4, 1):
        res = []
        for F, seeds in ((1, (4, 8)), (2, (5, 6))):       # frames with runs of three to five rejections
            frs = [synth.make_frame(smpl, s) for s in seeds[:F]] if F == 2 else None
            for s in (seeds if F == 1 else (0,)):
                fl = [synth.make_frame(smpl, s)] if F == 1 else frs
                ctx = api.Context(gmodel, 24, pm, 60000, F).set_tuning(nspec=nspec)
                assert ctx.tuning().nspec == nspec
                p, q, w, st = ctx.optimize_batch([f["data"] for f in fl], [f["labels"] for f in fl], Options.demo(icp_iters=2),
                                                 np.array([f["start"][1] for f in fl]), np.array([api.rot_to_quat(f["start"][2]) for f in fl]),
                                                 np.array([f["start"][0] for f in fl]))
                res += [p.ravel(), q.ravel(), w.ravel(), np.array([x.final_cost for x in st]), np.array([x.accepted_steps for x in st], float)]
        outs[str(nspec)] = np.concatenate(res)
    assert np.array_equal(outs["0"], outs["4"]) and np.array_equal(outs["0"], outs["1"])
    assert outs["0"].size > 400


def test_slab_scan_against_nanoflann_goldens(smpl, omodel, gmodel):
    """The throughput shape of the nearest neighbour (k_compact sorting each part's visible candidates by (y, vertex id), k_nn_part
    walking outwards from the wave's slab of queries until the y gap alone exceeds the worst best distance) on the reference's own
    nanoflann outputs: the four small cases, 38 k and 125 k queries, and the coarse 6-part map whose parts are larger than the
    sort's capacity (they take the unsorted full scan).  avt_tuning.nn_force_part routes the stand-alone avt_nn through that shape.
    Bit-exact, and identical with the slab switched off (avt_tuning.nn_slab = 0)."""
    import os
    import sys
    from avatar_amd import api
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import make_nn_golden_full as mk
    zf = np.load(os.path.join(here, "golden", "nn_golden_full.npz"))
    zs = np.load(os.path.join(here, "golden", "nn_golden.npz"))
    for k in range(int(zs["ncase"])):
        pm = zs[f"part_map_{k}"]; npart = int(zs[f"num_parts_{k}"])
        ctx = api.Context(gmodel, npart, pm, 60000, 1).set_tuning(nn_force_part=1)
        got = ctx.nn(zs[f"cloud_{k}"], zs[f"vis_{k}"], zs[f"data_{k}"], zs[f"labels_{k}"])
        assert np.array_equal(got, zs[f"idx_{k}"]), f"small case {k}"
    for k, c in enumerate(mk.CASES):
        pm, npart, cloud, vis, data, labels = mk.case_inputs(smpl, omodel, c)
        ctx = api.Context(gmodel, npart, pm, len(labels), 1, device=0).set_tuning(nn_force_part=1)
        got = ctx.nn(cloud, vis, data, labels)
        assert np.array_equal(got, zf[f"idx_{k}"]), (k, int((got != zf[f"idx_{k}"]).sum()))
        ctx.set_tuning(nn_slab=0)
        assert np.array_equal(ctx.nn(cloud, vis, data, labels), got)


def test_slab_scan_settles_exact_ties_by_vertex_id(smpl, omodel, gmodel):
    """Model points that coincide exactly (distances tie bit for bit): the winner is the smallest vertex id, as in an ascending
    scan with strict '<' (nanoflann.hpp:175-199 semantics of the oracle's ordered brute force), although the slab scan meets
    the candidates in y order.  Every vertex of a part is given a twin with a different id."""
    from avatar_amd import api
    pm = synth.identity_part_map()
    fr = synth.make_frame(smpl, 2)
    w0, p0, R0 = fr["start"]
    c0, _, _ = omodel.update(w0, p0, R0)
    vis = omodel.visibility(c0, True)
    part = np.asarray(pm)[synth.main_joint(smpl)]
    cloud = c0.copy()
    rng = np.random.default_rng(5)
    for q in range(24):                       # within every part: random pairs (a, b) of visible vertices, b moved onto a
        ids = np.nonzero((part == q) & (vis != 0))[0]
        ids = rng.permutation(ids)
        half = len(ids) // 2
        cloud[ids[half:2 * half]] = cloud[ids[:half]]
    sel = slice(0, None, 3)
    data, labels = fr["data"][sel], fr["labels"][sel]
    ref = omodel.nn(pm, 24, cloud, vis, data, labels)
    ctx = api.Context(gmodel, 24, pm, 60000, 1).set_tuning(nn_force_part=1)
    got = ctx.nn(cloud, vis, data, labels)
    assert np.array_equal(got, ref), int((got != ref).sum())
    # the ties were real: most matched vertices have a twin at the same position with another id
    m = ref[ref >= 0]
    d = np.abs(cloud[:, None, :] - cloud[None, m[:200], :]).max(-1) == 0
    assert (d.sum(0) >= 2).mean() > 0.5


def test_limit_one_joint_per_point_matches_oracle(smpl, frame0):
    """AvatarModel(dir, limit_one_joint_per_point = true) (Avatar.h:76-77, AvatarModel.cpp:190-196): the optimiser's forward model
    and Jacobians bind every point to its largest-weight joint (weight 1), Avatar::update() keeps all weights.  GPU against the
    oracle built from the same model description: update() identical to the unrestricted model, the fit identical to the oracle's."""
    from avatar_amd import api
    from oracle import oracle as orc
    pm = synth.identity_part_map()
    g1 = api.AvatarModel(smpl, limit_one_joint_per_point=True)
    g0 = api.AvatarModel(smpl)
    o1 = orc.OracleModel(smpl, limit_one_joint_per_point=True)
    w0, p0, R0 = frame0["start"]
    c1 = g1.default_ctx().lbs_update(w0[None], p0[None], R0[None])[0]
    c0 = g0.default_ctx().lbs_update(w0[None], p0[None], R0[None])[0]
    assert np.array_equal(c1, c0)                                 # `weights` (all entries) drive update()
    sel = slice(0, None, 4)
    data, labels = frame0["data"][sel], frame0["labels"][sel]
    q0 = api.rot_to_quat(R0)
    opt = Options.demo(icp_iters=2, max_iters_per_icp=5)
    ref = o1.optimize(pm, 24, data, labels, opt, p0, q0, w0, aggregate=1)
    ctx = api.Context(g1, 24, pm, 16384, 1)
    p, q, w, st = ctx.optimize_batch([data], [labels], opt, p0[None], q0[None], w0[None])
    assert np.array_equal(ctx.correspondences(0, len(labels)), ref["corr"])
    assert st[0].accepted_steps == ref["stats"].accepted_steps
    assert abs(st[0].final_cost - ref["stats"].final_cost) <= 1e-9 * abs(ref["stats"].final_cost)
    assert np.abs(ctx.cloud(0) - ref["cloud"]).max() < 1e-6 and np.abs(p[0] - ref["p"]).max() < 1e-7
    # and it is a different fit from the blended model's (the flag does something)
    ref0 = orc.OracleModel(smpl).optimize(pm, 24, data, labels, opt, p0, q0, w0, aggregate=1)
    assert abs(ref0["stats"].final_cost - ref["stats"].final_cost) > 1e-6 * abs(ref["stats"].final_cost)


@pytest.mark.gpu
def test_folded_accept_tests_twelve_seeds_both_policies(smpl, gmodel):
    """the folded accept test forms the objective in reduce_spec_cost, the unfolded one in k_solve - one spelling of the expression now
    (avt_device.h, lm_objective_*: individually rounded operations, nothing for the compiler to contract differently in the two places).  The
    twelve bench seeds, both damping policies (the fixed factors reject in runs, which is where tests are folded): bit-identical with and without."""
    from avatar_amd import api
    pm = synth.identity_part_map()
    folded_somewhere = 0
    for seed in range(12):
        fr = synth.make_frame(smpl, seed)
        data, labels = fr["data"][::3], fr["labels"][::3]
        p0, q0, w0 = _start(fr)
        for policy in (0, 1):
            opt = Options.counted(lm_policy=policy)
            out = []
            for sc in (0, 1):
                ctx = api.Context(gmodel, 24, pm, len(labels), 1)
                ctx.set_tuning(spec_cost=sc)
                p, q, w, st = ctx.optimize_batch([data], [labels], opt, p0[None], q0[None], w0[None])
                out.append((p, q, w, np.array([st[0].gn_iterations, st[0].accepted_steps]), np.array([st[0].lambda_, st[0].final_cost]), ctx.cost_trace(0)))
            for a_, b_ in zip(*out):
                assert np.array_equal(a_, b_), (seed, policy)
            tr = out[0][5]
            folded_somewhere += int(sum(1 for k in range(9) if tr[k + 1] == tr[k] and tr[k + 2] == tr[k + 1]) > 0)      # two rejections in a row
    assert folded_somewhere > 0


def test_folded_accept_tests_do_not_change_a_bit(smpl, omodel, gmodel):
    """Riding shapes: k_eval evaluates the COST of the queued speculative steps beside the trial point, and the solve launch that rejects
    the trial point takes the accept tests of the steps that would be rejected as well at once (avt_tuning.spec_cost; 0 = one launch
    pair per rejection).  Same tests on the same numbers in the same order: parameters, objective trace, damping
    and iteration counts are bit-identical with and without, on frames whose rejections come in runs of two to five, over several ICP
    iterations (the iteration budget is per ICP iteration), with both damping policies - and equal to the oracle's sequence."""
    from avatar_amd import api
    pm = synth.identity_part_map()
    for seed, policy, icp in ((8, 0, 1), (4, 0, 3), (5, 1, 2)):
        fr = synth.make_frame(smpl, seed)
        p0, q0, w0 = _start(fr)
        n = len(fr["labels"])
        opt = Options.counted(icp_iters=icp, lm_policy=policy)
        out = {}
        for sc in (0, 1):
            ctx = api.Context(gmodel, 24, pm, n, 1)
            ctx.set_tuning(spec_cost=sc)
            p, q, w, st = ctx.optimize_batch([fr["data"]], [fr["labels"]], opt, p0[None], q0[None], w0[None])
            # (the debug trace is indexed by the CUMULATIVE iteration count: 10 icp + 1 entries, the last ICP iteration's are the last eleven)
            out[sc] = (p, q, w, st[0].gn_iterations, st[0].accepted_steps, st[0].lambda_, st[0].final_cost, ctx.cost_trace(0, n=10 * icp + 1), ctx.cloud(0))
        for sc in (1,):
            for a, b in zip(out[0], out[sc]):
                assert np.array_equal(np.asarray(a), np.asarray(b)), (seed, sc)
        ref = omodel.optimize(pm, 24, fr["data"], fr["labels"], opt, p0, q0, w0, aggregate=1)
        assert out[1][3] == ref["stats"].gn_iterations == 10 * icp and out[1][4] == ref["stats"].accepted_steps
        assert np.abs(out[1][8] - ref["cloud"]).max() < 1e-6
        tr = out[1][7][-11:]
        assert [int(tr[k + 1] < tr[k]) for k in range(10)] == [int(a == 1) for a in ref["trace_acc"][-10:]]
        assert np.allclose(tr, ref["trace_cost"][-11:], rtol=1e-9, atol=0)
