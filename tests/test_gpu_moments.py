"""GPU: the moment form of the ICP data term (avatar_amd/csrc/avt_moments.hip, include/avt.h AVT_DATA_TERM_MOMENTS) against the
row form (k_eval: the residual / Jacobian rows rebuilt every iteration) and against the oracle's literal per-block formulas
(AvatarOptimizer.cpp:505-582, :609-644), through the C ABI."""
import numpy as np
import pytest

from avatar_amd import synth
from avatar_amd.capi import Options

pytestmark = pytest.mark.gpu


def _start(fr):
    from avatar_amd import api
    w0, p0, R0 = fr["start"]
    return p0, api.rot_to_quat(R0), w0


def _rel(a, b):
    return np.abs(a - b).max() / max(1e-300, np.abs(b).max())


@pytest.mark.parametrize("seed,dense", [(19, False), (3, False), (0, True)])
def test_moment_normal_equations_equal_the_row_form_and_the_oracle(smpl, omodel, gmodel, seed, dense):
    from avatar_amd import api
    pm = synth.identity_part_map()
    fr = synth.make_frame(smpl, seed, dense=dense)
    p0, q0, w0 = _start(fr)
    n = len(fr["labels"])
    opt = Options.demo(max_iters_per_icp=4)
    for frames in (1, 5):
        ctx = api.Context(gmodel, 24, pm, n, frames)
        ctx.set_data_term(ctx.DATA_TERM_ROWS)
        p, q, w, st = ctx.optimize_batch([fr["data"]] * frames, [fr["labels"]] * frames, opt, np.repeat(p0[None], frames, 0),
                                         np.repeat(q0[None], frames, 0), np.repeat(w0[None], frames, 0))
        f = frames - 1
        Hr, gr, cr = ctx.normal_equations(f)
        ctx.set_data_term(ctx.DATA_TERM_MOMENTS)          # makes the moments of the resident correspondences
        Hm, gm, cm = ctx.normal_equations(f)
        corr = ctx.correspondences(f, n)
        oc, og, oH, _ = omodel.evaluate(p[f], q[f], w[f], corr, fr["data"], 0.0, 0.0, aggregate=1)
        for name, H, g in (("rows", Hr, gr), ("moments", Hm, gm)):
            assert _rel(H, oH) < 1e-10, (name, _rel(H, oH))
            assert _rel(g, og) < 1e-9, (name, _rel(g, og))
            assert np.abs(H - H.T).max() == 0.0, name
        assert _rel(Hm, Hr) < 1e-11 and _rel(gm, gr) < 1e-9
        # structural zeros: a left-leg joint and a right-arm joint never share a model point
        assert np.all(Hm[3 + 3 * 4: 6 + 3 * 4, 3 + 3 * 19: 6 + 3 * 19] == 0.0)


def test_fit_with_moments_equals_fit_with_rows_and_oracle(smpl, omodel, gmodel):
    """Ten GN iterations with either form of the data term: same accept / reject sequence, same fit to 1e-7 (the parity bar is 1e-4)."""
    from avatar_amd import api
    pm = synth.identity_part_map()
    for seed in (0, 7):
        fr = synth.make_frame(smpl, seed)
        p0, q0, w0 = _start(fr)
        n = len(fr["labels"])
        opt = Options.demo(icp_iters=2)
        res = {}
        for form in (0, 1):
            ctx = api.Context(gmodel, 24, pm, n, 1)
            ctx.set_data_term(form)
            assert ctx.data_term() == form
            res[form] = ctx.optimize_batch([fr["data"]], [fr["labels"]], opt, p0[None], q0[None], w0[None])
        ref = omodel.optimize(pm, 24, fr["data"], fr["labels"], opt, p0, q0, w0, aggregate=1)
        for form in (0, 1):
            p, q, w, st = res[form]
            assert st[0].accepted_steps == ref["stats"].accepted_steps and st[0].gn_iterations == ref["stats"].gn_iterations
            assert np.abs(p[0] - ref["p"]).max() < 1e-7 and np.abs(q[0] - ref["q"]).max() < 1e-7 and np.abs(w[0] - ref["w"]).max() < 1e-6
            assert abs(st[0].final_cost - ref["stats"].final_cost) < 1e-8 * ref["stats"].final_cost
        assert np.abs(res[0][0] - res[1][0]).max() < 1e-9 and np.abs(res[0][1] - res[1][1]).max() < 1e-9


def test_moments_batch_matches_single(smpl, gmodel):
    """A batch of different frames through the moment form equals the same frames one at a time, bit for bit."""
    from avatar_amd import api
    pm = synth.identity_part_map()
    frs = [synth.make_frame(smpl, s) for s in (21, 22, 23, 24, 25, 26, 27)]
    starts = [_start(fr) for fr in frs]
    opt = Options.demo(max_iters_per_icp=6)
    ctx = api.Context(gmodel, 24, pm, 60000, len(frs))
    assert ctx.data_term() == ctx.DATA_TERM_AUTO
    ctx.set_data_term(ctx.DATA_TERM_MOMENTS)
    P, Q, W, st = ctx.optimize_batch([fr["data"] for fr in frs], [fr["labels"] for fr in frs], opt, np.array([s[0] for s in starts]),
                                     np.array([s[1] for s in starts]), np.array([s[2] for s in starts]))
    one = api.Context(gmodel, 24, pm, 60000, 1)
    one.set_data_term(one.DATA_TERM_MOMENTS)
    for i, fr in enumerate(frs):
        p, q, w, s1 = one.optimize_batch([fr["data"]], [fr["labels"]], opt, starts[i][0][None], starts[i][1][None], starts[i][2][None])
        assert np.array_equal(p[0], P[i]) and np.array_equal(q[0], Q[i]) and np.array_equal(w[0], W[i])


def test_tuning_is_a_structure_not_the_environment(smpl, gmodel):
    """include/avt.h avt_tuning: read back, changed through the setter, validated; the AUTO data term follows mom_min_frames."""
    from avatar_amd import api, capi
    pm = synth.identity_part_map()
    ctx = api.Context(gmodel, 24, pm, 60000, 4)
    t = ctx.tuning()
    assert t.as_dict() == {k: v for k, v in capi.Tuning.DEFAULTS.items() if k != "reserved"} and t.non_default() == {}
    ctx.set_tuning(nspec=2, mom_min_frames=2)
    assert ctx.tuning().non_default() == {"nspec": 2, "mom_min_frames": 2}
    with pytest.raises(api.AvtError):
        ctx.set_tuning(ride_strips=5)
    with pytest.raises(KeyError):
        ctx.set_tuning(no_such_knob=1)
    # four frames with the moment form from two frames per launch on == the same frames with the form selected explicitly
    frs = [synth.make_frame(smpl, 30 + s) for s in range(4)]
    starts = [_start(fr) for fr in frs]
    args = ([fr["data"] for fr in frs], [fr["labels"] for fr in frs], Options.demo(max_iters_per_icp=4), np.array([s[0] for s in starts]),
            np.array([s[1] for s in starts]), np.array([s[2] for s in starts]))
    a = ctx.optimize_batch(*args)
    ctx2 = api.Context(gmodel, 24, pm, 60000, 4)
    ctx2.set_data_term(ctx2.DATA_TERM_MOMENTS)
    b = ctx2.optimize_batch(*args)
    assert all(np.array_equal(x, y) for x, y in zip(a[:3], b[:3]))
    ctx3 = api.Context(gmodel, 24, pm, 60000, 4)      # default: rows for four frames - close, not bit-equal
    c = ctx3.optimize_batch(*args)
    assert np.abs(c[0] - a[0]).max() < 1e-9 and not np.array_equal(c[0], a[0])


def test_moment_form_on_other_model_shapes(smpl):
    """The run-time-K instantiations (k_pairpass<0>, k_assemble<0>; SMPL's K = 10 has its own) and a model whose points carry one joint
    each (limit_one_joint_per_point: only diagonal pairs): normal equations equal to the row form's and the oracle's, the fit equal to the oracle's."""
    from avatar_amd import api
    from oracle import oracle as orc
    pm = synth.identity_part_map()
    fr = synth.make_frame(smpl, 12)
    p0, q0, _ = _start(fr)
    n = len(fr["labels"])
    small = dict(smpl)
    small["shapedirs"] = np.ascontiguousarray(np.asarray(smpl["shapedirs"])[:, :, :7])
    for model, limit, K in ((small, False, 7), (smpl, True, 10)):
        gm = api.AvatarModel(model, limit_one_joint_per_point=limit)
        om = orc.OracleModel(model, limit_one_joint_per_point=limit)
        w0 = np.asarray(fr["start"][0])[:K]
        opt = Options.demo(max_iters_per_icp=5)
        ctx = api.Context(gm, 24, pm, n, 2)
        ctx.set_data_term(ctx.DATA_TERM_MOMENTS)
        p, q, w, st = ctx.optimize_batch([fr["data"]] * 2, [fr["labels"]] * 2, opt, np.repeat(p0[None], 2, 0), np.repeat(q0[None], 2, 0), np.repeat(w0[None], 2, 0))
        ref = om.optimize(pm, 24, fr["data"], fr["labels"], opt, p0, q0, w0, aggregate=1)
        assert st[1].accepted_steps == ref["stats"].accepted_steps
        assert np.abs(p[1] - ref["p"]).max() < 1e-7 and np.abs(q[1] - ref["q"]).max() < 1e-7 and np.abs(w[1] - ref["w"]).max() < 1e-6
        Hm, gm_, _ = ctx.normal_equations(1)
        ctx.set_data_term(ctx.DATA_TERM_ROWS)
        Hr, gr, _ = ctx.normal_equations(1)
        corr = ctx.correspondences(1, n)
        _, og, oH, _ = om.evaluate(p[1], q[1], w[1], corr, fr["data"], 0.0, 0.0, aggregate=1)
        assert _rel(Hm, oH) < 1e-10 and _rel(gm_, og) < 1e-9 and _rel(Hm, Hr) < 1e-11 and _rel(gm_, gr) < 1e-9, (K, limit)


def test_moment_form_ragged_batch_with_empty_frame(smpl, omodel, gmodel):
    """An empty frame and a sparse one inside a batch through the moment form: the empty frame is left untouched, the others fit like the oracle."""
    from avatar_amd import api
    pm = synth.identity_part_map()
    frs = [synth.make_frame(smpl, s) for s in (15, 16)]
    datas = [frs[0]["data"], np.zeros((0, 3)), frs[1]["data"][:777]]
    labs = [frs[0]["labels"], np.zeros(0, np.int32), frs[1]["labels"][:777]]
    starts = [_start(frs[0]), _start(frs[0]), _start(frs[1])]
    opt = Options.demo(max_iters_per_icp=5)
    ctx = api.Context(gmodel, 24, pm, 60000, 3)
    ctx.set_data_term(ctx.DATA_TERM_MOMENTS)
    args = (datas, labs, opt, np.array([s[0] for s in starts]), np.array([s[1] for s in starts]), np.array([s[2] for s in starts]))
    p1, q1, w1, st1 = ctx.optimize_batch(*args)
    p2, q2, w2, st2 = ctx.optimize_batch(*args)
    assert np.array_equal(p1, p2) and np.array_equal(q1, q2) and np.array_equal(w1, w2)      # bit-wise reproducible
    assert st1[1].num_correspondences == 0 and np.array_equal(p1[1], starts[1][0]) and np.array_equal(w1[1], starts[1][2])
    for f in (0, 2):
        ref = omodel.optimize(pm, 24, datas[f], labs[f], opt, *starts[f], aggregate=1)
        assert st1[f].accepted_steps == ref["stats"].accepted_steps
        assert np.abs(p1[f] - ref["p"]).max() < 1e-6 and np.abs(w1[f] - ref["w"]).max() < 1e-5


def test_switching_the_form_back_and_forth_reproduces_the_system_and_the_cost(smpl, gmodel):
    """avt_get_normal_equations makes what the selected form needs on demand; the two forms' cost constants share one buffer,
    so every switch must rebuild it: rows -> moments -> rows -> moments gives the same system and cost each time."""
    from avatar_amd import api
    pm = synth.identity_part_map()
    fr = synth.make_frame(smpl, 11)
    p0, q0, w0 = _start(fr)
    n = len(fr["labels"])
    ctx = api.Context(gmodel, 24, pm, n, 3)
    ctx.set_data_term(ctx.DATA_TERM_ROWS)
    ctx.optimize_batch([fr["data"]] * 3, [fr["labels"]] * 3, Options.demo(max_iters_per_icp=3), np.repeat(p0[None], 3, 0), np.repeat(q0[None], 3, 0), np.repeat(w0[None], 3, 0))
    seen = {}
    for rnd in range(2):
        for name, term in (("rows", ctx.DATA_TERM_ROWS), ("moments", ctx.DATA_TERM_MOMENTS)):
            ctx.set_data_term(term)
            H, g, cost = ctx.normal_equations(1)
            if name in seen:
                assert np.array_equal(H, seen[name][0]) and np.array_equal(g, seen[name][1]) and cost == seen[name][2], (name, rnd)
            seen[name] = (H, g, cost)
    assert abs(seen["moments"][2] - seen["rows"][2]) <= 1e-9 * abs(seen["rows"][2])


def test_moment_form_launch_shapes_agree_bit_for_bit(smpl, gmodel):
    """The pose prior of the moment form rides in the pair pass's grid up to 256 frames per launch and is a launch of its own above
    (avt_moments.hip launch_assemble); a frame's fit must not depend on which: 260 frames in ONE launch group, 260 frames in two groups
    of 130 and the frames one at a time give the same bits (every second point of the clouds: the volume of the 130-frame test this was)."""
    from avatar_amd import api
    pm = synth.identity_part_map()
    frs = [synth.make_frame(smpl, s) for s in (41, 42, 43)]
    for fr in frs:
        fr["data"] = np.ascontiguousarray(fr["data"][::2]); fr["labels"] = np.ascontiguousarray(fr["labels"][::2])
    starts = [_start(fr) for fr in frs]
    opt = Options.demo(max_iters_per_icp=4)
    n = max(len(fr["labels"]) for fr in frs)
    F = 260
    pick = [i % 3 for i in range(F)]
    args = ([frs[i]["data"] for i in pick], [frs[i]["labels"] for i in pick], opt, np.array([starts[i][0] for i in pick]),
            np.array([starts[i][1] for i in pick]), np.array([starts[i][2] for i in pick]))
    res = []
    for groups in (1, 2):
        ctx = api.Context(gmodel, 24, pm, n, F)
        ctx.set_tuning(groups=groups)
        ctx.set_data_term(ctx.DATA_TERM_MOMENTS)
        res.append(ctx.optimize_batch(*args))
        del ctx
    one = api.Context(gmodel, 24, pm, n, 1)
    one.set_data_term(one.DATA_TERM_MOMENTS)
    for i in range(3):
        p, q, w, _ = one.optimize_batch([frs[i]["data"]], [frs[i]["labels"]], opt, starts[i][0][None], starts[i][1][None], starts[i][2][None])
        for r in res:
            for j in (i, i + 3, F - 3 + i):      # first, second and last occurrence of the frame in the batch (pick[F - 3 + i] == (F - 3 + i) % 3)
                if pick[j] != i:
                    continue
                assert np.array_equal(r[0][j], p[0]) and np.array_equal(r[1][j], q[0]) and np.array_equal(r[2][j], w[0]), (i, j)


def test_batch_with_three_icp_iterations_the_forms_agree(smpl, gmodel):
    """66 frames (two launch groups of 33: the default takes the moment form there, the prior's workgroups ride in the pair pass), three ICP
    iterations: the default equals the moment form selected explicitly bit for bit, and the row form to 1e-10."""
    from avatar_amd import api
    pm = synth.identity_part_map()
    frs = [synth.make_frame(smpl, 50 + s) for s in range(4)]
    F = 66
    pick = [i % 4 for i in range(F)]
    st = [_start(fr) for fr in frs]
    args = ([frs[i]["data"] for i in pick], [frs[i]["labels"] for i in pick], Options.demo(icp_iters=3), np.array([st[i][0] for i in pick]),
            np.array([st[i][1] for i in pick]), np.array([st[i][2] for i in pick]))
    n = max(len(fr["labels"]) for fr in frs)
    res = {}
    for name, term in (("auto", None), ("rows", 0), ("moments", 1)):
        ctx = api.Context(gmodel, 24, pm, n, F)
        if term is not None:
            ctx.set_data_term(term)
        res[name] = ctx.optimize_batch(*args)
        del ctx
    assert all(np.array_equal(a, b) for a, b in zip(res["auto"][:3], res["moments"][:3]))
    assert [s.accepted_steps for s in res["rows"][3]] == [s.accepted_steps for s in res["moments"][3]]
    for a, b in zip(res["rows"][:3], res["moments"][:3]):
        assert np.abs(a - b).max() < 1e-10


@pytest.mark.gpu
def test_dense_batch_on_the_moment_form_is_the_oracles_fit(smpl, omodel, gmodel):
    """What bench.py's `dense_64` leg times: a batch of DENSE frames (2560x1440 renders, ~150 k points) as two frame
    groups, AUTO => the moment form - checked as a FIT against the oracle, not only through avt_get_normal_equations: frames at both
    ends of both groups have bit-exact correspondences, the oracle's accept / reject sequence and objective after every GN
    iteration, and its vertices to 1e-6 (AvatarOptimizer.cpp:505-582 is what the moments replace)."""
    from avatar_amd import api
    F = 64
    pm = synth.identity_part_map()
    gts = [synth.sample_ground_truth(smpl, 3000 + f) for f in range(F)]
    starts = [synth.perturb_start(*gts[f], 3000 + f) for f in range(F)]
    ctx = api.Context(gmodel, 24, pm, 200000, F, device=0)
    npts = ctx.render_frames(np.array([g[0] for g in gts]), np.array([g[1] for g in gts]), np.array([g[2] for g in gts]), res_scale=2)
    assert npts.min() > 60000
    p0 = np.array([s[1] for s in starts]); w0 = np.array([s[0] for s in starts])
    q0 = api.rot_to_quat(np.array([s[2] for s in starts]).reshape(-1, 3, 3)).reshape(F, 24, 4)
    ctx.state_upload(p0, q0, w0)
    opt = Options.counted()
    assert ctx.data_term() == ctx.DATA_TERM_AUTO and ctx.launch_shape()[:2] == (2, 32)      # two groups of 32 frames per launch: AUTO => moments
    ctx.optimize_resident(opt)
    p, q, w, st = ctx.state_download()
    assert ctx.mfma_count(0)["eval_rows"] is None and ctx.mfma_count(0)["moments"] > 0      # the moment form ran (no matched-point records exist)
    for i in (0, 31, 32, 63):
        d, l = ctx.frame_download(i)
        ref = omodel.optimize(pm, 24, d, l, opt, p0[i], q0[i], w0[i], aggregate=1)
        assert np.array_equal(ctx.correspondences(i, len(l)), ref["corr"])
        assert np.abs(ctx.cloud(i) - ref["cloud"]).max() < 1e-6
        tr = ctx.cost_trace(i)
        assert np.allclose(tr, ref["trace_cost"][:11], rtol=1e-9, atol=0)
        assert [int(tr[k + 1] < tr[k]) for k in range(10)] == [int(a == 1) for a in ref["trace_acc"][:10]]
        assert st[i].accepted_steps == ref["stats"].accepted_steps and st[i].gn_iterations == 10
        assert abs(st[i].lambda_ - ref["stats"].lambda_) <= 1e-9 * abs(ref["stats"].lambda_)      # (gain ratio: lambda follows a ratio of cost differences)
        assert np.abs(p[i] - ref["p"]).max() < 1e-7 and np.abs(w[i] - ref["w"]).max() < 1e-6


def test_normal_equations_after_a_replayed_graph_are_made_from_fresh_moments(smpl, gmodel):
    """what exists for the resident correspondences (moments / matched-point records) must describe the last optimize()
    that RAN, also when it ran as the replay of a cached hipGraph.  rows(S1) -> moments(S1) -> rows(S2) replays the rows graph; the
    moments in memory then belong to S1's correspondences, and avt_get_normal_equations(MOMENTS) has to rebuild them: it must give what a
    fresh context gives for rows(S2) -> MOMENTS."""
    from avatar_amd import api
    pm = synth.identity_part_map()
    fr = synth.make_frame(smpl, 5)
    p0, q0, w0 = _start(fr)
    n = len(fr["labels"])
    opt = Options.demo(max_iters_per_icp=3)
    p2 = p0 + np.array([0.03, -0.02, 0.01])          # a second start state: other correspondences, same launch shape
    ctx = api.Context(gmodel, 24, pm, n, 1)
    ctx.frames_upload([fr["data"]], [fr["labels"]])
    for form, p in ((ctx.DATA_TERM_ROWS, p0), (ctx.DATA_TERM_MOMENTS, p0), (ctx.DATA_TERM_ROWS, p2)):
        ctx.set_data_term(form)
        ctx.state_upload(p[None], q0[None], w0[None])
        ctx.optimize_resident(opt)
    corr_s2 = ctx.correspondences(0, n)
    ctx.set_data_term(ctx.DATA_TERM_MOMENTS)
    H1, g1, c1 = ctx.normal_equations(0)
    fresh = api.Context(gmodel, 24, pm, n, 1)
    fresh.frames_upload([fr["data"]], [fr["labels"]])
    fresh.set_data_term(fresh.DATA_TERM_ROWS)
    fresh.state_upload(p2[None], q0[None], w0[None])
    fresh.optimize_resident(opt)
    assert np.array_equal(fresh.correspondences(0, n), corr_s2)
    ctx0 = api.Context(gmodel, 24, pm, n, 1)
    ctx0.frames_upload([fr["data"]], [fr["labels"]]); ctx0.set_data_term(ctx0.DATA_TERM_ROWS)
    ctx0.state_upload(p0[None], q0[None], w0[None]); ctx0.optimize_resident(opt)
    assert not np.array_equal(ctx0.correspondences(0, n), corr_s2), "the two start states must lead to different correspondences for this test to see anything"
    fresh.set_data_term(fresh.DATA_TERM_MOMENTS)
    H2, g2, c2 = fresh.normal_equations(0)
    assert np.array_equal(H1, H2) and np.array_equal(g1, g2) and c1 == c2


def test_assembly_as_role_workgroups_gives_the_one_workgroup_assembly_bit_for_bit(smpl, gmodel):
    """k_assemble_parts (six independent 256-thread role workgroups per frame, the default) computes every entry of the system by the
    item of k_assemble (one 1024-thread workgroup) that computes it there: same operands, same order, same bits - for the system
    itself and for a fit of several ICP iterations, one frame and a batch."""
    from avatar_amd import api
    pm = synth.identity_part_map()
    frs = [synth.make_frame(smpl, 40 + s) for s in range(5)]
    nmax = max(len(f["labels"]) for f in frs)
    p0 = np.array([f["start"][1] for f in frs]); q0 = np.array([api.rot_to_quat(f["start"][2]) for f in frs]); w0 = np.array([f["start"][0] for f in frs])
    opt = Options.demo(icp_iters=2, max_iters_per_icp=5)
    res = {}
    for parts in (0, 1):
        ctx = api.Context(gmodel, 24, pm, nmax, len(frs))
        ctx.set_data_term(ctx.DATA_TERM_MOMENTS)
        ctx.set_tuning(asm_parts=parts)
        assert ctx.tuning().asm_parts == parts
        out = ctx.optimize_batch([f["data"] for f in frs], [f["labels"] for f in frs], opt, p0, q0, w0)
        res[parts] = (out, [ctx.normal_equations(i) for i in (0, 4)], [ctx.cost_trace(i) for i in range(len(frs))])
    (pa, qa, wa, sta), nea, tra = res[0]
    (pb, qb, wb, stb), neb, trb = res[1]
    assert np.array_equal(pa, pb) and np.array_equal(qa, qb) and np.array_equal(wa, wb)
    for (Ha, ga, ca), (Hb, gb, cb) in zip(nea, neb):
        assert np.array_equal(Ha, Hb) and np.array_equal(ga, gb) and ca == cb
        assert np.abs(Hb - Hb.T).max() == 0.0
    assert all(np.array_equal(a, b) for a, b in zip(tra, trb))
    assert [s.accepted_steps for s in sta] == [s.accepted_steps for s in stb]


def test_skinning_several_frames_per_workgroup_gives_the_same_bits(smpl, gmodel):
    """k_lbs_multi (frame batches: a thread loads a vertex's shape planes and weights once and skins 2 / 4 frames with them) against
    k_lbs (one frame per workgroup): same operations in the same order per frame - clouds, joint positions and the whole fit equal bit
    for bit, with a frame count that is not a multiple of the frames per workgroup."""
    from avatar_amd import api
    pm = synth.identity_part_map()
    frs = [synth.make_frame(smpl, 60 + s) for s in range(7)]
    sel = slice(0, None, 4)
    datas = [f["data"][sel] for f in frs]; labs = [f["labels"][sel] for f in frs]
    nmax = max(len(l) for l in labs)
    p0 = np.array([f["start"][1] for f in frs]); q0 = np.array([api.rot_to_quat(f["start"][2]) for f in frs]); w0 = np.array([f["start"][0] for f in frs])
    opt = Options.demo(icp_iters=2, max_iters_per_icp=4)
    res = {}
    for ft in (1, 2, 4):
        ctx = api.Context(gmodel, 24, pm, nmax, len(frs))
        ctx.set_data_term(ctx.DATA_TERM_MOMENTS)
        ctx.set_tuning(lbs_frames=ft)
        out = ctx.optimize_batch(datas, labs, opt, p0, q0, w0)
        res[ft] = (out, [ctx.cloud(i) for i in range(len(frs))], [ctx.posed(i) for i in (0, 6)], [ctx.correspondences(i, len(labs[i])) for i in range(len(frs))])
    for ft in (2, 4):
        for a, b in zip(res[1][0][:3], res[ft][0][:3]):
            assert np.array_equal(a, b), ft
        assert all(np.array_equal(a, b) for a, b in zip(res[1][1], res[ft][1])), ft
        for pa, pb in zip(res[1][2], res[ft][2]):
            assert all(np.array_equal(a, b) for a, b in zip(pa, pb)), ft
        assert all(np.array_equal(a, b) for a, b in zip(res[1][3], res[ft][3])), ft


@pytest.mark.parametrize("frames,form", [(21, 1), (53, 1), (13, 0)])
def test_frames_mapped_to_xcds_give_the_same_bits(smpl, gmodel, frames, form):
    """avt_tuning.xcd_frames (frame-batch kernels take their (frame, block) from a remap of the grid that keeps a frame's workgroups on one
    XCD, avt_device.h xcd_frame_block / xcd_frame_1d) against the grid order: a permutation of which workgroup does what - every result equal
    bit for bit, with launch sizes that are not multiples of 8 (the remap's uneven split) and with two frame groups."""
    from avatar_amd import api
    pm = synth.identity_part_map()
    frs = [synth.make_frame(smpl, 80 + (s % 5)) for s in range(frames)]
    sel = slice(0, None, 6)
    datas = [f["data"][sel] for f in frs]; labs = [f["labels"][sel] for f in frs]
    nmax = max(len(l) for l in labs)
    rng = np.random.default_rng(5)
    p0 = np.array([f["start"][1] for f in frs]) + 0.01 * rng.standard_normal((frames, 3))
    q0 = np.array([api.rot_to_quat(f["start"][2]) for f in frs]); w0 = np.array([f["start"][0] for f in frs])
    opt = Options.demo(icp_iters=2, max_iters_per_icp=3)
    res = {}
    for xf in (0, 1):
        ctx = api.Context(gmodel, 24, pm, nmax, frames)
        ctx.set_data_term(ctx.DATA_TERM_MOMENTS if form else ctx.DATA_TERM_ROWS)
        ctx.set_tuning(xcd_frames=xf)
        assert ctx.tuning().xcd_frames == xf
        out = ctx.optimize_batch(datas, labs, opt, p0, q0, w0)
        res[xf] = (out, [ctx.cloud(i) for i in range(frames)], [ctx.correspondences(i, len(labs[i])) for i in range(frames)],
                   [(s.accepted_steps, s.final_cost, s.lambda_) for s in out[3]])
    for a, b in zip(res[0][0][:3], res[1][0][:3]):
        assert np.array_equal(a, b)
    assert all(np.array_equal(a, b) for a, b in zip(res[0][1], res[1][1]))
    assert all(np.array_equal(a, b) for a, b in zip(res[0][2], res[1][2]))
    assert res[0][3] == res[1][3]
    assert len({tuple(x) for x in res[1][0][0].round(12).tolist()}) > 1        # the frames are not all the same frame
