"""GPU: the reference's stopping rule (options.function_tolerance = 1e-4, AvatarOptimizer.cpp:1333; avt_options.function_tolerance) in lock
step with the oracle's: an accepted step whose decrease is at most that fraction of the objective ends the frame's Gauss-Newton iterations
of the ICP iteration - on the riding one-frame shape (speculative steps, folded accept tests), the few-frames row form with its own
reduction launch, and frame batches on the moment form, where some frames of a launch stop while their neighbours go on."""
import numpy as np
import pytest

from avatar_amd import synth
from avatar_amd.capi import Options

pytestmark = pytest.mark.gpu


def _start(fr):
    from avatar_amd import api
    w0, p0, R0 = fr["start"]
    return p0, api.rot_to_quat(R0), w0


def _frames(smpl, n, first=0, step=4):
    frs = [synth.make_frame(smpl, first + s) for s in range(n)]
    return [fr["data"][::step] for fr in frs], [fr["labels"][::step] for fr in frs], [_start(fr) for fr in frs]


def _check_against_oracle(omodel, pm, opt, datas, labels, starts, P, Q, W, st, which):
    stopped = 0
    for i in which:
        ref = omodel.optimize(pm, 24, datas[i], labels[i], opt, *starts[i], aggregate=1)
        budget = opt.icp_iters * opt.max_iters_per_icp
        assert st[i].gn_iterations == ref["stats"].gn_iterations <= budget, (i, st[i].gn_iterations, ref["stats"].gn_iterations)
        assert st[i].accepted_steps == ref["stats"].accepted_steps
        assert np.abs(P[i] - ref["p"]).max() < 1e-7 and np.abs(Q[i] - ref["q"]).max() < 1e-7 and np.abs(W[i] - ref["w"]).max() < 1e-6
        assert abs(st[i].final_cost - ref["stats"].final_cost) < 1e-8 * ref["stats"].final_cost
        assert abs(st[i].lambda_ - ref["stats"].lambda_) < 1e-6 * ref["stats"].lambda_
        stopped += ref["stats"].gn_iterations < budget
    return stopped


@pytest.mark.parametrize("frames,form,tol,policy", [(1, 0, 1e-4, 1), (1, 0, 1e-2, 1), (1, 0, 1e-2, 0), (2, 0, 1e-2, 1), (5, 0, 1e-4, 1), (5, 0, 1e-2, 1), (5, 1, 1e-2, 1), (9, 0, 1e-2, 1)])
def test_stopping_rule_matches_oracle(smpl, omodel, gmodel, frames, form, tol, policy):
    from avatar_amd import api
    pm = synth.identity_part_map()
    datas, labels, starts = _frames(smpl, frames, first=0)
    opt = Options.demo(icp_iters=3, function_tolerance=tol, lm_policy=policy)
    ctx = api.Context(gmodel, 24, pm, 20000, frames)
    ctx.set_data_term(form)
    P, Q, W, st = ctx.optimize_batch(datas, labels, opt, np.array([s[0] for s in starts]), np.array([s[1] for s in starts]), np.array([s[2] for s in starts]))
    stopped = _check_against_oracle(omodel, pm, opt, datas, labels, starts, P, Q, W, st, range(frames))
    assert stopped > 0, "no frame of this case met the rule: the test exercises nothing"
    # the rule off on the same context (same graph key, other option block): the full budget again
    off = Options.counted(icp_iters=3, lm_policy=policy)
    _, _, _, st0 = ctx.optimize_batch(datas, labels, off, np.array([s[0] for s in starts]), np.array([s[1] for s in starts]), np.array([s[2] for s in starts]))
    assert all(s.gn_iterations == 30 for s in st0)
    # and on again: bit-identical to the first run (nothing of the idle launches leaks into the next call)
    P2, Q2, W2, st2 = ctx.optimize_batch(datas, labels, opt, np.array([s[0] for s in starts]), np.array([s[1] for s in starts]), np.array([s[2] for s in starts]))
    assert np.array_equal(P, P2) and np.array_equal(Q, Q2) and np.array_equal(W, W2) and [s.gn_iterations for s in st] == [s.gn_iterations for s in st2]


def test_stopping_rule_in_a_frame_batch_of_two_groups(smpl, omodel, gmodel):
    """70 frames = two frame groups on the moment form (AUTO): frames that stop early share their launches with frames that do not."""
    from avatar_amd import api
    F = 70
    pm = synth.identity_part_map()
    datas, labels, starts = _frames(smpl, F, first=100, step=5)
    opt = Options.demo(icp_iters=2)      # function_tolerance = 1e-4, the reference's
    ctx = api.Context(gmodel, 24, pm, 16384, F)
    p0 = np.array([s[0] for s in starts]); q0 = np.array([s[1] for s in starts]); w0 = np.array([s[2] for s in starts])
    P, Q, W, st = ctx.optimize_batch(datas, labels, opt, p0, q0, w0)
    assert ctx.launch_shape()[0] == 2 and ctx.mfma_count(0)["moments"] > 0
    its = np.array([s.gn_iterations for s in st])
    assert its.max() <= 20 and its.min() >= 2
    # the frames with the fewest and the most iterations, and both ends of both groups, against the oracle
    which = sorted(set([0, 34, 35, 69, int(np.argmin(its)), int(np.argmax(its))]))
    _check_against_oracle(omodel, pm, opt, datas, labels, starts, P, Q, W, st, which)
    # a looser tolerance: most frames stop in the first ICP iteration already
    loose = Options.demo(icp_iters=2, function_tolerance=2e-2)
    P, Q, W, st = ctx.optimize_batch(datas, labels, loose, p0, q0, w0)
    its2 = np.array([s.gn_iterations for s in st])
    assert (its2 < 20).sum() > F // 2 and (its2 <= its).all()
    _check_against_oracle(omodel, pm, loose, datas, labels, starts, P, Q, W, st, [0, 35, 69, int(np.argmin(its2))])


def test_stopping_rule_rejects_bad_tolerance(smpl, gmodel):
    from avatar_amd import api
    pm = synth.identity_part_map()
    datas, labels, starts = _frames(smpl, 1)
    ctx = api.Context(gmodel, 24, pm, 20000, 1)
    for bad in (-1e-4, 1.0, float("nan")):
        with pytest.raises(api.AvtError):
            ctx.optimize_batch(datas, labels, Options.demo(function_tolerance=bad), starts[0][0][None], starts[0][1][None], starts[0][2][None])
