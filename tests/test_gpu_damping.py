"""GPU: the gain-ratio damping schedule (avt_options.lm_policy = 1) runs in lock step with the oracle's - on the
few-frames launch shapes (speculative steps, accept test inside k_lbs), on a frame batch, and through the moment form."""
import numpy as np
import pytest

from avatar_amd import synth
from avatar_amd.capi import Options

pytestmark = pytest.mark.gpu


def _start(fr):
    from avatar_amd import api
    w0, p0, R0 = fr["start"]
    return p0, api.rot_to_quat(R0), w0


@pytest.mark.parametrize("frames,form,lm_up", [(1, 0, None), (2, 0, None), (5, 0, None), (5, 1, None), (9, 0, None), (1, 0, 2.0), (5, 1, 2.0)])
def test_gain_ratio_schedule_matches_oracle(smpl, omodel, gmodel, frames, form, lm_up):
    from avatar_amd import api
    pm = synth.identity_part_map()
    frs = [synth.make_frame(smpl, 40 + s) for s in range(frames)]
    starts = [_start(fr) for fr in frs]
    opt = Options.counted(icp_iters=2, lm_policy=1) if lm_up is None else Options.counted(icp_iters=2, lm_policy=1, lm_up=lm_up, lm_down=0.25)      # (None: Options.GAIN_LM_UP)
    ctx = api.Context(gmodel, 24, pm, 60000, frames)
    ctx.set_data_term(form)
    P, Q, W, st = ctx.optimize_batch([fr["data"] for fr in frs], [fr["labels"] for fr in frs], opt, np.array([s[0] for s in starts]),
                                     np.array([s[1] for s in starts]), np.array([s[2] for s in starts]))
    fixed = Options.counted(icp_iters=2, lm_policy=0)
    differs = 0
    for i, fr in enumerate(frs):
        ref = omodel.optimize(pm, 24, fr["data"], fr["labels"], opt, *starts[i], aggregate=1)
        assert st[i].accepted_steps == ref["stats"].accepted_steps and st[i].gn_iterations == 20, (i, st[i].accepted_steps, ref["stats"].accepted_steps)
        assert np.abs(P[i] - ref["p"]).max() < 1e-7 and np.abs(Q[i] - ref["q"]).max() < 1e-7 and np.abs(W[i] - ref["w"]).max() < 1e-6
        assert abs(st[i].final_cost - ref["stats"].final_cost) < 1e-8 * ref["stats"].final_cost
        assert abs(st[i].lambda_ - ref["stats"].lambda_) < 1e-6 * ref["stats"].lambda_
        differs += ref["stats"].accepted_steps != omodel.optimize(pm, 24, fr["data"], fr["labels"], fixed, *starts[i], aggregate=1)["stats"].accepted_steps
    assert differs > 0        # the schedule is not the fixed-factor one in disguise


def test_gain_ratio_schedule_is_reproducible_and_rejects_bad_policy(smpl, gmodel):
    from avatar_amd import api
    pm = synth.identity_part_map()
    fr = synth.make_frame(smpl, 3)
    p0, q0, w0 = _start(fr)
    ctx = api.Context(gmodel, 24, pm, 60000, 1)
    opt = Options.demo(lm_policy=1)
    a = ctx.optimize_batch([fr["data"]], [fr["labels"]], opt, p0[None], q0[None], w0[None])
    b = ctx.optimize_batch([fr["data"]], [fr["labels"]], opt, p0[None], q0[None], w0[None])
    assert all(np.array_equal(x, y) for x, y in zip(a[:3], b[:3]))
    for bad in (dict(lm_policy=7), dict(lm_up=1.0), dict(lm_down=1.0), dict(lm_policy=1, lm_down=0.0)):
        with pytest.raises(api.AvtError):
            ctx.optimize_batch([fr["data"]], [fr["labels"]], Options.demo(**bad), p0[None], q0[None], w0[None])
