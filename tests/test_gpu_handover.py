"""The in-launch hand-over of the few-frames launch shape (reduction workgroups riding in k_solve's grid, avt_lm.hip) must
never turn a scheduling accident into a silently different fit: either the bits are the bits of an undisturbed run, or the
call that hands out the result fails with AVT_STATUS_DEVICE_FAULT."""
import os

import numpy as np
import pytest

from avatar_amd import synth
from avatar_amd.capi import Options

pytestmark = pytest.mark.gpu

SEEDS = {1: (4,), 2: (5, 6), 3: (0, 5, 8)}


def _ctx(api, gmodel, pm, F):
    """The hand-over belongs to the ROW form of the data term (k_eval's partial tiles reduced inside k_solve's launch); the moment
    form, the default, has no partial tiles and no hand-over."""
    ctx = api.Context(gmodel, 24, pm, 60000, F)
    ctx.set_data_term(ctx.DATA_TERM_ROWS)
    return ctx


def _run(ctx, api, frames, opt):
    p, q, w, st = ctx.optimize_batch([f["data"] for f in frames], [f["labels"] for f in frames], opt,
                                     np.array([f["start"][1] for f in frames]), np.array([api.rot_to_quat(f["start"][2]) for f in frames]),
                                     np.array([f["start"][0] for f in frames]))
    return np.concatenate([p.ravel(), q.ravel(), w.ravel(), np.array([x.final_cost for x in st]), np.array([x.accepted_steps for x in st], float)])


def test_a_solver_that_gives_up_waiting_is_an_error_not_a_different_fit(smpl, gmodel):
    """avt_tuning.ride_timeout_us = 0: a solver role whose reduction has not yet delivered gives up at
    once.  Every call then either fails with status 3 or - the reduction happened to be there - returns the undisturbed bits."""
    from avatar_amd import api
    pm = synth.identity_part_map()
    opt = Options.demo(icp_iters=2)
    faults = clean = 0
    for F in (1, 2, 3):
        frames = [synth.make_frame(smpl, s) for s in SEEDS[F]]
        good = _run(_ctx(api, gmodel, pm, F), api, frames, opt)
        ctx = _ctx(api, gmodel, pm, F).set_tuning(ride_timeout_us=0)
        for _ in range(4):
            try:
                out = _run(ctx, api, frames, opt)
            except api.AvtError as e:
                assert e.status == 3 and "fault" in str(e)
                faults += 1
                continue
            assert np.array_equal(out, good)
            clean += 1
    assert faults > 0, "the timeout path was never taken: the test does not exercise it"
    # a fault is reported once and cleared: an ordinary context on the same device is unaffected
    frames = [synth.make_frame(smpl, s) for s in SEEDS[1]]
    ctx = _ctx(api, gmodel, pm, 1)
    assert np.array_equal(_run(ctx, api, frames, opt), _run(ctx, api, frames, opt))


def test_few_frame_shapes_under_compute_pressure_from_another_stream(smpl, gmodel):
    """The one-, two- and three-frame shapes while another stream keeps the whole chip busy (large fp32 matrix products queued
    through torch on a side stream: every CU holds their workgroups, the riding launch's workgroups get CUs as they come free).
    Results are bit-identical to the undisturbed run, or the call reports a device fault - never a silent difference."""
    import torch
    from avatar_amd import api
    pm = synth.identity_part_map()
    opt = Options.demo(icp_iters=2)
    dev = torch.device("cuda:0")
    a = torch.randn(8192, 8192, device=dev); b = torch.randn(8192, 8192, device=dev)
    side = torch.cuda.Stream(device=dev)
    reported = 0
    for F in (1, 2, 3):
        frames = [synth.make_frame(smpl, s) for s in SEEDS[F]]
        ctx = _ctx(api, gmodel, pm, F)
        good = _run(ctx, api, frames, opt)
        for rep in range(3):
            with torch.cuda.stream(side):
                for _ in range(12):
                    c = a @ b                      # ~1.1 TFLOP each: tens of milliseconds of a full chip, queued asynchronously
            try:
                out = _run(ctx, api, frames, opt)
            except api.AvtError as e:
                assert e.status == 3
                reported += 1
                continue
            finally:
                side.synchronize()
            assert np.array_equal(out, good), f"{F} frame(s), repetition {rep}: silently different bits under pressure"
    print(f"handover under pressure: {reported} reported faults")
