"""GPU synthetic-frame generator against the host generator: bit-identical clouds and labels."""
import numpy as np
import pytest

from avatar_amd import synth

pytestmark = pytest.mark.gpu


def test_gpu_render_matches_host_renderer(smpl, gmodel):
    from avatar_amd import api
    pm = synth.identity_part_map()
    ws, ps, Rs = [], [], []
    for s in (0, 1, 2):
        w, p, R = synth.sample_ground_truth(smpl, s)
        ws.append(w); ps.append(p); Rs.append(R)
    ctx = api.Context(gmodel, 24, pm, 60000, 4)
    n = ctx.render_frames(np.array(ws), np.array(ps), np.array(Rs))
    cloud, _, _ = ctx.lbs_update(np.array(ws), np.array(ps), np.array(Rs))
    n2 = ctx.render_frames(np.array(ws), np.array(ps), np.array(Rs))      # lbs_update does not disturb the frames
    assert np.array_equal(n, n2)
    for f in range(3):
        data_h, lab_h = synth.render_cloud(smpl, cloud[f], pm)             # host z-buffer on the SAME posed vertices
        data_g, lab_g = ctx.frame_download(f)
        assert len(lab_g) == len(lab_h) and 15000 < len(lab_g) < 60000
        assert np.array_equal(lab_g, lab_h)
        assert np.array_equal(data_g, data_h)


def test_render_then_optimize_resident(smpl, omodel, gmodel):
    """Rendered frames are directly usable by avt_optimize_resident (no host round trip)."""
    from avatar_amd import api
    from avatar_amd.capi import Options
    pm = synth.identity_part_map()
    w, p, R = synth.sample_ground_truth(smpl, 5)
    ctx = api.Context(gmodel, 24, pm, 60000, 1)
    ctx.render_frames(w[None], p[None], R[None])
    data, lab = ctx.frame_download(0)
    w0, p0, R0 = synth.perturb_start(w, p, R, 5)
    q0 = api.rot_to_quat(R0)
    opt = Options.demo()
    ctx.state_upload(p0[None], q0[None], w0[None])
    ctx.optimize_resident(opt)
    pg, qg, wg, st = ctx.state_download()
    ref = omodel.optimize(pm, 24, data, lab, opt, p0, q0, w0, aggregate=1)
    assert np.array_equal(ctx.correspondences(0, len(lab)), ref["corr"])
    assert np.abs(ctx.cloud(0) - ref["cloud"]).max() < 1e-6
