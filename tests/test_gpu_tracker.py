"""Frame protocol (demo.cpp:215-290): interval subsampling, reinit policy, ICP budgets, warm start."""
import numpy as np
import pytest

from avatar_amd import synth
from avatar_amd.capi import Options


def _sequence(smpl, n=3):
    w, p, R = synth.sample_ground_truth(smpl, 21, use_gmm=False)
    w = 0.5 * w
    frames = []
    for k in range(n):
        Rk = R.copy()
        Rk[16] = R[16] @ synth.rodrigues([0.0, 0.0, 0.12 * k])     # left shoulder swings
        Rk[4] = R[4] @ synth.rodrigues([0.10 * k, 0.0, 0.0])       # left knee bends
        pk = p + np.array([0.02 * k, 0.0, -0.01 * k])
        verts = synth.pose_vertices(smpl, w, pk, Rk)
        xyz, mask, n_fg = synth.render_images(smpl, verts, synth.identity_part_map())
        frames.append((xyz, mask, verts))
    return frames


def test_subsample_cpu(smpl):
    """Pure host logic (no GPU): subsampling matches a literal restatement of the demo loop."""
    from avatar_amd.tracker import FrameTracker

    class _Opt:           # stand-in with the two attributes subsample() reads
        numParts = 24
        ava = None
    (xyz, mask, _), = _sequence(smpl, 1)
    tr = FrameTracker.__new__(FrameTracker)
    tr.opt = _Opt(); tr.interval = 12
    ys, xs = np.nonzero(mask != 255)
    bbox = (ys.min(), xs.min(), ys.max(), xs.max())
    data, labels = tr.subsample(xyz, mask, bbox)
    ref_pts, ref_lab = [], []
    for r in range(bbox[0], bbox[2] + 1, 12):
        for c in range(bbox[1], bbox[3] + 1, 12):
            if mask[r, c] == 255:
                continue
            ref_pts.append([xyz[r, c, 0], -xyz[r, c, 1], xyz[r, c, 2]]); ref_lab.append(mask[r, c])
    assert np.array_equal(data, np.array(ref_pts, np.float64)) and np.array_equal(labels, np.array(ref_lab, np.int32))
    assert 100 < len(labels) < 1000


@pytest.mark.gpu
def test_tracker_sequence_matches_oracle(smpl, omodel, gmodel):
    from avatar_amd import api
    from avatar_amd.tracker import FrameTracker
    from oracle import oracle as orc
    frames = _sequence(smpl, 3)
    pm = synth.identity_part_map()
    ava = api.Avatar(gmodel)
    opt = api.AvatarOptimizer(ava, None, (1280, 720), 24, pm, max_points=4096)
    opt.betaPose, opt.betaShape = 0.05, 0.12
    tr = FrameTracker(opt, interval=6, frame_icp_iters=2, reinit_icp_iters=3, reinit_cnz=1000)
    # oracle mirror of the same protocol
    o_w = np.zeros(10); o_p = np.zeros(3); o_R = np.tile(np.eye(3), (24, 1, 1)); reinit = True
    errs = []
    for xyz, mask, gt in frames:
        ys, xs = np.nonzero(mask != 255)
        bbox = (ys.min(), xs.min(), ys.max(), xs.max())
        assert tr.process(xyz, mask, bbox)
        data, labels = tr.subsample(xyz, mask, bbox)
        icp = 2
        if reinit:
            o_p = data.mean(0); o_w = np.zeros(10); o_R = np.tile(np.eye(3), (24, 1, 1))
            o_R[0] = np.array([[-1.0, 0, 0], [0, 1.0, 0], [0, 0, -1.0]]); reinit = False; icp = 3
        o = Options.demo(icp_iters=icp)
        ref = omodel.optimize(pm, 24, data, labels, o, o_p, orc.rot_to_quat(o_R), o_w, aggregate=1)
        o_p, o_w, o_R = ref["p"], ref["w"], orc.quat_to_rot(ref["q"])
        assert np.abs(ava.cloud - ref["cloud"]).max() < 1e-5
        errs.append(np.abs(ava.cloud - gt).mean())
    assert errs[-1] < 0.05          # the warm-started fit stays on the subject
    # tracking loss: an (almost) empty mask flips the tracker back to reinit
    empty = np.full_like(frames[0][1], 255)
    assert not tr.process(frames[0][0], empty, (0, 0, 719, 1279)) and tr.reinit
