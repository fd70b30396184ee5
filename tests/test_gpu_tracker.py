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


@pytest.mark.gpu
def test_first_fit_wants_every_body_part(smpl, gmodel):
    """live-demo.cpp:376-383: before the FIRST fit every body part must be seen (initialPerPartCnz pixels at interval 1); a frame
    with a part missing is skipped and asks for a reinitialisation, later frames are not held to it.  Off (0) = demo.cpp."""
    from avatar_amd import api
    from avatar_amd.tracker import FrameTracker
    xyz, mask, _ = _sequence(smpl, 1)[0]
    pm = synth.identity_part_map()
    ava = api.Avatar(gmodel)
    opt = api.AvatarOptimizer(ava, None, (1280, 720), 24, pm, max_points=4096)
    tr = FrameTracker(opt, interval=6, frame_icp_iters=1, reinit_icp_iters=1, reinit_cnz=500, initial_per_part_cnz=36)
    ys, xs = np.nonzero(mask != 255)
    bbox = (ys.min(), xs.min(), ys.max(), xs.max())
    part = int(np.bincount(mask[mask != 255]).argmax())
    holed = mask.copy(); holed[mask == part] = 255            # one body part not seen at all
    assert not tr.process(xyz, holed, bbox) and tr.reinit and tr.firstTime
    assert tr.process(xyz, mask, bbox) and not tr.firstTime
    assert tr.process(xyz, holed, bbox)                        # after the first fit the per-part rule no longer applies


def write_sequence(path, frames, interval, frame_icp, reinit_icp, reinit_cnz):
    """sequence.bin of tests/cpp/tracker_demo.cpp: header, then per frame bbox + XYZ map (float32) + part mask (uint8)."""
    import struct
    H, W = frames[0][1].shape
    with open(path, "wb") as f:
        f.write(struct.pack("7i", len(frames), W, H, interval, frame_icp, reinit_icp, reinit_cnz))
        for xyz, mask, bbox in frames:
            f.write(struct.pack("4i", *[int(v) for v in bbox]))
            f.write(np.ascontiguousarray(xyz, np.float32).tobytes()); f.write(np.ascontiguousarray(mask, np.uint8).tobytes())


@pytest.mark.gpu
def test_cpp_frame_tracker_matches_oracle(smpl, omodel, tmp_path):
    """include/ark/FrameTracker.h (the demo.cpp:215-290 loop in C++ over the C ABI) on a rendered sequence, against the
    oracle mirror of the same protocol; an empty frame in the middle must flip it to reinit and the next frame must
    reinitialise (centroid start, reinit ICP budget)."""
    import os
    import subprocess
    from oracle import oracle as orc
    from tests.test_gpu_facade import write_model_dir
    here = os.path.dirname(os.path.abspath(__file__))
    exe = os.path.join(here, "cpp", "tracker_demo")
    assert os.path.exists(exe), "tests/cpp/tracker_demo not built (make -C avatar_amd/csrc facade)"
    mdir = str(tmp_path / "model")
    write_model_dir(smpl, mdir)
    seq = _sequence(smpl, 3)
    frames = []
    for xyz, mask, gt in seq:
        ys, xs = np.nonzero(mask != 255)
        frames.append((xyz, mask, (ys.min(), xs.min(), ys.max(), xs.max())))
    empty = (seq[0][0], np.full_like(seq[0][1], 255), (0, 0, 719, 1279))
    order = [frames[0], frames[1], empty, frames[2]]
    spath, opath = str(tmp_path / "seq.bin"), str(tmp_path / "out.bin")
    write_sequence(spath, order, 6, 2, 3, 1000)
    r = subprocess.run([exe, mdir, spath, opath], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = open(opath, "rb").read()
    V, off, got = 6890, 0, []
    for _ in order:
        fitted = int(np.frombuffer(raw, np.int32, 1, off)[0]); off += 4
        if fitted:
            cloud = np.frombuffer(raw, np.float64, 3 * V, off).reshape(V, 3); off += 8 * 3 * V
            pw = np.frombuffer(raw, np.float64, 13, off); off += 8 * 13
            got.append((cloud, pw[:3], pw[3:]))
        else:
            got.append(None)
    assert got[2] is None and all(g is not None for g in (got[0], got[1], got[3]))
    # oracle mirror
    from avatar_amd.tracker import FrameTracker

    class _Opt:
        numParts = 24
    sub = FrameTracker.__new__(FrameTracker); sub.opt = _Opt(); sub.interval = 6
    pm = synth.identity_part_map()
    o_w = np.zeros(10); o_p = np.zeros(3); o_R = np.tile(np.eye(3), (24, 1, 1)); reinit = True
    for k, (xyz, mask, bbox) in enumerate(order):
        data, labels = sub.subsample(xyz, mask, bbox)
        if len(labels) < 1000 // 36:
            reinit = True
            continue
        icp = 2
        if reinit:
            o_p = data.mean(0); o_w = np.zeros(10); o_R = np.tile(np.eye(3), (24, 1, 1))
            o_R[0] = np.array([[-1.0, 0, 0], [0, 1.0, 0], [0, 0, -1.0]]); reinit = False; icp = 3
        ref = omodel.optimize(pm, 24, data, labels, Options.demo(icp_iters=icp), o_p, orc.rot_to_quat(o_R), o_w, aggregate=1)
        o_p, o_w, o_R = ref["p"], ref["w"], orc.quat_to_rot(ref["q"])
        cloud, p, w = got[k]
        assert np.abs(cloud - ref["cloud"]).max() < 1e-5, k
        assert np.abs(p - ref["p"]).max() < 1e-6 and np.abs(w - ref["w"]).max() < 1e-5
