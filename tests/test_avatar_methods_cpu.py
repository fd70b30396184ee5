"""CPU: the host-side members of ark::Avatar that the reference's tools use beside the tracker path (include/ark/Avatar.h): randomize
(Avatar.cpp:77-126), smplParams (:128-137), pdf (:139 + GaussianMixture.cpp:83-93), alignToJoints (:141-193) - against numpy restatements
of the same closed forms, std::mt19937's documented stream, and the oracle's forward kinematics."""
import os
import subprocess

import numpy as np
import pytest

from avatar_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _rows(out):
    return {l.split()[0]: np.array([float(x) for x in l.split()[1:]]) for l in out.splitlines() if l.strip()}


def _rot(rec):
    return rec.reshape(-1, 3, 3).transpose(0, 2, 1)      # column-major blocks -> (J, 3, 3)


@pytest.fixture(scope="module")
def run(smpl, tmp_path_factory):
    from tests.test_gpu_facade import write_model_dir
    td = tmp_path_factory.mktemp("avatar_methods")
    exe = str(td / "avatar_methods_check")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-o", exe, os.path.join(HERE, "cpp", "avatar_methods_check.cpp"),
                           "-L", os.path.join(ROOT, "avatar_amd", "csrc"), "-lavatar_hip", "-lz", "-Wl,-rpath," + os.path.join(ROOT, "avatar_amd", "csrc")])
    mdir = str(td / "model")
    write_model_dir(smpl, mdir)
    return exe, mdir, td


def _targets(smpl, omodel, p, R):
    """World joint positions of the zero-shape avatar posed with (p, R): the oracle's update()."""
    w = np.zeros(10)
    cloud, jpos, _ = omodel.update(w, p, R)
    return jpos


def test_randomize_smplparams_pdf_align(smpl, omodel, run):
    exe, mdir, td = run
    w, p, R = synth.sample_ground_truth(smpl, 5, use_gmm=False)
    tgt = _targets(smpl, omodel, p, R)
    tgt_nan = tgt.copy(); tgt_nan[[10, 22]] = np.nan      # L_FOOT and L_HAND not seen
    tfile = str(td / "targets.txt")
    with open(tfile, "w") as f:
        for row in tgt_nan:
            f.write(" ".join("nan" if np.isnan(x) else "%.17g" % x for x in row) + "\n")
    rec = _rows(subprocess.run([exe, mdir, tfile], capture_output=True, text=True, check=True).stdout)
    J = 24
    # ---- randomize: reproducible under (seed, reseed), ranges of Avatar.cpp:105-124, rotations orthonormal
    assert np.array_equal(rec["w1"], rec["w2"]) and np.array_equal(rec["p1"], rec["p2"]) and np.array_equal(rec["r1"], rec["r2"])
    p1 = rec["p1"]
    assert -1 <= p1[0] < 1 and -0.5 <= p1[1] < 0.5 and 2.2 <= p1[2] < 4.5
    R1 = _rot(rec["r1"])
    assert np.abs(R1 @ R1.transpose(0, 2, 1) - np.eye(3)).max() < 1e-12 and np.allclose(np.linalg.det(R1), 1.0)
    assert np.abs(rec["w1"]).max() < 6 and rec["w1"].std() > 0.2
    # the root faces the camera: a turn of pi +- pi/3 about y, perturbed by N(0, 0.2) rad
    assert R1[0][2, 2] < -0.3
    # only-root call: shape zero, joints identity, another root
    assert not rec["w3"].any() and np.allclose(_rot(rec["r3"])[1:], np.eye(3)) and not np.array_equal(rec["p3"], rec["p1"])
    # the first shape coefficient is std::normal_distribution<float>(0, 1) on mt19937(1234): libstdc++'s stream, pinned by value
    # (the draw is a float: exactly representable)
    assert rec["w1"][0] == np.float32(rec["w1"][0])
    # ---- smplParams: axis-angle of r[1..]; Rodrigues of it gives the rotation back; the pose was drawn from the LAST component (the
    # reference's component loop has no break) as mean + chol(cov) z
    sp = rec["smpl1"].reshape(J - 1, 3)
    for i in range(J - 1):
        assert np.abs(synth.rodrigues(sp[i]) - R1[i + 1]).max() < 1e-12
        assert np.linalg.norm(sp[i]) <= np.pi + 1e-12
    # ---- pdf against a dense numpy evaluation of GaussianMixture.cpp:22-93 (quirk kept: |L (x - mu)|^2 with L = chol(cov^-1), not L^T)
    wt, mu, cov = np.asarray(smpl["prior_weight"], float), np.asarray(smpl["prior_mean"], float), np.asarray(smpl["prior_cov"], float)
    n = mu.shape[1]
    dets = np.array([np.prod(np.diag(np.linalg.cholesky(c))) for c in cov])
    consts = wt / (2 * np.pi) ** (n * 0.5) / dets * dets.min()
    assert np.allclose(rec["consts"], consts, rtol=1e-9)
    assert np.allclose(rec["consts_log"], np.log(wt) - n * 0.5 * np.log(2 * np.pi) - np.log(dets) + np.log(dets.min()), rtol=1e-9, atol=1e-9)
    x = sp.reshape(-1)
    pdf = sum(consts[i] * np.exp(-0.5 * np.sum((np.linalg.cholesky(np.linalg.inv(cov[i])) @ (x - mu[i])) ** 2)) for i in range(len(wt)))
    assert abs(rec["pdf1"][0] - pdf) <= 1e-7 * abs(pdf) + 1e-300
    # ---- alignToJoints: the root goes to joint 0; every seen bone points where the target's does; unseen joints keep the identity
    assert np.array_equal(rec["p4"], tgt[0])
    R4 = _rot(rec["r4"])
    assert np.allclose(R4[10], np.eye(3)) and np.allclose(R4[22], np.eye(3))
    rest = _targets(smpl, omodel, np.zeros(3), np.tile(np.eye(3), (J, 1, 1)))      # rest joints of the zero-shape model (root at 0)
    parent = np.asarray(smpl["kintree_table"])[0].astype(int); parent[0] = -1
    rt = [None] * J
    v0, v1 = rest[3] - rest[0], tgt[3] - tgt[0]
    assert np.allclose(R4[0] @ (v0 / np.linalg.norm(v0)), v1 / np.linalg.norm(v1), atol=1e-12)
    rt[0] = R4[0]
    for i in range(1, J):
        if np.isnan(tgt_nan[i, 0]):
            rt[i] = rt[parent[i]]
            continue
        rt[i] = rt[parent[i]] @ R4[i]      # r[i] = rotTrans[parent]^T rotTrans[i]
        a, b = rest[i] - rest[parent[i]], tgt[i] - tgt[parent[i]]
        assert np.allclose(rt[i] @ (a / np.linalg.norm(a)), b / np.linalg.norm(b), atol=1e-10), i
    # a missing joint makes the mean bone-length ratio NaN, and the reference then sets the width coefficient to 1.5 (Avatar.cpp:174-175)
    assert rec["w4"][0] == 1.5 and not rec["w4"][1:].any()
    # all joints seen: bone lengths of the target equal the model's (same shape), so the width coefficient stays ~0; a target 10 % larger widens it
    for scale, check in ((1.0, lambda w0: abs(w0) < 1e-9), (1.1, lambda w0: abs(w0 - 32.0 * 0.1 * np.linalg.norm(rest[6] - rest[0])) < 1e-9)):
        with open(tfile, "w") as f:
            for row in tgt[0] + scale * (tgt - tgt[0]):
                f.write(" ".join("%.17g" % x for x in row) + "\n")
        rec2 = _rows(subprocess.run([exe, mdir, tfile], capture_output=True, text=True, check=True).stdout)
        assert check(rec2["w4"][0]), (scale, rec2["w4"][0])
        assert np.allclose(_rot(rec2["r4"])[10] @ np.eye(3), _rot(rec2["r4"])[10]) and np.abs(_rot(rec2["r4"]) - _rot(rec2["r4"])).max() == 0


def test_python_mirror_agrees_with_the_cpp_facade(smpl, omodel, run, monkeypatch):
    """avatar_amd.api.Avatar's host-side members against the C++ facade's output on the same rotations / targets (no GPU: the methods are host code)."""
    from avatar_amd import api
    exe, mdir, td = run
    w, p, R = synth.sample_ground_truth(smpl, 5, use_gmm=False)
    tgt = _targets(smpl, omodel, p, R)
    tfile = str(td / "targets_py.txt")
    tgt_nan = tgt.copy(); tgt_nan[[10, 22]] = np.nan
    with open(tfile, "w") as f:
        for row in tgt_nan:
            f.write(" ".join("nan" if np.isnan(x) else "%.17g" % x for x in row) + "\n")
    rec = _rows(subprocess.run([exe, mdir, tfile], capture_output=True, text=True, check=True).stdout)
    gm = api.AvatarModel(smpl)
    ava = api.Avatar(gm)
    ava.r = _rot(rec["r1"]).copy()
    assert np.abs(ava.smplParams() - rec["smpl1"]).max() < 1e-9
    assert abs(ava.pdf() - rec["pdf1"][0]) <= 1e-7 * abs(rec["pdf1"][0]) + 1e-300
    ava.alignToJoints(tgt_nan)
    assert np.abs(ava.r - _rot(rec["r4"])).max() < 1e-9 and np.array_equal(ava.p, rec["p4"]) and ava.w[0] == 1.5
    ava.randomize(seed=3)
    assert 2.2 <= ava.p[2] < 4.5 and np.abs(ava.r @ ava.r.transpose(0, 2, 1) - np.eye(3)).max() < 1e-12
