"""The reference's deprecated ad-hoc model format (AvatarModel.cpp:128-288: skeleton.txt, model.pcd, shapekey/*.pcd,
joint_shape_regressor.txt or joint_regressor.txt, mesh.txt) read by the C++ facade (include/ark/Avatar.h) and by the Python mirror
(avatar_amd/api.py): the same model written in both formats gives the same model data.  Host side only (avt_model_create needs no GPU)."""
import os
import subprocess

import numpy as np
import pytest

from avatar_amd import api, capi, synth

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def write_pcd(path, pts):
    pts = np.asarray(pts, np.float64).reshape(-1, 3)
    with open(path, "w") as f:
        f.write("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\n")
        f.write(f"WIDTH {len(pts)}\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {len(pts)}\nDATA ascii\n")
        for p in pts:
            f.write("%.17g %.17g %.17g\n" % tuple(p))


def write_legacy_dir(smpl, d, with_jsr, rng):
    """The files smpl-to-legacy conversion would produce (the layout AvatarModel.cpp:128-288 parses)."""
    os.makedirs(os.path.join(d, "shapekey"), exist_ok=True)
    V, J = smpl["v_template"].shape[0], smpl["weights"].shape[1]
    K = smpl["shapedirs"].shape[2]
    parent = np.asarray(smpl["kintree_table"])[0].astype(np.int64); parent[0] = -1
    write_pcd(os.path.join(d, "model.pcd"), smpl["v_template"])
    for k in range(K):
        write_pcd(os.path.join(d, "shapekey", f"shape{k:03d}.pcd"), smpl["shapedirs"][:, :, k])
    rest = np.asarray(smpl["J_regressor"]) @ np.asarray(smpl["v_template"])
    with open(os.path.join(d, "skeleton.txt"), "w") as f:
        f.write(f"{J} {V}\n")
        for j in range(J):
            f.write(f"{j} {parent[j]} joint{j} %.17g %.17g %.17g\n" % tuple(rest[j]))
        W = np.asarray(smpl["weights"])
        for v in range(V):
            nz = np.nonzero(W[v])[0]
            nz = nz[rng.permutation(len(nz))]                  # any order: the loader sorts
            f.write(str(len(nz)) + "".join(" %d %.17g" % (j, W[v, j]) for j in nz) + "\n")
    JR = np.asarray(smpl["J_regressor"], np.float64)
    if with_jsr:
        base = (JR @ np.asarray(smpl["v_template"], np.float64)).reshape(-1)                                 # 3J
        reg = np.stack([(JR @ np.asarray(smpl["shapedirs"], np.float64)[:, :, k]).reshape(-1) for k in range(K)], 1)   # 3J x K
        with open(os.path.join(d, "joint_shape_regressor.txt"), "w") as f:
            f.write(f"{K}\n" + " ".join("%.17g" % x for x in base) + "\n")
            for i in range(3 * J):
                f.write(" ".join("%.17g" % x for x in reg[i]) + "\n")
    else:
        with open(os.path.join(d, "joint_regressor.txt"), "w") as f:
            f.write(f"{J}\n")
            for j in range(J):
                nz = np.nonzero(JR[j])[0]
                f.write(str(len(nz)) + "".join(" %d %.17g" % (v, JR[j, v]) for v in nz) + "\n")
    F = np.asarray(smpl["f"])
    with open(os.path.join(d, "mesh.txt"), "w") as f:
        f.write(f"{len(F)}\n" + "\n".join("%d %d %d" % tuple(t) for t in F) + "\n")
    synth.write_pose_prior_txt(smpl, os.path.join(d, "pose_prior.txt"))


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("legacy") / "legacy_model_check")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-o", exe, os.path.join(HERE, "cpp", "legacy_model_check.cpp"),
                           "-L", os.path.join(ROOT, "avatar_amd", "csrc"), "-lavatar_hip", "-lz", "-Wl,-rpath," + os.path.join(ROOT, "avatar_amd", "csrc")])
    return exe


def _run(exe, d, limit=0):
    out = subprocess.run([exe, d, str(limit)], capture_output=True, text=True, check=True).stdout
    rec = {l.split()[0]: l.split()[1:] for l in out.splitlines()}
    return ([int(x) for x in rec["dims"]], np.array(rec["parent"], int), np.array(rec["main_joint"], int),
            np.array(rec["initial_joint_pos"], float), np.array(rec["joint_shape_reg"], float), int(rec["use_jsr"][0]))


@pytest.mark.parametrize("with_jsr", [True, False])
def test_legacy_format_gives_the_model_of_the_npz(smpl, tmp_path, checker, with_jsr):
    from tests.test_gpu_facade import write_model_dir
    rng = np.random.default_rng(1)
    dn, dl = str(tmp_path / "npz"), str(tmp_path / "legacy")
    write_model_dir(smpl, dn)
    write_legacy_dir(smpl, dl, with_jsr, rng)
    dims_n, par_n, mj_n, ijp_n, jsr_n, _ = _run(checker, dn)
    dims_l, par_l, mj_l, ijp_l, jsr_l, use_l = _run(checker, dl)
    assert dims_l == dims_n and np.array_equal(par_l, par_n) and np.array_equal(mj_l, mj_n)
    assert use_l == (1 if with_jsr else 0)                    # AvatarModel.cpp:243, :263
    # (the text files carry 17 significant digits: the numbers are the npz's; the regressor path sums in the same order)
    assert np.allclose(ijp_l, ijp_n, rtol=0, atol=1e-14) and np.allclose(jsr_l, jsr_n, rtol=0, atol=1e-14)
    # the Python mirror reads the same directory to the same model
    ml = api.AvatarModel(dl)
    mn = api.AvatarModel(dn)
    assert np.array_equal(ml.mainJoint, mn.mainJoint)
    assert np.allclose(ml.initialJointPos, mn.initialJointPos, atol=1e-14) and np.allclose(ml.jointShapeReg, mn.jointShapeReg, atol=1e-14)
    assert np.allclose(ml.initialJointPos.reshape(-1), ijp_l, atol=1e-15)


def test_limit_one_joint_per_point_host_side(smpl, omodel):
    """AvatarModel.cpp:190-196: assignedJoints keeps the largest weight only (set to 1): same main joints, and the oracle built from
    the same description sees one ancestor chain per point."""
    from oracle import oracle as orc
    m1 = api.AvatarModel(smpl, limit_one_joint_per_point=True)
    m0 = api.AvatarModel(smpl)
    assert np.array_equal(m1.mainJoint, m0.mainJoint)
    o1 = orc.OracleModel(smpl, limit_one_joint_per_point=True)
    parent = np.asarray(smpl["kintree_table"])[0].astype(int); parent[0] = -1
    for v in (0, 100, 3000, 6889):
        chain, j = [], int(m0.mainJoint[v])
        while j != -1:
            chain.append(j); j = parent[j]
        assert sorted(o1.ancestors(v).tolist()) == sorted(chain)
        assert len(omodel.ancestors(v)) >= len(chain)
