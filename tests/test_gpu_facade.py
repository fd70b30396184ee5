"""The C++ facade (include/ark/*.h: ark::AvatarModel / Avatar / AvatarOptimizer over the C ABI) driven with the
reference's call protocol (demo.cpp:137-143, :252-268) and checked against the oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

from avatar_amd import synth
from avatar_amd.capi import Options

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def write_model_dir(smpl, d, compressed=False):
    os.makedirs(d, exist_ok=True)
    keys = {k: smpl[k] for k in ("v_template", "f", "kintree_table", "J_regressor", "weights", "shapedirs")}
    (np.savez_compressed if compressed else np.savez)(os.path.join(d, "model.npz"), **keys)
    synth.write_pose_prior_txt(smpl, os.path.join(d, "pose_prior.txt"))


@pytest.mark.gpu
def test_cpp_facade_matches_oracle(smpl, omodel, tmp_path):
    from oracle import oracle as orc
    exe = os.path.join(HERE, "cpp", "facade_demo")
    assert os.path.exists(exe), "tests/cpp/facade_demo not built (make -C avatar_amd/csrc facade)"
    mdir = str(tmp_path / "model")
    write_model_dir(smpl, mdir)
    fr = synth.make_frame(smpl, 12)
    data, labels = fr["data"][::3], fr["labels"][::3]
    w0, p0, R0 = fr["start"]
    fpath, opath = str(tmp_path / "frame.bin"), str(tmp_path / "out.bin")
    with open(fpath, "wb") as f:
        f.write(struct.pack("i", len(labels)))
        f.write(np.ascontiguousarray(data, np.float64).tobytes()); f.write(np.ascontiguousarray(labels, np.int32).tobytes())
        f.write(w0.astype(np.float64).tobytes()); f.write(p0.astype(np.float64).tobytes())
        f.write(np.ascontiguousarray(np.transpose(R0, (0, 2, 1))).tobytes())      # column-major 3x3 blocks
    r = subprocess.run([exe, mdir, fpath, opath], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = np.fromfile(opath, np.float64)
    V = 6890
    cloud = out[:3 * V].reshape(V, 3); p = out[3 * V:3 * V + 3]; w = out[3 * V + 3:3 * V + 13]
    ref = omodel.optimize(synth.identity_part_map(), 24, data, labels, Options.demo(), p0, orc.rot_to_quat(R0), w0, aggregate=1)
    assert np.abs(cloud - ref["cloud"]).max() < 1e-6
    assert np.abs(p - ref["p"]).max() < 1e-7 and np.abs(w - ref["w"]).max() < 1e-6
    assert abs(out[-1] - ref["stats"].final_cost) < 1e-8 * abs(ref["stats"].final_cost)


@pytest.mark.parametrize("compressed", [False, True])
def test_npz_reader_cpu(smpl, tmp_path, compressed):
    """include/ark/Npz.h (the cnpy replacement for model ingest, AvatarModel.cpp:23-127) against numpy."""
    mdir = str(tmp_path / "model")
    write_model_dir(smpl, mdir, compressed)
    exe = str(tmp_path / "npz_check")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-o", exe,
                           os.path.join(HERE, "cpp", "npz_check.cpp"), "-lz"])
    out = subprocess.run([exe, os.path.join(mdir, "model.npz")], capture_output=True, text=True, check=True).stdout
    seen = {}
    for line in out.strip().splitlines():
        t = line.split()
        seen[t[0]] = (int(t[1]), float(t[2]), float(t[3]), float(t[4]), tuple(int(x) for x in t[5:]))
    for k in ("v_template", "f", "kintree_table", "J_regressor", "weights", "shapedirs"):
        a = np.asarray(smpl[k])
        n, s, first, last, shape = seen[k]
        assert shape == a.shape and n == a.size
        flat = a.reshape(-1).astype(np.float64)
        acc = 0.0
        for x in flat[:2000]:
            acc += x
        assert first == flat[0] and last == flat[-1]
        assert abs(s - float(np.sum(flat))) <= 1e-9 * max(1.0, np.abs(flat).sum())


def test_npz_reader_tolerates_foreign_members_and_rejects_damage(smpl, tmp_path):
    """SMPL exports carry string / bool / object members next to the six arrays the model needs: they are skipped (cnpy
    tolerates them too); a truncated or corrupted file ends in an error, never in an out-of-bounds read."""
    mdir = tmp_path / "model"
    os.makedirs(mdir)
    keys = {k: smpl[k] for k in ("v_template", "f", "kintree_table", "J_regressor", "weights", "shapedirs")}
    path = str(mdir / "model.npz")
    np.savez(path, bs_style=np.array("lbs"), bs_type=np.array(["lrotmin"]), flag=np.array([True, False]), **keys)
    exe = str(tmp_path / "npz_check")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-o", exe,
                           os.path.join(HERE, "cpp", "npz_check.cpp"), "-lz"])
    r = subprocess.run([exe, path], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    names = {line.split()[0] for line in r.stdout.strip().splitlines()}
    assert set(keys) <= names and "bs_style" not in names and "flag" not in names
    raw = open(path, "rb").read()
    rng = np.random.default_rng(0)
    bad = [raw[:len(raw) // 2], raw[:-30], raw[:100], raw[len(raw) // 3:]]
    for _ in range(6):                                   # random damage in the central directory / headers
        b = bytearray(raw)
        for pos in rng.integers(len(raw) - 2000, len(raw), 12):
            b[pos] = int(rng.integers(0, 256))
        bad.append(bytes(b))
    for i, b in enumerate(bad):
        q = str(tmp_path / f"bad{i}.npz")
        open(q, "wb").write(b)
        r = subprocess.run([exe, q], capture_output=True, text=True)
        assert r.returncode in (0, 3), (i, r.returncode, r.stderr)      # clean result or a reported error; never a signal
        if r.returncode == 3:
            assert "npz" in r.stderr
