"""Host side of the batch split (include/avt_shard.h): the frame partition, the packed model the broadcast ships.
No GPU, no RCCL (the world-size-2 gloo run of the N>1 path is tests/test_oracle_cpu.py::test_frame_sharding_world_size_2_gloo)."""
import ctypes
import os

import numpy as np
import pytest

from avatar_amd import capi, shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_is_a_bijection():
    lib = capi.load_library()
    for B in (1, 5, 64, 512, 513):
        for W in (1, 2, 3, 4, 8):
            seen = []
            for r in range(W):
                n = lib.avt_shard_local_count(B, r, W)
                assert n == len(shard.frames_of_rank(B, r, W))
                for i in range(n):
                    f = lib.avt_shard_global_frame(i, r, W)
                    assert f == shard.frames_of_rank(B, r, W)[i]
                    assert lib.avt_shard_owner(f, W) == r and lib.avt_shard_local_index(f, W) == i
                    seen.append(f)
            assert sorted(seen) == list(range(B))
    assert lib.avt_shard_local_count(3, 5, 8) == 0            # more ranks than frames: the surplus ranks hold nothing


def test_packed_model_round_trip(smpl, omodel):
    """avt_model_pack -> bytes -> avt_model_unpack yields the same derived model data as avt_model_create."""
    lib = capi.load_library()
    arr = capi.ModelArrays(smpl)
    block = shard.pack_model(arr)
    assert block[:8] == b"AVTMODEL" and len(block) % 8 == 0
    h = shard.unpack_model(block)
    V, J, K, F, P = (ctypes.c_int() for _ in range(5))
    assert lib.avt_model_dims(h, ctypes.byref(V), ctypes.byref(J), ctypes.byref(K), ctypes.byref(F), ctypes.byref(P)) == 0
    assert (V.value, J.value, K.value, F.value, P.value) == (arr.V, arr.J, arr.K, arr.F, arr.P)
    mj = np.empty(arr.V, np.int32)
    assert lib.avt_model_main_joint(h, capi.iptr(mj)) == 0
    assert np.array_equal(mj, omodel.main_joint())
    ijp = np.empty(3 * arr.J); jsr = np.empty(3 * arr.J * arr.K)
    assert lib.avt_model_joint_regression(h, capi.dptr(ijp), capi.dptr(jsr)) == 0
    o_ijp, o_jsr = omodel.joint_regression()
    assert np.array_equal(ijp.reshape(-1, 3), o_ijp) and np.array_equal(jsr.reshape(arr.K, -1).T, o_jsr)
    lib.avt_model_destroy(h)
    # the block is position independent: a copy at another address unpacks too
    h2 = shard.unpack_model(bytes(bytearray(block)))
    lib.avt_model_destroy(h2)


def test_packed_model_rejects_damage(smpl):
    from avatar_amd.api import AvtError
    block = shard.pack_model(capi.ModelArrays(smpl))
    for bad in (block[:100], block[:-8], b"XXXXXXXX" + block[8:], block[:8] + b"\x07" + block[9:]):
        with pytest.raises(AvtError):
            shard.unpack_model(bad)
    # a corrupted sparse column pointer must be caught before avt_model_create walks it
    hdr = np.frombuffer(block[:56], np.int32)
    V = int(hdr[3])
    off_colptr = 56 + 8 * 3 * V + 8 * 3 * V * int(hdr[5]) + ((4 * int(hdr[4]) + 7) & ~7) + ((4 * 3 * int(hdr[6]) + 7) & ~7)
    b = bytearray(block)
    b[off_colptr + 4 * V:off_colptr + 4 * V + 4] = np.int32(7).tobytes()          # weights_colptr[V] != nnz
    with pytest.raises(AvtError):
        shard.unpack_model(bytes(b))


def test_a_bogus_rccl_path_does_not_crash_the_loader():
    """dlerror() was called twice and a NULL went into a std::string - any candidate that failed to open crashed
    the process (rc 139) instead of advancing to the next one.  With AVT_RCCL_LIB pointing nowhere the loader must move on to
    the system copies, or fail with a message."""
    import subprocess
    import sys
    code = ("import ctypes, sys; sys.path.insert(0, %r); from avatar_amd import capi; lib = capi.load_library(); "
            "buf = ctypes.create_string_buffer(128); rc = lib.avt_shard_unique_id(buf); "
            "print('rc', rc, lib.avt_last_error().decode() if rc else 'ok')" % ROOT)
    env = dict(os.environ, AVT_RCCL_LIB="/nonexistent/librccl.so.1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, f"the loader crashed: rc {r.returncode}\n{r.stderr[-400:]}"
    assert "rc 0 ok" in r.stdout or "librccl" in r.stdout or "nccl" in r.stdout


def test_packed_model_carries_the_legacy_format_fields(smpl):
    """limit_one_joint_per_point and a joint shape regressor given directly (joint_shape_regressor.txt of the legacy format) travel in
    the packed block the model broadcast ships."""
    lib = capi.load_library()
    m = dict(smpl)
    J, K = m["weights"].shape[1], m["shapedirs"].shape[2]
    rng = np.random.default_rng(2)
    m["joint_shape_reg_base"] = rng.standard_normal(3 * J)
    m["joint_shape_reg"] = rng.standard_normal((3 * J, K))
    arr = capi.ModelArrays(m, limit_one_joint_per_point=True)
    h = shard.unpack_model(shard.pack_model(arr))
    ijp = np.empty(3 * J); jsr = np.empty(3 * J * K)
    assert lib.avt_model_joint_regression(h, ijp.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), jsr.ctypes.data_as(ctypes.POINTER(ctypes.c_double))) == 0
    assert np.array_equal(ijp, m["joint_shape_reg_base"]) and np.array_equal(jsr.reshape(K, 3 * J).T, m["joint_shape_reg"])
    lib.avt_model_destroy(h)
    plain = shard.pack_model(capi.ModelArrays(smpl))
    assert len(shard.pack_model(arr)) > len(plain)            # the two extra arrays are there only when given
