"""Row f2 pinned by the reference's own reader: the same model.npz goes through cnpy::npz_load (cnpy.cpp:246-300, compiled from
/root/reference where it lies into oracle/_ref/libcnpy_ref.so) and through include/ark/Npz.h; shapes and every element agree.
The reference turns a member's raw bytes into a matrix in util::loadFloatMatrix / loadUintMatrix (Util.cpp:249-300: word size 4 or
8, column-major map when fortran_order, row-major otherwise); that rule is applied to cnpy's bytes here, member by member, with
the interpretation AvatarModel.cpp:25-127 chooses (float for v_template / shapedirs / J_regressor / weights, unsigned for f /
kintree_table).  Fixtures are generated, not committed."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libcnpy_ref.so")

FLOAT_KEYS = ("v_template", "shapedirs", "J_regressor", "weights", "posedirs")
UINT_KEYS = ("f", "kintree_table")


def _ref_lib():
    if not os.path.exists(REF_SO):
        if os.path.exists("/root/reference/cnpy.cpp"):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], stdout=subprocess.DEVNULL)
        else:
            pytest.skip("the reference's cnpy reader was not built (no reference tree here)")
    lib = C.CDLL(REF_SO)
    lib.cnpyref_open.restype = C.c_void_p; lib.cnpyref_open.argtypes = [C.c_char_p]
    lib.cnpyref_name.restype = C.c_char_p; lib.cnpyref_name.argtypes = [C.c_void_p, C.c_int]
    lib.cnpyref_count.argtypes = [C.c_void_p]
    lib.cnpyref_info.restype = C.c_longlong
    lib.cnpyref_info.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_longlong)]
    lib.cnpyref_bytes.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.cnpyref_close.argtypes = [C.c_void_p]
    return lib


@pytest.fixture(scope="module")
def ark_lib(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("npzcapi") / "libnpz_capi.so")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-I", os.path.join(ROOT, "include"), "-o", so,
                           os.path.join(HERE, "cpp", "npz_capi.cpp"), "-lz"])
    lib = C.CDLL(so)
    lib.arknpz_open.restype = C.c_void_p; lib.arknpz_open.argtypes = [C.c_char_p]
    lib.arknpz_error.restype = C.c_char_p; lib.arknpz_error.argtypes = [C.c_void_p]
    lib.arknpz_name.restype = C.c_char_p; lib.arknpz_name.argtypes = [C.c_void_p, C.c_int]
    lib.arknpz_count.argtypes = [C.c_void_p]
    lib.arknpz_info.restype = C.c_longlong
    lib.arknpz_info.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_longlong)]
    lib.arknpz_values.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.arknpz_close.argtypes = [C.c_void_p]
    return lib


def _load_ref(lib, path):
    """{name: (word_size, fortran, shape, raw bytes)} through cnpy::npz_load."""
    h = lib.cnpyref_open(path.encode())
    assert h, "cnpy::npz_load threw"
    out = {}
    for i in range(lib.cnpyref_count(h)):
        ws, fo, nd = C.c_int(), C.c_int(), C.c_int()
        shape = (C.c_longlong * 8)()
        nbytes = lib.cnpyref_info(h, i, C.byref(ws), C.byref(fo), C.byref(nd), shape)
        buf = np.empty(nbytes, np.uint8)
        lib.cnpyref_bytes(h, i, buf.ctypes.data_as(C.c_void_p))
        out[lib.cnpyref_name(h, i).decode()] = (ws.value, bool(fo.value), tuple(shape[k] for k in range(nd.value)), buf)
    lib.cnpyref_close(h)
    return out


def _load_ark(lib, path):
    h = lib.arknpz_open(path.encode())
    err = lib.arknpz_error(h).decode()
    assert not err, err
    out = {}
    for i in range(lib.arknpz_count(h)):
        ii, nd = C.c_int(), C.c_int()
        shape = (C.c_longlong * 8)()
        n = lib.arknpz_info(h, i, C.byref(ii), C.byref(nd), shape)
        f = np.empty(n, np.float64); iv = np.empty(n, np.int64)
        lib.arknpz_values(h, i, f.ctypes.data_as(C.c_void_p), iv.ctypes.data_as(C.c_void_p))
        out[lib.arknpz_name(h, i).decode()] = (bool(ii.value), tuple(shape[k] for k in range(nd.value)), iv if ii.value else f)
    lib.arknpz_close(h)
    return out


def _reference_matrix(ws, fortran, shape, raw, kind):
    """What util::loadFloatMatrix / loadUintMatrix make of a member (Util.cpp:249-300), as a logical array in C order."""
    assert ws in (4, 8)
    dt = {("f", 4): "<f4", ("f", 8): "<f8", ("u", 4): "<u4", ("u", 8): "<u8"}[(kind, ws)]
    flat = raw.view(dt)
    a = flat.reshape(shape, order="F" if fortran else "C")
    if kind == "f":
        return np.ascontiguousarray(a).astype(np.float64)          # .cast<double>()
    return np.ascontiguousarray(a).astype(np.uint32).astype(np.int32).astype(np.int64)   # .cast<uint32_t>() then .cast<int>()


def _model_arrays(smpl, rng):
    V, J, K = smpl["v_template"].shape[0], smpl["weights"].shape[1], smpl["shapedirs"].shape[2]
    kt = np.asarray(smpl["kintree_table"], np.int64).astype(np.uint32)   # root parent -1 -> 4294967295 like SMPL exports
    return dict(v_template=smpl["v_template"], shapedirs=smpl["shapedirs"], J_regressor=smpl["J_regressor"], weights=smpl["weights"],
                f=np.asarray(smpl["f"], np.uint32), kintree_table=kt, posedirs=rng.standard_normal((V, 3, 9)))


@pytest.mark.parametrize("variant", ["stored", "deflated", "fortran_f32", "wide_ints_foreign"])
def test_ark_npz_reads_what_the_references_cnpy_reads(smpl, tmp_path, ark_lib, variant):
    ref = _ref_lib()
    rng = np.random.default_rng(3)
    arrs = _model_arrays(smpl, rng)
    if variant == "fortran_f32":        # Fortran-order members and float32 members (SMPL exports carry both)
        arrs["weights"] = np.asfortranarray(arrs["weights"])
        arrs["J_regressor"] = np.asfortranarray(arrs["J_regressor"].astype(np.float32))
        arrs["v_template"] = arrs["v_template"].astype(np.float32)
        arrs["f"] = np.asfortranarray(arrs["f"])
    if variant == "wide_ints_foreign":  # 64-bit index arrays plus members that are not numeric arrays at all
        arrs["f"] = arrs["f"].astype(np.uint64)
        arrs["kintree_table"] = arrs["kintree_table"].astype(np.int64)
        arrs["bs_style"] = np.array("lbs"); arrs["is_female"] = np.array(True); arrs["names"] = np.array(["a", "bcd"])
    path = str(tmp_path / "model.npz")
    (np.savez_compressed if variant == "deflated" else np.savez)(path, **arrs)
    r = _load_ref(ref, path)
    a = _load_ark(ark_lib, path)
    assert set(r) == set(arrs)                            # the reference's reader keeps every member
    for key in FLOAT_KEYS + UINT_KEYS:
        ws, fo, shape, raw = r[key]
        assert fo == np.isfortran(arrs[key]) or arrs[key].ndim < 2
        is_int, shape_a, vals = a[key]
        assert shape_a == shape == arrs[key].shape
        want = _reference_matrix(ws, fo, shape, raw, "f" if key in FLOAT_KEYS else "u")
        if key in UINT_KEYS:
            assert is_int
            got = vals.reshape(shape).astype(np.uint32).astype(np.int32).astype(np.int64)     # the facade's own cast to int
            assert np.array_equal(got, want)
        else:
            assert not is_int
            assert np.array_equal(vals.reshape(shape), want)                                  # bit-equal doubles
            assert np.array_equal(want, np.asarray(arrs[key], np.float64))                    # and both are what numpy wrote
    # members that are not numeric arrays: the reference's reader carries their bytes along, ark::npz skips them; neither fails
    for key in set(arrs) - set(FLOAT_KEYS) - set(UINT_KEYS):
        assert key in r and key not in a


def _npy_bytes(descr, fortran, shape_text, payload=b""):
    hdr = ("{'descr': '%s', 'fortran_order': %s, 'shape': (%s), }" % (descr, "True" if fortran else "False", shape_text)).encode()
    pad = 64 - (10 + len(hdr) + 1) % 64
    hdr += b" " * (pad % 64) + b"\n"
    return b"\x93NUMPY\x01\x00" + len(hdr).to_bytes(2, "little") + hdr + payload


def test_ark_npz_rejects_shapes_whose_product_wraps_and_skips_odd_headers(tmp_path, ark_lib):
    """two dimensions of 2^40 wrapped size_t (n = 0 passed the truncation check while shape said 2^80 elements); a
    member whose header makes std::stoull throw std::out_of_range aborted the whole load instead of being skipped."""
    import zipfile
    good = np.arange(12, dtype=np.float64).reshape(3, 4)
    path = str(tmp_path / "crafted.npz")
    with zipfile.ZipFile(path, "w", zipfile.ZIP_STORED) as z:
        import io
        b = io.BytesIO(); np.save(b, good); z.writestr("v_template.npy", b.getvalue())
        z.writestr("wraps.npy", _npy_bytes("<f8", False, "1099511627776, 1099511627776"))            # 2^40 x 2^40
        z.writestr("huge_dim.npy", _npy_bytes("<f8", False, "99999999999999999999999999, 2"))        # stoull: out_of_range
        z.writestr("odd_descr.npy", _npy_bytes("<f999999999999", False, "2, 2"))                     # stoi: out_of_range
    a = _load_ark(ark_lib, path)
    assert set(a) == {"v_template"}                       # the crafted members are skipped, the sound one is read
    assert np.array_equal(a["v_template"][2].reshape(3, 4), good)
    h = ark_lib.arknpz_open(path.encode())
    assert ark_lib.arknpz_error(h) == b""
    ark_lib.arknpz_close(h)
