"""ctypes binding of oracle/librender_oracle.so: the reference's painter's-order renderer restated (render_oracle.cpp).
TEST INFRASTRUCTURE ONLY: imported by tests/ only."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "librender_oracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "render_oracle.cpp")
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "librender_oracle.so"], stdout=subprocess.DEVNULL)
        _lib = C.CDLL(_SO)
        _lib.orc_backproject.restype = C.c_int
    return _lib


def render(cloud, mesh, vertex_part, intrin, width, height, stable=False, return_ties=False):
    """(depth (H,W) float32, part mask (H,W) uint8) of posed vertices `cloud` (V,3): AvatarRenderer::renderDepth / renderPartMask.
    stable=True orders faces of equal sort key by face id instead of leaving them to std::sort; return_ties adds the number of
    equal adjacent keys in the sorted order."""
    cloud = np.ascontiguousarray(cloud, np.float64); mesh = np.ascontiguousarray(mesh, np.int32)
    vp = np.ascontiguousarray(vertex_part, np.int32)
    depth = np.empty((height, width), np.float32); mask = np.empty((height, width), np.uint8)
    ties = C.c_int(0)
    lib().orc_render_ex(C.c_int(cloud.shape[0]), C.c_int(mesh.shape[0]), cloud.ctypes.data_as(C.POINTER(C.c_double)),
                        mesh.ctypes.data_as(C.POINTER(C.c_int)), vp.ctypes.data_as(C.POINTER(C.c_int)),
                        C.c_float(intrin["fx"]), C.c_float(intrin["fy"]), C.c_float(intrin["cx"]), C.c_float(intrin["cy"]),
                        C.c_int(width), C.c_int(height), depth.ctypes.data_as(C.POINTER(C.c_float)), mask.ctypes.data_as(C.POINTER(C.c_ubyte)),
                        C.c_int(1 if stable else 0), C.byref(ties))
    return (depth, mask, ties.value) if return_ties else (depth, mask)


def backproject(depth, mask, intrin):
    """(data (N,3) float64, labels (N,) int32): optim.cpp:104-120."""
    H, W = depth.shape
    cap = int((depth > 0).sum())
    xyz = np.empty((cap, 3), np.float64); lab = np.empty(cap, np.int32)
    n = lib().orc_backproject(C.c_int(W), C.c_int(H), np.ascontiguousarray(depth).ctypes.data_as(C.POINTER(C.c_float)),
                              np.ascontiguousarray(mask).ctypes.data_as(C.POINTER(C.c_ubyte)), C.c_float(intrin["fx"]), C.c_float(intrin["fy"]),
                              C.c_float(intrin["cx"]), C.c_float(intrin["cy"]), C.c_int(cap), xyz.ctypes.data_as(C.POINTER(C.c_double)),
                              lab.ctypes.data_as(C.POINTER(C.c_int)))
    assert n == cap
    return xyz, lab
