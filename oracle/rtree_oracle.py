"""ctypes view of oracle/librtree_oracle.so (CPU restatement of the reference's RTree inference; TEST INFRASTRUCTURE:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this)."""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_DIR, "librtree_oracle.so")
_L = None


def lib():
    global _L
    if _L is None:
        if not os.path.exists(_SO):
            subprocess.check_call(["make", "-C", _DIR, "librtree_oracle.so"], stdout=subprocess.DEVNULL)
        _L = C.CDLL(_SO)
        _L.orc_rtree_create.restype = C.c_void_p
        _L.orc_rtree_load.restype = C.c_void_p
    return _L


def _fp(a): return a.ctypes.data_as(C.POINTER(C.c_float))
def _ip(a): return a.ctypes.data_as(C.POINTER(C.c_int))
def _up(a): return a.ctypes.data_as(C.POINTER(C.c_ubyte))


class OracleRTree:
    def __init__(self, handle):
        self._h = C.c_void_p(handle)
        n, nl, npp, pml, pmt = (C.c_int() for _ in range(5))
        lib().orc_rtree_dims(self._h, C.byref(n), C.byref(nl), C.byref(npp), C.byref(pml), C.byref(pmt))
        self.numParts, self.partMapType = npp.value, pmt.value
        self.feature = np.empty((n.value, 5), np.float32); self.links = np.empty((n.value, 3), np.int32)
        self.leafData = np.empty((nl.value, npp.value), np.float32); self.leafBestMatch = np.empty(nl.value, np.uint8)
        self.partMap = np.empty(pml.value, np.int32)
        lib().orc_rtree_get(self._h, _fp(self.feature), _ip(self.links), _fp(self.leafData), _up(self.leafBestMatch), _ip(self.partMap))

    @classmethod
    def from_arrays(cls, feature, links, leaf_data, num_parts):
        f = np.ascontiguousarray(feature, np.float32); l = np.ascontiguousarray(links, np.int32)
        d = np.ascontiguousarray(leaf_data, np.float32).reshape(-1, num_parts)
        return cls(lib().orc_rtree_create(C.c_int(len(l)), _fp(f), _ip(l), C.c_int(len(d)), _fp(d), C.c_int(num_parts)))

    @classmethod
    def load(cls, path):
        h = lib().orc_rtree_load(path.encode())
        if not h:
            raise RuntimeError("oracle: cannot load " + path)
        return cls(h)

    def export(self, path):
        return lib().orc_rtree_export(self._h, path.encode()) == 0

    def __del__(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().orc_rtree_destroy(self._h)
            self._h = C.c_void_p()

    def predictBest(self, depth, interval=1, top_left=(0, 0), bot_right=(-1, -1), fill_in_gaps=True):
        d = np.ascontiguousarray(depth, np.float32)
        out = np.empty(d.shape, np.uint8)
        lib().orc_rtree_predict_best(self._h, _fp(d), C.c_int(d.shape[0]), C.c_int(d.shape[1]), C.c_int(interval), C.c_int(top_left[0]),
                                     C.c_int(top_left[1]), C.c_int(bot_right[0]), C.c_int(bot_right[1]), C.c_int(1 if fill_in_gaps else 0), _up(out))
        return out

    def predict(self, depth):
        """(numParts, H, W) float32 distributions (RTree::predict(depth))."""
        d = np.ascontiguousarray(depth, np.float32)
        out = np.empty((self.numParts,) + d.shape, np.float32)
        lib().orc_rtree_predict(self._h, _fp(d), C.c_int(d.shape[0]), C.c_int(d.shape[1]), _fp(out))
        return out

    def postProcess(self, image, com_pre=None, interval=1, top_left=(0, 0), bot_right=(-1, -1), dist_to_pre_weight=0.001):
        valid = com_pre is not None and com_pre.shape == (2, self.numParts)
        cp = np.ascontiguousarray(com_pre.T, np.float64) if valid else np.zeros((self.numParts, 2))
        lib().orc_rtree_post_process(self._h, _up(image), C.c_int(image.shape[0]), C.c_int(image.shape[1]),
                                     cp.ctypes.data_as(C.POINTER(C.c_double)), C.c_int(1 if valid else 0), C.c_int(interval), C.c_int(top_left[0]),
                                     C.c_int(top_left[1]), C.c_int(bot_right[0]), C.c_int(bot_right[1]), C.c_double(dist_to_pre_weight))
        return np.ascontiguousarray(cp.T)
