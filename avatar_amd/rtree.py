"""`ark::RTree` (RTree.h:12-184) over the C ABI of include/avt_rtree.h: the body-part forest that labels a foreground
depth image right before AvatarOptimizer::optimize() (demo.cpp:196-268).  SURVEY.md §8 row f4.

Inference runs on the GPU (avatar_amd/csrc/avt_rtree.hip); there is no CPU fallback: without libavatar_hip.so every
call raises."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi

RTREE_SYMBOLS = [
    "avt_rtree_create", "avt_rtree_load", "avt_rtree_export", "avt_rtree_destroy", "avt_rtree_info", "avt_rtree_get",
    "avt_rtree_predict_best", "avt_rtree_predict", "avt_rtree_images_upload", "avt_rtree_predict_best_resident", "avt_rtree_labels_download",
    "avt_rtree_sync", "avt_rtree_post_process",
]


class RTreeDesc(C.Structure):
    _fields_ = [("n_nodes", C.c_int), ("n_leafs", C.c_int), ("num_parts", C.c_int), ("feature", C.POINTER(C.c_float)),
                ("links", C.POINTER(C.c_int)), ("leaf_data", C.POINTER(C.c_float)), ("part_map_len", C.c_int),
                ("part_map", C.POINTER(C.c_int)), ("part_map_type", C.c_int)]


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _up(a):
    return a.ctypes.data_as(C.POINTER(C.c_ubyte))


def _check(lib, rc):
    if rc != 0:
        lib.avt_last_error.restype = C.c_char_p
        raise RuntimeError((lib.avt_last_error() or b"?").decode())


class RTree:
    """Same members and call protocol as the reference class: numParts, partMap, nodes / leafData (as arrays),
    loadFile, exportFile, predictBest, postProcess."""

    def __init__(self, path: str | None = None, device: int = 0):
        self._lib = capi.load_library()
        self._h = C.c_void_p()
        self.device = device
        self.numParts = 0
        self.partMap = np.zeros(0, np.int32)
        self.partMapType = 0
        if path is not None and not self.loadFile(path):
            raise RuntimeError("RTree failed to initialize from %s" % path)     # RTree.cpp:2961-2965

    @classmethod
    def from_arrays(cls, feature, links, leaf_data, num_parts, part_map=None, part_map_type=0, device=0):
        """feature (n,5) float32 [u.x u.y v.x v.y thresh], links (n,3) int32 [lnode rnode leafid], leaf_data (nl, num_parts)."""
        self = cls(None, device)
        f = np.ascontiguousarray(feature, np.float32); l = np.ascontiguousarray(links, np.int32)
        d = np.ascontiguousarray(leaf_data, np.float32).reshape(-1, num_parts)
        pm = np.ascontiguousarray(part_map if part_map is not None else np.zeros(0), np.int32)
        desc = RTreeDesc(len(l), len(d), num_parts, _fp(f), _ip(l), _fp(d), len(pm), _ip(pm), part_map_type)
        _check(self._lib, self._lib.avt_rtree_create(C.byref(desc), C.c_int(device), C.byref(self._h)))
        self._refresh()
        return self

    def __del__(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.avt_rtree_destroy(self._h)
            self._h = C.c_void_p()

    def _refresh(self):
        n, nl, np_, pml, pmt = (C.c_int() for _ in range(5))
        _check(self._lib, self._lib.avt_rtree_info(self._h, C.byref(n), C.byref(nl), C.byref(np_), C.byref(pml), C.byref(pmt)))
        self.numParts, self.partMapType = np_.value, pmt.value
        self.feature = np.empty((n.value, 5), np.float32); self.links = np.empty((n.value, 3), np.int32)
        self.leafData = np.empty((nl.value, np_.value), np.float32); self.leafBestMatch = np.empty(nl.value, np.uint8)
        self.partMap = np.empty(pml.value, np.int32)
        _check(self._lib, self._lib.avt_rtree_get(self._h, _fp(self.feature), _ip(self.links), _fp(self.leafData), _up(self.leafBestMatch),
                                                  _ip(self.partMap)))

    def loadFile(self, path: str) -> bool:
        if self._h.value:
            self._lib.avt_rtree_destroy(self._h)
            self._h = C.c_void_p()
        if self._lib.avt_rtree_load(path.encode(), C.c_int(self.device), C.byref(self._h)) != 0:
            return False
        self._refresh()
        return True

    def exportFile(self, path: str) -> bool:
        return self._lib.avt_rtree_export(self._h, path.encode()) == 0

    def predictBest(self, depth, num_threads=0, interval=1, top_left=(0, 0), bot_right=(-1, -1), fill_in_gaps=True):
        """cv::Mat RTree::predictBest(depth, num_threads, interval, top_left, bot_right, fill_in_gaps); points are (x, y)."""
        d = np.ascontiguousarray(depth, np.float32)
        out = np.empty(d.shape, np.uint8)
        _check(self._lib, self._lib.avt_rtree_predict_best(self._h, _fp(d), C.c_int(d.shape[0]), C.c_int(d.shape[1]), C.c_int(interval),
                                                           C.c_int(top_left[0]), C.c_int(top_left[1]), C.c_int(bot_right[0]),
                                                           C.c_int(bot_right[1]), C.c_int(1 if fill_in_gaps else 0), _up(out)))
        return out

    def predict(self, depth):
        """std::vector<cv::Mat> RTree::predict(depth): (numParts, H, W) float32 leaf distributions (0 where depth <= 0)."""
        d = np.ascontiguousarray(depth, np.float32)
        out = np.empty((self.numParts,) + d.shape, np.float32)
        _check(self._lib, self._lib.avt_rtree_predict(self._h, _fp(d), C.c_int(d.shape[0]), C.c_int(d.shape[1]), _fp(out)))
        return out

    def postProcess(self, image, com_pre=None, interval=1, num_threads=1, top_left=(0, 0), bot_right=(-1, -1), dist_to_pre_weight=0.001):
        """In-place on `image` (H,W) uint8; com_pre (2, numParts) float64 is updated and returned (None: first frame)."""
        assert image.dtype == np.uint8 and image.flags.c_contiguous
        valid = com_pre is not None and com_pre.shape == (2, self.numParts)
        cp = np.ascontiguousarray(com_pre.T, np.float64) if valid else np.zeros((self.numParts, 2))
        _check(self._lib, self._lib.avt_rtree_post_process(self._h, _up(image), C.c_int(image.shape[0]), C.c_int(image.shape[1]), capi.dptr(cp),
                                                           C.c_int(1 if valid else 0), C.c_int(interval), C.c_int(top_left[0]),
                                                           C.c_int(top_left[1]), C.c_int(bot_right[0]), C.c_int(bot_right[1]),
                                                           C.c_double(dist_to_pre_weight)))
        return np.ascontiguousarray(cp.T)

    # ---- resident batch (bench.py) ----
    def upload_images(self, depth_stack):
        d = np.ascontiguousarray(depth_stack, np.float32)
        _check(self._lib, self._lib.avt_rtree_images_upload(self._h, C.c_int(d.shape[0]), C.c_int(d.shape[1]), C.c_int(d.shape[2]), _fp(d)))
        self._shape = d.shape

    def predict_resident(self, interval=1, top_left=(0, 0), bot_right=(-1, -1), fill_in_gaps=True):
        _check(self._lib, self._lib.avt_rtree_predict_best_resident(self._h, C.c_int(interval), C.c_int(top_left[0]), C.c_int(top_left[1]),
                                                                    C.c_int(bot_right[0]), C.c_int(bot_right[1]), C.c_int(1 if fill_in_gaps else 0)))

    def sync(self):
        _check(self._lib, self._lib.avt_rtree_sync(self._h))

    def download_labels(self, image):
        out = np.empty(self._shape[1:], np.uint8)
        _check(self._lib, self._lib.avt_rtree_labels_download(self._h, C.c_int(image), _up(out)))
        return out
