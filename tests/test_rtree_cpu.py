"""CPU tests of the body-part forest stage: the oracle against hand-computed known answers of
RTree.cpp's arithmetic, the file formats, the part map, and the product's host code (file IO, best-match table,
post-processing) against the oracle.  No GPU needed: the product tree is created host-only (device = -1)."""
import ctypes
import os
import re

import numpy as np
import pytest

from avatar_amd import capi, rtree
from oracle import rtree_oracle as ro

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "forest_small.srtr")


def _host_tree(path=None, arrays=None, part_map=None, part_map_type=0):
    if path is not None:
        t = rtree.RTree(None, device=-1)
        assert t.loadFile(path)
        return t
    f, l, d, npp = arrays
    return rtree.RTree.from_arrays(f, l, d, npp, part_map=part_map, part_map_type=part_map_type, device=-1)


def _stump():
    """root: u=(3,0), v=(0,-2), thresh 0.5 -> left leaf (part 1) / right leaf (part 0)"""
    f = np.array([[3, 0, 0, -2, 0.5], [0, 0, 0, 0, 0], [0, 0, 0, 0, 0]], np.float32)
    l = np.array([[1, 2, -1], [-1, -1, 0], [-1, -1, 1]], np.int32)
    d = np.array([[0.2, 0.8], [0.6, 0.4]], np.float32)
    return f, l, d, 2


def test_abi_exports_every_symbol_of_avt_rtree_h():
    hdr = open(os.path.join(ROOT, "include", "avt_rtree.h")).read()
    declared = set(re.findall(r"\b(avt_rtree_[a-z_]+)\s*\(", hdr))
    assert declared == set(rtree.RTREE_SYMBOLS), declared ^ set(rtree.RTREE_SYMBOLS)
    lib = capi.load_library()
    for s in declared:
        assert hasattr(lib, s), s


def test_oracle_known_answers_of_the_feature_arithmetic():
    """scoreByFeature by hand (RTree.cpp:53-68, :3209-3246): offsets are divided by the pixel's depth, rounded half away
    from zero, probes outside the REGION OF INTEREST or on background read 20 m, and row top_left.y itself is skipped."""
    t = ro.OracleRTree.from_arrays(*_stump())
    assert list(t.leafBestMatch) == [1, 0]                       # first strict maximum (RTree.cpp:3451-3463)
    depth = np.zeros((8, 10), np.float32)
    depth[2:7, 2:8] = 2.0
    depth[3, 5] = 1.0
    out = t.predictBest(depth, interval=1, fill_in_gaps=False)
    assert (out[0] == 255).all() and (out[depth == 0] == 255).all()
    # pixel (r=3, c=5), depth 1: u -> (8, 3): depth 0 -> 20; v -> (5, 1): background -> 20; 0 < 0.5 -> left -> part 1
    assert out[3, 5] == 1
    # pixel (r=4, c=4), depth 2: u = 1.5 -> rounds to 2 -> (6, 4) = 2.0; v = -1 -> (4, 3) = 2.0; 0 < 0.5 -> part 1
    assert out[4, 4] == 1
    # pixel (r=4, c=6), depth 2: u -> (8, 4) = 0 -> 20; v -> (6, 3) = 2.0; 18 >= 0.5 -> right -> part 0
    assert out[4, 6] == 0
    # pixel (r=3, c=4), depth 2: u -> (6, 3) = 2.0; v -> (4, 2) = 2.0 -> part 1; with a region of interest that excludes
    # row 2 the v probe reads background instead: 2 - 20 < 0.5 -> still part 1; excluding column 6 flips u to 20 -> part 0
    assert out[3, 4] == 1
    roi = t.predictBest(depth, interval=1, top_left=(2, 2), bot_right=(5, 6), fill_in_gaps=False)
    assert roi[3, 4] == 0 and (roi[2] == 255).all()              # row top_left.y is never labelled
    # interval 2 with fill: rows 2, 4, 6 (first row skipped), cells are copied to the right / below
    up = t.predictBest(depth, interval=2, fill_in_gaps=True)
    assert (up[0:2] == 255).all() and up[4, 4] == up[5, 5] == up[4, 5] == 1


def test_file_formats_and_part_map(tmp_path):
    t = ro.OracleRTree.load(GOLD)
    assert t.numParts == 24 and len(t.partMap) == 24 and t.partMapType == 0 and (t.partMap == np.arange(24)).all()
    # binary round trip through the oracle and through the product
    p1, p2 = str(tmp_path / "a.srtr"), str(tmp_path / "b.srtr")
    assert t.export(p1)
    assert open(p1, "rb").read() == open(GOLD, "rb").read()
    prod = _host_tree(path=GOLD)
    assert prod.exportFile(p2) and open(p2, "rb").read() == open(GOLD, "rb").read()
    for a, b in ((prod.feature, t.feature), (prod.links, t.links), (prod.leafData, t.leafData), (prod.leafBestMatch, t.leafBestMatch),
                 (prod.partMap, t.partMap)):
        assert np.array_equal(a, b)
    # legacy text format (RTree.cpp:3020-3048)
    f, l, d, npp = _stump()
    txt = str(tmp_path / "legacy.txt")
    with open(txt, "w") as fh:
        fh.write("3 2 2\n-1 1 2 0.5 3 0 0 -2\n0\n1\n0.2 0.8\n0.6 0.4\n")
    for tree in (ro.OracleRTree.load(txt), _host_tree(path=txt)):
        assert np.array_equal(tree.links, l) and np.allclose(tree.feature[0], f[0]) and np.allclose(tree.leafData, d)
    # a 'disjoint' part map that merges parts
    pm = str(tmp_path / "c.srtr")
    assert t.export(pm)
    with open(pm + ".partmap", "w") as fh:
        fh.write("partmap disjoint\nsrc 3\nhead arm leg\ndest 2\nupper lower\nleg lower\nhead upper\narm upper\n")
    for tree in (ro.OracleRTree.load(pm), _host_tree(path=pm)):
        assert tree.partMapType == 1 and list(tree.partMap) == [0, 0, 1]


def test_product_rejects_malformed_trees(tmp_path):
    f, l, d, npp = _stump()
    bad = l.copy(); bad[0, 0] = 0                                 # child pointing at its parent: a cycle
    with pytest.raises(RuntimeError):
        rtree.RTree.from_arrays(f, bad, d, npp, device=-1)
    with pytest.raises(RuntimeError):
        rtree.RTree.from_arrays(f, l, d[:, :1].repeat(200, 1), 200, device=-1)      # labels must stay below 128
    p = str(tmp_path / "trunc.srtr")
    open(p, "wb").write(open(GOLD, "rb").read()[:1000])
    assert not rtree.RTree(None, device=-1).loadFile(p)
    host_only = _host_tree(arrays=_stump())
    with pytest.raises(RuntimeError):                             # no CPU fallback for inference
        host_only.predictBest(np.ones((4, 4), np.float32))


def _blobs(rng, rows=40, cols=56, parts=4):
    img = np.full((rows, cols), 255, np.uint8)
    for _ in range(14):
        r, c = rng.integers(0, rows - 6), rng.integers(0, cols - 8)
        img[r:r + rng.integers(2, 7), c:c + rng.integers(2, 9)] = rng.integers(0, parts)
    return img


@pytest.mark.parametrize("interval", [1, 2])
@pytest.mark.parametrize("ptype", [0, 1])
def test_post_process_product_equals_oracle(interval, ptype, tmp_path):
    """Largest-component selection / small-piece removal incl. the centre-of-mass memory across frames and the
    interval > 1 behaviour on up-scaled images (RTree.cpp:125-323, :3422-3449)."""
    rng = np.random.default_rng(10 * interval + ptype)
    f, l, d, _ = _stump()
    d4 = np.zeros((2, 4), np.float32); d4[:, :2] = d
    path = str(tmp_path / "t.srtr")
    assert ro.OracleRTree.from_arrays(f, l, d4, 4).export(path)
    with open(path + ".partmap", "w") as fh:
        fh.write("partmap %s\nsrc 4\na b c d\ndest 4\nw x y z\na w\nb x\nc y\nd z\n" % ("disjoint" if ptype else "contiguous"))
    orc, prod = ro.OracleRTree.load(path), _host_tree(path=path)
    assert prod.partMapType == ptype
    com_o = com_p = None
    for frame in range(3):
        img = _blobs(rng)
        if interval > 1:                                          # what predictBest(..., fill_in_gaps=true) hands over
            img = np.repeat(np.repeat(img[::interval, ::interval], interval, 0), interval, 1)[:40, :56].copy()
        a, b = img.copy(), img.copy()
        roi = dict(top_left=(2, 2), bot_right=(51, 35)) if frame == 1 else {}
        com_o = orc.postProcess(a, com_o, interval=interval, dist_to_pre_weight=0.01, **roi)
        com_p = prod.postProcess(b, com_p, interval=interval, dist_to_pre_weight=0.01, **roi)
        assert np.array_equal(a, b)
        assert np.array_equal(com_o, com_p)
        if ptype == 0 and interval == 1 and not roi:              # one 4-connected blob per part survives
            from scipy import ndimage
            for part in range(4):
                assert ndimage.label(a == part)[1] <= 1
        assert not ((a >= 128) & (a != 255)).any()
