"""GPU parity of the body-part forest inference: k_rtree_predict through the C ABI against the
oracle — labels are uint8, the bar is bit-exact — on synthetic depth renders, all interval / region / fill variants,
edge cases, a resident batch, and the tracker driven by predicted labels."""
import os

import numpy as np
import pytest

from avatar_amd import api, rtree, synth, synth_forest
from avatar_amd.tracker import FrameTracker
from oracle import rtree_oracle as ro

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "forest_small.srtr")


@pytest.fixture(scope="module")
def trees():
    return rtree.RTree(GOLD), ro.OracleRTree.load(GOLD)


def _render(smpl, seed):
    w, p, R = synth.sample_ground_truth(smpl, seed)
    xyz, mask, _ = synth.render_images(smpl, synth.pose_vertices(smpl, w, p, R), synth.identity_part_map())
    return xyz, mask, synth_forest.depth_of(xyz), (w, p, R)


def _bbox(mask):
    rr, cc = np.nonzero(mask != 255)
    return (int(cc.min()), int(rr.min())), (int(cc.max()), int(rr.max()))


def test_members_match_oracle(trees):
    g, o = trees
    assert g.numParts == o.numParts == 24 and np.array_equal(g.partMap, o.partMap) and g.partMapType == o.partMapType
    assert np.array_equal(g.leafBestMatch, o.leafBestMatch) and np.array_equal(g.feature, o.feature)


@pytest.mark.parametrize("seed", [21, 22])
def test_predict_best_bit_exact(smpl, trees, seed):
    g, o = trees
    _, mask, depth, _ = _render(smpl, seed)
    tl, br = _bbox(mask)
    for kw in (dict(interval=1, fill_in_gaps=False), dict(interval=1), dict(interval=2), dict(interval=2, fill_in_gaps=False),
               dict(interval=3), dict(interval=2, top_left=tl, bot_right=br), dict(interval=1, top_left=tl, bot_right=br, fill_in_gaps=False),
               dict(interval=2, top_left=(tl[0] + 40, tl[1] + 60), bot_right=(br[0] - 30, br[1] - 50))):
        a, b = g.predictBest(depth, **kw), o.predictBest(depth, **kw)
        assert np.array_equal(a, b), kw
    full = g.predictBest(depth, interval=1, fill_in_gaps=False)
    fg = mask != 255
    fg[0] = False
    assert (full[fg] != 255).all() and (full[~fg] == 255).all()
    assert (full[fg] == mask[fg]).mean() > 0.15                  # the toy tree is weak but far above chance (1/24)


def test_predict_best_edge_cases(trees):
    g, o = trees
    rng = np.random.default_rng(5)
    empty = np.zeros((37, 53), np.float32)
    assert (g.predictBest(empty, interval=2) == 255).all()
    noise = rng.uniform(0.3, 6.0, (37, 53)).astype(np.float32)   # every pixel foreground, probes leave the image often
    noise[rng.random((37, 53)) < 0.2] = 0
    for kw in (dict(interval=1), dict(interval=2), dict(interval=5), dict(interval=4, top_left=(3, 1), bot_right=(52, 36)),
               dict(interval=1, top_left=(10, 10), bot_right=(10, 11))):
        assert np.array_equal(g.predictBest(noise, **kw), o.predictBest(noise, **kw)), kw
    with pytest.raises(RuntimeError):
        g.predictBest(noise, interval=0)
    with pytest.raises(RuntimeError):
        g.predictBest(noise, top_left=(0, 0), bot_right=(53, 36))


def test_resident_batch_equals_single(smpl, trees):
    g, o = trees
    depths = np.stack([_render(smpl, s)[2] for s in (23, 24, 25)])
    g.upload_images(depths)
    g.predict_resident(interval=2)
    for i in range(3):
        assert np.array_equal(g.download_labels(i), o.predictBest(depths[i], interval=2))


def test_tracker_with_predicted_labels(smpl, gmodel, trees):
    """demo.cpp:196-268 end to end: depth -> predictBest(interval 2, bounding box) -> postProcess -> subsample -> optimize.
    The product pipeline must agree with the same pipeline on oracle labels (the toy tree's labels are too weak to
    judge the fit against the ground truth)."""
    g, o = trees
    xyz, mask, depth, (w, p, R) = _render(smpl, 26)
    tl, br = _bbox(mask)
    lab_g = g.predictBest(depth, interval=2, top_left=tl, bot_right=br)
    lab_o = o.predictBest(depth, interval=2, top_left=tl, bot_right=br)
    com_g = g.postProcess(lab_g, None, interval=2, top_left=tl, bot_right=br)
    com_o = o.postProcess(lab_o, None, interval=2, top_left=tl, bot_right=br)
    assert np.array_equal(lab_g, lab_o) and np.array_equal(com_g, com_o)
    ava = api.Avatar(gmodel)
    opt = api.AvatarOptimizer(ava, None, (1280, 720), g.numParts, g.partMap, max_points=8192)
    opt.betaPose, opt.betaShape = 0.05, 0.12
    trk = FrameTracker(opt, interval=3, rtree=g)
    assert trk.process_depth(xyz, (tl[1], tl[0], br[1], br[0]))
    assert np.array_equal(trk.comPre, com_o)
    # the same protocol fed with the oracle's labels lands on the same state, and the fit made progress
    ava2 = api.Avatar(gmodel)
    opt2 = api.AvatarOptimizer(ava2, None, (1280, 720), o.numParts, o.partMap, max_points=8192)
    opt2.betaPose, opt2.betaShape = 0.05, 0.12
    assert FrameTracker(opt2, interval=3).process(xyz, lab_o, (tl[1], tl[0], br[1], br[0]))
    assert np.array_equal(ava.p, ava2.p) and np.array_equal(ava.w, ava2.w) and np.array_equal(ava.r, ava2.r)
    st = opt.last_stats
    assert st.num_correspondences > 1000 and st.final_cost < st.initial_cost


def test_cpp_facade_labels_like_the_oracle(smpl, trees, tmp_path):
    """include/ark/RTree.h through tests/cpp/rtree_demo.cpp: predictBest(interval 2, bounding box) + postProcess."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp", "rtree_demo")
    assert os.path.exists(exe), "tests/cpp/rtree_demo not built (make -C avatar_amd/csrc facade)"
    _, o = trees
    _, mask, depth, _ = _render(smpl, 27)
    tl, br = _bbox(mask)
    inp, outp = str(tmp_path / "depth.bin"), str(tmp_path / "out.bin")
    with open(inp, "wb") as fh:
        np.array([depth.shape[0], depth.shape[1], tl[0], tl[1], br[0], br[1]], np.int32).tofile(fh)
        depth.tofile(fh)
    subprocess.check_call([exe, GOLD, inp, outp])
    raw = np.fromfile(outp, np.uint8)
    got = raw[:depth.size].reshape(depth.shape)
    com = np.frombuffer(raw[depth.size:].tobytes(), np.float64).reshape(-1, 2).T
    ref = o.predictBest(depth, interval=2, top_left=tl, bot_right=br)
    com_ref = o.postProcess(ref, None, interval=2, top_left=tl, bot_right=br)
    assert np.array_equal(got, ref) and np.array_equal(com, com_ref)


def _random_tree(rng, depth, num_parts):
    """Random full-ish binary tree in parent-before-children order with random probe offsets and thresholds."""
    feature, links, leaves = [], [], []
    todo = [(0, -1, 0)]
    while todo:
        dep, parent, side = todo.pop(0)
        me = len(feature)
        if parent >= 0:
            links[parent][side] = me
        if dep < depth and (dep < 2 or rng.random() < 0.8):
            u, v = rng.uniform(-60, 60, 2), rng.uniform(-60, 60, 2)
            feature.append([u[0], u[1], v[0], v[1], rng.normal(0, 0.4)]); links.append([-1, -1, -1])
            todo.append((dep + 1, me, 0)); todo.append((dep + 1, me, 1))
        else:
            feature.append([0, 0, 0, 0, 0]); links.append([-1, -1, len(leaves)])
            d = rng.random(num_parts) * (rng.random(num_parts) < 0.4)
            if d.sum() == 0:
                d[rng.integers(num_parts)] = 1.0
            leaves.append(d / d.sum())
    return np.asarray(feature, np.float32), np.asarray(links, np.int32), np.asarray(leaves, np.float32), num_parts


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_random_trees_and_images_bit_exact(seed):
    rng = np.random.default_rng(seed)
    f, l, d, npp = _random_tree(rng, depth=int(rng.integers(3, 12)), num_parts=int(rng.integers(2, 40)))
    g, o = rtree.RTree.from_arrays(f, l, d, npp), ro.OracleRTree.from_arrays(f, l, d, npp)
    assert np.array_equal(g.leafBestMatch, o.leafBestMatch)
    for _ in range(6):
        H, W = int(rng.integers(1, 90)), int(rng.integers(1, 120))
        depth = rng.choice([0.0, 0.6, 1.5, 2.5, 7.0], (H, W), p=[0.3, 0.1, 0.3, 0.2, 0.1]).astype(np.float32)
        depth *= (1 + 0.05 * rng.standard_normal((H, W))).astype(np.float32)
        interval = int(rng.integers(1, 6))
        x0, y0 = int(rng.integers(0, W)), int(rng.integers(0, H))
        x1, y1 = int(rng.integers(x0, W)), int(rng.integers(y0, H))
        for kw in (dict(interval=interval), dict(interval=interval, fill_in_gaps=False),
                   dict(interval=interval, top_left=(x0, y0), bot_right=(x1, y1))):
            assert np.array_equal(g.predictBest(depth, **kw), o.predictBest(depth, **kw)), (H, W, kw)


def test_predict_distributions_bit_exact(smpl, trees):
    """RTree::predict(depth): leaf distributions per pixel, probes bounded by the image, every row."""
    g, o = trees
    _, mask, depth, _ = _render(smpl, 28)
    tile = np.ascontiguousarray(depth[200:520:2, 400:900:2])
    a, b = g.predict(tile), o.predict(tile)
    assert a.shape == (24,) + tile.shape and np.array_equal(a, b)
    fg = tile > 0
    assert np.allclose(a.sum(0)[fg], 1.0, atol=1e-5) and (a[:, ~fg] == 0).all()
