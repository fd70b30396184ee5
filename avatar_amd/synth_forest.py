"""A small body-part forest trained on synthetic renders, so that the RTree stage has a model to
run: the reference ships none (its trainers, RTree.cpp:330-2955, need hours on rendered SMPL data).  One tree, depth-
difference features (u, v) / depth as in scoreByFeature (RTree.cpp:53-68), greedy information-gain splits on sampled
foreground pixels — the scheme of the reference's trainers at toy scale.  numpy only; deterministic for a given seed.
"""
from __future__ import annotations

import numpy as np

from . import synth

BACKGROUND_DEPTH = np.float32(20.0)   # RTree.cpp:325


def depth_of(xyz):
    """float32 depth image (metres, 0 = background) from the XYZ map of synth.render_images."""
    return np.ascontiguousarray(xyz[:, :, 2], np.float32)


def _probe(depth_stack, img, r, c, off, sample):
    """depth at pixel + round(off / sample), BACKGROUND_DEPTH outside the image or on background (float32 arithmetic)."""
    H, W = depth_stack.shape[1:]
    t = (off[None, :] / sample[:, None]).astype(np.float32)
    pc = np.where(t >= 0, np.floor(t + np.float32(0.5)), np.ceil(t - np.float32(0.5))).astype(np.int64)   # std::round
    x, y = pc[:, 0] + c, pc[:, 1] + r
    inside = (x >= 0) & (y >= 0) & (x < W) & (y < H)
    z = np.full(len(r), BACKGROUND_DEPTH, np.float32)
    zi = depth_stack[img[inside], y[inside], x[inside]]
    z[inside] = np.where(zi == 0, BACKGROUND_DEPTH, zi)
    return z


def train(model, num_images=24, points_per_image=1500, num_features=40, threshes_per_feature=8, max_probe_offset=170.0,
          min_samples=40, max_depth=13, seed=0, part_map=None):
    """Returns (feature (n,5) float32, links (n,3) int32, leaf_data (nl, num_parts) float32, num_parts)."""
    rng = np.random.default_rng(seed)
    part_map = synth.identity_part_map() if part_map is None else np.asarray(part_map, np.int32)
    num_parts = int(part_map.max()) + 1
    depths, imgs, rows, cols, labs = [], [], [], [], []
    for i in range(num_images):
        w, p, R = synth.sample_ground_truth(model, 5000 + i)
        xyz, mask, _ = synth.render_images(model, synth.pose_vertices(model, w, p, R), part_map)
        depths.append(depth_of(xyz))
        rr, cc = np.nonzero(mask != 255)
        pick = rng.choice(len(rr), min(points_per_image, len(rr)), replace=False)
        imgs.append(np.full(len(pick), i)); rows.append(rr[pick]); cols.append(cc[pick]); labs.append(mask[rr[pick], cc[pick]])
    D = np.stack(depths)
    img, r, c, y = (np.concatenate(a).astype(np.int64) for a in (imgs, rows, cols, labs))
    sample = D[img, r, c]

    def entropy(counts):
        n = counts.sum(-1, keepdims=True)
        p = counts / np.maximum(n, 1)
        return -(np.where(p > 0, p * np.log2(np.maximum(p, 1e-30)), 0.0)).sum(-1)

    feature, links, leaves = [], [], []
    todo = [(np.arange(len(y)), 0, -1, 0)]            # (sample ids, depth, parent, side)
    while todo:
        ids, dep, parent, side = todo.pop(0)
        me = len(feature)
        if parent >= 0:
            links[parent][side] = me
        hist = np.bincount(y[ids], minlength=num_parts).astype(np.float64)
        best = None
        if dep < max_depth and len(ids) >= min_samples and (hist > 0).sum() > 1:
            base = entropy(hist)
            u = rng.uniform(-max_probe_offset, max_probe_offset, (num_features, 2)).astype(np.float32)
            v = rng.uniform(-max_probe_offset, max_probe_offset, (num_features, 2)).astype(np.float32)
            v[rng.random(num_features) < 0.5] = 0      # half the features compare against the pixel's own depth
            for k in range(num_features):
                score = _probe(D, img[ids], r[ids], c[ids], u[k], sample[ids]) - _probe(D, img[ids], r[ids], c[ids], v[k], sample[ids])
                for th in rng.choice(score, min(threshes_per_feature, len(score)), replace=False):
                    left = score < th
                    nl = int(left.sum())
                    if nl == 0 or nl == len(ids):
                        continue
                    hl = np.bincount(y[ids][left], minlength=num_parts).astype(np.float64)
                    gain = base - (nl * entropy(hl) + (len(ids) - nl) * entropy(hist - hl)) / len(ids)
                    if best is None or gain > best[0]:
                        best = (gain, u[k], v[k], np.float32(th), left)
        if best is None or best[0] <= 1e-9:
            feature.append([0, 0, 0, 0, 0]); links.append([-1, -1, len(leaves)])
            leaves.append((hist / hist.sum()).astype(np.float32))
        else:
            _, bu, bv, th, left = best
            feature.append([bu[0], bu[1], bv[0], bv[1], th]); links.append([-1, -1, -1])
            todo.append((ids[left], dep + 1, me, 0)); todo.append((ids[~left], dep + 1, me, 1))
    return (np.asarray(feature, np.float32), np.asarray(links, np.int32), np.asarray(leaves, np.float32).reshape(-1, num_parts), num_parts)


def write_part_map(path, part_map, contiguous=True):
    """<tree>.partmap in the layout RTree::readPartMap parses (RTree.cpp:3465-3509)."""
    part_map = np.asarray(part_map)
    n_src, n_dst = len(part_map), int(part_map.max()) + 1
    with open(path, "w") as f:
        f.write("partmap %s\nsrc %d\n" % ("contiguous" if contiguous else "disjoint", n_src))
        f.write(" ".join("j%d" % i for i in range(n_src)) + "\n")
        f.write("dest %d\n" % n_dst)
        f.write(" ".join("p%d" % i for i in range(n_dst)) + "\n")
        for i in range(n_src):
            f.write("j%d p%d\n" % (i, part_map[i]))
