#!/usr/bin/env python3
"""CPU simulation of the slab scan of k_nn_part (avt_nn.hip): which fraction of (query, candidate) pairs a wave of 64 consecutive
bucketed queries evaluates when the part's visible candidates are sorted by y and the scan walks outwards from the slab in rounds of
CH candidates a side until the y gap alone exceeds the wave's worst best distance.  `order`: how the queries of a part are ordered
inside their bucket - "pixel" (original row-major order: what a stable bucketing gives) or "tile" (the unordered scatter: tiles of
2048 points in arbitrary order, arbitrary order inside a tile).  Usage: nn_slab_sim.py [CH]"""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from avatar_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

CH = int(sys.argv[1]) if len(sys.argv) > 1 else 32
smpl = synth.load_model(0); om = orc.OracleModel(smpl); pm = synth.identity_part_map()
part = np.asarray(pm)[synth.main_joint(smpl)]


def sim(seed, dense, order):
    fr = synth.make_frame(smpl, seed, dense=dense)
    w0, p0, R0 = fr["start"]
    cloud = synth.pose_vertices(smpl, w0, p0, R0)
    vis = om.visibility(cloud)
    data, lab = fr["data"], fr["labels"]
    rng = np.random.default_rng(seed)
    tot = ev = 0
    for q in range(24):
        cand = np.nonzero((part == q) & (vis != 0))[0]
        qi = np.nonzero(lab == q)[0]
        if len(cand) == 0 or len(qi) == 0:
            continue
        if order == "tile":
            tiles = qi // 2048
            key = rng.permutation(tiles.max() + 1)[tiles] * 1e6 + rng.random(len(qi))
            qi = qi[np.argsort(key)]
        C = cloud[cand]; C = C[np.argsort(C[:, 1], kind="stable")]; ty = C[:, 1]; n = len(C)
        Q = data[qi]
        for s in range(0, len(qi), 64):
            qq = Q[s:s + 64]
            ylo, yhi = qq[:, 1].min(), qq[:, 1].max()
            st = int(np.searchsorted(ty, 0.5 * (ylo + yhi))) & ~3
            R = L = st
            best = np.full(len(qq), np.inf)
            rdone, ldone = R >= n, L <= 0
            while not (rdone and ldone):
                if not rdone:
                    e = min(R + CH, n); best = np.minimum(best, ((qq[:, None, :] - C[None, R:e, :]) ** 2).sum(-1).min(1)); ev += (e - R) * len(qq); R = e
                if not ldone:
                    b = max(L - CH, 0); best = np.minimum(best, ((qq[:, None, :] - C[None, b:L, :]) ** 2).sum(-1).min(1)); ev += (L - b) * len(qq); L = b
                bound = best.max() * (1 + 1e-12)
                rdone = rdone or R >= n or (ty[R] > yhi and (ty[R] - yhi) ** 2 > bound)
                ldone = ldone or L <= 0 or (ty[L - 1] < ylo and (ylo - ty[L - 1]) ** 2 > bound)
            tot += n * len(qq)
    return tot, ev


for dense in (False, True):
    for order in ("pixel", "tile"):
        T = E = 0
        for seed in (0, 1, 2, 3):
            t, e = sim(seed, dense, order); T += t; E += e
        print(("dense " if dense else "sparse"), f"queries in {order:5s} order, {CH} candidates a side per round: evaluated fraction {E / T:.3f}")
