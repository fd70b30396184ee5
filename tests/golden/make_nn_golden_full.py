#!/usr/bin/env python3
"""FULL-SIZE nearest-neighbour goldens from the reference's own nanoflann KD-tree search (build container only).

Complements make_nn_golden.py (sub-sampled queries with inputs stored): here the inputs are NOT stored - they are
regenerated from seeds (synth.make_frame is deterministic) - and only the reference's output index arrays are committed:
    case 0: frame seed 0, ~38 k queries (BASELINE configs[1] size), identity part map, 24 parts
    case 1: frame seed 3, dense ~151 k queries (configs[4] size), identity part map
    case 2: frame seed 5, ~38 k queries, coarse 6-part map
Model cloud = Avatar::update() of the frame's start state, visibility = back-face test, exactly what findNN sees in the
first ICP iteration (AvatarOptimizer.cpp:841-907).  Usage: python tests/golden/make_nn_golden_full.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from avatar_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

COARSE = np.array([0, 1, 2, 0, 1, 2, 0, 1, 2, 0, 1, 2, 3, 3, 3, 3, 4, 5, 4, 5, 4, 5, 4, 5], np.int32)
CASES = [dict(seed=0, dense=False, coarse=False), dict(seed=3, dense=True, coarse=False), dict(seed=5, dense=False, coarse=True)]


def case_inputs(smpl, om, c):
    """(part_map, num_parts, model cloud, visibility, data, labels) of a case, from its seed."""
    pm, npart = (COARSE, 6) if c["coarse"] else (synth.identity_part_map(), 24)
    fr = synth.make_frame(smpl, c["seed"], dense=c["dense"], part_map=pm)
    w0, p0, R0 = fr["start"]
    cloud, _, _ = om.update(w0, p0, R0)
    vis = om.visibility(cloud, True)
    return pm, npart, cloud, vis, fr["data"], fr["labels"]


def main():
    assert orc.have_reference_nn(), "oracle/_ref not built: run `make -C oracle ref` in the build container"
    smpl = synth.load_model(0)
    om = orc.OracleModel(smpl)
    mj = om.main_joint()
    out = {"ncase": len(CASES)}
    for k, c in enumerate(CASES):
        pm, npart, cloud, vis, data, labels = case_inputs(smpl, om, c)
        idx, _ = orc.reference_nn(pm[mj].astype(np.int32), cloud, vis, data, labels, npart)
        out[f"idx_{k}"] = idx.astype(np.int32)
        out[f"n_{k}"] = np.int64(len(labels))
        print(f"case {k}: {len(labels)} queries, {int((idx >= 0).sum())} matched, {len(np.unique(idx[idx >= 0]))} distinct model points")
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "nn_golden_full.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
