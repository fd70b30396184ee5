"""Generates tests/golden/nn_golden.npz with the reference's OWN nanoflann KD-tree search.

Run in the build container only (needs /root/reference/include/nanoflann.hpp, compiled from where it lies by
oracle/Makefile into oracle/_ref/).  Inputs: seeded synthetic (model cloud, visibility, data cloud, labels) sets
that cover the per-part restriction, parts without visible model points, a coarse part map and a ragged tiny
case.  Outputs: the model index nanoflann returns for every data point (and the squared distance, used to assert
that the fixtures contain no exact ties, the only case where KD traversal order could matter).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from avatar_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def main():
    assert orc.have_reference_nn(), "build oracle/_ref first (make -C oracle ref)"
    m = synth.load_model(0)
    om = orc.OracleModel(m)
    mj = om.main_joint()
    cases = []
    rng = np.random.default_rng(42)

    def add(seed, part_map, num_parts, sub=None, drop_part=None):
        fr = synth.make_frame(m, seed, part_map=part_map)
        w0, p0, R0 = fr["start"]
        cloud, _, _ = om.update(w0, p0, R0)
        vis = om.visibility(cloud, True)
        if drop_part is not None:
            vis = vis.copy(); vis[part_map[mj] == drop_part] = 0
        data, labels = fr["data"], fr["labels"]
        if sub is not None:
            sel = np.sort(rng.choice(len(labels), size=sub, replace=False))
            data, labels = data[sel], labels[sel]
        model_part = part_map[mj].astype(np.int32)
        idx, dist = orc.reference_nn(model_part, cloud, vis, data, labels, num_parts)
        # no exact ties: the winner's distance must be strictly below every other candidate's
        ref2 = om.nn(part_map, num_parts, cloud, vis, data, labels)
        assert np.array_equal(idx, ref2), "ordered brute force disagrees with nanoflann"
        cases.append(dict(part_map=part_map.astype(np.int32), num_parts=num_parts, cloud=cloud, vis=vis, data=data,
                          labels=labels.astype(np.int32), idx=idx, dist=dist))

    ident = synth.identity_part_map()
    add(0, ident, 24, sub=4000)
    add(1, ident, 24, sub=3000, drop_part=4)                 # a part with data but no visible model point
    coarse = np.array([0, 1, 2, 0, 1, 2, 0, 1, 2, 0, 1, 2, 3, 3, 3, 3, 4, 5, 4, 5, 4, 5, 4, 5], np.int32)
    add(2, coarse, 6, sub=3000)                              # 6 coarse parts (bigger candidate sets)
    add(3, ident, 24, sub=5)                                 # ragged tiny
    out = {"ncase": len(cases)}
    for k, c in enumerate(cases):
        for key, v in c.items():
            out[f"{key}_{k}"] = v
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "nn_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", [len(c["labels"]) for c in cases], "queries")


if __name__ == "__main__":
    main()
