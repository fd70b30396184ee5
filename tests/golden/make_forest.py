"""Generates tests/golden/forest_small.srtr (+ .partmap): the toy body-part tree the RTree tests and bench.py run
(avatar_amd/synth_forest.py, default configuration, seed 0), written in the reference's binary format by the oracle's
exporter.  Run from the repo root: python tests/golden/make_forest.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from avatar_amd import synth, synth_forest  # noqa: E402
from oracle import rtree_oracle  # noqa: E402

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "forest_small.srtr")
model = synth.load_model(0)
f, l, d, num_parts = synth_forest.train(model)
tree = rtree_oracle.OracleRTree.from_arrays(f, l, d, num_parts)
assert tree.export(out)
synth_forest.write_part_map(out + ".partmap", synth.identity_part_map(), contiguous=True)
print(out, os.path.getsize(out), "bytes,", len(l), "nodes,", len(d), "leaves")
