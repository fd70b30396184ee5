"""world_size-2 gloo check of the multi-GPU batch split (bench.py's N>1 path): frames are independent, rank r owns
frames r, r+W, ...; no data-path collective; results gathered; timing is a MAX all-reduce.  CPU only: each rank runs
its shard through the oracle and the gathered result must equal the single-process result."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from avatar_amd import synth  # noqa: E402
from avatar_amd.capi import Options  # noqa: E402
from avatar_amd.shard import frames_of_rank, gather_results  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def main():
    dist.init_process_group(backend="gloo", init_method="env://")
    rank, world = dist.get_rank(), dist.get_world_size()
    smpl = synth.load_model(0)
    om = orc.OracleModel(smpl)
    B = 5
    opt = Options.demo(max_iters_per_icp=2)
    pm = synth.identity_part_map()

    def run(f):
        fr = synth.make_frame(smpl, f)
        w0, p0, R0 = fr["start"]
        r = om.optimize(pm, 24, fr["data"][::20], fr["labels"][::20], opt, p0, orc.rot_to_quat(R0), w0, aggregate=1)
        return np.concatenate([r["p"], r["q"].reshape(-1), r["w"]])

    mine = frames_of_rank(B, rank, world)
    local = np.array([run(f) for f in mine]).reshape(len(mine), 109)
    allres = gather_results(local, B, rank, world, dist)
    t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert abs(t.item() - 0.1 * world) < 1e-12
    if rank == 0:
        ref = np.array([run(f) for f in range(B)])
        assert allres.shape == (B, 109) and np.array_equal(allres, ref)
        assert sorted(sum([frames_of_rank(B, r, world) for r in range(world)], [])) == list(range(B))
        print("SHARD_OK")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
