"""Batch split of independent frames over ranks: frame f -> rank f mod W.  No collective on the data
path; the only exchanges are the result gather (109 doubles per frame) and the timing all-reduce."""
import numpy as np


def frames_of_rank(num_frames, rank, world):
    return list(range(rank, num_frames, world))


def gather_results(local, num_frames, rank, world, dist):
    """local: (n_local, D) results of this rank's frames (in frames_of_rank order) -> (num_frames, D) on every rank."""
    import torch
    D = local.shape[1] if local.size else 0
    dmax = torch.tensor([D], dtype=torch.int64)
    dist.all_reduce(dmax, op=dist.ReduceOp.MAX)
    D = int(dmax.item())
    per = (num_frames + world - 1) // world
    buf = torch.zeros(per, D, dtype=torch.float64)
    if local.size:
        buf[:local.shape[0]] = torch.from_numpy(np.ascontiguousarray(local))
    out = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    res = np.zeros((num_frames, D))
    for r in range(world):
        fr = frames_of_rank(num_frames, r, world)
        res[fr] = out[r][:len(fr)].numpy()
    return res
