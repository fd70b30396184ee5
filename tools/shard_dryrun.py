#!/usr/bin/env python3
"""Dry run of the RCCL batch split with W ranks that all share GPU 0 (a 1-GPU box has nothing else):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29555 tools/shard_dryrun.py
The rendezvous runs over gloo.  RCCL is expected to REFUSE the communicator ("Duplicate GPU detected"): the log records
that, and that the same code path with one rank per GPU (world 1 on this box: tests/test_gpu_shard.py) works."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist  # noqa: E402

from avatar_amd import shard  # noqa: E402


def main():
    dist.init_process_group(backend="gloo", init_method="env://")
    rank, world = dist.get_rank(), dist.get_world_size()
    uid = shard.exchange_unique_id(dist, rank)
    try:
        s = shard.Shard(0, rank, world, uid)
        print(f"[rank {rank}] communicator of {world} ranks on ONE GPU built: {s.backend}", flush=True)
        s.barrier()
        print(f"[rank {rank}] barrier ok", flush=True)
        s.close()
    except Exception as e:
        print(f"[rank {rank}] avt_shard_create failed as expected on a shared GPU: {e}", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
