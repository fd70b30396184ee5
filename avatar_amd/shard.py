"""Batch split of independent frames over the GPUs of one node, Python view of include/avt_shard.h.

frame f -> rank f mod W; model constants replicated; no collective inside optimize().  The three exchanges (model
broadcast, cloud scatter, result all-gather) run inside libavatar_hip.so on device buffers over RCCL
(avatar_amd/csrc/avt_shard.cpp); this module only marshals arguments.  The rendezvous (128 opaque bytes from rank 0 to
everybody) travels over whatever the host program has - `exchange_unique_id` uses a torch.distributed process group.

`gather_results` is the torch.distributed equivalent of the result gather (device tensors under nccl, host tensors
under gloo): the CPU tests use it, and bench.py falls back to it - loudly - if the RCCL communicator cannot be built.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi
from .capi import Stats, dptr, iptr

ID_BYTES = 128


def frames_of_rank(num_frames, rank, world):
    return list(range(rank, num_frames, world))


def _check(rc):
    if rc != 0:
        from .api import AvtError
        e = AvtError(capi.load_library().avt_last_error().decode())
        e.status = rc
        raise e


def pack_model(arrays: capi.ModelArrays) -> bytes:
    """avt_model_pack: the relocatable byte block the model broadcast ships."""
    lib = capi.load_library()
    desc = arrays.desc()
    n = C.c_size_t()
    _check(lib.avt_model_pack_size(C.byref(desc), C.byref(n)))
    buf = C.create_string_buffer(n.value)
    _check(lib.avt_model_pack(C.byref(desc), buf, n))
    return buf.raw


def unpack_model(block: bytes):
    """avt_model_unpack -> avt_model* (c_void_p)."""
    lib = capi.load_library()
    h = C.c_void_p()
    buf = C.create_string_buffer(block, len(block))
    _check(lib.avt_model_unpack(buf, C.c_size_t(len(block)), C.byref(h)))
    return h


def exchange_unique_id(dist, rank, src=0, rccl=True):
    """Rank `src` creates the rendezvous bytes - the RCCL unique id, or (rccl=False: the shared-memory transport, which only
    needs bytes all ranks agree on) 128 random ones; everybody receives them through the torch.distributed store."""
    lib = capi.load_library()
    box = [None]
    if rank == src:
        if rccl:
            buf = C.create_string_buffer(ID_BYTES)
            _check(lib.avt_shard_unique_id(buf))
            box[0] = buf.raw
        else:
            import os
            box[0] = os.urandom(ID_BYTES)
    dist.broadcast_object_list(box, src=src)
    return box[0]


class Shard:
    """avt_shard: one RCCL communicator rank bound to one GPU."""

    def __init__(self, device, rank, world, unique_id: bytes = None, loopback_group: str = None, shm: bool = False):
        """unique_id: the rendezvous bytes.  loopback_group: instead of RCCL, the in-process loop-back transport - the
        ranks are threads of this process that name the same group (avt_shard_create_loopback).  shm: the ranks are processes
        of one node that exchange through a shared-memory segment named after unique_id (avt_shard_create_shm)."""
        self._lib = capi.load_library()
        self.h = C.c_void_p()
        if loopback_group is not None:
            _check(self._lib.avt_shard_create_loopback(C.c_int(device), C.c_int(rank), C.c_int(world), loopback_group.encode(), C.byref(self.h)))
        elif shm:
            _check(self._lib.avt_shard_create_shm(C.c_int(device), C.c_int(rank), C.c_int(world), unique_id, C.byref(self.h)))
        else:
            _check(self._lib.avt_shard_create(C.c_int(device), C.c_int(rank), C.c_int(world), unique_id, C.byref(self.h)))
        self.rank, self.world = rank, world
        self.backend = self._lib.avt_shard_backend(self.h).decode()

    def close(self):
        if self.h:
            self._lib.avt_shard_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def local_frames(self, num_frames):
        return frames_of_rank(num_frames, self.rank, self.world)

    def broadcast_model(self, arrays, root=0):
        """arrays: capi.ModelArrays on root (ignored elsewhere). Returns an avt_model* built from the broadcast bytes."""
        h = C.c_void_p()
        desc = arrays.desc() if (self.rank == root and arrays is not None) else None
        _check(self._lib.avt_shard_broadcast_model(self.h, C.c_int(root), C.byref(desc) if desc is not None else None, C.byref(h)))
        return h

    def scatter_frames(self, ctx, num_frames, datas=None, labels=None, p=None, q=None, w=None, root=0):
        """Root passes the whole batch (lists of per-frame arrays + (B,3)/(B,J,4)/(B,K) start states)."""
        if self.rank == root:
            offs = np.zeros(num_frames + 1, np.int32)
            for f in range(num_frames):
                offs[f + 1] = offs[f] + len(labels[f])
            data = np.ascontiguousarray(np.concatenate([np.asarray(d, np.float64).reshape(-1, 3) for d in datas], 0))
            lab = np.ascontiguousarray(np.concatenate([np.asarray(l, np.int32) for l in labels]))
            p = np.ascontiguousarray(np.asarray(p, np.float64).reshape(num_frames, 3))
            q = np.ascontiguousarray(np.asarray(q, np.float64).reshape(num_frames, -1))
            w = np.ascontiguousarray(np.asarray(w, np.float64).reshape(num_frames, -1))
            args = (dptr(data), iptr(lab), iptr(offs), dptr(p), dptr(q), dptr(w))
        else:
            args = (None,) * 6
        _check(self._lib.avt_shard_scatter_frames(self.h, ctx.h, C.c_int(root), C.c_int(num_frames), *args))
        nloc = len(self.local_frames(num_frames))
        ctx._F = nloc

    def gather_enqueue(self, ctx, num_frames):
        _check(self._lib.avt_shard_gather_enqueue(self.h, ctx.h, C.c_int(num_frames)))

    def gather_wait(self):
        """Blocks until the last enqueued all-gather is complete."""
        _check(self._lib.avt_shard_gather_wait(self.h))

    def gather_download(self, ctx, num_frames):
        m = ctx.model
        p = np.empty((num_frames, 3)); q = np.empty((num_frames, m.numJoints() * 4)); w = np.empty((num_frames, m.numShapeKeys()))
        st = (Stats * num_frames)()
        _check(self._lib.avt_shard_gather_download(self.h, ctx.h, C.c_int(num_frames), dptr(p), dptr(q), dptr(w), st))
        return p, q.reshape(num_frames, -1, 4), w, list(st)

    def gather_results(self, ctx, num_frames):
        self.gather_enqueue(ctx, num_frames)
        return self.gather_download(ctx, num_frames)

    def set_self_exchange(self, on=True):
        """Dry runs: a rank's own blocks go through the transport as well (include/avt_shard.h, avt_shard_set_self_exchange)."""
        _check(self._lib.avt_shard_set_self_exchange(self.h, C.c_int(int(bool(on)))))

    def barrier(self, ctx=None):
        _check(self._lib.avt_shard_barrier(self.h, ctx.h if ctx is not None else None))


def gather_results(local, num_frames, rank, world, dist, device=None):
    """torch.distributed result gather: local (n_local, D) results of this rank's frames (frames_of_rank order) ->
    (num_frames, D) on every rank.  `device`: a torch device for the exchange buffers (cuda:<i> under nccl; None = host,
    for gloo)."""
    import torch
    D = local.shape[1] if local.size else 0
    dmax = torch.tensor([D], dtype=torch.int64, device=device)
    dist.all_reduce(dmax, op=dist.ReduceOp.MAX)
    D = int(dmax.item())
    per = (num_frames + world - 1) // world
    buf = torch.zeros(per, D, dtype=torch.float64, device=device)
    if local.size:
        buf[:local.shape[0]] = torch.from_numpy(np.ascontiguousarray(local)).to(buf.device)
    out = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    res = np.zeros((num_frames, D))
    for r in range(world):
        fr = frames_of_rank(num_frames, r, world)
        res[fr] = out[r][:len(fr)].cpu().numpy()
    return res
