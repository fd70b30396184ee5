"""The batch split of include/avt_shard.h on hardware: a ONE-rank RCCL communicator on this GPU (the only size a 1-GPU
box can build: RCCL refuses two ranks on one device) pushes the model broadcast, the cloud scatter (grouped
ncclSend/ncclRecv, the root's own block looped through RCCL by avt_shard_set_self_exchange) and the result all-gather
through the real library on device buffers, and everything must come out bit-identical to the plain single-context path."""
import ctypes
import os

import numpy as np
import pytest

from avatar_amd import synth
from avatar_amd.capi import Options

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def one_rank_shard():
    from avatar_amd import capi, shard
    lib = capi.load_library()
    buf = ctypes.create_string_buffer(shard.ID_BYTES)
    assert lib.avt_shard_unique_id(buf) == 0, lib.avt_last_error()
    s = shard.Shard(0, 0, 1, buf.raw)
    yield s
    s.close()


def test_backend_is_rccl(one_rank_shard):
    assert one_rank_shard.backend.startswith("rccl ") and "librccl" in one_rank_shard.backend


def test_broadcast_model_builds_the_same_model(smpl, gmodel, one_rank_shard):
    from avatar_amd import api
    h = one_rank_shard.broadcast_model(gmodel.arrays, root=0)
    m2 = api.AvatarModel(smpl, handle=h)
    assert np.array_equal(m2.mainJoint, gmodel.mainJoint) and np.array_equal(m2.jointShapeReg, gmodel.jointShapeReg)
    w, p, R = synth.sample_ground_truth(smpl, 3)
    c1 = gmodel.default_ctx().lbs_update(w[None], p[None], R[None])[0]
    c2 = m2.default_ctx().lbs_update(w[None], p[None], R[None])[0]
    assert np.array_equal(c1, c2)


def test_scatter_optimize_gather_equals_plain_path(smpl, gmodel, one_rank_shard):
    from avatar_amd import api
    pm = synth.identity_part_map()
    B = 3
    frames = [synth.make_frame(smpl, 40 + f) for f in range(B)]
    datas = [fr["data"][::5 + f] for f, fr in enumerate(frames)]            # ragged point counts
    labels = [fr["labels"][::5 + f] for f, fr in enumerate(frames)]
    p0 = np.array([fr["start"][1] for fr in frames]); w0 = np.array([fr["start"][0] for fr in frames])
    q0 = np.array([api.rot_to_quat(fr["start"][2]) for fr in frames])
    opt = Options.demo(max_iters_per_icp=4)
    # plain path
    ctx_a = api.Context(gmodel, 24, pm, 16384, B, device=0)
    pa, qa, wa, sta = ctx_a.optimize_batch(datas, labels, opt, p0, q0, w0)
    # split path (one rank owns every frame), the root's block going through ncclSend/ncclRecv
    one_rank_shard.set_self_exchange(True)
    try:
        ctx_b = api.Context(gmodel, 24, pm, 16384, B, device=0)
        one_rank_shard.scatter_frames(ctx_b, B, datas, labels, p0, q0, w0, root=0)
    finally:
        one_rank_shard.set_self_exchange(False)
    ctx_b._N = np.array([len(l) for l in labels], np.int32)
    for f in range(B):
        d, l = ctx_b.frame_download(f)
        assert np.array_equal(d, datas[f]) and np.array_equal(l, labels[f])
    ctx_b.optimize_resident(opt)
    pg, qg, wg, stg = one_rank_shard.gather_results(ctx_b, B)
    pl, ql, wl, stl = ctx_b.state_download()
    assert np.array_equal(pg, pl) and np.array_equal(qg, ql) and np.array_equal(wg, wl)
    assert np.array_equal(pg, pa) and np.array_equal(qg, qa) and np.array_equal(wg, wa)       # bit-reproducible pipeline
    for f in range(B):
        assert stg[f].final_cost == sta[f].final_cost and stg[f].num_correspondences == sta[f].num_correspondences
        assert stg[f].gn_iterations == 4 and stg[f].matched_model_points == sta[f].matched_model_points
    # the exchange of one step runs on the shard's stream beside the next step: after reset -> optimize -> enqueue, three
    # times without any host synchronisation, the gathered block is the last step's (= the same results again)
    for _ in range(3):
        ctx_b.state_reset()
        ctx_b.optimize_resident(opt)
        one_rank_shard.gather_enqueue(ctx_b, B)
    one_rank_shard.gather_wait()
    pg2, qg2, wg2, stg2 = one_rank_shard.gather_download(ctx_b, B)
    assert np.array_equal(pg2, pa) and np.array_equal(qg2, qa) and np.array_equal(wg2, wa)
    assert all(stg2[f].final_cost == sta[f].final_cost for f in range(B))
    one_rank_shard.barrier(ctx_b)
    # the gathered block is a SNAPSHOT at every world size - enqueue, then a further optimize() that moves the states on (warm start:
    # no reset), then download: the rows are those of the call the gather was enqueued behind, not the later call's
    ctx_b.state_reset(); ctx_b.optimize_resident(opt)
    one_rank_shard.gather_enqueue(ctx_b, B)
    ctx_b.optimize_resident(opt)
    pg3, qg3, wg3, stg3 = one_rank_shard.gather_download(ctx_b, B)
    pl3, ql3, wl3, _ = ctx_b.state_download()
    assert np.array_equal(pg3, pa) and np.array_equal(qg3, qa) and np.array_equal(wg3, wa)
    assert not np.array_equal(pl3, pa)


def test_frames_swap_under_resident_state(smpl, gmodel):
    """avt_frames_upload followed by avt_optimize_resident must run with the NEW frames' point counts."""
    from avatar_amd import api
    pm = synth.identity_part_map()
    fa, fb = synth.make_frame(smpl, 50), synth.make_frame(smpl, 51)
    da, la = fa["data"][::6], fa["labels"][::6]
    db, lb = fb["data"][::9], fb["labels"][::9]           # a different, smaller point count
    opt = Options.demo(max_iters_per_icp=3)
    w0, p0, R0 = fa["start"]
    q0 = api.rot_to_quat(R0)
    ctx = api.Context(gmodel, 24, pm, 16384, 1, device=0)
    ctx.frames_upload([da], [la]); ctx.state_upload(p0[None], q0[None], w0[None])
    ctx.optimize_resident(opt)
    p1, q1, w1, _ = ctx.state_download()
    ctx.frames_upload([db], [lb])                          # warm start: keep the state, swap the frame
    ctx.optimize_resident(opt)
    p2, q2, w2, st2 = ctx.state_download()
    ref = api.Context(gmodel, 24, pm, 16384, 1, device=0)
    pr, qr, wr, str_ = ref.optimize_batch([db], [lb], opt, p1, q1, w1)
    assert np.array_equal(p2, pr) and np.array_equal(q2, qr) and np.array_equal(w2, wr)
    assert st2[0].num_correspondences == str_[0].num_correspondences == int((ctx.correspondences(0, len(lb)) >= 0).sum())


def test_stand_alone_calls_invalidate_resident_state(smpl, gmodel, frame0):
    from avatar_amd import api
    pm = synth.identity_part_map()
    d, l = frame0["data"][::8], frame0["labels"][::8]
    w0, p0, R0 = frame0["start"]
    ctx = api.Context(gmodel, 24, pm, 16384, 1, device=0)
    ctx.frames_upload([d], [l]); ctx.state_upload(p0[None], api.rot_to_quat(R0)[None], w0[None])
    ctx.lbs_update(w0[None], p0[None], R0[None])           # uses the frame slots as scratch
    with pytest.raises(api.AvtError, match="no frames resident"):
        ctx.optimize_resident(Options.demo())


def test_failed_context_creation_releases_everything(gmodel):
    """avt_ctx_create error paths clean up: many failing creations must not exhaust streams or memory."""
    from avatar_amd import api
    bad_map = np.full(24, 99, np.int32)                    # part ids outside [0, num_parts)
    for _ in range(200):
        with pytest.raises(api.AvtError):
            api.Context(gmodel, 24, bad_map, 1 << 20, 64, device=0)


# ---- W > 1 ranks on one GPU: the loop-back transport (threads of this process) runs every line of the multi-peer exchange ----

def _run_ranks(W, fn):
    """fn(rank) on W threads; returns the results, re-raising the first exception after all threads ended."""
    import threading
    out, err = [None] * W, [None] * W

    def body(r):
        try:
            out[r] = fn(r)
        except BaseException as e:      # noqa: BLE001 - reported below
            err[r] = e
    th = [threading.Thread(target=body, args=(r,)) for r in range(W)]
    for t in th: t.start()
    for t in th: t.join(120)
    assert not any(t.is_alive() for t in th), "a rank hangs in an exchange"
    return out, err


@pytest.mark.parametrize("W,B", [(2, 5), (3, 7), (8, 5), (8, 17)])
def test_loopback_ranks_scatter_optimize_gather(smpl, gmodel, W, B):
    """W ranks as W threads (one context + one loop-back shard each, all on this GPU): model broadcast, cloud scatter with
    W - 1 peers (ragged batches; with W = 8, B = 5 three ranks own NO frame), optimize, all-gather.  Every rank ends up with
    every frame's result in global frame order, bit-identical to the single-context run; every rank's resident block is
    frames_of_rank(B, rank, W)."""
    import uuid
    from avatar_amd import api, shard
    pm = synth.identity_part_map()
    frames = [synth.make_frame(smpl, 60 + f) for f in range(B)]
    datas = [fr["data"][::7 + (f % 3)] for f, fr in enumerate(frames)]
    labels = [fr["labels"][::7 + (f % 3)] for f, fr in enumerate(frames)]
    p0 = np.array([fr["start"][1] for fr in frames]); w0 = np.array([fr["start"][0] for fr in frames])
    q0 = np.array([api.rot_to_quat(fr["start"][2]) for fr in frames])
    opt = Options.demo(max_iters_per_icp=3)
    per = (B + W - 1) // W
    # the plain path, rank share by rank share: results are bit-reproducible for a given launch SHAPE (the number of frames in a
    # call fixes how many workgroups share a frame's sums, hence their rounding), so the yard-stick fits each rank's share with
    # one ordinary context holding exactly those frames
    pa = np.empty((B, 3)); qa = np.empty((B, gmodel.numJoints(), 4)); wa = np.empty((B, gmodel.numShapeKeys())); sta = [None] * B
    for r in range(W):
        mine = shard.frames_of_rank(B, r, W)
        if not mine: continue
        pr, qr, wr, sr = api.Context(gmodel, 24, pm, 8192, per, device=0).optimize_batch(
            [datas[f] for f in mine], [labels[f] for f in mine], opt, p0[mine], q0[mine], w0[mine])
        pa[mine] = pr; qa[mine] = qr.reshape(len(mine), -1, 4); wa[mine] = wr
        for i, f in enumerate(mine): sta[f] = sr[i]
    group = "t-" + uuid.uuid4().hex

    def rank_main(r):
        sh = shard.Shard(0, r, W, loopback_group=group)
        assert "loop-back" in sh.backend and sh.world == W
        h = sh.broadcast_model(gmodel.arrays if r == 0 else None, root=0)
        model = api.AvatarModel(smpl, handle=h)
        ctx = api.Context(model, 24, pm, 8192, per, device=0)
        if r == 0: sh.scatter_frames(ctx, B, datas, labels, p0, q0, w0, root=0)
        else: sh.scatter_frames(ctx, B, root=0)
        mine = shard.frames_of_rank(B, r, W)
        ctx._N = np.array([len(labels[f]) for f in mine], np.int32)
        for i, f in enumerate(mine):                       # this rank's resident block is exactly its share, in order
            d, l = ctx.frame_download(i)
            assert np.array_equal(d, datas[f]) and np.array_equal(l, labels[f])
        if mine:
            ctx.optimize_resident(opt)
        res = sh.gather_results(ctx, B)
        sh.barrier(ctx)
        sh.close()
        return res

    out, err = _run_ranks(W, rank_main)
    for e in err:
        if e is not None: raise e
    for r in range(W):
        pg, qg, wg, stg = out[r]
        assert np.array_equal(pg, pa) and np.array_equal(qg, qa) and np.array_equal(wg, wa), f"rank {r} of {W}"
        assert all(stg[f].final_cost == sta[f].final_cost and stg[f].num_correspondences == sta[f].num_correspondences for f in range(B))


def test_loopback_a_bad_rank_fails_everybody_and_nobody_hangs(smpl, gmodel):
    """one rank's context is too small for its share.  The scatter fails on EVERY rank (agreed before any cloud
    moves), no rank is left inside an exchange; a gather with one rank not holding its share marks that rank's rows faulty:
    the offender reports at once, its peers at download."""
    import uuid
    from avatar_amd import api, shard
    pm = synth.identity_part_map()
    W, B = 3, 6
    frames = [synth.make_frame(smpl, 70 + f) for f in range(B)]
    datas = [fr["data"][::9] for fr in frames]; labels = [fr["labels"][::9] for fr in frames]
    p0 = np.array([fr["start"][1] for fr in frames]); w0 = np.array([fr["start"][0] for fr in frames])
    q0 = np.array([api.rot_to_quat(fr["start"][2]) for fr in frames])
    group = "t-" + uuid.uuid4().hex

    def rank_main(r):
        sh = shard.Shard(0, r, W, loopback_group=group)
        small = api.Context(gmodel, 24, pm, 8192, 1 if r == 1 else 2, device=0)      # rank 1 cannot hold its two frames
        with pytest.raises(api.AvtError, match="rejected the batch|exceeds the context"):
            sh.scatter_frames(small, B, *( (datas, labels, p0, q0, w0) if r == 0 else () ), root=0)
        # same communicator, now a sound batch; rank 2 then "forgets" its frames before the gather
        ctx = api.Context(gmodel, 24, pm, 8192, 2, device=0)
        sh.scatter_frames(ctx, B, *((datas, labels, p0, q0, w0) if r == 0 else ()), root=0)
        ctx.optimize_resident(Options.demo(max_iters_per_icp=2))
        if r == 2:
            w, p, R = synth.sample_ground_truth(smpl, 1)
            ctx.lbs_update(w[None], p[None], R[None])                                  # invalidates the resident frames
        with pytest.raises(api.AvtError, match="differ from this rank|device fault"):
            sh.gather_results(ctx, B)
        sh.close()
        return True

    out, err = _run_ranks(W, rank_main)
    for e in err:
        if e is not None: raise e
    assert all(out)
