#!/usr/bin/env python3
"""Is the host ahead of the GPU?  K optimize() steps are enqueued back to back (no synchronisation): time until the last enqueue
returns against time until the GPU is done, for F frames per GPU.  Usage: enqueue_vs_total.py [F ...]"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from avatar_amd import api, synth  # noqa: E402
from avatar_amd.capi import Options  # noqa: E402

smpl = synth.load_model(0); gm = api.AvatarModel(smpl); pm = synth.identity_part_map()
for F in [int(a) for a in sys.argv[1:]] or [1, 64]:
    gts = [synth.sample_ground_truth(smpl, g) for g in range(F)]
    st = [synth.perturb_start(*gts[i], i) for i in range(F)]
    ctx = api.Context(gm, 24, pm, 65536, F, device=0)
    ctx.render_frames(np.array([g[0] for g in gts]), np.array([g[1] for g in gts]), np.array([g[2] for g in gts]))
    ctx.state_upload(np.array([s[1] for s in st]), api.rot_to_quat(np.array([s[2] for s in st]).reshape(-1, 3, 3)).reshape(F, 24, 4), np.array([s[0] for s in st]))
    opt = Options.demo()
    for _ in range(3):
        ctx.state_reset(); ctx.optimize_resident(opt)
    ctx.sync()
    K = 20
    t0 = time.perf_counter()
    for _ in range(K):
        ctx.state_reset(); ctx.optimize_resident(opt)
    t1 = time.perf_counter()
    ctx.sync()
    t2 = time.perf_counter()
    print(f"{F} frames: enqueue of {K} steps returned after {(t1 - t0) / K * 1e3:.3f} ms per step, GPU done after {(t2 - t0) / K * 1e3:.3f} ms per step")
