"""Stress: optimize() of resident frame batches (one and two frame groups, moment form, frames mapped to XCDs) repeated 150 times must give
bit-identical results every time.  Usage (on the GPU box): python tools/stress_batch.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avatar_amd import api, synth
from avatar_amd.capi import Options
smpl = synth.load_model(0); gm = api.AvatarModel(smpl); pm = synth.identity_part_map()
bad = 0
for F in (9, 30, 45, 64, 130):
    gts = [synth.sample_ground_truth(smpl, g % 16) for g in range(F)]
    st = [synth.perturb_start(*gts[i], i % 16) for i in range(F)]
    ctx = api.Context(gm, 24, pm, 65536, F)
    ctx.render_frames(np.array([g[0] for g in gts]), np.array([g[1] for g in gts]), np.array([g[2] for g in gts]))
    ctx.state_upload(np.array([s[1] for s in st]), api.rot_to_quat(np.array([s[2] for s in st]).reshape(-1, 3, 3)).reshape(F, 24, 4), np.array([s[0] for s in st]))
    opt = Options.demo(icp_iters=2)
    ctx.state_reset(); ctx.optimize_resident(opt); ref = ctx.state_download()
    for i in range(150):
        ctx.state_reset(); ctx.optimize_resident(opt)
        p, q, w, s_ = ctx.state_download()
        if not (np.array_equal(p, ref[0]) and np.array_equal(q, ref[1]) and np.array_equal(w, ref[2])):
            bad += 1
    print("F=%d: mismatches so far %d, shape %s, finite %s" % (F, bad, ctx.launch_shape(), np.isfinite(ref[0]).all()))
print("STRESS", "FAILED" if bad else "OK")
