import sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
from avatar_amd import api, synth
from avatar_amd.capi import Options
smpl = synth.load_model(0); gm = api.AvatarModel(smpl); pm = synth.identity_part_map()
gt = synth.sample_ground_truth(smpl, 0); st = synth.perturb_start(*gt, 0)
ctx = api.Context(gm, 24, pm, 65536, 1, device=0)
ctx.render_frames(gt[0][None], gt[1][None], gt[2][None])
ctx.state_upload(st[1][None], api.rot_to_quat(st[2].reshape(-1,3,3)).reshape(1,24,4), st[0][None])
opt = Options.demo()
def run(nreset, K=200):
    for _ in range(5):
        for _ in range(nreset): ctx.state_reset()
        ctx.optimize_resident(opt)
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(K):
        for _ in range(nreset): ctx.state_reset()
        ctx.optimize_resident(opt)
    ctx.sync(); return (time.perf_counter() - t0) / K * 1e3
for rep in range(2):
    print("resets per step 1:", round(run(1), 4), "ms; 2:", round(run(2), 4), "ms; 3:", round(run(3), 4), "ms")
