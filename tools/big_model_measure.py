"""Step time and per-class kernel times of the large-skeleton launch shapes (52 joints = SMPL-H sized, 55 = SMPL-X sized),
synthetic models grown from the SMPL-shaped one by tests/bigmodel.py.  Run on the GPU box:
gpurun -- 'python tools/big_model_measure.py [frames]'"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from avatar_amd import api, synth
from avatar_amd.capi import Options
from oracle import oracle as orc            # only to pose the synthetic frame (tests/bigmodel.make_frame)
from bigmodel import extend_model, make_frame

F = int(sys.argv[1]) if len(sys.argv) > 1 else 1
smpl = synth.load_model(0)
for joints in (52, 55):
    m = extend_model(smpl, joints); gm = api.AvatarModel(m); om = orc.OracleModel(m)
    frs = [make_frame(m, om, smpl, 3 + s % 4) for s in range(min(F, 4))]
    frs = [frs[s % len(frs)] for s in range(F)]
    pm = frs[0]["part_map"]
    ctx = api.Context(gm, joints, pm, max(len(f["labels"]) for f in frs), F, device=0)
    opt = Options.demo()
    ctx.frames_upload([f["data"] for f in frs], [f["labels"] for f in frs])
    ctx.state_upload(np.array([f["start"][1] for f in frs]), np.array([api.rot_to_quat(f["start"][2]) for f in frs]),
                     np.array([f["start"][0] for f in frs]))
    for i in range(5):
        ctx.state_reset(); ctx.optimize_resident(opt)
    ctx.sync()
    ctx.profile_begin(); ctx.state_reset(); ctx.optimize_resident(opt); ctx.sync(); prof = ctx.profile_end()
    ctx.sync(); t0 = time.perf_counter()
    for i in range(50):
        ctx.state_reset(); ctx.optimize_resident(opt)
    ctx.sync(); dt = (time.perf_counter() - t0) / 50
    print("J=%d P=%d F=%d: %.3f ms per optimize(), %.0f GN it/s | %s" % (
        joints, gm.arrays.P, F, dt * 1e3, F * opt.icp_iters * opt.max_iters_per_icp / dt,
        " ".join("%s %.4f/%d" % (k, v[0], v[1]) for k, v in prof.items() if v[1])))
