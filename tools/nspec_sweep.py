"""ms per optimize() of single frames (seeds 0..N-1) for the current AVT_NSPEC: the rejection runs differ from frame to frame."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avatar_amd import api, synth
from avatar_amd.capi import Options
smpl = synth.load_model(0); gm = api.AvatarModel(smpl); pm = synth.identity_part_map()
ms = []
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    fr = synth.make_frame(smpl, seed)
    ctx = api.Context(gm, 24, pm, 60000, 1)
    ctx.frames_upload([fr["data"]], [fr["labels"]]); w0, p0, R0 = fr["start"]
    ctx.state_upload(p0[None], api.rot_to_quat(R0)[None], w0[None])
    opt = Options.demo()
    for i in range(5): ctx.state_reset(); ctx.optimize_resident(opt)
    ctx.sync(); t0 = time.perf_counter()
    for i in range(200): ctx.state_reset(); ctx.optimize_resident(opt)
    ctx.sync(); ms.append((time.perf_counter() - t0) / 200 * 1e3)
print("AVT_NSPEC=%s: ms per frame %s | mean %.4f" % (os.environ.get("AVT_NSPEC", "default"), " ".join("%.3f" % m for m in ms), np.mean(ms)))
