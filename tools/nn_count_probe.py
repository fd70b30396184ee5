import sys
import numpy as np
sys.path.insert(0, '/root/repo')
from avatar_amd import api, synth, capi
from avatar_amd.capi import Options
import ctypes as C
smpl = synth.load_model(0); gm = api.AvatarModel(smpl); pm = synth.identity_part_map()
F = 32
gts = [synth.sample_ground_truth(smpl, g) for g in range(F)]
st = [synth.perturb_start(*gts[i], i) for i in range(F)]
ctx = api.Context(gm, 24, pm, 65536, F, device=0)
ctx.render_frames(np.array([g[0] for g in gts]), np.array([g[1] for g in gts]), np.array([g[2] for g in gts]))
ctx.state_upload(np.array([s[1] for s in st]), api.rot_to_quat(np.array([s[2] for s in st]).reshape(-1, 3, 3)).reshape(F, 24, 4), np.array([s[0] for s in st]))
ctx.optimize_resident(Options.demo(max_iters_per_icp=1)); ctx.sync()
lib = capi.load_library()
tot = np.zeros(6)
for f in range(F):
    out = np.zeros(64)
    lib.avt_debug_trace(ctx.h, C.c_int(f), out.ctypes.data_as(C.POINTER(C.c_double)))
    tot += out[58:64]
print("waves", tot[0], "waves in the tie path", tot[1]); tot = tot[2:]; print("evaluated fraction", tot[0] / tot[1], "rounds per wave", tot[2] / (tot[1] / (tot[1]/tot[2]) ) if False else "", "mean candidates per wave", tot[1], tot[0], "rounds", tot[2], "sum slab width", tot[3])
