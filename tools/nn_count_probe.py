"""What the slab scan of k_nn_part does on the synthetic frames (counting build: make -C avatar_amd/csrc libavatar_hip_nn_count.so;
AVT_LIB=avatar_amd/csrc/libavatar_hip_nn_count.so AVT_ONE_GROUP=1 python tools/nn_count_probe.py)."""
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
waves, tie_waves, ev, avail, rounds, width = tot
print(f"{int(waves)} waves of k_nn_part's slab scan over {F} frames: {ev / avail:.3f} of the candidates evaluated, {rounds / waves:.2f} rounds and "
      f"{avail / waves:.0f} candidates available per wave, mean slab width {100 * width / waves:.2f} cm, {int(tie_waves)} waves in the tie path")
