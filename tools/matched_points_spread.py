import numpy as np, sys
sys.path.insert(0,'/root/repo')
from avatar_amd import api, synth
from avatar_amd.capi import Options
smpl=synth.load_model(0); gm=api.AvatarModel(smpl); pm=synth.identity_part_map()
F=64
gts=[synth.sample_ground_truth(smpl,g) for g in range(F)]
starts=[synth.perturb_start(*gts[i],i) for i in range(F)]
ctx=api.Context(gm,24,pm,65536,F,device=0)
n=ctx.render_frames(np.array([g[0] for g in gts]),np.array([g[1] for g in gts]),np.array([g[2] for g in gts]))
J=24
ctx.state_upload(np.array([s[1] for s in starts]), api.rot_to_quat(np.array([s[2] for s in starts]).reshape(-1,3,3)).reshape(F,J,4), np.array([s[0] for s in starts]))
ctx.optimize_resident(Options.demo())
st=ctx.state_download()[3]
M=np.array([s.matched_model_points for s in st]); 
print("N", n.min(), n.mean(), n.max()); print("M", M.min(), M.mean(), M.max(), "batches", np.ceil(M/16).min(), np.ceil(M/16).max())
nb=np.ceil(M/16)
print("batches group 0 (frames 0-31):", nb[:32].sum(), "max", nb[:32].max(), " group 1 (frames 32-63):", nb[32:].sum(), "max", nb[32:].max())
print("points group 0:", n[:32].sum(), " group 1:", n[32:].sum())
