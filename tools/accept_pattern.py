"""Accept / reject pattern of the LM iterations of the bench frames (cost trace of avt_debug_trace): which GN iterations were
rejected, i.e. followed by a re-solve of the SAME system with a larger lambda."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avatar_amd import api, capi, synth
from avatar_amd.capi import Options
smpl = synth.load_model(0); gm = api.AvatarModel(smpl); pm = synth.identity_part_map()
lib = capi.load_library()
tot = [0, 0]
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    fr = synth.make_frame(smpl, seed)
    ctx = api.Context(gm, 24, pm, 60000, 1)
    w0, p0, R0 = fr["start"]
    p, q, w, st = ctx.optimize_batch([fr["data"]], [fr["labels"]], Options.demo(), p0[None], api.rot_to_quat(R0)[None], w0[None])
    buf = np.zeros(64); lib.avt_debug_trace(ctx.h, 0, buf.ctypes.data_as(C.POINTER(C.c_double)))
    c = buf[0:11]
    pat = "".join("A" if c[i + 1] < c[i] else "R" for i in range(10))
    tot[0] += pat.count("A"); tot[1] += pat.count("R")
    print("frame seed %2d: %s  accepted %d  cost %.3f -> %.3f" % (seed, pat, st[0].accepted_steps, c[0], c[10]))
print("accepted %d, rejected %d" % tuple(tot))
