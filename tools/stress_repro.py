"""Stress: optimize() of the same frames repeated many times must give bit-identical results every time (the few-frames launch
shapes hand the reduced system over inside a launch: a lost or early hand-over would show up here).
Usage: python tools/stress_repro.py [repeats]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avatar_amd import api, synth
from avatar_amd.capi import Options

R = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
smpl = synth.load_model(0); gm = api.AvatarModel(smpl); pm = synth.identity_part_map()
bad = 0
for F in (1, 2, 3, 4):
    frs = [synth.make_frame(smpl, 5 + s) for s in range(F)]
    ctx = api.Context(gm, 24, pm, 60000, F)
    ctx.frames_upload([f["data"] for f in frs], [f["labels"] for f in frs])
    ctx.state_upload(np.array([f["start"][1] for f in frs]), np.array([api.rot_to_quat(f["start"][2]) for f in frs]), np.array([f["start"][0] for f in frs]))
    opt = Options.demo()
    ctx.state_reset(); ctx.optimize_resident(opt); ref = ctx.state_download()
    for i in range(R):
        ctx.state_reset(); ctx.optimize_resident(opt)
        if True:
            p, q, w, st = ctx.state_download()
            if not (np.array_equal(p, ref[0]) and np.array_equal(q, ref[1]) and np.array_equal(w, ref[2])):
                bad += 1
    print("F=%d: %d repeats, mismatching downloads so far: %d (launch shape %s)" % (F, R, bad, ctx.launch_shape()))
print("STRESS", "FAILED" if bad else "OK")
