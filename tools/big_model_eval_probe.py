"""Per-wave phase clocks of the generic evaluation kernel (k_eval<0,0,12>) on the 52- / 55-joint synthetic models.
Needs: make -C avatar_amd/csrc libavatar_hip_timing.so;  AVT_LIB=avatar_amd/csrc/libavatar_hip_timing.so python tools/big_model_eval_probe.py [frames]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from avatar_amd import api, capi, synth
from avatar_amd.capi import Options
from oracle import oracle as orc            # only to pose the synthetic frame (tests/bigmodel.make_frame)
from bigmodel import extend_model, make_frame
F = int(sys.argv[1]) if len(sys.argv) > 1 else 1
smpl = synth.load_model(0)
for joints in (52, 55):
    m = extend_model(smpl, joints); gm = api.AvatarModel(m); om = orc.OracleModel(m)
    fr = make_frame(m, om, smpl, 3)
    ctx = api.Context(gm, joints, fr["part_map"], len(fr["labels"]), F, device=0)
    w0, p0, R0 = fr["start"]
    for i in range(2):
        ctx.optimize_batch([fr["data"]] * F, [fr["labels"]] * F, Options.demo(), np.repeat(p0[None], F, 0), np.repeat(api.rot_to_quat(R0)[None], F, 0), np.repeat(w0[None], F, 0))
    lib = capi.load_library(); buf = np.zeros(64)
    lib.avt_debug_trace(ctx.h, 0, buf.ctypes.data_as(C.POINTER(C.c_double)))
    names = ['A-wait', 'mfma', 'records-wait', 'build', 'B-wait']
    print("J=%d F=%d launch shape %s; workgroup 0 of frame 0, last full evaluation; wall %.2f us" % (joints, F, ctx.launch_shape(), buf[56] / 100.0))
    for wv in range(4):
        t = buf[16 + 8 * wv:24 + 8 * wv]
        print("  wave %d: total %8.0f | " % (wv, t.sum()) + " | ".join("%s %7.0f" % (n, v) for n, v in zip(names, t)))
