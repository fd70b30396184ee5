"""In-kernel phase timing (clock64, thread 0) of the last k_solve<1024,true> launch on the 52- / 55-joint synthetic models.
Needs the instrumented library: make -C avatar_amd/csrc libavatar_hip_timing_lm.so, then
    AVT_LIB=avatar_amd/csrc/libavatar_hip_timing_lm.so python tools/big_model_phase_probe.py"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from avatar_amd import api, capi, synth
from avatar_amd.capi import Options
from oracle import oracle as orc            # only to pose the synthetic frame (tests/bigmodel.make_frame)
from bigmodel import extend_model, make_frame

smpl = synth.load_model(0)
for joints in (52, 55):
    m = extend_model(smpl, joints); gm = api.AvatarModel(m); om = orc.OracleModel(m)
    fr = make_frame(m, om, smpl, 3)
    ctx = api.Context(gm, joints, fr["part_map"], len(fr["labels"]), 1, device=0)
    w0, p0, R0 = fr["start"]
    for i in range(2):
        ctx.optimize_batch([fr["data"]], [fr["labels"]], Options.demo(), p0[None], api.rot_to_quat(R0)[None], w0[None])
    lib = capi.load_library(); buf = np.zeros(64)
    lib.avt_debug_trace(ctx.h, 0, buf.ctypes.data_as(C.POINTER(C.c_double)))
    s = np.diff(buf[40:47])
    print("J=%d k_solve<1024> clocks: loads + LM decision %.0f | system assembly %.0f | LDL^T %.0f | back substitution %.0f | retraction %.0f | skeleton pass %.0f | total %.0f"
          % (joints, s[0], s[1], s[2], s[3], s[4], s[5], buf[46] - buf[40]))
    print("   skeleton pass: joint positions + barrier %.0f | level loop %.0f | outputs %.0f" % (buf[62] - buf[45], buf[63] - buf[62], buf[46] - buf[63]))
