import sys, numpy as np, ctypes as C
sys.path.insert(0,'/root/repo')
from avatar_amd import api, synth, capi
from avatar_amd.capi import Options
smpl=synth.load_model(0); gm=api.AvatarModel(smpl)
fr=synth.make_frame(smpl,0); pm=synth.identity_part_map()
ctx=api.Context(gm,24,pm,60000,1)
w0,p0,R0=fr['start']; q0=api.rot_to_quat(R0)
opt=Options.demo()
for i in range(3):
    ctx.optimize_batch([fr['data']],[fr['labels']],opt,p0[None],q0[None],w0[None])
# read trace buffer: fb.trace is internal; expose via normal_equations? use hipMemcpy through torch? simpler: use avt debug getter
lib=capi.load_library()
buf=np.zeros(64)
lib.avt_debug_trace.argtypes=[C.c_void_p,C.c_int,C.POINTER(C.c_double)]
lib.avt_debug_trace(ctx.h,0,buf.ctypes.data_as(C.POINTER(C.c_double)))
t=buf[40:48]
print('clock deltas:', np.diff(t))
print('total', t[7]-t[0])
