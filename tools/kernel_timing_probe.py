"""In-kernel phase timing (s_memtime) of k_eval / k_solve.  Needs a -DAVT_TIMING build: `touch avt_lm.hip && make -C avatar_amd/csrc EXTRA=-DAVT_TIMING` for k_solve, and avt_eval.o compiled by hand with -DAVT_TIMING for k_eval (the Makefile's EXTRA only reaches avt_lm.o).  Usage: python tools/kernel_timing_probe.py [frames]"""
import sys, numpy as np, ctypes as C
sys.path.insert(0,'/root/repo')
from avatar_amd import api, synth, capi
from avatar_amd.capi import Options
F=int(sys.argv[1]) if len(sys.argv)>1 else 64
smpl=synth.load_model(0); gm=api.AvatarModel(smpl)
frs=[synth.make_frame(smpl,s%8) for s in range(F)]; pm=synth.identity_part_map()
ctx=api.Context(gm,24,pm,60000,F)
p0=np.array([f['start'][1] for f in frs]); q0=np.array([api.rot_to_quat(f['start'][2]) for f in frs]); w0=np.array([f['start'][0] for f in frs])
opt=Options.demo()
for i in range(3):
    ctx.optimize_batch([f['data'] for f in frs],[f['labels'] for f in frs],opt,p0,q0,w0)
lib=capi.load_library(); buf=np.zeros(64)

lib.avt_debug_trace(ctx.h,0,buf.ctypes.data_as(C.POINTER(C.c_double)))
names=['barrier(top)','rec->LDS+zero','xhat+xk+T','jac+shape','barrier(mid)','mfma(last batch)+tail']
t=buf[48:54]; print('F',F,'eval block0 cycles per phase:'); [print('  %-14s %10.0f  %5.1f%%'%(n,v,100*v/t.sum())) for n,v in zip(names,t)]; print('  total',t.sum(), 'cycles; wall (100 MHz counter) %.2f us -> shader clock %.2f GHz' % (buf[54]/100.0, t.sum()/(buf[54]*10.0)))
s=np.diff(buf[41:47]); print('k_solve cycles: system assembly %.0f | LDL^T %.0f | back substitution %.0f | retraction %.0f | skeleton pass %.0f' % tuple(s))



# skeleton-pass internal probes live in the last two doubles of the prep block of the try slot (AVT_TIMING builds)
print('skeleton pass: joint positions + barrier %.0f | level loop %.0f | outputs %.0f cycles' % (buf[62]-buf[45], buf[63]-buf[62], buf[46]-buf[63]))
