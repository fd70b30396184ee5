"""In-kernel phase timing of the moment-form assembly (avt_moments.hip, -DAVT_TIMING build):
    make -C avatar_amd/csrc libavatar_hip_timing_mom.so
    AVT_LIB=avatar_amd/csrc/libavatar_hip_timing_mom.so python tools/moment_phase_probe.py [frames]"""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, '/root/repo')
from avatar_amd import api, synth, capi
from avatar_amd.capi import Options

F = int(sys.argv[1]) if len(sys.argv) > 1 else 1
smpl = synth.load_model(0); gm = api.AvatarModel(smpl)
frs = [synth.make_frame(smpl, s % 8) for s in range(F)]; pm = synth.identity_part_map()
ctx = api.Context(gm, 24, pm, 60000, F)
p0 = np.array([f['start'][1] for f in frs]); q0 = np.array([api.rot_to_quat(f['start'][2]) for f in frs]); w0 = np.array([f['start'][0] for f in frs])
opt = Options.demo()
for i in range(3):
    ctx.optimize_batch([f['data'] for f in frs], [f['labels'] for f in frs], opt, p0, q0, w0)
lib = capi.load_library(); buf = np.zeros(64)
lib.avt_debug_trace(ctx.h, 0, buf.ctypes.data_as(C.POINTER(C.c_double)))
names = ["staging (skeleton, lists, X16, Z)", "B-a (data side, partner sums of the records)", "B-b (per-joint sums)", "B-c (subtree sums)", "B-d (blocks but rot-rot)", "rot-rot"]
t = np.diff(buf[40:47])
print('F', F, 'k_assemble, clocks per phase (thread 0):')
for n, v in zip(names, t):
    print('  %-48s %9.0f  %5.1f%%' % (n, v, 100 * v / t.sum()))
print('  total %.0f clocks' % t.sum())

t = np.diff(buf[50:57])
print('k_moments, pair workgroup 20 of frame 0 (last pass), clocks: setup+list loads+cnt gather %.0f | scan+compaction %.0f | rounds %.0f | barrier %.0f | cross-wave sum %.0f | stores %.0f | total %.0f' % (t[0], t[1], t[2], t[3], t[4], t[5], t.sum()))

t = np.diff(buf[24:30])
print('k_pairpass, workgroup 5 of frame 0, clocks: ids + loads issued + staged in LDS %.0f | contraction (99 LDS reads per lane) %.0f | group sums %.0f | records, stores %.0f | workgroup sum of Z %.0f | total %.0f' % (t[0], t[1], t[2], t[3], t[4], t.sum()))
