"""In-kernel phase timing (s_memtime) of the evaluation kernels, workgroup 0 of the first frame of the launch, per wave.
Needs the instrumented library: make -C avatar_amd/csrc libavatar_hip_timing.so, then
    AVT_LIB=avatar_amd/csrc/libavatar_hip_timing.so python tools/eval_phase_probe.py [frames]
Phases (cycles summed over the workgroup's batches): A-wait (barrier before the builder: the other waves' matrix phase) |
mfma (the wave's own matrix phase) | records-wait (the prefetched records arrive and go to LDS) | build (rows of the tile) | B-wait (barrier before the matrix phase)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avatar_amd import api, capi, synth  # noqa: E402
from avatar_amd.capi import Options  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 64
smpl = synth.load_model(0); gm = api.AvatarModel(smpl)
frs = [synth.make_frame(smpl, s % 8) for s in range(min(F, 8))]
frs = [frs[s % len(frs)] for s in range(F)]
pm = synth.identity_part_map()
ctx = api.Context(gm, 24, pm, 60000, F)
p0 = np.array([f['start'][1] for f in frs]); q0 = np.array([api.rot_to_quat(f['start'][2]) for f in frs]); w0 = np.array([f['start'][0] for f in frs])
opt = Options.demo()
for i in range(2):
    ctx.optimize_batch([f['data'] for f in frs], [f['labels'] for f in frs], opt, p0, q0, w0)
g, nfg, G = ctx.launch_shape()
lib = capi.load_library(); buf = np.zeros(64)
lib.avt_debug_trace(ctx.h, 0, buf.ctypes.data_as(C.POINTER(C.c_double)))
names = ['A-wait', 'mfma', 'records-wait', 'build', 'B-wait', '-', '-', '-']
print(f"F={F} groups={g} frames/launch={nfg} G={G}; last full evaluation launch, workgroup 0 of frame 0; wall {buf[56] / 100.0:.2f} us")
for wv in range(4):
    t = buf[16 + 8 * wv:24 + 8 * wv]
    print(f"  wave {wv}: total {t.sum():8.0f} cycles | " + " | ".join(f"{n} {v:7.0f}" for n, v in zip(names, t) if n != '-'))
