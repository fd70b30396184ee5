"""Where and when every k_eval workgroup of the last GN iteration ran (needs the -DAVT_TIMING build; see kernel_timing_probe.py).
Usage: python tools/eval_block_timeline.py [frames]"""
import sys, numpy as np, ctypes as C
sys.path.insert(0, '/root/repo')
from avatar_amd import api, synth, capi
from avatar_amd.capi import Options
F = int(sys.argv[1]) if len(sys.argv) > 1 else 64
smpl = synth.load_model(0); gm = api.AvatarModel(smpl)
frs = [synth.make_frame(smpl, s % 8) for s in range(F)]; pm = synth.identity_part_map()
ctx = api.Context(gm, 24, pm, 60000, F)
p0 = np.array([f['start'][1] for f in frs]); q0 = np.array([api.rot_to_quat(f['start'][2]) for f in frs]); w0 = np.array([f['start'][0] for f in frs])
opt = Options.demo()
for i in range(3):
    ctx.optimize_batch([f['data'] for f in frs], [f['labels'] for f in frs], opt, p0, q0, w0)
lib = capi.load_library()
rows = []
for f in range(F):
    buf = np.zeros(64); lib.avt_debug_trace(ctx.h, f, buf.ctypes.data_as(C.POINTER(C.c_double)))
    for g in range(8):
        a, b, w = buf[12 + 3 * g: 15 + 3 * g]
        if b > a > 0: rows.append((f, g, a, b, int(w)))
rows = np.array(rows)
t0 = rows[:, 2].min()
st, en = (rows[:, 2] - t0) / 100.0, (rows[:, 3] - t0) / 100.0
hw = rows[:, 4].astype(int); xcc = hw >> 16; cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
place = xcc * 1000 + se * 100 + sh * 10 + cu
print('blocks', len(rows), 'kernel span %.1f us; block duration mean %.1f min %.1f max %.1f us' % (en.max(), (en - st).mean(), (en - st).min(), (en - st).max()))
print('start times: %d blocks within 5 us of the first, latest start %.1f us' % ((st < 5).sum(), st.max()))
u, c = np.unique(place, return_counts=True)
print('distinct (xcc,se,sh,cu) places', len(u), 'blocks per place histogram', np.bincount(c))
print('blocks per XCC', np.bincount(xcc))
