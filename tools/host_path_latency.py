"""Wall time of the host-buffer entry points (what a tracker pays per frame): avt_optimize through the facade with a
~38k-point frame and with the tracker's subsampled frame.  Usage: python tools/host_path_latency.py"""
import sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
from avatar_amd import api, synth

smpl = synth.load_model(0); gm = api.AvatarModel(smpl)
fr = synth.make_frame(smpl, 0)
pm = synth.identity_part_map()
for name, sel in (("full frame", slice(None)), ("every 9th pixel (tracker interval 3)", slice(0, None, 9))):
    data, labels = fr["data"][sel], fr["labels"][sel]
    ava = api.Avatar(gm)
    opt = api.AvatarOptimizer(ava, None, (1280, 720), 24, pm, max_points=65536)
    opt.betaPose, opt.betaShape = 0.05, 0.12
    w0, p0, R0 = fr["start"]
    ts = []
    for i in range(30):
        ava.w, ava.p, ava.r = w0.copy(), p0.copy(), R0.copy()
        t0 = time.perf_counter()
        opt.optimize(data, labels, 1, 4)
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts[5:]) * 1e3
    print("%-40s N=%6d  optimize() wall: median %.3f ms  min %.3f ms" % (name, len(labels), np.median(ts), ts.min()))

# breakdown of one facade call
import ctypes as C
data, labels = fr["data"], fr["labels"]
ava = api.Avatar(gm)
opt = api.AvatarOptimizer(ava, None, (1280, 720), 24, pm, max_points=65536)
opt.betaPose, opt.betaShape = 0.05, 0.12
w0, p0, R0 = fr["start"]
acc = {"rot_to_quat": 0.0, "optimize_batch": 0.0, "quat_to_rot": 0.0, "update": 0.0}
for i in range(30):
    ava.w, ava.p, ava.r = w0.copy(), p0.copy(), R0.copy()
    t0 = time.perf_counter(); q = api.rot_to_quat(ava.r); t1 = time.perf_counter()
    p, qq, w, st = opt.ctx.optimize_batch([data], [labels], opt.options(1, 4), ava.p[None], q[None], ava.w[None]); t2 = time.perf_counter()
    ava.p, ava.w = p[0], w[0]; ava.r = api.quat_to_rot(qq[0]); t3 = time.perf_counter()
    ava.update(); t4 = time.perf_counter()
    if i >= 5:
        for k, v in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)): acc[k] += v / 25
print({k: round(v * 1e3, 4) for k, v in acc.items()}, "ms")
