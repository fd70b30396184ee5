"""Where the host-to-host call avt_optimize(host pointers) spends its wall time on top of the resident fit: the four stages of
avt_optimize_batch timed one by one through the C ABI (prepared contiguous arrays, no Python array work inside the timed calls)."""
import ctypes as C, sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avatar_amd import api, capi, synth
from avatar_amd.capi import Options, Stats, dptr, iptr

smpl = synth.load_model(0); gm = api.AvatarModel(smpl); pm = synth.identity_part_map()
fr = synth.make_frame(smpl, 0)
data = np.ascontiguousarray(fr["data"], np.float64); lab = np.ascontiguousarray(fr["labels"], np.int32)
w0, p0, R0 = fr["start"]; q0 = api.rot_to_quat(R0)
ctx = api.Context(gm, 24, pm, 65536, 1)
lib = ctx._lib; h = ctx.h
opt = Options.demo()
offs = np.array([0, len(lab)], np.int32)
p = p0.copy(); q = q0.ravel().copy(); w = w0.copy(); st = Stats()
def T(fn, n=200):
    for _ in range(20): fn()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n * 1e6
up_f = lambda: lib.avt_frames_upload(h, 1, dptr(data), iptr(lab), iptr(offs))
up_s = lambda: lib.avt_state_upload(h, 1, dptr(p0), dptr(q0.ravel()), dptr(w0))
def run():
    lib.avt_state_reset(h); lib.avt_optimize_resident(h, C.byref(opt)); lib.avt_sync(h)
dn = lambda: lib.avt_state_download(h, dptr(p), dptr(q), dptr(w), C.byref(st))
def whole():
    p[:] = p0; q[:] = q0.ravel(); w[:] = w0
    lib.avt_optimize(h, dptr(data), iptr(lab), len(lab), C.byref(opt), dptr(p), dptr(q), dptr(w), C.byref(st))
up_f(); up_s(); run()
print("N = %d points (%.2f MB host to device)" % (len(lab), 28 * len(lab) / 1e6))
print("avt_frames_upload   %8.1f us" % T(up_f))
print("avt_state_upload    %8.1f us" % T(up_s))
print("reset + optimize_resident + sync %8.1f us" % T(run))
print("avt_state_download  %8.1f us" % T(dn))
print("avt_optimize (all of it, host pointers) %8.1f us" % T(whole))
