"""In-kernel phase timing (clock64, thread 0) of the last k_solve launch of an optimize() call.
Needs the instrumented library: make -C avatar_amd/csrc libavatar_hip_timing_lm.so, then
    AVT_LIB=avatar_amd/csrc/libavatar_hip_timing_lm.so python tools/solve_phase_probe.py [frames]
The per-round line needs libavatar_hip_timing_lm_rounds.so (its probes inflate the phase totals: read those from the plain timing build)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avatar_amd import api, capi, synth  # noqa: E402
from avatar_amd.capi import Options  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 1
smpl = synth.load_model(0); gm = api.AvatarModel(smpl)
frs = [synth.make_frame(smpl, s % 8) for s in range(min(F, 8))]
frs = [frs[s % len(frs)] for s in range(F)]
pm = synth.identity_part_map()
ctx = api.Context(gm, 24, pm, 60000, F)
p0 = np.array([f['start'][1] for f in frs]); q0 = np.array([api.rot_to_quat(f['start'][2]) for f in frs]); w0 = np.array([f['start'][0] for f in frs])
opt = Options.demo(**({"lm_policy": int(os.environ["PROBE_LM_POLICY"])} if "PROBE_LM_POLICY" in os.environ else {}))
if "PROBE_BETA_POSE" in os.environ: opt.beta_pose = float(os.environ["PROBE_BETA_POSE"])
for i in range(2):
    ctx.optimize_batch([f['data'] for f in frs], [f['labels'] for f in frs], opt, p0, q0, w0)
lib = capi.load_library(); buf = np.zeros(64)
lib.avt_debug_trace(ctx.h, 0, buf.ctypes.data_as(C.POINTER(C.c_double)))
s = np.diff(buf[40:47])
print("F=%d k_solve (last full solve of frame 0), shader clocks: loads + LM decision %.0f | system assembly %.0f | LDL^T %.0f | back substitution %.0f | retraction %.0f | skeleton pass %.0f | total %.0f"
      % (F, s[0], s[1], s[2], s[3], s[4], s[5], buf[46] - buf[40]))
if buf[48] > 0:
    print("back substitution: fold %.0f | barrier %.0f | chain + unknowns %.0f clocks" % (buf[48] - buf[43], buf[49] - buf[48], buf[44] - buf[49]))
if buf[50] > 0:
    print("riding hand-over: kernel start to poll %.0f | polling %.0f (%d spins) | poll end to decision made %.0f clocks" % (buf[50] - buf[40], buf[51] - buf[50], int(buf[52]), buf[41] - buf[51]))
    print("  poll end to: system requested %.0f | barrier (everything arrived) %.0f | accept test taken %.0f | control block written, roles sorted %.0f" % (buf[53] - buf[51], buf[54] - buf[53], buf[55] - buf[54], buf[41] - buf[55]))
if buf[50] <= 0 and buf[53] > 0:
    print("  staging requested and factor zeroed %.0f | ... index arithmetic up to the system's first request %.0f | its requests issued %.0f" % (buf[58] - buf[40], buf[56] - buf[58], buf[57] - buf[56]))
    print("batch shape, from the kernel's first instruction to: system requested %.0f | barrier (everything arrived) %.0f | accept test taken %.0f | prior's entries requested, control block written %.0f" % (buf[53] - buf[40], buf[54] - buf[53], buf[55] - buf[54], buf[41] - buf[55]))
# skeleton-pass internal probes live in the last two doubles of the prep block of the try slot
print("skeleton pass: joint positions + barrier %.0f | level loop %.0f | outputs %.0f clocks" % (buf[62] - buf[45], buf[63] - buf[62], buf[46] - buf[63]))

try:
    lib.avt_debug_mf_phases.argtypes = [C.POINTER(C.c_longlong), C.c_int]
    lib.avt_debug_mf_phases(None, 1)
    ctx.optimize_batch([f['data'] for f in frs], [f['labels'] for f in frs], opt, p0, q0, w0)
    ph = (C.c_longlong * 8)()
    lib.avt_debug_mf_phases(ph, 0)
    n = 10.0 * 22       # full solves of the call x rounds
    print("LDL^T rounds of workgroup 0, clocks per round: wave 0: barrier A %.0f | row phase %.0f | barrier B %.0f | matrix phase %.0f   wave 3: %.0f | %.0f | %.0f | %.0f"
          % tuple(v / n for v in ph))
except AttributeError:
    pass
