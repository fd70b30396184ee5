#!/usr/bin/env python3
"""Fixed-factor LM damping (x4 up, /3 down) against the gain-ratio schedule (avt_options.lm_policy = 1) on the bench frames,
with the CPU oracle (same objective, same step rule as the GPU): accepted steps of 10 and final objective per seed.
`python tools/damping_policy_compare.py sweep` adds the table Options.GAIN_LM_UP was chosen from (lm_up = the multiplier of the
first rejection after an accepted step under the gain-ratio schedule; Nielsen's value is 2)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from avatar_amd import synth  # noqa: E402
from avatar_amd.capi import Options  # noqa: E402
from oracle import oracle as orc  # noqa: E402

smpl = synth.load_model(0)
om = orc.OracleModel(smpl)
pm = synth.identity_part_map()
orc.set_nn_implementation("nanoflann")
rows = []
for seed in range(12):
    fr = synth.make_frame(smpl, seed)
    w0, p0, R0 = fr["start"]
    q0 = orc.rot_to_quat(R0)
    out = []
    for pol in (0, 1):
        for icp in (1, 3):
            opt = Options.demo(icp_iters=icp, lm_policy=pol)
            r = om.optimize(pm, 24, fr["data"], fr["labels"], opt, p0, q0, w0, aggregate=1)
            out.append((r["stats"].accepted_steps, r["stats"].gn_iterations, r["stats"].final_cost))
    rows.append(out)
    print("seed %2d | fixed: %2d/%2d cost %.6f ; 3 ICP %2d/%2d cost %.6f | gain ratio: %2d/%2d cost %.6f ; 3 ICP %2d/%2d cost %.6f" % (
        seed, out[0][0], out[0][1], out[0][2], out[1][0], out[1][1], out[1][2], out[2][0], out[2][1], out[2][2], out[3][0], out[3][1], out[3][2]))
a = np.array([[x[0] / x[1] for x in r] for r in rows])
c = np.array([[x[2] for x in r] for r in rows])
print("accepted fraction, mean over seeds: fixed %.3f (1 ICP) %.3f (3 ICP); gain ratio %.3f / %.3f" % tuple(a.mean(0)[[0, 1, 2, 3]]))
print("seeds where the gain-ratio schedule ends at a lower objective: %d of 12 (1 ICP), %d of 12 (3 ICP); mean ratio of final objectives %.4f / %.4f" % (
    (c[:, 2] < c[:, 0]).sum(), (c[:, 3] < c[:, 1]).sum(), (c[:, 2] / c[:, 0]).mean(), (c[:, 3] / c[:, 1]).mean()))

if len(sys.argv) > 1 and sys.argv[1] == "sweep":
    frames = []
    for seed in range(12):
        fr = synth.make_frame(smpl, seed)
        w0, p0, R0 = fr["start"]
        frames.append((fr, w0, p0, orc.rot_to_quat(R0)))
    print("lm_policy lm_up lm_down | accepted fraction 1 ICP, 3 ICP | mean final objective 1 ICP, 3 ICP")
    for pol, up, down in [(0, 4, 1 / 3), (0, 16, 1 / 3), (0, 64, 1 / 3), (1, 2, 1 / 3), (1, 4, 1 / 3), (1, 8, 1 / 3), (1, 16, 1 / 3), (1, 32, 1 / 3), (1, 16, 0.5)]:
        a = []
        for fr, w0, p0, q0 in frames:
            row = []
            for icp in (1, 3):
                r = om.optimize(pm, 24, fr["data"], fr["labels"], Options.demo(icp_iters=icp, lm_policy=pol, lm_up=up, lm_down=down), p0, q0, w0, aggregate=1)
                row += [r["stats"].accepted_steps / r["stats"].gn_iterations, r["stats"].final_cost]
            a.append(row)
        a = np.array(a)
        print("%d %5g %.3f | %.3f %.3f | %.4f %.4f" % (pol, up, down, a[:, 0].mean(), a[:, 2].mean(), a[:, 1].mean(), a[:, 3].mean()))
