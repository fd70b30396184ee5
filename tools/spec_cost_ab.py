"""A/B of the folded accept tests (avt_tuning.spec_cost): ms per optimize() of one frame over the twelve bench seeds, with the accept /
reject pattern of each, both settings in one process (same box, alternating)."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avatar_amd import api, synth
from avatar_amd.capi import Options
smpl = synth.load_model(0); gm = api.AvatarModel(smpl); pm = synth.identity_part_map()
opt = Options.demo()
SETTINGS = tuple(int(a) for a in sys.argv[1:]) or (2, 0)
ctxs = {}
for sc in SETTINGS:
    ctxs[sc] = api.Context(gm, 24, pm, 65536, 1)
    ctxs[sc].set_tuning(spec_cost=sc)
tot = {sc: 0.0 for sc in SETTINGS}
for sd in range(12):
    gt = synth.sample_ground_truth(smpl, sd); st = synth.perturb_start(*gt, sd)
    row = {}
    for sc in SETTINGS:
        ctx = ctxs[sc]
        ctx.render_frames(gt[0][None], gt[1][None], gt[2][None])
        ctx.state_upload(st[1][None], api.rot_to_quat(st[2].reshape(-1, 3, 3)).reshape(1, 24, 4), st[0][None])
        for _ in range(5):
            ctx.state_reset(); ctx.optimize_resident(opt)
        ctx.sync()
        best = 1e9
        for rep in range(5):
            t0 = time.perf_counter()
            for _ in range(40):
                ctx.state_reset(); ctx.optimize_resident(opt)
            ctx.sync()
            best = min(best, (time.perf_counter() - t0) / 40 * 1e3)
        p, q, w, s = ctx.state_download()
        tr = ctx.cost_trace(0)
        row[sc] = (best, "".join("A" if tr[i + 1] < tr[i] else "R" for i in range(10)), p, q, w, s[0].gn_iterations, s[0].lambda_)
        tot[sc] += best
    ref = row[SETTINGS[-1]]
    same = all(np.array_equal(row[sc][2], ref[2]) and np.array_equal(row[sc][3], ref[3]) and np.array_equal(row[sc][4], ref[4]) and row[sc][5] == ref[5] and row[sc][6] == ref[6] for sc in SETTINGS)
    print("seed %2d  %s  " % (sd, ref[1]) + "  ".join("spec_cost=%d %.4f ms" % (sc, row[sc][0]) for sc in SETTINGS) + "  same bits %s  gn %d" % (same, ref[5]))
print("mean over the seeds: " + "  ".join("spec_cost=%d %.4f ms" % (sc, tot[sc] / 12) for sc in SETTINGS))
