"""The speculative steps must not change a single bit: optimize() of several frames with AVT_NSPEC=0 (every step factored when it
is asked for) and with the default, in two processes, downloads compared.  Usage: python tools/spec_equivalence.py"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    from avatar_amd import api, synth
    from avatar_amd.capi import Options
    smpl = synth.load_model(0); gm = api.AvatarModel(smpl); pm = synth.identity_part_map()
    out = []
    for F in (1, 2):
        for seed in range(0, 12, F):
            frs = [synth.make_frame(smpl, seed + s) for s in range(F)]
            ctx = api.Context(gm, 24, pm, 60000, F)
            for opt in (Options.demo(), Options.demo(max_iters_per_icp=7, icp_iters=2)):
                p, q, w, st = ctx.optimize_batch([f["data"] for f in frs], [f["labels"] for f in frs], opt, np.array([f["start"][1] for f in frs]),
                                                 np.array([api.rot_to_quat(f["start"][2]) for f in frs]), np.array([f["start"][0] for f in frs]))
                out += [p.ravel(), q.ravel(), w.ravel(), np.array([s.final_cost for s in st]), np.array([s.accepted_steps for s in st], float)]
    np.save(sys.argv[2], np.concatenate(out))
    sys.exit(0)
res = []
for n in ("0", "2", "3"):
    env = dict(os.environ, AVT_NSPEC=n)
    path = "/tmp/spec_eq_%s.npy" % n
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "child", path], env=env)
    res.append(np.load(path))
same = all(np.array_equal(res[0], r) for r in res[1:])
print("values compared:", res[0].size, "| bit-identical with 0 / 2 / 3 speculative workgroups:", same)
sys.exit(0 if same else 1)
