"""The C++ tracker (tests/cpp/tracker_demo) with and without the stopping rule, alternately, same sequence (bench.py's tracker_stage frames)."""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from avatar_amd import synth
from tests.test_gpu_facade import write_model_dir
from tests.test_gpu_tracker import write_sequence
smpl = synth.load_model(0)
w, p, R = synth.sample_ground_truth(smpl, 21, use_gmm=False)
w = 0.5 * w
frames = []
for k in range(6):
    Rk = R.copy()
    Rk[16] = R[16] @ synth.rodrigues([0.0, 0.0, 0.05 * k]); Rk[4] = R[4] @ synth.rodrigues([0.04 * k, 0.0, 0.0])
    xyz, mask, _ = synth.render_images(smpl, synth.pose_vertices(smpl, w, p + np.array([0.01 * k, 0.0, 0.0]), Rk), synth.identity_part_map())
    ys, xs = np.nonzero(mask != 255)
    frames.append((xyz, mask, (ys.min(), xs.min(), ys.max(), xs.max())))
exe = os.path.join(ROOT, "tests", "cpp", "tracker_demo")
reps = sys.argv[1] if len(sys.argv) > 1 else "20"
with tempfile.TemporaryDirectory() as td:
    write_model_dir(smpl, os.path.join(td, "model"))
    write_sequence(os.path.join(td, "seq.bin"), frames, 3, 3, 6, 1000)
    for tol in ("1e-4", "0", "1e-4", "0", "1e-2", "1e-4"):
        r = subprocess.run([exe, os.path.join(td, "model"), os.path.join(td, "seq.bin"), os.path.join(td, "out.bin"), reps, tol], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, **({"AVT_NO_GRAPH": "1"} if len(sys.argv) > 2 else {})))
        print(tol, [l for l in r.stdout.splitlines() if "timing" in l], r.stderr[-200:])
