"""The launch path a multi-GPU node takes, with two ranks on ONE GPU: `bench.py --gpus 2` starts its ranks
through torch.distributed.run, every rank builds its avt_shard, takes the model from rank 0's broadcast, runs its share of the
batch with the result all-gather inside every step, the scatter / gather round trip is checked on every rank, and rank 0 prints the
ONE compact line.  RCCL refuses two ranks on one device, so the exchanges go through the shared-memory transport
(avt_shard_create_shm, AVT_BENCH_SHARE_GPU0=1); everything above the transport is the code an 8-GPU run executes."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_two_ranks_one_gpu_spawn_shard_gather_compact_line(tmp_path):
    detail = tmp_path / "detail.json"
    env = dict(os.environ, AVT_BENCH_SHARE_GPU0="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--regions", "3",
                        "--detail-file", str(detail)], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, f"exactly one stdout line expected, got {len(lines)}: {r.stdout[-500:]}"
    assert len(lines[0]) < 4096
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    bs = d["batch_split"]
    assert bs["enabled"] is True and bs["world"] == 2 and bs["ok"] is True, bs
    assert "shared memory" in bs["backend"]
    full = json.loads(detail.read_text())
    run = full["batch_split"]["run"]
    assert run["gathered_equals_local"] and run["gathered_equals_local_on_every_rank"] and run["all_ranks_finite"] and run["frames_total"] == 2
    chk = full["batch_split"]["check"]
    assert chk.get("scatter_bit_exact_on_every_rank") and chk.get("gather_equals_local_on_every_rank") and chk.get("gathered_equals_single_process_run"), chk
    # the 64-frames-per-GPU leg ran on both ranks with the gather inside every step
    t = full["throughput_config"]
    assert t["frames_per_gpu"] == 64 and t["shard"]["frames_total"] == 128 and t["shard"]["gathered_equals_local_on_every_rank"]
