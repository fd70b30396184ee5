"""bench.py's contract line: the line the driver parses stays small and carries `roofline` and
`cpu_baseline`; `--gpus N` decides by itself whether it must start the N ranks."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def canned():
    # a full result object of the round-3 run (25 KB as one line: what the driver could no longer parse)
    return json.load(open(os.path.join(ROOT, "profiles", "r03_bench_default.json")))


def test_compact_line_is_small_and_complete():
    out = canned()
    assert len(json.dumps(out)) > 20000
    line = bench.compact_line(out)
    assert "\n" not in line and len(line) < bench.COMPACT_LIMIT
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert d[k] == out[k], k
    assert d["config"]["workload"] and "model" not in d["config"]
    ro = d["roofline"]
    assert ro["frac"] == pytest.approx(ro["achieved"] / ro["peak"], rel=1e-3)
    assert ro["bound"] in ("hbm", "mfma") and ro["unit"] in ("GB/s", "TFLOP/s") and "traffic" in ro
    cb = d["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] in ("port", "reference") and cb["sample"]
    assert set(d["configs"]) == {"64_frames", "512_frames", "dense_1", "dense_64"}
    for c in d["configs"].values():
        assert c["value"] > 0 and c["roofline_frac"] > 0


def test_compact_line_survives_oversized_fields():
    out = canned()
    out["config"]["workload"] = "x" * 5000
    out["cpu_baseline"]["sample"] = "y" * 5000
    out["roofline"]["limiter"] = "z" * 5000
    line = bench.compact_line(out)
    assert len(line) < bench.COMPACT_LIMIT
    assert json.loads(line)["roofline"]["frac"] > 0


def test_spawn_decision():
    assert bench.spawn_plan(1, {}, 8) is None                          # one GPU: run in place
    assert bench.spawn_plan(4, {"WORLD_SIZE": "4"}, 8) is None         # already a rank of somebody's launch
    plan = bench.spawn_plan(4, {}, 8)
    assert plan[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in plan and "127.0.0.1" in plan
    with pytest.raises(SystemExit):
        bench.spawn_plan(8, {}, 1)                                     # fewer devices than ranks: refuse, loudly
    assert bench.spawn_plan(2, {}, 1, share_gpu0=True) is not None     # the explicit single-GPU dry run


def test_round5_accounting_fields():
    """(i) the nearest-neighbour class' PMC traffic is k_compact + k_nn_part, (ii) matrix-pipe utilisation is executed
    instructions over time - never above 1 -, (iii) k_solve under the moment form is priced on its own bytes with the SURVEY 8(d) figure
    beside it, (iv) the line carries a host-to-host value."""
    out = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_default.json")))
    line = bench.compact_line(out)
    assert len(line) < bench.COMPACT_LIMIT
    d = json.loads(line)
    assert d["value_host_to_host"] > 0 and d["value_host_to_host"] < d["value"] and d["host_to_host"]["equals_resident_run"] is True
    assert 0 < d["mfma"]["frac"] <= 1.0 and d["mfma"]["kernel"] == "k_eval"
    for name, c in d["configs"].items():
        assert c["mfma_frac"] is not None and 0 < c["mfma_frac"] <= 1.0, name
        assert c["mfma_kernel"] == ("k_moments" if c["data_term"] == "moments" else "k_eval"), name
        if c["data_term"] == "moments" and c["roofline_kernel"] == "k_solve":
            assert c["roofline_frac"] < 0.05 and c["chain_us"] > 0, name          # a latency chain, not a third of HBM
    t64 = out["throughput_config"]["roofline"]
    assert t64["survey_8d_equivalent"]["frac"] > t64["frac"] and "chain_us" in t64
    # (i): on the committed round-4 records the class sums both kernels
    keep = bench.ROUND
    try:
        bench.ROUND = "r04"
        rec = json.load(open(os.path.join(ROOT, "profiles", "r04_pmc_256_frames_per_launch.json")))
        both = sum(v["hbm_bytes"] for k, v in rec["kernels"].items() if "k_compact" in k or "k_nn_part" in k)
        assert bench.pmc_traffic(256, "nn", rec["points_per_frame"]) == both > 9e8
        one = json.load(open(os.path.join(ROOT, "profiles", "r04_pmc_1_frames_per_launch.json")))
        assert bench.pmc_traffic(1, "nn", one["points_per_frame"]) == [v for k, v in one["kernels"].items() if "k_nn_vis" in k][0]["hbm_bytes"]
        assert bench.pmc_traffic(32, "eval_moments", 29408.0) is not None      # (k_prior is optional: it rides in k_pairpass up to 128 frames per launch)
    finally:
        bench.ROUND = keep
