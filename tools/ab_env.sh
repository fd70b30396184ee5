#!/bin/bash
# A/B of one environment knob over bench configurations on the GPU box: tools/ab_env.sh VAR "v1 v2 .." "frames .." [extra bench args]
# prints ms per step (median region) and the per-launch class means of the instrumented pass for every (value, frames); "64d" = 64 dense frames.
VAR=$1; VALS=$2; FRAMES=$3; shift 3
R=${GRAFT_REPO_ROOT:-/root/repo}
for F in $FRAMES; do
  for V in $VALS; do
    DENSE=""; FF=${F%d}; [ "$F" != "$FF" ] && DENSE="--dense"
    env $VAR=$V timeout 180 python $R/bench.py --frames $FF $DENSE --steps 10 --warmup 3 --regions 5 --no-cpu-baseline --no-shard --no-label-stage --no-render-stage --no-seed-spread --detail-file /tmp/ab_detail.json "$@" 2>/dev/null | tail -1 > /tmp/ab_line.json
    python - "$VAR" "$V" "$F" <<'PY'
import json, sys
d = json.load(open("/tmp/ab_line.json"))
full = json.load(open("/tmp/ab_detail.json"))
k = {n: round(v["ms"] / max(1, v["launches"]) * 1e3, 1) for n, v in full.get("kernels", {}).items()}
print(f"{sys.argv[1]}={sys.argv[2]:>3s} frames {sys.argv[3]:>4s}: {d['ms_per_step']:.4f} ms/step  value {d['value']:.0f}  chain {d['roofline'].get('chain_us_per_gn_iteration')}  us/launch {k}")
PY
  done
done
