#!/bin/bash
# headline frame + 12-seed spread with a tuning knob at several values: tools/spec_ab6.sh VAR "v1 v2 .."
R=${GRAFT_REPO_ROOT:-/root/repo}
VAR=$1; VALS=$2
for rep in 1 2; do for V in $VALS; do
  env $VAR=$V python $R/bench.py --no-cpu-baseline --no-throughput-config --no-label-stage --no-render-stage --no-shard --no-dense-config --saturation-frames 0 --detail-file /tmp/d_$V.json > /dev/null 2>&1
  python - $VAR $V <<'PY'
import json, sys
d = json.load(open("/tmp/d_%s.json" % sys.argv[2]))
s = d["single_frame_spread"]
print("%s=%s headline %.4f ms  12 seeds mean %.4f median %.4f  by seed %s" % (sys.argv[1], sys.argv[2], d["ms_per_step"], s["ms_per_step"]["mean"], s["ms_per_step"]["median"], s["by_seed_ms"]))
PY
done; done
