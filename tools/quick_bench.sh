#!/bin/bash
# quick numbers for the three headline configurations (run through gpurun from the repo root): tools/quick_bench.sh [lib]
R=${GRAFT_REPO_ROOT:-/root/repo}
[ -n "$1" ] && export AVT_LIB=$R/$1
python $R/bench.py --no-cpu-baseline --no-dense-config --no-seed-spread --no-render-stage --no-label-stage 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('1 frame: %.4f ms/step (%.0f it/s)' % (d['ms_per_step'], d['value']))
for k, c in d.get('configs', {}).items():
    print('%s: %.4f ms/step (%.0f it/s)' % (k, c['ms_per_step'], c['value']))
"
