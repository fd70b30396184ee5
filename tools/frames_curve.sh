#!/bin/bash
# frames per GPU -> ms per optimize() with the library's defaults (run through gpurun): tools/frames_curve.sh > gpurun_out/r06_frames_per_gpu_curve.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
echo "# frames per GPU -> ms per optimize() (1 ICP x 10 GN iterations, gain-ratio default, stopping rule off), GN iterations/s, data term; bench.py --frames F, round-6 defaults"
for F in 1 2 3 4 6 8 12 16 24 32 44 48 64 96 128 192 256 384 512; do
  steps=$([ $F -le 64 ] && echo 20 || echo 5)
  python $R/bench.py --frames $F --steps $steps --warmup 3 --regions 5 --no-cpu-baseline --no-throughput-config --no-label-stage --no-render-stage --no-shard --no-dense-config --no-seed-spread --saturation-frames 0 --detail-file /tmp/fc.json 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); f = json.load(open('/tmp/fc.json'))
print('%4d frames  %.4f ms  %10.0f it/s  %s  groups x frames per launch %s' % ($F, d['ms_per_step'], d['value'], f['tuning']['data_term_run'], f['roofline']['launch_shape']))"
done
