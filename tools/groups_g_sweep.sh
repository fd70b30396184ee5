for fr in 64 96; do for ng in 2 3 4; do for g in 8 12 16 24; do
  AVT_GROUPS=$ng AVT_G=$g python bench.py --frames $fr --steps 6 --warmup 2 --regions 5 --no-cpu-baseline --no-shard 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('frames $fr groups $ng Gcap $g', d['ms_per_step'], d['value'], d['roofline']['launch_shape'])"
done; done; done
