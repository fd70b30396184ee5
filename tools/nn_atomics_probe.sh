#!/bin/bash
# what the integer atomics of the nearest-neighbour bookkeeping cost: the nn class of bench.py's instrumented pass with the shipped
# library and with a build whose nn_record skips them (make -C avatar_amd/csrc libavatar_hip_nn_noatom.so; results are wrong there)
for fr in ${FRS:-64 512}; do for lib in avatar_amd/csrc/libavatar_hip.so avatar_amd/csrc/libavatar_hip_nn_noatom.so avatar_amd/csrc/libavatar_hip_nn_nomerge.so; do
  AVT_LIB=$PWD/$lib python bench.py --frames $fr --steps 5 --warmup 2 --regions 3 --no-cpu-baseline --no-shard 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']['nn']; print('frames $fr lib $lib: nn class', k['ms'], 'ms in', k['launches'], 'launches; step', d['ms_per_step'], 'ms')"
done; done
