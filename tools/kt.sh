#!/bin/bash
# kernel trace of one bench configuration: tools/kt.sh <frames> [lib]   (run through gpurun from the repo root)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; F=$1
[ -n "$2" ] && export AVT_LIB=$R/$2
COMMON="--no-cpu-baseline --no-throughput-config --no-label-stage --no-render-stage --no-shard --no-dense-config --no-seed-spread --saturation-frames 0 --regions 3"
rm -rf $O/prof_kt_$F
rocprofv3 --kernel-trace --stats -d $O/prof_kt_$F -o p -- python $R/bench.py --frames $F --steps 5 --warmup 2 $COMMON > $O/prof_kt_$F.log 2>&1
grep "^{\"metric\"" $O/prof_kt_$F.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('frames', d['config']['frames_per_gpu'], 'value', d['value'], 'ms/step', d['ms_per_step'])"
python $R/tools/rocpd_stats.py $(find $O/prof_kt_$F -name "*.db" | head -1) | head -${3:-16}
