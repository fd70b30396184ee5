cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
COMMON="--no-cpu-baseline --no-throughput-config --no-label-stage --no-render-stage --no-shard --no-dense-config --no-seed-spread --saturation-frames 0 --regions 3"
for F in 1 64; do
rm -rf $O/prof_m_$F
rocprofv3 --kernel-trace --stats -d $O/prof_m_$F -o p -- python $R/bench.py --frames $F --data-term moments --steps 5 --warmup 2 $COMMON > $O/prof_m_$F.log 2>&1
grep '^{"metric"' $O/prof_m_$F.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('frames', d['config']['frames_per_gpu'], 'moments forced: value', d['value'], 'ms/step', d['ms_per_step'])"
python $R/tools/rocpd_stats.py $(find $O/prof_m_$F -name "*.db" | head -1) | head -12
rm -rf $O/prof_m_$F
done
