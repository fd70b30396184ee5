#!/bin/bash
# Reproduces the rocprofv3 artefacts under profiles/ on an MI355X box (run through gpurun from the repo root):
#   kernel traces (--kernel-trace --stats) and, in separate passes, the FETCH_SIZE / WRITE_SIZE counters for the bench
#   configurations, plus SQ counters of the evaluation, solve and nearest-neighbour kernels.  A frame batch runs as frame
#   groups, so every summary is PER LAUNCH SHAPE (tools/rocpd_stats.py, pmc_summary.py, pmc_counters.py group by grid);
#   the instrumented pass of bench.py uses the same shapes as the graph replay.
# Output: gpurun_out/prof_<tag>/ and gpurun_out/<round>_*.txt / *.json (copy what is to be judged into profiles/).
set -u
RND=${RND:-r05}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
COMMON="--no-cpu-baseline --no-throughput-config --no-label-stage --no-render-stage --no-shard --no-dense-config --no-seed-spread --saturation-frames 0 --regions 3"
# configurations: "<frames>[d]" (d = dense 150k-point frames)
for CFG in ${FRAMES:-1 64 512 1d 16d 64d}; do
  F=${CFG%d}; DENSE=""; TAG=""; [ "$CFG" != "$F" ] && DENSE="--dense" && TAG="_dense"
  nfg=$(python -c "print($F if $F < 32 else ($F + 1) // 2)")     # frames per launch: two frame groups from 32 frames on
  steps=$([ $F = 1 ] && echo 25 || echo 5)
  cmd="python $R/bench.py --frames $F $DENSE --steps $steps --warmup 2 $COMMON"
  rocprofv3 --kernel-trace --stats -d $O/prof_kt_$CFG -o p -- $cmd > $O/prof_kt_$CFG.log 2>&1
  python $R/tools/rocpd_stats.py $(find $O/prof_kt_$CFG -name "*.db" | head -1) > $O/${RND}_rocprof_kernel_trace_${F}_frames${TAG}.txt
  # the workload the records belong to (bench.py only uses a record whose points per frame match what it times)
  npts=$(grep -h '"points_per_frame"' $O/prof_kt_$CFG.log | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['config']['points_per_frame'])" 2>/dev/null || echo 0)
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr -d $O/prof_pmc_${CFG}_$ctr -o p -- $cmd > $O/prof_pmc_${CFG}_$ctr.log 2>&1
  done
  python $R/tools/pmc_summary.py $(find $O/prof_pmc_${CFG}_FETCH_SIZE -name "*.db" | head -1) $(find $O/prof_pmc_${CFG}_WRITE_SIZE -name "*.db" | head -1) \
      "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} -- python bench.py --frames $F $DENSE ($nfg frames per launch, $npts points per frame)" \
      $O/${RND}_pmc_${nfg}_frames_per_launch${TAG}.json $npts > /dev/null
done
if [ "${SQ:-1}" = 1 ]; then
  for F in ${SQ_FRAMES:-1 64 512}; do
    cmd="python $R/bench.py --frames $F --steps 3 --warmup 1 $COMMON"
    out=$O/${RND}_pmc_sq_counters_${F}_frames.txt
    : > $out
    for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_WAIT_ANY"; do
      n=$(echo $set | cut -d" " -f1)
      rocprofv3 --pmc $set -d $O/prof_sq_${F}_$n -o p -- $cmd > $O/prof_sq_${F}_$n.log 2>&1
      python $R/tools/pmc_counters.py $(find $O/prof_sq_${F}_$n -name "*.db" | head -1) k_eval k_solve k_nn k_reduce k_pairpass k_assemble k_moments k_compact >> $out
    done
  done
fi
# the rocpd databases are tens of MiB each: only the summaries travel back (gpurun merges at most 64 MiB)
rm -rf $O/prof_kt_* $O/prof_pmc_* $O/prof_sq_*
ls -la $O/${RND}_*
