#!/bin/bash
# Reproduces the rocprofv3 artefacts under profiles/ on an MI355X box (run through gpurun from the repo root):
#   kernel traces (--kernel-trace --stats) and, in separate passes, the FETCH_SIZE / WRITE_SIZE counters for the two
#   bench configurations, plus the SQ counters of the evaluation kernel.  Output: gpurun_out/prof_<tag>/ and *.txt / *.json.
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
for F in 1 64; do
  tag=$([ $F = 1 ] && echo single_frame || echo 64_frames)
  steps=$([ $F = 1 ] && echo 25 || echo 5)
  cmd="python $R/bench.py --frames $F --steps $steps --warmup 2 --no-cpu-baseline --no-throughput-config --no-label-stage --no-render-stage"
  rocprofv3 --kernel-trace --stats -d $O/prof_kt_$tag -o p -- $cmd > $O/prof_kt_$tag.log 2>&1
  python $R/tools/rocpd_stats.py $(find $O/prof_kt_$tag -name "*.db" | head -1) > $O/r01_rocprof_kernel_trace_$tag.txt
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr -d $O/prof_pmc_${tag}_$ctr -o p -- $cmd > $O/prof_pmc_${tag}_$ctr.log 2>&1
  done
  python $R/tools/pmc_summary.py $(find $O/prof_pmc_${tag}_FETCH_SIZE -name "*.db" | head -1) $(find $O/prof_pmc_${tag}_WRITE_SIZE -name "*.db" | head -1) \
      "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} -- python bench.py --frames $F" $O/r01_pmc_$tag.json > /dev/null
done
cmd="python $R/bench.py --frames 64 --steps 3 --warmup 1 --no-cpu-baseline --no-throughput-config --no-label-stage --no-render-stage"
: > $O/r01_pmc_eval_sq_counters_64_frames.txt
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_WAIT_ANY"; do
  n=$(echo $set | cut -d" " -f1)
  rocprofv3 --pmc $set -d $O/prof_sq_$n -o p -- $cmd > $O/prof_sq_$n.log 2>&1
  python $R/tools/pmc_counters.py $(find $O/prof_sq_$n -name "*.db" | head -1) k_eval >> $O/r01_pmc_eval_sq_counters_64_frames.txt
done
ls -la $O/*.txt $O/*.json
