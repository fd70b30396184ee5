cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cmd="python $R/bench.py --frames 64 --steps 3 --warmup 1 --no-cpu-baseline --no-throughput-config --no-label-stage"
: > $O/sq_now.txt
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_WAIT_ANY"; do
  n=$(echo $set | cut -d" " -f1)
  rocprofv3 --pmc $set -d $O/prof_sq2_$n -o p -- $cmd > $O/prof_sq2_$n.log 2>&1
  python $R/tools/pmc_counters.py $(find $O/prof_sq2_$n -name "*.db" | head -1) k_eval >> $O/sq_now.txt
done
cat $O/sq_now.txt
