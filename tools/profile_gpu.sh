#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats, then PMC passes, each in its own run.
# Usage: bash tools/profile_gpu.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-r01}; shift
ARGS=${@:---frames 64 --steps 2 --warmup 1 --no-cpu-baseline}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
timeout -s KILL 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $B $ARGS > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" \
            "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
            "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | tr ' ' '+' | cut -c1-40)
  timeout -s KILL 200 rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$name -- python3 $B $ARGS > $OUT/pmc_$name.log 2>&1 || { echo "pmc pass failed: $pass"; tail -3 $OUT/pmc_$name.log; }
done
python3 $GRAFT_REPO_ROOT/tools/summarize_prof.py $OUT $ARGS > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
