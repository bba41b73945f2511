#!/bin/bash
# GPU box: the 10 M-triangle field at 4K (BASELINE configs[4]) on one GPU: a plain bench line, then the two HBM
# traffic counters, each in a run of its own.  Usage: bash tools/profile_field.sh <tag> [frames]
set -o pipefail
TAG=${1:-field}
FRAMES=${2:-32}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
ARGS="--workload field_4k_d8_spp1024 --frames $FRAMES --steps 2 --warmup 1 --no-cpu-baseline --sky-size 2048"
timeout -k 10 280 python3 $B $ARGS > $OUT/bench.json 2> $OUT/bench.err || { echo "bench failed"; tail -5 $OUT/bench.err; exit 1; }
echo "bench done"
for pass in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 280 rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$pass -- python3 $B $ARGS > $OUT/pmc_$pass.log 2>&1 || { echo "pmc pass failed: $pass"; tail -3 $OUT/pmc_$pass.log; exit 1; }
  echo "$pass done"
done
python3 $GRAFT_REPO_ROOT/tools/summarize_prof.py $OUT --frames $FRAMES --workload field_4k_d8_spp1024 > $OUT/summary.txt 2>&1
cat $OUT/summary.txt; cat $OUT/bench.json
