#!/bin/bash
# GPU box: the 10 M-triangle field at 4K (BASELINE configs[4]) on one GPU.  First the two memory-traffic counters, each in
# a run of its own; their record goes where bench.py looks for it (profiles/<round>/<tag>/pmc_traffic.json, in the box's
# copy of the tree); then the bench line, which therefore carries the traffic of this very session.
# Usage: bash tools/profile_field.sh <round dir, e.g. r03> <tag, e.g. field128> [frames] [extra bench args, e.g. --rebuild-bvh]
set -o pipefail
ROUND=${1:-r03}
TAG=${2:-field128}
FRAMES=${3:-128}
EXTRA=${4:-}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${ROUND}_$TAG
mkdir -p $OUT $GRAFT_REPO_ROOT/profiles/$ROUND/$TAG
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
ARGS="--workload field_4k_d8_spp1024 --frames $FRAMES --steps 2 --warmup 1 --no-cpu-baseline --sky-size 2048 $EXTRA"
for pass in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 280 rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$pass -- python3 $B $ARGS > $OUT/pmc_$pass.log 2>&1 || { echo "pmc pass failed: $pass"; tail -3 $OUT/pmc_$pass.log; exit 1; }
  echo "$pass done"
done
python3 $GRAFT_REPO_ROOT/tools/summarize_prof.py $OUT $ARGS > $OUT/summary.txt 2>&1
cp $OUT/pmc_traffic.json $GRAFT_REPO_ROOT/profiles/$ROUND/$TAG/pmc_traffic.json || exit 1
timeout -k 10 280 python3 $B $ARGS > $OUT/bench.json 2> $OUT/bench.err || { echo "bench failed"; tail -5 $OUT/bench.err; exit 1; }
cat $OUT/summary.txt; cat $OUT/bench.json
