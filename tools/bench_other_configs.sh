#!/bin/bash
# GPU box: one bench line for each of the other configurations (the reference-shaped single dispatch at depth 8 and 32,
# the 4K / 4096-frame mesh scene, the sphere-only scene, the Cornell box) and the short-launch table, into
# gpurun_out/r03_final/.  Usage: bash tools/bench_other_configs.sh
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_final; mkdir -p $O
for wl in interactive_1080p_d8_spp1 interactive_1080p_d32_spp1 bunny_4k_d8_spp4096 three_balls_1080p_d8_spp256 cornell_640x360_d4_spp16; do
  timeout -k 10 250 python bench.py --workload $wl --no-cpu-baseline > $O/bench_$wl.json 2> $O/bench_$wl.err || { echo "failed $wl"; tail -3 $O/bench_$wl.err; exit 1; }
  cut -c1-200 $O/bench_$wl.json
done
timeout -k 10 200 python tools/frame_latency.py > $O/frame_latency.txt 2>&1 || exit 1
cat $O/frame_latency.txt
