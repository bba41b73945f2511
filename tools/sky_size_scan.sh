#!/bin/bash
# GPU box: the bench at several sky cube sizes (how much of a step is the latency of sky texels that miss the caches?)
for s in 64 1024 2048 4096; do
  echo -n "sky $s: "
  timeout -k 10 120 python bench.py --sky-size $s --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%.0f Mrays/s %.2f ms" % (d["value"], d["ms_per_step"]))' || exit 1
done
