#!/bin/bash
export PTRT_LIB=${PTRT_LIB:-$PWD/path-tracer_amd/libptrt_tuning.so}
run() { echo -n "$* -> "; env "$@" timeout -k 10 120 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%.0f Mrays/s %.2f ms" % (d["value"], d["ms_per_step"]))' || exit 1; }
run X=0
for g in 1024 768 512 256; do run PTRT_POOL_GROUPS=$g; done
