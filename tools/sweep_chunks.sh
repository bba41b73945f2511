#!/bin/bash
# the PTRT_* overrides are read by the tuning build only (make -C path-tracer_amd/csrc tuning)
export PTRT_LIB=${PTRT_LIB:-$PWD/path-tracer_amd/libptrt_tuning.so}
# GPU box: shortest planned item (frames) against items per resident slot, bench scene.
run() { echo -n "$* -> "; env "$@" timeout -k 10 120 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%.0f Mrays/s %.2f ms" % (d["value"], d["ms_per_step"]))' || exit 1; }
run X=0
for c in 1 2; do for s in 16 24 32 40 48 56; do run PTRT_MIN_CHUNK=$c PTRT_ITEMS_PER_SLOT=$s; done; done
run X=1
