#!/bin/bash
# the PTRT_* overrides are read by the tuning build only (make -C path-tracer_amd/csrc tuning)
export PTRT_LIB=${PTRT_LIB:-$PWD/path-tracer_amd/libptrt_tuning.so}
# GPU box: sweep the pooled kernel's scheduling knobs (environment overrides) with short bench runs.
run() { echo -n "$* -> "; env "$@" timeout -k 10 120 python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%.0f Mrays/s %.2f ms" % (d["value"], d["ms_per_step"]))' || exit 1; }
run X=0
run X=1
for v in 4 16 32; do run PTRT_POOL_STEPS=$v; done
for v in 8 32 48; do run PTRT_POOL_TOPUP=$v; done
for v in 16 24 48; do run PTRT_REFILL_LANES=$v; done
for v in 32 64 256; do run PTRT_POOL_SHADE_DENSE=$v; done
for v in 4 8 32; do run PTRT_POOL_DENSE=$v; done
for v in 0 16 64; do run PTRT_POOL_START=$v; done
for v in 16 24 32 64; do run PTRT_ITEMS_PER_SLOT=$v; done
for v in 2 8; do run PTRT_MIN_CHUNK=$v; done
for v in 16 64; do run PTRT_POOL_MISS_ITERS=$v; done
for v in 32 40 56; do run PTRT_POOL_MISS_LANES=$v; done
