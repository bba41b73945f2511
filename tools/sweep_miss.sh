#!/bin/bash
# the PTRT_* overrides are read by the tuning build only (make -C path-tracer_amd/csrc tuning)
export PTRT_LIB=${PTRT_LIB:-$PWD/path-tracer_amd/libptrt_tuning.so}
for it in 4 8 16 64; do for ln in 8 24 32 48; do
echo "iters=$it lanes=$ln" 
PTRT_POOL_MISS_ITERS=$it PTRT_POOL_MISS_LANES=$ln timeout -k 10 100 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(' bench',round(d['value']),round(d['ms_per_step'],2))" || exit 1
PTRT_KERNEL=pool PTRT_POOL_MISS_ITERS=$it PTRT_POOL_MISS_LANES=$ln timeout -k 10 60 python tools/sky_only.py 2>/dev/null | tail -1 || exit 1
done; done
