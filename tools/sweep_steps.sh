#!/bin/bash
export PTRT_LIB=${PTRT_LIB:-$PWD/path-tracer_amd/libptrt_tuning.so}
runb() { echo -n "bench $* -> "; env "$@" timeout -k 10 120 python bench.py --steps 6 --warmup 2 --no-cpu-baseline $EXTRA 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%.0f Mrays/s %.2f ms" % (d["value"], d["ms_per_step"]))' || exit 1; }
runf() { echo -n "field $* -> "; env "$@" timeout -k 10 280 python bench.py --workload field_4k_d8_spp1024 --frames 128 --steps 2 --warmup 1 --no-cpu-baseline --sky-size 2048 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%.0f Mrays/s %.2f ms" % (d["value"], d["ms_per_step"]))' || exit 1; }
EXTRA=""
for v in 8 12 16 24; do runb PTRT_POOL_STEPS=$v; done
EXTRA="--frames 128"
for v in 32 48 64; do runb PTRT_ITEMS_PER_SLOT=$v; done
runb PTRT_POOL_STEPS=16 PTRT_ITEMS_PER_SLOT=48
for v in 12 24; do runf PTRT_POOL_STEPS=$v; done
runf PTRT_POOL_STEPS=16 PTRT_ITEMS_PER_SLOT=48
runf PTRT_POOL_STEPS=16 PTRT_ITEMS_PER_SLOT=64
