#!/bin/bash
# the PTRT_* overrides are read by the tuning build only (make -C path-tracer_amd/csrc tuning)
export PTRT_LIB=${PTRT_LIB:-$PWD/path-tracer_amd/libptrt_tuning.so}
# GPU box: the pooled kernel's scheduling knobs on the 10 M-triangle field (4K, 128 frames per launch).
run() { echo -n "$* -> "; env "$@" timeout -k 10 280 python bench.py --workload field_4k_d8_spp1024 --frames 128 --steps 2 --warmup 1 --no-cpu-baseline --sky-size 2048 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%.0f Mrays/s %.2f ms" % (d["value"], d["ms_per_step"]))' || exit 1; }
run X=0
for v in 4 16; do run PTRT_POOL_STEPS=$v; done
for v in 8 32; do run PTRT_POOL_TOPUP=$v; done
for v in 16 48; do run PTRT_ITEMS_PER_SLOT=$v; done
for v in 16 64; do run PTRT_POOL_START=$v; done
run PTRT_POOL_DENSE=32
