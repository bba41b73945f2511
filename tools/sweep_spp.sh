#!/bin/bash
# the PTRT_* overrides are read by the tuning build only (make -C path-tracer_amd/csrc tuning)
export PTRT_LIB=${PTRT_LIB:-$PWD/path-tracer_amd/libptrt_tuning.so}
# GPU box: the pooled kernel's scheduling knobs at several samples per frame (SPP, FRAMES from the environment).
export SPP=${SPP:-2} FRAMES=${FRAMES:-64}
run() { echo -n "$* -> "; env "$@" timeout -k 10 120 python tools/kernel_choice.py 76 | sed 's/.*pooled *\([0-9]*\) Mrays.*/\1/' || exit 1; }
run X=0
run X=0
for v in 16 48; do run PTRT_REFILL_LANES=$v; done
for v in 32 64; do run PTRT_POOL_SHADE_DENSE=$v; done
for v in 16 64; do run PTRT_POOL_START=$v; done
for v in 8 32; do run PTRT_POOL_TOPUP=$v; done
for v in 4 16; do run PTRT_POOL_STEPS=$v; done
for v in 16 64; do run PTRT_POOL_MISS_ITERS=$v; done
for v in 12 32 48; do run PTRT_ITEMS_PER_SLOT=$v; done
for v in 1 4; do run PTRT_MIN_CHUNK=$v; done
