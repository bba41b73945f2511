#!/bin/bash
# the PTRT_* overrides are read by the tuning build only (make -C path-tracer_amd/csrc tuning)
export PTRT_LIB=${PTRT_LIB:-$PWD/path-tracer_amd/libptrt_tuning.so}
run() { echo -n "$* -> "; env "$@" timeout -k 10 120 python tools/kernel_choice.py 4 8 12 16 | sed 's/.*triangles= *\([0-9]*\).*per-tile *\([0-9]*\) Mrays.*/\1:\2/' | tr '\n' ' '; echo; }
export FRAMES=48 SPP=1
run X=0
run X=0
run PTRT_NO_PLAN=1
run PTRT_REFILL_LANES=24
run PTRT_ITEMS_PER_SLOT=6
run PTRT_ALWAYS_PLAN=1
run SPP=2 FRAMES=24
