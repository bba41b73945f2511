run() { echo -n "$* -> "; env "$@" timeout -k 10 120 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 tools/bench_val.py || exit 1; }
run X=0
for v in 4 6 12; do run PTRT_POOL_STEPS=$v; done
for v in 4 8 24; do run PTRT_POOL_TOPUP=$v; done
for v in 16 32 96; do run PTRT_POOL_START=$v; done
for v in 32 40 56; do run PTRT_POOL_MISS_LANES=$v; done
for v in 8 32; do run PTRT_POOL_MISS_ITERS=$v; done
for v in 36 52; do run PTRT_ITEMS_PER_SLOT=$v; done
for v in 8 32; do run PTRT_POOL_DENSE=$v; done
