#!/bin/bash
# GPU box: two SQ counter passes over a short bench run; prints per-dispatch means for the main kernel.
# Usage: [PTRT_KERNEL=pool|tile] bash tools/pmc_quick.sh <tag>
set -o pipefail
TAG=${1:-q}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
ARGS="--frames 64 --steps 2 --warmup 1 --no-cpu-baseline"
i=0
for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" \
            "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d $OUT/p$i -- python3 $B $ARGS > $OUT/p$i.log 2>&1 || { echo "pmc pass failed: $pass"; tail -3 $OUT/p$i.log; exit 1; }
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections
res = collections.OrderedDict()
for f in sorted(glob.glob(sys.argv[1] + "/p*/**/*_counter_collection.csv", recursive=True)):
    by = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "pool_kernel" in k or "trace_kernel" in k:
            d = by[r["Dispatch_Id"]]
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    if not by: continue
    first = list(next(iter(by.values())).keys())[0]
    top = sorted(by.values(), key=lambda d: -d.get(first, 0))[:2]
    for k in top[0]:
        res[k] = sum(d[k] for d in top) / len(top)
for k, v in res.items():
    print(f"  {k:28s} {v:.4g}")
if "SQ_ACTIVE_INST_VALU" in res and "GRBM_GUI_ACTIVE" in res:
    print("  VALU busy per SIMD: %.1f%%" % (100 * res["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / res["GRBM_GUI_ACTIVE"]))
PY
