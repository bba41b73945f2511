#!/bin/bash
# Runs on the GPU box: FETCH_SIZE and raw TCC read-request counters for the two calibration patterns.
OUT=$GRAFT_REPO_ROOT/gpurun_out/fetch_calib; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for mode in 0 1; do
  for pass in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum"; do
    name=m${mode}_$(echo $pass | cut -d' ' -f1)
    timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d $OUT/$name -- $GRAFT_REPO_ROOT/tools/ubench/fetch_calib $mode > $OUT/$name.log 2>&1 || echo "failed $name"
  done
done
python3 - <<'PY'
import csv, glob, os
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/fetch_calib"
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "read" in r["Kernel_Name"]:
            print(os.path.relpath(f, root).split("/")[0], r["Kernel_Name"][:20], r["Counter_Name"], r["Counter_Value"])
PY
