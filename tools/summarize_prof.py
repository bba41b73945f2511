"""Condenses rocprofv3 csv output (kernel stats + pmc passes) into a short text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats:", os.path.relpath(f, root))
    for row in csv.DictReader(open(f)):
        print("  %-60s calls=%s total_ns=%s avg_ns=%s pct=%s" % (row.get("Name", "")[:60], row.get("Calls"), row.get("TotalDurationNs"), row.get("AverageNs"), row.get("Percentage")))
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "trace_kernel" in r.get("Kernel_Name", "")]
    if rows:
        r = rows[-1]
        print("== last trace_kernel dispatch: VGPR=%s SGPR=%s LDS=%s scratch=%s grid=%s wg=%s dur_ns=%d" % (
            r.get("VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Scratch_Size"), r.get("Grid_Size"), r.get("Workgroup_Size"),
            int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
print("== pmc (per trace_kernel dispatch, mean over dispatches)")
for f in sorted(glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    acc, n = defaultdict(float), defaultdict(int)
    for row in csv.DictReader(open(f)):
        if "trace_kernel" not in row.get("Kernel_Name", ""):
            continue
        acc[row["Counter_Name"]] += float(row["Counter_Value"])
        n[row["Counter_Name"]] += 1
    for k in acc:
        print("  %-32s %.6g  (n=%d)" % (k, acc[k] / n[k], n[k]))
