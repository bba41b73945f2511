"""Condenses rocprofv3 csv output (kernel stats + pmc passes) into a short text summary."""
import csv
import glob
import os
import sys
import re
from collections import defaultdict


def is_mega(name):
    return "trace_kernel" in name or "pool_kernel" in name


def is_planned(name):
    """Planned launches (item table + partial sums): the kernels' last template argument, DIRECT, is false."""
    m = re.search(r"_kernel<([^>]*)>", name)
    return bool(m) and m.group(1).split(",")[-1].strip() == "false"


root = sys.argv[1]
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats:", os.path.relpath(f, root))
    for row in csv.DictReader(open(f)):
        print("  %-60s calls=%s total_ns=%s avg_ns=%s pct=%s" % (row.get("Name", "")[:60], row.get("Calls"), row.get("TotalDurationNs"), row.get("AverageNs"), row.get("Percentage")))
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if is_mega(r.get("Kernel_Name", ""))]
    if rows:
        print("== megakernel dispatches (ms): " + ", ".join("%s %.2f" % ("planned" if is_planned(r["Kernel_Name"]) else "direct", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6) for r in rows))
        r = rows[-1]
        print("== last megakernel dispatch: VGPR=%s SGPR=%s LDS=%s scratch=%s grid=%s wg=%s dur_ns=%d" % (
            r.get("VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Scratch_Size"), r.get("Grid_Size"), r.get("Workgroup_Size"),
            int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
print("== pmc (per megakernel dispatch, planned launches only, mean over dispatches)")
for f in sorted(glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    acc, n = defaultdict(float), defaultdict(int)
    for row in csv.DictReader(open(f)):
        kn = row.get("Kernel_Name", "")
        if not (is_mega(kn) and is_planned(kn) and "_kernel<false" in kn):  # planned launches of the plain (no detailed stats) build
            continue
        acc[row["Counter_Name"]] += float(row["Counter_Value"])
        n[row["Counter_Name"]] += 1
    for k in acc:
        print("  %-32s %.6g  (n=%d)" % (k, acc[k] / n[k], n[k]))

# HBM traffic of the planned megakernel launches (the largest dispatches), corrected as measured by
# tools/ubench/fetch_calib: FETCH_SIZE tallies every TCC->fabric read request as 64 B although a
# coalesced stream moves 128 B per request (factor 2); WRITE_SIZE is exact.
import json
vals = {}
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    xs = []
    for f in glob.glob(os.path.join(root, "pmc_" + name, "**", "*counter_collection.csv"), recursive=True):
        xs += [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if is_mega(r.get("Kernel_Name", "")) and is_planned(r["Kernel_Name"]) and "_kernel<false" in r["Kernel_Name"] and r["Counter_Name"] == name]
    big = [x for x in xs if x > 0.5 * max(xs)] if xs else []
    vals[name] = sum(big) / len(big) if big else None
# The launch's other kernels: the sky-only tiles' loop (it runs beside the megakernel) and the addition of the partial sums.
side = {}
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    for kern in ("sky_tiles_kernel", "accumulate_kernel"):
        xs = []
        for f in glob.glob(os.path.join(root, "pmc_" + name, "**", "*counter_collection.csv"), recursive=True):
            xs += [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if kern in r.get("Kernel_Name", "") and r["Counter_Name"] == name]
        big = [x for x in xs if x > 0.5 * max(xs)] if xs and max(xs) > 0 else []
        side[(name, kern)] = sum(big) / len(big) if big else 0.0
print("== per launch, KB:", {f"{k[1]} {k[0]}": round(v) for k, v in side.items()}, "megakernel", vals)
if vals["FETCH_SIZE"] is not None and vals["WRITE_SIZE"] is not None:
    for name in ("FETCH_SIZE", "WRITE_SIZE"):
        vals[name] += side[(name, "sky_tiles_kernel")] + side[(name, "accumulate_kernel")]
    args = sys.argv[2:]
    frames = int(args[args.index("--frames") + 1]) if "--frames" in args else 256
    workload = args[args.index("--workload") + 1] if "--workload" in args else "bunny_1080p_d8_spp256"
    rec = {"workload": workload, "frames": frames, "bvh": "sah" if "--rebuild-bvh" in args else "reference-walk-only" if "--reference-walk" in args else "reference",
           "fetch_size_kb": vals["FETCH_SIZE"], "write_size_kb": vals["WRITE_SIZE"],
           "hbm_bytes_per_launch": 2.0 * vals["FETCH_SIZE"] * 1024.0 + vals["WRITE_SIZE"] * 1024.0,
           "kernels": "megakernel + sky_tiles_kernel + accumulate_kernel of one planned launch",
           "correction": "2 x FETCH_SIZE (128-B read requests tallied at 64 B on gfx950) + WRITE_SIZE"}
    json.dump(rec, open(os.path.join(root, "pmc_traffic.json"), "w"))
    print("== traffic:", rec)
