"""Joins tools/step_latency.py's list of launches with the ray_query_kernel durations of the rocprofv3 kernel trace of
the same run: nanoseconds per traversal step of the launch's longest walk."""
import csv, glob, json, os, sys
root = sys.argv[1]
launches = json.load(open(os.path.join(root, "step_latency_launches.json")))
rows = []
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if "ray_query_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
assert len(rows) == len(launches), (len(rows), len(launches))
for l, r in zip(launches, rows):
    ns = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    print(f"{l['what']:32s} rays {l['rays']:6d}  longest walk {l['steps_max']:4d} steps  kernel {ns / 1e3:8.1f} us  {ns / max(l['steps_max'], 1):7.1f} ns per step of the longest walk")
