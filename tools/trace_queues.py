"""Prints, from a rocprofv3 kernel trace (CSV), the hardware queue, start and duration of the megakernel and of the
sky-only tiles' loop in the last launches: do the two overlap, and are they on different queues?
Usage: python tools/trace_queues.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
keep = [r for r in rows if any(k in r["Kernel_Name"] for k in ("pool_kernel", "trace_kernel", "sky_tiles", "accumulate", "probe"))]
t0 = int(keep[0]["Start_Timestamp"]) if keep else 0
for r in keep[-16:]:
    name = next(k for k in ("pool_kernel", "trace_kernel", "sky_tiles_kernel", "accumulate_kernel", "stream_probe_kernel", "sky_tiles", "accumulate", "probe") if k in r["Kernel_Name"])
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"queue {r.get('Queue_Id', '?'):>3}  start {(s - t0) / 1e6:10.3f} ms  dur {(e - s) / 1e6:8.3f} ms  {name}")
