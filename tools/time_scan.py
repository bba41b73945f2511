"""GPU box: single frames over a span of time values; prints the frames whose time or work stands out."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkgload
pt = _pkgload.load()
scenes = importlib.import_module("path_tracer_amd.scenes")
n, lo, hi = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
scene = scenes.bunny(n=n)
r = pt.Renderer(scene, 1920, 1080, settings=pt.Settings(1, 8, 1e-4, 1000.0), flags=pt.PT_FLAG_DETAILED_STATS | pt.PT_FLAG_TILE_KERNEL)
r.set_sky_cube(scenes.sky_cube_direct(64))
r.render(1); r.sync()
prev = r.stats()
rows = []
for t0 in range(lo, hi):
    t = time.perf_counter()
    r.render(t0); r.sync()
    dt = (time.perf_counter() - t) * 1e3
    s = r.stats()
    rows.append((t0, dt, s["rays"] - prev["rays"], s["interior"] - prev["interior"], s["tri_tests"] - prev["tri_tests"]))
    prev = s
a = np.array(rows, dtype=np.float64)
med = np.median(a, axis=0)
print(f"{len(a)} frames, median: {med[1]:.3f} ms, {med[2]:.0f} rays, {med[3]:.0f} node visits, {med[4]:.0f} triangle tests")
for row in a:
    if row[1] > 2 * med[1] or row[3] > 1.2 * med[3] or row[2] > 1.05 * med[2]:
        print(f"time {int(row[0])}: {row[1]:.3f} ms, {row[2]:.0f} rays, {row[3]:.0f} node visits, {row[4]:.0f} triangle tests")
r.close()
