"""GPU box: per-frame work counts of a mesh scene against the shader's time value (does the cost move with the time?)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkgload
pt = _pkgload.load()
scenes = importlib.import_module("path_tracer_amd.scenes")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
scene = scenes.bunny(n=n)
r = pt.Renderer(scene, 1920, 1080, settings=pt.Settings(1, 8, 1e-4, 1000.0), flags=pt.PT_FLAG_DETAILED_STATS | pt.PT_FLAG_TILE_KERNEL)
r.set_sky_cube(scenes.sky_cube_direct(64))
prev = r.stats()
for t0 in (1, 500, 1000, 1200, 1345, 1400, 1500, 1700, 2000, 3000, 5000, 100000, 1 << 24, 1 << 31):
    t = time.perf_counter()
    r.render_frames(t0, 1, 8); r.sync()
    dt = time.perf_counter() - t
    s = r.stats()
    d = {k: s[k] - prev[k] for k in s if k not in ("frames", "bvh_depth")}
    prev = s
    print(f"time {t0:>10d}: {dt * 1e3:6.2f} ms, rays {d['rays'] / 8e6:.3f} M/frame, node visits/ray {d['interior'] / d['rays']:.2f}, "
          f"triangle tests/ray {d['tri_tests'] / d['rays']:.2f}, bounces/path {d['bounces'] / d['paths']:.3f}", flush=True)
r.close()
