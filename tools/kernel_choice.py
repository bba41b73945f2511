"""GPU box: pooled vs per-tile kernel on meshes of growing size (bunny(n), 1080p, depth 8, FRAMES x 1 spp per launch,
FRAMES from the environment, default 64)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkgload
pt = _pkgload.load()
scenes = importlib.import_module("path_tracer_amd.scenes")
sky = scenes.sky_cube_direct(64)
st = pt.Settings(int(os.environ.get("SPP", "1")), 8, 1e-4, 1000.0)
frames = int(os.environ.get("FRAMES", "64"))
for n in [int(a) for a in sys.argv[1:]] or [2, 4, 8, 16, 32, 76]:
    scene = scenes.bunny(n=n)
    res = []
    for flags in (pt.PT_FLAG_POOL_KERNEL, pt.PT_FLAG_TILE_KERNEL):
        r = pt.Renderer(scene, 1920, 1080, settings=st, flags=flags)
        r.set_sky_cube(sky)
        r.render_frames(1, 1, frames); r.sync()
        best = 0.0
        for rep in range(3):  # best of three launches: a single one varies by 10 % and more
            rays0 = r.stats()["rays"]
            t0 = time.perf_counter()
            r.render_frames(1000 + rep * frames, 1, frames); r.sync()
            dt = time.perf_counter() - t0
            best = max(best, (r.stats()["rays"] - rays0) / dt / 1e6)
        res.append(best)
        depth = r.stats()["bvh_depth"]
        r.close()
    print(f"n={n:3d} triangles={len(scene.triangles):7d} bvh_depth={depth:2d}: pooled {res[0]:8.0f} Mrays/s, per-tile {res[1]:8.0f} Mrays/s, ratio {res[0]/res[1]:.2f}")
