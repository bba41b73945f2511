"""GPU box: time of single-dispatch renders (pt_render, 1 spp) and of short fused launches, bench scene."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkgload
pt = _pkgload.load()
scenes = importlib.import_module("path_tracer_amd.scenes")
scene = scenes.bunny()
r = pt.Renderer(scene, 1920, 1080, settings=pt.Settings(1, 8, 1e-4, 1000.0))
r.set_sky_equirect(scenes.sky_equirect(4096, 2048), 4096)
for n in (1, 2, 4, 8, 15, 16, 32):
    r.render_frames(1, 1, n); r.sync()
    rays0 = r.stats()["rays"]
    reps = max(1, 64 // n)
    t0 = time.perf_counter()
    for k in range(reps):
        r.render_frames(100 + k * n, 1, n)
    r.sync()
    dt = (time.perf_counter() - t0) / reps
    rays = (r.stats()["rays"] - rays0) / reps
    print(f"{n:3d} frames per launch: {dt * 1e3:7.3f} ms per launch, {dt * 1e3 / n:6.3f} ms per frame, {rays / dt / 1e6:8.0f} Mrays/s")
r.close()
