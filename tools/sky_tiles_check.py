import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkgload
pt = _pkgload.load()
scenes = importlib.import_module("path_tracer_amd.scenes")
import numpy as np
r = pt.Renderer(scenes.bunny(), 1920, 1080, settings=pt.Settings(1, 8, 1e-4, 1000.0))
r.set_sky_equirect(scenes.sky_equirect(4096, 2048), 1024)
for rep in range(3):
    r.render_frames(1 + 256 * rep, 1, 256)
r.sync()
f = r.read_sky_tiles()
print("flagged fraction", f.mean(), "rows flagged:", f.mean(axis=1)[::8])
t0 = time.perf_counter()
for rep in range(5):
    r.render_frames(5000 + 256 * rep, 1, 256)
r.sync()
print("ms per launch", (time.perf_counter() - t0) / 5 * 1e3, os.environ.get("PTRT_NO_SKY_TILES"))
