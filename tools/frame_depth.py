"""GPU box: time of one 1-spp 1080p dispatch of the bench scene against path depth and sky size (is a single frame
bound by the longest path's chain of dependent fetches?)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkgload
pt = _pkgload.load()
scenes = importlib.import_module("path_tracer_amd.scenes")
scene = scenes.bunny()
for sky in (64, 4096):
    for depth in (1, 2, 4, 8, 16):
        r = pt.Renderer(scene, 1920, 1080, settings=pt.Settings(1, depth, 1e-4, 1000.0))
        r.set_sky_equirect(scenes.sky_equirect(min(4096, sky * 2), min(2048, sky)), sky)
        r.render(1); r.sync()
        rays0 = r.stats()["rays"]
        t0 = time.perf_counter()
        for k in range(32):
            r.render(100 + k)
        r.sync()
        dt = (time.perf_counter() - t0) / 32
        rays = (r.stats()["rays"] - rays0) / 32
        print(f"sky {sky:4d} depth {depth:2d}: {dt * 1e3:6.3f} ms per frame, {rays / 1e6:5.2f} Mrays, {rays / dt / 1e6:7.0f} Mrays/s")
        r.close()
