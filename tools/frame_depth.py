"""GPU box: time of one 1-spp 1080p dispatch of the bench scene against path depth (is a single frame bound by the
longest path's chain of dependent traversal steps?).  PTRT_KERNEL=pool|tile forces a megakernel; REBUILD=1 walks the
library's own tree (PT_FLAG_REBUILD_BVH); REFWALK=1 walks the reference tree only (PT_FLAG_REFERENCE_WALK)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkgload
pt = _pkgload.load()
scenes = importlib.import_module("path_tracer_amd.scenes")
scene = scenes.bunny()
flags = (pt.PT_FLAG_REBUILD_BVH if os.environ.get("REBUILD") == "1" else 0) | (pt.PT_FLAG_REFERENCE_WALK if os.environ.get("REFWALK") == "1" else 0)
for sky in (int(os.environ.get("SKY", "4096")),):
    pano = scenes.sky_equirect(min(4096, sky * 2), min(2048, sky))
    for depth in (1, 2, 4, 8, 16, 32):
        r = pt.Renderer(scene, 1920, 1080, settings=pt.Settings(1, depth, 1e-4, 1000.0), flags=flags)
        r.set_sky_equirect(pano, sky)
        r.render(1); r.sync()
        rays0 = r.stats()["rays"]
        t0 = time.perf_counter()
        for k in range(32):
            r.render(100 + k)
        r.sync()
        dt = (time.perf_counter() - t0) / 32
        rays = (r.stats()["rays"] - rays0) / 32
        print(f"sky {sky:4d} depth {depth:2d}: {dt * 1e3:6.3f} ms per frame, {rays / 1e6:5.2f} Mrays, {rays / dt / 1e6:7.0f} Mrays/s", flush=True)
        r.close()
