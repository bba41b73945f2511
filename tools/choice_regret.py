"""GPU box: the library's own kernel choice against both kernels forced, over mesh sizes, launch lengths and samples
per frame (1080p, depth 8, best of three launches each): prints auto / best-forced."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkgload
pt = _pkgload.load()
scenes = importlib.import_module("path_tracer_amd.scenes")
sky = scenes.sky_cube_direct(64)
os.environ.pop("PTRT_KERNEL", None)


def rate(scene, st, frames, flags):
    r = pt.Renderer(scene, 1920, 1080, settings=st, flags=flags)
    r.set_sky_cube(sky)
    r.render_frames(1, 1, frames); r.sync()
    best = 0.0
    for rep in range(3):
        rays0 = r.stats()["rays"]
        t0 = time.perf_counter()
        r.render_frames(1000 + rep * frames, 1, frames); r.sync()
        best = max(best, (r.stats()["rays"] - rays0) / (time.perf_counter() - t0) / 1e6)
    r.close()
    return best


worst = 1.0
for n in (8, 16, 27, 48, 76):
    scene = scenes.bunny(n=n)
    for spp, frames in ((1, 16), (1, 48), (1, 64), (1, 192), (2, 24), (2, 96), (4, 16), (4, 48), (16, 4), (64, 1)):
        st = pt.Settings(spp, 8, 1e-4, 1000.0)
        a, p, t = (rate(scene, st, frames, f) for f in (0, pt.PT_FLAG_POOL_KERNEL, pt.PT_FLAG_TILE_KERNEL))
        worst = min(worst, a / max(p, t))
        print(f"{len(scene.triangles):6d} triangles, {spp:2d} spp x {frames:3d} frames: auto {a:7.0f} pooled {p:7.0f} per-tile {t:7.0f}  auto/best {a / max(p, t):.2f}", flush=True)
print(f"worst auto/best {worst:.2f}")
