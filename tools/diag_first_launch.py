"""GPU box: the first launch of a context (no cost measurement yet), whole or with a short planned stage first."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkgload
pt = _pkgload.load()
scenes = importlib.import_module("path_tracer_amd.scenes")
sky = scenes.sky_cube_direct(64)
for n, frames in ((8, 192), (16, 192), (76, 256), (76, 64)):
    scene = scenes.bunny(n=n)
    for base in (1, 1345, 100001):
        for flag, name in ((pt.PT_FLAG_TILE_KERNEL, "per-tile"), (pt.PT_FLAG_POOL_KERNEL, "pooled")):
            res = []
            for stage in (0, 24):
                r = pt.Renderer(scene, 1920, 1080, settings=pt.Settings(1, 8, 1e-4, 1000.0), flags=flag)
                r.set_sky_cube(sky)
                r.sync()
                t0 = time.perf_counter()
                if stage:
                    r.render_frames(base, 1, stage)
                r.render_frames(base + stage, 1, frames - stage); r.sync()
                res.append((time.perf_counter() - t0) * 1e3)
                r.close()
            print(f"{len(scene.triangles):6d} triangles, {frames} frames from time {base:6d}, {name:8s}: whole {res[0]:6.2f} ms, staged {res[1]:6.2f} ms", flush=True)
