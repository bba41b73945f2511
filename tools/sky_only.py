"""GPU box: a scene where every ray misses (one tiny sphere behind the camera): cost of the miss path alone."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkgload
pt = _pkgload.load()
scenes = importlib.import_module("path_tracer_amd.scenes")
scene = pt.Scene([pt.Sphere((0.0, 0.0, 50.0), 0.1, (0.8, 0.8, 0.0), pt.Material.Diffuse)], scenes.cornell().triangles[:0], scenes.default_camera())
r = pt.Renderer(scene, 1920, 1080, settings=pt.Settings(1, 8, 1e-4, 1000.0))
r.set_sky_equirect(scenes.sky_equirect(4096, 2048), 4096)
for rep in range(3):
    r.render_frames(1 + 256 * rep, 1, 256)
r.sync()
rays0 = r.stats()["rays"]
t0 = time.perf_counter()
r.render_frames(5000, 1, 256); r.sync()
dt = time.perf_counter() - t0
print(os.environ.get("PTRT_KERNEL"), f"{(r.stats()['rays'] - rays0) / dt / 1e6:.0f} Mrays/s, {dt * 1e3:.2f} ms")
