"""GPU box: time of each of a series of identical launches (plan, plan, then reuse), per-tile kernel, small mesh."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkgload
pt = _pkgload.load()
scenes = importlib.import_module("path_tracer_amd.scenes")
n, frames = int(sys.argv[1]), int(sys.argv[2])
flag = pt.PT_FLAG_POOL_KERNEL if len(sys.argv) > 3 and sys.argv[3] == "pool" else pt.PT_FLAG_TILE_KERNEL
scene = scenes.bunny(n=n)
if os.environ.get("PRIOR"):  # another context first, as a comparison script would have
    q = pt.Renderer(scene, 1920, 1080, settings=pt.Settings(1, 8, 1e-4, 1000.0), flags=pt.PT_FLAG_POOL_KERNEL)
    q.set_sky_cube(scenes.sky_cube_direct(64))
    for k in range(4):
        q.render_frames(1 + k * frames, 1, frames)
    q.sync(); q.close()
r = pt.Renderer(scene, 1920, 1080, settings=pt.Settings(int(os.environ.get("SPP", "1")), 8, 1e-4, 1000.0), flags=flag)
r.set_sky_cube(scenes.sky_cube_direct(64))
out = []
for k in range(10):
    if os.environ.get("STATS"):
        r.stats()
    t0 = time.perf_counter()
    r.render_frames(int(os.environ.get('BASE', '1')) + (0 if os.environ.get('SAME_TIME') else k) * frames, 1, frames); r.sync()
    out.append((time.perf_counter() - t0) * 1e3)
print(f"{len(scene.triangles)} triangles, {frames} frames:", " ".join(f"{v:.2f}" for v in out), "ms")
r.close()
