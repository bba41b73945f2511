"""GPU box: launches of 1 to 32 frames of the bench scene (1080p, depth 8, 1 spp) with each megakernel forced and
with the library's own choice: ms per launch.  Usage: short_launch_kernels.py [mesh subdivision [regroup]]
(regroup: with PT_FLAG_REGROUP_SHORT)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkgload
pt = _pkgload.load()
scenes = importlib.import_module("path_tracer_amd.scenes")
n_arg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
scene = scenes.bunny(n=n_arg) if n_arg else scenes.bunny()
sky = scenes.sky_equirect(4096, 2048)
extra = pt.PT_FLAG_REGROUP_SHORT if len(sys.argv) > 2 and sys.argv[2] == "regroup" else 0
print(f"{len(scene.triangles)} triangles" + (", PT_FLAG_REGROUP_SHORT" if extra else ""))
for n in (1, 2, 4, 8, 12, 15, 16, 24, 32):
    row = []
    for flags in (0, pt.PT_FLAG_POOL_KERNEL, pt.PT_FLAG_TILE_KERNEL):
        r = pt.Renderer(scene, 1920, 1080, settings=pt.Settings(1, 8, 1e-4, 1000.0), flags=flags | extra)
        r.set_sky_equirect(sky, 4096)
        for k in range(3):
            r.render_frames(1 + k * n, 1, n)
        r.sync()
        reps = max(2, 48 // n)
        t0 = time.perf_counter()
        for k in range(reps):
            r.render_frames(100 + k * n, 1, n)
        r.sync()
        row.append((time.perf_counter() - t0) / reps * 1e3)
        r.close()
    print(f"{n:3d} frames: auto {row[0]:7.3f} ms  pooled {row[1]:7.3f} ms  per-tile {row[2]:7.3f} ms  pooled/per-tile time {row[1] / row[2]:.2f}", flush=True)
