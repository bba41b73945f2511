"""GPU box, tuning build (PTRT_LIB=path-tracer_amd/libptrt_tuning.so): steps of the second tree's walk for camera rays
towards the bench mesh, four children to a record against two (PTRT_NO_QUADS=1), and the time of the batch."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkgload
pt = _pkgload.load()
scenes = importlib.import_module("path_tracer_amd.scenes")
scene = scenes.bunny()
r = pt.Renderer(scene, 64, 64, settings=pt.Settings(1, 8, 1e-4, 1000.0))
rng = np.random.RandomState(3)
n = 2000000
o = np.tile(np.array(scene.camera.origin, np.float32), (n, 1))
d = (np.array([0.0, 0.5, -1.0], np.float32) + rng.normal(0, 0.55, (n, 3)).astype(np.float32) - o).astype(np.float32)
d /= np.linalg.norm(d, axis=1, keepdims=True)  # (the second tree's reach is scaled by the direction's length where that exceeds one)
r.trace_rays_walked(o[:1000], d[:1000])
t0 = time.perf_counter()
out = r.trace_rays_walked(o, d)
dt = time.perf_counter() - t0
walked = out[5]
steps = walked >> 2
print(f"quads {'off' if os.environ.get('PTRT_NO_QUADS') else 'on'}: trees {np.bincount(walked & 3, minlength=3)}, steps mean {steps[steps > 0].mean():.2f} max {steps.max()}, hit share {(out[0] != 0xFFFFFFFF).mean():.3f}, call {dt * 1e3:.1f} ms")
r.close()
