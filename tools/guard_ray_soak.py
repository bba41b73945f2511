"""GPU box: explicit rays aimed at the corner cases of the second tree's walk (vertices, edges, grazing angles, origins on
the surface, degenerate direction components: tests/test_gpu_guard.py::_aimed_rays) through pt_trace_rays with the second
tree and with PT_FLAG_REFERENCE_WALK, compared bit for bit -- ids, t, point, normal, facing.  (The reference walk of
pt_trace_rays is itself compared with the CPU oracle by the test suite.)  Usage: python tools/guard_ray_soak.py [millions of rays]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _pkgload
pt = _pkgload.load()
scenes = importlib.import_module("path_tracer_amd.scenes")
from test_gpu_guard import _aimed_rays
want = int(sys.argv[1]) if len(sys.argv) > 1 else 20
st = pt.Settings(1, 4, 1e-4, 1000.0)
total, tally = 0, np.zeros(3, np.int64)
seed = 0
while total < want * 1000000:
    n_mesh = (32, 48, 76)[seed % 3]
    base = scenes.bunny(n=n_mesh)
    spheres = list(base.spheres) + [pt.Sphere((0.9, 0.6, -0.2), 0.35, (1, 1, 1), pt.Material.Dielectric)]
    scene = pt.Scene(spheres, base.triangles, base.camera)
    a = pt.Renderer(scene, 16, 16, settings=st)
    b = pt.Renderer(scene, 16, 16, settings=st, flags=pt.PT_FLAG_REFERENCE_WALK)
    assert a.stats()["guard_nodes"] > 0
    for rep in range(4):
        o, d, kind = _aimed_rays(scene.triangles, np.random.RandomState(1000 + 17 * seed + rep), 2000000)
        ga = a.trace_rays_walked(o, d)
        gb = b.trace_rays_walked(o, d)
        assert not gb[5].any()
        for k, name in enumerate(("primitive id", "t", "point", "normal", "facing")):
            x, y = ga[k], gb[k]
            if x.dtype == np.float32:
                x, y = x.view(np.uint32), y.view(np.uint32)
            bad = (x != y)
            if bad.any():
                rows = np.nonzero(bad.reshape(len(o), -1).any(1))[0]
                print(f"MISMATCH in {name}: {len(rows)} rays, first {rows[:5]}, kinds {kind[rows[:5]]}, trees {ga[5][rows[:5]]}, mesh n={n_mesh}, seed {seed} rep {rep}")
                sys.exit(1)
        tally += np.bincount(ga[5], minlength=3)[:3]
        total += len(o)
    a.close(); b.close()
    seed += 1
    print(f"{total / 1e6:.0f} M rays: reference tree only {tally[0]}, second tree's hit taken {tally[1]}, walked both {tally[2]}", flush=True)
print("ok")
