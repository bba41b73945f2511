"""Pool-vs-classic on a sphere-only scene: which pixels differ and what they hit (GPU)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkgload
pt = _pkgload.load()
scenes = importlib.import_module("path_tracer_amd.scenes")
scene = scenes.three_balls()
scene.planes = [] if hasattr(scene, "planes") else None
w, h = 64, 40
st = pt.Settings(int(os.environ.get("DBG_SPP", "1")), int(os.environ.get("DBG_DEPTH", "4")), 1e-4, 1000.0)
sky = np.ones((6, 2, 2, 4), np.float32)
out = {}
for pool in ("0", "1"):
    os.environ["PTRT_POOL"] = pool
    r = pt.Renderer(scene, w, h, settings=st, flags=pt.PT_FLAG_AOV)
    r.set_sky_cube(sky)
    r.render(7)
    prim, t = r.read_aov()
    out[pool] = (r.read_sum().copy(), r.stats()["rays"], prim.copy())
    r.close()
a, ra, pa = out["0"]; b, rb, pb = out["1"]
print("rays", ra, rb, "prim equal", np.array_equal(pa, pb))
diff = np.argwhere((a.view(np.uint32) != b.view(np.uint32)).any(axis=2))
print("differing pixels:", len(diff), "of", w * h)
mats = {i: s.material.kind if hasattr(s.material, "kind") else s.material for i, s in enumerate(scene.spheres)}
print("materials", mats)
import collections
print("prims of differing pixels", collections.Counter(int(pa[y, x]) & 0xFFFF for y, x in diff))
print("prims of all pixels", collections.Counter(int(v) & 0xFFFF for v in pa.ravel()))
for y, x in diff[:10]:
    print(y, x, hex(pa[y, x]), a[y, x], b[y, x])
