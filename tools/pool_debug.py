"""A/B the pooled kernel against the classic one on the golden scene (GPU): per-pixel sums, ray counts."""
import importlib.util, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkgload
pt = _pkgload.load()
spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
scene, sky, st, w, h, times = mod.golden_inputs()
out = {}
flags = pt.PT_FLAG_AOV | (pt.PT_FLAG_DETAILED_STATS if os.environ.get("DBG_STATS") else 0)
nt = int(os.environ.get("DBG_TIMES", "1"))
for pool in ("0", "1"):
    os.environ["PTRT_POOL"] = pool
    r = pt.Renderer(scene, w, h, settings=st, flags=flags)
    r.set_sky_cube(sky)
    for t in times[:nt]:
        r.render(t)
        print(pool, t, r.stats()["rays"])
    out[pool] = (r.read_sum().copy(), r.stats())
    r.close()
a, sa = out["0"]; b, sb = out["1"]
print("classic", sa); print("pool   ", sb)
diff = np.argwhere((a.view(np.uint32) != b.view(np.uint32)).any(axis=2))
print("differing pixels:", len(diff), "of", w * h)
for y, x in diff[:20]:
    print(y, x, a[y, x], b[y, x])
