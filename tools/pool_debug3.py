"""Pool-vs-classic on cornell 96x64 spp1 depth32, separate render() calls: where do sums differ (GPU)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkgload
pt = _pkgload.load()
scenes = importlib.import_module("path_tracer_amd.scenes")
scene = scenes.cornell()
w, h = 96, 64
st = pt.Settings(1, 32, 1e-4, 1000.0)
sky = np.ones((6, 2, 2, 4), np.float32)
nf = int(os.environ.get("DBG_FRAMES", "3"))
out = {}
for pool in ("0", "1"):
    os.environ["PTRT_POOL"] = pool
    r = pt.Renderer(scene, w, h, settings=st, flags=pt.PT_FLAG_AOV)
    r.set_sky_cube(sky)
    per = []
    for t in range(nf):
        r.render(10 + t)
        per.append((r.read_sum().copy(), r.stats()["rays"]))
    out[pool] = per
    r.close()
for t in range(nf):
    a, ra = out["0"][t]; b, rb = out["1"][t]
    diff = np.argwhere((a.view(np.uint32) != b.view(np.uint32)).any(axis=2))
    print(f"after frame {t}: rays {ra} {rb}; differing pixels {len(diff)}")
    if len(diff):
        ys, xs = diff[:, 0], diff[:, 1]
        print("   tiles (ty,tx):", sorted(set(zip((ys // 8).tolist(), (xs // 8).tolist()))))
        for y, x in diff[:6]:
            print("   ", y, x, a[y, x], b[y, x])
