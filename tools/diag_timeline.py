"""Diagnostic (GPU box): per-wave timeline of one trace launch from the PT_DIAG build.
Run: PTRT_LIB=path-tracer_amd/libptrt_diag.so python tools/diag_timeline.py [frames]"""
import ctypes as C
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkgload

pt = _pkgload.load()
scenes = importlib.import_module("path_tracer_amd.scenes")
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 16
sky = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
scene, w, h, st, _ = scenes.CONFIGS["bunny_1080p_d8_spp256"][0](), 1920, 1080, pt.Settings(1, 8, 1e-4, 1000.0), 0
r = pt.Renderer(scene, w, h, settings=st)
pw = min(4096, sky)
r.set_sky_equirect(scenes.sky_equirect(pw, pw // 2), sky)
r.render_frames(1, 1, frames); r.sync()
r.reset_ready_samples()
r.render_frames(1, 1, frames); r.sync()
nw = ((w + 15) // 16) * ((h + 15) // 16) * 4
buf = np.zeros((nw, 4), np.uint64)
L = pt.lib()
L.pt_diag_read.argtypes = [C.c_void_p, C.c_size_t]
assert L.pt_diag_read(buf.ctypes.data_as(C.c_void_p), buf.nbytes) == 0
t0, t1, hw, misc = buf[:, 0].astype(np.int64), buf[:, 1].astype(np.int64), buf[:, 2], buf[:, 3]
iters, rays = (misc & 0xFFFFFFFF).astype(np.int64), (misc >> 32).astype(np.int64)
base = t0.min()
dur = (t1 - t0) / 100.0  # us (100 MHz)
total = (t1.max() - base) / 100.0
print(f"frames={frames} waves={nw} kernel span={total:.1f} us rays={rays.sum()} -> {rays.sum()/total:.1f} Mrays/s")
print("wave duration us: min %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % tuple(np.percentile(dur, [0, 50, 90, 99, 100])))
print("iters per wave: min %d p50 %d p90 %d max %d ; mean rays/iter %.1f" % (iters.min(), np.median(iters), np.percentile(iters, 90), iters.max(), rays.sum() / iters.sum()))
print("us per iter: p50 %.2f p90 %.2f" % (np.median(dur / np.maximum(iters, 1)), np.percentile(dur / np.maximum(iters, 1), 90)))
# concurrency over time
ev = np.concatenate([np.stack([t0, np.ones_like(t0)], 1), np.stack([t1, -np.ones_like(t1)], 1)])
ev = ev[np.argsort(ev[:, 0], kind="stable")]
conc = np.cumsum(ev[:, 1])
ts = (ev[:, 0] - base) / 100.0
for frac in (0.05, 0.25, 0.5, 0.75, 0.9, 0.95, 0.99):
    i = np.searchsorted(ts, frac * total)
    print(f"  t={frac*total:9.1f} us ({int(frac*100):2d}%): resident waves {int(conc[min(i, len(conc)-1)])}")
avg_conc = dur.sum() / total
print(f"average resident waves {avg_conc:.0f} = {avg_conc/256:.1f} per CU")
# rows of tiles: time by tile row
bx = (w + 15) // 16
row = (np.arange(nw) // 4) // bx
for rr in range(0, (h + 15) // 16, 6):
    m = row == rr
    print(f"  tile row {rr:2d}: mean wave us {dur[m].mean():8.1f} iters {iters[m].mean():7.1f} rays/iter {rays[m].sum()/max(iters[m].sum(),1):5.1f} start {((t0[m].min()-base)/100):8.1f}")
