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
L = pt.lib()
L.pt_diag_clear.argtypes = []
r.render_frames(1, 1, frames); r.sync()
r.reset_ready_samples()
r.render_frames(1, 1, frames); r.sync()
L.pt_diag_clear()
r.reset_ready_samples()
r.render_frames(1, 1, frames); r.sync()
import math
chunk = int(os.environ.get("PTRT_CHUNK_FRAMES", "0")) or (frames if frames < 16 else min(32, max(4, (frames + 7) // 8)))
nw = 1 << 18
buf = np.zeros((nw, 8), np.uint64)
L.pt_diag_read.argtypes = [C.c_void_p, C.c_size_t]
assert L.pt_diag_read(buf.ctypes.data_as(C.c_void_p), buf.nbytes) == 0
buf = buf[buf[:, 1] > 0]
nw = len(buf)
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
top = np.argsort(-dur)[:12]
print("longest waves: " + ", ".join(f"[wid {int(i)} start {(t0[i]-base)/100:.0f}us dur {dur[i]:.0f}us refills {int(iters[i])} steps {int(buf[i,7])} walk_cyc {int(buf[i,4])} refill_cyc {int(buf[i,5])} rays {int(rays[i])}]" for i in top[:6]))
late = np.argsort(-(t1 - base))[:6]
print("last to finish: " + ", ".join(f"[wid {int(i)} start {(t0[i]-base)/100:.0f}us dur {dur[i]:.0f}us]" for i in late))
marks = np.zeros((64, 8), np.uint64)
L.pt_diag_read_marks.argtypes = [C.c_void_p, C.c_size_t]
L.pt_diag_read_marks(marks.ctypes.data_as(C.c_void_p), marks.nbytes)
for wi in (0, 2, 8):
    ms = [(int(m) - int(base)) / 100.0 for m in marks[wi] if m]
    print(f"wave {wi}: time (us) at every 1024th loop trip: " + " ".join(f"{m:.0f}" for m in ms))
avg_conc = dur.sum() / total
print(f"average resident waves {avg_conc:.0f} = {avg_conc/256:.1f} per CU")
ta, tb, tc = buf[:, 4].astype(np.float64), buf[:, 5].astype(np.float64), buf[:, 6].astype(np.float64)
tot = ta.sum() + tb.sum() + tc.sum()
steps = buf[:, 7].astype(np.float64)
print("shares of in-loop wave time: walk steps %.1f%%  shade+new ray %.1f%%" % (100 * ta.sum() / tot, 100 * tb.sum() / tot))
print("per wave: refills %.1f, walk iterations %.1f; cycles per walk iteration %.0f, per refill %.0f; rays per refill %.1f" % (
    iters.mean(), steps.mean(), ta.sum() / max(steps.sum(), 1), tb.sum() / max(iters.sum(), 1), rays.sum() / max(iters.sum(), 1)))
dc = buf[:, 6]
walkers, leaf_lanes, leaf_trips = (dc >> np.uint64(40)).astype(np.float64), ((dc >> np.uint64(20)) & np.uint64(0xFFFFF)).astype(np.float64), (dc & np.uint64(0xFFFFF)).astype(np.float64)
print("walk trips: %.3g total, mean active walkers %.1f, trips with a leaf lane %.1f%%, mean leaf lanes in those %.1f" % (
    steps.sum(), walkers.sum() / steps.sum(), 100 * leaf_trips.sum() / steps.sum(), leaf_lanes.sum() / max(leaf_trips.sum(), 1)))
heavy = steps > np.percentile(steps, 90)
print("top-10%% waves by trips hold %.1f%% of trips; their mean active walkers %.1f" % (100 * steps[heavy].sum() / steps.sum(), walkers[heavy].sum() / steps[heavy].sum()))
pi = ta / np.maximum(iters, 1)
print("hit_scene cycles per wave-iteration: p10 %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f" % tuple(np.percentile(pi[iters > 0], [10, 50, 90, 99, 100])))
if not os.environ.get("PTRT_NO_ORDER"):
    sys.exit(0)
# rows of tiles: time by tile row
bx = (w + 15) // 16
row = (np.arange(nw) // 4) // bx
for rr in range(0, (h + 15) // 16, 3):
    m = row == rr
    print(f"  tile row {rr:2d}: mean wave us {dur[m].mean():8.1f} iters {iters[m].mean():7.1f} rays/iter {rays[m].sum()/max(iters[m].sum(),1):5.1f} hit_scene cyc/iter {ta[m].sum()/max(iters[m].sum(),1):8.0f} shade {tb[m].sum()/max(iters[m].sum(),1):6.0f} adv {tc[m].sum()/max(iters[m].sum(),1):6.0f}")
