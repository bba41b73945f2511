"""Diagnostic (GPU box): per-wave section counters of the pooled kernel from the PT_DIAG build.
Run: PTRT_LIB=path-tracer_amd/libptrt_diag.so python tools/diag_pool.py [frames]"""
import ctypes as C
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkgload

os.environ.setdefault("PTRT_KERNEL", "pool")  # these counters exist in the pooled kernel only
pt = _pkgload.load()
scenes = importlib.import_module("path_tracer_amd.scenes")
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 64
scene, w, h, st = scenes.CONFIGS["bunny_1080p_d8_spp256"][0](), 1920, 1080, pt.Settings(1, 8, 1e-4, 1000.0)
r = pt.Renderer(scene, w, h, settings=st)
r.set_sky_equirect(scenes.sky_equirect(4096, 2048), 4096)
L = pt.lib()
L.pt_diag_clear.argtypes = []
L.pt_diag_read.argtypes = [C.c_void_p, C.c_size_t]
r.render_frames(1, 1, frames); r.sync()
r.reset_ready_samples()
L.pt_diag_clear()
r.render_frames(1, 1, frames); r.sync()
rays = r.stats()["rays"]
buf = np.zeros((1 << 18, 8), np.uint64)
assert L.pt_diag_read(buf.ctypes.data_as(C.c_void_p), buf.nbytes) == 0
buf = buf[buf[:, 1] > 0]
M32 = np.uint64(0xFFFFFFFF)
t0, t1 = buf[:, 0].astype(np.int64), buf[:, 1].astype(np.int64)
trips, idle = (buf[:, 2] & M32).astype(np.float64), (buf[:, 2] >> np.uint64(32)).astype(np.float64)
shades, ready = (buf[:, 3] & M32).astype(np.float64), (buf[:, 3] >> np.uint64(32)).astype(np.float64)
shade_cyc = buf[:, 4].astype(np.float64)
steps, busy = (buf[:, 5] & M32).astype(np.float64), (buf[:, 5] >> np.uint64(32)).astype(np.float64)
walk_cyc = buf[:, 6].astype(np.float64)
claim_cyc, claims = (buf[:, 7] & np.uint64((1 << 40) - 1)).astype(np.float64), (buf[:, 7] >> np.uint64(40)).astype(np.float64)
span = (t1.max() - t0.min()) / 100.0
dur = (t1 - t0) / 100.0
print(f"waves {len(buf)} span {span:.0f} us; wave duration p50 {np.median(dur):.0f} max {dur.max():.0f} us; rays (this launch incl. earlier) {rays}")
tot = shade_cyc.sum() + walk_cyc.sum() + claim_cyc.sum()
print("cycle shares: shade %.1f%% walk %.1f%% claim %.1f%%" % (100 * shade_cyc.sum() / tot, 100 * walk_cyc.sum() / tot, 100 * claim_cyc.sum() / tot))
print("trips/wave %.0f, idle trips %.1f%%" % (trips.mean(), 100 * idle.sum() / trips.sum()))
print("shading passes/wave %.0f, mean ready lanes %.1f, cycles per pass %.0f" % (shades.mean(), ready.sum() / shades.sum(), shade_cyc.sum() / shades.sum()))
print("walk steps/wave %.0f, mean busy lanes %.1f, cycles per step %.0f" % (steps.mean(), busy.sum() / max(steps.sum(), 1), walk_cyc.sum() / max(steps.sum(), 1)))
print("claims/wave %.0f, claim cycles per trip %.0f" % (claims.mean(), claim_cyc.sum() / trips.sum()))
ws = steps.reshape(-1, 4) if len(steps) % 4 == 0 else None
if ws is not None:
    srt = np.sort(ws, axis=1)[:, ::-1]
    print("walk steps by rank within the group (most-walking wave first): " + " ".join(f"{v:.0f}" for v in srt.mean(axis=0)))

marks = np.zeros((320, 8), np.uint64)
L.pt_diag_read_marks.argtypes = [C.c_void_p, C.c_size_t]
L.pt_diag_read_marks(marks.ctypes.data_as(C.c_void_p), marks.nbytes)
sec = marks[64:128].astype(np.float64).sum(axis=0)
hist = marks[128:192].astype(np.float64).sum(axis=0)[:5]
extra = marks[128:192].astype(np.float64).sum(axis=0)[5:8]
len_n = marks[192:256].astype(np.float64).sum(axis=0)
len_steps = marks[256:320].astype(np.float64).sum(axis=0)
print("walk steps by busy lanes [1-8, 9-16, 17-32, 33-48, 49-64]: " + " ".join(f"{100 * v / max(hist.sum(), 1):.1f}%" for v in hist))
print("finished walks by steps [1, 2, 3-4, 5-8, 9-16, 17-32, 33-64, 65+]: " + " ".join(f"{100 * v / max(len_n.sum(), 1):.1f}%" for v in len_n))
print("  their share of all lane-steps:                                  " + " ".join(f"{100 * v / max(len_steps.sum(), 1):.1f}%" for v in len_steps) + f"   (mean {len_steps.sum() / max(len_n.sum(), 1):.1f} steps per walk)")
marks = marks[:64]
m = marks.astype(np.float64).sum(axis=0)
if m[3] > 0:
    hp, hl, hc, ap, al, ac = m[:6]
    print("hit passes %.1f%% of passes: %.1f lanes, %.0f cycles; miss passes: %.1f lanes, %.0f cycles" % (
        100 * hp / ap, hl / max(hp, 1), hc / max(hp, 1), (al - hl) / max(ap - hp, 1), (ac - hc) / max(ap - hp, 1)))
    nm = max(ap - hp, 1)
    # (the kernel's stamps dg_sec[0..6] in miss_pass: ring pop; slot state from LDS; sky texel asked for + frame counters + RNG
    # and camera ray; walk_begin_pool of the new ray; everything up to the end of the round's lane work -- the retired
    # path's sums, a sphere hit shaded in place with its bounce ray's own walk_begin_pool, and the hand-over stores of
    # lanes that leave the loop; the wait for loads still outstanding at the round's end; after the loop)
    names = ["pop", "state load", "sky issue + new path", "spheres + root visit", "sums + sphere hit shaded in place (with the bounce ray's spheres + root visit) + hand-over stores",
             "wait for outstanding loads", "write-back + jobs + pushes"]
    print("miss pass rounds per pass %.2f, lanes per round %.1f" % (m[6] / nm, m[7] / max(m[6], 1)))
    print("after a miss pass: job hand-out %.0f cycles per pass (%.1f lanes finished their job), ring pushes %.0f" % (extra[0] / nm, extra[2] / nm, extra[1] / nm))
    print("miss pass sections (cycles): " + ", ".join(f"{n} {v / nm:.0f}" for n, v in zip(names, sec)))
