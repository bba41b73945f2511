"""Writes tests/golden/cornell_40x24.npz from the CPU oracle.  The reference has no fixtures of its
own (and cannot run here), so this file pins the ORACLE's output against later edits; inputs are
exact binary constants, so the file regenerates bit-identically on any machine.
Run: python tests/golden/make_golden.py"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import _pkgload

pt = _pkgload.load()
scenes = importlib.import_module("path_tracer_amd.scenes")
from oracle import pt_oracle as po


def golden_inputs():
    scene = scenes.cornell(pt.Camera((1.0, 1.0, 4.75), (0.0, 0.0, -1.0), (1.0, 0.0, 0.0), (0.0, 1.0, 0.0), 1.0, 60.0))
    sky = np.zeros((6, 4, 4, 4), np.float32)
    for f in range(6):
        for j in range(4):
            for i in range(4):
                sky[f, j, i] = (0.25 * (f + 1), 0.125 * (i + 1), 0.0625 * (j + 1), 1.0)
    return scene, sky, pt.Settings(2, 5, 1e-4, 1000.0), 40, 24, [3, 4]


if __name__ == "__main__":
    scene, sky, st, w, h, times = golden_inputs()
    osc = po.from_scene(scene, w, h, st, sky)
    acc = np.zeros((h, w, 4), np.float32)
    rays = 0
    for t in times:
        acc, prim, tt, stats = osc.render(t, acc, want_aov=True)
        rays += stats["rays"]
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cornell_40x24.npz")
    np.savez_compressed(out, accum=acc, prim=prim, t=tt, rays=np.int64(rays))
    print("wrote", out, "rays", rays)
