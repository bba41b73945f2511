"""Analysis on the CPU oracle (test infrastructure; not a test, not collected): how long are the BVH walks of the bench
scene's rays, bounce level by bounce level, and what is the chain of dependent steps that bounds a single frame --
(a) for a scheduler that ties a tile's 64 pixels together bounce level by bounce level (sum over levels of the tile's
longest walk), (b) for one that lets every path run on its own (the longest single path)?  Optionally with another tree
over the same triangles (--sah: the library's binned-SAH builder, if it has one) to see what a better tree would save,
and with that builder's tree over padded triangle bounds (--second-tree: the padding restated here from pt_set_mesh's
plan_guard -- 48 * 2^-24 * 1e5 * |e1| |e2| * reach, reach = 8 * (mesh radius + longest edge) -- to see what the tree
that the megakernels walk by default costs over the unpadded one).

Usage: python tests/analysis/walk_stats.py [--depth 16] [--width 1920 --height 1080] [--time 100] [--n 76] [--sah] [--second-tree] [--field]"""
import argparse
import ctypes as C
import importlib
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))
LEVELS = 32


def build():
    so = os.path.join(HERE, "libwalk_stats.so")
    src = os.path.join(HERE, "walk_stats.c")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(os.path.join(ROOT, "oracle", "pt_oracle.c"))):
        subprocess.check_call(["gcc", "-O2", "-std=c11", "-ffp-contract=off", "-mfma", "-fPIC", "-shared", "-pthread", src, "-o", so, "-lm"])
    return C.CDLL(so)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--depth", type=int, default=16)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--time", type=int, default=100)
    ap.add_argument("--n", type=int, default=76)
    ap.add_argument("--sah", action="store_true")
    ap.add_argument("--second-tree", action="store_true")
    ap.add_argument("--field", action="store_true")
    a = ap.parse_args()
    import _pkgload
    pt = _pkgload.load()
    scenes = importlib.import_module("path_tracer_amd.scenes")
    from oracle import pt_oracle as po
    L = build()
    scene = scenes.instanced_field(n=a.n) if a.field else scenes.bunny(n=a.n)
    st = pt.Settings(1, a.depth, 1e-4, 1000.0)
    sky = scenes.sky_cube_direct(16)
    osc = po.from_scene(scene, a.width, a.height, st, sky)
    trees = [("reference (midpoint split, bvh.rs)", osc)]
    if a.sah:
        nodes, idx = pt.bvh_build_sah(scene.triangles)
        trees.append(("binned SAH", po.OracleScene(a.width, a.height, scene.camera.buffer(), st.buffer(), po.pack_spheres(scene.spheres),
                                                   scene.triangles, nodes, idx, sky, 0)))
    if a.second_tree:
        tr = scene.triangles
        A, B, Cc = (tr[k].astype(np.float64) for k in ("a", "b", "c"))
        e1, e2, e3 = np.linalg.norm(B - A, axis=1), np.linalg.norm(Cc - A, axis=1), np.linalg.norm(Cc - B, axis=1)
        allv = np.concatenate([A, B, Cc])
        centre = 0.5 * (allv.min(0) + allv.max(0))
        reach = 8.0 * (np.linalg.norm(allv - centre, axis=1).max() + max(e1.max(), e2.max(), e3.max()))
        pad = 48.0 * 2.0 ** -24 * 1e5 * reach * e1 * e2 + 8.0 * 2.0 ** -24 * (np.abs(allv).max() + 2.0 * reach + np.abs(centre).max())
        print(f"second tree: reach {reach:.3f}, padding median {np.median(pad):.2e}, max {pad.max():.2e}; triangle extent median "
              f"{np.median((np.maximum(np.maximum(A, B), Cc) - np.minimum(np.minimum(A, B), Cc)).max(1)):.2e}")
        nodes, idx = pt.bvh_build_sah(tr, pad=pad.astype(np.float32))
        trees.append(("binned SAH over padded bounds (the megakernels' second tree)",
                      po.OracleScene(a.width, a.height, scene.camera.buffer(), st.buffer(), po.pack_spheres(scene.spheres), tr, nodes, idx, sky, 0)))
    for name, o in trees:
        out = np.zeros((a.height, a.width, LEVELS), np.uint16)
        acc = np.zeros((a.height, a.width, 4), np.float32)
        stats = po.Stats()
        rc = L.ws_render(C.byref(o.struct), a.time, out.ctypes.data_as(C.c_void_p), acc.ctypes.data_as(C.c_void_p), C.byref(stats), min(os.cpu_count() or 1, 16))
        assert rc == 0
        s = stats.as_dict()
        steps = out.astype(np.int64)
        walked = steps > 1                      # more than the root visit
        print(f"== {name}: {len(o.nodes)} nodes; {s['rays']} rays, {s['interior']} interior steps, {s['tri_tests']} triangle tests")
        print(f"   rays that walk (more than the root visit): {int(walked.sum())} = {walked.sum() / s['rays']:.3%} of rays; "
              f"steps per walking ray: mean {steps[walked].mean():.1f}, median {np.median(steps[walked]):.0f}, 99 % {np.percentile(steps[walked], 99):.0f}, max {steps.max()}")
        print("   level: walking rays, mean steps, max steps")
        for lv in range(min(a.depth, LEVELS)):
            w = walked[..., lv]
            if w.any():
                print(f"   {lv:5d}: {int(w.sum()):9d} {steps[..., lv][w].mean():8.1f} {steps[..., lv].max():6d}")
        per_path = steps.sum(-1)
        print(f"   longest single path: {per_path.max()} steps (99.99 % of walking paths: {np.percentile(per_path[per_path > 1], 99.99):.0f})")
        h8, w8 = (a.height // 8) * 8, (a.width // 8) * 8
        tiles = steps[:h8, :w8].reshape(h8 // 8, 8, w8 // 8, 8, LEVELS).max(axis=(1, 3))     # per tile and level: the longest walk
        chain = tiles.sum(-1)
        print(f"   per 8x8 tile, sum over levels of the tile's longest walk: max {chain.max()}, 99.9 % {np.percentile(chain, 99.9):.0f}; "
              f"sum over levels of the image's longest walk: {steps.max(axis=(0, 1)).sum()}")
        # Two phases: a dense one in which a pixel gives up once its path has taken more than S steps in all (the ray in
        # flight is started again in the second phase), and a sparse one for the paths that gave up.
        cum = np.cumsum(steps, -1)
        print("   budget S: paths over S, their remaining steps (restarting the ray in flight): total / longest; dense phase: longest tile chain")
        for S in (64, 96, 128, 192, 256, 384):
            before = cum - steps                       # steps taken when a ray starts
            over_lv = (cum > S)                        # the ray during which the budget runs out, and all later ones
            gave_up = over_lv.any(-1)
            rest = np.where(over_lv, steps, 0).sum(-1)  # the ray in flight is redone in full
            done = np.where(over_lv, np.minimum(steps, np.maximum(S - before, 0)), steps)   # steps the dense phase spends
            t = done[:h8, :w8].reshape(h8 // 8, 8, w8 // 8, 8, LEVELS).max(axis=(1, 3)).sum(-1)
            print(f"   {S:8d}: {int(gave_up.sum()):7d} {int(rest.sum()):9d} / {int(rest.max()):5d}; {int(t.max()):5d} (99.9 % of tiles {np.percentile(t, 99.9):.0f})")
        if name.startswith("reference"):
            ref_img = acc.copy()
        else:
            d = np.abs(acc - ref_img)[..., :3]
            print(f"   image against the reference tree's: {int((d > 0).any(-1).sum())} of {a.width * a.height} pixels differ (one sample per pixel), max abs {d.max():.3g}")


if __name__ == "__main__":
    main()
