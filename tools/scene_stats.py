"""GPU box: per-ray work counts of the bench scene (detailed-stats build of the kernels), for sizing optimisations."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkgload

pt = _pkgload.load()
scenes = importlib.import_module("path_tracer_amd.scenes")
name = sys.argv[1] if len(sys.argv) > 1 else "bunny_1080p_d8_spp256"
factory, w, h, st, _ = scenes.CONFIGS[name]
scene = factory()
r = pt.Renderer(scene, w, h, settings=st, flags=pt.PT_FLAG_DETAILED_STATS)
r.set_sky_cube(scenes.sky_cube_direct(64))
r.render_frames(1, 1, 16)
s = r.stats()
rays = s["rays"]
print(name, {k: round(v / rays, 3) for k, v in s.items() if k not in ("frames", "bvh_depth", "rays")}, "rays", rays)
leaf2 = sum(1 for n in scene.nodes if n["count"] == 2)
leaf1 = sum(1 for n in scene.nodes if n["count"] == 1)
leafn = sum(1 for n in scene.nodes if n["count"] > 2)
print("leaves with 1 / 2 / more triangles:", leaf1, leaf2, leafn)
r.close()
