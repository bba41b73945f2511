"""GPU box: how long pt_set_mesh takes with and without the planning and building of the second tree (bench mesh, field)."""
import importlib, os, sys, time
sys.path.insert(0, os.getcwd())
import _pkgload
pt = _pkgload.load()
scenes = importlib.import_module("path_tracer_amd.scenes")
for name, scene in (("bench mesh", scenes.bunny()), ("field", scenes.instanced_field())):
    for what, flags in (("default", 0), ("PT_FLAG_REFERENCE_WALK", pt.PT_FLAG_REFERENCE_WALK)):
        r = pt.Renderer(scene, 64, 64, flags=flags)
        t0 = time.perf_counter()
        r.set_mesh(scene.triangles, scene.nodes, scene.triangle_indices)
        dt = time.perf_counter() - t0
        print(f"{name}: pt_set_mesh, {what}: {dt:.3f} s, second tree nodes {r.stats()['guard_nodes']}", flush=True)
        r.close()
