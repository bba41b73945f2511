"""Host timing (GPU box or anywhere): the two tree builders on the bench mesh and on the 10 M-triangle field."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkgload
pt = _pkgload.load()
scenes = importlib.import_module("path_tracer_amd.scenes")
for name, tris in (("bench mesh", scenes.blob_mesh(76)), ("field", scenes.instanced_field().triangles)):
    for what, fn in (("reference tree (pt_bvh_build)", pt.bvh_build), ("surface-area tree (pt_bvh_build_sah)", pt.bvh_build_sah)):
        t0 = time.perf_counter()
        nodes, _ = fn(tris)
        print(f"{name}, {len(tris)} triangles, {what}: {time.perf_counter() - t0:.2f} s, {len(nodes)} nodes", flush=True)
