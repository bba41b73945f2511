"""GPU box, tuning build (make tuning; PTRT_LIB=path-tracer_amd/libptrt_tuning.so), under rocprofv3 --kernel-trace:
how long does ONE ray's BVH walk take on an otherwise idle GPU, per traversal step?  Explicit ray queries
(pt_trace_rays -> ray_query_kernel, the plain one-visit-per-trip walk) of camera rays into the bench scene's mesh: first
a batch to find rays with long walks (the tuning build reports each ray's step count), then each of the longest alone,
then 64 copies of it in one wave, then 64 different long rays in one wave.  Prints the step counts in launch order; the
kernel durations are in the trace (tools/step_latency_report.py joins the two)."""
import importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkgload
pt = _pkgload.load()
scenes = importlib.import_module("path_tracer_amd.scenes")
scene = scenes.bunny()
r = pt.Renderer(scene, 64, 64, settings=pt.Settings(1, 8, 1e-4, 1000.0))
cam = scene.camera
rng = np.random.RandomState(3)
n = 20000
# rays from around the camera towards points near the mesh's silhouette
o = np.tile(np.array(cam.origin, np.float32), (n, 1))
target = np.array([0.0, 0.5, -1.0], np.float32) + rng.normal(0, 0.55, (n, 3)).astype(np.float32)
d = (target - o).astype(np.float32)
prim, t, p, nrm, front = r.trace_rays(o, d)
steps = front >> 1
order = np.argsort(-steps.astype(np.int64))
launches = [{"what": "batch", "rays": n, "steps_max": int(steps.max()), "steps_mean": float(steps.mean())}]
for k in order[:6]:
    for rep in range(2):
        _, _, _, _, f1 = r.trace_rays(o[k:k + 1], d[k:k + 1])
        launches.append({"what": "one ray alone", "rays": 1, "steps_max": int(f1[0] >> 1)})
    _, _, _, _, f64 = r.trace_rays(np.tile(o[k], (64, 1)), np.tile(d[k], (64, 1)))
    launches.append({"what": "64 copies of it in one wave", "rays": 64, "steps_max": int((f64 >> 1).max())})
top = order[:64]
_, _, _, _, ft = r.trace_rays(o[top], d[top])
launches.append({"what": "the 64 longest, one wave", "rays": 64, "steps_max": int((ft >> 1).max()), "steps_mean": float((ft >> 1).mean())})
# k long rays per wave, the wave's other lanes given rays that end at once (pointing away from everything)
away_o, away_d = np.array(cam.origin, np.float32), np.array([0.0, 1.0, 0.3], np.float32)
for k in (2, 4, 8, 16, 32):
    oo, dd = np.tile(away_o, (64 * (64 // k), 1)), np.tile(away_d, (64 * (64 // k), 1))
    for wv in range(64 // k):
        oo[64 * wv:64 * wv + k] = o[top[wv * k:(wv + 1) * k]]
        dd[64 * wv:64 * wv + k] = d[top[wv * k:(wv + 1) * k]]
    _, _, _, _, fk = r.trace_rays(oo, dd)
    launches.append({"what": f"the 64 longest, {k} per wave ({64 // k} waves)", "rays": len(oo), "steps_max": int((fk >> 1).max()), "steps_mean": float((fk >> 1)[(fk >> 1) > 1].mean())})
mid = order[2000:2064]
_, _, _, _, fm = r.trace_rays(o[mid], d[mid])
launches.append({"what": "64 middling rays, one wave", "rays": 64, "steps_max": int((fm >> 1).max()), "steps_mean": float((fm >> 1).mean())})
json.dump(launches, open(os.path.join(os.environ.get("PT_OUT", "."), "step_latency_launches.json"), "w"), indent=1)
print(json.dumps(launches, indent=1))
r.close()
