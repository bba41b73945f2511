"""GPU box: the bench scene at 1080p from cameras all around (and inside) the mesh, rendered in exact short launches
(15 frames: every frame added on its own, in order) by both megakernels with the second tree and with
PT_FLAG_REFERENCE_WALK; the four accumulated images and ray counts of every camera must be the same bits.
Usage: python -u tools/guard_render_soak.py [launches per camera]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkgload
pt = _pkgload.load()
scenes = importlib.import_module("path_tracer_amd.scenes")
launches = int(sys.argv[1]) if len(sys.argv) > 1 else 50
base = scenes.bunny()
sky = scenes.sky_cube_direct(64)
rng = np.random.RandomState(7)
centre = np.array([0.0, 0.5, -1.0])
total_rays = 0
for cam_i in range(12):
    dist = (0.3, 1.6, 2.5, 4.0, 5.9, 7.0, 9.0, 14.0)[cam_i % 8]
    yaw, pitch = float(rng.uniform(-180, 180)), float(rng.uniform(-40, 5))
    fwd = np.array([np.cos(np.radians(yaw)) * np.cos(np.radians(pitch)), np.sin(np.radians(pitch)), np.sin(np.radians(yaw)) * np.cos(np.radians(pitch))])
    cam = pt.Camera.from_yaw_pitch(tuple(centre - fwd * dist), yaw, pitch, 1.0, float(rng.uniform(30, 90)))
    scene = pt.Scene(base.spheres, base.triangles, cam)
    depth = int(rng.choice([2, 8, 32]))
    images = {}
    for kernel, kf in (("pooled", pt.PT_FLAG_POOL_KERNEL), ("per-tile", pt.PT_FLAG_TILE_KERNEL)):
        for walk, wf in (("both trees", 0), ("reference", pt.PT_FLAG_REFERENCE_WALK)):
            r = pt.Renderer(scene, 1920, 1080, settings=pt.Settings(1, depth, 1e-4, 1000.0), flags=kf | wf)
            r.set_sky_cube(sky)
            for k in range(launches):
                r.render_frames(1 + 15 * k, 1, 15)
            s = r.stats()
            images[kernel, walk] = (r.read_sum().copy(), s["rays"], s["guarded"], s["rewalks"])
            r.close()
    ref = images["pooled", "reference"]
    for key, (img, rays, guarded, rewalks) in images.items():
        if rays != ref[1] or not np.array_equal(img.view(np.uint32), ref[0].view(np.uint32)):
            print(f"MISMATCH camera {cam_i} (distance {dist}, depth {depth}): {key}: {int((img.view(np.uint32) != ref[0].view(np.uint32)).sum())} sums differ, rays {rays} vs {ref[1]}")
            sys.exit(1)
    total_rays += ref[1]
    g = images["pooled", "both trees"]
    print(f"camera {cam_i}: {dist} from the mesh's centre, depth {depth}, {15 * launches} frames, {ref[1]} rays per rendering; on the second tree {g[2]} (pooled) / "
          f"{images['per-tile', 'both trees'][2]} (per-tile), walked both {g[3]} / {images['per-tile', 'both trees'][3]}: four renderings identical", flush=True)
print(f"ok: {total_rays} rays per mode")
