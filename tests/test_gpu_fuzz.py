"""Seeded random scenes through both megakernels against the CPU oracle: triangle soups with random
materials, random sphere sets (all four materials, spheres inside spheres), random cameras, odd image sizes, random
spp/depth, single dispatches and short fused launches.  Everything must agree bit for bit, ray counts included."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
EXTRA = int(os.environ.get("PTRT_FUZZ_EXTRA", "0"))  # more seeds for a one-off soak, e.g. after touching a megakernel


def _random_scene(pt, scenes, rng):
    M = pt.Material
    n_planes = int(rng.integers(0, 40))
    tris = []
    for _ in range(n_planes):
        q = rng.uniform(-2.0, 2.0, 3)
        u = rng.normal(0.0, 0.8, 3)
        v = rng.normal(0.0, 0.8, 3)
        tris.append(pt.Plane(tuple(q), tuple(u), tuple(v), tuple(rng.uniform(0.1, 1.0, 3)), M(int(rng.integers(0, 3)))).triangles())
    if tris and rng.random() < 0.3:
        tris.append(tris[0].copy())  # coincident triangles: equal t, order decides
    if rng.random() < 0.35:  # a closed mesh with a deeper tree
        tris.append(scenes.blob_mesh(int(rng.integers(3, 22)), center=tuple(rng.uniform(-0.5, 0.5, 3)), radius=float(rng.uniform(0.3, 1.2))))
    triangles = np.concatenate(tris) if tris else np.zeros(0, pt.TRIANGLE_DTYPE)
    spheres = []
    for _ in range(int(rng.integers(0 if n_planes else 1, 7))):
        spheres.append(pt.Sphere(tuple(rng.uniform(-1.5, 1.5, 3)), float(rng.uniform(0.1, 0.9)), tuple(rng.uniform(0.2, 1.0, 3)),
                                 M(int(rng.integers(0, 4)))))
    if rng.random() < 0.5:
        spheres.append(pt.Sphere((0.0, -100.5, -1.0), 100.0, (0.8, 0.8, 0.0), M.Diffuse))
    cam = pt.Camera.from_yaw_pitch(tuple(rng.uniform(-1.0, 1.0, 3) + np.array([0.0, 0.5, 3.0])), float(rng.uniform(-120.0, -60.0)),
                                   float(rng.uniform(-25.0, 15.0)), vfov=float(rng.uniform(40.0, 100.0)))
    return pt.Scene(spheres, triangles, cam)


@pytest.mark.parametrize("seed", range(40 + EXTRA))
def test_random_scene(pt, po, scenes, seed):
    rng = np.random.default_rng(1000 + seed)
    scene = _random_scene(pt, scenes, rng)
    w, h = int(rng.integers(9, 130)), int(rng.integers(9, 90))
    st = pt.Settings(int(rng.integers(1, 4)), int(rng.integers(1, 12)), 1e-4, 1000.0)
    sky = scenes.sky_cube_direct(int(rng.choice([1, 4, 16]))) if rng.random() < 0.8 else None
    flags = pt.PT_FLAG_AOV | (pt.PT_FLAG_TRIANGLE_MATERIALS if rng.random() < 0.5 else 0) | (pt.PT_FLAG_REFERENCE_DISPATCH if rng.random() < 0.25 else 0)
    oflags = (po.FLAG_REFERENCE_DISPATCH if flags & pt.PT_FLAG_REFERENCE_DISPATCH else 0) | (po.FLAG_TRIANGLE_MATERIALS if flags & pt.PT_FLAG_TRIANGLE_MATERIALS else 0)
    r = pt.Renderer(scene, w, h, settings=st, flags=flags)
    if sky is not None:
        r.set_sky_cube(sky)
    osc = po.from_scene(scene, w, h, st, sky, flags=oflags)
    t0 = int(rng.integers(0, 2 ** 32 - 64))
    r.render(t0)
    want, prim, tt, ost = osc.render(t0, want_aov=True)
    rays = ost["rays"]
    gprim, gt = r.read_aov()
    what = f"seed {seed}: {len(scene.triangles)} triangles, {len(scene.spheres)} spheres, {w}x{h}, spp {st.samples_per_pixel}, depth {st.depth}"
    assert np.array_equal(gprim, prim), what
    assert np.array_equal(gt.view(np.uint32), tt.view(np.uint32)), what
    n = int(rng.integers(2, 15))
    stride = int(rng.integers(1, 5))
    r.render_frames(t0 + 1, stride, n)
    for f in range(n):
        want, _, _, ost = osc.render(t0 + 1 + f * stride, want)
        rays += ost["rays"]
    got = r.read_sum()
    same = got.view(np.uint32) == want.view(np.uint32)
    assert same.all(), f"{what}: {int((~same).sum())} of {got.size} sums differ"
    assert r.stats()["rays"] == rays, what
    r.close()


@pytest.mark.parametrize("seed", range(10 + EXTRA))
def test_random_scene_planned_launch(pt, po, scenes, seed):
    """Fused launches of 16 frames and more are split into work items by measured cost: same samples and ray
    counts as the oracle's frame-by-frame sums, equal up to fp32 summation order; a second launch reuses the plan."""
    rng = np.random.default_rng(5000 + seed)
    scene = _random_scene(pt, scenes, rng)
    w, h = int(rng.integers(40, 200)), int(rng.integers(30, 120))
    st = pt.Settings(int(rng.integers(1, 3)), int(rng.integers(1, 10)), 1e-4, 1000.0)
    sky = scenes.sky_cube_direct(8)
    r = pt.Renderer(scene, w, h, settings=st)
    r.set_sky_cube(sky)
    osc = po.from_scene(scene, w, h, st, sky)
    want = np.zeros((h, w, 4), np.float32)
    rays, total = 0, 0
    for t0 in (7, 1000):
        n = int(rng.integers(16, 48))
        r.render_frames(t0, 1, n)
        for f in range(n):
            want, _, _, ost = osc.render(t0 + f, want)
            rays += ost["rays"]
        total += n
    got = r.read_sum()
    assert r.stats()["rays"] == rays and (got[..., 3] == total).all()
    np.testing.assert_allclose(got, want, rtol=total * 2.0 ** -23, atol=1e-6)
    r.close()
