"""The configurations of the headline benchmark at their full sizes, against the CPU oracle: the reference's own
small case in full (bit-exact), and 1080p frames of the sphere scene and of the 69 k-triangle mesh scene -- single
dispatches bit-exact, a fused launch long enough to be split into work items equal up to fp32 summation order,
ray counts exact.  Sized so that the oracle's share stays at a few seconds on 16 host threads."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _exact(got, want, what):
    got, want = np.asarray(got), np.asarray(want)
    same = got.view(np.uint32) == want.view(np.uint32)
    assert same.all(), f"{what}: {int((~same).sum())} of {got.size} values differ (max abs {np.abs(got - want).max():.3e})"


def _config(pt, scenes, name, sky_size):
    factory, w, h, settings, frames = scenes.CONFIGS[name]
    return factory(), w, h, settings, frames, scenes.sky_cube_direct(sky_size)


def test_cornell_640x360_spp16_in_full(pt, po, scenes):
    """BASELINE configs[0]: one dispatch of 16 samples per pixel, depth 4."""
    scene, w, h, st, frames, sky = _config(pt, scenes, "cornell_640x360_d4_spp16", 64)
    assert (w, h, st.samples_per_pixel, st.depth, frames) == (640, 360, 16, 4, 1)
    r = pt.Renderer(scene, w, h, settings=st, flags=pt.PT_FLAG_AOV)
    r.set_sky_cube(sky)
    r.render(1)
    want, prim, tt, ost = po.from_scene(scene, w, h, st, sky).render(1, want_aov=True)
    gprim, gt = r.read_aov()
    assert np.array_equal(gprim, prim)
    _exact(gt, tt, "primary-hit t")
    _exact(r.read_sum(), want, "radiance")
    assert r.stats()["rays"] == ost["rays"]
    r.close()


@pytest.mark.parametrize("name", ["three_balls_1080p_d8_spp256", "bunny_1080p_d8_spp256"])
def test_1080p_frames_match_oracle(pt, po, scenes, name):
    scene, w, h, st, frames, sky = _config(pt, scenes, name, 256)
    assert (w, h, st.depth, frames) == (1920, 1080, 8, 256)
    osc = po.from_scene(scene, w, h, st, sky)
    r = pt.Renderer(scene, w, h, settings=st, flags=pt.PT_FLAG_AOV)
    r.set_sky_cube(sky)
    # two dispatches, frame after frame: the same bits
    r.render(1)
    want, prim, tt, ost = osc.render(1, want_aov=True)
    rays = ost["rays"]
    gprim, gt = r.read_aov()
    assert np.array_equal(gprim, prim)
    _exact(gt, tt, "primary-hit t")
    r.render(2)
    want, _, _, ost = osc.render(2, want)
    rays += ost["rays"]
    _exact(r.read_sum(), want, "two 1080p frames")
    assert r.stats()["rays"] == rays
    # a fused launch that is planned into work items (>= 16 frames): same samples, sums equal up to their order
    n = 24
    r.reset_ready_samples()
    r.render_frames(100, 1, n)
    want = np.zeros((h, w, 4), np.float32)
    rays = 0
    for f in range(n):
        want, _, _, ost = osc.render(100 + f, want)
        rays += ost["rays"]
    got = r.read_sum()
    assert r.stats()["rays"] == rays and (got[..., 3] == n).all()
    np.testing.assert_allclose(got, want, rtol=n * 2.0 ** -23, atol=1e-6)
    r.close()


def test_1080p_several_samples_per_frame_match_oracle(pt, po, scenes):
    """The mesh scene at 3 samples per pixel and frame: one dispatch bit-exact, then a 16-frame launch (48 samples per
    pixel: planned into work items by samples) with exact ray counts and sums equal up to summation order."""
    scene, w, h, st, _, sky = _config(pt, scenes, "bunny_1080p_d8_spp256", 64)
    st = pt.Settings(3, st.depth, st.t_min, st.t_max)
    osc = po.from_scene(scene, w, h, st, sky)
    r = pt.Renderer(scene, w, h, settings=st)
    r.set_sky_cube(sky)
    r.render(9)
    want, _, _, ost = osc.render(9)
    _exact(r.read_sum(), want, "one 1080p frame of 3 samples")
    rays, n = ost["rays"], 16
    r.render_frames(50, 2, n)
    for f in range(n):
        want, _, _, ost = osc.render(50 + 2 * f, want)
        rays += ost["rays"]
    got = r.read_sum()
    assert r.stats()["rays"] == rays and (got[..., 3] == n + 1).all()
    np.testing.assert_allclose(got, want, rtol=(n + 1) * 2.0 ** -23, atol=1e-6)
    r.close()


def test_bunny_4k_frames_match_oracle(pt, po, scenes):
    """BASELINE configs[3] on one GPU: a 3840x2160 dispatch bit-exact (radiance, primary-hit ids and t, ray count), then
    a 16-frame launch planned into work items: ray count exact, sums equal up to summation order."""
    scene, w, h, st, frames, sky = _config(pt, scenes, "bunny_4k_d8_spp4096", 256)
    assert (w, h, st.samples_per_pixel, st.depth, frames) == (3840, 2160, 1, 8, 4096)
    osc = po.from_scene(scene, w, h, st, sky)
    r = pt.Renderer(scene, w, h, settings=st, flags=pt.PT_FLAG_AOV)
    r.set_sky_cube(sky)
    r.render(1)
    want, prim, tt, ost = osc.render(1, want_aov=True)
    gprim, gt = r.read_aov()
    assert np.array_equal(gprim, prim)
    _exact(gt, tt, "primary-hit t")
    _exact(r.read_sum(), want, "one 4K frame")
    assert r.stats()["rays"] == ost["rays"]
    n, rays = 16, ost["rays"]
    r.render_frames(4000, 1, n)
    for f in range(n):
        want, _, _, ost = osc.render(4000 + f, want)
        rays += ost["rays"]
    got = r.read_sum()
    assert r.stats()["rays"] == rays and (got[..., 3] == n + 1).all()
    np.testing.assert_allclose(got, want, rtol=(n + 1) * 2.0 ** -23, atol=1e-6)
    r.close()


@pytest.fixture(scope="module")
def field(pt, po, scenes):
    """BASELINE configs[4]: the 9 980 928-triangle instanced field, built once for both kernels; the oracle's own
    builder must give the same 12.4 M nodes and the same index permutation as the library's threaded one."""
    factory, w, h, st, frames = scenes.CONFIGS["field_4k_d8_spp1024"]
    assert (w, h, st.samples_per_pixel, st.depth, frames) == (3840, 2160, 1, 8, 1024)
    scene = factory()
    nodes, idx = po.bvh_build(scene.triangles)
    assert np.array_equal(nodes.view(np.uint8), scene.nodes.view(np.uint8)) and np.array_equal(idx, scene.triangle_indices)
    # depth of the deepest node, level by level over the oracle's nodes
    depth, level = 0, np.zeros(1, np.int64)
    while True:
        inner = level[nodes["count"][level] == 0]
        if inner.size == 0:
            break
        left = nodes["left"][inner].astype(np.int64)
        level = np.concatenate([left, left + 1])
        depth += 1
    return scene, nodes, idx, depth, w, h, st


def test_field_4k_frame_matches_oracle(pt, po, scenes, field):
    """One full-resolution dispatch of the 10 M-triangle scene (tree deeper than either kernel's on-chip stack, leaf
    tokens with 24-bit slot numbers): radiance, primary-hit ids and t bit-exact, ray count and tree depth equal."""
    scene, nodes, idx, depth, w, h, st = field
    assert len(scene.triangles) == 9980928 and depth > 24
    sky = scenes.sky_cube_direct(64)
    osc = po.OracleScene(w, h, scene.camera.buffer(), st.buffer(), po.pack_spheres(scene.spheres), scene.triangles, nodes, idx, sky)
    r = pt.Renderer(scene, w, h, settings=st, flags=pt.PT_FLAG_AOV)
    r.set_sky_cube(sky)
    r.render(1)
    want, prim, tt, ost = osc.render(1, want_aov=True)
    gprim, gt = r.read_aov()
    assert np.array_equal(gprim, prim) and int((prim < 0x80000000).sum()) > 1000000
    _exact(gt, tt, "primary-hit t")
    _exact(r.read_sum(), want, "one 4K frame of the field")
    s = r.stats()
    assert s["rays"] == ost["rays"] and s["bvh_depth"] == depth and ost["max_stack"] <= depth
    r.close()


def test_field_planned_launch_matches_oracle(pt, po, scenes, field):
    """The same scene in a 16-frame planned launch at 960x540 (the stacks spill to global memory in planned launches
    too): ray count exact, sums equal up to summation order."""
    scene, nodes, idx, depth, _, _, st = field
    w, h, n = 960, 540, 16
    sky = scenes.sky_cube_direct(64)
    osc = po.OracleScene(w, h, scene.camera.buffer(), st.buffer(), po.pack_spheres(scene.spheres), scene.triangles, nodes, idx, sky)
    r = pt.Renderer(scene, w, h, settings=st)
    r.set_sky_cube(sky)
    r.render_frames(1, 1, n)
    want, rays = np.zeros((h, w, 4), np.float32), 0
    for f in range(n):
        want, _, _, ost = osc.render(1 + f, want)
        rays += ost["rays"]
    got = r.read_sum()
    assert r.stats()["rays"] == rays and (got[..., 3] == n).all()
    np.testing.assert_allclose(got, want, rtol=n * 2.0 ** -23, atol=1e-6)
    r.close()


@pytest.mark.library_choice
def test_bench_sky_cube_4096_matches_oracle(pt, po, scenes):
    """The benchmark's own sky: the 4096 x 2048 procedural panorama through equirect_kernel into the 4096^2 cube
    (shaders/equirectangular.wgsl:12-81, renderer.rs:218), every texel of every face against the oracle's
    pto_equirect_to_cube, bit for bit -- and every 64th row once more on its own, so that a failure names the face and row."""
    pano = scenes.sky_equirect(4096, 2048)
    r = pt.Renderer(scenes.three_balls(), 64, 40, settings=pt.Settings(1, 2, 1e-4, 1000.0))
    r.set_sky_equirect(pano, 4096)
    got = r.read_sky_cube()
    r.close()
    want = po.equirect_to_cube(pano, 4096)
    assert got.shape == want.shape == (6, 4096, 4096, 4)
    for face in range(6):
        for row in range(0, 4096, 64):
            assert np.array_equal(got[face, row].view(np.uint32), want[face, row].view(np.uint32)), f"face {face}, row {row}"
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
