"""The two implementations of the megakernel -- the pooled one (PT_FLAG_POOL_KERNEL) and the
one-wave-per-tile one (PT_FLAG_TILE_KERNEL, also the fallback for depth 0) -- must produce the same bits:
sums, primary-hit AOV, ray count and the detailed counters.  The other GPU tests pin each of them to the
oracle (tests/conftest.py runs every GPU test once per kernel)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def megakernel(pt):
    """These tests choose the kernel themselves, through the flags (nothing is forced from outside)."""
    pt.force_kernel(None)
    yield None


def _render(pt, scene, w, h, st, sky, flags, calls):
    r = pt.Renderer(scene, w, h, settings=st, flags=pt.PT_FLAG_AOV | flags)
    if sky is not None:
        r.set_sky_cube(sky)
    for c in calls:
        if isinstance(c, tuple):
            r.render_frames(*c)
        else:
            r.render(c)
    # (which tree a ray walked is the pooled kernel's own business: not part of what the two kernels must agree on)
    counters = {k: v for k, v in r.stats().items() if k not in ("guarded", "rewalks")}
    out = (r.read_sum().copy(),) + tuple(a.copy() for a in r.read_aov()) + (counters,)
    r.close()
    return out


def _same(pt, scene, w, h, st, sky, calls, flags=0):
    a = _render(pt, scene, w, h, st, sky, flags | pt.PT_FLAG_POOL_KERNEL, calls)
    b = _render(pt, scene, w, h, st, sky, flags | pt.PT_FLAG_TILE_KERNEL, calls)
    assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)), \
        f"sums: {int((a[0].view(np.uint32) != b[0].view(np.uint32)).sum())} of {a[0].size} values differ"
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32)), "primary-hit AOV differs"
    assert a[3] == b[3], f"counters differ: {a[3]} vs {b[3]}"
    return a


def _gizmo_scene(pt, scenes):
    S, M = pt.Sphere, pt.Material
    spheres = [S((0, 0, -1), 0.5, (0.8, 0.3, 0.3), M.Diffuse), S((1, 0, -1), 0.5, (1, 1, 1), M.Dielectric),
               S((-1, 0, -1), 0.5, (0.8, 0.6, 0.2), M.Metal), S((0, 1.0, -1), 0.4, (0.2, 0.9, 0.2), M.Gizmo),
               S((0, -100.5, -1), 100.0, (0.8, 0.8, 0.0), M.Diffuse)]
    cam = pt.Camera.from_yaw_pitch((0.0, 0.6, 1.8), -90.0, -8.0)
    return pt.Scene(spheres, scenes.cornell().triangles, cam)


@pytest.mark.parametrize("spp,depth", [(1, 8), (3, 5)])
def test_single_dispatches_match(pt, scenes, spp, depth):
    """All four compiled variants (spp 1 or more, with and without a gizmo sphere), odd image size."""
    sky = scenes.sky_cube_direct(16)
    st = pt.Settings(spp, depth, 1e-4, 1000.0)
    _same(pt, scenes.cornell(), 100, 59, st, sky, [1, 2, 7])
    _same(pt, _gizmo_scene(pt, scenes), 100, 59, st, sky, [3, 4])


def test_fused_frames_match(pt, scenes):
    """Fused launches on a mesh with a deep tree.  Short launches add every frame in order: same bits.
    Long ones are split into work items by measured cost, which the two kernels measure differently,
    so their partial sums are grouped differently: same rays, sums equal up to summation order."""
    scene = scenes.bunny(n=24)
    sky = scenes.sky_cube_direct(32)
    st = pt.Settings(1, 8, 1e-4, 1000.0)
    a = _same(pt, scene, 192, 108, st, sky, [(1, 1, 5), (6, 1, 15), (100, 3, 9)])
    assert (a[0][..., 3] == 29).all() and a[3]["frames"] == 29
    calls = [(1, 1, 5), (6, 1, 40), (100, 3, 40)]
    p = _render(pt, scene, 192, 108, st, sky, pt.PT_FLAG_POOL_KERNEL, calls)
    t = _render(pt, scene, 192, 108, st, sky, pt.PT_FLAG_TILE_KERNEL, calls)
    assert p[3] == t[3] and np.array_equal(p[1], t[1]) and np.array_equal(p[0][..., 3], t[0][..., 3]) and (p[0][..., 3] == 85).all()
    np.testing.assert_allclose(p[0], t[0], rtol=85 * 2.0 ** -23, atol=1e-6)
    p2 = _render(pt, scene, 192, 108, st, sky, pt.PT_FLAG_POOL_KERNEL, calls)
    assert np.array_equal(p[0].view(np.uint32), p2[0].view(np.uint32)), "pooled kernel is not reproducible run to run"


def test_detailed_stats_and_reference_dispatch_match(pt, scenes):
    sky = scenes.sky_cube_direct(16)
    st = pt.Settings(2, 6, 1e-4, 1000.0)
    a = _same(pt, _gizmo_scene(pt, scenes), 100, 59, st, sky, [5, (6, 1, 12)], flags=pt.PT_FLAG_DETAILED_STATS | pt.PT_FLAG_REFERENCE_DISPATCH)
    assert a[3]["interior"] > 0 and a[3]["tri_tests"] > 0 and a[3]["sky_lookups"] > 0
    assert (a[0][48:] == 0).all() and (a[0][:, 96:] == 0).all()  # outside floor(w/16)*16 x floor(h/16)*16


def test_depth_zero_uses_the_tile_kernel(pt, po, scenes):
    """depth 0: no ray is ever traced and every path is white (compute.wgsl:131-137, 203)."""
    st = pt.Settings(2, 0, 1e-4, 1000.0)
    r = pt.Renderer(scenes.cornell(), 40, 24, settings=st, flags=pt.PT_FLAG_POOL_KERNEL)
    r.render_frames(1, 1, 3)
    s = r.read_sum()
    assert (s[..., :3] == 3.0).all() and (s[..., 3] == 3.0).all() and r.stats()["rays"] == 0
    r.close()


def test_automatic_choice_by_mesh_size(pt, scenes):
    """Without flags the library takes the per-tile kernel for small meshes and the pooled one for big ones;
    either way the image is the same as with the kernel forced."""
    sky = scenes.sky_cube_direct(16)
    st = pt.Settings(1, 4, 1e-4, 1000.0)
    for n in (4, 76):  # 192 and 69 312 triangles
        scene = scenes.bunny(n=n)
        auto = _render(pt, scene, 96, 54, st, sky, 0, [(1, 1, 6)])
        forced = _render(pt, scene, 96, 54, st, sky, pt.PT_FLAG_TILE_KERNEL, [(1, 1, 6)])
        assert np.array_equal(auto[0].view(np.uint32), forced[0].view(np.uint32)) and auto[3] == forced[3]
    # a long launch on the big mesh goes to the pooled kernel: same bits as with that kernel forced
    auto = _render(pt, scene, 96, 54, st, sky, 0, [(1, 1, 130)])
    forced = _render(pt, scene, 96, 54, st, sky, pt.PT_FLAG_POOL_KERNEL, [(1, 1, 130)])
    assert np.array_equal(auto[0].view(np.uint32), forced[0].view(np.uint32)) and auto[3] == forced[3]
    # several samples per frame count as that many frames: one frame of 50 samples goes to the pooled kernel too
    # (a single frame is summed in sample order by both kernels: same bits), and so do 16 frames of 3
    st = pt.Settings(50, 4, 1e-4, 1000.0)
    auto = _render(pt, scene, 96, 54, st, sky, 0, [7])
    for flag in (pt.PT_FLAG_POOL_KERNEL, pt.PT_FLAG_TILE_KERNEL):
        forced = _render(pt, scene, 96, 54, st, sky, flag, [7])
        assert np.array_equal(auto[0].view(np.uint32), forced[0].view(np.uint32)) and auto[3] == forced[3]
    st = pt.Settings(3, 4, 1e-4, 1000.0)
    auto = _render(pt, scene, 96, 54, st, sky, 0, [(1, 1, 16)])
    forced = _render(pt, scene, 96, 54, st, sky, pt.PT_FLAG_POOL_KERNEL, [(1, 1, 16)])
    tile = _render(pt, scene, 96, 54, st, sky, pt.PT_FLAG_TILE_KERNEL, [(1, 1, 16)])
    assert auto[3] == forced[3] == tile[3] and np.array_equal(auto[1], tile[1])
    np.testing.assert_allclose(auto[0], forced[0], rtol=16 * 2.0 ** -23, atol=1e-6)
    np.testing.assert_allclose(auto[0], tile[0], rtol=16 * 2.0 ** -23, atol=1e-6)


@pytest.mark.parametrize("flag", ["PT_FLAG_POOL_KERNEL", "PT_FLAG_TILE_KERNEL"])
def test_repeated_launches_reuse_their_plan(pt, po, scenes, flag):
    """Pooled kernel: from the third identical launch on the work-item table of the previous launch is run again
    (no planning, no cost measurement), the ninth measures the cost again and the tenth plans anew; setting anything
    on the context ends that.  The per-tile kernel plans every launch.  Every launch must still trace every sample:
    ray counts and frame counts exact, sums equal to the oracle's up to summation order, and two contexts fed the
    same call sequence agree bit for bit."""
    scene = scenes.bunny(n=12)
    sky = scenes.sky_cube_direct(16)
    st = pt.Settings(1, 6, 1e-4, 1000.0)
    w, h, n = 120, 72, 20
    osc = po.from_scene(scene, w, h, st, sky)
    want = np.zeros((h, w, 4), np.float32)
    rays = 0

    def run(r):
        for k in range(12):             # pooled: launches 3..8 reuse the plan of launch 2, 9 measures, 10 plans, 11.. reuse
            r.render_frames(1 + k * n, 1, n)
        moved = pt.Scene(scene.spheres, scene.triangles, pt.Camera.from_yaw_pitch((0.8, 1.1, 4.2), -92.0, -3.0))
        r.update_scene(moved)           # new camera: the old plan must not be run again blindly (it would still be correct)
        for k in range(3):
            r.render_frames(1000 + k * n, 1, n)
        return r.read_sum().copy(), r.stats()

    def make():
        r = pt.Renderer(scene, w, h, settings=st, flags=getattr(pt, flag))
        r.set_sky_cube(sky)
        return r

    r1, r2 = make(), make()
    got, stats = run(r1)
    got2, stats2 = run(r2)
    r1.close()
    r2.close()
    assert np.array_equal(got.view(np.uint32), got2.view(np.uint32)) and stats == stats2
    for k in range(12):
        for f in range(n):
            want, _, _, ost = osc.render(1 + k * n + f, want)
            rays += ost["rays"]
    moved = pt.Scene(scene.spheres, scene.triangles, pt.Camera.from_yaw_pitch((0.8, 1.1, 4.2), -92.0, -3.0))
    osc2 = po.from_scene(moved, w, h, st, sky)
    for k in range(3):
        for f in range(n):
            want, _, _, ost = osc2.render(1000 + k * n + f, want)
            rays += ost["rays"]
    assert stats["rays"] == rays and stats["frames"] == 15 * n and (got[..., 3] == 15 * n).all()
    np.testing.assert_allclose(got, want, rtol=15 * n * 2.0 ** -23, atol=1e-6)


def test_regrouped_short_launches(pt, po, scenes):
    """PT_FLAG_REGROUP_SHORT: launches of 4 to 15 frames are split into items of several frames like longer ones -- the
    same samples and ray counts as the oracle's frame-after-frame sums, equal up to fp32 summation order, reproducible
    run to run; without the flag such a launch stays bit-identical to separate dispatches; launches of 2 and 3 frames
    are exact either way."""
    scene = scenes.bunny(n=12)
    sky = scenes.sky_cube_direct(16)
    st = pt.Settings(1, 6, 1e-4, 1000.0)
    w, h = 160, 90
    osc = po.from_scene(scene, w, h, st, sky)
    for n in (3, 4, 9, 15):
        want = np.zeros((h, w, 4), np.float32)
        rays = 0
        for f in range(n):
            want, _, _, s = osc.render(50 + f, want)
            rays += s["rays"]
        got = []
        for flags in (0, pt.PT_FLAG_REGROUP_SHORT, pt.PT_FLAG_REGROUP_SHORT):
            r = pt.Renderer(scene, w, h, settings=st, flags=flags)
            r.set_sky_cube(sky)
            r.render_frames(50, 1, n)
            got.append(r.read_sum())
            assert r.stats()["rays"] == rays and (got[-1][..., 3] == n).all()
            r.close()
        assert np.array_equal(got[0].view(np.uint32), want.view(np.uint32)), f"{n} frames without the flag: not the sequential sums"
        assert np.array_equal(got[1].view(np.uint32), got[2].view(np.uint32)), f"{n} frames regrouped: not reproducible"
        if n < 4:
            assert np.array_equal(got[1].view(np.uint32), want.view(np.uint32))
        else:
            np.testing.assert_allclose(got[1], want, rtol=n * 2.0 ** -23, atol=1e-6)


def test_rebuilt_bvh_flag(pt, po, scenes):
    """PT_FLAG_REBUILD_BVH (opt-in, outside the bit-for-bit contract): the library walks a tree of its own over the same
    triangles.  Against the oracle on the reference's tree: primary-hit ids may differ in at most 1e-5 of the pixels
    (ties and grazing rays), the converged mean by at most 1e-3 RMSE; and the walks must be shorter."""
    scene = scenes.bunny(n=24)
    sky = scenes.sky_cube_direct(16)
    st = pt.Settings(1, 8, 1e-4, 1000.0)
    w, h, frames = 320, 180, 24
    osc = po.from_scene(scene, w, h, st, sky)
    want, prim, tt, s = osc.render(1, want_aov=True)
    for f in range(1, frames):
        osc.render(1 + f, want)
    r = pt.Renderer(scene, w, h, settings=st, flags=pt.PT_FLAG_AOV | pt.PT_FLAG_REBUILD_BVH | pt.PT_FLAG_DETAILED_STATS)
    r.set_sky_cube(sky)
    r.render(1)
    gprim, gt = r.read_aov()
    st1 = r.stats()
    assert (gprim != prim).sum() <= max(1, int(1e-5 * w * h))
    same = gprim == prim
    assert np.array_equal(gt[same].view(np.uint32), tt[same].view(np.uint32))
    assert st1["rays"] == s["rays"]
    assert st1["interior"] + st1["tri_tests"] < 0.9 * (s["interior"] + s["tri_tests"])
    r.render_frames(2, 1, frames - 1)
    got = r.read_mean()[..., :3]
    ref = want[..., :3] / frames
    rmse = float(np.sqrt(np.mean((got - ref) ** 2)))
    assert rmse <= 1e-3, rmse
    # the flag acts when the mesh is set: taking it away and setting the mesh again gives the reference's tree back
    r.set_flags(pt.PT_FLAG_AOV)
    r.set_mesh(scene.triangles, scene.nodes, scene.triangle_indices)
    r.reset_ready_samples()
    r.render(1)
    assert np.array_equal(r.read_sum().view(np.uint32), osc.render(1)[0].view(np.uint32))
    r.close()
