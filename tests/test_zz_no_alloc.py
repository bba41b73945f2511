"""pt_render_frames_static -- the entry point that allocates nothing -- against pt_render_frames through the C ABI:
the same launches bit for bit on two calls in a row (the second smaller, so that the shared static array still holds
the first call's tables), with another context dirtying the array in between, on a tree deep enough to need the
spill space; device memory in use must not move during the static calls; a launch beyond the array is refused."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _free_bytes():
    import torch
    return torch.cuda.mem_get_info(0)[0]


def _pair(pt, scene, w, h, st, sky):
    rs = []
    for _ in range(2):
        r = pt.Renderer(scene, w, h, settings=st, flags=pt.PT_FLAG_AOV)
        r.set_sky_cube(sky)
        rs.append(r)
    return rs


def test_static_entry_matches_render_frames(pt, scenes):
    scene = scenes.bunny(n=24)          # 6912 triangles, tree deeper than either kernel's on-chip stack
    small = scenes.cornell()
    sky = scenes.sky_cube_direct(32)
    st = pt.Settings(1, 8, 1e-4, 1000.0)
    w, h = 192, 108
    ref, sta = _pair(pt, scene, w, h, st, sky)
    other = pt.Renderer(small, 64, 40, settings=pt.Settings(2, 4, 1e-4, 1000.0))
    other.set_sky_cube(sky)
    assert ref.stats()["bvh_depth"] > 8
    # every kernel is loaded and the usual entry point has grown its buffers before memory is watched
    warm = pt.Renderer(scene, w, h, settings=st)
    warm.set_sky_cube(sky)
    warm.render_frames(1, 1, 40)
    warm.render_frames_static(1, 1, 40)
    warm.render(3)
    warm.sync()
    warm.close()

    calls = [(1, 1, 40), (100, 1, 20), (7, 2, 9), (500, 0, 1)]   # planned, planned and smaller, one item per tile and frame, one frame
    for k, c in enumerate(calls):
        ref.render_frames(*c)
        want = ref.read_sum().copy()
        free0 = _free_bytes()
        sta.render_frames_static(*c)
        got = sta.read_sum().copy()
        assert abs(_free_bytes() - free0) <= (2 << 20), "device memory in use moved during a static call"
        same = _bits(got) == _bits(want)
        assert same.all(), f"call {k} {c}: {int((~same).sum())} of {same.size} values differ"
        assert sta.stats() == ref.stats()
        if k == 0:
            pr, tr = ref.read_aov()
            ps, ts = sta.read_aov()
            assert np.array_equal(pr, ps) and np.array_equal(_bits(tr), _bits(ts))
            # another context's launch overwrites the shared array between the two calls
            other.render_frames_static(9, 1, 24)
            other.sync()
    # the other context's image is what its own usual entry point gives
    o2 = pt.Renderer(small, 64, 40, settings=pt.Settings(2, 4, 1e-4, 1000.0))
    o2.set_sky_cube(sky)
    o2.render_frames(9, 1, 24)
    assert (_bits(other.read_sum()) == _bits(o2.read_sum())).all()
    for r in (ref, sta, other, o2):
        r.close()


def test_static_entry_with_spilling_traversal_stacks(pt, scenes):
    """A chain-shaped tree (60 sheets stacked ever closer to z = 0), deeper than both kernels' on-chip stacks: the
    spill space comes from the static array too."""
    n = 60
    tris = np.zeros(n, pt.TRIANGLE_DTYPE)
    z = -8.0 * 2.0 ** -np.arange(n, dtype=np.float64)
    tris["a"] = np.stack([-1 * np.ones(n), -1 * np.ones(n), z], 1)
    tris["b"] = np.stack([3 * np.ones(n), -1 * np.ones(n), z], 1)
    tris["c"] = np.stack([-1 * np.ones(n), 3 * np.ones(n), z], 1)
    tris["na"] = tris["nb"] = tris["nc"] = (0, 0, 1)
    tris["albedo"] = 1.0
    scene = pt.Scene([pt.Sphere((0.5, 0.5, 1.0), 0.2, (0.8, 0.8, 0.0))], tris, pt.Camera(origin=(0.5, 0.5, 2.0), vfov=50.0))
    st = pt.Settings(1, 6, 1e-4, 1000.0)
    ref, sta = _pair(pt, scene, 96, 64, st, scenes.sky_cube_direct(8))
    assert ref.stats()["bvh_depth"] > 40
    for c in [(1, 1, 24), (50, 1, 3), (9, 0, 1)]:
        ref.render_frames(*c)
        sta.render_frames_static(*c)
        assert (_bits(sta.read_sum()) == _bits(ref.read_sum())).all(), c
        assert sta.stats() == ref.stats()
    ref.close()
    sta.close()


def test_static_entry_refuses_what_does_not_fit(pt, scenes):
    scene = scenes.bunny(n=8)
    r = pt.Renderer(scene, 1920, 1080, settings=pt.Settings(1, 4, 1e-4, 1000.0), flags=pt.PT_FLAG_POOL_KERNEL)
    with pytest.raises(pt.PtError) as e:
        r.render_frames_static(1, 1, 512)       # ~350 000 work items: beyond the 192 MiB array
    assert e.value.code == -5 and "static scratch" in str(e.value)
    assert r.stats()["frames"] == 0 and r.stats()["rays"] == 0
    r.render_frames_static(1, 1, 16)            # a launch that fits still works afterwards
    assert r.stats()["frames"] == 16 and (r.read_sum()[..., 3] == 16).all()
    r.set_window(4)
    with pytest.raises(pt.PtError) as e:
        r.render_frames_static(1, 1, 2)
    assert e.value.code == -4
    r.close()
