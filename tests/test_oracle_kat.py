"""CPU tests of the oracle itself: the integer facts derivable from the reference's source
(HybridTaus known answers, struct layouts, BVH invariants) and analytic intersection cases.

The reference ships no tests or golden vectors (parity unpinned); these are the only pins."""
import math

import numpy as np
import pytest

# (x, y, time) -> four draws (u32 xor of the state, IEEE bits of the value), state after four.
# Computed from the integer definition in shaders/include/utils.wgsl:33-58.
TAUS_KAT = [
    ((0, 0, 0), [(0x3c6ef35f, 0x3e71bbcd), (0x47502932, 0x3e8ea052), (0xd1ccf6e9, 0x3f51ccf7), (0xaaf95334, 0x3f2af953)],
     (0, 0, 0, 0xaaf95334)),
    ((100, 200, 7), [(0x503e3b63, 0x3ea07c77), (0x69e31cd9, 0x3ed3c63a), (0x3c400943, 0x3e710025), (0x550bd172, 0x3eaa17a3)],
     (0xce4001, 0xc80000, 0x1860000, 0x548b9173)),
    ((1919, 1079, 123456), [(0xacedb3a3, 0x3f2cedb4), (0x6fc0f24e, 0x3edf81e5), (0xf36ff4bf, 0x3f736ff5), (0xb2a1de21, 0x3f32a1de)],
     (0x8e8be01d, 0x4300000, 0xb85b03a7, 0x80413d9b)),
    ((639, 359, 1), [(0x650183ac, 0x3eca0307), (0x9096295e, 0x3f109629), (0x196d90db, 0x3dcb6c87), (0x71f51710, 0x3ee3ea2e)],
     (0x84dbe009, 0x1600000, 0xa280005, 0xfe66f71c)),
    ((960, 540, 0xFFFFFFF0), [(0x7737f477, 0x3eee6fe9), (0x0ea8b719, 0x3d6a8b72), (0x8c6f6358, 0x3f0c6f63), (0x1b205008, 0x3dd90280)],
     (0x7bc000f, 0x2180000, 0xefb0007, 0x107f5000)),
]


@pytest.mark.parametrize("seed,draws,final", TAUS_KAT)
def test_hybrid_taus_known_answers(po, seed, draws, final):
    got, state = po.hybrid_taus(*seed, 4)
    assert got == draws
    assert tuple(state) == final


def test_hybrid_taus_value_range_is_closed(po):
    # u >= 2^32 - 128 rounds to 2^32 in f32, so the value reaches exactly 1.0
    st = np.array([0xFFFFFFFF, 0, 0, 0], np.uint32)
    u = np.float32(0xFFFFFF80)
    assert np.float32(2.3283064365387e-10) * u == np.float32(1.0)


def test_struct_layouts(pt, po):
    assert pt.TRIANGLE_DTYPE.itemsize == 112 and pt.NODE_DTYPE.itemsize == 32
    assert pt.SPHERE_DATA_DTYPE.itemsize == 8208 and pt.CAMERA_DTYPE.itemsize == 64 and pt.SETTINGS_DTYPE.itemsize == 16
    assert pt.TRIANGLE_DTYPE.fields["na"][1] == 48 and pt.TRIANGLE_DTYPE.fields["albedo"][1] == 96
    assert pt.TRIANGLE_DTYPE.fields["material"][1] == 108
    assert pt.NODE_DTYPE.fields["left"][1] == 12 and pt.NODE_DTYPE.fields["count"][1] == 28
    assert pt.CAMERA_DTYPE.fields["vfov"][1] == 28 and pt.CAMERA_DTYPE.fields["up"][1] == 48
    assert po.TRIANGLE_DTYPE == pt.TRIANGLE_DTYPE and po.NODE_DTYPE == pt.NODE_DTYPE


def test_canonical_math_accuracy(po):
    """The fixed sin/cos/atan2/asin are within a few ulp of the true functions (they need only be a
    valid WGSL implementation; what matters is that host and device evaluate the same sequence)."""
    L = po.lib()
    xs = np.concatenate([np.linspace(-100, 100, 20001), np.linspace(-3e4, 3e4, 5001)]).astype(np.float32)
    s = np.array([L.pto_sinf(float(x)) for x in xs], np.float32)
    c = np.array([L.pto_cosf(float(x)) for x in xs], np.float32)
    assert np.abs(s - np.sin(xs.astype(np.float64))).max() < 3e-7
    assert np.abs(c - np.cos(xs.astype(np.float64))).max() < 3e-7
    rng = np.random.RandomState(0)
    yx = rng.uniform(-2, 2, (5000, 2)).astype(np.float32)
    a = np.array([L.pto_atan2f(float(y), float(x)) for y, x in yx], np.float32)
    assert np.abs(a - np.arctan2(yx[:, 0].astype(np.float64), yx[:, 1])).max() < 1e-6
    zs = np.linspace(-1, 1, 4001).astype(np.float32)
    a = np.array([L.pto_asinf(float(z)) for z in zs], np.float32)
    assert np.abs(a - np.arcsin(zs.astype(np.float64))).max() < 1e-6
    assert L.pto_atan2f(0.0, -1.0) == np.float32(math.pi) and L.pto_atan2f(0.0, 0.0) == 0.0


def test_hash_rand_in_unit_interval(po):
    L = po.lib()
    rng = np.random.RandomState(1)
    v = np.array([L.pto_hash_rand(float(a), float(b)) for a, b in rng.uniform(-50, 50, (4000, 2)).astype(np.float32)])
    assert (v >= 0).all() and (v <= 1).all() and 0.4 < v.mean() < 0.6
    u = np.zeros(3, np.float32)
    for a, b in rng.uniform(-1, 1, (200, 2)).astype(np.float32):
        L.pto_random_unit(float(a), float(b), u.ctypes.data)
        assert abs(float(np.linalg.norm(u.astype(np.float64))) - 1.0) < 1e-5


def _scene(pt, po, spheres=(), tris=None, w=4, h=4, t_min=1e-4, t_max=1000.0):
    tris = np.zeros(0, pt.TRIANGLE_DTYPE) if tris is None else tris
    sc = pt.Scene(list(spheres), tris, pt.Camera())
    return po.from_scene(sc, w, h, pt.Settings(1, 4, t_min, t_max))


def test_ray_sphere_roots(pt, po):
    s = [pt.Sphere((0, 0, -5), 1.0, (1, 1, 1), pt.Material.Diffuse)]
    osc = _scene(pt, po, s)
    prim, t, p, n, front = osc.trace_rays([[0, 0, 0], [0, 0, -5], [0, 0, 0], [0, 0, 0]],
                                          [[0, 0, -1], [0, 0, -1], [0, 0, -2], [0, 1, 0]])
    assert prim[0] == 0x80000000 and t[0] == np.float32(4.0) and front[0] == 1 and np.allclose(n[0], [0, 0, 1])
    # origin inside: near root is negative, far root is taken, normal faces the ray
    assert prim[1] == 0x80000000 and t[1] == np.float32(1.0) and front[1] == 0 and np.allclose(n[1], [0, 0, 1])
    # un-normalised direction: t is in units of |d|
    assert t[2] == np.float32(2.0)
    assert prim[3] == 0xFFFFFFFF


def test_sphere_order_and_t_window(pt, po):
    s = [pt.Sphere((0, 0, -5), 1.0, (1, 0, 0)), pt.Sphere((0, 0, -5), 1.0, (0, 1, 0)), pt.Sphere((0, 0, -3), 0.5, (0, 0, 1))]
    osc = _scene(pt, po, s)
    prim, t, *_ = osc.trace_rays([[0, 0, 0]], [[0, 0, -1]])
    assert prim[0] == (0x80000000 | 2) and t[0] == np.float32(2.5)
    # identical spheres: strict '<' keeps the first
    osc = _scene(pt, po, s[:2])
    assert osc.trace_rays([[0, 0, 0]], [[0, 0, -1]])[0][0] == 0x80000000
    # t_max excludes the hit (open interval)
    osc = _scene(pt, po, s[:1], t_max=4.0)
    assert osc.trace_rays([[0, 0, 0]], [[0, 0, -1]])[0][0] == 0xFFFFFFFF


def test_gizmo_sphere_only_hits_its_rim(pt, po):
    s = [pt.Sphere((0, 0, -5), 1.0, (1, 1, 1), pt.Material.Gizmo)]
    osc = _scene(pt, po, s)
    prim, *_ = osc.trace_rays([[0, 0, 0], [0.995, 0, 0]], [[0, 0, -1], [0, 0, -1]])
    assert prim[0] == 0xFFFFFFFF       # head-on: |dot(d, n)| = 1 > 0.2
    assert prim[1] == 0x80000000       # grazing the silhouette


def _tri(pt, a, b, c, n=(0, 0, 1)):
    t = np.zeros(1, pt.TRIANGLE_DTYPE)
    t["a"], t["b"], t["c"] = a, b, c
    t["na"] = t["nb"] = t["nc"] = n
    t["albedo"] = 1.0
    return t


def test_moller_trumbore_edges(pt, po):
    tri = _tri(pt, (0, 0, -2), (1, 0, -2), (0, 1, -2))
    osc = _scene(pt, po, tris=tri)
    o = [[0.25, 0.25, 0], [0.0, 0.5, 0], [0.5, 0.5, 0], [0.75, 0.75, 0], [-0.01, 0.5, 0], [0.25, 0.25, -4]]
    d = [[0, 0, -1]] * 5 + [[0, 0, 1]]
    prim, t, p, n, front = osc.trace_rays(o, d)
    assert prim[0] == 0 and t[0] == np.float32(2.0) and front[0] == 1 and np.allclose(n[0], [0, 0, 1])
    assert prim[1] == 0            # u == 0 is inside
    assert prim[2] == 0            # u + v == 1 is inside
    assert prim[3] == 0xFFFFFFFF and prim[4] == 0xFFFFFFFF
    assert prim[5] == 0 and front[5] == 0 and np.allclose(n[5], [0, 0, -1])   # two-sided, normal flipped


def test_small_determinant_is_rejected(pt, po):
    # |det| < 1e-5 rejects: a 1e-3-sized triangle with a unit direction has det ~1e-6
    tri = _tri(pt, (0, 0, -2), (1e-3, 0, -2), (0, 1e-3, -2))
    osc = _scene(pt, po, tris=tri)
    assert osc.trace_rays([[2e-4, 2e-4, 0]], [[0, 0, -1]])[0][0] == 0xFFFFFFFF
    # the same triangle is found when the direction is long enough
    assert osc.trace_rays([[2e-4, 2e-4, 0]], [[0, 0, -100]])[0][0] == 0


def test_hit_aabb_cases(po):
    L = po.lib()
    f = lambda *v: np.array(v, np.float32)
    box = (f(-1, -1, -3), f(1, 1, -2))
    call = lambda o, d: L.pto_hit_aabb(o.ctypes.data, d.ctypes.data, box[0].ctypes.data, box[1].ctypes.data)
    assert call(f(0, 0, 0), f(0, 0, -1)) == np.float32(2.0)
    assert call(f(0, 0, -2.5), f(0, 0, -1)) == np.float32(-0.5)       # inside: negative entry distance
    assert call(f(0, 0, -5), f(0, 0, -1)) == np.float32(99999.0)      # behind
    assert call(f(3, 0, 0), f(0, 0, -1)) == np.float32(99999.0)       # beside (0-component -> inf slabs)


def test_bvh_invariants(pt, po, scenes):
    for tris in (scenes.cornell().triangles, scenes.blob_mesh(6)):
        nodes, idx = po.bvh_build(tris)
        n = len(tris)
        assert sorted(idx.tolist()) == list(range(n)) and len(nodes) <= 2 * n - 1
        covered = np.zeros(n, int)
        for i, nd in enumerate(nodes):
            if nd["count"] == 0:
                l = int(nd["left"])
                assert l > i and l + 1 < len(nodes)
                for c in (l, l + 1):      # children are inside the parent
                    assert (nodes[c]["min"] >= nd["min"]).all() and (nodes[c]["max"] <= nd["max"]).all()
            else:
                sl = idx[nd["left"]: nd["left"] + nd["count"]]
                covered[nd["left"]: nd["left"] + nd["count"]] += 1
                v = np.concatenate([tris[sl]["a"], tris[sl]["b"], tris[sl]["c"]])
                assert np.array_equal(v.min(0), nd["min"]) and np.array_equal(v.max(0), nd["max"])
        assert (covered == 1).all()


def test_bvh_small_cases(pt, po):
    one = _tri(pt, (0, 0, 0), (1, 0, 0), (0, 1, 0))
    nodes, idx = po.bvh_build(one)
    assert len(nodes) == 1 and nodes[0]["count"] == 1 and idx.tolist() == [0]
    three = np.concatenate([_tri(pt, (k, 0, 0), (k + 0.5, 0, 0), (k, 0.5, 0)) for k in range(3)])
    nodes, idx = po.bvh_build(three)
    assert nodes[0]["count"] == 0 and nodes[0]["left"] == 1 and len(nodes) == 3
    # identical centroids on the far side of the midpoint: the left side stays empty, the node a leaf
    far = _tri(pt, (0, 0, 0), (1, 0, 0), (1, 0.1, 0))
    nodes, idx = po.bvh_build(np.concatenate([far] * 5))
    assert len(nodes) == 1 and nodes[0]["count"] == 5
    # identical centroids on the near side: the partition loop stops at i == j without looking at
    # the last element (bvh.rs:115-124), so one triangle always peels off to the right
    nodes, idx = po.bvh_build(np.concatenate([one] * 5))
    assert len(nodes) == 7 and [int(n["count"]) for n in nodes if n["count"]] == [1, 1, 2, 1]
    nodes, idx = po.bvh_build(np.zeros(0, pt.TRIANGLE_DTYPE))
    assert len(nodes) == 0 and len(idx) == 0


def test_sky_lookup_faces(po, scenes):
    size = 8
    cube = np.zeros((6, size, size, 4), np.float32)
    for f in range(6):
        cube[f, :, :, 0] = f
        cube[f, :, :, 1] = np.arange(size)[None, :]       # s index
        cube[f, :, :, 2] = np.arange(size)[:, None]       # t index
    L = po.lib()
    out = np.zeros(3, np.float32)
    look = lambda *d: (L.pto_sky_lookup(cube.ctypes.data, size, np.array(d, np.float32).ctypes.data, out.ctypes.data), out.copy())[1]
    assert look(1, 0, 0)[0] == 0 and look(-1, 0, 0)[0] == 1 and look(0, 1, 0)[0] == 2
    assert look(0, -1, 0)[0] == 3 and look(0, 0, 1)[0] == 4 and look(0, 0, -1)[0] == 5
    assert look(1, 1, 1)[0] == 4 and look(1, 1, 0)[0] == 2      # ties: z, then y, then x
    # +Z face: s grows with x, t grows with -y
    assert look(-0.9, 0.9, 1).tolist() == [4, 0, 0] and look(0.9, -0.9, 1).tolist() == [4, size - 1, size - 1]
    assert look(0.999999, 0, 1)[1] == size - 1                  # clamp to edge


def test_depth_exhaustion_returns_throughput_and_absorption_sees_sky(pt, po):
    """compute.wgsl:206 returns the accumulated colour (not black) when depth runs out."""
    s = [pt.Sphere((0, 0, 0), 50.0, (0.5, 0.5, 0.5), pt.Material.Metal)]      # camera inside a mirror ball
    sc = pt.Scene(s, np.zeros(0, pt.TRIANGLE_DTYPE), pt.Camera(origin=(0, 0, 0)))
    st = pt.Settings(1, 3, 1e-4, 1000.0)
    sky = np.full((6, 2, 2, 4), 7.0, np.float32)
    acc, *_ = po.from_scene(sc, 8, 8, st, sky).render(3)
    vals = np.unique(np.round(acc[..., 0], 6))
    # 0.5^3 = depth exhausted; 0.5^k * 7 (k<=3) = reflected ray pointing into the surface
    assert set(vals.tolist()) <= {0.125, 3.5, 1.75, 0.875}
    assert 0.125 in vals.tolist()
    assert (acc[..., 3] == 1).all()


def test_progressive_window_bookkeeping(po):
    """renderer.rs:539-541, 560, 623-645 by hand: ready_samples starts at 0, the copy pass uses the count from
    before the increment, reset_ready_samples sets 1; texels are round(clamp(c) * 255) and the mean is over 8-bit values."""
    win = po.ProgressiveWindow(3)
    shown = []
    for k in range(5):
        win.frame(np.full((1, 2, 4), 0.1 * (k + 1), np.float32))
        shown.append(win.shown)
    assert shown == [0, 1, 2, 3, 3]
    # newest three frames hold 0.5, 0.4, 0.3 -> texels 128 (127.5 rounds to even), 102, 76 (76.5 -> 76)
    want = (np.float32(128) / np.float32(255) + np.float32(102) / np.float32(255) + np.float32(76) / np.float32(255)) / np.float32(3)
    assert (win.displayed() == want).all()
    win.reset_ready_samples()
    win.frame(np.array([[[2.0, -1.0, np.nan, 1.0]]], np.float32))
    assert win.shown == 1 and (win.displayed() == np.array([1.0, 0.0, 0.0, 1.0], np.float32)).all()
    fresh = po.ProgressiveWindow(4)
    fresh.frame(np.zeros((1, 1, 4), np.float32))
    with pytest.raises(ValueError):
        fresh.displayed()
