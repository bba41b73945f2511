"""More HIP-vs-oracle parity through the C ABI: explicit ray queries, every material, deep trees
whose stack spills past LDS, the equirect->cube kernel, read-outs and input validation."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _exact(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)), \
        f"{what}: {int((a.view(np.uint32) != b.view(np.uint32)).sum())} of {a.size} values differ"


def _pair(pt, po, scene, w, h, st, sky, flags=0):
    r = pt.Renderer(scene, w, h, settings=st, flags=pt.PT_FLAG_AOV | flags)
    if sky is not None:
        r.set_sky_cube(sky)
    return r, po.from_scene(scene, w, h, st, sky)


def test_trace_rays_matches_oracle(pt, po, scenes):
    scene = scenes.bunny(n=12)
    scene.spheres.append(pt.Sphere((0.5, 0.3, 0.5), 0.3, (1, 1, 1), pt.Material.Dielectric))
    st = pt.Settings(1, 4, 1e-4, 1000.0)
    r, osc = _pair(pt, po, scene, 16, 16, st, None)
    rng = np.random.RandomState(5)
    n = 3000
    o = rng.uniform(-2, 2, (n, 3)).astype(np.float32) + np.array([0, 1.5, 2], np.float32)
    d = (np.array([0, 0.5, -1], np.float32) - o + rng.normal(0, 0.6, (n, 3))).astype(np.float32)
    d[::7] *= 0.01          # un-normalised directions scale t
    d[5::11, 0] = 0.0       # axis-parallel components -> infinite inverse direction
    gp, gt, gpos, gn, gf = r.trace_rays(o, d)
    op, ot, opos, on, of = osc.trace_rays(o, d)
    assert np.array_equal(gp, op), f"{int((gp != op).sum())} primitive ids differ"
    hit = op != 0xFFFFFFFF
    assert hit.sum() > 200 and (op[hit] < 0x80000000).sum() > 50
    _exact(gt[hit], ot[hit], "t")
    _exact(gpos[hit], opos[hit], "hit point")
    _exact(gn[hit], on[hit], "normal")
    assert np.array_equal(gf[hit], of[hit])
    r.close()


def test_all_materials_and_spp_gt_1(pt, po, scenes):
    """Metal, dielectric, gizmo and an out-of-range material id, several samples per dispatch."""
    S, M = pt.Sphere, pt.Material
    spheres = [S((0, 0, -1), 0.5, (0.8, 0.3, 0.3), M.Diffuse), S((1, 0, -1), 0.5, (1, 1, 1), M.Dielectric),
               S((-1, 0, -1), 0.5, (0.8, 0.6, 0.2), M.Metal), S((0, 1.0, -1), 0.4, (0.2, 0.9, 0.2), M.Gizmo),
               S((0, -100.5, -1), 100.0, (0.8, 0.8, 0.0), M.Diffuse), S((2.2, 0.2, -1.5), 0.7, (0.3, 0.3, 0.9), M.Diffuse)]
    cam = pt.Camera.from_yaw_pitch((0.0, 0.6, 1.8), -90.0, -8.0)
    scene = pt.Scene(spheres, scenes.cornell().triangles, cam)
    sky = scenes.sky_cube_direct(16)
    st = pt.Settings(3, 12, 1e-4, 1000.0)
    r, osc = _pair(pt, po, scene, 128, 72, st, sky)
    # material id 7 takes the shader's default branch (compute.wgsl:196-200)
    buf = pt.pack_spheres(spheres)
    buf["spheres"][0][5]["material"] = 7.0
    r._ck(r._l.pt_set_spheres(r._ctx, buf.ctypes.data))
    osc.spheres[:] = buf.view(np.uint8).reshape(-1)
    acc = np.zeros((72, 128, 4), np.float32)
    rays = 0
    for t in (11, 12):
        r.render(t)
        acc, prim, tt, s = osc.render(t, acc, want_aov=True)
        rays += s["rays"]
    gprim, gt = r.read_aov()
    assert np.array_equal(gprim, prim)
    assert {0x80000001, 0x80000002, 0x80000003, 0x80000005} <= set(np.unique(prim).tolist())
    _exact(gt, tt, "primary t")
    _exact(r.read_sum(), acc, "radiance")
    assert r.stats()["rays"] == rays
    r.close()


def test_deep_tree_spills_past_lds_stack(pt, po, scenes):
    """Sheets stacked ever closer to z = 0 give the midpoint builder a chain (each split peels off
    the farthest sheet); a ray towards them enters the near child and pushes the far one at every
    level, so the stack grows to the depth of the tree -- far past its LDS part."""
    n = 60
    tris = np.zeros(n, pt.TRIANGLE_DTYPE)
    z = -8.0 * 2.0 ** -np.arange(n, dtype=np.float64)
    tris["a"] = np.stack([-1 * np.ones(n), -1 * np.ones(n), z], 1)
    tris["b"] = np.stack([3 * np.ones(n), -1 * np.ones(n), z], 1)
    tris["c"] = np.stack([-1 * np.ones(n), 3 * np.ones(n), z], 1)
    tris["na"] = tris["nb"] = tris["nc"] = (0, 0, 1)
    tris["albedo"] = 1.0
    scene = pt.Scene([pt.Sphere((0.5, 0.5, 1.0), 0.2, (0.8, 0.8, 0.0))], tris,
                     pt.Camera(origin=(0.5, 0.5, 2.0), vfov=50.0))
    st = pt.Settings(1, 6, 1e-4, 1000.0)
    r, osc = _pair(pt, po, scene, 96, 64, st, scenes.sky_cube_direct(8))
    assert r.stats()["bvh_depth"] > 40
    acc = np.zeros((64, 96, 4), np.float32)
    for t in (1, 2, 3):
        r.render(t)
        acc, prim, tt, s = osc.render(t, acc, want_aov=True)
    assert s["max_stack"] > 24, "scene does not exercise the spill path"
    gprim, gt = r.read_aov()
    assert np.array_equal(gprim, prim) and (prim < 0x80000000).sum() > 20
    _exact(r.read_sum(), acc, "radiance")
    # many frames in one call: planned work items with spilling stacks; sums differ from the
    # frame-by-frame ones by summation order only
    r.reset_ready_samples()
    r.render_frames(1, 1, 20)
    acc = np.zeros((64, 96, 4), np.float32)
    for t in range(1, 21):
        osc.render(t, acc)
    got = r.read_sum()
    assert np.array_equal(got[..., 3], acc[..., 3])
    np.testing.assert_allclose(got, acc, rtol=20 * 2.0 ** -23, atol=1e-6)
    r.close()


def test_sphere_only_and_empty_scenes(pt, po, scenes):
    sky = scenes.sky_cube_direct(8)
    st = pt.Settings(2, 5, 1e-4, 1000.0)
    for spheres in ([], scenes.three_balls().spheres):
        scene = pt.Scene(list(spheres), np.zeros(0, pt.TRIANGLE_DTYPE), scenes.three_balls().camera)
        r, osc = _pair(pt, po, scene, 64, 40, st, sky)
        r.render(9)
        acc, prim, tt, s = osc.render(9, want_aov=True)
        gprim, _ = r.read_aov()
        assert np.array_equal(gprim, prim)
        _exact(r.read_sum(), acc, "radiance")
        r.close()


def test_depth_zero_and_no_sky(pt, po, scenes):
    scene = scenes.cornell()
    for depth in (0, 1):
        st = pt.Settings(2, depth, 1e-4, 1000.0)
        r, osc = _pair(pt, po, scene, 32, 20, st, None)
        r.render(4)
        acc, *_ = osc.render(4)
        _exact(r.read_sum(), acc, f"depth {depth}")
        r.close()


def test_equirect_to_cube_matches_oracle(pt, po, scenes):
    pano = scenes.sky_equirect(256, 128)
    pano[..., 0] += np.arange(256, dtype=np.float32)[None, :] / 256.0      # make columns distinguishable
    scene = scenes.three_balls()
    r = pt.Renderer(scene, 32, 20, settings=pt.Settings(1, 4, 1e-4, 1000.0))
    r.set_sky_equirect(pano, 64)
    got = r.read_sky_cube()
    want = po.equirect_to_cube(pano, 64)
    _exact(got, want, "cube texels")
    # and the renderer uses it
    r.render(2)
    acc, *_ = po.from_scene(scene, 32, 20, pt.Settings(1, 4, 1e-4, 1000.0), want).render(2)
    _exact(r.read_sum(), acc, "radiance with generated cube")
    r.close()


def test_hdr_file_to_sky_cube_matches_oracle(pt, po, scenes):
    """The whole of CubeTexture::from_equirectangular_hdri (src/texture.rs:172-270): .hdr bytes -> panorama -> cube."""
    pano = scenes.sky_equirect(128, 64)[..., :3].astype(np.float64)
    mant, expo = np.frexp(pano.max(axis=-1))                 # shared exponent of the largest channel
    rgbe = np.zeros((64, 128, 4), np.uint8)
    rgbe[..., :3] = np.clip(pano / np.ldexp(1.0, expo)[..., None] * 256.0, 0, 255).astype(np.uint8)
    rgbe[..., 3] = np.where(pano.max(axis=-1) < 1e-32, 0, expo + 128)
    data = b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 64 +X 128\n" + rgbe.tobytes()  # 128 px rows, flat coding
    decoded = pt.decode_hdr(data)
    want_pano = po.hdr_decode(data)
    _exact(decoded, want_pano, "decoded panorama")
    assert np.abs(decoded[..., :3] - pano).max() <= pano.max() / 128.0
    r = pt.Renderer(scenes.three_balls(), 32, 20, settings=pt.Settings(1, 4, 1e-4, 1000.0))
    r.set_sky_equirect(decoded, 32)
    _exact(r.read_sky_cube(), po.equirect_to_cube(want_pano, 32), "cube texels from the .hdr")
    r.close()


def test_display_window_matches_oracle(pt, po, scenes):
    """ProgressiveRendering + copy.wgsl (renderer.rs:492-516, 613-646): the newest frames as 8-bit textures and
    their mean, through single frames, a fused call, a reset and more frames than the window holds."""
    scene = scenes.cornell()
    sky = scenes.sky_cube_direct(8)
    st = pt.Settings(2, 4, 1e-4, 1000.0)
    w, h = 72, 40  # reference dispatch covers 64 x 32 of it
    flags = pt.PT_FLAG_REFERENCE_DISPATCH
    r = pt.Renderer(scene, w, h, settings=st, flags=flags)
    r.set_sky_cube(sky)
    r.set_window(3)
    osc = po.from_scene(scene, w, h, st, sky, flags)
    win = po.ProgressiveWindow(3)
    acc = np.zeros((h, w, 4), np.float32)

    def oracle_frames(times):
        for t in times:
            frame, *_ = osc.render(t)
            win.frame(frame)
            acc[...] = acc + frame

    def check(what):
        mean, q, n = r.read_window()
        want = win.displayed()
        assert n == win.shown, what
        _exact(mean, want, what + ": displayed mean")
        assert np.array_equal(q, np.rint(np.clip(want, 0, 1) * np.float32(255.0)).astype(np.uint8)), what
        _exact(r.read_sum(), acc, what + ": running sums")

    r.render(5)
    oracle_frames([5])
    assert r.window_frames_shown() == 0
    with pytest.raises(pt.PtError):  # the reference's first frame averages zero textures
        r.read_window()
    r.render(6)
    oracle_frames([6])
    check("second frame")
    assert r.window_frames_shown() == 1
    r.render_frames(10, 3, 4)  # fused call: times 10, 13, 16, 19; the window is full afterwards
    oracle_frames([10, 13, 16, 19])
    check("after a fused call")
    assert r.window_frames_shown() == 3 and (r.read_window()[0][32:] == 0).all()
    r.reset_ready_samples()
    win.reset_ready_samples()
    acc[...] = 0
    r.render(30)
    oracle_frames([30])
    check("first frame after a reset")
    assert r.window_frames_shown() == 1
    r.render_frames(31, 1, 2)
    oracle_frames([31, 32])
    check("window refilling")
    # without the window the sums are the same bits
    r2 = pt.Renderer(scene, w, h, settings=st, flags=flags)
    r2.set_sky_cube(sky)
    r2.render(30)
    r2.render_frames(31, 1, 2)
    _exact(r.read_sum(), r2.read_sum(), "sums with and without a window")
    r2.close()
    r.set_window(0)
    with pytest.raises(pt.PtError):
        r.read_window()
    with pytest.raises(pt.PtError):
        r.set_window(257)
    r.close()


def test_degenerate_deep_tree_is_refused(pt, scenes):
    """5000 copies of one triangle whose centroid lies before the middle of its box: the reference's builder peels
    them off one per level; a tree that deep is refused instead of sizing traversal stacks for it."""
    one = np.zeros(1, pt.TRIANGLE_DTYPE)
    one["a"][0], one["b"][0], one["c"][0] = (0, 0, 0), (1, 0, 0), (0, 0.1, 0)
    one["na"][0] = one["nb"][0] = one["nc"][0] = (0, 0, 1)
    tris = np.repeat(one, 5000)
    nodes, idx = pt.bvh_build(tris)
    assert len(nodes) == 2 * 4999 - 1  # one level per triangle until two are left
    r = pt.Renderer(scenes.three_balls(), 16, 16, settings=pt.Settings(1, 2, 1e-4, 1000.0))
    with pytest.raises(pt.PtError) as e:
        r.set_mesh(tris, nodes, idx)
    assert e.value.code == -5 and "depth" in str(e.value)  # PT_ERR_LIMIT
    r.render(1)  # the context is still usable (spheres only)
    assert r.stats()["rays"] > 0
    r.close()


def _far_cluster(pt, origin, size, copies):
    """`copies` identical triangles whose centroid lies beyond the middle of their box on every axis the builder can
    pick: the partition leaves its left side empty and the node stays one leaf of `copies` triangles (bvh.rs:115-129)."""
    t = np.zeros(1, pt.TRIANGLE_DTYPE)
    o = np.asarray(origin, np.float32)
    t["a"][0], t["b"][0], t["c"][0] = o, o + np.float32(size) * np.array([1, 0, 0], np.float32), o + np.float32(size) * np.array([1, 0.9, 0], np.float32)
    t["na"][0] = t["nb"][0] = t["nc"][0] = (0, 0, 1)
    t["albedo"][0] = (0.7, 0.6, 0.5)
    return np.repeat(t, copies)


def test_leaves_of_more_than_seven_triangles(pt, po, scenes):
    """Leaves too big for a token's inline count (8 and more triangles) are resolved through the side table: primary-hit
    ids (the first of the identical triangles in leaf order wins the tie), t, radiance and ray counts must be the
    oracle's, in single dispatches and in a planned launch; also with such a leaf as the root."""
    cam = pt.Camera.from_yaw_pitch((2.0, 0.4, 2.2), -90.0, 0.0)
    st = pt.Settings(1, 5, 1e-4, 1000.0)
    sky = scenes.sky_cube_direct(16)
    meshes = {
        "mixed": np.concatenate([_far_cluster(pt, (0.0, 0.0, 0.0), 0.8, 9), _far_cluster(pt, (1.2, 0.1, 0.2), 0.7, 20),
                                 _far_cluster(pt, (2.4, 0.0, -0.3), 0.6, 7), _far_cluster(pt, (3.6, 0.2, 0.0), 0.5, 8)]),
        "root": _far_cluster(pt, (1.5, 0.0, 0.0), 1.0, 12),
    }
    for name, tris in meshes.items():
        scene = pt.Scene([pt.Sphere((2.0, -100.0, 0.0), 100.0, (0.5, 0.5, 0.5), pt.Material.Diffuse)], tris, cam)
        counts = sorted(int(c) for c in scene.nodes["count"] if c)
        assert counts == ([7, 8, 9, 20] if name == "mixed" else [12]), counts
        w, h = 96, 64
        r, osc = _pair(pt, po, scene, w, h, st, sky)
        acc = np.zeros((h, w, 4), np.float32)
        rays = 0
        for t in (3, 4):
            r.render(t)
            acc, prim, tt, s = osc.render(t, acc, want_aov=True)
            rays += s["rays"]
        gprim, gt = r.read_aov()
        assert np.array_equal(gprim, prim), (name, int((gprim != prim).sum()))
        assert len(np.unique(prim[prim < 0x80000000])) == (4 if name == "mixed" else 1), "every cluster must be seen, one winner each"
        _exact(gt, tt, name + ": primary t")
        _exact(r.read_sum(), acc, name + ": radiance")
        assert r.stats()["rays"] == rays
        # a planned launch (the pooled kernel's walk in two halves when that kernel is chosen)
        r.reset_ready_samples()
        acc = np.zeros((h, w, 4), np.float32)
        before = r.stats()["rays"]
        rays = 0
        for f in range(16):
            acc, _, _, s = osc.render(20 + f, acc)
            rays += s["rays"]
        r.render_frames(20, 1, 16)
        got = r.read_sum()
        assert r.stats()["rays"] - before == rays
        np.testing.assert_allclose(got, acc, rtol=16 * 2.0 ** -23, atol=1e-6)
        # explicit rays through the plain (one record per step) walk of the query kernel
        rng = np.random.RandomState(11)
        o = (np.array([2.0, 0.4, 3.0], np.float32) + rng.uniform(-0.5, 0.5, (500, 3))).astype(np.float32)
        d = (np.array([2.0, 0.3, 0.0], np.float32) + rng.uniform(-2, 2, (500, 3)) * np.array([1, 0.2, 0.1], np.float32) - o).astype(np.float32)
        gp, gtt, _, gn, _ = r.trace_rays(o, d)
        op, ott, _, on, _ = osc.trace_rays(o, d)
        assert np.array_equal(gp, op)
        hit = op != 0xFFFFFFFF
        _exact(gtt[hit], ott[hit], name + ": query t")
        _exact(gn[hit], on[hit], name + ": query normal")
        r.close()


def test_mesh_beyond_the_walk_table_limit_is_refused(pt, scenes):
    """Node pairs and triangle records share one device array addressed with 32-bit byte offsets: counts that would
    take it past 4 GiB are refused from the counts alone, before any array is read."""
    cube = scenes.cornell()
    r = pt.Renderer(scenes.three_balls(), 16, 16, settings=pt.Settings(1, 2, 1e-4, 1000.0))
    tris = np.ascontiguousarray(cube.triangles, pt.TRIANGLE_DTYPE)
    nodes = np.ascontiguousarray(cube.nodes, pt.NODE_DTYPE)
    idx = np.ascontiguousarray(cube.triangle_indices, np.uint32)
    ptr = lambda a: a.ctypes.data_as(__import__("ctypes").c_void_p)
    for n_nodes, n_idx in (((1 << 27) - 1, 4), (3, 1 << 26), (1 << 26, 1 << 25)):
        rc = r._l.pt_set_mesh(r._ctx, ptr(tris), len(tris), ptr(nodes), n_nodes, ptr(idx), n_idx)
        assert rc == -5, (n_nodes, n_idx, rc)  # PT_ERR_LIMIT
        assert b"walk table" in r._l.pt_last_error(r._ctx)
    r.set_mesh(tris, nodes, idx)  # the context is still usable
    r.render(1)
    assert r.stats()["rays"] > 0
    r.close()


def test_readouts(pt, scenes):
    scene = scenes.three_balls()
    r = pt.Renderer(scene, 64, 36, settings=pt.Settings(1, 4, 1e-4, 1000.0))
    r.set_sky_cube(scenes.sky_cube_direct(8))
    for t in range(4):
        r.render(t)
    s, m, q = r.read_sum(), r.read_mean(), r.read_rgba8()
    assert (s[..., 3] == 4).all() and (m[..., 3] == 1).all()
    np.testing.assert_array_equal(m[..., :3], s[..., :3] / np.float32(4.0))
    want = np.rint(np.clip(m, 0, 1) * np.float32(255.0)).astype(np.uint8)
    assert np.array_equal(q, want)
    r.reset_ready_samples()
    assert not r.read_sum().any() and r.stats()["frames"] == 0 and r.stats()["rays"] == 0
    assert not r.read_mean().any()
    r.close()


def test_bind_accum_to_torch_tensor(pt, scenes):
    import torch
    scene = scenes.three_balls()
    a = pt.Renderer(scene, 64, 36, settings=pt.Settings(1, 4, 1e-4, 1000.0))
    b = pt.Renderer(scene, 64, 36, settings=pt.Settings(1, 4, 1e-4, 1000.0), stream=torch.cuda.current_stream().cuda_stream)
    tens = torch.zeros((36, 64, 4), dtype=torch.float32, device="cuda")
    b.bind_accum(tens.data_ptr())
    assert b.accum_device_ptr() == tens.data_ptr()
    for r in (a, b):
        r.set_sky_cube(scenes.sky_cube_direct(8))
        r.render(1)
        r.render(2)
        r.sync()
    assert np.array_equal(a.read_sum(), tens.cpu().numpy())
    a.close()
    b.close()


def test_malformed_bvh_is_rejected(pt, scenes):
    scene = scenes.cornell()
    r = pt.Renderer(scene, 16, 16)
    nodes = scene.nodes.copy()
    bad = nodes.copy()
    bad["left"][0] = 0                      # child index not after its parent: would loop forever
    with pytest.raises(pt.PtError):
        r.set_mesh(scene.triangles, bad, scene.triangle_indices)
    bad = nodes.copy()
    bad["left"][0] = len(nodes)             # children out of range
    with pytest.raises(pt.PtError):
        r.set_mesh(scene.triangles, bad, scene.triangle_indices)
    leaf = int(np.nonzero(nodes["count"])[0][0])
    bad = nodes.copy()
    bad["count"][leaf] = 1000               # leaf range past the index array
    with pytest.raises(pt.PtError):
        r.set_mesh(scene.triangles, bad, scene.triangle_indices)
    assert nodes["count"][1] == 0 and nodes["count"][2] != 0, "test assumes node 1 interior, node 2 leaf"
    bad = nodes.copy()
    bad["count"][2] = 0
    bad["left"][2] = nodes["left"][1]       # nodes 1 and 2 share a pair of children: not a tree
    with pytest.raises(pt.PtError):
        r.set_mesh(scene.triangles, bad, scene.triangle_indices)
    idx = scene.triangle_indices.copy()
    idx[0] = 99                             # triangle index out of range
    with pytest.raises(pt.PtError):
        r.set_mesh(scene.triangles, nodes, idx)
    r.set_mesh(scene.triangles, nodes, scene.triangle_indices)      # still usable afterwards
    r.render(1)
    assert r.read_sum()[..., 3].min() == 1
    st = pt.Settings(0, 4, 1e-4, 1000.0)
    r.set_settings(st)
    with pytest.raises(pt.PtError):
        r.render(1)
    r.close()


def test_reference_dispatch_smaller_than_one_group(pt, scenes):
    """PT_FLAG_REFERENCE_DISPATCH renders floor(w/16) x floor(h/16) groups of 16x16 pixels (renderer.rs:565-569): an
    image lower than 16 pixels gets none, in single dispatches, short fused launches and planned ones alike."""
    r = pt.Renderer(scenes.cornell(), 58, 10, settings=pt.Settings(2, 3, 1e-4, 1000.0), flags=pt.PT_FLAG_REFERENCE_DISPATCH)
    r.render(5)
    r.render_frames(6, 3, 13)
    r.render_frames(100, 1, 40)
    assert (r.read_sum() == 0).all() and r.stats()["rays"] == 0 and r.stats()["frames"] == 54
    r.close()


def test_camera_ray_culling_rectangles(pt, po, scenes):
    """Camera rays of pixels outside the screen rectangles of the root's two child boxes skip the root visit
    (it is known to end the walk).  Cameras that stress the rectangles: far away (mesh a few pixels wide), close and
    off-centre (a box leaves the screen on one side), inside the mesh's box (a corner behind the camera: no
    statement), looking away (everything behind), narrow and wide fields of view, a basis that is not orthonormal,
    an image smaller than the margin.  Radiance, primary-hit ids and ray counts must stay the oracle's."""
    base = scenes.bunny(n=10)
    sky = scenes.sky_cube_direct(16)
    st = pt.Settings(2, 4, 1e-4, 1000.0)
    C = pt.Camera
    cams = [
        (C.from_yaw_pitch((1.0, 1.0, 40.0), -90.0, 0.0, 1.0, 30.0), 96, 54),
        (C.from_yaw_pitch((1.6, 0.6, 1.2), -100.0, -5.0, 1.0, 75.0), 96, 54),
        (C.from_yaw_pitch((0.0, 0.5, -1.0), -90.0, 0.0, 1.0, 75.0), 64, 40),
        (C.from_yaw_pitch((1.0, 1.0, 4.7), 90.0, 0.0, 1.0, 75.0), 64, 40),
        (C.from_yaw_pitch((1.0, 1.0, 4.7), -90.0, 0.0, 1.0, 5.0), 64, 40),
        (C.from_yaw_pitch((0.3, 0.8, 2.5), -95.0, -10.0, 0.5, 140.0), 96, 54),
        (C(origin=(1.0, 1.0, 4.7), forward=(0.1, -0.05, -1.0), right=(1.0, 0.2, 0.0), up=(0.1, 1.0, 0.1), focal_length=1.3, vfov=60.0), 80, 48),
        (C.from_yaw_pitch((1.0, 1.0, 4.7), -90.0, 0.0, 1.0, 75.0), 3, 2),
        # far from the origin with a narrow field of view: a pixel is as wide as an ulp of the camera's position, so the
        # float direction lands pixels away from the exact projection (the rectangles' margin has a term for that) ...
        (C.from_yaw_pitch((1.0, 1.0, 1000.0), -90.0, 0.0, 1.0, 0.25), 96, 54),
        (C.from_yaw_pitch((300.0, 200.0, 900.0), -108.42, -11.86, 1.0, 0.3), 96, 54),
        # ... and so far out that the margin would cover the image: no statement
        (C.from_yaw_pitch((1.0, 1.0, 20000.0), -90.0, 0.0, 1.0, 0.01), 64, 40),
    ]
    for cam, w, h in cams:
        scene = pt.Scene(base.spheres, base.triangles, cam)
        r, osc = _pair(pt, po, scene, w, h, st, sky)
        r.render(3)
        gprim, _ = r.read_aov()
        r.render_frames(10, 1, 5)
        acc, prim, tt, s = osc.render(3, want_aov=True)
        rays = s["rays"]
        for f in range(5):
            acc, _, _, s = osc.render(10 + f, acc)
            rays += s["rays"]
        assert np.array_equal(gprim, prim)
        _exact(r.read_sum(), acc, f"camera at {cam.origin}")
        assert r.stats()["rays"] == rays
        r.close()


def test_short_launch_scratch_is_bounded(pt, scenes):
    """Launches of 2..15 frames plan one 1 KiB row of partial sums per tile and frame so that the frames run side by
    side -- as long as those rows stay below 640 MiB; above that they run frame after frame with no rows at all.
    Device memory in use may therefore grow by at most that much, whatever the image size, and the image is the same
    frame-by-frame sum either way."""
    import torch
    scene = scenes.bunny(n=6)
    sky = scenes.sky_cube_direct(8)
    st = pt.Settings(1, 3, 1e-4, 1000.0)
    for (w, h, n) in ((1920, 1080, 15), (3840, 2160, 8)):
        r = pt.Renderer(scene, w, h, settings=st)
        r.set_sky_cube(sky)
        r.render(1)
        r.sync()
        free0 = torch.cuda.mem_get_info(0)[0]
        r.render_frames(2, 1, n)
        r.sync()
        grown = free0 - torch.cuda.mem_get_info(0)[0]
        assert grown <= (680 << 20), f"{w}x{h}, {n} frames: {grown >> 20} MiB of scheduling memory"
        got = r.read_sum()
        r2 = pt.Renderer(scene, w, h, settings=st)
        r2.set_sky_cube(sky)
        for f in range(n + 1):
            r2.render(1 + f)
        same = got.view(np.uint32) == r2.read_sum().view(np.uint32)
        assert same.all(), f"{w}x{h}: {int((~same).sum())} values differ from frame-by-frame dispatches"
        r.close()
        r2.close()


def test_sky_only_tiles(pt, po, scenes):
    """Planned launches sum the tiles whose camera rays provably all leave the scene in a loop of their own,
    frame after frame.  Cameras that stress the proof: above a huge ground sphere with the horizon in view (rays that
    point at the sphere's side of the camera but pass it), looking up (everything behind), level with a small
    sphere, inside a sphere (no statement), a sphere whose silhouette grazes tile borders, no mesh, no spheres,
    PT_FLAG_REFERENCE_DISPATCH, several samples per frame.  Radiance (bit for bit in launches of fewer than 16 frames), primary-hit ids and ray
    counts must stay the oracle's, and the flagged tiles must be exactly tiles without any primary hit."""
    S, M, C = pt.Sphere, pt.Material, pt.Camera
    mesh = scenes.bunny(n=8).triangles
    none = mesh[:0]
    ground = S((0.0, -100.5, -1.0), 100.0, (0.8, 0.8, 0.0), M.Diffuse)
    small = S((0.4, 0.9, -1.2), 0.35, (0.7, 0.2, 0.2), M.Metal)
    glass = S((-0.9, 0.3, -0.6), 0.3, (1.0, 1.0, 1.0), M.Dielectric)
    sky = scenes.sky_cube_direct(16)
    st = pt.Settings(1, 5, 1e-4, 1000.0)
    cases = [
        ([ground], mesh, scenes.default_camera(), 160, 96, 0, True),
        ([ground, small, glass], mesh, C.from_yaw_pitch((0.5, 1.4, 3.5), -95.0, -6.0), 160, 96, 0, True),
        ([ground, small], none, C.from_yaw_pitch((0.0, 0.4, 2.0), -90.0, 35.0), 96, 56, 0, True),      # looking up
        ([small], none, C.from_yaw_pitch((0.4, 0.9, 2.0), -90.0, 0.0, 1.0, 40.0), 128, 72, 0, True),   # one small sphere
        ([S((0.0, 0.0, 0.0), 50.0, (0.5, 0.5, 0.5), M.Diffuse)], none, C.from_yaw_pitch((0.0, 1.0, 3.0), -90.0, 0.0), 64, 40, 0, False),  # inside
        ([], mesh, scenes.default_camera(), 128, 72, 0, True),                                          # mesh only
        ([], none, scenes.default_camera(), 64, 40, 0, True),                                           # nothing at all
        ([ground, small], mesh, scenes.default_camera(), 100, 59, pt.PT_FLAG_REFERENCE_DISPATCH, True),
        # far away, narrow field of view: float rays land pixels away from the exact projection the proof starts from
        ([small], mesh, C.from_yaw_pitch((1.0, 1.0, 1000.0), -90.0, 0.0, 1.0, 0.4), 160, 96, 0, None),
        ([ground, small], mesh, C.from_yaw_pitch((300.0, 200.0, 900.0), -108.42, -11.86, 1.0, 0.5), 160, 96, 0, None),
    ]
    for spheres, tris, cam, w, h, flags, expect_some in cases:
        scene = pt.Scene(spheres, tris, cam)
        osc = po.from_scene(scene, w, h, st, sky, po.FLAG_REFERENCE_DISPATCH if flags & pt.PT_FLAG_REFERENCE_DISPATCH else 0)
        r = pt.Renderer(scene, w, h, settings=st, flags=pt.PT_FLAG_AOV | flags)
        r.set_sky_cube(sky)
        # first call: four frames straight (no cost measurement yet), the rest planned; second call: planned, with the AOV
        r.render_frames(1, 1, 9)
        r.reset_ready_samples()
        r.render_frames(20, 1, 7)
        flagged = r.read_sky_tiles().astype(bool)
        gprim, gt = r.read_aov()
        got = r.read_sum()
        acc, prim, tt, s = osc.render(20, want_aov=True)
        rays = s["rays"]
        for f in range(1, 7):
            acc, _, _, s = osc.render(20 + f, acc)
            rays += s["rays"]
        what = f"{len(spheres)} spheres, {len(tris)} triangles, {w}x{h}, flags {flags:#x}"
        assert np.array_equal(gprim, prim) and np.array_equal(gt.view(np.uint32), tt.view(np.uint32)), what
        assert np.array_equal(got.view(np.uint32), acc.view(np.uint32)), what
        assert r.stats()["rays"] == rays, what
        if expect_some is not None:
            assert flagged.any() == expect_some, what
        # a flagged tile has no primary hit in any of its rendered pixels
        hit = prim != 0xFFFFFFFF
        ty, tx = np.nonzero(flagged)
        for a, b in zip(ty, tx):
            assert not hit[a * 8:a * 8 + 8, b * 8:b * 8 + 8].any(), what
        # a long launch (sums regrouped for the other tiles) agrees up to summation order, and exactly in flagged tiles
        r.reset_ready_samples()
        r.render_frames(100, 1, 24)
        got = r.read_sum()
        acc = None
        for f in range(24):
            acc, _, _, _ = osc.render(100 + f, acc) if acc is not None else osc.render(100 + f)
        np.testing.assert_allclose(got, acc, rtol=24 * 2.0 ** -23, atol=1e-6, err_msg=what)
        mask = np.kron(r.read_sky_tiles(), np.ones((8, 8), np.uint32))[:h, :w].astype(bool)
        assert np.array_equal(got[mask].view(np.uint32), acc[mask].view(np.uint32)), what
        r.close()
    # several samples per frame: the loop draws every sample's three random numbers and sums the frame's samples first
    scene = pt.Scene([ground, small], mesh, scenes.default_camera())
    st3 = pt.Settings(3, 5, 1e-4, 1000.0)
    osc = po.from_scene(scene, 128, 72, st3, sky)
    r = pt.Renderer(scene, 128, 72, settings=st3, flags=pt.PT_FLAG_AOV)
    r.set_sky_cube(sky)
    r.render_frames(1, 1, 9)
    r.reset_ready_samples()
    r.render_frames(20, 1, 5)
    assert r.read_sky_tiles().any()
    acc, prim, tt, s = osc.render(20, want_aov=True)
    rays = s["rays"]
    for f in range(1, 5):
        acc, _, _, s = osc.render(20 + f, acc)
        rays += s["rays"]
    gprim, gt = r.read_aov()
    assert np.array_equal(gprim, prim) and np.array_equal(r.read_sum().view(np.uint32), acc.view(np.uint32)) and r.stats()["rays"] == rays
    r.close()
    # per-ray statistics leave every tile to the megakernels
    r = pt.Renderer(scene, 96, 56, settings=st, flags=pt.PT_FLAG_DETAILED_STATS)
    r.set_sky_cube(sky)
    r.render_frames(1, 1, 9)
    r.render_frames(20, 1, 7)
    assert not r.read_sky_tiles().any()
    r.close()
