"""The megakernels' second tree (see PT_FLAG_REFERENCE_WALK in include/ptrt.h): rays within reach of the mesh walk a
surface-area tree that pt_set_mesh builds over padded triangle bounds, every triangle hit found there is checked to be
the one a walk of the caller's tree returns, and whatever fails the check walks the caller's tree.  The bar is the one of
every other parity test: the same bits as a walk of the caller's tree alone, and as the oracle.  These cases aim at the
mechanism itself: that it is in use where expected, that it steps aside where it must (flat meshes, duplicated
triangles, cameras far away, coarse meshes), and that each of those paths gives the reference's picture."""
import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.library_choice]


def _render(pt, scene, w, h, st, sky, flags, frames, time0=11):
    r = pt.Renderer(scene, w, h, settings=st, flags=flags | pt.PT_FLAG_AOV)
    r.set_sky_cube(sky)
    r.render_frames(time0, 1, frames)
    out = r.read_sum().copy(), tuple(a.copy() for a in r.read_aov()), r.stats()
    r.close()
    return out


def _both_walks(pt, scene, w, h, st, sky, frames=12):
    """Both kernels with the second tree, both on the caller's tree only: bit for bit (a launch of fewer than 16 frames
    adds every frame on its own, in order, whichever kernel runs it).  Returns the pooled kernel's counters."""
    assert frames < 16
    runs = {}
    for kernel, kflag in (("pooled", pt.PT_FLAG_POOL_KERNEL), ("per-tile", pt.PT_FLAG_TILE_KERNEL)):
        for walk, wflag in (("both trees", 0), ("caller's tree", pt.PT_FLAG_REFERENCE_WALK)):
            runs[kernel, walk] = _render(pt, scene, w, h, st, sky, kflag | wflag, frames)
    g = runs["pooled", "both trees"]
    for (kernel, walk), other in runs.items():
        if walk == "caller's tree":
            assert other[2]["guard_nodes"] == 0 and other[2]["guarded"] == 0 and other[2]["rewalks"] == 0
        else:
            assert other[2]["guard_nodes"] == g[2]["guard_nodes"] and (other[2]["guarded"] > 0) == (g[2]["guarded"] > 0)
        diff = int((g[0].view(np.uint32) != other[0].view(np.uint32)).sum())
        assert diff == 0, f"{diff} of {g[0].size} sums differ from the {kernel} kernel on {walk}"
        assert np.array_equal(g[1][0], other[1][0]) and np.array_equal(g[1][1].view(np.uint32), other[1][1].view(np.uint32))
        assert g[2]["rays"] == other[2]["rays"]
    # a long launch, split into work items of several frames: the two pooled runs plan alike (nothing measured yet)
    g2 = _render(pt, scene, w, h, st, sky, pt.PT_FLAG_POOL_KERNEL, 40, time0=500)
    r2 = _render(pt, scene, w, h, st, sky, pt.PT_FLAG_POOL_KERNEL | pt.PT_FLAG_REFERENCE_WALK, 40, time0=500)
    assert np.array_equal(g2[0].view(np.uint32), r2[0].view(np.uint32)) and g2[2]["rays"] == r2[2]["rays"]
    return g[2]


def test_second_tree_is_walked_and_changes_nothing(pt, po, scenes):
    scene, sky, st = scenes.bunny(n=32), scenes.sky_cube_direct(16), pt.Settings(1, 8, 1e-4, 1000.0)
    w, h = 256, 144
    s = _both_walks(pt, scene, w, h, st, sky)
    assert s["guard_nodes"] > 0 and s["guarded"] > 0
    assert s["rewalks"] * 20 <= s["guarded"], "on a smooth mesh nearly every hit passes the check"
    # ... and the oracle (fp32 summation order of the fused launch aside)
    osc = po.from_scene(scene, w, h, st, sky)
    acc, rays = np.zeros((h, w, 4), np.float32), 0
    for t in range(11, 11 + 12):
        rays += osc.render(t, acc)[3]["rays"]
    r = pt.Renderer(scene, w, h, settings=st, flags=pt.PT_FLAG_POOL_KERNEL)
    r.set_sky_cube(sky)
    r.render_frames(11, 1, 12)
    assert r.stats()["rays"] == rays
    assert np.array_equal(r.read_sum().view(np.uint32), acc.view(np.uint32)), "frames added one by one: the oracle's bits"
    r.close()


def test_spp_and_materials_variants(pt, scenes):
    """The other compiled variants of the pooled kernel: several samples per frame, a gizmo sphere, triangle materials."""
    S, M = pt.Sphere, pt.Material
    base = scenes.bunny(n=32)
    spheres = list(base.spheres) + [S((1.4, 0.4, -0.4), 0.4, (1, 1, 1), M.Dielectric), S((-1.3, 0.3, -0.2), 0.3, (0.9, 0.6, 0.2), M.Metal),
                                   S((0.0, 1.7, -1.0), 0.3, (0.2, 0.9, 0.2), M.Gizmo)]
    scene = pt.Scene(spheres, base.triangles, base.camera)
    s = _both_walks(pt, scene, 160, 96, pt.Settings(3, 6, 1e-4, 1000.0), scenes.sky_cube_direct(8))
    assert s["guarded"] > 0


def test_flat_mesh_sends_every_hit_to_the_callers_tree(pt, scenes):
    """A grid in the plane y = const: no hit point lies strictly inside its triangle's bounds (they have no extent in y),
    so no hit can be certified and every one of them walks the caller's tree as well."""
    n = 100
    g = np.linspace(-1.0, 1.0, n + 1).astype(np.float32)
    tris = np.zeros(2 * n * n, pt.TRIANGLE_DTYPE)
    k = 0
    for i in range(n):
        for j in range(n):
            p00, p10, p01, p11 = (g[i], 0.25, g[j] - 1), (g[i + 1], 0.25, g[j] - 1), (g[i], 0.25, g[j + 1] - 1), (g[i + 1], 0.25, g[j + 1] - 1)
            for a, b, c in ((p00, p01, p10), (p10, p01, p11)):
                tris[k]["a"], tris[k]["b"], tris[k]["c"] = a, b, c
                tris[k]["na"] = tris[k]["nb"] = tris[k]["nc"] = (0, 1, 0)
                k += 1
    tris["albedo"] = 1.0
    base = scenes.bunny(n=4)
    scene = pt.Scene(base.spheres, tris, pt.Camera.from_yaw_pitch((0.3, 1.6, 1.4), -90.0, -30.0, 1.0, 75.0))
    s = _both_walks(pt, scene, 160, 96, pt.Settings(1, 6, 1e-4, 1000.0), scenes.sky_cube_direct(8))
    assert s["guard_nodes"] > 0 and s["guarded"] > 0 and s["rewalks"] > 0


def test_duplicated_triangles_tie_and_walk_the_callers_tree(pt, scenes):
    """Every triangle twice: each hit comes with a second accepted hit at exactly the same t, and which of the two
    wins is the tree's order of tests -- so the second tree's answer is never taken."""
    tris = scenes.blob_mesh(32)
    scene0 = scenes.bunny(n=4)
    scene = pt.Scene(scene0.spheres, np.concatenate([tris, tris]), scene0.camera)
    s = _both_walks(pt, scene, 160, 96, pt.Settings(1, 6, 1e-4, 1000.0), scenes.sky_cube_direct(8))
    assert s["guard_nodes"] > 0 and s["rewalks"] > 0


def test_far_camera_is_out_of_reach(pt, scenes):
    base = scenes.bunny(n=32)
    far = pt.Camera.from_yaw_pitch((0.0, 4.0, 39.0), -90.0, -5.0, 1.0, 20.0)
    s = _both_walks(pt, pt.Scene(base.spheres, base.triangles, far), 160, 96, pt.Settings(1, 6, 1e-4, 1000.0), scenes.sky_cube_direct(8))
    # (camera rays start 40 units from a mesh of radius one: none of them may use the second tree; bounce rays
    # off the mesh and off the ground near it start within reach and do)
    assert s["guard_nodes"] > 0
    r = pt.Renderer(pt.Scene(base.spheres, base.triangles, far), 160, 96, settings=pt.Settings(1, 1, 1e-4, 1000.0), flags=pt.PT_FLAG_POOL_KERNEL)
    r.set_sky_cube(scenes.sky_cube_direct(8))
    r.render_frames(3, 1, 16)
    assert r.stats()["guarded"] == 0, "depth 1: camera rays only, all out of reach"
    r.close()


def test_coarse_mesh_gets_no_second_tree(pt, scenes):
    """Triangles that are large against the mesh: the padding would be no small part of a triangle, no tree is built."""
    base = scenes.bunny(n=4)      # 192 triangles of about a third of the mesh's radius
    s = _both_walks(pt, base, 128, 72, pt.Settings(1, 5, 1e-4, 1000.0), scenes.sky_cube_direct(8))
    assert s["guard_nodes"] == 0 and s["guarded"] == 0


def test_t_max_beyond_the_walks_own_horizon(pt, scenes):
    """t_max above 9999: the triangle test would accept hits that far out, the reference walk never gets there (its
    prune distance starts at 9999, compute.wgsl:242); so must the second tree's."""
    base = scenes.bunny(n=32)
    cam = pt.Camera.from_yaw_pitch((0.0, 0.5, 6.0), -90.0, 0.0, 1.0, 40.0)
    s = _both_walks(pt, pt.Scene(base.spheres, base.triangles, cam), 96, 64, pt.Settings(1, 4, 1e-4, 1e9), scenes.sky_cube_direct(8))
    assert s["guard_nodes"] > 0


def _aimed_rays(tris, rng, n):
    """Rays aimed where the two trees could disagree: at vertices and along edges (several triangles accept the same
    point, some at exactly the same t), at grazing angles (the triangle test's error grows like 1 / cos), from points on
    the surface itself (bounce rays), with un-normalised, tiny and zero direction components."""
    a, b, c = (tris[k].astype(np.float64) for k in ("a", "b", "c"))
    pick = rng.randint(0, len(tris), n)
    A, B, C = a[pick], b[pick], c[pick]
    kind = rng.randint(0, 6, n)
    u, v = rng.uniform(0, 1, n), rng.uniform(0, 1, n)
    flip = u + v > 1
    u[flip], v[flip] = 1 - u[flip], 1 - v[flip]
    target = A + u[:, None] * (B - A) + v[:, None] * (C - A)                      # kind 0: inside
    corner = np.where((rng.randint(0, 3, n) == 0)[:, None], A, np.where((rng.randint(0, 2, n) == 0)[:, None], B, C))
    target = np.where((kind == 1)[:, None], corner, target)                        # kind 1: a vertex, exactly
    s = rng.uniform(0, 1, n)[:, None]
    edge = np.where((rng.randint(0, 3, n) == 0)[:, None], A + s * (B - A), np.where((rng.randint(0, 2, n) == 0)[:, None], A + s * (C - A), B + s * (C - B)))
    target = np.where((kind == 2)[:, None], edge, target)                          # kind 2: on an edge
    nrm = np.cross(B - A, C - A)
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-30)
    tang = (B - A) / np.maximum(np.linalg.norm(B - A, axis=1, keepdims=True), 1e-30)
    origin = np.array([0.0, 0.5, -1.0]) + rng.normal(0, 1, (n, 3)) * rng.uniform(0.05, 5.0, (n, 1))   # anywhere within reach, inside the mesh too
    graze = target + tang * rng.uniform(0.02, 3.0, (n, 1)) + nrm * (10.0 ** rng.uniform(-7, -2, (n, 1))) * rng.choice([-1, 1], (n, 1))
    origin = np.where((kind == 3)[:, None], graze, origin)                         # kind 3: grazing
    other = rng.randint(0, len(tris), n)
    surf = a[other] + 0.3 * (b[other] - a[other]) + 0.3 * (c[other] - a[other])
    origin = np.where((kind == 4)[:, None], surf, origin)                          # kind 4: from the surface
    o = origin.astype(np.float32)
    d = (target - o.astype(np.float64))
    d *= 10.0 ** rng.uniform(-1.5, 0.5, (n, 1))                                     # un-normalised: t scales
    d = d.astype(np.float32)
    k5 = kind == 5                                                                  # kind 5: degenerate components
    d[k5 & (pick % 3 == 0), 0] = 0.0
    d[k5 & (pick % 3 == 1), 1] = np.float32(1e-30)
    d[k5 & (pick % 3 == 2), 2] *= np.float32(1e-6)
    return o, d, kind


def test_rays_aimed_at_edges_vertices_and_grazing_angles(pt, po, scenes):
    """pt_trace_rays runs the pooled kernel's scheme ray by ray and says which tree answered: on rays aimed at the places
    where a different tree could give a different answer, every output equals the oracle's walk of the reference tree."""
    base = scenes.bunny(n=32)
    spheres = list(base.spheres) + [pt.Sphere((0.9, 0.6, -0.2), 0.35, (1, 1, 1), pt.Material.Dielectric)]
    scene = pt.Scene(spheres, base.triangles, base.camera)
    st = pt.Settings(1, 4, 1e-4, 1000.0)
    r = pt.Renderer(scene, 16, 16, settings=st)
    assert r.stats()["guard_nodes"] > 0
    osc = po.from_scene(scene, 16, 16, st, None)
    rng = np.random.RandomState(17)
    n = 400000
    o, d, kind = _aimed_rays(scene.triangles, rng, n)
    gp, gt, gpos, gn, gf, walked = r.trace_rays_walked(o, d)
    op, ot, opos, on, of = osc.trace_rays(o, d)
    bad = gp != op
    assert not bad.any(), f"{int(bad.sum())} primitive ids differ (kinds {np.bincount(kind[bad], minlength=6)}, trees {np.bincount(walked[bad], minlength=3)})"
    hit = op != 0xFFFFFFFF
    for name, g, w in (("t", gt, ot), ("hit point", gpos, opos), ("normal", gn, on)):
        gb, wb = g[hit].view(np.uint32), w[hit].view(np.uint32)
        assert np.array_equal(gb, wb), f"{name}: {int((gb != wb).sum())} values differ"
    assert np.array_equal(gf[hit], of[hit])
    share = np.bincount(walked, minlength=3) / n
    # the scheme is in use, and its fallback is exercised: vertex and edge rays tie or sit on the bounds' faces
    assert share[1] > 0.5 and share[2] > 0.01, share
    # (a ray aimed at a vertex of the far side hits the near side first, somewhere inside a triangle: the rates are well below one)
    back = [(walked[kind == k] == 2).mean() for k in range(3)]
    assert back[1] > 0.05 and back[1] > 4 * back[0] and back[2] > back[0], f"rays at vertices and edges fall back more often than rays at interiors: {back}"
    zero = (d == 0).any(1)
    assert zero.sum() > 1000 and (walked[zero] == 0).all(), "a zero direction component keeps a ray on the reference tree"
    # the same rays with the second tree switched off: nothing differs, nothing is reported
    r2 = pt.Renderer(scene, 16, 16, settings=st, flags=pt.PT_FLAG_REFERENCE_WALK)
    rp, rt, rpos, rn, rf, rwalked = r2.trace_rays_walked(o, d)
    assert not rwalked.any() and np.array_equal(rp, gp) and np.array_equal(rt.view(np.uint32), gt.view(np.uint32))
    r.close()
    r2.close()


def test_callers_tree_with_boxes_that_do_not_contain_their_triangles(pt, po, scenes):
    """The check that certifies a hit leans on the caller's boxes containing what is below them.  A caller's tree whose
    boxes are too small is still a tree the shader would walk (missing what its boxes hide): no second tree is built
    for it, and the picture is the oracle's walk of those very nodes."""
    scene = scenes.bunny(n=32)
    nodes = scene.nodes.copy()
    mid = 0.5 * (nodes["min"] + nodes["max"])
    nodes["min"] = (mid + 0.9 * (nodes["min"] - mid)).astype(np.float32)
    nodes["max"] = (mid + 0.9 * (nodes["max"] - mid)).astype(np.float32)
    scene.nodes = nodes
    w, h, st, sky = 160, 96, pt.Settings(1, 6, 1e-4, 1000.0), scenes.sky_cube_direct(8)
    osc = po.OracleScene(w, h, scene.camera.buffer(), st.buffer(), po.pack_spheres(scene.spheres), scene.triangles, nodes, scene.triangle_indices, sky, 0)
    want = np.zeros((h, w, 4), np.float32)
    for t in range(11, 23):
        want = osc.render(t, want)[0]
    for flag in (pt.PT_FLAG_POOL_KERNEL, pt.PT_FLAG_TILE_KERNEL):
        got, _, stats = _render(pt, scene, w, h, st, sky, flag, 12)
        assert stats["guard_nodes"] == 0 and stats["guarded"] == 0
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # the same mesh with its own tree does get one
    assert _render(pt, scenes.bunny(n=32), w, h, st, sky, pt.PT_FLAG_POOL_KERNEL, 2)[2]["guard_nodes"] > 0


def test_flag_set_after_the_mesh_keeps_rays_off_the_second_tree(pt, scenes):
    scene, sky, st = scenes.bunny(n=32), scenes.sky_cube_direct(8), pt.Settings(1, 5, 1e-4, 1000.0)
    for kernel in (pt.PT_FLAG_POOL_KERNEL, pt.PT_FLAG_TILE_KERNEL):
        r = pt.Renderer(scene, 128, 72, settings=st, flags=kernel)
        r.set_sky_cube(sky)
        r.render_frames(5, 1, 12)
        with_tree, s1 = r.read_sum().copy(), r.stats()
        assert s1["guard_nodes"] > 0 and s1["guarded"] > 0
        r.set_flags(kernel | pt.PT_FLAG_REFERENCE_WALK)
        r.reset_ready_samples()
        r.render_frames(5, 1, 12)
        s2 = r.stats()   # (the counters start over with the reset)
        assert s2["guarded"] == 0 and s2["rays"] == s1["rays"], "no ray walked the second tree while the flag was set"
        assert np.array_equal(r.read_sum().view(np.uint32), with_tree.view(np.uint32))
        r.set_flags(kernel)
        r.reset_ready_samples()
        r.render_frames(5, 1, 12)
        assert r.stats()["guarded"] == s1["guarded"] and np.array_equal(r.read_sum().view(np.uint32), with_tree.view(np.uint32))
        r.close()
