"""pt_group_* through the C ABI: several renderer contexts behind one handle, frames dealt over the members
(member g renders time0 + g, time0 + g + G, ...), one reduce onto a root member.  This box has one GPU, so the members
share GPU 0 -- the dealing, the peer copies, the fixed-order addition and the clearing of the other members are the code
an 8-GPU node runs; only the copies' route differs.  RCCL refuses two ranks on one device, so its path is rehearsed with
a one-member group (library found with dlopen, communicator, the collective call on the member's stream)."""
import numpy as np
import pytest

import multi_worker

pytestmark = pytest.mark.gpu


def _oracle_sum(po, scene, w, h, st, sky, times):
    osc = po.from_scene(scene, w, h, st, sky)
    acc, rays = np.zeros((h, w, 4), np.float32), 0
    for t in times:
        _, _, _, s = osc.render(t, acc)
        rays += s["rays"]
    return acc, rays


def test_group_of_two_members_on_one_gpu_matches_one_context_and_the_oracle(pt, po, scenes):
    scene, sky, st, w, h, time0, frames = multi_worker.job(pt, scenes)
    g = pt.RendererGroup(scene, w, h, [0, 0], settings=st)
    assert len(g) == 2
    g.set_sky_cube(sky)
    g.render_frames(time0, frames)
    # before the reduce every member holds its own share
    a, b = g.read_sum(0), g.read_sum(1)
    assert (a[..., 3] == (frames + 1) // 2).all() and (b[..., 3] == frames // 2).all()
    g.reduce(0)
    got = g.read_sum(0)
    assert (got[..., 3] == frames).all()
    assert not g.read_sum(1).any(), "the other member's sums are handed over, not kept"
    one = pt.Renderer(scene, w, h, settings=st)
    one.set_sky_cube(sky)
    one.render_frames(time0, 1, frames)
    ref, rays = one.read_sum(), one.stats()["rays"]
    one.close()
    np.testing.assert_allclose(got, ref, rtol=frames * 2.0 ** -23, atol=1e-6)
    want, orays = _oracle_sum(po, scene, w, h, st, sky, range(time0, time0 + frames))
    assert g.stats()["rays"] == rays == orays and g.stats()["frames"] == frames
    np.testing.assert_allclose(got, want, rtol=frames * 2.0 ** -23, atol=1e-6)
    # the group goes on rendering and reduces again: nothing is counted twice, and the sum's order is fixed
    g.render_frames(time0 + frames, 5)
    g.reduce(0)
    again = g.read_sum(0)
    assert (again[..., 3] == frames + 5).all()
    want2, _ = _oracle_sum(po, scene, w, h, st, sky, range(time0, time0 + frames + 5))
    np.testing.assert_allclose(again, want2, rtol=(frames + 5) * 2.0 ** -23, atol=1e-6)
    g.reset_ready_samples()
    g.render_frames(time0, frames)
    g.reduce(0)
    assert np.array_equal(g.read_sum(0).view(np.uint32), got.view(np.uint32)), "same frames, same members, same order: same bits"
    # another root: the staging buffers move with it
    g.reset_ready_samples()
    g.render_frames(time0, frames)
    g.reduce(1)
    np.testing.assert_allclose(g.read_sum(1), ref, rtol=frames * 2.0 ** -23, atol=1e-6)
    assert not g.read_sum(0).any()
    g.close()


def test_group_of_three_deals_every_frame_once(pt, po, scenes):
    scene, sky, st = scenes.cornell(), scenes.sky_cube_direct(8), pt.Settings(2, 4, 1e-4, 1000.0)
    w, h, time0, frames = 64, 40, 0xFFFFFFFE, 7        # the time values wrap like the shader's u32 seed
    g = pt.RendererGroup(scene, w, h, [0, 0, 0], settings=st)
    g.set_sky_cube(sky)
    g.render_frames(time0, frames)
    assert [int(g.read_sum(m)[0, 0, 3]) for m in range(3)] == [3, 2, 2]
    g.reduce(0)
    want, orays = _oracle_sum(po, scene, w, h, st, sky, [(time0 + f) & 0xFFFFFFFF for f in range(frames)])
    assert g.stats()["rays"] == orays
    np.testing.assert_allclose(g.read_sum(0), want, rtol=frames * 2.0 ** -23, atol=1e-6)
    # fewer frames than members: the last members render nothing
    g.reset_ready_samples()
    g.render_frames(3, 2)
    g.reduce(0)
    assert (g.read_sum(0)[..., 3] == 2).all()
    g.close()


@pytest.mark.library_choice
def test_one_member_rccl_group(pt, po, scenes):
    scene, sky, st, w, h, time0, frames = multi_worker.job(pt, scenes)
    g = pt.RendererGroup(scene, w, h, [0], settings=st, group_flags=pt.PT_GROUP_RCCL)
    g.set_sky_cube(sky)
    g.render_frames(time0, frames)
    g.reduce(0)          # ncclReduce over a communicator of one rank, in place, on the member's stream
    got = g.read_sum(0)
    one = pt.Renderer(scene, w, h, settings=st)
    one.set_sky_cube(sky)
    one.render_frames(time0, 1, frames)
    assert np.array_equal(got.view(np.uint32), one.read_sum().view(np.uint32))
    one.close()
    g.close()


@pytest.mark.library_choice
def test_group_argument_errors(pt, scenes):
    import ctypes as C
    l = pt.lib()
    out = C.c_void_p()
    devs = (C.c_int * 2)(0, 99)
    assert l.pt_group_create(devs, 2, 32, 32, 0, C.byref(out)) == -1 and not out.value      # device out of range
    assert b"member 1" in l.pt_group_last_error(None)
    assert l.pt_group_create(devs, 0, 32, 32, 0, C.byref(out)) == -1
    assert l.pt_group_create(devs, 1, 32, 32, 2, C.byref(out)) == -1                          # unknown flag
    g = pt.RendererGroup(scenes.cornell(), 32, 20, [0, 0])
    with pytest.raises(pt.PtError):
        g.reduce(2)
    assert l.pt_group_member(g._g, 2) is None and l.pt_group_member(g._g, -1) is None
    g.close()


@pytest.mark.library_choice
def test_bench_group_mode_on_one_gpu(tmp_path):
    """`bench.py --gpus 1 --group`: the bench step through pt_group_* in one process (a one-member group: the reduce is a
    no-op, everything else is the path an 8-GPU run takes) gives a well-formed line with the same ray count per step as the
    plain run's frames."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("PTRT_KERNEL", None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--group", "--steps", "2", "--warmup", "1", "--frames", "32",
                          "--sky-size", "256"], env=env, capture_output=True, text=True, timeout=110)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 1 and d["unit"] == "Mrays/s" and d["value"] > 0 and d["config"]["frames_in_reduced_image"] == 32
    assert "pt_group" in d["config"]["parallelism"]
