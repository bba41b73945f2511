"""World-size-2 test of the sample split + sum-reduce on CPU (gloo).  The per-rank renderer here is
the CPU oracle -- the point is the frame dealing and the collective, which are the same code the
GPU ranks run (path_tracer_amd.multi)."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_frames_for_rank_partitions_the_job(pt):
    multi = importlib.import_module("path_tracer_amd.multi")
    for world in (1, 2, 3, 8):
        for n in (0, 1, 5, 8, 17):
            seen = []
            for r in range(world):
                t0, stride, cnt = multi.frames_for_rank(r, world, 100, n)
                seen += [t0 + i * stride for i in range(cnt)]
            assert sorted(seen) == list(range(100, 100 + n))
    assert multi.frames_for_rank(1, 2, 0xFFFFFFFF, 4)[0] == 0     # u32 wrap like the shader's seed
    with pytest.raises(ValueError):
        multi.frames_for_rank(2, 2, 0, 4)


def test_group_entry_points_without_a_gpu(pt):
    """pt_group_*: argument checks and the failure path on a machine without a GPU (a message, a status, no handle); the
    frame shares pt_group_render_frames deals are those of frames_for_rank (stated in include/ptrt.h: member g renders
    time0 + g with stride G)."""
    import ctypes as C
    import torch
    l = pt.lib()
    out = C.c_void_p(1)
    devs = (C.c_int * 2)(0, 1)
    assert l.pt_group_create(None, 2, 64, 64, 0, C.byref(out)) == -1 and not out.value
    assert l.pt_group_create(devs, 65, 64, 64, 0, C.byref(out)) == -1
    assert l.pt_group_create(devs, 2, 64, 64, 0x10, C.byref(out)) == -1 and b"flag" in l.pt_group_last_error(None)
    assert l.pt_group_create(devs, 2, 64, 64, 0, None) == -1
    if not torch.cuda.is_available():
        rc = l.pt_group_create(devs, 2, 64, 64, 0, C.byref(out))
        assert rc == -2 and not out.value and b"member 0" in l.pt_group_last_error(None)
    for fn in (l.pt_group_sync, l.pt_group_reset):
        assert fn(None) == -1
    assert l.pt_group_size(None) == 0 and l.pt_group_member(None, 0) is None
    l.pt_group_destroy(None)
    multi = importlib.import_module("path_tracer_amd.multi")
    text = open(os.path.join(ROOT, "path-tracer_amd", "csrc", "pt_group.hip")).read()
    assert "pt_render_frames(g->members[m], time0 + m, G, count)" in text and "(n_frames - g + G - 1u) / G" in text
    for G in (1, 2, 3, 8):
        for n in (0, 1, 7, 8, 9):
            for g in range(G):
                assert multi.frames_for_rank(g, G, 10, n)[2] == ((n - g + G - 1) // G if n > g else 0)


def _worker(rank, world, port, n_frames, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import _pkgload
    pt = _pkgload.load()
    scenes = importlib.import_module("path_tracer_amd.scenes")
    multi = importlib.import_module("path_tracer_amd.multi")
    from oracle import pt_oracle as po
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    scene, sky, st = scenes.cornell(), scenes.sky_cube_direct(8), pt.Settings(1, 4, 1e-4, 1000.0)
    osc = po.from_scene(scene, 48, 27, st, sky)
    t0, stride, cnt = multi.frames_for_rank(rank, world, 5, n_frames)
    acc = np.zeros((27, 48, 4), np.float32)
    for i in range(cnt):
        osc.render(t0 + i * stride, acc, threads=2)
    tens = torch.from_numpy(acc)
    multi.reduce_sum_to_root(tens, 0)
    if rank == 0:
        np.save(out_path, tens.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sum_equals_single_rank(pt, po, scenes, tmp_path):
    import torch.multiprocessing as mp
    n_frames = 7
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "sum.npy")
    mp.spawn(_worker, args=(2, port, n_frames, out), nprocs=2, join=True)
    got = np.load(out)
    st = pt.Settings(1, 4, 1e-4, 1000.0)
    osc = po.from_scene(scenes.cornell(), 48, 27, st, scenes.sky_cube_direct(8))
    want = np.zeros((27, 48, 4), np.float32)
    for f in range(n_frames):
        osc.render(5 + f, want, threads=2)
    assert (got[..., 3] == n_frames).all()
    np.testing.assert_allclose(got, want, rtol=n_frames * 2.0 ** -23, atol=1e-6)   # summation order only


def test_bench_self_launch_starts_the_ranks(pt, po, scenes, tmp_path):
    """`python bench.py --gpus N` without a torch.distributed environment starts N ranks through
    torch.distributed.run on 127.0.0.1.  The same launch line, pointed at the CPU worker instead of bench.py, must
    bring up two ranks that find each other through the environment and produce the single-rank image."""
    import subprocess
    sys.path.insert(0, ROOT)
    import bench
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import multi_worker
    args = bench.parse_args(["--gpus", "2", "--steps", "3"])
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = bench.self_launch_command(args, ["--gpus", "2", "--steps", "3"], port)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=2" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "2", "--steps", "3"]
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    out = str(tmp_path / "sum.npz")
    worker = cmd[:script] + [os.path.join(ROOT, "tests", "multi_worker.py"), "--renderer", "oracle", "--out", out]
    assert subprocess.run(worker, timeout=170).returncode == 0
    got = np.load(out)
    scene, sky, st, w, h, time0, frames = multi_worker.job(pt, scenes)
    osc = po.from_scene(scene, w, h, st, sky)
    want, rays = np.zeros((h, w, 4), np.float32), 0
    for f in range(frames):
        _, _, _, s = osc.render(time0 + f, want, threads=2)
        rays += s["rays"]
    assert int(got["rays"]) == rays and (got["sum"][..., 3] == frames).all()
    np.testing.assert_allclose(got["sum"], want, rtol=frames * 2.0 ** -23, atol=1e-6)


def test_bench_frame_dealing(pt):
    """Weak scaling gives every rank the configuration's frame count, strong scaling deals it; steps never share a time value."""
    sys.path.insert(0, ROOT)
    import bench
    multi = importlib.import_module("path_tracer_amd.multi")
    for scaling, per_rank in (("weak", 256), ("strong", 64)):
        seen = set()
        for step in range(3):
            for g in range(4):
                t0, stride, cnt = bench.frames_of_step(multi, g, 4, scaling, 256, step)
                assert cnt == per_rank
                times = {t0 + i * stride for i in range(cnt)}
                assert not (times & seen)
                seen |= times
        assert len(seen) == 3 * 4 * per_rank
