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
