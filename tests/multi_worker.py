"""One rank of the sample-split job used by the multi-rank tests (not a test module itself).

Started either by torch.distributed.run (RANK / WORLD_SIZE / MASTER_* in the environment) or with explicit
--rank/--world/--port.  Renders its share of the frames -- with the HIP renderer on GPU 0 (--renderer hip) or with the
CPU oracle (--renderer oracle, for the rendezvous and dealing logic on machines without a GPU) -- sums the ranks'
images through gloo on host copies, and rank 0 writes the sum and the total ray count to --out (.npz)."""
import argparse
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def job(pt, scenes):
    """The job every party of the test renders: (scene, sky, settings, width, height, time0, frames)."""
    return scenes.bunny(n=12), scenes.sky_cube_direct(16), pt.Settings(1, 6, 1e-4, 1000.0), 160, 96, 5, 8


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--renderer", choices=["hip", "oracle"], required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--rank", type=int, default=None)
    ap.add_argument("--world", type=int, default=None)
    ap.add_argument("--port", type=int, default=None)
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    import _pkgload
    pt = _pkgload.load()
    scenes = importlib.import_module("path_tracer_amd.scenes")
    multi = importlib.import_module("path_tracer_amd.multi")
    if a.rank is None:
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        dist.init_process_group("gloo")  # env:// rendezvous, as under torch.distributed.run
    else:
        rank, world = a.rank, a.world
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{a.port}", rank=rank, world_size=world)
    scene, sky, st, w, h, time0, frames = job(pt, scenes)
    t0, stride, cnt = multi.frames_for_rank(rank, world, time0, frames)
    if a.renderer == "hip":
        r = pt.Renderer(scene, w, h, device=0, settings=st)
        r.set_sky_cube(sky)
        r.render_frames(t0, stride, cnt)
        acc, rays = r.read_sum(), r.stats()["rays"]
        r.close()
    else:
        from oracle import pt_oracle as po
        osc = po.from_scene(scene, w, h, st, sky)
        acc, rays = np.zeros((h, w, 4), np.float32), 0
        for i in range(cnt):
            _, _, _, s = osc.render(t0 + i * stride, acc, threads=2)
            rays += s["rays"]
    tens = torch.from_numpy(acc)
    multi.reduce_sum_to_root(tens, 0)
    total = torch.tensor([rays], dtype=torch.int64)
    dist.all_reduce(total)
    if rank == 0:
        np.savez(a.out, sum=tens.numpy(), rays=int(total.item()))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
