#!/usr/bin/env python3
"""Benchmark of the path-tracing hot path on MI355X.

Metric: Mrays/s, 1 ray = 1 hitScene call (compute.wgsl:138), counted by the kernel itself.
Default workload (BASELINE.json configs[2]): ~69k-triangle mesh + ground sphere, 1920x1080, 8 bounces,
256 spp rendered as 256 dispatch-equivalents of 1 spp fused into one launch, sky = 4096^2 cube
map generated on the GPU from a procedural panorama.  One "step" = that whole job on every rank:
clear the accumulation buffer, one launch of the megakernel, and (N > 1) one RCCL sum-reduce of the
fp32 accumulation buffers to rank 0.

Scaling over N GPUs (frames are independent given their time value, compute.wgsl:85; rank g renders the
frames f = g (mod N) of the step's frame range):
  weak   (default)  every rank renders the configuration's frame count: the reduced image has N times the samples;
  strong            the configuration's frame count is dealt over the ranks (BASELINE configs[3]: 4096 spp split
                    across 2/4/8 GPUs): fixed total work.

Usage: python bench.py [--gpus N --steps K --warmup W] [--workload NAME] [--scaling weak|strong] [--no-cpu-baseline]
With --gpus N > 1 and no torch.distributed environment the script starts its own N worker processes
(python -m torch.distributed.run on 127.0.0.1) and relays rank 0's JSON line; under an external
torch.distributed.run it reads RANK / LOCAL_RANK / WORLD_SIZE as usual.
"""
from __future__ import annotations

import argparse
import glob
import importlib
import json
import os
import re
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_VALU_PEAK_TFLOPS = 157.3   # MI355X vector FP32 peak (packed issue), spec
HBM_PEAK_GBS = 8000.0
INFINITY_CACHE_BYTES = 256 << 20  # memory-side cache in front of HBM: a working set below it is not served by HBM
DEFAULT_SCALING = {"bunny_4k_d8_spp4096": "strong", "field_4k_d8_spp1024": "strong"}


def algorithmic_flops(st: dict) -> float:
    """Work model of the shader as written (flop = add/mul/min/max/cmp 1, fma 2, div/sqrt/sin 1):
    interior step 57, triangle test 60 (+40 accepted), sphere test 30 (+20 accepted),
    shaded bounce 120, path set-up + sky + accumulate 60."""
    return (57.0 * st["interior"] + 60.0 * st["tri_tests"] + 40.0 * st["tri_accept"] + 30.0 * st["sphere_tests"]
            + 20.0 * st["sphere_accept"] + 120.0 * st["bounces"] + 60.0 * st["paths"])


def algorithmic_bytes(st: dict, pixel_frames: int) -> float:
    """interior step 64 B (two child nodes), triangle test 52 B, closest hit 48 B of normals,
    sky texel 16 B, accumulation 32 B per pixel and frame."""
    return (64.0 * st["interior"] + 52.0 * st["tri_tests"] + 48.0 * st["tri_accept"] + 16.0 * st["sky_lookups"]
            + 32.0 * pixel_frames)


def working_set_bytes(scene, second_tree_nodes: int = 0) -> int:
    """What a BVH walk reads, as the library stores it: 32 B per node and 64 B per triangle in the walk table, 48 B of
    vertex normals per triangle; with the library's second tree the same again for its nodes and its own order of the
    triangles, plus its four-child records (64 B each, about one per three nodes)."""
    one = 32 * int(len(scene.nodes)) + (64 + 48) * int(len(scene.triangles))
    if second_tree_nodes:
        one += 32 * second_tree_nodes + (64 + 48) * int(len(scene.triangles)) + 64 * (second_tree_nodes // 3)
    return one


def sky_loop_pixels(flags, w: int, h: int) -> int:
    """Pixels of the image that lie in tiles flagged sky-only (8x8 tiles, the last ones clipped by the image)."""
    import numpy as np
    ty, tx = np.nonzero(flags)
    return int((np.minimum(8, h - 8 * ty) * np.minimum(8, w - 8 * tx)).sum())


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="bunny_1080p_d8_spp256")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="default: strong for the configurations BASELINE.json splits across GPUs, weak otherwise")
    ap.add_argument("--frames", type=int, default=0, help="override the configuration's frame count")
    ap.add_argument("--sky-size", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rebuild-bvh", action="store_true",
                    help="PT_FLAG_REBUILD_BVH: walk the library's own binned-SAH tree instead of the reference's (opt-in mode outside "
                         "the bit-for-bit contract; never the headline number)")
    ap.add_argument("--reference-walk", action="store_true",
                    help="PT_FLAG_REFERENCE_WALK: the pooled kernel walks the reference tree only (same results bit for bit; measures "
                         "what the library's second tree is worth)")
    ap.add_argument("--group", action="store_true",
                    help="one process drives all --gpus GPUs through the C ABI's pt_group_* (frames dealt over the GPUs, one "
                         "pt_group_reduce per step) instead of one torch.distributed rank per GPU")
    ap.add_argument("--group-rccl", action="store_true", help="with --group: sum with RCCL instead of peer copies in a fixed order")
    return ap.parse_args(argv)


def self_launch_command(args, argv, port: int) -> list[str]:
    """The worker launch of `python bench.py --gpus N` when no torch.distributed environment is present."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def frames_of_step(multi, rank: int, world: int, scaling: str, frames: int, step: int):
    """(time0, stride, count) of this rank's frames in step `step`; steps use disjoint time ranges."""
    total = frames * world if scaling == "weak" else frames
    return multi.frames_for_rank(rank, world, 1 + step * total, total)


def find_traffic(workload: str, frames_per_launch: int, bvh: str = "reference"):
    """HBM bytes per launch from the newest committed PMC record of this workload, launch length and tree
    (profiles/rNN/<anything>/pmc_traffic.json, written by tools/summarize_prof.py)."""
    best = None
    for path in glob.glob(os.path.join(ROOT, "profiles", "**", "pmc_traffic.json"), recursive=True):
        try:
            rec = json.load(open(path))
        except Exception:
            continue
        if rec.get("workload") != workload or rec.get("frames") != frames_per_launch or rec.get("bvh", "reference") != bvh:
            continue
        m = re.search(r"profiles/r(\d+)/", path.replace(os.sep, "/"))
        key = (int(m.group(1)) if m else 0, os.path.getmtime(path))
        if best is None or key > best[0]:
            best = (key, rec, os.path.relpath(path, ROOT))
    return (best[1].get("hbm_bytes_per_launch"), best[2]) if best else (None, None)


def cpu_baseline(pt, scenes, scene, w, h, settings, sky_cube, budget_s: float = 20.0):
    """Times the CPU oracle (a scalar C port of the shader, NOT lavapipe) on a bounded sample of
    the same workload: whole 1-spp frames at full resolution until ~budget_s is used."""
    import numpy as np
    from oracle import pt_oracle as po
    cores = min(os.cpu_count() or 1, 16)
    osc = po.from_scene(scene, w, h, settings, sky_cube)
    acc = np.zeros((h, w, 4), np.float32)
    rays, frames, t0 = 0, 0, time.perf_counter()
    while True:
        acc, _, _, st = osc.render(1 + frames, acc, threads=cores)
        rays += st["rays"]
        frames += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or dt * (frames + 1) / frames > 1.25 * budget_s:
            break
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": f"{frames} full-resolution {settings.samples_per_pixel}-spp frames of the same scene ({rays} rays, {dt:.1f} s); "
                      f"oracle/pt_oracle.c, gcc -O2 -ffp-contract=off, pthreads"}


def main_group(args):
    """`--group`: the step of main() with one process and one host thread for all GPUs -- clear, the step's frames dealt
    over the GPUs by pt_group_render_frames, one pt_group_reduce onto GPU 0 -- timed by the host clock around a
    pt_group_sync on both sides.  No roofline block (that is the per-rank run's business), same metric."""
    import numpy as np
    import _pkgload
    pt = _pkgload.load()
    scenes = importlib.import_module("path_tracer_amd.scenes")
    factory, w, h, settings, frames = scenes.CONFIGS[args.workload]
    scene = factory()
    if args.frames:
        frames = args.frames
    scaling = args.scaling or DEFAULT_SCALING.get(args.workload, "weak")
    total = frames * args.gpus if scaling == "weak" else frames
    g = pt.RendererGroup(scene, w, h, list(range(args.gpus)), settings=settings, group_flags=pt.PT_GROUP_RCCL if args.group_rccl else 0)
    pano_w = min(4096, max(8, args.sky_size))
    g.set_sky_equirect(scenes.sky_equirect(pano_w, pano_w // 2), args.sky_size)

    def step(i: int):
        g.reset_ready_samples()
        g.render_frames(1 + i * total, total)
        g.reduce(0)

    for i in range(args.warmup):
        step(args.steps + i)
    g.sync()
    t0 = time.perf_counter()
    rays = 0
    for i in range(args.steps):
        step(i)
        rays += g.stats()["rays"]  # (reset clears the counters; reading them waits for the step: steps do not overlap)
    g.sync()
    elapsed = time.perf_counter() - t0
    frames_seen = int(g.read_sum(0)[0, 0, 3])
    out = {
        "metric": "Mrays/s, bunny 1080p 8-bounce, 1/2/4/8 GPUs; % FP32 VALU peak",
        "value": rays / elapsed / 1e6, "unit": "Mrays/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": args.workload, "width": w, "height": h, "depth": settings.depth,
                   "spp_per_step": total * settings.samples_per_pixel, "triangles": int(len(scene.triangles)),
                   "rays_per_step": rays // max(args.steps, 1), "frames_in_reduced_image": frames_seen,
                   "parallelism": f"pt_group x{args.gpus} ({scaling}), one process, one {'RCCL' if args.group_rccl else 'peer-copy'} reduce per step"},
        "roofline": None, "cpu_baseline": None,
    }
    assert frames_seen == total, (frames_seen, total)
    print(json.dumps(out), flush=True)
    g.close()


def main():
    args = parse_args()
    if args.group:
        return main_group(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Nothing in this process has touched the GPU yet: start the ranks as children and pass their output through.
        cmd = self_launch_command(args, sys.argv[1:], free_port())
        sys.exit(subprocess.run(cmd).returncode)

    import numpy as np
    import torch
    import torch.distributed as dist
    import _pkgload

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # (PTRT_BENCH_FORCE_REDUCE=1: a one-rank process group, so that the reduce path can be rehearsed on a single GPU)
    reduce_steps = world > 1 or os.environ.get("PTRT_BENCH_FORCE_REDUCE") == "1"
    if reduce_steps:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(free_port()))
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        else:
            dist.init_process_group("nccl", device_id=dev)

    pt = _pkgload.load()
    scenes = importlib.import_module("path_tracer_amd.scenes")
    multi = importlib.import_module("path_tracer_amd.multi")
    factory, w, h, settings, frames = scenes.CONFIGS[args.workload]
    scene = factory()
    if args.frames:
        frames = args.frames
    scaling = args.scaling or DEFAULT_SCALING.get(args.workload, "weak")

    # the renderer accumulates straight into a torch tensor so RCCL can reduce it in place
    accum = torch.zeros((h, w, 4), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev)
    base_flags = (pt.PT_FLAG_REBUILD_BVH if args.rebuild_bvh else 0) | (pt.PT_FLAG_REFERENCE_WALK if args.reference_walk else 0)
    r = pt.Renderer(scene, w, h, device=local_rank, settings=settings, stream=stream.cuda_stream, flags=base_flags)
    r.bind_accum(accum.data_ptr())
    pano_w = min(4096, max(8, args.sky_size))
    pano = scenes.sky_equirect(pano_w, pano_w // 2)
    r.set_sky_equirect(pano, args.sky_size)

    def step(i: int, events=None):
        accum.zero_()
        if events:
            events[0].record(stream)
        r.render_frames(*frames_of_step(multi, rank, world, scaling, frames, i))
        if events:
            events[1].record(stream)
        if reduce_steps:
            dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM)  # in place, on the launch stream's order (multi.reduce_sum_to_root)

    def fence():
        torch.cuda.synchronize(dev)
        if reduce_steps:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for i in range(args.warmup):
        step(args.steps + i)
    fence()
    rays_before = r.stats()["rays"]
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, ev[i])
    fence()
    elapsed = time.perf_counter() - t0
    rays = r.stats()["rays"] - rays_before
    kernel_ms = [a.elapsed_time(b) for a, b in ev]
    timed_stats = r.stats()  # (cumulative over warm-up and timed steps)

    tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    rr = torch.tensor([float(rays)], dtype=torch.float64, device=dev)
    if reduce_steps:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(rr, op=dist.ReduceOp.SUM)
    elapsed_max, rays_total = float(tt.item()), float(rr.item())

    # What kind of rays those were.  Tiles whose camera rays provably all leave the scene are summed by a plain loop
    # (sky_tiles_kernel: RNG, camera ray, sky texel, add -- no sphere or box test): flagged pixels x frames x samples of
    # every timed launch.  The rest is the megakernel's; of those, the rays that entered the BVH below the root are counted
    # by the detailed-stats build of the same kernel on the same frames (untimed; it renders every tile itself, the ray
    # set is the same).  Its first launch also prices the algorithmic work of one launch for the roofline.
    spp = max(settings.samples_per_pixel, 1)
    sky_px = sky_loop_pixels(r.read_sky_tiles(), w, h)
    launches = [frames_of_step(multi, rank, world, scaling, frames, i) for i in range(args.steps)]
    sky_rays = sum(sky_px * l[2] * spp for l in launches)
    r.set_flags(pt.PT_FLAG_DETAILED_STATS | base_flags)
    r.reset_ready_samples()
    st, walked = None, 0
    for l in launches:
        r.render_frames(*l)
        cum = r.stats()
        if st is None:
            st = dict(cum)
        walked = cum["walks"]
    r.set_flags(base_flags)
    kinds = torch.tensor([float(sky_rays), float(walked)], dtype=torch.float64, device=dev)
    if reduce_steps:
        dist.all_reduce(kinds, op=dist.ReduceOp.SUM)
    sky_rays_total, walked_total = (int(x) for x in kinds.tolist())

    if rank == 0:
        launch = launches[0]
        flops, abytes = algorithmic_flops(st), algorithmic_bytes(st, w * h * launch[2])
        k_ms = float(np.mean(kernel_ms))
        k_s = k_ms * 1e-3
        valu_tflops = flops / k_s / 1e12
        valu_frac = valu_tflops / FP32_VALU_PEAK_TFLOPS
        traffic, traffic_src = find_traffic(args.workload, launch[2], "sah" if args.rebuild_bvh else "reference-walk-only" if args.reference_walk else "reference")
        fabric_gbs = traffic / k_s / 1e9 if traffic else None
        fabric_frac = fabric_gbs / HBM_PEAK_GBS if traffic else None
        # Which roof: where the walk's working set lives.  Nodes + triangles + normals below the Infinity Cache's size are
        # served by L2 and that cache, not by HBM: the launch is bound by vector instruction issue and its figure is the
        # fraction of the FP32 vector peak (the metric's own).  The traffic counter (TCC read/write requests to the fabric)
        # then tallies cache-served misses too, so its rate is reported as `fabric_frac`, not as a fraction of an HBM roof.
        # Only a working set beyond that cache (the 10 M-triangle field) makes the counter an HBM figure and HBM the bound.
        ws = working_set_bytes(scene, timed_stats["guard_nodes"])
        hbm_bound = ws > INFINITY_CACHE_BYTES
        if hbm_bound and fabric_frac is not None:
            roof = {"bound": "hbm", "achieved": fabric_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": fabric_frac,
                    "hbm_frac": fabric_frac, "hbm_traffic_GBps": fabric_gbs}
        elif hbm_bound:
            # no committed PMC record for this launch length: the algorithmic bytes stand in (an upper bound on HBM bytes)
            roof = {"bound": "hbm", "achieved": abytes / k_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": abytes / k_s / 1e9 / HBM_PEAK_GBS,
                    "hbm_frac": None, "hbm_traffic_GBps": None}
        else:
            roof = {"bound": "valu", "achieved": valu_tflops, "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": valu_frac,
                    "fabric_frac": fabric_frac, "fabric_traffic_GBps": fabric_gbs}
        roof.update({
            "traffic": traffic, "traffic_source": traffic_src, "working_set_bytes": ws,
            "valu_frac": valu_frac, "valu_TFLOPs": valu_tflops,
            "kernel": "pool_kernel (+ accumulate_kernel; plan_kernel in two launches of every eight)" if r_uses_pool(pt, scene, settings, launch[2])
                      else "trace_kernel (+ plan_kernel and accumulate_kernel)",
            "kernel_ms": k_ms, "frames_per_launch": launch[2], "flops_per_launch": flops,
            "algorithmic_bytes_per_launch": abytes, "algorithmic_GBps": abytes / k_s / 1e9,
            "note": "no MFMA: nothing here is a dense contraction; valu_frac prices the shader's own flop count against the packed-FP32 "
                    "vector peak; traffic = memory-side bytes of one launch from the newest committed PMC record of this workload and "
                    "launch length (null without one): an HBM figure only when the working set exceeds the 256 MiB Infinity Cache"})
        rays_step = int(rays_total) // max(args.steps, 1)
        sky_step, walked_step = sky_rays_total // max(args.steps, 1), walked_total // max(args.steps, 1)
        out = {
            "metric": "Mrays/s, bunny 1080p 8-bounce, 1/2/4/8 GPUs; % FP32 VALU peak",
            "value": rays_total / elapsed_max / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "width": w, "height": h, "depth": settings.depth,
                       "spp_per_step": (frames * world if scaling == "weak" else frames) * settings.samples_per_pixel,
                       "spp_per_gpu_per_step": launch[2] * settings.samples_per_pixel, "triangles": int(len(scene.triangles)),
                       "bvh_nodes": int(len(scene.nodes)), "bvh_depth": st["bvh_depth"], "sky_cube": args.sky_size,
                       "rays_per_step": rays_step,
                       # disjoint kinds, summing to rays_per_step (means over the timed steps): summed by the sky-only tiles' loop
                       # without any intersection test; handled by the megakernel and ended at its sphere loop + root-box test;
                       # handled by the megakernel and walked through the BVH
                       "rays_sky_loop": sky_step, "rays_megakernel": rays_step - sky_step - walked_step, "rays_walked": walked_step,
                       # the library's second tree (bit-identical results): of all rays rendered so far, those that walked it, and
                       # of their hits those that failed the check and walked the reference tree as well
                       "second_tree_nodes": timed_stats["guard_nodes"],
                       "second_tree_walk_share": timed_stats["guarded"] / max(timed_stats["rays"], 1),
                       "second_tree_rewalk_share": timed_stats["rewalks"] / max(timed_stats["guarded"], 1),
                       "sky_loop_beside": st["sky_loop_beside"], "bvh": "library's binned SAH (PT_FLAG_REBUILD_BVH)" if args.rebuild_bvh else "reference (bvh.rs midpoint split)",
                       "parallelism": f"sample-split x{world} ({scaling}), one RCCL sum-reduce per step" if world > 1 else "single GPU"},
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            sky_cube = r.read_sky_cube()
            out["cpu_baseline"] = cpu_baseline(pt, scenes, scene, w, h, settings, sky_cube)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    r.close()
    if reduce_steps:
        dist.barrier()
        dist.destroy_process_group()


def r_uses_pool(pt, scene, settings, frames_per_launch: int) -> bool:
    """Mirror of the library's kernel choice (pt_kernels.h: PT_POOL_*), for the roofline block's label only."""
    tris, spp = len(scene.triangles), max(settings.samples_per_pixel, 1)
    return ((tris >= 128 and frames_per_launch >= 16) or (frames_per_launch == 1 and spp >= 48 and tris >= 8192)
            or (spp == 1 and frames_per_launch >= 8 and tris >= 512) or (frames_per_launch >= 4 and tris >= 8192))


if __name__ == "__main__":
    main()
