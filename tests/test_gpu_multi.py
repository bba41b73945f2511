"""The sample split with the HIP renderer on every rank: two fresh worker processes share GPU 0, each renders its
share f = g (mod 2) of 8 frames (frames are independent given their time value, compute.wgsl:85), the images are
summed through gloo on host copies, and the result must be the one-process HIP image up to fp32 summation order,
with the oracle's ray count."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import multi_worker

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_hip_ranks_sum_to_the_single_rank_image(pt, po, scenes, tmp_path, megakernel):
    scene, sky, st, w, h, time0, frames = multi_worker.job(pt, scenes)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "sum.npz")
    env = dict(os.environ, PTRT_KERNEL=megakernel or "")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "multi_worker.py"), "--renderer", "hip", "--out", out,
                               "--rank", str(g), "--world", "2", "--port", str(port)], env=env) for g in range(2)]
    codes = [p.wait(timeout=110) for p in procs]
    assert codes == [0, 0]
    got = np.load(out)
    # one process, the same frames frame after frame
    r = pt.Renderer(scene, w, h, settings=st)
    r.set_sky_cube(sky)
    r.render_frames(time0, 1, frames)
    one, rays = r.read_sum(), r.stats()["rays"]
    r.close()
    assert (got["sum"][..., 3] == frames).all() and int(got["rays"]) == rays
    np.testing.assert_allclose(got["sum"], one, rtol=frames * 2.0 ** -23, atol=1e-6)
    osc = po.from_scene(scene, w, h, st, sky)
    want, orays = np.zeros((h, w, 4), np.float32), 0
    for f in range(frames):
        _, _, _, s = osc.render(time0 + f, want)
        orays += s["rays"]
    assert rays == orays
    np.testing.assert_allclose(got["sum"], want, rtol=frames * 2.0 ** -23, atol=1e-6)


@pytest.mark.library_choice
def test_bench_step_with_the_reduce_path_on_one_rank(tmp_path):
    """bench.py's N > 1 step -- RCCL group, sum-reduce of the device sums, barrier, all-reduces of time and rays -- run
    on this one GPU as a one-rank group (PTRT_BENCH_FORCE_REDUCE=1), beside the plain one-process run of the same
    steps: same rays, a well-formed line, and -- the mechanism, not the clock -- the GPU's stream pair still runs side by
    side with RCCL's streams in the process, so that the sky-only tiles' loop runs beside the megakernel (a step is 1.08 x
    as long when it does not; the wall-clock bound below is only a coarse guard against that and worse)."""
    import json
    args = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "3", "--no-cpu-baseline"]
    lines = []
    for forced in (False, True):
        env = dict(os.environ)
        env.pop("PTRT_KERNEL", None)
        if forced:
            env["PTRT_BENCH_FORCE_REDUCE"] = "1"
        out = subprocess.run(args, env=env, capture_output=True, text=True, timeout=110)
        assert out.returncode == 0, out.stderr[-2000:]
        lines.append(json.loads(out.stdout.strip().splitlines()[-1]))
    plain, forced = lines
    for d in lines:
        assert d["n_gpus"] == 1 and d["steps"] == 4 and d["unit"] == "Mrays/s" and d["roofline"]["kernel_ms"] > 0
    assert forced["config"]["rays_per_step"] == plain["config"]["rays_per_step"]
    for d in lines:
        assert d["config"]["sky_loop_beside"] == 1
        c = d["config"]
        assert c["rays_sky_loop"] > 0 and c["rays_walked"] > 0 and c["rays_sky_loop"] + c["rays_megakernel"] + c["rays_walked"] == c["rays_per_step"]
        assert d["roofline"]["bound"] == "valu" and d["roofline"]["frac"] == d["roofline"]["valu_frac"]
    assert forced["ms_per_step"] < 1.25 * plain["ms_per_step"], (forced["ms_per_step"], plain["ms_per_step"])
