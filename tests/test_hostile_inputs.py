"""Hostile numbers -- NaN and infinite cameras, zero-area, enormous and tiny triangles, zero and negative radii, NaN
and infinite sky texels: the oracle stays in bounds under AddressSanitizer/UBSan (host driver tests/csrc/
hostile_inputs.c), and the HIP path, which evaluates the same index arithmetic, neither faults nor disagrees."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_in_bounds_under_sanitizers(tmp_path):
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    exe = tmp_path / "hostile"
    cmd = ["gcc", "-O1", "-g", "-std=c11", "-ffp-contract=off", "-mfma", "-pthread", "-fsanitize=address,undefined",
           "-fsanitize=float-cast-overflow", "-fno-sanitize-recover=undefined", os.path.join(ROOT, "tests", "csrc", "hostile_inputs.c"),
           "-o", str(exe), "-lm"]
    built = subprocess.run(cmd, capture_output=True, text=True)
    if built.returncode != 0 and "sanitize" in built.stderr and "cannot find" in built.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert built.returncode == 0, built.stderr
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:verify_asan_link_order=0")
    ran = subprocess.run([str(exe)], capture_output=True, text=True, env=env, timeout=300)
    assert ran.returncode == 0 and ran.stdout.strip().endswith("ok"), ran.stdout[-2000:] + ran.stderr[-4000:]


def _hostile_scenes(pt):
    M = pt.Material
    al = (0.5, 0.5, 0.5)
    tris = np.concatenate([
        pt.Plane((-1, -1, -2), (2, 0, 0), (0, 2, 0), al, M.Diffuse).triangles(),
        pt.Plane((0, 0, -1), (0, 0, 0), (0, 0, 0), al, M.Metal).triangles(),                     # zero area
        pt.Plane((-1e30, -1e30, -3), (2e30, 0, 0), (0, 2e30, 0), al, M.Dielectric).triangles(),  # enormous
        pt.Plane((0.3, 0.2, -1.5), (1e-30, 0, 0), (0, 1e-30, 0), al, M.Diffuse).triangles(),     # subnormal products
    ])
    spheres = [pt.Sphere((0, 0, -1), 0.5, (0.8, 0.3, 0.3), M.Dielectric), pt.Sphere((1, 0, -1), 0.0, (1, 1, 1), M.Metal),
               pt.Sphere((-1, 0, -1), -0.5, (1, 1, 1), M.Gizmo), pt.Sphere((0, float("nan"), -1), 0.5, (1, 1, 1), M.Diffuse)]
    nan, inf = float("nan"), float("inf")
    cams = [pt.Camera((0, 0, 1)), pt.Camera((nan, 0, 1)), pt.Camera((0, 0, 1), forward=(0, 0, inf)), pt.Camera((0, 0, 1), vfov=180.0),
            pt.Camera((0, 0, 1), focal_length=0.0), pt.Camera((0, 0, 1), right=(1e38, 0, 0), up=(0, -1e38, 0))]
    for k, cam in enumerate(cams):
        for mesh in (False, True):
            yield f"camera {k}, mesh {mesh}", pt.Scene(spheres, tris if mesh else np.zeros(0, pt.TRIANGLE_DTYPE), cam)


@pytest.mark.gpu
def test_hip_matches_oracle_on_hostile_inputs(pt, po):
    sky = np.full((6, 8, 8, 4), 0.25, np.float32)
    sky.reshape(-1)[::97] = np.nan
    sky.reshape(-1)[::89] = np.inf
    st = pt.Settings(2, 6, 1e-4, 1000.0)
    w, h = 37, 23
    for what, scene in _hostile_scenes(pt):
        for flags in (0, pt.PT_FLAG_TRIANGLE_MATERIALS):
            r = pt.Renderer(scene, w, h, settings=st, flags=flags)
            r.set_sky_cube(sky)
            osc = po.from_scene(scene, w, h, st, sky, flags=po.FLAG_TRIANGLE_MATERIALS if flags else 0)
            want = np.zeros((h, w, 4), np.float32)
            rays = 0
            for t in (0xFFFFFFF0, 0xFFFFFFF7, 5):
                r.render(t)
                want, _, _, ost = osc.render(t, want)
                rays += ost["rays"]
            got = r.read_sum()
            # NaN payloads may differ between the two instruction sets; everything else is the same bits
            same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
            assert same.all(), f"{what}, flags {flags}: {int((~same).sum())} of {got.size} sums differ"
            assert r.stats()["rays"] == rays, what
            r.close()
