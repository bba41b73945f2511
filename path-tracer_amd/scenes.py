"""Synthetic scenes for the benchmark configurations.

The reference's assets (bunny.obj, the .hdr panoramas) are not part of its repository
snapshot, so every scene here is generated from fixed formulas: the same arrays feed the HIP
renderer and the CPU oracle.  Default camera and sphere pattern follow src/app.rs:107-140 and
src/scene/camera.rs:34-44, 127-128, 214-226.
"""
from __future__ import annotations

import numpy as np

from . import (TRIANGLE_DTYPE, Camera, Material, Plane, Scene, Settings, Sphere)


def default_camera() -> Camera:
    # Camera::new origin, then the controller's yaw -90 / pitch 0 basis
    return Camera.from_yaw_pitch((1.0, 1.0, 4.7), -90.0, 0.0, 1.0, 75.0)


def cornell(camera: Camera | None = None) -> Scene:
    """Open-front box of five `Plane`s (10 triangles) around x,y in [0,2], z in [-1,1], two spheres."""
    W = (1.0, 1.0, 1.0)
    planes = [
        Plane((0, 0, -1), (2, 0, 0), (0, 0, 2), W),   # floor
        Plane((0, 2, -1), (2, 0, 0), (0, 0, 2), W),   # ceiling
        Plane((0, 0, -1), (2, 0, 0), (0, 2, 0), W),   # back
        Plane((0, 0, -1), (0, 0, 2), (0, 2, 0), W),   # left
        Plane((2, 0, -1), (0, 0, 2), (0, 2, 0), W),   # right
    ]
    tris = np.concatenate([p.triangles() for p in planes])
    spheres = [
        Sphere((0.6, 0.4, -0.3), 0.4, (0.8, 0.3, 0.3), Material.Diffuse),
        Sphere((1.4, 0.4, 0.3), 0.4, (1.0, 1.0, 1.0), Material.Dielectric),
    ]
    return Scene(spheres, tris, camera or default_camera())


def three_balls(camera: Camera | None = None) -> Scene:
    spheres = [
        Sphere((0.0, 0.0, -1.0), 0.5, (0.8, 0.3, 0.3), Material.Diffuse),
        Sphere((1.0, 0.0, -1.0), 0.5, (1.0, 1.0, 1.0), Material.Dielectric),
        Sphere((-1.0, 0.0, -1.0), 0.5, (0.8, 0.6, 0.2), Material.Metal),
        Sphere((0.0, -100.5, -1.0), 100.0, (0.8, 0.8, 0.0), Material.Diffuse),
    ]
    cam = camera or Camera.from_yaw_pitch((0.0, 0.6, 1.8), -90.0, -8.0, 1.0, 75.0)
    return Scene(spheres, np.zeros(0, TRIANGLE_DTYPE), cam)


def blob_mesh(n: int = 76, center=(0.0, 0.5, -1.0), radius: float = 1.0, yaw: float = 0.0) -> np.ndarray:
    """Closed displaced cube-sphere standing in for the Stanford bunny: 12*n*n triangles of
    similar size (69 312 at n = 76), smooth area-weighted vertex normals, O(1) scale."""
    g = np.linspace(-1.0, 1.0, n + 1)
    u, v = np.meshgrid(g, g, indexing="ij")
    one = np.ones_like(u)
    # six faces, each oriented so that (du x dv) points outwards
    faces_uv = [np.stack([one, u, v], -1), np.stack([-one, v, u], -1), np.stack([v, one, u], -1),
                np.stack([u, -one, v], -1), np.stack([u, v, one], -1), np.stack([v, u, -one], -1)]
    verts, tris_idx, base = [], [], 0
    ii, jj = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    for fv in faces_uv:
        verts.append(fv.reshape(-1, 3))
        a = base + ii * (n + 1) + jj
        b, c, d = a + (n + 1), a + 1, a + (n + 1) + 1
        tris_idx.append(np.stack([a, b, c], -1).reshape(-1, 3))
        tris_idx.append(np.stack([c, b, d], -1).reshape(-1, 3))
        base += (n + 1) * (n + 1)
    V = np.concatenate(verts)
    F = np.concatenate(tris_idx)
    # weld the shared edge/corner vertices of neighbouring faces
    key = np.round(V * n).astype(np.int64)
    _, first, inverse = np.unique(key, axis=0, return_index=True, return_inverse=True)
    V, F = V[first], inverse.reshape(-1)[F]
    D = V / np.linalg.norm(V, axis=1, keepdims=True)
    th, ph = np.arccos(np.clip(D[:, 1], -1, 1)), np.arctan2(D[:, 2], D[:, 0])
    r = 0.72 + 0.14 * np.sin(3 * th) * np.sin(4 * ph + 0.5) * np.sin(th) ** 2 + 0.08 * np.sin(7 * th + 1.0) * np.cos(5 * ph) * np.sin(th) ** 2 \
        + 0.05 * np.cos(6 * D[:, 0] + 2.0) * np.cos(5 * D[:, 1]) * np.cos(4 * D[:, 2] + 1.0)
    P = D * (r * radius)[:, None]
    cy, sy = np.cos(yaw), np.sin(yaw)
    P = np.stack([cy * P[:, 0] + sy * P[:, 2], P[:, 1], -sy * P[:, 0] + cy * P[:, 2]], -1)
    fn = np.cross(P[F[:, 1]] - P[F[:, 0]], P[F[:, 2]] - P[F[:, 0]])
    N = np.zeros_like(P)
    for k in range(3):
        np.add.at(N, F[:, k], fn)
    ln = np.linalg.norm(N, axis=1, keepdims=True)
    N = np.where(ln > 0, N / np.maximum(ln, 1e-30), np.array([0.0, 1.0, 0.0]))
    P = P + np.asarray(center, np.float64)
    tris = np.zeros(len(F), TRIANGLE_DTYPE)
    for k, (pn, nn) in enumerate((("a", "na"), ("b", "nb"), ("c", "nc"))):
        tris[pn] = P[F[:, k]].astype(np.float32)
        tris[nn] = N[F[:, k]].astype(np.float32)
    tris["albedo"] = 1.0
    tris["material"] = 0
    return tris


def bunny(camera: Camera | None = None, n: int = 76) -> Scene:
    """Mesh (~69k triangles at n = 76) on a large ground sphere, default camera."""
    tris = blob_mesh(n)
    spheres = [Sphere((0.0, -100.5, -1.0), 100.0, (0.8, 0.8, 0.0), Material.Diffuse)]
    return Scene(spheres, tris, camera or default_camera())


def instanced_field(grid: int = 12, pitch: float = 2.5, n: int = 76) -> Scene:
    """grid x grid copies of the mesh flattened into one triangle list (12x12 -> ~10.0 M)."""
    rng = np.random.RandomState(2)
    parts = []
    for i in range(grid):
        for j in range(grid):
            cx, cz = (i - (grid - 1) / 2) * pitch + 1.0, -1.0 - j * pitch
            parts.append(blob_mesh(n, (cx, 0.5, cz), 1.0, float(rng.uniform(0, 2 * np.pi))))
    spheres = [Sphere((0.0, -100.5, -1.0), 100.0, (0.8, 0.8, 0.0), Material.Diffuse)]
    cam = Camera.from_yaw_pitch((1.0, 6.0, 8.0), -90.0, -25.0, 1.0, 75.0)
    return Scene(spheres, np.concatenate(parts), cam)


def sky_equirect(width: int = 4096, height: int = 2048) -> np.ndarray:
    """Procedural panorama (RGBA32F, row 0 = zenith): horizon-to-zenith gradient plus a sun lobe."""
    v = (np.arange(height, dtype=np.float64) + 0.5) / height
    u = (np.arange(width, dtype=np.float64) + 0.5) / width
    lat = (0.5 - v)[:, None] * np.pi          # +pi/2 at row 0
    lon = (u[None, :] - 0.5) * 2.0 * np.pi
    d = np.stack([np.cos(lat) * np.cos(lon), np.sin(lat) * np.ones_like(lon), np.cos(lat) * np.sin(lon)], -1)
    up = np.clip(d[..., 1], -1.0, 1.0)
    t = 0.5 * (up + 1.0)
    sky = (1.0 - t)[..., None] * np.array([1.0, 1.0, 1.0]) + t[..., None] * np.array([0.5, 0.7, 1.0])
    ground = np.array([0.35, 0.32, 0.30])
    col = np.where((up < 0)[..., None], ground * (0.6 + 0.4 * (1 + up))[..., None], sky)
    sun = np.array([0.45, 0.75, 0.48])
    sun = sun / np.linalg.norm(sun)
    lobe = np.clip(d @ sun, 0.0, 1.0) ** 600.0
    col = col + lobe[..., None] * np.array([40.0, 36.0, 30.0])
    out = np.ones((height, width, 4), np.float32)
    out[..., :3] = col.astype(np.float32)
    return out


def sky_cube_direct(size: int) -> np.ndarray:
    """Small analytic cube map (6, size, size, 4) for tests that do not need the panorama path.
    Face orientation follows the Vulkan cube-map convention the lookup uses."""
    j, i = np.meshgrid(np.arange(size), np.arange(size), indexing="ij")
    s = (i + 0.5) / size * 2.0 - 1.0
    t = (j + 0.5) / size * 2.0 - 1.0
    one = np.ones_like(s)
    dirs = [np.stack([one, -t, -s], -1), np.stack([-one, -t, s], -1), np.stack([s, one, t], -1),
            np.stack([s, -one, -t], -1), np.stack([s, -t, one], -1), np.stack([-s, -t, -one], -1)]
    out = np.ones((6, size, size, 4), np.float32)
    for f, d in enumerate(dirs):
        d = d / np.linalg.norm(d, axis=-1, keepdims=True)
        tt = 0.5 * (d[..., 1] + 1.0)
        col = (1.0 - tt)[..., None] * np.array([1.0, 1.0, 1.0]) + tt[..., None] * np.array([0.5, 0.7, 1.0])
        col = col + (np.clip(d @ (np.array([0.45, 0.75, 0.48]) / np.linalg.norm([0.45, 0.75, 0.48])), 0, 1) ** 200.0)[..., None] * 8.0
        out[f, ..., :3] = col.astype(np.float32)
    return out


# BASELINE.json configurations: name -> (scene factory, width, height, Settings, frames)
CONFIGS = {
    "cornell_640x360_d4_spp16": (cornell, 640, 360, Settings(16, 4, 1e-4, 1000.0), 1),
    "three_balls_1080p_d8_spp256": (three_balls, 1920, 1080, Settings(1, 8, 1e-4, 1000.0), 256),
    "bunny_1080p_d8_spp256": (bunny, 1920, 1080, Settings(1, 8, 1e-4, 1000.0), 256),
    "bunny_4k_d8_spp4096": (bunny, 3840, 2160, Settings(1, 8, 1e-4, 1000.0), 4096),
    "field_4k_d8_spp1024": (instanced_field, 3840, 2160, Settings(1, 8, 1e-4, 1000.0), 1024),
    # the reference's own per-frame call: one dispatch of 1 spp per step (Renderer::render, renderer.rs:562-570),
    # at its default depth (renderer.rs:411-416) and at the benchmark's
    "interactive_1080p_d32_spp1": (bunny, 1920, 1080, Settings(1, 32, 1e-4, 1000.0), 1),
    "interactive_1080p_d8_spp1": (bunny, 1920, 1080, Settings(1, 8, 1e-4, 1000.0), 1),
}
