"""Host-side binding of libptrt.so (the C ABI declared in include/ptrt.h).

This mirrors the reference's host objects so that tests read like its own code would:
``Scene`` (src/scene/mod.rs:26-43), ``Camera`` (src/scene/camera.rs:11-44),
``Sphere``/``Material`` (src/scene/sphere.rs, src/scene/mod.rs:18-24), ``Plane``
(src/scene/plane.rs) and ``Renderer`` with ``render`` / ``reset_ready_samples``
(src/renderer.rs:13-37, 551-557, 635-637).  All compute happens in the HIP library; if it
cannot be loaded every entry point raises -- there is no CPU fallback.

The directory name contains a hyphen, so import it through ``load()`` in ``_pkgload.py`` at the
repository root (or ``importlib``), under the module name ``path_tracer_amd``.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from enum import IntEnum
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PTRT_LIB", os.path.join(_HERE, "libptrt.so"))  # PTRT_LIB: diagnostic builds only

PT_FLAG_REFERENCE_DISPATCH = 1
PT_FLAG_AOV = 2
PT_FLAG_DETAILED_STATS = 4
PT_FLAG_TILE_KERNEL = 8
PT_FLAG_POOL_KERNEL = 16
PT_FLAG_TRIANGLE_MATERIALS = 32
PT_FLAG_REGROUP_SHORT = 64
PT_FLAG_REBUILD_BVH = 128
PT_FLAG_REFERENCE_WALK = 256
PT_GROUP_RCCL = 1
PRIM_NONE = 0xFFFFFFFF
PRIM_SPHERE = 0x80000000

# numpy views of the reference's Pod structs
TRIANGLE_DTYPE = np.dtype(
    [("a", "<f4", 3), ("_p0", "<f4"), ("b", "<f4", 3), ("_p1", "<f4"), ("c", "<f4", 3), ("_p2", "<f4"),
     ("na", "<f4", 3), ("_p3", "<f4"), ("nb", "<f4", 3), ("_p4", "<f4"), ("nc", "<f4", 3), ("_p5", "<f4"),
     ("albedo", "<f4", 3), ("material", "<u4")])
NODE_DTYPE = np.dtype([("min", "<f4", 3), ("left", "<u4"), ("max", "<f4", 3), ("count", "<u4")])
SPHERE_DTYPE = np.dtype([("center", "<f4", 3), ("radius", "<f4"), ("albedo", "<f4", 3), ("material", "<f4")])
SPHERE_DATA_DTYPE = np.dtype([("count", "<u4"), ("_pad", "<u4", 3), ("spheres", SPHERE_DTYPE, 256)])
CAMERA_DTYPE = np.dtype([("origin", "<f4", 3), ("focal_length", "<f4"), ("forward", "<f4", 3), ("vfov", "<f4"),
                         ("right", "<f4", 3), ("_p1", "<u4"), ("up", "<f4", 3), ("_p2", "<u4")])
SETTINGS_DTYPE = np.dtype([("spp", "<u4"), ("depth", "<u4"), ("t_min", "<f4"), ("t_max", "<f4")])
assert TRIANGLE_DTYPE.itemsize == 112 and NODE_DTYPE.itemsize == 32 and SPHERE_DTYPE.itemsize == 32
assert SPHERE_DATA_DTYPE.itemsize == 8208 and CAMERA_DTYPE.itemsize == 64 and SETTINGS_DTYPE.itemsize == 16


class PtError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"ptrt error {code}: {message}")
        self.code = code


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rays", "paths", "interior", "tri_tests", "tri_accept", "sphere_tests",
                                          "sky_lookups", "bounces", "sphere_accept", "frames", "bvh_depth", "walks",
                                          "sky_loop_beside", "guarded", "rewalks", "guard_nodes")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


_lib = None
# PTRT_KERNEL=pool|tile in the environment (read once, here; the library itself does not look at it) forces a
# megakernel for every Renderer of the process: used by the scripts under tools/.
_forced_flags = {"p": PT_FLAG_POOL_KERNEL, "t": PT_FLAG_TILE_KERNEL}.get(os.environ.get("PTRT_KERNEL", " ")[:1], 0)


def force_kernel(kind: Optional[str]):
    """Test aid: every Renderer made from now on gets PT_FLAG_POOL_KERNEL ("pool") or PT_FLAG_TILE_KERNEL ("tile")
    added to its flags; None restores the library's own choice."""
    global _forced_flags
    _forced_flags = {None: 0, "pool": PT_FLAG_POOL_KERNEL, "tile": PT_FLAG_TILE_KERNEL}[kind]

# every symbol include/ptrt.h declares: (name, restype, argtypes)
_VP, _U32, _I = C.c_void_p, C.c_uint32, C.c_int
ABI = [
    ("pt_create", _I, [_I, _U32, _U32, C.POINTER(_VP)]),
    ("pt_destroy", None, [_VP]),
    ("pt_last_error", C.c_char_p, [_VP]),
    ("pt_set_stream", _I, [_VP, _VP]),
    ("pt_set_flags", _I, [_VP, _U32]),
    ("pt_set_mesh", _I, [_VP, _VP, _U32, _VP, _U32, _VP, _U32]),
    ("pt_set_sky_cube", _I, [_VP, _VP, _U32]),
    ("pt_set_sky_equirect", _I, [_VP, _VP, _U32, _U32, _U32]),
    ("pt_read_sky_cube", _I, [_VP, _VP, C.c_size_t, C.POINTER(_U32)]),
    ("pt_set_camera", _I, [_VP, _VP]),
    ("pt_set_spheres", _I, [_VP, _VP]),
    ("pt_set_settings", _I, [_VP, _VP]),
    ("pt_render", _I, [_VP, _U32]),
    ("pt_render_frames", _I, [_VP, _U32, _U32, _U32]),
    ("pt_render_frames_static", _I, [_VP, _U32, _U32, _U32]),
    ("pt_reset", _I, [_VP]),
    ("pt_sync", _I, [_VP]),
    ("pt_read_sum_rgba32f", _I, [_VP, _VP]),
    ("pt_read_mean_rgba32f", _I, [_VP, _VP]),
    ("pt_read_rgba8", _I, [_VP, _VP]),
    ("pt_set_window", _I, [_VP, _U32]),
    ("pt_read_window", _I, [_VP, _VP, _VP, C.POINTER(_U32)]),
    ("pt_read_aov", _I, [_VP, _VP, _VP]),
    ("pt_read_sky_tiles", _I, [_VP, _VP, C.c_size_t, C.POINTER(_U32)]),
    ("pt_stats", _I, [_VP, C.POINTER(Stats)]),
    ("pt_accum_device_ptr", _VP, [_VP]),
    ("pt_bind_accum", _I, [_VP, _VP]),
    ("pt_trace_rays", _I, [_VP, _VP, _VP, _U32, _VP, _VP, _VP, _VP, _VP]),
    ("pt_trace_rays_walked", _I, [_VP, _VP, _VP, _U32, _VP, _VP, _VP, _VP, _VP, _VP]),
    ("pt_group_create", _I, [C.POINTER(_I), _I, _U32, _U32, _U32, C.POINTER(_VP)]),
    ("pt_group_destroy", None, [_VP]),
    ("pt_group_last_error", C.c_char_p, [_VP]),
    ("pt_group_size", _I, [_VP]),
    ("pt_group_member", _VP, [_VP, _I]),
    ("pt_group_set_flags", _I, [_VP, _U32]),
    ("pt_group_set_mesh", _I, [_VP, _VP, _U32, _VP, _U32, _VP, _U32]),
    ("pt_group_set_sky_cube", _I, [_VP, _VP, _U32]),
    ("pt_group_set_sky_equirect", _I, [_VP, _VP, _U32, _U32, _U32]),
    ("pt_group_set_camera", _I, [_VP, _VP]),
    ("pt_group_set_spheres", _I, [_VP, _VP]),
    ("pt_group_set_settings", _I, [_VP, _VP]),
    ("pt_group_render_frames", _I, [_VP, _U32, _U32]),
    ("pt_group_reduce", _I, [_VP, _I]),
    ("pt_group_reset", _I, [_VP]),
    ("pt_group_sync", _I, [_VP]),
    ("pt_group_stats", _I, [_VP, C.POINTER(Stats)]),
    ("pt_bvh_build", C.c_int64, [_VP, _U32, _VP, _VP]),
    ("pt_bvh_build_sah", C.c_int64, [_VP, _U32, _VP, _VP]),
    ("pt_bvh_build_sah_padded", C.c_int64, [_VP, _U32, _VP, _VP, _VP]),
    ("pt_plane_triangles", _I, [_VP, _VP, _VP, _VP, _U32, _VP]),
    ("pt_pack_spheres", _I, [_VP, _U32, _VP]),
    ("pt_obj_load", _I, [C.c_char_p, _VP, _U32, C.POINTER(_U32)]),
    ("pt_hdr_decode", _I, [_VP, C.c_size_t, _VP, C.c_size_t, C.POINTER(_U32), C.POINTER(_U32)]),
    ("pt_host_error", C.c_char_p, []),
]


def lib() -> C.CDLL:
    """Loads libptrt.so (built in-tree by __graft_entry__.build()).  Raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PtError(-4, f"{LIB_PATH} not built; run `make -C path-tracer_amd/csrc` (there is no CPU fallback)")
        try:
            # torch bundles its own HIP runtime under the same soname; whichever copy is loaded
            # first serves the whole process, and torch only works with its own.
            import torch  # noqa: F401
        except ImportError:
            pass
        handle = C.CDLL(LIB_PATH)
        for name, res, args in ABI:
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _host_check(rc: int):
    if rc < 0:
        raise PtError(int(rc), lib().pt_host_error().decode())


class Material(IntEnum):  # src/scene/mod.rs:18-24
    Diffuse = 0
    Metal = 1
    Dielectric = 2
    Gizmo = 3


@dataclass
class Sphere:  # SphereDescriptor, src/scene/sphere.rs:10-15
    center: Sequence[float]
    radius: float
    albedo: Sequence[float]
    material: Material = Material.Diffuse


@dataclass
class Plane:  # src/scene/plane.rs:7-13
    q: Sequence[float]
    u: Sequence[float]
    v: Sequence[float]
    albedo: Sequence[float] = (1.0, 1.0, 1.0)
    material: Material = Material.Diffuse

    def triangles(self) -> np.ndarray:
        out = np.zeros(2, TRIANGLE_DTYPE)
        f = lambda s: np.ascontiguousarray(s, np.float32)
        q, u, v, al = f(self.q), f(self.u), f(self.v), f(self.albedo)
        _host_check(lib().pt_plane_triangles(_ptr(q), _ptr(u), _ptr(v), _ptr(al), int(self.material), _ptr(out)))
        return out


@dataclass
class Camera:  # src/scene/camera.rs:11-44
    origin: Sequence[float] = (1.0, 1.0, 4.7)
    forward: Sequence[float] = (0.0, 0.0, -1.0)
    right: Sequence[float] = (1.0, 0.0, 0.0)
    up: Sequence[float] = (0.0, 1.0, 0.0)
    focal_length: float = 1.0
    vfov: float = 75.0

    @staticmethod
    def from_yaw_pitch(origin, yaw_deg=-90.0, pitch_deg=0.0, focal_length=1.0, vfov=75.0) -> "Camera":
        """Basis as CameraController::update_camera builds it (camera.rs:214-226), in f32."""
        f32 = np.float32
        yaw, pitch = np.deg2rad(f32(yaw_deg), dtype=f32), np.deg2rad(f32(pitch_deg), dtype=f32)
        fwd = np.array([np.cos(yaw) * np.cos(pitch), np.sin(pitch), np.sin(yaw) * np.cos(pitch)], f32)
        norm = lambda v: (v * (f32(1.0) / np.sqrt(np.sum(v * v, dtype=f32), dtype=f32))).astype(f32)
        fwd = norm(fwd)
        right = norm(np.cross(fwd, np.array([0, 1, 0], f32)).astype(f32))
        up = norm(np.cross(right, fwd).astype(f32))
        return Camera(tuple(origin), tuple(fwd), tuple(right), tuple(up), focal_length, vfov)

    def buffer(self) -> np.ndarray:  # CameraBuffer::from, camera.rs:86-99
        b = np.zeros(1, CAMERA_DTYPE)
        b["origin"], b["forward"], b["right"], b["up"] = self.origin, self.forward, self.right, self.up
        b["focal_length"], b["vfov"] = self.focal_length, self.vfov
        return b


@dataclass
class Settings:  # src/renderer.rs:604-611, defaults :411-416
    samples_per_pixel: int = 1
    depth: int = 32
    t_min: float = 0.0001
    t_max: float = 1000.0

    def buffer(self) -> np.ndarray:
        b = np.zeros(1, SETTINGS_DTYPE)
        b["spp"], b["depth"], b["t_min"], b["t_max"] = self.samples_per_pixel, self.depth, self.t_min, self.t_max
        return b


def pack_spheres(spheres: Sequence[Sphere]) -> np.ndarray:
    """SphereDataBuffer::from(&Vec<Sphere>), src/scene/sphere.rs:113-130."""
    raw = np.zeros(len(spheres), SPHERE_DTYPE)
    for i, s in enumerate(spheres):
        raw[i] = (tuple(s.center), s.radius, tuple(s.albedo), float(int(s.material)))
    out = np.zeros(1, SPHERE_DATA_DTYPE)
    _host_check(lib().pt_pack_spheres(_ptr(raw), len(spheres), _ptr(out)))
    return out


def bvh_build(triangles: np.ndarray):
    """Bvh::from_triangles, src/scene/bvh.rs:34-66 -> (nodes, triangle_indices)."""
    triangles = np.ascontiguousarray(triangles, TRIANGLE_DTYPE)
    n = len(triangles)
    if n == 0:
        return np.zeros(0, NODE_DTYPE), np.zeros(0, np.uint32)
    nodes = np.zeros(2 * n - 1, NODE_DTYPE)
    idx = np.zeros(n, np.uint32)
    used = lib().pt_bvh_build(_ptr(triangles), n, _ptr(nodes), _ptr(idx))
    _host_check(used)
    return nodes[:used].copy(), idx


def bvh_build_sah(triangles: np.ndarray, pad=None):
    """pt_bvh_build_sah: a binned-SAH tree in the reference's node format -> (nodes, triangle_indices).  Not the
    reference's tree: what PT_FLAG_REBUILD_BVH walks (opt-in, outside the bit-for-bit contract).  With `pad` (one float
    per triangle) pt_bvh_build_sah_padded: the tree over bounds grown by that much."""
    triangles = np.ascontiguousarray(triangles, TRIANGLE_DTYPE)
    n = len(triangles)
    if n == 0:
        return np.zeros(0, NODE_DTYPE), np.zeros(0, np.uint32)
    nodes = np.zeros(2 * n - 1, NODE_DTYPE)
    idx = np.zeros(n, np.uint32)
    if pad is None:
        used = lib().pt_bvh_build_sah(_ptr(triangles), n, _ptr(nodes), _ptr(idx))
    else:
        pad = np.ascontiguousarray(pad, np.float32)
        assert pad.shape == (n,)
        used = lib().pt_bvh_build_sah_padded(_ptr(triangles), n, _ptr(pad), _ptr(nodes), _ptr(idx))
    if used < 0:
        raise PtError(int(used), "pt_bvh_build_sah failed")
    return nodes[:used].copy(), idx


def load_obj(path: str) -> np.ndarray:
    """Model::from_obj, src/model.rs:111-213 (v / vn / f subset) -> triangle records."""
    n = C.c_uint32(0)
    _host_check(lib().pt_obj_load(path.encode(), None, 0, C.byref(n)))
    out = np.zeros(n.value, TRIANGLE_DTYPE)
    if n.value:
        _host_check(lib().pt_obj_load(path.encode(), _ptr(out), n.value, C.byref(n)))
    return out


def decode_hdr(data: bytes) -> np.ndarray:
    """Radiance .hdr bytes -> (h, w, 4) float32 panorama, as CubeTexture::from_equirectangular_hdri decodes
    its `data` argument (src/texture.rs:179-188); feed it to Renderer.set_sky_equirect."""
    buf = np.frombuffer(bytes(data), np.uint8)
    w, h = C.c_uint32(0), C.c_uint32(0)
    _host_check(lib().pt_hdr_decode(_ptr(buf), buf.size, None, 0, C.byref(w), C.byref(h)))
    out = np.zeros((h.value, w.value, 4), np.float32)
    _host_check(lib().pt_hdr_decode(_ptr(buf), buf.size, _ptr(out), out.shape[0] * out.shape[1], C.byref(w), C.byref(h)))
    return out


@dataclass
class Scene:
    """Scene::new builds the BVH from the triangle list (src/scene/mod.rs:35-43)."""
    spheres: List[Sphere]
    triangles: np.ndarray
    camera: Camera
    nodes: np.ndarray = field(init=False)
    triangle_indices: np.ndarray = field(init=False)

    def __post_init__(self):
        self.triangles = np.ascontiguousarray(self.triangles, TRIANGLE_DTYPE)
        self.nodes, self.triangle_indices = bvh_build(self.triangles)


class Renderer:
    """Renderer::new(device, queue, surface_config, scene) / render / reset_ready_samples."""

    def __init__(self, scene: Scene, width: int, height: int, device: int = 0, settings: Optional[Settings] = None,
                 flags: int = 0, stream: int = 0):
        self._ctx = C.c_void_p()
        self._l = lib()
        rc = self._l.pt_create(device, width, height, C.byref(self._ctx))
        if rc != 0:
            raise PtError(rc, self._l.pt_last_error(None).decode())
        self.width, self.height = width, height
        self.settings = settings or Settings()
        if stream:
            self._ck(self._l.pt_set_stream(self._ctx, C.c_void_p(stream)))
        if flags or _forced_flags:
            self.set_flags(flags)
        self.set_mesh(scene.triangles, scene.nodes, scene.triangle_indices)
        self.update_scene(scene)

    def _ck(self, rc: int):
        if rc != 0:
            raise PtError(rc, self._l.pt_last_error(self._ctx).decode())

    def close(self):
        if self._ctx:
            self._l.pt_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_flags(self, flags: int):
        if not flags & (PT_FLAG_POOL_KERNEL | PT_FLAG_TILE_KERNEL):
            flags |= _forced_flags
        self._ck(self._l.pt_set_flags(self._ctx, flags))

    def set_mesh(self, triangles: np.ndarray, nodes: np.ndarray, tri_idx: np.ndarray):
        triangles = np.ascontiguousarray(triangles, TRIANGLE_DTYPE)
        nodes = np.ascontiguousarray(nodes, NODE_DTYPE)
        tri_idx = np.ascontiguousarray(tri_idx, np.uint32)
        self._ck(self._l.pt_set_mesh(self._ctx, _ptr(triangles), len(triangles), _ptr(nodes), len(nodes),
                                     _ptr(tri_idx), len(tri_idx)))

    def set_sky_cube(self, faces: np.ndarray):
        faces = np.ascontiguousarray(faces, np.float32)
        assert faces.ndim == 4 and faces.shape[0] == 6 and faces.shape[1] == faces.shape[2] and faces.shape[3] == 4
        self._ck(self._l.pt_set_sky_cube(self._ctx, _ptr(faces), faces.shape[1]))

    def set_sky_equirect(self, pano: np.ndarray, cube_size: int):
        pano = np.ascontiguousarray(pano, np.float32)
        assert pano.ndim == 3 and pano.shape[2] == 4
        self._ck(self._l.pt_set_sky_equirect(self._ctx, _ptr(pano), pano.shape[1], pano.shape[0], cube_size))

    def read_sky_cube(self) -> np.ndarray:
        size = C.c_uint32(0)
        self._ck(self._l.pt_read_sky_cube(self._ctx, None, 0, C.byref(size)))
        out = np.zeros((6, size.value, size.value, 4), np.float32)
        if size.value:
            self._ck(self._l.pt_read_sky_cube(self._ctx, _ptr(out), out.size, C.byref(size)))
        return out

    def update_scene(self, scene: Scene):
        """The per-frame uniform uploads of Renderer::update_buffers (renderer.rs:524-548)."""
        cam, sph, st = scene.camera.buffer(), pack_spheres(scene.spheres), self.settings.buffer()
        self._ck(self._l.pt_set_camera(self._ctx, _ptr(cam)))
        self._ck(self._l.pt_set_spheres(self._ctx, _ptr(sph)))
        self._ck(self._l.pt_set_settings(self._ctx, _ptr(st)))

    def set_settings(self, settings: Settings):
        self.settings = settings
        st = settings.buffer()
        self._ck(self._l.pt_set_settings(self._ctx, _ptr(st)))

    def render(self, time: int):
        self._ck(self._l.pt_render(self._ctx, time & 0xFFFFFFFF))

    def render_frames(self, time0: int, stride: int, n_frames: int):
        self._ck(self._l.pt_render_frames(self._ctx, time0 & 0xFFFFFFFF, stride, n_frames))

    def render_frames_static(self, time0: int, stride: int, n_frames: int):
        """pt_render_frames without any device allocation (scheduling memory from the library's static array)."""
        self._ck(self._l.pt_render_frames_static(self._ctx, time0 & 0xFFFFFFFF, stride, n_frames))

    def reset_ready_samples(self):
        self._ck(self._l.pt_reset(self._ctx))

    def sync(self):
        self._ck(self._l.pt_sync(self._ctx))

    def read_sum(self) -> np.ndarray:
        out = np.zeros((self.height, self.width, 4), np.float32)
        self._ck(self._l.pt_read_sum_rgba32f(self._ctx, _ptr(out)))
        return out

    def read_mean(self) -> np.ndarray:
        out = np.zeros((self.height, self.width, 4), np.float32)
        self._ck(self._l.pt_read_mean_rgba32f(self._ctx, _ptr(out)))
        return out

    def read_rgba8(self) -> np.ndarray:
        out = np.zeros((self.height, self.width, 4), np.uint8)
        self._ck(self._l.pt_read_rgba8(self._ctx, _ptr(out)))
        return out

    def set_window(self, sample_size: int):
        """ProgressiveRendering.sample_size (renderer.rs:613-646): keep the newest frames as rgba8unorm images; 0 = off."""
        self._ck(lib().pt_set_window(self._ctx, sample_size))

    def window_frames_shown(self) -> int:
        n = C.c_uint32(0)
        self._ck(lib().pt_read_window(self._ctx, None, None, C.byref(n)))
        return n.value

    def read_window(self):
        """What copy.wgsl:27-37 shows after the last frame: (mean rgba32f, the same as 8-bit linear, frames averaged)."""
        mean = np.zeros((self.height, self.width, 4), np.float32)
        q = np.zeros((self.height, self.width, 4), np.uint8)
        n = C.c_uint32(0)
        self._ck(lib().pt_read_window(self._ctx, _ptr(mean), _ptr(q), C.byref(n)))
        return mean, q, n.value

    def read_aov(self):
        prim = np.zeros((self.height, self.width), np.uint32)
        t = np.zeros((self.height, self.width), np.float32)
        self._ck(self._l.pt_read_aov(self._ctx, _ptr(prim), _ptr(t)))
        return prim, t

    def read_sky_tiles(self) -> np.ndarray:
        """(ceil(height/8), ceil(width/8)) flags: tiles the last planned launch summed in its sky-only loop."""
        n = C.c_uint32(0)
        self._ck(self._l.pt_read_sky_tiles(self._ctx, None, 0, C.byref(n)))
        out = np.zeros(n.value, np.uint32)
        self._ck(self._l.pt_read_sky_tiles(self._ctx, _ptr(out), out.size, C.byref(n)))
        return out.reshape((self.height + 7) // 8, (self.width + 7) // 8)

    def stats(self) -> dict:
        s = Stats()
        self._ck(self._l.pt_stats(self._ctx, C.byref(s)))
        return s.as_dict()

    def accum_device_ptr(self) -> int:
        return int(self._l.pt_accum_device_ptr(self._ctx) or 0)

    def bind_accum(self, device_ptr: int):
        self._ck(self._l.pt_bind_accum(self._ctx, C.c_void_p(device_ptr)))

    def trace_rays(self, origins: np.ndarray, dirs: np.ndarray):
        origins = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        dirs = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        n = len(origins)
        prim, t = np.zeros(n, np.uint32), np.zeros(n, np.float32)
        p, nrm, front = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32), np.zeros(n, np.uint32)
        self._ck(self._l.pt_trace_rays(self._ctx, _ptr(origins), _ptr(dirs), n, _ptr(prim), _ptr(t), _ptr(p),
                                       _ptr(nrm), _ptr(front)))
        return prim, t, p, nrm, front

    def trace_rays_walked(self, origins: np.ndarray, dirs: np.ndarray):
        """pt_trace_rays_walked: trace_rays' outputs plus, per ray, which tree answered (0 / 1 / 2, see include/ptrt.h)."""
        origins = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        dirs = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        n = len(origins)
        prim, t, walked = np.zeros(n, np.uint32), np.zeros(n, np.float32), np.zeros(n, np.uint32)
        p, nrm, front = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32), np.zeros(n, np.uint32)
        self._ck(self._l.pt_trace_rays_walked(self._ctx, _ptr(origins), _ptr(dirs), n, _ptr(prim), _ptr(t), _ptr(p),
                                              _ptr(nrm), _ptr(front), _ptr(walked)))
        return prim, t, p, nrm, front, walked


class RendererGroup:
    """pt_group_*: one Renderer per GPU behind one handle, driven from this one thread.  Frames are dealt over the members
    (member g renders time0 + g, time0 + g + G, ...: frames are independent given their time value, compute.wgsl:85) and
    `reduce` adds every member's sums onto one member's."""

    def __init__(self, scene: Scene, width: int, height: int, devices: Sequence[int], settings: Optional[Settings] = None,
                 flags: int = 0, group_flags: int = 0):
        self._l = lib()
        self._g = C.c_void_p()
        devs = (C.c_int * len(devices))(*devices)
        rc = self._l.pt_group_create(devs, len(devices), width, height, group_flags, C.byref(self._g))
        if rc != 0:
            raise PtError(rc, self._l.pt_group_last_error(None).decode())
        self.width, self.height, self.devices = width, height, list(devices)
        self.settings = settings or Settings()
        if flags or _forced_flags:
            self.set_flags(flags)
        tri = np.ascontiguousarray(scene.triangles, TRIANGLE_DTYPE)
        nodes = np.ascontiguousarray(scene.nodes, NODE_DTYPE)
        idx = np.ascontiguousarray(scene.triangle_indices, np.uint32)
        self._ck(self._l.pt_group_set_mesh(self._g, _ptr(tri), len(tri), _ptr(nodes), len(nodes), _ptr(idx), len(idx)))
        self.update_scene(scene)

    def _ck(self, rc: int):
        if rc != 0:
            raise PtError(rc, self._l.pt_group_last_error(self._g).decode())

    def close(self):
        if self._g:
            self._l.pt_group_destroy(self._g)
            self._g = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return int(self._l.pt_group_size(self._g))

    def set_flags(self, flags: int):
        if not flags & (PT_FLAG_POOL_KERNEL | PT_FLAG_TILE_KERNEL):
            flags |= _forced_flags
        self._ck(self._l.pt_group_set_flags(self._g, flags))

    def update_scene(self, scene: Scene):
        cam, sph, st = scene.camera.buffer(), pack_spheres(scene.spheres), self.settings.buffer()
        self._ck(self._l.pt_group_set_camera(self._g, _ptr(cam)))
        self._ck(self._l.pt_group_set_spheres(self._g, _ptr(sph)))
        self._ck(self._l.pt_group_set_settings(self._g, _ptr(st)))

    def set_sky_cube(self, faces: np.ndarray):
        faces = np.ascontiguousarray(faces, np.float32)
        assert faces.ndim == 4 and faces.shape[0] == 6 and faces.shape[1] == faces.shape[2] and faces.shape[3] == 4
        self._ck(self._l.pt_group_set_sky_cube(self._g, _ptr(faces), faces.shape[1]))

    def set_sky_equirect(self, pano: np.ndarray, cube_size: int):
        pano = np.ascontiguousarray(pano, np.float32)
        assert pano.ndim == 3 and pano.shape[2] == 4
        self._ck(self._l.pt_group_set_sky_equirect(self._g, _ptr(pano), pano.shape[1], pano.shape[0], cube_size))

    def render_frames(self, time0: int, n_frames: int):
        self._ck(self._l.pt_group_render_frames(self._g, time0 & 0xFFFFFFFF, n_frames))

    def reduce(self, root: int = 0):
        self._ck(self._l.pt_group_reduce(self._g, root))

    def reset_ready_samples(self):
        self._ck(self._l.pt_group_reset(self._g))

    def sync(self):
        self._ck(self._l.pt_group_sync(self._g))

    def stats(self) -> dict:
        s = Stats()
        self._ck(self._l.pt_group_stats(self._g, C.byref(s)))
        return s.as_dict()

    def _member_call(self, i: int, fn, *args):
        ctx = self._l.pt_group_member(self._g, i)
        if not ctx:
            raise PtError(-1, f"member {i} out of range")
        rc = fn(C.c_void_p(ctx), *args)
        if rc != 0:
            raise PtError(rc, self._l.pt_last_error(C.c_void_p(ctx)).decode())

    def read_sum(self, member: int = 0) -> np.ndarray:
        out = np.zeros((self.height, self.width, 4), np.float32)
        self._member_call(member, self._l.pt_read_sum_rgba32f, _ptr(out))
        return out

    def read_mean(self, member: int = 0) -> np.ndarray:
        out = np.zeros((self.height, self.width, 4), np.float32)
        self._member_call(member, self._l.pt_read_mean_rgba32f, _ptr(out))
        return out
