"""ctypes binding of the CPU oracle (oracle/libpt_oracle.so).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the
product.  Build with `make -C oracle`."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpt_oracle.so")
FLAG_REFERENCE_DISPATCH = 1

TRIANGLE_DTYPE = np.dtype(
    [("a", "<f4", 3), ("_p0", "<f4"), ("b", "<f4", 3), ("_p1", "<f4"), ("c", "<f4", 3), ("_p2", "<f4"),
     ("na", "<f4", 3), ("_p3", "<f4"), ("nb", "<f4", 3), ("_p4", "<f4"), ("nc", "<f4", 3), ("_p5", "<f4"),
     ("albedo", "<f4", 3), ("material", "<u4")])
NODE_DTYPE = np.dtype([("min", "<f4", 3), ("left", "<u4"), ("max", "<f4", 3), ("count", "<u4")])


class SceneStruct(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("camera", C.c_void_p), ("settings", C.c_void_p),
                ("spheres", C.c_void_p), ("triangles", C.c_void_p), ("n_tris", C.c_uint32),
                ("nodes", C.c_void_p), ("n_nodes", C.c_uint32), ("tri_idx", C.c_void_p), ("n_idx", C.c_uint32),
                ("sky", C.c_void_p), ("sky_size", C.c_uint32), ("flags", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rays", "paths", "interior", "tri_tests", "tri_accept", "sphere_tests",
                                          "sky_lookups", "bounces", "sphere_accept", "overflow15", "max_stack")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        L = C.CDLL(LIB_PATH)
        vp, u32, f32 = C.c_void_p, C.c_uint32, C.c_float
        L.pto_render.restype = C.c_int
        L.pto_render.argtypes = [C.POINTER(SceneStruct), u32, vp, vp, vp, C.POINTER(Stats), C.c_int]
        L.pto_trace_ray.restype = C.c_int
        L.pto_trace_ray.argtypes = [C.POINTER(SceneStruct), vp, vp, vp, vp, vp, vp, vp, vp]
        L.pto_hit_aabb.restype = f32
        L.pto_hit_aabb.argtypes = [vp, vp, vp, vp]
        L.pto_sky_lookup.restype = None
        L.pto_sky_lookup.argtypes = [vp, u32, vp, vp]
        L.pto_random_unit.restype = None
        L.pto_random_unit.argtypes = [f32, f32, vp]
        L.pto_hash_rand.restype = f32
        L.pto_hash_rand.argtypes = [f32, f32]
        L.pto_hybrid_taus.restype = f32
        L.pto_hybrid_taus.argtypes = [vp]
        for name in ("pto_sinf", "pto_cosf", "pto_tanf", "pto_asinf"):
            getattr(L, name).restype = f32
            getattr(L, name).argtypes = [f32]
        L.pto_atan2f.restype = f32
        L.pto_atan2f.argtypes = [f32, f32]
        L.pto_bvh_build.restype = C.c_int64
        L.pto_bvh_build.argtypes = [vp, u32, vp, vp]
        L.pto_plane_triangles.restype = None
        L.pto_plane_triangles.argtypes = [vp, vp, vp, vp, u32, vp]
        L.pto_equirect_to_cube.restype = None
        L.pto_equirect_to_cube.argtypes = [vp, u32, u32, vp, u32]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class OracleScene:
    """Holds the byte buffers the kernel would be bound to (compute.wgsl:72-81) and keeps them alive."""

    def __init__(self, width, height, camera_buf, settings_buf, sphere_buf, triangles, nodes, tri_idx, sky=None, flags=0):
        self.width, self.height = int(width), int(height)
        self.camera = np.ascontiguousarray(camera_buf).view(np.uint8).copy()
        self.settings = np.ascontiguousarray(settings_buf).view(np.uint8).copy()
        self.spheres = np.ascontiguousarray(sphere_buf).view(np.uint8).copy()
        self.triangles = np.ascontiguousarray(triangles, TRIANGLE_DTYPE)
        self.nodes = np.ascontiguousarray(nodes, NODE_DTYPE)
        self.tri_idx = np.ascontiguousarray(tri_idx, np.uint32)
        self.sky = None if sky is None else np.ascontiguousarray(sky, np.float32)
        assert self.camera.nbytes == 64 and self.settings.nbytes == 16 and self.spheres.nbytes == 8208
        s = SceneStruct()
        s.width, s.height = self.width, self.height
        s.camera, s.settings, s.spheres = _p(self.camera).value, _p(self.settings).value, _p(self.spheres).value
        s.triangles, s.n_tris = (_p(self.triangles).value if len(self.triangles) else None), len(self.triangles)
        s.nodes, s.n_nodes = (_p(self.nodes).value if len(self.nodes) else None), len(self.nodes)
        s.tri_idx, s.n_idx = (_p(self.tri_idx).value if len(self.tri_idx) else None), len(self.tri_idx)
        if self.sky is not None:
            assert self.sky.ndim == 4 and self.sky.shape[0] == 6 and self.sky.shape[1] == self.sky.shape[2]
            s.sky, s.sky_size = _p(self.sky).value, self.sky.shape[1]
        s.flags = flags
        self.struct = s

    def render(self, time, accum=None, want_aov=False, threads=None):
        """One dispatch added into accum (H, W, 4).  Returns (accum, prim, t, stats dict)."""
        if accum is None:
            accum = np.zeros((self.height, self.width, 4), np.float32)
        prim = np.full((self.height, self.width), 0xFFFFFFFF, np.uint32) if want_aov else None
        t = np.zeros((self.height, self.width), np.float32) if want_aov else None
        st = Stats()
        threads = threads or min(os.cpu_count() or 1, 16)
        rc = lib().pto_render(C.byref(self.struct), time & 0xFFFFFFFF, _p(accum), _p(prim), _p(t), C.byref(st), threads)
        if rc != 0:
            raise RuntimeError("pto_render rejected its input")
        return accum, prim, t, st.as_dict()

    def trace_rays(self, origins, dirs):
        origins = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        dirs = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        n = len(origins)
        prim, t = np.zeros(n, np.uint32), np.zeros(n, np.float32)
        p, nrm, front = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32), np.zeros(n, np.uint32)
        hit = C.c_uint32()
        pr, tt, fr = C.c_uint32(), C.c_float(), C.c_uint32()
        L = lib()
        for i in range(n):
            L.pto_trace_ray(C.byref(self.struct), _p(origins[i]), _p(dirs[i]), C.addressof(hit), C.addressof(pr),
                            C.addressof(tt), _p(p[i]), _p(nrm[i]), C.addressof(fr))
            prim[i], t[i], front[i] = pr.value, tt.value, fr.value
        return prim, t, p, nrm, front


def from_scene(scene, width, height, settings, sky=None, flags=0) -> OracleScene:
    """Builds the oracle's view of a path_tracer_amd Scene WITHOUT using the product's BVH builder:
    the tree comes from the oracle's own restatement of bvh.rs."""
    nodes, idx = bvh_build(scene.triangles)
    return OracleScene(width, height, scene.camera.buffer(), settings.buffer(), pack_spheres(scene.spheres),
                       scene.triangles, nodes, idx, sky, flags)


def pack_spheres(spheres):
    """SphereDataBuffer::from (src/scene/sphere.rs:113-130) in numpy: count, 12 B pad, 256 x 32 B."""
    buf = np.zeros(8208, np.uint8)
    n = min(len(spheres), 256)
    buf[:4].view(np.uint32)[0] = n
    rec = buf[16:].view(np.float32).reshape(256, 8)
    for i, s in enumerate(spheres[:256]):
        rec[i, 0:3], rec[i, 3], rec[i, 4:7], rec[i, 7] = s.center, s.radius, s.albedo, float(int(s.material))
    return buf


def bvh_build(triangles):
    triangles = np.ascontiguousarray(triangles, TRIANGLE_DTYPE)
    n = len(triangles)
    if n == 0:
        return np.zeros(0, NODE_DTYPE), np.zeros(0, np.uint32)
    nodes = np.zeros(2 * n - 1, NODE_DTYPE)
    idx = np.zeros(n, np.uint32)
    used = lib().pto_bvh_build(_p(triangles), n, _p(nodes), _p(idx))
    assert used > 0
    return nodes[:used].copy(), idx


def plane_triangles(q, u, v, albedo, material):
    out = np.zeros(2, TRIANGLE_DTYPE)
    f = lambda s: np.ascontiguousarray(s, np.float32)
    q, u, v, albedo = f(q), f(u), f(v), f(albedo)
    lib().pto_plane_triangles(_p(q), _p(u), _p(v), _p(albedo), int(material), _p(out))
    return out


def hybrid_taus(x, y, time, n):
    """n draws from the seed of pixel (x, y) at `time`: list of (u32 xor, f32 bits), final state."""
    st = np.array([x, y, (x + time) & 0xFFFFFFFF, (y + time) & 0xFFFFFFFF], np.uint32)
    out = []
    for _ in range(n):
        v = lib().pto_hybrid_taus(_p(st))
        out.append((int(st[0] ^ st[1] ^ st[2] ^ st[3]), int(np.float32(v).view(np.uint32))))
    return out, [int(s) for s in st]


def equirect_to_cube(pano, size):
    pano = np.ascontiguousarray(pano, np.float32)
    out = np.zeros((6, size, size, 4), np.float32)
    lib().pto_equirect_to_cube(_p(pano), pano.shape[1], pano.shape[0], _p(out), size)
    return out
