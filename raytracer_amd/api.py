"""ctypes binding of the C-ABI in include/rt_api.h (libraytracer_amd.so).

Host-side plumbing only: scene arrays in numpy, device pointers as integers.
There is no CPU rendering path here -- if the HIP library is missing or no GPU
is visible the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_LIB") or os.path.join(_HERE, "libraytracer_amd.so")     # RT_LIB: another build of the same library (experiments)

RT_OK = 0
(RT_ERR_INVALID, RT_ERR_IO, RT_ERR_PARSE, RT_ERR_NO_DEVICE, RT_ERR_HIP, RT_ERR_NOMEM, RT_ERR_UNSUPPORTED,
 RT_ERR_NO_SCENE) = range(1, 9)
RT_LIGHT_AMBIENT, RT_LIGHT_DIRECTIONAL, RT_LIGHT_POINT, RT_LIGHT_SPOT = 0, 1, 2, 3
RT_PROJ_ORTHO, RT_PROJ_PERSP = 0, 1
RT_ACCEL_AUTO, RT_ACCEL_BRUTE, RT_ACCEL_LBVH, RT_ACCEL_LBVH_PACKET, RT_ACCEL_LBVH_WIDE, RT_ACCEL_LBVH_TEAM = 0, 1, 2, 3, 4, 5
RT_ACCEL_ALT_TREE = 16      # query_rays only: OR-ed to an LBVH mode, walks the PLOC hierarchy held until the first render
RT_ACCEL_SAH_TREE = 32      # the same for the binned surface-area build (where load_scene built it)
TREE_NAMES = ("karras", "ploc", "sah")
RT_PARSE_REF_COMPAT = 1
RT_TILE = 32


class RtError(RuntimeError):
    def __init__(self, status: int, where: str):
        lib = load_library()
        msg = lib.rt_strerror(status).decode()
        detail = lib.rt_last_error().decode()
        super().__init__(f"{where}: {msg} ({status}){': ' + detail if detail else ''}")
        self.status = status


class RtLight(C.Structure):
    _fields_ = [("type", C.c_int32), ("color", C.c_float * 3), ("pos", C.c_float * 3),
                ("dir", C.c_float * 3), ("angle1", C.c_float), ("angle2", C.c_float)]


class RtSceneDesc(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("spp", C.c_int32),
                ("max_depth", C.c_int32), ("proj", C.c_int32), ("seed", C.c_uint32),
                ("cam_pos", C.c_float * 3), ("cam_dir", C.c_float * 3), ("cam_up", C.c_float * 3),
                ("cam_half_angle", C.c_float), ("background", C.c_float * 3), ("ambient", C.c_float * 3),
                ("n_spheres", C.c_int32), ("n_triangles", C.c_int32), ("n_materials", C.c_int32),
                ("n_lights", C.c_int32),
                ("spheres", C.c_void_p), ("sphere_mat", C.c_void_p), ("tri_verts", C.c_void_p),
                ("tri_normals", C.c_void_p), ("tri_flags", C.c_void_p), ("tri_mat", C.c_void_p),
                ("materials", C.c_void_p), ("lights", C.c_void_p),
                ("bvh_threshold", C.c_int32), ("bvh_depth", C.c_int32)]


class RtOptions(C.Structure):
    _fields_ = [("device", C.c_int32), ("accel", C.c_int32), ("lbvh_threshold", C.c_int32),
                ("batch_samples", C.c_int32), ("pool_rays", C.c_int64), ("stream", C.c_void_p)]


class RtStats(C.Structure):
    _fields_ = [("rays_primary", C.c_uint64), ("rays_secondary", C.c_uint64), ("rays_shadow", C.c_uint64),
                ("nodes_visited", C.c_uint64 * 2), ("sphere_tests", C.c_uint64 * 2), ("tri_tests", C.c_uint64 * 2),
                ("lane_node_tests", C.c_uint64 * 2), ("lane_prim_tests", C.c_uint64 * 2),
                ("ms_render", C.c_double), ("ms_extend", C.c_double), ("ms_shadow", C.c_double),
                ("ms_shade", C.c_double), ("ms_other", C.c_double),
                ("launches_extend", C.c_int32), ("launches_shadow", C.c_int32),
                ("batches", C.c_int32), ("retries", C.c_int32), ("accel_used", C.c_int32),
                ("generations", C.c_int32), ("ms_host_enqueue", C.c_double)]

    @property
    def rays(self) -> int:
        return self.rays_primary + self.rays_secondary + self.rays_shadow

    def as_dict(self) -> dict:
        return {n: (list(getattr(self, n)) if hasattr(getattr(self, n), '__len__') else getattr(self, n))
                for n, _ in self._fields_}


class RtTreeStats(C.Structure):
    _fields_ = [("primitives", C.c_int32), ("in_use", C.c_int32), ("depth", C.c_int32 * 3), ("reserved", C.c_int32),
                ("node_visits", C.c_double * 3), ("prim_tests", C.c_double * 3)]

    def as_dict(self) -> dict:
        return {"in_use": TREE_NAMES[self.in_use] if self.in_use >= 0 else None,
                "trees": {TREE_NAMES[t]: {"depth": self.depth[t], "expected_node_visits": round(self.node_visits[t], 3),
                                          "expected_prim_tests": round(self.prim_tests[t], 3)}
                          for t in range(3) if self.node_visits[t] > 0}}


PROGRESS_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint64, C.c_uint64)

_lib = None


def load_library():
    """Load libraytracer_amd.so from the package directory; raise if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C raytracer_amd/csrc` "
            "(or __graft_entry__.build()); there is no fallback implementation")
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, u32, u64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_uint64
    sig = {
        "rt_create": (C.c_int, [C.POINTER(vp), C.POINTER(RtOptions)]),
        "rt_destroy": (None, [vp]),
        "rt_strerror": (C.c_char_p, [C.c_int]),
        "rt_last_error": (C.c_char_p, []),
        "rt_validate_scene": (C.c_int, [C.POINTER(RtSceneDesc)]),
        "rt_load_scene": (C.c_int, [vp, C.POINTER(RtSceneDesc)]),
        "rt_set_progress_callback": (C.c_int, [vp, PROGRESS_FN, vp]),
        "rt_set_stage_times": (C.c_int, [vp, C.c_int]),
        "rt_render_multi": (C.c_int, [C.POINTER(vp), C.c_int, vp, C.POINTER(RtStats)]),
        "rt_render": (C.c_int, [vp, vp, vp, C.POINTER(RtStats)]),
        "rt_render_device": (C.c_int, [vp, vp, C.POINTER(RtStats)]),
        "rt_shard_tiles": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
        "rt_shard_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
        "rt_render_shard": (C.c_int, [vp, C.c_int, C.c_int, vp, C.POINTER(RtStats)]),
        "rt_assemble_shards": (C.c_int, [vp, vp, C.c_int, vp]),
        "rt_assemble_shards_noalloc": (C.c_int, [vp, vp, C.c_int, vp]),
        "rt_query_rays": (C.c_int, [vp, i64, vp, vp, vp, C.c_int, C.c_int, vp, vp]),
        "rt_primary_rays": (C.c_int, [vp, i64, vp, vp, vp, vp, vp]),
        "rt_jitter_hash": (u32, [u32, u64]),
        "rt_parse_scene_file": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.POINTER(RtSceneDesc)), C.c_char_p, C.c_size_t]),
        "rt_free_scene": (None, [C.POINTER(RtSceneDesc)]),
        "rt_output_name": (C.c_int, [C.c_char_p, C.c_char_p, C.c_size_t]),
        "rt_write_bmp24": (C.c_int, [C.c_char_p, vp, C.c_int, C.c_int]),
        "rt_set_work_counters": (C.c_int, [vp, C.c_int]),
        "rt_set_calibration": (C.c_int, [vp, C.c_int]),
        "rt_set_graphs": (C.c_int, [vp, C.c_int, C.POINTER(C.c_uint64)]),
        "rt_get_tree_stats": (C.c_int, [vp, C.POINTER(RtTreeStats)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)   # AttributeError if the library lacks a declared entry point
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


EXPORTED_SYMBOLS = [
    "rt_create", "rt_destroy", "rt_strerror", "rt_last_error", "rt_load_scene", "rt_render",
    "rt_render_device", "rt_shard_tiles", "rt_shard_bytes", "rt_render_shard", "rt_assemble_shards",
    "rt_query_rays", "rt_primary_rays", "rt_jitter_hash", "rt_parse_scene_file", "rt_free_scene",
    "rt_output_name", "rt_write_bmp24", "rt_set_work_counters", "rt_set_calibration", "rt_validate_scene", "rt_set_progress_callback",
    "rt_set_stage_times", "rt_render_multi", "rt_assemble_shards_noalloc", "rt_set_graphs", "rt_get_tree_stats",
]


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


@dataclass
class SceneArrays:
    """The scene in the flat layout of rt_scene_desc, held as numpy arrays."""
    width: int = 640
    height: int = 480
    spp: int = 1
    max_depth: int = 5
    proj: int = RT_PROJ_PERSP
    seed: int = 1
    cam_pos: np.ndarray = field(default_factory=lambda: np.zeros(3, np.float32))
    cam_dir: np.ndarray = field(default_factory=lambda: np.array([0, 0, 1], np.float32))
    cam_up: np.ndarray = field(default_factory=lambda: np.array([0, 1, 0], np.float32))
    cam_half_angle: float = float(np.float32(45.0 * 0.01745329251))
    background: np.ndarray = field(default_factory=lambda: np.zeros(3, np.float32))
    ambient: np.ndarray = field(default_factory=lambda: np.zeros(3, np.float32))
    spheres: np.ndarray = field(default_factory=lambda: np.zeros((0, 4), np.float32))
    sphere_mat: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    tri_verts: np.ndarray = field(default_factory=lambda: np.zeros((0, 9), np.float32))
    tri_normals: np.ndarray = field(default_factory=lambda: np.zeros((0, 9), np.float32))
    tri_flags: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    tri_mat: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    materials: np.ndarray = field(default_factory=lambda: np.array(
        [[0, 0, 0, 1, 1, 1, 0, 0, 0, 5, 0, 0, 0, 1]], np.float32))
    lights: np.ndarray = field(default_factory=lambda: np.zeros((0, 12), np.float32))  # type,col3,pos3,dir3,a1,a2
    output_name: str = "raytraced.bmp"
    bvh_threshold: int = 0      # `bvh_threshold n` of the scene file, 0 = not given
    bvh_depth: int = 0

    _ARRAYS = ("cam_pos", "cam_dir", "cam_up", "background", "ambient", "spheres", "sphere_mat",
               "tri_verts", "tri_normals", "tri_flags", "tri_mat", "materials", "lights")
    _SCALARS = ("width", "height", "spp", "max_depth", "proj", "seed", "cam_half_angle")

    def normalise(self) -> "SceneArrays":
        f32 = lambda a, shape: np.ascontiguousarray(np.asarray(a, np.float32).reshape(shape))
        i32 = lambda a: np.ascontiguousarray(np.asarray(a, np.int32).reshape(-1))
        self.cam_pos, self.cam_dir, self.cam_up = f32(self.cam_pos, 3), f32(self.cam_dir, 3), f32(self.cam_up, 3)
        self.background, self.ambient = f32(self.background, 3), f32(self.ambient, 3)
        self.spheres = f32(self.spheres, (-1, 4))
        self.tri_verts, self.tri_normals = f32(self.tri_verts, (-1, 9)), f32(self.tri_normals, (-1, 9))
        self.materials, self.lights = f32(self.materials, (-1, 14)), f32(self.lights, (-1, 12))
        self.sphere_mat, self.tri_flags, self.tri_mat = i32(self.sphere_mat), i32(self.tri_flags), i32(self.tri_mat)
        return self

    def desc(self):
        """(RtSceneDesc, keepalive) -- the descriptor borrows the numpy buffers."""
        self.normalise()
        nl = len(self.lights)
        la = (RtLight * max(nl, 1))()
        for i in range(nl):
            row = self.lights[i]
            la[i].type = int(row[0])
            la[i].color[:] = row[1:4]
            la[i].pos[:] = row[4:7]
            la[i].dir[:] = row[7:10]
            la[i].angle1, la[i].angle2 = float(row[10]), float(row[11])
        d = RtSceneDesc()
        d.width, d.height, d.spp, d.max_depth = int(self.width), int(self.height), int(self.spp), int(self.max_depth)
        d.proj, d.seed = int(self.proj), int(self.seed)
        d.cam_pos[:], d.cam_dir[:], d.cam_up[:] = self.cam_pos, self.cam_dir, self.cam_up
        d.cam_half_angle = float(self.cam_half_angle)
        d.background[:], d.ambient[:] = self.background, self.ambient
        d.n_spheres, d.n_triangles = len(self.spheres), len(self.tri_verts)
        d.n_materials, d.n_lights = len(self.materials), nl
        d.spheres, d.sphere_mat = _ptr(self.spheres).value, _ptr(self.sphere_mat).value
        d.tri_verts, d.tri_normals = _ptr(self.tri_verts).value, _ptr(self.tri_normals).value
        d.tri_flags, d.tri_mat = _ptr(self.tri_flags).value, _ptr(self.tri_mat).value
        d.materials = _ptr(self.materials).value
        d.lights = C.cast(la, C.c_void_p).value
        d.bvh_threshold, d.bvh_depth = int(self.bvh_threshold), int(self.bvh_depth)
        return d, (la, self)

    def copy(self, **changes) -> "SceneArrays":
        kw = {n: np.array(getattr(self, n), copy=True) for n in self._ARRAYS}
        kw.update({n: getattr(self, n) for n in self._SCALARS})
        kw["output_name"] = self.output_name
        kw["bvh_threshold"], kw["bvh_depth"] = self.bvh_threshold, self.bvh_depth
        kw.update(changes)
        return SceneArrays(**kw).normalise()

    def save(self, path: str) -> None:
        self.normalise()
        np.savez_compressed(path, **{n: getattr(self, n) for n in self._ARRAYS},
                            scalars=np.array([self.width, self.height, self.spp, self.max_depth, self.proj, self.seed], np.int64),
                            cam_half_angle=np.float32(self.cam_half_angle),
                            output_name=np.frombuffer(self.output_name.encode(), np.uint8),
                            **({"bvh": np.array([self.bvh_threshold, self.bvh_depth], np.int64)}
                               if (self.bvh_threshold or self.bvh_depth) else {}))

    @classmethod
    def load(cls, path: str) -> "SceneArrays":
        z = np.load(path, allow_pickle=False)
        sc = z["scalars"]
        kw = {n: z[n] for n in cls._ARRAYS}
        if "bvh" in z.files:
            kw["bvh_threshold"], kw["bvh_depth"] = int(z["bvh"][0]), int(z["bvh"][1])
        return cls(width=int(sc[0]), height=int(sc[1]), spp=int(sc[2]), max_depth=int(sc[3]), proj=int(sc[4]),
                   seed=int(sc[5]), cam_half_angle=float(z["cam_half_angle"]),
                   output_name=bytes(z["output_name"]).decode(), **kw).normalise()

    @classmethod
    def from_desc(cls, d: RtSceneDesc, output_name: str = "raytraced.bmp") -> "SceneArrays":
        def arr(ptr, n, dtype, cols=None):
            if n == 0 or not ptr:
                return np.zeros((0, cols) if cols else 0, dtype)
            count = n * (cols or 1)
            buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(ptr)
            a = np.frombuffer(buf, dtype=dtype, count=count).copy()
            return a.reshape(n, cols) if cols else a
        lights = np.zeros((d.n_lights, 12), np.float32)
        if d.n_lights:
            la = C.cast(d.lights, C.POINTER(RtLight))
            for i in range(d.n_lights):
                L = la[i]
                lights[i] = [L.type, *L.color, *L.pos, *L.dir, L.angle1, L.angle2]
        return cls(width=d.width, height=d.height, spp=d.spp, max_depth=d.max_depth, proj=d.proj, seed=d.seed,
                   cam_pos=np.array(d.cam_pos[:], np.float32), cam_dir=np.array(d.cam_dir[:], np.float32),
                   cam_up=np.array(d.cam_up[:], np.float32), cam_half_angle=float(d.cam_half_angle),
                   background=np.array(d.background[:], np.float32), ambient=np.array(d.ambient[:], np.float32),
                   spheres=arr(d.spheres, d.n_spheres, np.float32, 4), sphere_mat=arr(d.sphere_mat, d.n_spheres, np.int32),
                   tri_verts=arr(d.tri_verts, d.n_triangles, np.float32, 9),
                   tri_normals=arr(d.tri_normals, d.n_triangles, np.float32, 9),
                   tri_flags=arr(d.tri_flags, d.n_triangles, np.int32), tri_mat=arr(d.tri_mat, d.n_triangles, np.int32),
                   materials=arr(d.materials, d.n_materials, np.float32, 14), lights=lights,
                   output_name=output_name, bvh_threshold=d.bvh_threshold, bvh_depth=d.bvh_depth).normalise()


def parse_scene_file(path: str, ref_compat: bool = False) -> SceneArrays:
    """Parse a scene file in the reference's text format (host only, no GPU needed)."""
    lib = load_library()
    out = C.POINTER(RtSceneDesc)()
    name = C.create_string_buffer(4096)
    rc = lib.rt_parse_scene_file(os.fsencode(path), RT_PARSE_REF_COMPAT if ref_compat else 0, C.byref(out), name, 4096)
    if rc != RT_OK:
        raise RtError(rc, f"rt_parse_scene_file({path})")
    try:
        return SceneArrays.from_desc(out.contents, name.value.decode(errors="replace"))
    finally:
        lib.rt_free_scene(out)


def output_name(image_name: str) -> str:
    lib = load_library()
    buf = C.create_string_buffer(4096)
    rc = lib.rt_output_name(image_name.encode(), buf, 4096)
    if rc != RT_OK:
        raise RtError(rc, "rt_output_name")
    return buf.value.decode()


def write_bmp24(path: str, rgb: np.ndarray) -> None:
    lib = load_library()
    rgb = np.ascontiguousarray(rgb, np.uint8)
    h, w, _ = rgb.shape
    rc = lib.rt_write_bmp24(os.fsencode(path), _ptr(rgb), w, h)
    if rc != RT_OK:
        raise RtError(rc, f"rt_write_bmp24({path})")


def validate_scene(scene: "SceneArrays") -> int:
    """The host-side descriptor checks of rt_load_scene (no GPU needed) -> status code."""
    d, keep = scene.desc()
    rc = int(load_library().rt_validate_scene(C.byref(d)))
    del keep
    return rc


def render_multi(tracers, want_stats: bool = True):
    """One frame over several Tracer contexts of this process (rt_render_multi) -> (rgb8 [H,W,3], RtStats)."""
    lib = load_library()
    s = tracers[0].scene
    rgb = np.zeros((s.height, s.width, 3), np.uint8)
    arr = (C.c_void_p * len(tracers))(*[t._ctx for t in tracers])
    st = RtStats()
    rc = lib.rt_render_multi(arr, len(tracers), _ptr(rgb), C.byref(st))
    if rc != RT_OK:
        raise RtError(rc, "rt_render_multi")
    return rgb, st


def jitter_hash(seed: int, counter: int) -> int:
    return int(load_library().rt_jitter_hash(seed, counter))


def shard_tiles(width: int, height: int, rank: int, world: int) -> int:
    return int(load_library().rt_shard_tiles(width, height, rank, world))


def shard_bytes(width: int, height: int, world: int) -> int:
    return int(load_library().rt_shard_bytes(width, height, world))


class Tracer:
    """One rt_ctx: the drop-in for `RayTrace` on one GPU."""

    def __init__(self, device: int = 0, accel: int = RT_ACCEL_AUTO, batch_samples: int = 0, pool_rays: int = 0,
                 lbvh_threshold: int = 0, stream: int | None = None):
        self._lib = load_library()
        self._ctx = C.c_void_p()
        opt = RtOptions(device, accel, lbvh_threshold, batch_samples, pool_rays, stream)
        self.stream = stream or None      # the caller's stream the context runs on (None: a stream of its own)
        rc = self._lib.rt_create(C.byref(self._ctx), C.byref(opt))
        if rc != RT_OK:
            raise RtError(rc, "rt_create")
        self.scene: SceneArrays | None = None

    def close(self) -> None:
        if getattr(self, "_ctx", None):
            self._lib.rt_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def load_scene(self, scene: SceneArrays) -> None:
        d, keep = scene.desc()
        rc = self._lib.rt_load_scene(self._ctx, C.byref(d))
        del keep
        if rc != RT_OK:
            raise RtError(rc, "rt_load_scene")
        self.scene = scene

    def render(self, want_float: bool = False):
        """-> (rgb8 [H,W,3], rgbf [H,W,3] or None, RtStats); host arrays."""
        s = self.scene
        h, w = (s.height, s.width) if s is not None else (1, 1)   # no scene: the library reports it
        rgb = np.zeros((h, w, 3), np.uint8)
        rgbf = np.zeros((h, w, 3), np.float32) if want_float else None
        st = RtStats()
        rc = self._lib.rt_render(self._ctx, _ptr(rgb), _ptr(rgbf), C.byref(st))
        if rc != RT_OK:
            raise RtError(rc, "rt_render")
        return rgb, rgbf, st

    def render_device(self, d_rgb: int) -> RtStats:
        st = RtStats()
        rc = self._lib.rt_render_device(self._ctx, d_rgb, C.byref(st))
        if rc != RT_OK:
            raise RtError(rc, "rt_render_device")
        return st

    def render_shard(self, rank: int, world: int, d_shard: int) -> RtStats:
        st = RtStats()
        rc = self._lib.rt_render_shard(self._ctx, rank, world, d_shard, C.byref(st))
        if rc != RT_OK:
            raise RtError(rc, "rt_render_shard")
        return st

    def assemble_shards(self, d_all: int, world: int, d_rgb: int) -> None:
        rc = self._lib.rt_assemble_shards(self._ctx, d_all, world, d_rgb)
        if rc != RT_OK:
            raise RtError(rc, "rt_assemble_shards")

    def assemble_shards_noalloc(self, d_all: int, world: int, d_rgb: int) -> None:
        """rt_assemble_shards without any allocation (a slice of a fixed per-GPU table; world * tile rows <= 2048, at most 8 contexts per GPU hold a slice)."""
        rc = self._lib.rt_assemble_shards_noalloc(self._ctx, d_all, world, d_rgb)
        if rc != RT_OK:
            raise RtError(rc, "rt_assemble_shards_noalloc")

    def set_stage_times(self, on):
        """Fill RtStats.ms_extend / ms_shadow / ms_shade / ms_other from the next render on
        (2: only ms_extend, 3: only ms_shadow, with a third of the events)."""
        rc = self._lib.rt_set_stage_times(self._ctx, int(on) if on in (2, 3) else (1 if on else 0))
        if rc != RT_OK:
            raise RtError(rc, "rt_set_stage_times")

    def set_calibration(self, on: bool):
        """False: no timing of hierarchies and walks at the first render after load_scene (one-frame callers)."""
        rc = self._lib.rt_set_calibration(self._ctx, 1 if on else 0)
        if rc != RT_OK:
            raise RtError(rc, "rt_set_calibration")

    def set_graphs(self, on=None) -> int:
        """Replay of thin batches as HIP graphs: True / False turns it on / off, None leaves it; returns how many
        batches of this context ran as a graph launch so far."""
        n = C.c_uint64(0)
        rc = self._lib.rt_set_graphs(self._ctx, -1 if on is None else (1 if on else 0), C.byref(n))
        if rc != RT_OK:
            raise RtError(rc, "rt_set_graphs")
        return int(n.value)

    def tree_stats(self) -> RtTreeStats:
        """The hierarchies built over the loaded scene, their surface-area cost and which one renders."""
        ts = RtTreeStats()
        rc = self._lib.rt_get_tree_stats(self._ctx, C.byref(ts))
        if rc != RT_OK:
            raise RtError(rc, "rt_get_tree_stats")
        return ts

    def set_progress_callback(self, fn):
        """fn(pixels_done, pixels_total) before the first and after every batch of a render; None removes it."""
        self._progress = PROGRESS_FN(lambda user, done, total: fn(int(done), int(total))) if fn else C.cast(None, PROGRESS_FN)
        rc = self._lib.rt_set_progress_callback(self._ctx, self._progress, None)
        if rc != RT_OK:
            raise RtError(rc, "rt_set_progress_callback")

    def set_work_counters(self, on: bool):
        """Count node / primitive fetches and tests into RtStats from the next render on (slower)."""
        rc = self._lib.rt_set_work_counters(self._ctx, 1 if on else 0)
        if rc != RT_OK:
            raise RtError(rc, "rt_set_work_counters")

    def query_rays(self, o, d, dmax=None, any_hit: bool = False, accel: int = RT_ACCEL_BRUTE):
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        n = len(o)
        dm = np.ascontiguousarray(dmax, np.float32).reshape(-1) if dmax is not None else None
        prim = np.full(n, -2, np.int32)
        d2 = np.zeros(n, np.float32)
        rc = self._lib.rt_query_rays(self._ctx, n, _ptr(o), _ptr(d), _ptr(dm), int(any_hit), accel, _ptr(prim), _ptr(d2))
        if rc != RT_OK:
            raise RtError(rc, "rt_query_rays")
        return prim, d2

    def primary_rays(self, x, y, j):
        x = np.ascontiguousarray(x, np.int32).reshape(-1)
        y = np.ascontiguousarray(y, np.int32).reshape(-1)
        j = np.ascontiguousarray(j, np.int32).reshape(-1)
        n = len(x)
        o = np.zeros((n, 3), np.float32)
        d = np.zeros((n, 3), np.float32)
        rc = self._lib.rt_primary_rays(self._ctx, n, _ptr(x), _ptr(y), _ptr(j), _ptr(o), _ptr(d))
        if rc != RT_OK:
            raise RtError(rc, "rt_primary_rays")
        return o, d
