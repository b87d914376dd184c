"""Bindings of the CHECKERS: oracle/liboracle.so (CPU restatement) and, when it was
built, oracle/_ref/libref_oracle.so (the unmodified reference).  Test
infrastructure only -- the product package never imports this module."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref_oracle.so")
REF_O0_SO = os.path.join(ROOT, "oracle", "_ref", "libref_oracle_O0.so")   # same sources, the flags the reference ships with (no -O)
REFERENCE_TREE = "/root/reference"


class OrcCounts(C.Structure):
    _fields_ = [("primary", C.c_uint64), ("secondary", C.c_uint64), ("shadow", C.c_uint64)]

    @property
    def total(self):
        return self.primary + self.secondary + self.shadow


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Oracle:
    """CPU restatement (oracle/rt_oracle.c).  Takes a raytracer_amd.SceneArrays: the
    orc_scene struct has the field layout of rt_scene_desc."""

    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            raise FileNotFoundError(f"{ORACLE_SO} missing: run `make -C oracle`")
        L = C.CDLL(ORACLE_SO)
        L.orc_render.restype = C.c_double
        L.orc_jitter_hash.restype = C.c_uint32
        L.orc_jitter_hash.argtypes = [C.c_uint32, C.c_uint64]
        L.orc_intersect.restype = C.c_int
        L.orc_intersect.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p]
        L.orc_get_ray.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_trace_ray.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_render.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        self.L = L

    def render(self, scene, roi=None, threads=8, want_float=True):
        d, keep = scene.desc()
        x0, y0, x1, y1 = roi if roi else (0, 0, scene.width, scene.height)
        rgb = np.zeros((y1 - y0, x1 - x0, 3), np.uint8)
        rgbf = np.zeros((y1 - y0, x1 - x0, 3), np.float32) if want_float else None
        cnt = OrcCounts()
        secs = self.L.orc_render(C.byref(d), x0, y0, x1, y1, threads, _p(rgb), _p(rgbf), C.byref(cnt))
        del keep
        return rgb, rgbf, cnt, secs

    def get_ray(self, scene, x, y, j):
        d, keep = scene.desc()
        o = np.zeros(3, np.float32); dd = np.zeros(3, np.float32)
        self.L.orc_get_ray(C.byref(d), x, y, j, _p(o), _p(dd))
        return o, dd

    def intersect(self, scene, o, dr, dmin=0.001, dmax=float("inf")):
        d, keep = scene.desc()
        o = np.ascontiguousarray(o, np.float32); dr = np.ascontiguousarray(dr, np.float32)
        out = np.zeros(8, np.float64)
        kind = self.L.orc_intersect(C.byref(d), _p(o), _p(dr), dmin, dmax, _p(out))
        return kind, out

    def intersect_many(self, scene, o, dr, dmax=None):
        """-> prim ids in the product's numbering (-1 miss, sphere i, n_spheres + triangle i), dist2"""
        d, keep = scene.desc()
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3); dr = np.ascontiguousarray(dr, np.float32).reshape(-1, 3)
        n = len(o)
        prim = np.full(n, -1, np.int32); d2 = np.full(n, np.inf, np.float32)
        out = np.zeros(8, np.float64)
        for i in range(n):
            k = self.L.orc_intersect(C.byref(d), _p(o[i]), _p(dr[i]), 0.001, float(dmax[i]) if dmax is not None else float("inf"), _p(out))
            if k:
                prim[i] = int(out[7]) + (scene.spheres.shape[0] if k == 2 else 0)
                d2[i] = out[3]
        return prim, d2

    def trace_ray(self, scene, o, dr, depth=0):
        d, keep = scene.desc()
        o = np.ascontiguousarray(o, np.float32); dr = np.ascontiguousarray(dr, np.float32)
        rgb = np.zeros(3, np.float32)
        self.L.orc_trace_ray(C.byref(d), _p(o), _p(dr), depth, _p(rgb), None)
        return rgb

    def quantise(self, rgb):
        rgb = np.ascontiguousarray(rgb, np.float32)
        out = np.zeros(3, np.uint8)
        self.L.orc_quantise(_p(rgb), _p(out))
        return out

    def jitter_hash(self, seed, counter):
        return int(self.L.orc_jitter_hash(seed, counter))

    def spot_branch_counts(self, reset=True):
        """(alpha < angle1, alpha > angle2, interpolated): spot fall-off branches taken since the last reset"""
        out = (C.c_ulonglong * 3)()
        self.L.orc_spot_branch_counts(out, 1 if reset else 0)
        return tuple(int(v) for v in out)


def have_reference():
    return os.path.exists(REF_SO)


class Reference:
    """The unmodified reference, through oracle/ref_driver.cpp.  Holds ONE global scene."""

    def __init__(self, path=REF_SO):
        L = C.CDLL(path)
        L.ref_render.restype = C.c_double
        L.ref_render.argtypes = [C.c_uint, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ref_output_name.restype = C.c_char_p
        L.ref_intersect.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p]
        self.L = L
        self.info = None

    def load(self, path, w=-1, h=-1, spp=-1, depth=-1):
        rc = self.L.ref_load_scene(os.fsencode(path))
        if rc < 0:
            return rc
        self.L.ref_override(w, h, spp, depth)
        info = (C.c_int * 8)()
        self.L.ref_scene_info(info)
        self.info = list(info)
        return rc

    def output_name(self):
        return self.L.ref_output_name().decode(errors="replace")

    def scene_arrays(self, seed=1):
        """The reference's parsed scene as raytracer_amd.SceneArrays (materials de-duplicated)."""
        from raytracer_amd import SceneArrays
        W, H, SPP, D, ns, nt, nl, proj = self.info
        sph = np.zeros((ns, 4), np.float32); sm = np.zeros((ns, 14), np.float32)
        tv = np.zeros((nt, 9), np.float32); tn = np.zeros((nt, 9), np.float32)
        tf = np.zeros(nt, np.int32); tm = np.zeros((nt, 14), np.float32)
        li = np.zeros((nl, 12), np.float32); misc = np.zeros(16, np.float32)
        self.L.ref_dump_scene(_p(sph), _p(sm), _p(tv), _p(tn), _p(tf), _p(tm), _p(li), _p(misc))
        mats = np.concatenate([sm, tm])
        if len(mats):
            um, inv = np.unique(mats.view(np.uint32), axis=0, return_inverse=True)
            um = um.view(np.float32); inv = inv.reshape(-1).astype(np.int32)
        else:
            um = np.zeros((1, 14), np.float32); inv = np.zeros(0, np.int32)
        return SceneArrays(width=W, height=H, spp=SPP, max_depth=D, proj=proj, seed=seed,
                           cam_pos=misc[0:3], cam_dir=misc[3:6], cam_up=misc[6:9], cam_half_angle=float(misc[9]),
                           background=misc[10:13], ambient=misc[13:16], spheres=sph, sphere_mat=inv[:ns],
                           tri_verts=tv, tri_normals=tn, tri_flags=tf, tri_mat=inv[ns:], materials=um, lights=li,
                           output_name=self.output_name()).normalise()

    def render(self, seed=1, roi=None, threads=8):
        W, H = self.info[0], self.info[1]
        x0, y0, x1, y1 = roi if roi else (0, 0, W, H)
        rgb = np.zeros((y1 - y0, x1 - x0, 3), np.uint8); rgbf = np.zeros((y1 - y0, x1 - x0, 3), np.float32)
        secs = self.L.ref_render(seed, x0, y0, x1, y1, threads, _p(rgb), _p(rgbf))
        return rgb, rgbf, secs

    def get_ray(self, x, y, j, seed=1):
        o = np.zeros(3, np.float32); d = np.zeros(3, np.float32)
        self.L.ref_get_ray(x, y, j, seed, _p(o), _p(d))
        return o, d

    def trace_ray(self, o, d, depth=0):
        o = np.ascontiguousarray(o, np.float32); d = np.ascontiguousarray(d, np.float32)
        rgb = np.zeros(3, np.float32)
        self.L.ref_trace_ray(_p(o), _p(d), depth, _p(rgb))
        return rgb

    def intersect(self, o, d, dmin=0.001, dmax=float("inf")):
        o = np.ascontiguousarray(o, np.float32); d = np.ascontiguousarray(d, np.float32)
        out = np.zeros(8, np.float64)
        hit = self.L.ref_intersect(_p(o), _p(d), dmin, dmax, _p(out))
        return hit, out

    def quantise(self, rgb):
        rgb = np.ascontiguousarray(rgb, np.float32)
        out = np.zeros(3, np.uint8)
        self.L.ref_quantise(_p(rgb), _p(out))
        return out

    def camera_frame(self):
        out = np.zeros(7, np.float64)
        self.L.ref_camera_frame(_p(out))
        return out

    def print_scene(self):
        """The reference's printScene() of the loaded scene, captured from stdout -> str"""
        import sys
        import tempfile
        sys.stdout.flush()
        with tempfile.TemporaryFile() as tmp:
            saved = os.dup(1)
            try:
                os.dup2(tmp.fileno(), 1)
                self.L.ref_print_scene()
            finally:
                os.dup2(saved, 1)
                os.close(saved)
            tmp.seek(0)
            return tmp.read().decode(errors="replace")

    def write_bmp(self, path, rgb):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        self.L.ref_write_bmp(os.fsencode(path), _p(rgb), rgb.shape[1], rgb.shape[0])
