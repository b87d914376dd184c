"""Synthetic large scene of the fifth benchmark configuration ("1M spheres + 10M triangles"), built directly
as arrays (a text .scn of that size would be ~0.7 GB and the reference parser is O(N^2)).

Deterministic for a given (seed, scale): scale = 1.0 is the full 10^6 spheres + 10^7
triangles; scale = 0.001 gives 10^3 spheres + 10^4 triangles that the CPU oracle can still
brute-force at thumbnail size.  Triangles come as coherent meshes: randomly placed, rotated
and radially perturbed "blobs" of 1 000 triangles each (a 20 x 25 latitude/longitude grid),
edge length about 0.1 .. 0.5; spheres are uniform in the cube with log-uniform radii.
Eight materials (2 mirrors, 2 glasses with ior 1.5, 4 diffuse), one directional and two
point lights, camera outside the cube looking at the origin, max_depth 5.
"""
from __future__ import annotations

import numpy as np

from .api import RT_PROJ_PERSP, SceneArrays

TRIS_PER_BLOB = 1000
_LAT, _LON = 20, 25          # 20 x 25 quads -> 1000 triangles


def _blob_template():
    """Unit-sphere lat/lon grid: vertex directions [(_LAT+1)*_LON, 3] and triangle indices [1000, 3]."""
    th = np.linspace(0.12 * np.pi, 0.88 * np.pi, _LAT + 1)            # open at the poles: no degenerate triangles
    ph = np.arange(_LON) * (2 * np.pi / _LON)
    t, p = np.meshgrid(th, ph, indexing="ij")
    dirs = np.stack([np.sin(t) * np.cos(p), np.cos(t), np.sin(t) * np.sin(p)], -1).reshape(-1, 3)
    idx = []
    for i in range(_LAT):
        for j in range(_LON):
            a, b = i * _LON + j, i * _LON + (j + 1) % _LON
            c, d = (i + 1) * _LON + j, (i + 1) * _LON + (j + 1) % _LON
            idx.append((a, c, b))
            idx.append((b, c, d))
    return dirs, np.array(idx, np.int64)


def _rotations(rng, n):
    """n random rotation matrices (from unit quaternions)."""
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    return np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                     2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                     2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1).reshape(n, 3, 3)


def synthetic_scene(scale: float = 1.0, seed: int = 0xC0FFEE, width: int = 7680, height: int = 4320,
                    spp: int = 64, extent: float = 100.0) -> SceneArrays:
    rng = np.random.default_rng(seed)
    n_sph = max(1, int(round(1_000_000 * scale)))
    n_blob = max(1, int(round(10_000 * scale)))
    # keep the density of the full scene: the cube shrinks with the cube root of the scale
    ext = extent * scale ** (1.0 / 3.0)

    # ---- spheres
    centres = rng.uniform(-ext, ext, (n_sph, 3))
    radii = np.exp(rng.uniform(np.log(0.05), np.log(0.5), n_sph))
    spheres = np.concatenate([centres, radii[:, None]], 1).astype(np.float32)
    sphere_mat = rng.integers(0, 8, n_sph).astype(np.int32)

    # ---- triangle blobs
    dirs, idx = _blob_template()
    nv = len(dirs)
    blob_r = rng.uniform(0.8, 2.0, n_blob)                               # edge ~ 0.1 .. 0.5
    blob_c = rng.uniform(-ext, ext, (n_blob, 3))
    rot = _rotations(rng, n_blob)
    tri_verts = np.empty((n_blob * TRIS_PER_BLOB, 9), np.float32)
    chunk = 512
    for b0 in range(0, n_blob, chunk):
        b1 = min(n_blob, b0 + chunk)
        nb = b1 - b0
        bump = 1.0 + 0.25 * rng.uniform(-1, 1, (nb, nv))                 # radial perturbation per vertex
        v = dirs[None] * (blob_r[b0:b1, None] * bump)[..., None]          # [nb, nv, 3]
        v = np.einsum("bij,bvj->bvi", rot[b0:b1], v) + blob_c[b0:b1, None, :]
        t = v[:, idx]                                                    # [nb, 1000, 3, 3]
        tri_verts[b0 * TRIS_PER_BLOB:b1 * TRIS_PER_BLOB] = t.reshape(nb * TRIS_PER_BLOB, 9).astype(np.float32)
    tri_mat = np.repeat(rng.integers(0, 8, n_blob).astype(np.int32), TRIS_PER_BLOB)

    # flat normals exactly as the reference parser derives them: norm(cross(v1 - v0, v2 - v0)) in float32
    v0, v1, v2 = tri_verts[:, 0:3], tri_verts[:, 3:6], tri_verts[:, 6:9]
    e1, e2 = (v1 - v0).astype(np.float32), (v2 - v0).astype(np.float32)
    cx = (e1[:, 1] * e2[:, 2]).astype(np.float32) - (e1[:, 2] * e2[:, 1]).astype(np.float32)
    cy = (e1[:, 2] * e2[:, 0]).astype(np.float32) - (e1[:, 0] * e2[:, 2]).astype(np.float32)
    cz = (e1[:, 0] * e2[:, 1]).astype(np.float32) - (e1[:, 1] * e2[:, 0]).astype(np.float32)
    ss = ((cx * cx).astype(np.float32) + (cy * cy).astype(np.float32)).astype(np.float32) + (cz * cz).astype(np.float32)
    mag = np.sqrt(ss.astype(np.float32)).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        n = np.stack([cx / mag, cy / mag, cz / mag], 1).astype(np.float32)
    tri_normals = np.tile(n, (1, 3)).astype(np.float32)

    materials = np.array([
        # amb3            dif3             spec3          ns   trans3         ior
        [.05, .05, .05,   .10, .10, .10,   .85, .85, .85, 64,  0, 0, 0,       1.0],   # mirror
        [.05, .04, .02,   .30, .25, .10,   .70, .60, .30, 32,  0, 0, 0,       1.0],   # gold mirror
        [.02, .02, .02,   .10, .10, .10,   .20, .20, .20, 64,  .80, .80, .80, 1.5],   # glass
        [.01, .03, .03,   .05, .15, .15,   .15, .15, .15, 48,  .60, .85, .85, 1.5],   # tinted glass
        [.10, .02, .02,   .80, .15, .15,   0, 0, 0,       8,   0, 0, 0,       1.0],   # red
        [.02, .10, .02,   .15, .75, .20,   0, 0, 0,       8,   0, 0, 0,       1.0],   # green
        [.02, .02, .10,   .20, .25, .85,   0, 0, 0,       8,   0, 0, 0,       1.0],   # blue
        [.08, .08, .08,   .70, .70, .70,   0, 0, 0,       8,   0, 0, 0,       1.0],   # grey
    ], np.float32)
    far = 2.6 * ext
    lights = np.array([
        [1, .9, .9, .85,   0, 0, 0,   -.4, -1, .3,   0, 0],                                         # directional
        [2, 4 * ext * ext, 3.6 * ext * ext, 3.2 * ext * ext,   1.5 * ext, 1.8 * ext, -1.2 * ext,   0, 1, 0,   0, 0],
        [2, 2.5 * ext * ext, 3 * ext * ext, 4 * ext * ext,   -1.6 * ext, 1.2 * ext, 1.4 * ext,   0, 1, 0,   0, 0],
    ], np.float32)
    cam_pos = np.array([0.35 * far, 0.3 * far, -far], np.float32)
    cam_dir = (-cam_pos / np.linalg.norm(cam_pos)).astype(np.float32)
    return SceneArrays(width=width, height=height, spp=spp, max_depth=5, proj=RT_PROJ_PERSP, seed=1,
                       cam_pos=cam_pos, cam_dir=cam_dir, cam_up=np.array([0, 1, 0], np.float32),
                       cam_half_angle=float(np.float32(20.0 * 0.01745329251)),
                       background=np.array([.02, .03, .06], np.float32), ambient=np.array([.3, .3, .3], np.float32),
                       spheres=spheres, sphere_mat=sphere_mat, tri_verts=tri_verts, tri_normals=tri_normals,
                       tri_flags=np.zeros(len(tri_verts), np.int32), tri_mat=tri_mat, materials=materials,
                       lights=lights, output_name="synthetic.bmp").normalise()
