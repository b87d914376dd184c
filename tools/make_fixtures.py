#!/usr/bin/env python3
"""Generate tests/golden/ from the UNMODIFIED reference (oracle/_ref/libref_oracle.so).

Runs only where /root/reference exists (the build container).  What it writes is
data, never reference source:
  tests/golden/scenes/<name>.npz   scenes as the reference's own parser produced them
                                   (flat arrays), incl. the two derived benchmark scenes
  tests/golden/render_<case>.npz   small reference renders: float colour, RGB8, and the
                                   per-class ray counts of the CPU restatement on the same
                                   input (the reference cannot count; the two images are
                                   bit-identical, which pins the counts)
  tests/golden/kat.npz             known-answer vectors: primary rays, intersections,
                                   ray-tree colours, quantiser cases, jitter values, BMP bytes
Usage: python tools/make_fixtures.py   (after `make -C oracle`)
"""
import hashlib
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from _oracle import Oracle, Reference, REFERENCE_TREE  # noqa: E402
import raytracer_amd as ra  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
SCENES = {
    "spheres1": "SphereExamples/spheres1.scn",
    "spheres2": "SphereExamples/spheres2.scn",
    "spot_sphere": "SphereExamples/spot_sphere.scn",
    "ambient_sphere": "SphereExamples/ambient_sphere.scn",
    "triangle": "TriangleExamples/triangle.scn",
    "outdoor": "TriangleExamples/outdoor.scn",
    "test_reasonable": "TriangleExamples/test_reasonable.scn",
    "bvhtest": "BVHTest/test.scn",
    "dragon": "ComplexExamples/dragon.scn",
    "plant_h": "ComplexExamples/plant-h.scn",
    "complex_test": "ComplexExamples/test.scn",
}
DEG = 0.01745329251


def f9(x):
    return "%.9g" % float(x)


def scene_to_scn(s: ra.SceneArrays, per_vertex_normals=None) -> str:
    """Write a SceneArrays back out in the reference's text format (floats as %.9g,
    which round-trips float32).  per_vertex_normals: optional [nt,3,3] -> normal_triangle."""
    out = []
    ha = np.float32(s.cam_half_angle)
    cand = np.float32(float(ha) / DEG)
    best = None
    for k in range(-8, 9):
        c = cand
        for _ in range(abs(k)):
            c = np.nextafter(c, np.float32(np.inf if k > 0 else -np.inf), dtype=np.float32)
        if np.float32(float(c) * DEG) == ha:
            best = c
            break
    assert best is not None, "camera half angle does not round-trip"
    out.append("camera " + " ".join(f9(v) for v in [*s.cam_pos, *s.cam_dir, *s.cam_up, best]))
    out.append(f"film_resolution {s.width} {s.height}")
    out.append(f"output_image {s.output_name}")
    out.append("background " + " ".join(f9(v) for v in s.background))
    out.append("ambient_light " + " ".join(f9(v) for v in s.ambient))
    out.append(f"max_depth {s.max_depth}")
    out.append(f"sample_rate {s.spp}")
    out.append("ray_type " + ("perspective" if s.proj == ra.RT_PROJ_PERSP else "orthographic"))
    for L in s.lights:
        t = int(L[0])
        if t == 1:
            out.append("directional_light " + " ".join(f9(v) for v in [*L[1:4], *L[7:10]]))
        elif t == 2:
            out.append("point_light " + " ".join(f9(v) for v in [*L[1:4], *L[4:7]]))
        else:
            out.append("spot_light " + " ".join(f9(v) for v in [*L[1:4], *L[4:7], *L[7:10], L[10], L[11]]))
    cur = None

    def mat(i):
        nonlocal cur
        if i != cur:
            out.append("material " + " ".join(f9(v) for v in s.materials[i]))
            cur = i
    for i, sp in enumerate(s.spheres):
        mat(int(s.sphere_mat[i]))
        out.append("sphere " + " ".join(f9(v) for v in sp))
    nv = nn = 0          # vertices and normals are pools of their own, indexed in order of appearance
    for i, tv in enumerate(s.tri_verts):
        mat(int(s.tri_mat[i]))
        for k in range(3):
            out.append("vertex " + " ".join(f9(v) for v in tv[3 * k:3 * k + 3]))
        if per_vertex_normals is not None:
            for k in range(3):
                nk = per_vertex_normals[i, k]
                # a degenerate normal is written as 0 0 0: the parser's norm() turns it
                # into NaN, as it does for the flat normal of a zero-area triangle
                out.append("normal " + (" ".join(f9(v) for v in nk) if np.all(np.isfinite(nk)) else "0 0 0"))
            out.append(f"normal_triangle {nv} {nv + 1} {nv + 2} {nn} {nn + 1} {nn + 2}")
            nn += 3
        elif s.tri_flags[i]:
            for k in range(3):
                nk = s.tri_normals[i, 3 * k:3 * k + 3]
                out.append("normal " + (" ".join(f9(v) for v in nk) if np.all(np.isfinite(nk)) else "0 0 0"))
            out.append(f"normal_triangle {nv} {nv + 1} {nv + 2} {nn} {nn + 1} {nn + 2}")
            nn += 3
        else:
            out.append(f"triangle {nv} {nv + 1} {nv + 2}")
        nv += 3
    return "\n".join(out) + "\n"


def smooth_normals(s: ra.SceneArrays) -> np.ndarray:
    """Area-weighted vertex normals merged on identical positions -> [nt,3,3] float32."""
    v = s.tri_verts.reshape(-1, 3, 3).astype(np.float64)
    fn = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0])          # length = 2 * area
    # orient every face like the normal the flat shading uses, so the sum is coherent
    flat = s.tri_normals.reshape(-1, 3, 3)[:, 0].astype(np.float64)
    fn *= np.sign(np.einsum("ij,ij->i", fn, flat))[:, None]
    keys = s.tri_verts.reshape(-1, 3).view(np.uint32)
    _, inv = np.unique(keys, axis=0, return_inverse=True)
    inv = inv.reshape(-1)
    acc = np.zeros((inv.max() + 1, 3))
    np.add.at(acc, inv, np.repeat(fn, 3, axis=0))
    n = acc[inv].reshape(-1, 3, 3)
    ln = np.linalg.norm(n, axis=2, keepdims=True)
    n = np.where(ln > 0, n / np.maximum(ln, 1e-300), flat[:, None, :])
    return n.astype(np.float32)


# Derived scene for the spot-light fall-off (src/raytrace.cpp:549-566).  The reference compares
# acos(max(0, N.I)) -- radians, at most pi/2 -- with the file's raw angle values, so every shipped
# spot light (45/90, 5/20) only ever takes the "acts like a point light" branch.  Angles in the
# radian range make all three branches run: light 1 (0.35 / 0.9) mostly interpolates, light 2
# (0.1 / 0.2) mostly contributes nothing, light 3 (45 / 90) is point-like, light 4 (0.6 / 1.3) mixes.
SPOT_FALLOFF_SCN = """# spot lights whose angles lie in the range of acos()
output_image spot_falloff.bmp.tail
camera 0 4 -40 0 -0.05 1 0 1 0 20
film_resolution 160 120
background .1 .2 .4
ambient_light .05 .05 .05
max_depth 4
material .2 .2 .2 .7 .6 .5 .3 .3 .3 20 0 0 0 1
sphere -6 -2 8 4
material 0 0 0 .3 .5 .7 .6 .6 .6 30 .8 .8 .8 1.4
sphere 6 0 4 3.5
material .1 .1 .1 .8 .8 .8 0 0 0 5 0 0 0 1
vertex -60 -6 -30
vertex 60 -6 -30
vertex 60 -6 90
vertex -60 -6 90
triangle 0 1 2
triangle 0 2 3
vertex -30 -6 40
vertex 30 -6 40
vertex 0 30 45
normal 0 0.2 -1
normal 0.3 0 -1
normal -0.3 0.1 -1
material 0 0 0 .6 .7 .6 .2 .2 .2 8 0 0 0 1
normal_triangle 4 5 6 0 1 2
spot_light 900 900 800 0 25 -5 0 -1 0 0.35 0.9
spot_light 400 100 100 -20 10 -10 1 -1 1 0.1 0.2
spot_light 200 300 200 15 12 -15 0 0 1 45 90
spot_light 500 500 100 0 2 20 0 -1 0 0.6 1.3
point_light 50 50 80 10 20 0
"""


def spot_falloff_fixtures(R, O, tmpdir):
    """scene, two reference renders and getColor known-answer rows for the spot fall-off branches"""
    p = os.path.join(tmpdir, "spot_falloff.scn")
    open(p, "w").write(SPOT_FALLOFF_SCN)
    assert R.load(p, w=160, h=120) > 0
    scene = R.scene_arrays()
    scene.save(os.path.join(GOLD, "scenes", "spot_falloff.npz"))
    for case, w, h, spp in (("spot_falloff", 160, 120, 1), ("spot_falloff_spp4", 128, 96, 4)):
        s = scene.copy(width=w, height=h, spp=spp)
        assert R.load(p, w=w, h=h, spp=spp) > 0
        rgb, rgbf, secs = R.render(seed=1)
        O.spot_branch_counts(reset=True)
        orgb, orgbf, cnt, _ = O.render(s)
        br = O.spot_branch_counts()
        assert min(br) > 0.02 * sum(br), f"{case}: a fall-off branch is (nearly) unused: {br}"
        assert np.array_equal(rgbf.view(np.uint32), orgbf.view(np.uint32)), f"{case}: restatement differs from reference"
        np.savez_compressed(os.path.join(GOLD, f"render_{case}.npz"), scene=np.frombuffer(b"spot_falloff", np.uint8),
                            params=np.array([w, h, spp, s.max_depth, s.proj, 1], np.int64),
                            roi=np.array((0, 0, w, h), np.int64), rgb=rgb, rgbf=rgbf,
                            counts=np.array([cnt.primary, cnt.secondary, cnt.shadow], np.uint64))
        print("render", case, rgb.shape, "ref %.2fs" % secs, "rays", cnt.total, "spot branches (<a1, >a2, between)", br)
    # getColor rows: evaluateRayTree at depth == max_depth is exactly one getColor call whose
    # children return the background untraced (src/raytrace.cpp:132-135), and the full recursion.
    assert R.load(p, w=160, h=120) > 0
    rng = np.random.default_rng(17)
    n = 96
    xs = rng.integers(0, 160, n); ys = rng.integers(30, 120, n)
    o = np.zeros((n, 3), np.float32); d = np.zeros((n, 3), np.float32)
    leaf = np.zeros((n, 3), np.float32); full = np.zeros((n, 3), np.float32)
    for i in range(n):
        o[i], d[i] = R.get_ray(int(xs[i]), int(ys[i]), 0, seed=1)
        leaf[i] = R.trace_ray(o[i], d[i], scene.max_depth)
        full[i] = R.trace_ray(o[i], d[i], 0)
    np.savez_compressed(os.path.join(GOLD, "kat_spot.npz"), xy=np.stack([xs, ys], 1).astype(np.int32), o=o, d=d,
                        col_leaf=leaf, col_full=full)
    print("kat_spot", n, "rows")


def main():
    assert os.path.isdir(REFERENCE_TREE), "reference tree absent: fixtures can only be generated where it is mounted"
    os.makedirs(os.path.join(GOLD, "scenes"), exist_ok=True)
    R, O = Reference(), Oracle()
    tmpdir = tempfile.mkdtemp(prefix="rtfix")
    spot_falloff_fixtures(R, O, tmpdir)
    if "--only-spot" in sys.argv:      # leave the other (unchanged) fixture files as they are
        return
    scenes, sources = {}, {}
    for name, rel in SCENES.items():
        sources[name] = os.path.join(REFERENCE_TREE, rel)
        assert R.load(sources[name]) > 0
        scenes[name] = R.scene_arrays()

    # ---- derived benchmark scenes -------------------------------------------------
    # K2: outdoor + a spot light (no shipped scene combines triangles with a spot light)
    k2 = scenes["outdoor"].copy()
    k2.lights = np.concatenate([k2.lights, np.array([[3, 600, 600, 600, 0, 30, 10, 0, -1, 0, 30, 60]], np.float32)])
    k2.output_name = "outdoor_spot.bmp"
    p = os.path.join(tmpdir, "outdoor_spot.scn")
    open(p, "w").write(scene_to_scn(k2))
    assert R.load(p, w=k2.width, h=k2.height) > 0
    scenes["outdoor_spot"] = R.scene_arrays()
    sources["outdoor_spot"] = p
    # K4: plant-h re-expressed with normal_triangle + max_depth 8
    ph = scenes["plant_h"].copy(max_depth=8, output_name="plant_smooth.bmp")
    p = os.path.join(tmpdir, "plant_smooth.scn")
    open(p, "w").write(scene_to_scn(ph, per_vertex_normals=smooth_normals(ph)))
    assert R.load(p, w=ph.width, h=ph.height) > 0
    # (the parser re-normalises every `normal`, and that is not idempotent in float:
    # this file, not a re-export of the parsed arrays, is what the cases below render)
    scenes["plant_h_smooth"] = R.scene_arrays()
    sources["plant_h_smooth"] = p
    assert scenes["plant_h_smooth"].tri_flags.all() and scenes["plant_h_smooth"].max_depth == 8
    assert len(scenes["plant_h_smooth"].tri_verts) == len(scenes["plant_h"].tri_verts)

    for name, s in scenes.items():
        s.save(os.path.join(GOLD, "scenes", name + ".npz"))
        print("scene", name, s.width, s.height, len(s.spheres), len(s.tri_verts), len(s.lights))

    # ---- reference renders ----------------------------------------------------------
    cases = [  # case, scene, w, h, spp, depth, proj, roi
        ("spheres1", "spheres1", 160, 120, 1, -1, None, None),
        ("spheres1_spp4", "spheres1", 96, 72, 4, -1, None, None),
        ("spheres1_ortho", "spheres1", 96, 72, 1, -1, 0, None),
        ("spheres2_spp3", "spheres2", 128, 96, 3, -1, None, None),
        ("spot_sphere", "spot_sphere", 128, 96, 1, -1, None, None),
        ("ambient_sphere", "ambient_sphere", 64, 48, 1, -1, None, None),
        ("triangle", "triangle", 64, 48, 1, -1, None, None),
        ("outdoor", "outdoor", 160, 120, 1, -1, None, None),
        ("outdoor_spot_spp4", "outdoor_spot", 192, 108, 4, -1, None, None),
        ("test_reasonable", "test_reasonable", 128, 96, 1, -1, None, None),
        ("bvhtest", "bvhtest", 64, 48, 1, -1, None, None),
        ("dragon", "dragon", 96, 96, 1, -1, None, None),
        ("dragon_roi_spp2", "dragon", 640, 640, 2, -1, None, (300, 280, 348, 320)),
        ("plant_h", "plant_h", 80, 80, 1, -1, None, None),
        ("plant_h_smooth", "plant_h_smooth", 80, 80, 1, -1, None, None),
        ("complex_test", "complex_test", 96, 96, 1, -1, None, None),
        ("spheres1_depth0", "spheres1", 64, 48, 1, 0, None, None),
    ]
    for case, sname, w, h, spp, depth, proj, roi in cases:
        s = scenes[sname].copy(width=w, height=h, spp=spp)
        if depth >= 0:
            s.max_depth = depth
        if proj is not None:
            s.proj = proj
        # the reference parses the scene's own file; size / sample rate / depth / projection
        # are then set through its public setters
        assert R.load(sources[sname], w=w, h=h, spp=spp, depth=depth) > 0
        if proj is not None:
            R.L.ref_set_proj(1 if proj == ra.RT_PROJ_ORTHO else 0)
            R.info[7] = proj
        parsed = R.scene_arrays()
        for n in ra.SceneArrays._ARRAYS:
            if n in ("materials", "sphere_mat", "tri_mat"):
                continue
            assert np.array_equal(getattr(parsed, n).view(np.uint32), getattr(s, n).view(np.uint32)), (case, n)
        assert np.array_equal(parsed.materials[parsed.tri_mat].view(np.uint32), s.materials[s.tri_mat].view(np.uint32))
        assert np.array_equal(parsed.materials[parsed.sphere_mat].view(np.uint32), s.materials[s.sphere_mat].view(np.uint32))
        assert (parsed.width, parsed.height, parsed.spp, parsed.max_depth, parsed.proj) == (s.width, s.height, s.spp, s.max_depth, s.proj), case
        assert np.float32(parsed.cam_half_angle) == np.float32(s.cam_half_angle)
        rgb, rgbf, secs = R.render(seed=1, roi=roi)
        orgb, orgbf, cnt, osecs = O.render(s, roi=roi)
        assert np.array_equal(rgbf.view(np.uint32), orgbf.view(np.uint32)), f"{case}: restatement differs from reference"
        np.savez_compressed(os.path.join(GOLD, f"render_{case}.npz"), scene=np.frombuffer(sname.encode(), np.uint8),
                            params=np.array([w, h, spp, s.max_depth, s.proj, 1], np.int64),
                            roi=np.array(roi if roi else (0, 0, w, h), np.int64), rgb=rgb, rgbf=rgbf,
                            counts=np.array([cnt.primary, cnt.secondary, cnt.shadow], np.uint64))
        print("render", case, rgb.shape, "ref %.2fs" % secs, "rays", cnt.total)

    # ---- known-answer vectors ---------------------------------------------------------
    rng = np.random.default_rng(7)
    kat = {}
    for sname, spp in (("spheres1", 1), ("spheres1", 5), ("outdoor", 3), ("dragon", 1)):
        s = scenes[sname].copy(spp=spp)
        assert R.load(sources[sname], spp=spp) > 0
        n = 64
        xs = rng.integers(0, s.width, n); ys = rng.integers(0, s.height, n); js = rng.integers(0, spp, n)
        xs[:4] = [0, s.width - 1, 0, s.width // 2]; ys[:4] = [0, s.height - 1, s.height - 1, s.height // 2]
        o = np.zeros((n, 3), np.float32); d = np.zeros((n, 3), np.float32)
        for i in range(n):
            o[i], d[i] = R.get_ray(int(xs[i]), int(ys[i]), int(js[i]), seed=1)
        key = f"{sname}_spp{spp}"
        kat[f"ray_{key}_xyj"] = np.stack([xs, ys, js], 1).astype(np.int32)
        kat[f"ray_{key}_o"] = o
        kat[f"ray_{key}_d"] = d
        kat[f"cam_{key}"] = R.camera_frame()
        # intersections + colours along those rays (and along their first bounce direction)
        hits = np.zeros((n, 9), np.float64); cols = np.zeros((n, 3), np.float32)
        for i in range(n):
            hit, out = R.intersect(o[i], d[i])
            hits[i, 0] = hit; hits[i, 1:] = out
            cols[i] = R.trace_ray(o[i], d[i], 0)
        kat[f"hit_{key}"] = hits
        kat[f"col_{key}"] = cols
    qin = np.array([[0, .5, 1], [.25, .05, .999999], [1.0000001, 2, -1], [np.nan, np.inf, -np.inf],
                    [1e10, -1e10, 8421504.5], [0.00392156, 0.00392157, 0.0039216], [127 / 255, 128 / 255, 254.9999 / 255]], np.float32)
    kat["quant_in"] = qin
    kat["quant_out"] = np.stack([R.quantise(q) for q in qin])
    # jitter sequence (frozen definition, not a reference output)
    ctr = np.array([0, 1, 2, 3, 640 * 480 * 2 - 1, 2**32 - 1, 2**32, 2**33 + 12345, 2**40 + 7], np.uint64)
    kat["jitter_counter"] = ctr
    kat["jitter_seed1"] = np.array([O.jitter_hash(1, int(c)) for c in ctr], np.uint32)
    kat["jitter_seed99"] = np.array([O.jitter_hash(99, int(c)) for c in ctr], np.uint32)
    # BMP bytes from the reference's own Image::Write (widths that need 0..3 bytes of row padding)
    for w, h in ((5, 3), (6, 2), (7, 2), (8, 1)):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        p = os.path.join(tmpdir, f"b{w}.bmp")
        R.write_bmp(p, img)
        kat[f"bmp_{w}x{h}_rgb"] = img
        kat[f"bmp_{w}x{h}_file"] = np.frombuffer(open(p, "rb").read(), np.uint8)
    big = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
    p = os.path.join(tmpdir, "big.bmp")
    R.write_bmp(p, big)
    kat["bmp_640x480_seed"] = np.array([7], np.int64)
    kat["bmp_640x480_sha256"] = np.frombuffer(hashlib.sha256(open(p, "rb").read()).hexdigest().encode(), np.uint8)
    kat["bmp_640x480_rgb_sha256"] = np.frombuffer(hashlib.sha256(big.tobytes()).hexdigest().encode(), np.uint8)
    np.savez_compressed(os.path.join(GOLD, "kat_bmp_big.npz"), rgb=big)
    np.savez_compressed(os.path.join(GOLD, "kat.npz"), **kat)
    print("kat", len(kat), "arrays")


if __name__ == "__main__":
    main()
