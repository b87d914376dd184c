#!/usr/bin/env python3
"""Differential fuzzing of the scene parser (rt_parse_scene_file) against the reference's own Scene::parseScene
(oracle/_ref, built from /root/reference/src/parse.cpp), on generated scene TEXT.  CPU only.

  python tools/fuzz_parser.py [--seeds A B]

Every seed writes a scene file of random directives in random order -- all number spellings the stream extraction
accepts (1e3, .5, 5., +2, -0), tabs, CRLF, comments, blank lines, unknown words, directives with too many or too
few arguments (the next token is then consumed, as `>>` does), triangle indices in and out of range -- and
compares what both parsers return: success or failure, and on success every array bit for bit."""
import argparse
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import raytracer_amd as ra  # noqa: E402
from _oracle import Reference, have_reference  # noqa: E402

ARGS = {"camera": 10, "film_resolution": 2, "sphere": 4, "background": 3, "material": 14, "directional_light": 6,
        "point_light": 6, "spot_light": 11, "ambient_light": 3, "max_depth": 1, "vertex": 3, "normal": 3,
        "triangle": 3, "normal_triangle": 6, "bvh_threshold": 1, "bvh_depth": 1, "sample_rate": 1}
INTS = {"film_resolution", "max_depth", "triangle", "normal_triangle", "bvh_threshold", "bvh_depth", "sample_rate"}


def number(rng, integer, hostile):
    """hostile: the LAST argument of a directive may be a token the extraction rejects or reads only in part.  (A
    rejected argument that is not the last leaves the ones after it uninitialised in the reference: nothing to
    compare with.)"""
    if integer:
        v = int(rng.integers(0, 40)) if rng.random() < 0.9 else int(rng.integers(-3, 2000))
        s = str(v)
        if hostile and rng.random() < 0.03:
            s = rng.choice(["+" + s, s + ".5", s + "e1", "0" + s])
        return s
    v = float(rng.normal()) * float(10.0 ** rng.integers(-3, 4))
    form = rng.integers(0, 8)
    if form == 0:
        s = repr(v)
    elif form == 1:
        s = f"{v:.3f}"
    elif form == 2:
        s = f"{v:.2e}"
    elif form == 3:
        s = str(int(v))
    elif form == 4:
        s = f"{abs(v) % 1:.4f}"[1:]          # .1234
    elif form == 5:
        s = f"{int(abs(v))}."                 # 12.
    elif form == 6:
        s = "+" + f"{abs(v):.2f}"
    else:
        s = f"{v:.1E}"
    if hostile and rng.random() < 0.02:
        s = rng.choice(["1e39", "-1e-50", "1.5.2", "3,4", "--1", "1e", "e5", "0x10", "nan", "inf", "1f"])
    return s


def scene_text(seed):
    rng = np.random.default_rng(seed)
    hostile = rng.random() < 0.5
    lines = []
    nv = nn = 0
    for _ in range(int(rng.integers(1, 60))):
        r = rng.random()
        if r < 0.06:
            lines.append("#" + rng.choice(["", " a comment", "sphere 1 2 3 4", " " + "x" * int(rng.integers(1, 300))]))
            continue
        if r < 0.10:
            lines.append("")
            continue
        if r < 0.13:
            lines.append(rng.choice(["output_image out.bmp", "output_image a.b.c", "ray_type perspective", "ray_type orthographic",
                                     "ray_type fisheye", "output_image", "ray_type"]))
            continue
        if hostile and r < 0.16:
            lines.append(rng.choice(["bogus 1 2 3", "Sphere 0 0 5 1", "sphere# 1 2 3 4", "12 13", "materials 1 2", "cam 0 0 0"]))
            continue
        cmd = rng.choice(list(ARGS))
        n = ARGS[cmd]
        if hostile and rng.random() < 0.05:
            n = max(0, n + int(rng.choice([-1, 1, 2])))         # one argument short (the next token is consumed), or too many
        if cmd in ("triangle", "normal_triangle") and not (hostile and rng.random() < 0.1):
            if nv == 0 or (cmd == "normal_triangle" and nn == 0):
                cmd = "vertex"; n = 3
        vals = []
        for k in range(n):
            if cmd == "triangle" or (cmd == "normal_triangle" and k < 3):
                vals.append(str(int(rng.integers(0, max(nv, 1))) if rng.random() < 0.97 else int(rng.integers(-2, nv + 3))))
            elif cmd == "normal_triangle":
                vals.append(str(int(rng.integers(0, max(nn, 1))) if rng.random() < 0.97 else int(rng.integers(-2, nn + 3))))
            elif cmd == "film_resolution":
                vals.append(str(int(rng.integers(1, 300))))
            else:
                vals.append(number(rng, cmd in INTS, hostile and k == ARGS[cmd] - 1))
        if cmd == "vertex" and n == 3:
            nv += 1
        if cmd == "normal" and n == 3:
            nn += 1
        sep = rng.choice([" ", "  ", "\t", " \t "])
        lines.append(rng.choice(["", " ", "\t"]) + cmd + (sep if vals else "") + sep.join(vals) + rng.choice(["", " ", "  # trailing"]))
    # a directive cut short by the END OF THE FILE leaves its missing arguments uninitialised in the reference (the
    # extraction is never attempted): nothing to compare with, so the file ends in a comment
    lines.append("# end")
    eol = "\r\n" if rng.random() < 0.2 else "\n"
    text = eol.join(lines)
    if rng.random() < 0.8:
        text += eol
    return text


FIELDS = ("cam_pos", "cam_dir", "cam_up", "background", "ambient", "spheres", "tri_verts", "tri_normals", "tri_flags", "lights")


def compare(a, b):
    """-> list of differences between two SceneArrays"""
    out = []
    for n in ("width", "height", "spp", "max_depth", "proj"):
        if getattr(a, n) != getattr(b, n):
            out.append(f"{n}: reference {getattr(a, n)} product {getattr(b, n)}")
    for n in FIELDS:
        x, y = getattr(a, n), getattr(b, n)
        if x.shape != y.shape or not np.array_equal(x.view(np.uint32), y.view(np.uint32)):
            out.append(f"{n}: shapes {x.shape} {y.shape}" + ("" if x.shape != y.shape else f", first difference at {np.argwhere(x.view(np.uint32) != y.view(np.uint32))[0].tolist()}"))
    if np.float32(a.cam_half_angle).view(np.uint32) != np.float32(b.cam_half_angle).view(np.uint32):
        out.append(f"cam_half_angle {a.cam_half_angle} {b.cam_half_angle}")
    for what, ia, ib in (("sphere materials", a.sphere_mat, b.sphere_mat), ("triangle materials", a.tri_mat, b.tri_mat)):
        if len(ia) == len(ib) and len(ia) and not np.array_equal(a.materials[ia].view(np.uint32), b.materials[ib].view(np.uint32)):
            out.append(what)
    if a.output_name != b.output_name:
        out.append(f"output name {a.output_name!r} {b.output_name!r}")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, nargs=2, default=[0, 2000])
    ap.add_argument("--keep", help="directory to keep the scene files of mismatches in")
    args = ap.parse_args()
    if not have_reference():
        raise SystemExit("needs oracle/_ref (make -C oracle ref, where /root/reference is mounted)")
    ref = Reference()
    bad = 0
    with tempfile.TemporaryDirectory() as td:
        for seed in range(*args.seeds):
            path = os.path.join(td, "fuzz.scn")
            text = scene_text(seed)
            with open(path, "w", newline="") as f:
                f.write(text)
            rc = ref.load(path)
            try:
                b = ra.parse_scene_file(path, ref_compat=True)
                ok = True
            except ra.RtError as e:
                ok, err = False, str(e)
            notes = []
            if (rc > 0) != ok:
                notes.append(f"reference returns {rc}, product {'parses' if ok else 'fails: ' + err}")
            elif ok:
                notes = compare(ref.scene_arrays(), b)
            if notes:
                bad += 1
                print(f"MISMATCH seed {seed}: " + "; ".join(notes[:4]), flush=True)
                if args.keep:
                    os.makedirs(args.keep, exist_ok=True)
                    with open(os.path.join(args.keep, f"fuzz_{seed}.scn"), "w", newline="") as f:
                        f.write(text)
    print(f"{bad} of {args.seeds[1] - args.seeds[0]} scene files parsed differently")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
