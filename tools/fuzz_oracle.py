#!/usr/bin/env python3
"""Differential fuzzing of the CPU oracle (oracle/rt_oracle.c, the restatement every GPU parity test is checked
against) against the UNMODIFIED reference (oracle/_ref, compiled from /root/reference/src), on the seeded random
scenes of tests/_fuzz_scenes.py.  CPU only.

  python tools/fuzz_oracle.py [--seeds A B]

A scene is written out as scene text, parsed by the reference's own parser, and the scene AS THE REFERENCE HOLDS IT
is rendered by both: float frames must be bit-identical (NaN == NaN), RGB8 identical."""
import argparse
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

from _fuzz_scenes import random_scene  # noqa: E402
from _oracle import Oracle, Reference, have_reference  # noqa: E402
from conftest import same_float_image  # noqa: E402
from make_fixtures import scene_to_scn  # noqa: E402


def check(ref, orc, scene, path, threads=8):
    """-> None, or a description of the difference"""
    # a half angle the scene text can carry: the file holds degrees as a float, the parser multiplies by a double
    deg = np.float32(float(np.float32(scene.cam_half_angle)) / 0.01745329251)
    scene = scene.copy(cam_half_angle=float(np.float32(float(deg) * 0.01745329251)))
    with open(path, "w") as f:
        f.write(scene_to_scn(scene))
    if ref.load(path, w=scene.width, h=scene.height, spp=scene.spp, depth=scene.max_depth) <= 0:
        return "the reference does not parse the scene"
    held = ref.scene_arrays(seed=scene.seed)
    rrgb, rrgbf, _ = ref.render(seed=scene.seed, threads=threads)
    orgb, orgbf, cnt, _ = orc.render(held, threads=threads)
    same = same_float_image(rrgbf, orgbf).all(axis=2)
    if not same.all() or not np.array_equal(rrgb, orgb):
        y, x = (int(v) for v in np.argwhere(~same)[0]) if not same.all() else (0, 0)
        return (f"{int((~same).sum())} of {same.size} pixels differ in float, {int((rrgb != orgb).any(axis=2).sum())} in RGB8; "
                f"first ({x}, {y}): reference {rrgbf[y, x].tolist()} oracle {orgbf[y, x].tolist()}")
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, nargs=2, default=[0, 500])
    args = ap.parse_args()
    if not have_reference():
        raise SystemExit("needs oracle/_ref (make -C oracle ref, where /root/reference is mounted)")
    ref, orc = Reference(), Oracle()
    bad = 0
    with tempfile.TemporaryDirectory() as td:
        for seed in range(*args.seeds):
            scene = random_scene(seed)
            why = check(ref, orc, scene, os.path.join(td, "fuzz.scn"))
            if why:
                bad += 1
                print(f"MISMATCH seed {seed} ({scene.width}x{scene.height}x{scene.spp}, depth {scene.max_depth}, {len(scene.spheres)} sph "
                      f"{len(scene.tri_verts)} tri {len(scene.lights)} lights): {why}", flush=True)
    print(f"{bad} of {args.seeds[1] - args.seeds[0]} scenes rendered differently")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
