#!/usr/bin/env python3
"""Differential fuzzing of the HIP path against the CPU oracle on seeded random scenes.

  python tools/fuzz_parity.py [--seeds A B] [--trees] [--save]

For every seed: the oracle's frame, then the HIP frame by brute force and through the hierarchy with every walk
forced in turn (and, with --trees, on both hierarchies).  Prints one line per mismatch and a summary; exit
status 1 if anything differed.  Needs a GPU."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: F401,E402  (before the tracer library: one HIP runtime initialisation order)
import raytracer_amd as ra  # noqa: E402
from _fuzz_scenes import WALKS, random_scene  # noqa: E402
from _oracle import Oracle  # noqa: E402
from conftest import same_float_image  # noqa: E402

ENV_KEYS = ("RT_PACKET_GENS", "RT_PACKET_SHADOW_GENS", "RT_LANE_WIDE", "RT_TREE")


def render(scene, accel, env):
    for k in ENV_KEYS:
        os.environ.pop(k, None)
    os.environ.update(env)
    try:
        with ra.Tracer(device=0, accel=accel) as t:
            t.load_scene(scene)
            return t.render(want_float=True)
    finally:
        for k in ENV_KEYS:
            os.environ.pop(k, None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, nargs=2, default=[0, 40])
    ap.add_argument("--trees", action="store_true", help="force each hierarchy (Karras, PLOC) in turn as well")
    ap.add_argument("--save", action="store_true", help="write the frames of a mismatch to gpurun_out/fuzz_<seed>_<variant>.npz")
    args = ap.parse_args()
    orc = Oracle()
    bad = 0
    t0 = time.time()
    for seed in range(*args.seeds):
        scene = random_scene(seed)
        orgb, orgbf, cnt, _ = orc.render(scene, threads=16)
        variants = [("brute", ra.RT_ACCEL_BRUTE, {})]
        for walk, env in WALKS.items():
            for tree in (("karras", "ploc") if args.trees else (None,)):
                e = dict(env)
                if tree:
                    e["RT_TREE"] = tree
                variants.append((walk + ("/" + tree if tree else ""), ra.RT_ACCEL_LBVH, e))
        variants.append(("calibrated", ra.RT_ACCEL_LBVH, {}))
        notes = []
        for label, accel, env in variants:
            try:
                rgb, rgbf, st = render(scene, accel, env)
            except ra.RtError as e:
                if "single" in str(e).lower() or len(scene.spheres) + len(scene.tri_verts) < 2:
                    continue
                notes.append(f"{label}: error {e}")
                continue
            snapf, snap8 = rgbf.copy(), rgb.copy()
            same = same_float_image(snapf, orgbf).all(axis=2)
            counts = (st.rays_primary, st.rays_secondary, st.rays_shadow) == (cnt.primary, cnt.secondary, cnt.shadow)
            if not same.all() or not np.array_equal(snap8, orgb) or not counts:
                notes.append(f"{label}: {int((~same).sum())} of {same.size} pixels differ in float, "
                             f"{int((snap8 != orgb).any(axis=2).sum())} in RGB8, rays {st.rays_primary}/{st.rays_secondary}/{st.rays_shadow} "
                             f"vs {cnt.primary}/{cnt.secondary}/{cnt.shadow}")
                if args.save:
                    np.savez(os.path.join(ROOT, "gpurun_out", f"fuzz_{seed}_{label.replace('/', '_')}.npz"), rgbf=snapf, rgb=snap8, orgbf=orgbf, orgb=orgb)
                for y, x in list(zip(*np.nonzero(~same)))[:4]:
                    notes.append(f"    pixel ({x}, {y}): {snapf[y, x].tolist()} vs the oracle's {orgbf[y, x].tolist()}")
        what = (f"seed {seed}: {scene.width}x{scene.height}x{scene.spp} depth {scene.max_depth} proj {scene.proj} "
                f"{len(scene.spheres)} sph {len(scene.tri_verts)} tri {len(scene.lights)} lights, "
                f"|pos| ~ {float(np.abs(scene.cam_pos).max()):.3g}, rays {cnt.total}")
        if notes:
            bad += 1
            print("MISMATCH " + what, flush=True)
            for n in notes:
                print("    " + n, flush=True)
        else:
            print(f"ok {what} [{time.time() - t0:.0f} s]", flush=True)
    print(f"{bad} of {args.seeds[1] - args.seeds[0]} seeds with a mismatch", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
