#!/usr/bin/env python3
"""Differential fuzzing of the HIP path against the CPU oracle on seeded random scenes.

  python tools/fuzz_parity.py [--seeds A B] [--trees] [--big] [--save]

For every seed: the oracle's frame, then the HIP path by brute force and through the hierarchy with every walk
forced in turn (and, with --trees, on both hierarchies): three frames per context (the second runs its thin
generations on two streams and is captured as a HIP graph, the third replays it), some contexts with small batches or a ray pool that overflows, and for two of the
variants the frame assembled from the shards of a random number of ranks.  Prints one line per mismatch and a summary; exit
status 1 if anything differed.  Needs a GPU.

FUZZ_THREADS=n: OpenMP threads of the oracle (default 16: spinning threads in the process are the condition under
which a stale completion handler of the HIP runtime was caught writing into freed host memory -- see acquire_stream in
raytracer_amd/csrc/rt_api.hip and tools/heap_probe.py -- so the campaigns keep that pressure on)."""
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

SCRATCH = []
ENV_KEYS = ("RT_PACKET_GENS", "RT_PACKET_SHADOW_GENS", "RT_LANE_WIDE", "RT_TREE", "RT_MAX_BLOCKS")


def render(scene, accel, env, frames=1, shards=0, **kw):
    """-> list of (rgb, rgbf, stats), one per frame rendered on ONE context (the second frame knows the ray counts of
    the first: thin generations then run on two streams); with shards = n one more entry: the frame assembled from the
    n shards (rgbf None, stats of shard 0)."""
    for k in ENV_KEYS:
        os.environ.pop(k, None)
    os.environ.update(env)
    try:
        with ra.Tracer(device=0, accel=accel, **kw) as t:
            t.load_scene(scene)
            out = []
            for k in range(frames):
                if os.environ.get("FUZZ_SCRATCH") == "1":
                    # the library never sees the arrays that are compared: it writes r2 / f2, numpy copies them over
                    rgb = np.zeros((scene.height, scene.width, 3), np.uint8)
                    r2, f2, st = t.render(want_float=True)
                    rgb[:] = r2
                    out.append((rgb, f2, st))
                    SCRATCH.append((k, r2, rgb))
                else:
                    out.append(t.render(want_float=True))
            if shards:
                sb = ra.shard_bytes(scene.width, scene.height, shards)
                allsh = torch.zeros(shards * sb, dtype=torch.uint8, device="cuda:0")
                sts = [t.render_shard(r, shards, allsh[r * sb:(r + 1) * sb].data_ptr()) for r in range(shards)]
                frame = torch.zeros((scene.height, scene.width, 3), dtype=torch.uint8, device="cuda:0")
                t.assemble_shards(allsh.data_ptr(), shards, frame.data_ptr())
                torch.cuda.synchronize()
                st = sts[0]
                st.rays_primary = sum(x.rays_primary for x in sts); st.rays_secondary = sum(x.rays_secondary for x in sts)
                st.rays_shadow = sum(x.rays_shadow for x in sts)
                out.append((frame.cpu().numpy(), None, st))
            return out
    finally:
        for k in ENV_KEYS:
            os.environ.pop(k, None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, nargs=2, default=[0, 40])
    ap.add_argument("--trees", action="store_true", help="force each hierarchy (Karras, PLOC, binned surface-area build) in turn as well")
    ap.add_argument("--big", action="store_true", help="thousands of primitives, larger films (seconds of oracle time per seed)")
    ap.add_argument("--save", action="store_true", help="write the frames of a mismatch to gpurun_out/fuzz_<seed>_<variant>.npz")
    args = ap.parse_args()
    orc = Oracle()
    bad = 0
    t0 = time.time()
    for seed in range(*args.seeds):
        scene = random_scene(seed, big=args.big)
        orgb, orgbf, cnt, _ = orc.render(scene, threads=int(os.environ.get("FUZZ_THREADS", "16")))
        variants = [("brute", ra.RT_ACCEL_BRUTE, {})]
        for walk, env in WALKS.items():
            for tree in (("karras", "ploc", "sah") if args.trees else (None,)):
                e = dict(env)
                if tree:
                    e["RT_TREE"] = tree
                variants.append((walk + ("/" + tree if tree else ""), ra.RT_ACCEL_LBVH, e))
        variants.append(("calibrated", ra.RT_ACCEL_LBVH, {}))
        notes = []
        rng = np.random.default_rng(seed)
        for label, accel, env in variants:
            if accel != ra.RT_ACCEL_BRUTE and len(scene.spheres) + len(scene.tri_verts) < 2:
                continue                                                   # a single primitive has no hierarchy
            kw = {}
            if rng.random() < 0.3:
                kw["batch_samples"] = int(rng.choice([1024, 4096, 20000]))
            if rng.random() < 0.2:
                kw["pool_rays"] = int(rng.choice([3000, 30000]))
            shards = int(rng.integers(1, 6)) if label in ("calibrated", "brute") else 0
            try:
                frames = render(scene, accel, env, frames=3, shards=shards, **kw)
            except ra.RtError as e:
                notes.append(f"{label} {kw}: error {e}")
                continue
            for k, (rgb, rgbf, st) in enumerate(frames):
                tag = f"{label} {kw or ''} " + ("frame %d" % k if rgbf is not None else "assembled from %d shards" % shards)
                def compare():
                    return same_float_image(rgbf, orgbf).all(axis=2) if rgbf is not None else (rgb == orgb).all(axis=2)
                same = compare()          # looked at once: a first-read difference is a mismatch
                counts = (st.rays_primary, st.rays_secondary, st.rays_shadow) == (cnt.primary, cnt.secondary, cnt.shadow)
                if not same.all() or not np.array_equal(rgb, orgb) or not counts:
                    notes.append(f"{tag}: {int((~same).sum())} of {same.size} pixels differ, "
                                 f"{int((rgb != orgb).any(axis=2).sum())} in RGB8, rays {st.rays_primary}/{st.rays_secondary}/{st.rays_shadow} "
                                 f"vs {cnt.primary}/{cnt.secondary}/{cnt.shadow}")
                    if args.save:
                        np.savez(os.path.join(ROOT, "gpurun_out", f"fuzz_{seed}_{label.replace('/', '_')}_{k}.npz"), rgbf=rgbf if rgbf is not None else 0, rgb=rgb, orgbf=orgbf, orgb=orgb)
                    if same.all():
                        for y, x in list(zip(*np.nonzero((rgb != orgb).any(axis=2))))[:4]:
                            notes.append(f"    RGB8 pixel ({x}, {y}): {rgb[y, x].tolist()} vs the oracle's {orgb[y, x].tolist()}; float {rgbf[y, x].tolist() if rgbf is not None else None}")
                    for y, x in list(zip(*np.nonzero(~same)))[:4]:
                        notes.append(f"    pixel ({x}, {y}): {(rgbf if rgbf is not None else rgb)[y, x].tolist()} vs the oracle's {(orgbf if rgbf is not None else orgb)[y, x].tolist()}")
        for k, r2, rgb in SCRATCH:
            if not np.array_equal(r2, orgb) or not np.array_equal(rgb, orgb):
                notes.append(f"    SCRATCH frame {k}: array the library wrote differs at bytes {np.nonzero(r2.ravel() != orgb.ravel())[0].tolist()[:8]}, "
                             f"the numpy copy of it at bytes {np.nonzero(rgb.ravel() != orgb.ravel())[0].tolist()[:8]}")
        del SCRATCH[:]
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
