#!/usr/bin/env python3
"""Host-heap canaries around the tracer library's calls: who writes into freed memory?

  python tools/heap_probe.py [--seeds A B] [--variants brute,packet,...] [--hold N] [--keep-alive]

The loop of tools/fuzz_parity.py (oracle on FUZZ_THREADS threads, then the HIP path, three frames per context), with
blocks of every small malloc size class taken from the C heap before and after every render call, filled with a
pattern and checked a few calls later.  A block that changed while nobody owned a pointer into it was written by a
stale pointer; what the block held when malloc handed it out (the freed object's bytes) is printed with the process's
memory map, so that pointers in it (vtables, neighbours) can be named.  Needs a GPU.  Exit status 1 if a canary changed.

What it found: with streams destroyed per context the process died within seconds (a pointer of a live runtime
object with its low half cleared, or glibc's "corrupted double-linked list") -- the writer is the completion handler
the HIP runtime of ROCm 7.0 registers for marker commands, which decrements a counter at +0x98 and clears the word at
+0x378 of the stream's queue object (920 bytes) after hipStreamDestroy has freed it.  PROBE_NO_TORCH=1 runs the same
loop on the system's runtime instead of the one torch bundles."""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

if os.environ.get("PROBE_NO_TORCH") != "1":      # torch first: the process then runs on the HIP runtime torch bundles
    import torch  # noqa: F401,E402
import raytracer_amd as ra  # noqa: E402
from _fuzz_scenes import WALKS, random_scene  # noqa: E402
from _oracle import Oracle  # noqa: E402

_TRACE = os.path.join(ROOT, "tools", "libsegv_trace.so")      # prints the call chain of a thread that faults (tools/segv_trace.c)
_trace = C.CDLL(_TRACE) if os.path.exists(_TRACE) else None
libc = C.CDLL("libc.so.6")
libc.malloc.restype = C.c_void_p
libc.malloc.argtypes = [C.c_size_t]
libc.free.argtypes = [C.c_void_p]
PATTERN = 0x5A
ENV_KEYS = ("RT_PACKET_GENS", "RT_PACKET_SHADOW_GENS", "RT_LANE_WIDE", "RT_TREE")


class Canaries:
    def __init__(self, sizes, hold, fill=True, each=1):
        self.sizes = sizes
        self.hold = hold
        self.fill = fill        # False: the blocks keep what they held (a stale writer then meets the values it expects)
        self.each = each
        self.live = []          # (tick, tag, size, address, bytes the block held when it was handed out)
        self.tick = 0
        self.hits = []

    def take(self, tag):
        for n in self.sizes * self.each:
            p = libc.malloc(n)
            if not p:
                continue
            before = C.string_at(p, n)
            if self.fill:
                C.memset(p, PATTERN, n)
            self.live.append((self.tick, tag, n, p, before))

    def check(self, now_tag, everything=False):
        self.tick += 1
        keep = []
        for rec in self.live:
            tick, tag, n, p, before = rec
            if not everything and self.tick - tick < self.hold:
                keep.append(rec)
                continue
            cur = np.frombuffer(C.string_at(p, n), np.uint8)
            was = np.frombuffer(before, np.uint8)
            diff = np.nonzero(cur != (PATTERN if self.fill else was))[0]
            if len(diff):
                self.hits.append((tag, now_tag, n, p, diff.tolist(), cur[diff].tolist(), before))
                print(f"CANARY size {n} at {p:#x} taken at [{tag}] checked at [{now_tag}]: bytes {diff.tolist()[:24]} now "
                      f"{[hex(v) for v in cur[diff].tolist()[:24]]} were {[hex(v) for v in (was[diff].tolist()[:24] if not self.fill else [PATTERN])]}", flush=True)
                words = np.frombuffer(before[: n // 8 * 8], np.uint64)
                print("    held before: " + " ".join(f"{8 * i}:{int(w):x}" for i, w in enumerate(words) if w), flush=True)
            libc.free(p)
        self.live = keep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, nargs=2, default=[200000, 200400])
    ap.add_argument("--variants", default="brute,packet,lane,lane-wide,team,calibrated")
    ap.add_argument("--hold", type=int, default=3, help="checks a canary block survives before it is verified and freed")
    ap.add_argument("--frames", type=int, default=3)
    ap.add_argument("--lo", type=int, default=24)
    ap.add_argument("--hi", type=int, default=2056)
    ap.add_argument("--no-fill", action="store_true", help="the blocks keep what they held when malloc handed them out")
    ap.add_argument("--each", type=int, default=1, help="blocks per size and probe")
    ap.add_argument("--keep-alive", action="store_true", help="contexts are destroyed at the end of the run, not per variant")
    args = ap.parse_args()
    threads = int(os.environ.get("FUZZ_THREADS", "16"))
    orc = Oracle()
    can = Canaries(list(range(args.lo, args.hi + 1, 16)), args.hold, fill=not args.no_fill, each=args.each)
    wanted = args.variants.split(",")
    alive = []
    t0 = time.time()
    for seed in range(*args.seeds):
        scene = random_scene(seed)
        orc.render(scene, threads=threads)
        can.check(f"seed {seed} after the oracle")
        variants = [("brute", ra.RT_ACCEL_BRUTE, {})] + [(w, ra.RT_ACCEL_LBVH, e) for w, e in WALKS.items()] + [("calibrated", ra.RT_ACCEL_LBVH, {})]
        for label, accel, env in variants:
            if label not in wanted:
                continue
            if accel != ra.RT_ACCEL_BRUTE and len(scene.spheres) + len(scene.tri_verts) < 2:
                continue
            for k in ENV_KEYS:
                os.environ.pop(k, None)
            os.environ.update(env)
            t = ra.Tracer(device=0, accel=accel)
            if _trace is not None:
                _trace.segv_trace_install()       # (the runtimes install handlers of their own when they start)
            try:
                t.load_scene(scene)
                for k in range(args.frames):
                    tag = f"seed {seed} {label} depth {scene.max_depth} frame {k}"
                    can.take(tag + " before")
                    t.render(want_float=True)
                    can.take(tag + " after")
                    can.check(tag)
            finally:
                if args.keep_alive:
                    alive.append(t)
                    if len(alive) > 64:
                        alive.pop(0).close()
                else:
                    t.close()
                for k in ENV_KEYS:
                    os.environ.pop(k, None)
        if seed % 100 == 0:
            print(f"seed {seed} [{time.time() - t0:.0f} s] {len(can.hits)} damaged canaries so far", flush=True)
    for t in alive:
        t.close()
    time.sleep(0.2)
    can.check("end", everything=True)
    print(f"{len(can.hits)} damaged canaries in {args.seeds[1] - args.seeds[0]} seeds", flush=True)
    if can.hits:
        print("memory map (file-backed mappings):")
        with open("/proc/self/maps") as f:
            for line in f:
                if "/" in line and (" r-xp " in line or " r--p " in line or " rw-p " in line) and ".so" in line:
                    print("    " + line.rstrip())
    return 1 if can.hits else 0


if __name__ == "__main__":
    sys.exit(main())
