#!/usr/bin/env python3
"""Throughput of the K3 frame with one or two frames in flight on ONE GPU: n contexts (a ray pool and a
stream each), a host thread per context, every thread renders its share of the frames back to back.  A frame's thin
late generations leave most of the chip idle; with a second frame in flight its fat generations can fill that time.
The frame itself is not faster -- this is a throughput figure for callers that render many frames (an animation, a
batch of views), not the bench line.     usage: frames_in_flight.py [frames per context = 20] [world = 1]
(world > 1: rank 0's shard of that many, as tools/sim_scaling.py renders it)"""
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402  (before the tracer library)
import raytracer_amd as ra  # noqa: E402


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    here = os.path.dirname(os.path.abspath(__file__))
    scene = ra.SceneArrays.load(os.path.join(here, "..", "tests", "golden", "scenes", "dragon.npz"))
    scene = scene.copy(width=3840, height=2160, spp=16, seed=1)
    nbytes = max(scene.width * scene.height * 3, ra.shard_bytes(scene.width, scene.height, world))

    def frame(t, b):
        return t.render_shard(0, world, b.data_ptr()) if world > 1 else t.render_device(b.data_ptr())
    for n in (1, 2):
        tracers, bufs = [], []
        for _ in range(n):
            t = ra.Tracer(device=0)
            t.load_scene(scene)
            b = torch.empty(nbytes, dtype=torch.uint8, device="cuda:0")
            for _ in range(3):
                frame(t, b)                             # calibration, pool growth, graph capture
            tracers.append(t); bufs.append(b)
        torch.cuda.synchronize()
        rays = [0] * n

        def work(k):
            for _ in range(frames):
                st = frame(tracers[k], bufs[k])
                rays[k] += st.rays

        th = [threading.Thread(target=work, args=(k,)) for k in range(n)]
        t0 = time.perf_counter()
        for x in th: x.start()
        for x in th: x.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({"world": world, "frames_in_flight": n, "frames": n * frames, "ms_per_frame_throughput": round(dt / (n * frames) * 1e3, 3),
                          "Mrays_per_s": round(sum(rays) / dt / 1e6, 1)}), flush=True)
        for t in tracers: t.close()


if __name__ == "__main__":
    main()
