"""Wall time per frame of the thin workloads with their batches queued launch by launch and replayed as HIP graphs
(rt_set_graphs), no timing events in either.  usage: python tools/graph_timing.py [frames]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import raytracer_amd as ra  # noqa: E402

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
torch.cuda.init()
FRAMES = int(sys.argv[1]) if len(sys.argv) > 1 else 200
CASES = [("k1", "spheres1", dict(width=640, height=480, spp=1), ra.RT_ACCEL_BRUTE),
         ("k2", "outdoor_spot", dict(width=1920, height=1080, spp=4), ra.RT_ACCEL_BRUTE),
         ("k4", "plant_h_smooth", dict(width=660, height=512, spp=1, max_depth=8), ra.RT_ACCEL_LBVH),
         ("k3/8", "dragon", dict(width=3840, height=2160, spp=16), ra.RT_ACCEL_LBVH),
         ("k3/2", "dragon", dict(width=3840, height=2160, spp=16), ra.RT_ACCEL_LBVH),
         ("k3", "dragon", dict(width=3840, height=2160, spp=16), ra.RT_ACCEL_LBVH)]
for tag, name, kw, accel in CASES:
    scene = ra.SceneArrays.load(os.path.join(ROOT, "tests", "golden", "scenes", name + ".npz")).copy(seed=1, **kw)
    line = {}
    ref = None
    for on in (False, True):
        with ra.Tracer(accel=accel) as t:
            t.set_graphs(on)
            t.load_scene(scene)
            shard = "/" in tag
            world = int(tag.split("/")[1]) if shard else 1
            nbytes = ra.shard_bytes(kw["width"], kw["height"], world) if shard else kw["width"] * kw["height"] * 3
            buf = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")

            def frame():
                if shard:
                    return t.render_shard(0, world, buf.data_ptr())
                return t.render_device(buf.data_ptr())
            for _ in range(4):
                frame()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(FRAMES if tag != 'k3' else 20):
                st = frame()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / (FRAMES if tag != 'k3' else 20) * 1e3
            out = buf.cpu().numpy().copy()
            if ref is None:
                ref = out
            line["graphs" if on else "direct"] = round(ms, 4)
            line["replays"] = t.set_graphs()
            line["same_bytes"] = bool(np.array_equal(ref, out))
            line["device_ms" if not on else "device_ms_graphs"] = round(st.ms_render, 4)
            line["enqueue_ms" if not on else "enqueue_ms_graphs"] = round(st.ms_host_enqueue, 4)
    print(tag, line, flush=True)
