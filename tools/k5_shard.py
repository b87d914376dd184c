#!/usr/bin/env python3
"""One GPU's share of the full K5 configuration (synthetic 10^6 spheres + 10^7 triangles,
7680x4320, 64 spp, image split over 8 GPUs): renders rank 0's shard of 8 on this GPU and
prints its time and ray rate.  8-GPU runs are the driver's to launch; this is the per-GPU work."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402  (before the tracer library, see tests/conftest.py)
import raytracer_amd as ra  # noqa: E402
from raytracer_amd.synth import synthetic_scene  # noqa: E402


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    scene = synthetic_scene(scale=1.0)
    t = ra.Tracer(device=0)
    t0 = time.perf_counter()
    t.load_scene(scene)
    load_s = time.perf_counter() - t0
    buf = torch.empty(ra.shard_bytes(scene.width, scene.height, world), dtype=torch.uint8, device="cuda:0")
    t0 = time.perf_counter()
    t.render_shard(0, world, buf.data_ptr())          # calibration + pool growth
    first_s = time.perf_counter() - t0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(frames):
        st = t.render_shard(0, world, buf.data_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / frames
    print(json.dumps({"workload": f"K5 full config, shard 0 of {world}", "ms_per_frame": round(dt * 1e3, 1),
                      "Mrays_per_s": round(st.rays / dt / 1e6, 1), "rays": st.rays, "primary": st.rays_primary,
                      "batches": st.batches, "retries": st.retries, "scene_load_s": round(load_s, 2),
                      "first_frame_s": round(first_s, 2)}))
    t.close()


if __name__ == "__main__":
    main()
