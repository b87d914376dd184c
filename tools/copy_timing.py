"""rt_render (frame copied into the caller's pageable host buffer) against rt_render_device (frame left in HBM), K3
frame, with the staged multi-threaded copy and with one plain copy (RT_FAST_COPY=0).  usage: python tools/copy_timing.py"""
import hashlib
import os
import sys
import time

import numpy as np
import torch

torch.cuda.init()
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import raytracer_amd as ra  # noqa: E402

scene = ra.SceneArrays.load(os.path.join(ROOT, "tests", "golden", "scenes", "dragon.npz")).copy(seed=1, width=3840, height=2160, spp=16)
buf = torch.zeros(3840 * 2160 * 3, dtype=torch.uint8, device="cuda")
for fast in ("1", "0"):
    os.environ["RT_FAST_COPY"] = fast
    with ra.Tracer(accel=ra.RT_ACCEL_LBVH) as t:
        t.load_scene(scene)
        out = {}
        for what in ("device", "host", "host+float"):
            def frame():
                if what == "device":
                    return t.render_device(buf.data_ptr())
                return t.render(want_float=(what == "host+float"))
            for _ in range(3):
                r = frame()
            t0 = time.perf_counter()
            for _ in range(10):
                r = frame()
            torch.cuda.synchronize()
            out[what] = round((time.perf_counter() - t0) / 10 * 1e3, 3)
            if what != "device":
                out[what + " sha"] = hashlib.sha256(np.ascontiguousarray(r[0]).tobytes()).hexdigest()[:12]
                if what == "host+float":
                    out["float sha"] = hashlib.sha256(np.ascontiguousarray(r[1]).tobytes()).hexdigest()[:12]
            else:
                out["device sha"] = hashlib.sha256(buf.cpu().numpy().tobytes()).hexdigest()[:12]
        print("RT_FAST_COPY=" + fast, out, flush=True)
