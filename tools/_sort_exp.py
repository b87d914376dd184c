import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import raytracer_amd as ra
here = os.path.dirname(os.path.abspath(__file__))
scene = ra.SceneArrays.load(os.path.join(here, "..", "tests", "golden", "scenes", "dragon.npz"))
rng = np.random.default_rng(5)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
tv = scene.tri_verts.reshape(-1, 3, 3)
# skip the two huge floor triangles half of the time: origins follow surface AREA loosely, like real bounces
area = 0.5 * np.linalg.norm(np.cross(tv[:, 1] - tv[:, 0], tv[:, 2] - tv[:, 0]), axis=1)
small = area < np.percentile(area, 99.9)
p = np.where(small, area, 0); p = p / p.sum()
ti = rng.choice(len(tv), size=n, p=p)
u = rng.random((n, 2)).astype(np.float32); f = u.sum(1) > 1; u[f] = 1 - u[f]
o = tv[ti, 0] + u[:, :1] * (tv[ti, 1] - tv[ti, 0]) + u[:, 1:] * (tv[ti, 2] - tv[ti, 0])
d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
o = (o + 1e-3 * d).astype(np.float32)
lo, hi = o.min(0), o.max(0)
q = np.clip(((o - lo) / (hi - lo) * 1023).astype(np.uint64), 0, 1023)
def spread(v):
    v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F
    v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249
    return v
mort = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
octant = ((d[:, 0] < 0).astype(np.uint64) | ((d[:, 1] < 0).astype(np.uint64) << 1) | ((d[:, 2] < 0).astype(np.uint64) << 2))
orders = {"unsorted": np.arange(n), "morton": np.argsort(mort, kind="stable"),
          "octant_morton": np.argsort((octant << 30) | mort, kind="stable"),
          "morton15_octant": np.argsort(((mort >> 15) << 3) | octant, kind="stable")}
import time
with ra.Tracer(accel=ra.RT_ACCEL_LBVH) as t:
    t.load_scene(scene)
    ref = None
    for name, idx in orders.items():
        for accel, an in ((ra.RT_ACCEL_LBVH_PACKET, "packet"), (ra.RT_ACCEL_LBVH, "lane"), (ra.RT_ACCEL_LBVH_WIDE, "wide")):
            for any_hit in (False, True):
                oo, dd = np.ascontiguousarray(o[idx]), np.ascontiguousarray(d[idx])
                prim, d2 = t.query_rays(oo, dd, any_hit=any_hit, accel=accel)
                if not any_hit:
                    back = np.empty_like(prim); back[idx] = prim
                    if ref is None: ref = back
                    assert np.array_equal(ref, back)
    print("hit fraction", float((ref >= 0).mean()))
