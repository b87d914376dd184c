"""GPU suite (-m gpu): the HIP path, called through the C-ABI, against the CPU oracle run
live on the same inputs and against the committed reference renders.

Tolerance.  The contract is <= 1 LSB per 8-bit channel on >= 99.9 % of pixels.  The HIP
kernels restate the reference's arithmetic operation for operation, so the tests demand
more: float colours bit-identical on >= 99.9 % of pixels (the only licensed difference is
the device libm rounding pow()/acos() differently from glibc in the last place), RGB8
within 1 LSB everywhere, and identical per-class ray counts."""
import numpy as np
import pytest

import raytracer_amd as ra
from conftest import bits, golden_cases, hip_runtime_version, load_case, load_scene, same_float_image

pytestmark = pytest.mark.gpu

CONTRACT_FRACTION = 0.999   # the contract: pixels within 1 LSB per channel
# Measured on MI355X: every comparison of this suite is bit-identical in float on EVERY pixel (device pow / acos
# round like glibc on all inputs met so far), so that is what is demanded; a case that needs the contract's
# slack must be listed here with its measured fraction.  The fractions of a run are written to
# gpurun_out/parity_report.json.
EXACT_FRACTION = 1.0        # pixels whose float colour must be bit-identical
LSB_FRACTION = 1.0          # pixels whose RGB8 must be within 1 LSB
LIBM_SLACK = {}             # what -> accepted bit-identical fraction (>= CONTRACT_FRACTION)
REPORT = {}


def crop(img, roi):
    x0, y0, x1, y1 = roi
    return img[y0:y1, x0:x1]


def check_image(rgb, rgbf, gold_rgb, gold_rgbf, what):
    same = same_float_image(rgbf, gold_rgbf).all(axis=2)
    frac = float(same.mean())
    lsb = float((np.abs(rgb.astype(np.int32) - gold_rgb.astype(np.int32)) <= 1).all(axis=2).mean())
    byte_same = float((rgb == gold_rgb).all(axis=2).mean())
    REPORT[what] = {"pixels": int(same.size), "float_bit_identical": frac, "rgb8_identical": byte_same, "rgb8_within_1lsb": lsb}
    need = max(CONTRACT_FRACTION, LIBM_SLACK.get(what, EXACT_FRACTION))
    assert lsb >= LSB_FRACTION, f"{what}: only {lsb:.6f} of pixels within 1 LSB"
    assert frac >= need, f"{what}: only {frac:.6f} of pixels bit-identical in float ({int(round((1 - frac) * same.size))} differ)"
    return frac


@pytest.fixture(scope="module", autouse=True)
def _parity_report():
    yield
    import json
    import os
    from conftest import ROOT
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_report.json"), "w") as f:
            json.dump(REPORT, f, indent=1, sort_keys=True)
    except OSError:
        pass


@pytest.mark.parametrize("accel", [ra.RT_ACCEL_BRUTE, ra.RT_ACCEL_LBVH])
@pytest.mark.parametrize("case", golden_cases())
def test_render_matches_reference_fixture_and_oracle(oracle, case, accel):
    scene, roi, gold = load_case(case)
    if accel == ra.RT_ACCEL_LBVH and len(scene.spheres) + len(scene.tri_verts) < 2:
        pytest.skip("LBVH needs two primitives")
    with ra.Tracer(accel=accel) as t:
        t.load_scene(scene)
        rgb, rgbf, st = t.render(want_float=True)
    check_image(crop(rgb, roi), crop(rgbf, roi), gold["rgb"], gold["rgbf"], f"{case} vs reference fixture")
    full = roi == (0, 0, scene.width, scene.height)
    if full:
        assert [st.rays_primary, st.rays_secondary, st.rays_shadow] == [int(v) for v in gold["counts"]]
        assert st.accel_used == accel


@pytest.mark.parametrize("name,w,h,spp", [("spheres1", 640, 480, 1), ("outdoor_spot", 480, 270, 4),
                                          ("spheres2", 200, 150, 5), ("test_reasonable", 320, 240, 2)])
def test_larger_frames_against_live_oracle(oracle, tracer, name, w, h, spp):
    scene = load_scene(name).copy(width=w, height=h, spp=spp, seed=11)
    tracer.load_scene(scene)
    rgb, rgbf, st = tracer.render(want_float=True)
    orgb, orgbf, cnt, _ = oracle.render(scene, threads=16)
    check_image(rgb, rgbf, orgb, orgbf, f"{name} vs live oracle")
    assert (st.rays_primary, st.rays_secondary, st.rays_shadow) == (cnt.primary, cnt.secondary, cnt.shadow)


@pytest.mark.parametrize("accel", [ra.RT_ACCEL_BRUTE, ra.RT_ACCEL_LBVH])
@pytest.mark.parametrize("w,h,spp", [(320, 240, 1), (256, 192, 4)])
def test_spot_light_falloff_branches(oracle, accel, w, h, spp):
    """Spot lights with angles in the range of acos(): the point-like, the zero and the interpolated branch
    (src/raytrace.cpp:549-566) all run -- counted in the oracle -- and go through the device acos()."""
    scene = load_scene("spot_falloff").copy(width=w, height=h, spp=spp, seed=9)
    oracle.spot_branch_counts(reset=True)
    orgb, orgbf, cnt, _ = oracle.render(scene, threads=16)
    br = oracle.spot_branch_counts()
    assert min(br) > 0.15 * sum(br), br
    with ra.Tracer(accel=accel) as t:
        t.load_scene(scene)
        rgb, rgbf, st = t.render(want_float=True)
    check_image(rgb, rgbf, orgb, orgbf, f"spot fall-off {w}x{h}x{spp} accel {accel}")
    assert (st.rays_primary, st.rays_secondary, st.rays_shadow) == (cnt.primary, cnt.secondary, cnt.shadow)


def test_k2_full_size_frame_against_the_oracle(oracle, tracer):
    """The second benchmark configuration at its stated size: outdoor + spot light, 1920x1080, 4 spp, the whole frame."""
    scene = load_scene("outdoor_spot").copy(width=1920, height=1080, spp=4, seed=1)
    tracer.load_scene(scene)
    rgb, rgbf, st = tracer.render(want_float=True)
    orgb, orgbf, cnt, _ = oracle.render(scene, threads=16)
    check_image(rgb, rgbf, orgb, orgbf, "K2 1920x1080x4 whole frame")
    assert (st.rays_primary, st.rays_secondary, st.rays_shadow) == (cnt.primary, cnt.secondary, cnt.shadow)


def test_k4_full_size_frame_against_the_oracle(oracle, tracer):
    """The fourth benchmark configuration at its stated size: plant-h as normal_triangle, 660x512, max_depth 8.  The
    oracle tests all 5 247 primitives per ray, so it renders three horizontal bands of the frame (a quarter of
    it); the ray counts of the whole frame are pinned by brute force on the GPU."""
    scene = load_scene("plant_h_smooth").copy(width=660, height=512, spp=1, max_depth=8, seed=1)
    tracer.load_scene(scene)
    rgb, rgbf, st = tracer.render(want_float=True)
    assert st.accel_used == ra.RT_ACCEL_LBVH
    for roi in [(0, 100, 660, 140), (0, 250, 660, 300), (0, 400, 660, 440)]:
        orgb, orgbf, _, _ = oracle.render(scene, roi=roi, threads=16)
        check_image(crop(rgb, roi), crop(rgbf, roi), orgb, orgbf, f"K4 660x512 depth 8, band {roi}")
    with ra.Tracer(accel=ra.RT_ACCEL_BRUTE) as t:
        t.load_scene(scene)
        brgb, brgbf, bst = t.render(want_float=True)
    assert np.array_equal(bits(brgbf), bits(rgbf)) and np.array_equal(brgb, rgb)
    assert (st.rays_primary, st.rays_secondary, st.rays_shadow) == (bst.rays_primary, bst.rays_secondary, bst.rays_shadow)


def test_k1_ray_count_pin(tracer):
    tracer.load_scene(load_scene("spheres1"))
    rgb, _, st = tracer.render()
    assert (st.rays_primary, st.rays_secondary, st.rays_shadow) == (307200, 1267376, 805513)
    assert tuple(rgb[0, 0]) == (12, 12, 12)


@pytest.mark.parametrize("key,scene_name,spp", [("spheres1_spp1", "spheres1", 1), ("spheres1_spp5", "spheres1", 5),
                                                ("outdoor_spp3", "outdoor", 3), ("dragon_spp1", "dragon", 1)])
def test_primary_rays_and_closest_hits_known_answers(oracle, tracer, kat, key, scene_name, spp):
    scene = load_scene(scene_name).copy(spp=spp)
    tracer.load_scene(scene)
    xyj = kat[f"ray_{key}_xyj"]
    o, d = tracer.primary_rays(xyj[:, 0], xyj[:, 1], xyj[:, 2])
    assert np.array_equal(bits(o), bits(kat[f"ray_{key}_o"]))
    assert np.array_equal(bits(d), bits(kat[f"ray_{key}_d"]))
    ref_hit = kat[f"hit_{key}"]
    want_prim, want_d2 = oracle.intersect_many(scene, o, d)
    for accel in (ra.RT_ACCEL_BRUTE,) + TREE_WALKS:
        prim, d2 = tracer.query_rays(o, d, accel=accel)
        assert np.array_equal(prim >= 0, ref_hit[:, 0] != 0)
        assert np.array_equal(prim, want_prim)
        hit = prim >= 0
        assert np.array_equal(d2[hit].astype(np.float64), ref_hit[hit, 4])   # reference dist (squared)


# every walk of every hierarchy: per-lane / packet / per-lane 4-wide / team of lanes per ray, on the Karras, the PLOC and
# the binned surface-area tree
TREE_WALKS = tuple(w | t for t in (0, ra.RT_ACCEL_ALT_TREE, ra.RT_ACCEL_SAH_TREE)
                   for w in (ra.RT_ACCEL_LBVH, ra.RT_ACCEL_LBVH_PACKET, ra.RT_ACCEL_LBVH_WIDE, ra.RT_ACCEL_LBVH_TEAM))


def walks_held(t):
    """TREE_WALKS without the walks of a hierarchy that was not built for the scene loaded in tracer t (the binned
    surface-area build is left out for scenes beyond 2^18 primitives)"""
    depth = t.tree_stats().depth
    return tuple(w for w in TREE_WALKS if depth[2 if w & ra.RT_ACCEL_SAH_TREE else 1 if w & ra.RT_ACCEL_ALT_TREE else 0] > 0)


def _random_rays(scene, n, rng):
    """Rays aimed at the primitives.  Four in five start inside the box of the primitives, where the rays of a
    render start and for which the hierarchies are built to be conservative; the rest start up to an extent
    outside it, where a query is answered by brute force whatever walk was asked for."""
    prim_pts = []
    if len(scene.spheres):
        prim_pts.append(scene.spheres[:, :3])
    if len(scene.tri_verts):
        prim_pts.append(scene.tri_verts.reshape(-1, 3))
    pts = np.concatenate(prim_pts)
    lo, hi = pts.min(0), pts.max(0)
    ext = np.maximum(hi - lo, 1e-3)
    o = lo + rng.random((n, 3)) * (hi - lo)
    far = rng.random(n) < 0.2
    o[far] = ((lo - ext) + rng.random((n, 3)) * 3 * ext)[far]
    tgt = pts[rng.integers(0, len(pts), n)] + rng.normal(0, 0.02, (n, 3)) * ext
    d = tgt - o
    d *= rng.uniform(0.2, 3.0, (n, 1)) / np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-30)   # un-normalised on purpose
    return o.astype(np.float32), d.astype(np.float32)


@pytest.mark.parametrize("name", ["dragon", "plant_h", "complex_test", "spheres2"])
def test_lbvh_equals_brute_force_on_random_rays(tracer, name):
    scene = load_scene(name)
    tracer.load_scene(scene)
    rng = np.random.default_rng(5)
    n = 200_000
    o, d = _random_rays(scene, n, rng)
    pb, db = tracer.query_rays(o, d, accel=ra.RT_ACCEL_BRUTE)
    assert (pb >= 0).mean() > 0.2, "ray generator barely hits the scene"
    for accel in TREE_WALKS:
        pl, dl = tracer.query_rays(o, d, accel=accel)
        assert np.array_equal(pb, pl), f"accel {accel}: {(pb != pl).sum()} of {n} closest hits differ"
        assert np.array_equal(bits(db), bits(dl))
    # any-hit with a finite window (the shadow query): same yes/no
    dmax = np.where(rng.random(n) < 0.5, db * rng.uniform(0.5, 1.5, n).astype(np.float32), np.float32(np.inf)).astype(np.float32)
    dmax[~np.isfinite(dmax)] = np.inf
    ab, _ = tracer.query_rays(o, d, dmax=dmax, any_hit=True, accel=ra.RT_ACCEL_BRUTE)
    for accel in TREE_WALKS:
        al, _ = tracer.query_rays(o, d, dmax=dmax, any_hit=True, accel=accel)
        assert np.array_equal(ab >= 0, al >= 0)


@pytest.mark.parametrize("name", ["dragon", "plant_h", "spheres2"])
def test_quantised_wide_nodes_equal_brute_force(monkeypatch, name):
    """Trees beyond the caches store their 4-wide nodes with 8-bit child boxes (conservatively rounded outward);
    forced here on scenes that would get float boxes: the per-lane wide walk of both hierarchies must still find
    the brute-force hit for every ray, closest-hit and any-hit, and render the same image."""
    scene = load_scene(name).copy(width=160, height=120, spp=2)
    rng = np.random.default_rng(17)
    o, d = _random_rays(scene, 100_000, rng)
    monkeypatch.setenv("RT_WIDE_QUANT", "1")
    monkeypatch.setenv("RT_PACKET_GENS", "0")
    monkeypatch.setenv("RT_PACKET_SHADOW_GENS", "0")
    monkeypatch.setenv("RT_LANE_WIDE", "1")
    with ra.Tracer(accel=ra.RT_ACCEL_LBVH) as t:
        t.load_scene(scene)
        pb, db = t.query_rays(o, d, accel=ra.RT_ACCEL_BRUTE)
        dmax = np.where(rng.random(len(o)) < 0.5, db * rng.uniform(0.5, 1.5, len(o)).astype(np.float32), np.float32(np.inf)).astype(np.float32)
        dmax[~np.isfinite(dmax)] = np.inf
        ab, _ = t.query_rays(o, d, dmax=dmax, any_hit=True, accel=ra.RT_ACCEL_BRUTE)
        for accel in (ra.RT_ACCEL_LBVH_WIDE, ra.RT_ACCEL_LBVH_WIDE | ra.RT_ACCEL_ALT_TREE, ra.RT_ACCEL_LBVH_WIDE | ra.RT_ACCEL_SAH_TREE,
                      ra.RT_ACCEL_LBVH_TEAM, ra.RT_ACCEL_LBVH_TEAM | ra.RT_ACCEL_ALT_TREE, ra.RT_ACCEL_LBVH_TEAM | ra.RT_ACCEL_SAH_TREE):
            pl, dl = t.query_rays(o, d, accel=accel)
            assert np.array_equal(pb, pl), f"accel {accel}: {(pb != pl).sum()} closest hits differ"
            assert np.array_equal(bits(db), bits(dl))
            al, _ = t.query_rays(o, d, dmax=dmax, any_hit=True, accel=accel)
            assert np.array_equal(ab >= 0, al >= 0)
        rgb_q, f_q, st_q = t.render(want_float=True)
    monkeypatch.setenv("RT_WIDE_QUANT", "0")
    with ra.Tracer(accel=ra.RT_ACCEL_LBVH) as t:
        t.load_scene(scene)
        rgb_f, f_f, st_f = t.render(want_float=True)
    assert np.array_equal(rgb_q, rgb_f) and np.array_equal(bits(f_q), bits(f_f)) and st_q.rays == st_f.rays


def test_query_matches_oracle_on_random_rays(oracle, tracer):
    scene = load_scene("complex_test")
    tracer.load_scene(scene)
    o, d = _random_rays(scene, 3000, np.random.default_rng(9))
    want_prim, want_d2 = oracle.intersect_many(scene, o, d)
    prim, d2 = tracer.query_rays(o, d, accel=ra.RT_ACCEL_BRUTE)
    assert np.array_equal(prim, want_prim)
    assert np.array_equal(bits(d2[prim >= 0]), bits(want_d2[prim >= 0]))


def test_image_is_independent_of_batching_and_pool_overflow_retries():
    scene = load_scene("spheres1").copy(width=320, height=240, spp=2)
    imgs = []
    for kw in ({}, {"batch_samples": 4096}, {"batch_samples": 65536, "pool_rays": 70000}):
        with ra.Tracer(**kw) as t:
            t.load_scene(scene)
            rgb, rgbf, st = t.render(want_float=True)
            imgs.append((rgb, rgbf, st))
    assert imgs[2][2].retries > 0, "the small pool was meant to overflow and retry"
    for rgb, rgbf, st in imgs[1:]:
        assert np.array_equal(bits(rgbf), bits(imgs[0][1])) and np.array_equal(rgb, imgs[0][0])
        assert (st.rays_secondary, st.rays_shadow) == (imgs[0][2].rays_secondary, imgs[0][2].rays_shadow)


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
def test_sharded_render_is_byte_identical_for_any_world_size(tracer, world):
    import torch
    from raytracer_amd import dist as rd
    scene = load_scene("outdoor_spot").copy(width=200, height=150, spp=2)
    tracer.load_scene(scene)
    full, _, st_full = tracer.render()
    sb = ra.shard_bytes(scene.width, scene.height, world)
    allsh = torch.zeros(world * sb, dtype=torch.uint8, device="cuda:0")
    rays = 0
    for r in range(world):
        st = tracer.render_shard(r, world, allsh[r * sb:(r + 1) * sb].data_ptr())
        rays += st.rays
        torch.cuda.synchronize()
        # the packed layout is the one the host mirror (used by the gloo test) produces
        n = ra.shard_tiles(scene.width, scene.height, r, world) * ra.RT_TILE * ra.RT_TILE * 3
        host = rd.pack_shard_host(full, r, world)
        got = allsh[r * sb:(r + 1) * sb].cpu().numpy()
        tiles = rd.tile_list(scene.width, scene.height, r, world)
        for i, (tx, ty) in enumerate(tiles):
            hh = min(ra.RT_TILE, scene.height - ty * ra.RT_TILE); ww = min(ra.RT_TILE, scene.width - tx * ra.RT_TILE)
            a = got[: n].reshape(-1, ra.RT_TILE, ra.RT_TILE, 3)[i, :hh, :ww]
            b = host[: n].reshape(-1, ra.RT_TILE, ra.RT_TILE, 3)[i, :hh, :ww]
            assert np.array_equal(a, b)
    assert rays == st_full.rays
    frame = torch.zeros((scene.height, scene.width, 3), dtype=torch.uint8, device="cuda:0")
    tracer.assemble_shards(allsh.data_ptr(), world, frame.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(frame.cpu().numpy(), full)
    assert np.array_equal(rd.assemble_host(allsh.cpu().numpy(), world, scene.width, scene.height), full)


def test_synthetic_scene_small_scale_against_oracle(oracle, tracer):
    """The fifth benchmark configuration at 1/1000 scale (10^3 spheres + 10^4 triangles): LBVH render vs the oracle."""
    from raytracer_amd.synth import synthetic_scene
    scene = synthetic_scene(scale=0.001, width=160, height=90, spp=2)
    tracer.load_scene(scene)
    rgb, rgbf, st = tracer.render(want_float=True)
    assert st.accel_used == ra.RT_ACCEL_LBVH
    orgb, orgbf, cnt, _ = oracle.render(scene, threads=16)
    check_image(rgb, rgbf, orgb, orgbf, "synthetic 0.001")
    assert (st.rays_primary, st.rays_secondary, st.rays_shadow) == (cnt.primary, cnt.secondary, cnt.shadow)


def test_synthetic_scene_lbvh_equals_brute_force(tracer):
    """1/100 scale (10^4 spheres + 10^5 triangles): every walk of the tree finds the brute-force hit."""
    from raytracer_amd.synth import synthetic_scene
    scene = synthetic_scene(scale=0.01, width=64, height=36, spp=1)
    tracer.load_scene(scene)
    o, d = _random_rays(scene, 20_000, np.random.default_rng(11))
    pb, db = tracer.query_rays(o, d, accel=ra.RT_ACCEL_BRUTE)
    assert (pb >= 0).mean() > 0.2
    for accel in TREE_WALKS:
        pl, dl = tracer.query_rays(o, d, accel=accel)
        assert np.array_equal(pb, pl), f"accel {accel}: {(pb != pl).sum()} closest hits differ"
        assert np.array_equal(bits(db), bits(dl))


def test_every_hierarchy_renders_the_same_image(tracer, monkeypatch):
    """The first render picks a hierarchy by timing; force each of the three in turn and compare.  Their surface-area
    costs are reported, and the one that rendered is the one that was asked for."""
    scene = load_scene("dragon").copy(width=192, height=128, spp=2)
    frames = {}
    for k, tree in enumerate(("karras", "ploc", "sah")):
        monkeypatch.setenv("RT_TREE", tree)
        tracer.load_scene(scene)
        ts = tracer.tree_stats()
        assert ts.in_use == -1 and ts.primitives == len(scene.tri_verts) + len(scene.spheres)
        assert all(ts.depth[t] > 0 and ts.node_visits[t] > 1.0 and ts.prim_tests[t] > 0.5 for t in range(3)), ts.as_dict()
        frames[tree] = tracer.render(want_float=True)
        assert tracer.tree_stats().in_use == k, tracer.tree_stats().as_dict()
    rgb_k, f_k, st_k = frames["karras"]
    for tree in ("ploc", "sah"):
        rgb_p, f_p, st_p = frames[tree]
        assert np.array_equal(rgb_k, rgb_p) and np.array_equal(bits(f_k), bits(f_p)), tree
        assert st_k.rays == st_p.rays


@pytest.mark.parametrize("name,kw", [("dragon", dict(width=256, height=160, spp=4)),
                                     ("plant_h_smooth", dict(width=165, height=128, spp=1, max_depth=8)),
                                     ("outdoor_spot", dict(width=200, height=120, spp=2)),
                                     ("dragon", dict(width=1280, height=720, spp=3))])
def test_every_walk_renders_the_same_image(tracer, monkeypatch, name, kw):
    """Per-lane walks everywhere, team walks everywhere, packet walks everywhere, brute force: one image, bit for bit.
    The last case is large enough (2.8 M primary rays: several chunks of two 64-ray groups per wave) for the per-lane
    walks to run with lane refill across chunk boundaries, the first generations with it, the last ones without."""
    scene = load_scene(name).copy(**kw)
    tracer.load_scene(scene)
    rgb0, f0, st0 = tracer.render(want_float=True)
    images = {}
    for label, n, wide in (("lane", "0", 0), ("lane-wide", "0", 1), ("team", "0", 2), ("packet-then-team", "1", 2), ("packet", "99", 0)):
        monkeypatch.setenv("RT_PACKET_GENS", n)
        monkeypatch.setenv("RT_PACKET_SHADOW_GENS", n)
        if wide:
            monkeypatch.setenv("RT_LANE_WIDE", str(wide))
        with ra.Tracer(device=0, accel=ra.RT_ACCEL_LBVH) as t:
            t.load_scene(scene)
            images[label] = t.render(want_float=True)
        monkeypatch.delenv("RT_LANE_WIDE", raising=False)
    monkeypatch.delenv("RT_PACKET_GENS")
    monkeypatch.delenv("RT_PACKET_SHADOW_GENS")
    with ra.Tracer(device=0, accel=ra.RT_ACCEL_BRUTE) as t:
        t.load_scene(scene)
        images["brute"] = t.render(want_float=True)
    for label, (rgb, f, st) in images.items():
        assert np.array_equal(rgb, rgb0), label
        assert np.array_equal(bits(f), bits(f0)), label
        assert st.rays == st0.rays, label


@pytest.mark.parametrize("name,kw", [("spheres1", dict(width=320, height=240, spp=1)),
                                     ("spheres2", dict(width=200, height=150, spp=4)),
                                     ("outdoor_spot", dict(width=240, height=135, spp=4)),
                                     ("spot_falloff", dict(width=160, height=120, spp=2)),
                                     ("test_reasonable", dict(width=97, height=65, spp=2)),
                                     ("bvhtest", dict(width=120, height=90, spp=2))])
def test_shadow_queries_inside_the_shade_launch_change_nothing(monkeypatch, name, kw):
    """Brute-force scenes that fit the shade kernel's LDS get no shadow launches: the shade launch answers the
    shadow queries itself.  Same floats, bytes and ray counts as with the shadow launches (RT_INLINE_SHADOWS=0)."""
    scene = load_scene(name).copy(**kw)
    out = {}
    for inline in ("1", "0"):
        monkeypatch.setenv("RT_INLINE_SHADOWS", inline)
        with ra.Tracer(accel=ra.RT_ACCEL_BRUTE) as t:
            t.load_scene(scene)
            out[inline] = t.render(want_float=True)
    (rgb1, f1, st1), (rgb0, f0, st0) = out["1"], out["0"]
    want = scene.max_depth + 1 if len(scene.lights) else 0          # (a scene without lights has no shadow launches either way)
    assert st1.launches_shadow == 0 and st0.launches_shadow == want, "the switch did not switch"
    assert np.array_equal(bits(f1), bits(f0)) and np.array_equal(rgb1, rgb0)
    assert (st1.rays_primary, st1.rays_secondary, st1.rays_shadow) == (st0.rays_primary, st0.rays_secondary, st0.rays_shadow)


@pytest.mark.parametrize("name,kw", [("dragon", dict(width=192, height=128, spp=2)), ("spheres2", dict(width=160, height=120, spp=2))])
def test_stage_times_are_optional_and_selective(name, kw):
    """rt_set_stage_times: 0 nothing, 1 every stage, 2 only the closest-hit launches, 3 only the shadow launches;
    the image never depends on it."""
    scene = load_scene(name).copy(**kw)
    with ra.Tracer() as t:
        t.load_scene(scene)
        frames = {}
        for mode in (0, 1, 2, 3, 0):
            t.set_stage_times(mode)
            rgb, f, st = t.render(want_float=True)
            frames.setdefault(mode, []).append((bits(f), st))
        ref = frames[0][0][0]
        for mode, runs in frames.items():
            for b, st in runs:
                assert np.array_equal(b, ref), mode
        st0, st1, st2, st3 = frames[0][0][1], frames[1][0][1], frames[2][0][1], frames[3][0][1]
        assert st0.ms_extend == 0 and st0.ms_shadow == 0 and st0.ms_shade == 0
        assert st1.ms_extend > 0 and st1.ms_shade > 0 and (st1.ms_shadow > 0) == (st1.launches_shadow > 0)
        assert st2.ms_extend > 0 and st2.ms_shadow == 0 and st2.ms_shade == 0 and st2.ms_other == 0
        # (a brute-force scene small enough for the shade launch to answer its own shadow queries has no shadow launches)
        assert st3.ms_extend == 0 and st3.ms_shade == 0 and st3.ms_other == 0 and (st3.ms_shadow > 0) == (st3.launches_shadow > 0)
        assert frames[0][1][1].ms_extend == 0


@pytest.mark.parametrize("name,kw", [("dragon", dict(width=256, height=160, spp=2)), ("plant_h_smooth", dict(width=165, height=128, spp=1, max_depth=8))])
def test_calibration_is_optional_and_does_not_change_the_image(name, kw):
    """rt_set_calibration(ctx, 0): no timing of hierarchies and walks before the first frame (one-frame callers);
    same image, same ray counts."""
    scene = load_scene(name).copy(**kw)
    out = []
    for on in (True, False):
        with ra.Tracer(accel=ra.RT_ACCEL_LBVH) as t:
            t.set_calibration(on)
            t.load_scene(scene)
            rgb, f, st = t.render(want_float=True)
            out.append((rgb, bits(f), st.rays))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]) and out[0][2] == out[1][2]


@pytest.mark.parametrize("name,kw,accel", [("spheres1", dict(width=640, height=480, spp=1), ra.RT_ACCEL_BRUTE),
                                            ("spheres2", dict(width=320, height=240, spp=2), ra.RT_ACCEL_LBVH),
                                            ("dragon", dict(width=320, height=200, spp=2), ra.RT_ACCEL_LBVH),
                                            ("plant_h_smooth", dict(width=330, height=256, spp=1, max_depth=8), ra.RT_ACCEL_LBVH)])
def test_batches_replayed_as_graphs_render_the_same_image(name, kw, accel):
    """rt_set_graphs: a thin batch is captured the second time its launches come and replayed from then on (both
    streams of the overlapped generations inside the graph); frames, float frames and ray counts equal those of the
    directly queued launches, also after a change of film (new launches, new graph) and back.  Batches that run on two
    streams (hierarchy scenes from the second frame on) become graphs only on HIP runtimes from 7.2 on: an older
    runtime frees the queue object of the graph's side stream under completion handlers that are still pending."""
    replayed = accel == ra.RT_ACCEL_BRUTE or hip_runtime_version() >= 70200000
    scene = load_scene(name).copy(**kw)
    other = scene.copy(width=kw["width"] // 2, height=kw["height"] // 2)
    with ra.Tracer(accel=accel) as t:
        t.set_graphs(False)
        t.load_scene(scene)
        rgb0, f0, st0 = t.render(want_float=True)
        t.load_scene(other)
        rgb1, f1, st1 = t.render(want_float=True)
        assert t.set_graphs() == 0
    with ra.Tracer(accel=accel) as t:
        t.load_scene(scene)
        for i in range(5):
            rgb, f, st = t.render(want_float=True)
            assert np.array_equal(rgb, rgb0) and np.array_equal(bits(f), bits(f0)) and st.rays == st0.rays, i
        n = t.set_graphs()
        assert (n >= 2) if replayed else (n == 0), "no batch ran as a graph launch"
        t.load_scene(other)
        for i in range(4):
            rgb, f, st = t.render(want_float=True)
            assert np.array_equal(rgb, rgb1) and np.array_equal(bits(f), bits(f1)) and st.rays == st1.rays, i
        assert (t.set_graphs() > n) if replayed else (t.set_graphs() == 0)
        t.load_scene(scene)
        for i in range(3):
            rgb, f, st = t.render(want_float=True)
            assert np.array_equal(rgb, rgb0) and np.array_equal(bits(f), bits(f0)) and st.rays == st0.rays, i
        # a frame with stage times is queued directly, and the graph is still good afterwards
        n = t.set_graphs()
        t.set_stage_times(1)
        rgb, f, st = t.render(want_float=True)
        assert np.array_equal(rgb, rgb0) and t.set_graphs() == n
        t.set_stage_times(0)
        rgb, f, st = t.render(want_float=True)
        assert np.array_equal(rgb, rgb0) and np.array_equal(bits(f), bits(f0))


def test_work_counters_are_optional_and_do_not_change_the_image(tracer):
    scene = load_scene("dragon").copy(width=128, height=96, spp=2)
    tracer.load_scene(scene)
    rgb0, f0, st0 = tracer.render(want_float=True)
    assert sum(st0.nodes_visited) == 0 and sum(st0.tri_tests) == 0        # off by default
    tracer.set_work_counters(True)
    try:
        rgb1, f1, st1 = tracer.render(want_float=True)
    finally:
        tracer.set_work_counters(False)
    assert np.array_equal(bits(f0), bits(f1)) and np.array_equal(rgb0, rgb1)
    assert st1.rays == st0.rays
    assert st1.nodes_visited[0] > 0 and st1.tri_tests[0] > 0 and st1.lane_node_tests[0] >= st1.nodes_visited[0]
    assert st1.nodes_visited[1] > 0                                        # shadow rays walk the tree too
    rgb2, f2, st2 = tracer.render(want_float=True)
    assert sum(st2.nodes_visited) == 0


def test_k3_full_size_frame_against_oracle_crops_and_across_partitions(oracle):
    """The bench configuration itself (dragon, 3840x2160, 16 spp, depth 5).  The oracle is brute force
    on the CPU and cannot render 133 M samples, so it renders crops of the full-resolution frame
    (same film, same sample indices) where the image is busiest and where it is empty; the size-
    independent properties cover the rest: the ray counts of the frame, and a frame put together
    from two GPUs' shards is byte-identical to the one rendered whole."""
    import torch
    scene = load_scene("dragon").copy(width=3840, height=2160, spp=16, seed=1)
    with ra.Tracer() as t:
        t.load_scene(scene)
        rgb, rgbf, st = t.render(want_float=True)
        assert (st.rays_primary, st.rays_secondary, st.rays_shadow) == (132710400, 109577458, 110625041)
        for roi in [(1900, 1000, 1932, 1024), (2400, 1300, 2432, 1324), (1300, 700, 1332, 716), (0, 0, 16, 8), (3828, 2150, 3840, 2160)]:
            orgb, orgbf, _, _ = oracle.render(scene, roi=roi, threads=16)
            check_image(crop(rgb, roi), crop(rgbf, roi), orgb, orgbf, f"K3 full size, crop {roi}")
        world = 2
        nbytes = ra.shard_bytes(scene.width, scene.height, world)
        shards = torch.zeros(world * nbytes, dtype=torch.uint8, device="cuda:0")
        whole = torch.empty(scene.height * scene.width * 3, dtype=torch.uint8, device="cuda:0")
        total = 0
        for rank in range(world):
            total += t.render_shard(rank, world, shards.data_ptr() + rank * nbytes).rays
        t.assemble_shards(shards.data_ptr(), world, whole.data_ptr())
        assert total == st.rays
        assert np.array_equal(whole.cpu().numpy().reshape(scene.height, scene.width, 3), rgb)


def test_k3_full_size_frame_by_brute_force_equals_the_hierarchy():
    """The bench frame itself (3840x2160, 16 spp, 353 M rays) rendered twice: with every ray tested against all
    10 002 primitives (the reference's `intersect`, five seconds on the GPU) and through the hierarchy.  All
    8.3 M pixels must have the same float bits and the ray counts must agree."""
    scene = load_scene("dragon").copy(width=3840, height=2160, spp=16, seed=1)
    frames = {}
    for label, accel in (("hierarchy", ra.RT_ACCEL_LBVH), ("brute force", ra.RT_ACCEL_BRUTE)):
        with ra.Tracer(accel=accel) as t:
            t.load_scene(scene)
            frames[label] = t.render(want_float=True)
    (rgb_h, f_h, st_h), (rgb_b, f_b, st_b) = frames["hierarchy"], frames["brute force"]
    differ = int((bits(f_h) != bits(f_b)).any(axis=2).sum())
    REPORT["K3 full frame, hierarchy against brute force on the GPU"] = {"pixels": int(f_h.shape[0] * f_h.shape[1]),
                                                                         "float_bit_identical": 1.0 - differ / (f_h.shape[0] * f_h.shape[1])}
    assert differ == 0, f"{differ} pixels differ in float"
    assert np.array_equal(rgb_h, rgb_b)
    assert (st_h.rays_primary, st_h.rays_secondary, st_h.rays_shadow) == (st_b.rays_primary, st_b.rays_secondary, st_b.rays_shadow)


class TestK5FullScale:
    """The synthetic scene at full scale (10^6 spheres + 10^7 triangles) on the full 7680x4320 film, 2 spp: rendered
    once, compared crop by crop."""

    @pytest.fixture(scope="class")
    def k5(self):
        from raytracer_amd.synth import synthetic_scene
        scene = synthetic_scene(scale=1.0).copy(spp=2)
        assert (scene.width, scene.height, len(scene.spheres), len(scene.tri_verts)) == (7680, 4320, 1000000, 10000000)
        t = ra.Tracer()
        t.load_scene(scene)
        rgb, rgbf, st = t.render(want_float=True)
        assert st.accel_used == ra.RT_ACCEL_LBVH and st.rays_shadow > 0 and st.rays_secondary > 0
        yield scene, t, rgb, rgbf
        t.close()

    @pytest.mark.parametrize("x0,y0,shard", [(3840, 2160, True), (3012, 1796, True), (4700, 2500, False), (2200, 2900, False),
                                             (2800, 1500, False), (4100, 3300, False), (3600, 1300, False), (4400, 3000, False)])
    def test_k5_full_scene_and_film_against_oracle_crops(self, oracle, k5, x0, y0, shard):
        """The oracle tests every primitive for every ray (half a second per ray), so each case compares one 4x4
        crop -- centre and off-centre, all on geometry; the last two hold ray trees five levels deep (843 and 2 065
        rays for their 32 samples), the others 128 rays -- in float (eight cases, one crop each, so that a run
        reports progress every minute); for two of them the crop's tile is also rendered as part of a shard of
        16, which must give the same bytes."""
        import torch
        from raytracer_amd import dist as rd
        scene, t, rgb, rgbf = k5
        world = 16
        roi = (x0, y0, x0 + 4, y0 + 4)
        orgb, orgbf, cnt, _ = oracle.render(scene, roi=roi, threads=16)
        assert cnt.shadow > 0, "the crop was meant to see geometry"
        check_image(crop(rgb, roi), crop(rgbf, roi), orgb, orgbf, f"K5 full scale 7680x4320x2, crop {roi}")
        if shard:
            buf = torch.zeros(ra.shard_bytes(scene.width, scene.height, world), dtype=torch.uint8, device="cuda:0")
            tx, ty = x0 // ra.RT_TILE, y0 // ra.RT_TILE
            rank = next(r for r in range(world) if (tx, ty) in rd.tile_list(scene.width, scene.height, r, world))
            t.render_shard(rank, world, buf.data_ptr())
            k = rd.tile_list(scene.width, scene.height, rank, world).index((tx, ty))
            tile = buf.cpu().numpy()[k * ra.RT_TILE * ra.RT_TILE * 3:(k + 1) * ra.RT_TILE * ra.RT_TILE * 3].reshape(ra.RT_TILE, ra.RT_TILE, 3)
            assert np.array_equal(tile, rgb[ty * ra.RT_TILE:(ty + 1) * ra.RT_TILE, tx * ra.RT_TILE:(tx + 1) * ra.RT_TILE]), (x0, y0)


def test_k5_full_scale_camera_and_surface_rays_every_walk_equals_brute_force():
    """The 11 M-primitive scene, rays as a render sends them: 40 000 primary rays of random pixels (the eye is
    hundreds of units from every sphere, where the reference's float sphere test accepts rays that pass a sphere
    of radius 0.05 several radii from its centre -- the leaf boxes have to hold those) and the rays that leave
    their hit points towards random primitives.  Brute force on the GPU (11 M exact tests per ray) is the
    reference's answer; every walk of both hierarchies must return the same primitive and distance bits."""
    from raytracer_amd.synth import synthetic_scene
    scene = synthetic_scene(scale=1.0).copy(spp=1)
    rng = np.random.default_rng(23)
    n = 40_000
    with ra.Tracer(accel=ra.RT_ACCEL_LBVH) as t:
        t.load_scene(scene)
        o, d = t.primary_rays(rng.integers(0, scene.width, n), rng.integers(0, scene.height, n), np.zeros(n, np.int32))
        pb, db = t.query_rays(o, d, accel=ra.RT_ACCEL_BRUTE)
        assert (pb >= 0).mean() > 0.5 and (pb[pb >= 0] < len(scene.spheres)).any()
        hit = pb >= 0
        tpar = np.sqrt(db[hit].astype(np.float64) / (d[hit].astype(np.float64) ** 2).sum(1))
        o2 = (o[hit] + d[hit] * tpar[:, None]).astype(np.float32)
        tgt = scene.spheres[rng.integers(0, len(scene.spheres), len(o2)), :3]
        d2 = (tgt - o2 + rng.normal(0, 0.3, o2.shape)).astype(np.float32)
        pb2, db2 = t.query_rays(o2, d2, accel=ra.RT_ACCEL_BRUTE)
        for accel in walks_held(t):
            pl, dl = t.query_rays(o, d, accel=accel)
            assert np.array_equal(pb, pl), f"walk {accel}, camera rays: {(pb != pl).sum()} of {n} closest hits differ"
            assert np.array_equal(bits(db[hit]), bits(dl[hit]))
            pl2, dl2 = t.query_rays(o2, d2, accel=accel)
            assert np.array_equal(pb2, pl2), f"walk {accel}, surface rays: {(pb2 != pl2).sum()} of {len(o2)} closest hits differ"
            assert np.array_equal(bits(db2[pb2 >= 0]), bits(dl2[pb2 >= 0]))


def test_coincident_primitives_and_tie_rules(oracle, monkeypatch):
    """Thousands of primitives at the same place: every box overlaps every other (the second hierarchy
    degenerates and is rejected), and every ray meets many primitives at exactly the same distance, where
    the reference keeps the lowest index of a list and prefers a sphere to an equally distant triangle
    (src/raytrace.cpp:172, :250, :361).  Each copy has its own material, so a wrong winner changes the colour."""
    rng = np.random.default_rng(3)
    n = 1500
    mats = np.zeros((2 * n, 14), np.float32)
    mats[:, 0:3] = rng.random((2 * n, 3)); mats[:, 3:6] = rng.random((2 * n, 3)); mats[:, 9] = 5; mats[:, 13] = 1
    mats[::7, 6:9] = 0.5                                    # some mirrors among them
    tri = np.tile(np.array([[-2, -1, 6, 2, -1, 6, 0, 2.5, 6]], np.float32), (n, 1))
    nrm = np.tile(np.array([[0, 0, -1] * 3], np.float32), (n, 1))
    sph = np.tile(np.array([[0.8, 0.3, 5.5, 0.9]], np.float32), (n, 1))
    sph[n // 2:, 0] = -1.0                                   # two piles of spheres, one pile of triangles
    lights = np.array([[2, 40, 40, 40, 3, 4, 0, 0, 0, 0, 0, 0], [1, .5, .5, .5, 0, 0, 0, .2, -1, .3, 0, 0]], np.float32)
    scene = ra.SceneArrays(width=96, height=72, spp=2, seed=5, spheres=sph, sphere_mat=np.arange(n, dtype=np.int32),
                           tri_verts=tri, tri_normals=nrm, tri_flags=np.zeros(n, np.int32),
                           tri_mat=np.arange(n, 2 * n, dtype=np.int32), materials=mats, lights=lights,
                           ambient=np.array([.1, .1, .1], np.float32), background=np.array([.2, .3, .4], np.float32))
    orgb, orgbf, cnt, _ = oracle.render(scene, threads=16)
    images = {}
    for label, env in [("brute", None), ("packet", {"RT_PACKET_GENS": "64", "RT_PACKET_SHADOW_GENS": "64"}),
                       ("lane", {"RT_PACKET_GENS": "0", "RT_PACKET_SHADOW_GENS": "0"}), ("calibrated", {})]:
        for k in ("RT_PACKET_GENS", "RT_PACKET_SHADOW_GENS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in (env or {}).items():
            monkeypatch.setenv(k, v)
        with ra.Tracer(accel=ra.RT_ACCEL_BRUTE if env is None else ra.RT_ACCEL_LBVH) as t:
            t.load_scene(scene)
            rgb, rgbf, st = t.render(want_float=True)
        check_image(rgb, rgbf, orgb, orgbf, f"coincident primitives, {label}")
        assert (st.rays_primary, st.rays_secondary, st.rays_shadow) == (cnt.primary, cnt.secondary, cnt.shadow), label


def _deep_scene(ties=2048):
    # Morton keys 0 (a pile of `ties` coincident spheres), 1, 2, 4, ... 2^62 (one tiny sphere each) and the
    # far corner: the hierarchy is a chain of 63 levels with the pile's index tie-break levels below it
    pts = [(0.0, 0.0, 0.0)] * ties
    for j in range(21):
        v = float(2.0 ** j) / float(2 ** 21)
        pts += [(v, 0, 0), (0, v, 0), (0, 0, v)]
    pts.append((1.0, 1.0, 1.0))
    pts = np.array(pts, np.float32) * 8.0
    n = len(pts)
    sph = np.concatenate([pts, np.full((n, 1), 1e-7, np.float32)], 1)
    sph[:ties, 3] = 0.6
    sph[-1, 3] = 0.8
    mats = np.zeros((n, 14), np.float32)
    mats[:, 0:3] = np.random.default_rng(1).random((n, 3)); mats[:, 3:6] = 0.5; mats[:, 6:9] = 0.3; mats[:, 9] = 5; mats[:, 13] = 1
    return ra.SceneArrays(width=80, height=60, spp=2, seed=9, spheres=sph, sphere_mat=np.arange(n, dtype=np.int32),
                          materials=mats, cam_pos=np.array([0.5, 0.4, -2.5], np.float32),
                          ambient=np.array([.3, .3, .3], np.float32), background=np.array([.1, .2, .3], np.float32),
                          lights=np.array([[2, 30, 30, 30, 1, 2, -3, 0, 0, 0, 0, 0]], np.float32))


def test_hierarchy_deeper_than_64_levels(oracle, monkeypatch):
    """A tree deeper than 64 takes the fall-backs: the packet walk leaves its one-register stack for the
    128-entry one, no 4-wide collapse exists.  Every walk must still agree with brute force and the oracle."""
    scene = _deep_scene()
    orgb, orgbf, cnt, _ = oracle.render(scene, threads=16)
    assert cnt.secondary > 0 and cnt.shadow > 0
    rng = np.random.default_rng(11)
    o, d = _random_rays(scene, 4096, rng)
    for label, env in [("brute", None), ("packet", {"RT_PACKET_GENS": "64", "RT_PACKET_SHADOW_GENS": "64"}),
                       ("lane", {"RT_PACKET_GENS": "0", "RT_PACKET_SHADOW_GENS": "0"}), ("calibrated", {})]:
        for k in ("RT_PACKET_GENS", "RT_PACKET_SHADOW_GENS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in (env or {}).items():
            monkeypatch.setenv(k, v)
        with ra.Tracer(accel=ra.RT_ACCEL_BRUTE if env is None else ra.RT_ACCEL_LBVH) as t:
            t.load_scene(scene)
            if label == "calibrated":
                ref_prim, ref_d2 = t.query_rays(o, d, accel=ra.RT_ACCEL_BRUTE)
                for accel in (ra.RT_ACCEL_LBVH, ra.RT_ACCEL_LBVH_PACKET):
                    prim, d2 = t.query_rays(o, d, accel=accel)
                    assert np.array_equal(prim, ref_prim) and np.array_equal(bits(d2[prim >= 0]), bits(ref_d2[ref_prim >= 0]))
                with pytest.raises(ra.RtError):                  # no 4-wide collapse of a tree this deep
                    t.query_rays(o, d, accel=ra.RT_ACCEL_LBVH_WIDE)
            rgb, rgbf, st = t.render(want_float=True)
        check_image(rgb, rgbf, orgb, orgbf, f"deep hierarchy, {label}")
        assert (st.rays_primary, st.rays_secondary, st.rays_shadow) == (cnt.primary, cnt.secondary, cnt.shadow), label


@pytest.mark.parametrize("w,h,spp", [(1, 1, 1), (37, 19, 3), (5, 70, 64)])
def test_many_lights_degenerate_triangles_and_odd_films(oracle, tracer, w, h, spp):
    """Nine lights of every kind (one work item per hit and light in the shadow kernel), triangles of zero
    area (detA == 0: the literal division path of the triangle test, NaN comparisons as the reference makes
    them), a film that is not a multiple of the 32x32 tile, and more samples per pixel than pixels."""
    base = load_scene("test_reasonable")
    rng = np.random.default_rng(21)
    lights = []
    for k in range(9):
        kind = 1 + k % 3                                      # directional, point, spot
        col = rng.uniform(2, 30, 3) if kind != 1 else rng.uniform(0.1, 0.6, 3)
        pos = rng.uniform(-6, 6, 3); dr = rng.normal(size=3)
        lights.append([kind, *col, *pos, *dr, 20 + 5 * k, 50 + 3 * k])
    tv = base.tri_verts.copy(); tn = base.tri_normals.copy(); tf = base.tri_flags.copy(); tm = base.tri_mat.copy()
    if len(tv) == 0:
        tv = np.zeros((0, 9), np.float32); tn = np.zeros((0, 9), np.float32)
    degenerate = np.array([[0, 0, 4, 0, 0, 4, 1, 1, 4], [1, 0, 3, 2, 0, 3, 3, 0, 3], [0, 0, 5, 0, 0, 5, 0, 0, 5]], np.float32)
    tv = np.concatenate([tv, degenerate]); tn = np.concatenate([tn, np.tile([[0, 0, -1] * 3], (3, 1)).astype(np.float32)])
    tf = np.concatenate([tf, np.array([0, 1, 0], np.int32)]); tm = np.concatenate([tm, np.zeros(3, np.int32)])
    scene = base.copy(width=w, height=h, spp=spp, seed=4, lights=np.array(lights, np.float32),
                      tri_verts=tv, tri_normals=tn, tri_flags=tf, tri_mat=tm)
    orgb, orgbf, cnt, _ = oracle.render(scene, threads=16)
    for accel in (ra.RT_ACCEL_BRUTE, ra.RT_ACCEL_LBVH):
        with ra.Tracer(accel=accel) as t:
            t.load_scene(scene)
            rgb, rgbf, st = t.render(want_float=True)
        check_image(rgb, rgbf, orgb, orgbf, f"{w}x{h}x{spp} accel {accel}")
        assert (st.rays_primary, st.rays_secondary, st.rays_shadow) == (cnt.primary, cnt.secondary, cnt.shadow)


@pytest.mark.parametrize("name,kw", [("dragon", dict(width=960, height=540, spp=4)), ("spheres2", dict(width=640, height=480, spp=2))])
def test_repeated_frames_are_byte_identical(tracer, name, kw):
    """Colours travel up the ray tree from whichever wave finishes a ray; every slot has one writer per
    frame, so twenty frames in a row must not differ in a single byte (a race would flicker)."""
    import torch
    scene = load_scene(name).copy(seed=2, **kw)
    tracer.load_scene(scene)
    first = torch.empty(scene.width * scene.height * 3, dtype=torch.uint8, device="cuda:0")
    again = torch.empty_like(first)
    tracer.render_device(first.data_ptr())
    for _ in range(20):
        again.zero_()
        tracer.render_device(again.data_ptr())
        assert bool((first == again).all())


def test_host_buffer_render_equals_the_frame_in_device_memory(monkeypatch):
    """rt_render brings a frame of 8 MiB and more to the caller's buffer through pinned staging chunks and host threads,
    a smaller one (or RT_FAST_COPY=0) by one plain copy: the bytes are those rt_render_device leaves in device memory,
    for the 8-bit and the float frame, also when a small frame follows a large one on the same context."""
    import torch
    big = load_scene("spheres1").copy(width=2048, height=1500, spp=1)          # 9.2 MB of RGB8, 36.9 MB of floats
    small = big.copy(width=640, height=480)
    frames = {}
    for fast in ("1", "0"):
        monkeypatch.setenv("RT_FAST_COPY", fast)
        with ra.Tracer(accel=ra.RT_ACCEL_BRUTE) as t:
            for scene in (big, small, big):
                t.load_scene(scene)
                rgb, f, st = t.render(want_float=True)
                dev = torch.zeros((scene.height, scene.width, 3), dtype=torch.uint8, device="cuda:0")
                t.render_device(dev.data_ptr())
                assert np.array_equal(rgb, dev.cpu().numpy()), (fast, scene.width)
                rgb2, _, _ = t.render()
                assert np.array_equal(rgb, rgb2)
                key = (scene.width, scene.height)
                if key in frames:
                    assert np.array_equal(frames[key][0], rgb) and np.array_equal(bits(frames[key][1]), bits(f)), (fast, key)
                frames[key] = (rgb, f)


def test_render_on_a_caller_provided_stream(tracer):
    """rt_options.stream: every launch of the context goes to the caller's HIP stream (here a torch stream),
    so work the caller queued on it before the call is seen by the render, and the result is the same image."""
    import torch
    scene = load_scene("spheres2").copy(width=320, height=240, spp=2)
    tracer.load_scene(scene)
    want, _, _ = tracer.render()
    stream = torch.cuda.Stream(device="cuda:0")
    with ra.Tracer(stream=stream.cuda_stream) as t:
        t.load_scene(scene)
        with torch.cuda.stream(stream):
            out = torch.full((scene.height, scene.width, 3), 7, dtype=torch.uint8, device="cuda:0")   # queued on the same stream
            st = t.render_device(out.data_ptr())
            got = out.cpu().numpy()
        stream.synchronize()
    assert np.array_equal(got, want) and st.rays > 0


@pytest.mark.parametrize("name,kw", [("dragon", dict(width=640, height=360, spp=4)),
                                     ("spheres2", dict(width=480, height=360, spp=3)),          # reflection AND refraction: resolve passes
                                     ("plant_h_smooth", dict(width=330, height=256, spp=2, max_depth=8))])
def test_two_stream_generations_render_the_same_image(monkeypatch, name, kw):
    """Thin generations run split over two streams (shadow + lighting of generation g beside the traversal of
    g + 1).  Whatever the split -- none, from the first generation on, from a later one -- the frame is the same."""
    scene = load_scene(name).copy(seed=3, **kw)
    images = []
    for threshold in ("0", str(1 << 40), "20000"):
        monkeypatch.setenv("RT_OVERLAP_RAYS", threshold)
        with ra.Tracer(accel=ra.RT_ACCEL_LBVH) as t:
            t.load_scene(scene)
            for _ in range(3):                       # the split follows the ray counts of the frame before
                rgb, rgbf, st = t.render(want_float=True)
            images.append((rgb, rgbf, st))
    for rgb, rgbf, st in images[1:]:
        assert np.array_equal(rgb, images[0][0]) and np.array_equal(bits(rgbf), bits(images[0][1]))
        assert st.rays == images[0][2].rays


@pytest.mark.parametrize("w,h,spp", [(60000, 50000, 4), (60000, 50000, 3), (4099, 2053, 64), (3840, 2160, 16), (641, 479, 5)])
def test_primary_rays_for_every_counter_and_sample_rate_shortcut(oracle, tracer, w, h, spp):
    """Sample index -> pixel and the jitter draw take shortcuts when sample_rate is a power of two and when
    the jitter counter of the whole film fits 32 bits; the first two films here need 64 bits.  All must give
    the oracle's rays bit for bit, also for the last pixels of the film (the largest counters)."""
    scene = load_scene("spheres1").copy(width=w, height=h, spp=spp, seed=77)
    tracer.load_scene(scene)
    rng = np.random.default_rng(w + spp)
    xs = np.concatenate([rng.integers(0, w, 60), [0, w - 1, w - 1]]).astype(np.int32)
    ys = np.concatenate([rng.integers(0, h, 60), [0, h - 1, h - 2]]).astype(np.int32)
    js = np.concatenate([rng.integers(0, spp, 60), [0, spp - 1, 0]]).astype(np.int32)
    o, d = tracer.primary_rays(xs, ys, js)
    for k in range(len(xs)):
        oo, od = oracle.get_ray(scene, int(xs[k]), int(ys[k]), int(js[k]))
        assert np.array_equal(bits(o[k]), bits(oo)) and np.array_equal(bits(d[k]), bits(od)), (xs[k], ys[k], js[k])


def test_negative_max_depth_returns_the_background_untraced(oracle, tracer):
    # src/raytrace.cpp:132-135: depth 0 > max_depth, so no ray is traced at all
    s = load_scene("spheres1").copy(width=50, height=40, spp=3, max_depth=-1)
    tracer.load_scene(s)
    rgb, rgbf, st = tracer.render(want_float=True)
    orgb, orgbf, cnt, _ = oracle.render(s, threads=2)
    assert np.array_equal(rgb, orgb) and np.array_equal(bits(rgbf), bits(orgbf))
    assert st.rays_secondary == 0 and st.rays_shadow == 0 and st.generations == 0


def test_descriptor_with_a_count_but_no_array_is_rejected():
    import ctypes as C
    from raytracer_amd import api
    d, keep = load_scene("spheres1").desc()
    d.spheres = None
    with ra.Tracer() as t:
        rc = t._lib.rt_load_scene(t._ctx, C.byref(d))
        assert rc == api.RT_ERR_INVALID and b"NULL" in t._lib.rt_last_error()
        with pytest.raises(ra.RtError):          # the failed load left no scene behind
            t.render()


def test_edge_cases(tracer):
    # empty scene: background everywhere, no secondary or shadow rays
    s = ra.SceneArrays(width=70, height=33, background=np.array([.25, .5, 1], np.float32))
    tracer.load_scene(s)
    rgb, _, st = tracer.render()
    assert (rgb == np.array([63, 127, 255], np.uint8)).all() and st.rays == 70 * 33
    # one sphere, no light, NaN-producing zero direction camera is not used; depth 0 cuts all recursion
    s = load_scene("spheres1").copy(width=64, height=48, max_depth=0)
    tracer.load_scene(s)
    _, _, st = tracer.render()
    assert st.rays_secondary == 0 and st.generations == 1
    # errors, not crashes
    with ra.Tracer() as t:
        with pytest.raises(ra.RtError):
            t.render()
        bad = load_scene("spheres1").copy(max_depth=200)
        with pytest.raises(ra.RtError):
            t.load_scene(bad)
