"""GPU suite (-m gpu): differential test of the HIP path against the CPU oracle on seeded random scenes
(tests/_fuzz_scenes.py), every walk of both hierarchies forced in turn.  The seeds include the ones on which
`tools/fuzz_parity.py` (the same loop over thousands of seeds) has found a difference: tiny spheres seen from
far away, which the reference's float sphere test "hits" well outside their radius."""
import os

import numpy as np
import pytest

import raytracer_amd as ra
from _fuzz_scenes import WALKS, edge_grazing_case, random_scene
from conftest import bits, same_float_image

pytestmark = pytest.mark.gpu

ENV_KEYS = ("RT_PACKET_GENS", "RT_PACKET_SHADOW_GENS", "RT_LANE_WIDE", "RT_TREE", "RT_MAX_BLOCKS")
FOUND = (79, 2989, 11138, 13308)          # seeds of earlier findings


def _same(rgbf, orgbf, what):
    """Per-pixel bit equality of two host arrays, looked at ONCE: a first-read difference fails its test."""
    return same_float_image(rgbf, orgbf).all(axis=2)


def _render(monkeypatch, scene, accel, env, frames=1):
    """frames rendered on one context: from the second on the ray counts are known and thin generations of a
    hierarchy run on two streams"""
    for k in ENV_KEYS:
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    with ra.Tracer(device=0, accel=accel) as t:
        t.load_scene(scene)
        out = [t.render(want_float=True) for _ in range(frames)]
    for k in ENV_KEYS:
        monkeypatch.delenv(k, raising=False)
    return out


@pytest.mark.parametrize("seed", list(FOUND) + list(range(16)) + list(range(20000, 20016)))
def test_random_scene_every_walk_against_the_oracle(oracle, monkeypatch, seed):
    scene = random_scene(seed)
    orgb, orgbf, cnt, _ = oracle.render(scene, threads=16)
    variants = [("brute", ra.RT_ACCEL_BRUTE, {}), ("calibrated", ra.RT_ACCEL_LBVH, {})]
    for walk, env in WALKS.items():
        for tree in ("karras", "ploc", "sah"):
            variants.append((f"{walk}/{tree}", ra.RT_ACCEL_LBVH, dict(env, RT_TREE=tree)))
    for label, accel, env in variants:
        if accel == ra.RT_ACCEL_LBVH and len(scene.spheres) + len(scene.tri_verts) < 2:
            continue                                   # a single primitive has no hierarchy
        for k, (rgb, rgbf, st) in enumerate(_render(monkeypatch, scene, accel, env, frames=2)):
            what = f"seed {seed}, {label}, frame {k}"
            same = _same(rgbf, orgbf, what)
            assert same.all(), f"{what}: {int((~same).sum())} of {same.size} pixels differ in float"
            assert _same(rgb, orgb, what + ", RGB8").all(), what
            assert (st.rays_primary, st.rays_secondary, st.rays_shadow) == (cnt.primary, cnt.secondary, cnt.shadow), what


def test_tiny_spheres_far_from_the_eye_are_hit_where_the_reference_hits_them(oracle, monkeypatch):
    """The reference's sphere test (src/raytrace.cpp:232-241) rounds |e|^2 - r^2 to float: seen from a distance of
    |e| a sphere is hit by rays that pass its centre within about sqrt(r^2 + 1e-6 |e|^2), here several times the
    radius.  A hierarchy whose boxes hold only the ideal sphere culls those hits; brute force does not."""
    rng = np.random.default_rng(17)
    n = 3000
    sph = np.zeros((n, 4), np.float32)
    sph[:, 0:3] = rng.normal(size=(n, 3)) * np.array([2.5, 1.5, 2.5]) + np.array([0, 0, 25.0])
    sph[:, 3] = 10.0 ** rng.uniform(-3.0, -1.7, n)
    mats = np.zeros((6, 14), np.float32)
    mats[:, 0:3] = rng.random((6, 3)); mats[:, 3:6] = rng.random((6, 3)); mats[:, 9] = 10; mats[:, 13] = 1
    mats[0, 6:9] = 0.6                                                     # one mirror: the noise hits send out rays too
    scene = ra.SceneArrays(width=320, height=200, spp=2, seed=3, spheres=sph, sphere_mat=rng.integers(0, 6, n).astype(np.int32),
                           materials=mats, cam_half_angle=0.12, ambient=np.array([.5, .5, .5], np.float32),
                           background=np.array([.0, .0, .1], np.float32),
                           lights=np.array([[2, 300, 300, 300, 5, 8, 0, 0, 0, 0, 0, 0]], np.float32)).normalise()
    orgb, orgbf, cnt, _ = oracle.render(scene, threads=16)
    hit = (orgbf != scene.background).any(axis=2)
    assert hit.mean() > 0.01, "the scene is meant to show spheres"
    for label, accel, env in [("brute", ra.RT_ACCEL_BRUTE, {})] + [(w, ra.RT_ACCEL_LBVH, e) for w, e in WALKS.items()]:
        rgb, rgbf, st = _render(monkeypatch, scene, accel, env)[0]
        same = _same(rgbf, orgbf, label)
        assert same.all(), f"{label}: {int((~same).sum())} of {same.size} pixels differ in float"
        assert (st.rays_primary, st.rays_secondary, st.rays_shadow) == (cnt.primary, cnt.secondary, cnt.shadow), label


@pytest.mark.parametrize("trial", [1, 2, 4, 5, 9])
def test_rays_grazing_triangle_edges_from_far_origins(trial):
    """Rays aimed within a few float roundings of triangle edges from up to thousands of units away: whether the
    reference's float determinants (src/raytrace.cpp:314-334) call them hits is rounding, brute force on the GPU
    follows the reference, and a leaf box may cull none of them.  (Boxes padded by 1e-4 of the coordinates alone lose
    hits on trials 1 and 9: triangles in a coordinate plane seen from 500 units.)"""
    scene, o, d = edge_grazing_case(trial)
    with ra.Tracer(accel=ra.RT_ACCEL_LBVH) as t:
        t.load_scene(scene)
        pb, db = t.query_rays(o, d, accel=ra.RT_ACCEL_BRUTE)
        assert 0.2 < (pb >= 0).mean() < 0.8, "the rays are meant to sit on the edge between hit and miss"
        for walk in (ra.RT_ACCEL_LBVH, ra.RT_ACCEL_LBVH_PACKET, ra.RT_ACCEL_LBVH_WIDE, ra.RT_ACCEL_LBVH_TEAM,
                     ra.RT_ACCEL_LBVH | ra.RT_ACCEL_ALT_TREE, ra.RT_ACCEL_LBVH_WIDE | ra.RT_ACCEL_ALT_TREE,
                     ra.RT_ACCEL_LBVH_PACKET | ra.RT_ACCEL_SAH_TREE, ra.RT_ACCEL_LBVH_WIDE | ra.RT_ACCEL_SAH_TREE):
            pl, dl = t.query_rays(o, d, accel=walk)
            assert np.array_equal(pb, pl), f"walk {walk}: {(pb != pl).sum()} of {len(pb)} closest hits differ"
            assert np.array_equal(bits(db[pb >= 0]), bits(dl[pb >= 0]))
