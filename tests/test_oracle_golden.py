"""CPU suite: the oracle (oracle/rt_oracle.c) against the committed golden vectors that
tools/make_fixtures.py generated from the unmodified reference."""
import numpy as np
import pytest

from conftest import bits, golden_cases, load_case, load_scene, same_float_image


@pytest.mark.parametrize("case", golden_cases())
def test_oracle_reproduces_reference_render(oracle, case):
    scene, roi, gold = load_case(case)
    rgb, rgbf, cnt, _ = oracle.render(scene, roi=roi, threads=8)
    assert same_float_image(rgbf, gold["rgbf"]).all(), "float colour differs from the reference render"
    assert np.array_equal(rgb, gold["rgb"])
    assert [cnt.primary, cnt.secondary, cnt.shadow] == [int(v) for v in gold["counts"]]


@pytest.mark.parametrize("key,scene_name,spp", [("spheres1_spp1", "spheres1", 1), ("spheres1_spp5", "spheres1", 5),
                                                ("outdoor_spp3", "outdoor", 3), ("dragon_spp1", "dragon", 1)])
def test_oracle_function_level_vectors(oracle, kat, key, scene_name, spp):
    s = load_scene(scene_name).copy(spp=spp)
    xyj = kat[f"ray_{key}_xyj"]
    for i, (x, y, j) in enumerate(xyj):
        o, d = oracle.get_ray(s, int(x), int(y), int(j))
        assert np.array_equal(bits(o), bits(kat[f"ray_{key}_o"][i]))
        assert np.array_equal(bits(d), bits(kat[f"ray_{key}_d"][i]))
        kind, out = oracle.intersect(s, o, d)
        ref = kat[f"hit_{key}"][i]
        assert (kind != 0) == (ref[0] != 0)
        if kind:
            # normal3, dist2, point3 (the reference's 8th slot is a material probe, not compared)
            assert np.array_equal(out[:7], ref[1:8]) or np.array_equal(np.isnan(out[:7]), np.isnan(ref[1:8]))
        col = oracle.trace_ray(s, o, d, 0)
        assert same_float_image(col, kat[f"col_{key}"][i]).all()


def test_quantiser_vectors(oracle, kat):
    for q, expect in zip(kat["quant_in"], kat["quant_out"]):
        assert np.array_equal(oracle.quantise(q), expect), q


def test_jitter_sequence_is_frozen(oracle, kat):
    for seed, key in ((1, "jitter_seed1"), (99, "jitter_seed99")):
        got = [oracle.jitter_hash(seed, int(c)) for c in kat["jitter_counter"]]
        assert got == [int(v) for v in kat[key]]
        assert all(0 <= g <= 2**31 - 1 for g in got)


def test_spheres1_full_frame_ray_count(oracle):
    """2 380 089 rays at 640x480 -- the structural pin of the ray tree (reflection before
    refraction, shadow ray per light per hit, cut at depth > max_depth)."""
    s = load_scene("spheres1")
    rgb, _, cnt, _ = oracle.render(s, threads=8, want_float=False)
    assert (cnt.primary, cnt.secondary, cnt.shadow) == (307200, 1267376, 805513)
    assert tuple(rgb[0, 0]) == (12, 12, 12)


def test_spot_falloff_known_answers_and_branch_coverage(oracle):
    """Spot lights whose angles lie in the range of acos() (src/raytrace.cpp:549-566): colours of single
    getColor calls (evaluateRayTree at depth == max_depth, whose children return the background untraced) and
    of the full recursion along 96 primary rays, as the reference computed them; and the scene makes all
    three branches of the fall-off run."""
    import os
    from conftest import GOLD
    z = np.load(os.path.join(GOLD, "kat_spot.npz"), allow_pickle=False)
    s = load_scene("spot_falloff")
    oracle.spot_branch_counts(reset=True)
    for i in range(len(z["o"])):
        assert same_float_image(oracle.trace_ray(s, z["o"][i], z["d"][i], s.max_depth), z["col_leaf"][i]).all(), i
        assert same_float_image(oracle.trace_ray(s, z["o"][i], z["d"][i], 0), z["col_full"][i]).all(), i
    below_a1, above_a2, between = oracle.spot_branch_counts()
    assert min(below_a1, above_a2, between) >= 40, (below_a1, above_a2, between)
    oracle.render(s, threads=4)
    br = oracle.spot_branch_counts()
    assert min(br) > 0.15 * sum(br), br
