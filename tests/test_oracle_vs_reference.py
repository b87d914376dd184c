"""CPU suite, build container only: the oracle and the product's parser against the
unmodified reference (oracle/_ref/libref_oracle.so built from /root/reference/src).
Skipped where that library or the reference tree is absent."""
import hashlib
import os

import numpy as np
import pytest

import raytracer_amd as ra
from _oracle import REFERENCE_TREE, Reference, have_reference
from conftest import bits, same_float_image

pytestmark = pytest.mark.skipif(not (have_reference() and os.path.isdir(REFERENCE_TREE)),
                                reason="reference build / tree not present")

# sha256 of the BMP the serial brute-force reference writes at 640x480 (first 16 hex digits)
PINS = {
    "SphereExamples/spheres1.scn": "09e1d34c5d168373",
    "TriangleExamples/outdoor.scn": "d06559d640854d74",
    "SphereExamples/spot_sphere.scn": "8266005ef7dbf0c5",
    "SphereExamples/ambient_sphere.scn": "4995cf7734a7ef30",
    "TriangleExamples/triangle.scn": "5f05c725865b6604",
    "BVHTest/test.scn": "95f0cfdd034b5122",
    "TriangleExamples/test_reasonable.scn": "26c88c9542d41c9d",
}
ALL_SCENES = list(PINS) + ["SphereExamples/spheres2.scn", "ComplexExamples/dragon.scn", "ComplexExamples/plant-h.scn",
                           "ComplexExamples/test.scn", "PlaneExamples/test.scn", "PlaneExamples/rect.scn", "src/sphere.scn"]


@pytest.fixture(scope="module")
def ref():
    return Reference()


@pytest.mark.parametrize("rel", list(PINS))
def test_full_frame_bit_identity_and_bmp_pin(ref, oracle, rel, tmp_path):
    assert ref.load(os.path.join(REFERENCE_TREE, rel)) > 0
    scene = ref.scene_arrays()
    rgb_r, rgbf_r, _ = ref.render()
    rgb_o, rgbf_o, _, _ = oracle.render(scene)
    assert same_float_image(rgbf_r, rgbf_o).all()
    assert np.array_equal(rgb_r, rgb_o)
    # the product's BMP writer on the oracle's pixels reproduces the reference's file
    out = tmp_path / "o.bmp"
    ra.write_bmp24(str(out), rgb_o)
    assert hashlib.sha256(out.read_bytes()).hexdigest()[:16] == PINS[rel]
    out_ref = tmp_path / "r.bmp"
    ref.write_bmp(str(out_ref), rgb_r)
    assert out.read_bytes() == out_ref.read_bytes()


@pytest.mark.parametrize("rel", ALL_SCENES)
def test_product_parser_matches_reference_parser(ref, rel):
    path = os.path.join(REFERENCE_TREE, rel)
    assert ref.load(path) > 0
    a = ref.scene_arrays()
    b = ra.parse_scene_file(path, ref_compat=True)
    assert (a.width, a.height, a.spp, a.max_depth, a.proj) == (b.width, b.height, b.spp, b.max_depth, b.proj)
    for n in ("cam_pos", "cam_dir", "cam_up", "background", "ambient", "spheres", "tri_verts", "tri_normals",
              "tri_flags", "lights"):
        x, y = getattr(a, n), getattr(b, n)
        assert x.shape == y.shape and np.array_equal(x.view(np.uint32), y.view(np.uint32)), n
    assert np.float32(a.cam_half_angle) == np.float32(b.cam_half_angle)
    assert np.array_equal(bits(a.materials[a.sphere_mat]), bits(b.materials[b.sphere_mat]))
    assert np.array_equal(bits(a.materials[a.tri_mat]), bits(b.materials[b.tri_mat]))
    assert a.output_name == b.output_name


def test_jittered_supersampling_matches(ref, oracle):
    assert ref.load(os.path.join(REFERENCE_TREE, "SphereExamples/spheres2.scn"), w=120, h=90, spp=6) > 0
    scene = ref.scene_arrays(seed=5)
    _, rgbf_r, _ = ref.render(seed=5)
    _, rgbf_o, _, _ = oracle.render(scene)
    assert same_float_image(rgbf_r, rgbf_o).all()


@pytest.mark.parametrize("w,h,spp", [(160, 120, 1), (200, 150, 4)])
def test_spot_light_falloff_branches_match(ref, oracle, tmp_path, w, h, spp):
    """src/raytrace.cpp:549-566 with angles in the range of acos(): all three branches run (counted in the
    restatement) and the float image is bit-identical to the reference's."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from make_fixtures import SPOT_FALLOFF_SCN
    p = tmp_path / "spot_falloff.scn"
    p.write_text(SPOT_FALLOFF_SCN)
    assert ref.load(str(p), w=w, h=h, spp=spp) > 0
    scene = ref.scene_arrays(seed=3)
    rgb_r, rgbf_r, _ = ref.render(seed=3)
    oracle.spot_branch_counts(reset=True)
    rgb_o, rgbf_o, cnt, _ = oracle.render(scene)
    br = oracle.spot_branch_counts()
    assert min(br) > 0.15 * sum(br), br
    assert same_float_image(rgbf_r, rgbf_o).all() and np.array_equal(rgb_r, rgb_o)
    assert cnt.shadow == 5 * (cnt.shadow // 5) and cnt.secondary > 0      # five lights per hit
    # the product's parser reads the same file the same way (ref-compat film: W x W before the override)
    b = ra.parse_scene_file(str(p), ref_compat=True)
    assert np.array_equal(bits(b.lights), bits(scene.lights)) and (b.width, b.height) == (160, 160)
    assert b.output_name == "spot_falloff.bmp.tail"


def test_generated_scene_text_parses_like_the_reference(ref, tmp_path):
    """The parser fuzzer's generator (tools/fuzz_parser.py: every number spelling the stream extraction accepts or
    rejects, out-of-range values, tabs, CRLF, comments, unknown words, directives one argument short or with
    too many, indices in and out of range), 400 seeds: same success or failure, same arrays bit for bit."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from fuzz_parser import compare, scene_text
    path = str(tmp_path / "fuzz.scn")
    parsed = 0
    for seed in range(400):
        with open(path, "w", newline="") as f:
            f.write(scene_text(seed))
        rc = ref.load(path)
        try:
            b = ra.parse_scene_file(path, ref_compat=True)
        except ra.RtError:
            assert rc < 0, f"seed {seed}: the reference parses what the product rejects"
            continue
        assert rc > 0, f"seed {seed}: the product parses what the reference rejects"
        assert compare(ref.scene_arrays(), b) == [], f"seed {seed}"
        parsed += 1
    assert parsed > 300


def test_random_scenes_render_like_the_reference(ref, oracle, tmp_path):
    """The oracle against the unmodified reference on the seeded random scenes the GPU fuzz test uses
    (tools/fuzz_oracle.py: scene text -> the reference's parser -> both renderers on the scene as the reference
    holds it): float frames bit-identical.  120 ordinary seeds and 60 with the rarer cases (radii <= 0, the eye
    inside a sphere, mirror chains 25 deep, 14 lights)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from _fuzz_scenes import random_scene
    from fuzz_oracle import check
    for seed in list(range(120)) + list(range(20000, 20060)) + [79, 2989, 11138, 13308]:
        why = check(ref, oracle, random_scene(seed), str(tmp_path / "fuzz.scn"))
        assert why is None, f"seed {seed}: {why}"
