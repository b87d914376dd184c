"""CPU suite: the C-ABI library loads and exports every declared symbol, and the host-only
pieces either side of the trace loop (scene parser, output-name rule, BMP writer, tile
partition) behave like the reference's.  No compute call is made: there is no GPU here."""
import ctypes as C
import hashlib
import os
import re

import numpy as np
import pytest

import raytracer_amd as ra
from raytracer_amd import api
from conftest import GOLD, ROOT


def test_library_exports_every_symbol_the_header_declares():
    hdr = open(os.path.join(ROOT, "include", "rt_api.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations found"
    assert declared == set(api.EXPORTED_SYMBOLS), declared ^ set(api.EXPORTED_SYMBOLS)
    lib = C.CDLL(api.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in rt_api.h but not exported"
    api.load_library()


def test_strerror_and_no_device_is_an_error_not_a_fallback():
    lib = api.load_library()
    assert lib.rt_strerror(0) == b"ok"
    assert b"no CPU fallback" in lib.rt_strerror(api.RT_ERR_NO_DEVICE)
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(ra.RtError) as e:
            ra.Tracer(device=0)
        assert e.value.status in (api.RT_ERR_NO_DEVICE, 5)


SCN = """# a comment line
camera -6 2 -4 .77 0 .64 0 1 0 35
film_resolution 320 200
output_image out/some_name.bmp trailing
#inline comment
max_vertices 4
material .75 .75 .75 .75 .75 .75 .3 .3 .3 32 .2 .2 .2 1.5
sphere 0 -50 0 50
vertex -1 0 0
vertex 1 0 0
vertex 0 1 0
vertex 0 0 2
normal 0 0 2
normal 0 3 0
normal 1 1 0
triangle 0 1 2
material 1 0 0 1 0 0 0 0 0 5 0 0 0 1
normal_triangle 0 1 3 0 1 2
sphere 1 2 3 .5
point_light 10 10 10 0 5 0
directional_light 1 1 1 -1 -1 1
spot_light 6 6 6 0 3 1 0 -1 0 30 60
ambient_light .25 .25 .25
background .05 .05 .05
ray_type orthographic
sample_rate 3
bvh_depth 7
plane 0 0 0 0 1 0
max_depth 4
"""


def _parse_text(tmp_path, text, **kw):
    p = tmp_path / "s.scn"
    p.write_bytes(text.encode())
    return ra.parse_scene_file(str(p), **kw)


def test_parser_grammar(tmp_path):
    s = _parse_text(tmp_path, SCN)
    assert (s.width, s.height, s.spp, s.max_depth, s.proj) == (320, 200, 3, 4, ra.RT_PROJ_ORTHO)
    assert s.output_name == "out/some_name.bmp"
    assert np.allclose(s.cam_pos, [-6, 2, -4]) and np.allclose(s.cam_dir, [.77, 0, .64])
    assert np.float32(s.cam_half_angle) == np.float32(np.float32(35) * 0.01745329251)
    assert s.spheres.shape == (2, 4) and s.tri_verts.shape == (2, 9)
    # material 0 is the parser default, sticky materials follow in order of appearance
    assert s.materials.shape == (3, 14) and list(s.materials[0]) == [0, 0, 0, 1, 1, 1, 0, 0, 0, 5, 0, 0, 0, 1]
    assert list(s.sphere_mat) == [1, 2] and list(s.tri_mat) == [1, 2] and list(s.tri_flags) == [0, 1]
    # flat normal = norm(cross(v1-v0, v2-v0)) thrice; file normals normalised on read
    assert np.allclose(s.tri_normals[0].reshape(3, 3), [[0, 0, 1]] * 3)
    assert np.allclose(s.tri_normals[1].reshape(3, 3), [[0, 0, 1], [0, 1, 0], [2 ** -.5, 2 ** -.5, 0]])
    assert [int(t) for t in s.lights[:, 0]] == [2, 1, 3]
    assert list(s.lights[2, 10:12]) == [30, 60]                      # spot angles stored raw
    assert np.allclose(s.lights[0, 7:10], [0, 1, 0]) and np.allclose(s.lights[1, 4:7], [0, 0, 0])
    assert np.allclose(s.ambient, .25) and np.allclose(s.background, .05)


def test_parser_reference_compat_resolution(tmp_path):
    assert _parse_text(tmp_path, "film_resolution 300 200\n", ref_compat=True).height == 300
    assert _parse_text(tmp_path, "film_resolution 300 200\n").height == 200


def test_parser_defaults_and_crlf(tmp_path):
    s = _parse_text(tmp_path, "sphere 0 0 5 1\r\nmax_depth 2\r\n")
    assert (s.width, s.height, s.spp, s.max_depth, s.proj) == (640, 480, 1, 2, ra.RT_PROJ_PERSP)
    assert s.output_name == "raytraced.bmp" and list(s.cam_dir) == [0, 0, 1]
    assert np.float32(s.cam_half_angle) == np.float32(45.0 * 0.01745329251)


def test_parser_errors(tmp_path):
    with pytest.raises(ra.RtError) as e:
        _parse_text(tmp_path, "vertex 0 0 0\nvertex 1 0 0\ntriangle 0 1 2\n")
    assert e.value.status == api.RT_ERR_PARSE and "vertex (2) in triangle (0)" in str(e.value)
    with pytest.raises(ra.RtError) as e:
        _parse_text(tmp_path, "vertex 0 0 0\nvertex 1 0 0\nvertex 0 1 0\nnormal 0 0 1\nnormal_triangle 0 1 2 0 0 1\n")
    assert "normal (1)" in str(e.value)
    with pytest.raises(ra.RtError) as e:
        ra.parse_scene_file(str(tmp_path / "missing.scn"))
    assert e.value.status == api.RT_ERR_IO


def test_parser_stream_quirks(tmp_path):
    # an over-long comment tail ends parsing silently; what was read before it stays
    s = _parse_text(tmp_path, "sphere 0 0 5 1\n# " + "x" * 400 + "\nsphere 0 0 9 1\n")
    assert len(s.spheres) == 1
    # a malformed number zero-fills the directive, applies it, and ends parsing
    s = _parse_text(tmp_path, "sphere 1 2 oops 4\nsphere 0 0 9 1\n")
    assert len(s.spheres) == 1 and list(s.spheres[0]) == [1, 2, 0, 0]
    # ints are read from the character stream: "5.0" yields 5 and leaves ".0" as a stray token
    s = _parse_text(tmp_path, "max_depth 5.0\nsample_rate 2\n")
    assert (s.max_depth, s.spp) == (5, 2)


def test_parser_number_out_of_float_range(tmp_path):
    """`scn >> float` on 1e39: the largest finite float of that sign is stored and the stream fails, which ends
    the parse (std::num_get; src/parse.cpp:102 `while (scn >> str)`); a value too small for a float is rounded
    (to zero here) and parsing goes on."""
    p = tmp_path / "range.scn"
    p.write_text("ambient_light .0068 15. 1e39\nbackground 1 2 3\n")
    s = ra.parse_scene_file(str(p))
    assert s.ambient.tolist() == [np.float32(.0068), 15.0, np.finfo(np.float32).max] and s.background.tolist() == [0, 0, 0]
    p.write_text("ambient_light .5 -1e39 7\nbackground 1 2 3\n")
    s = ra.parse_scene_file(str(p))
    assert s.ambient.tolist() == [0.5, -np.finfo(np.float32).max, 0.0] and s.background.tolist() == [0, 0, 0]
    p.write_text("ambient_light .5 1e-50 7\nbackground 1 2 3\n")
    s = ra.parse_scene_file(str(p))
    assert s.ambient.tolist() == [0.5, 0.0, 7.0] and s.background.tolist() == [1, 2, 3]


def test_output_name_rule():
    assert ra.output_name("dragon.bmp") == "dragon.bmp"
    assert ra.output_name("a.bmp.tail") == "a.bmp"
    assert ra.output_name("C:\\raytracer\\spheres1.bmp") == "C:\\raytracer\\spheres1.bmp"
    assert ra.output_name("rectangle.scn") == "rec"      # no ".bmp": first three characters


def test_bmp_writer_matches_reference_bytes(kat, tmp_path):
    for w, h in ((5, 3), (6, 2), (7, 2), (8, 1)):
        p = tmp_path / f"{w}.bmp"
        ra.write_bmp24(str(p), kat[f"bmp_{w}x{h}_rgb"])
        assert p.read_bytes() == kat[f"bmp_{w}x{h}_file"].tobytes()
    big = np.load(os.path.join(GOLD, "kat_bmp_big.npz"))["rgb"]
    assert hashlib.sha256(big.tobytes()).hexdigest() == bytes(kat["bmp_640x480_rgb_sha256"]).decode()
    p = tmp_path / "big.bmp"
    ra.write_bmp24(str(p), big)
    data = p.read_bytes()
    assert hashlib.sha256(data).hexdigest() == bytes(kat["bmp_640x480_sha256"]).decode()
    import struct
    assert struct.unpack("<2sIHHIIiiHHIIiiII", data[:54]) == (b"BM", 921654, 0, 0, 54, 40, 640, 480, 1, 24, 0, 921600, 3780, 3780, 0, 0)


def test_jitter_hash_export(kat):
    assert [ra.jitter_hash(1, int(c)) for c in kat["jitter_counter"]] == [int(v) for v in kat["jitter_seed1"]]
    assert [ra.jitter_hash(99, int(c)) for c in kat["jitter_counter"]] == [int(v) for v in kat["jitter_seed99"]]


@pytest.mark.parametrize("w,h", [(640, 480), (1920, 1080), (33, 65), (3840, 2160), (1, 1)])
def test_tile_partition_covers_the_film_once(w, h):
    from raytracer_amd import dist as rd
    ntx, nty = -(-w // ra.RT_TILE), -(-h // ra.RT_TILE)
    for world in (1, 2, 3, 4, 8):
        seen = np.zeros((nty, ntx), np.int32)
        most = 0
        for r in range(world):
            tiles = rd.tile_list(w, h, r, world)
            assert len(tiles) == ra.shard_tiles(w, h, r, world)
            most = max(most, len(tiles))
            for tx, ty in tiles:
                seen[ty, tx] += 1
        assert (seen == 1).all()
        assert ra.shard_bytes(w, h, world) == most * ra.RT_TILE * ra.RT_TILE * 3


# ---------------------------------------------------------------- descriptor validation (no GPU needed)
def test_descriptor_limits_are_rejected_on_the_host():
    lib = api.load_library()
    base = ra.SceneArrays()
    assert ra.validate_scene(base) == api.RT_OK
    assert ra.validate_scene(base.copy(max_depth=63)) == api.RT_OK
    for depth in (64, 200, 2**31 - 1):          # INT_MAX used to overflow `max_depth + 1`
        assert ra.validate_scene(base.copy(max_depth=depth)) == api.RT_ERR_UNSUPPORTED
        assert b"max_depth" in lib.rt_last_error()
    assert ra.validate_scene(base.copy(width=0)) == api.RT_ERR_INVALID
    assert ra.validate_scene(base.copy(spp=0)) == api.RT_ERR_INVALID
    assert ra.validate_scene(base.copy(width=65536 * 32)) == api.RT_ERR_UNSUPPORTED
    # counts beyond what node offsets / hit records can address: only the counts are read before the rejection
    d, keep = base.desc()
    d.n_triangles = (1 << 26) + 1
    d.tri_verts = d.tri_normals = d.tri_flags = d.tri_mat = 1      # non-NULL, never dereferenced
    assert lib.rt_validate_scene(C.byref(d)) == api.RT_ERR_UNSUPPORTED and b"2^26" in lib.rt_last_error()
    d, keep = base.desc()
    d.n_materials = 1 << 29
    assert lib.rt_validate_scene(C.byref(d)) == api.RT_ERR_UNSUPPORTED and b"materials" in lib.rt_last_error()
    # a material index out of range
    bad = base.copy(spheres=np.array([[0, 0, 3, 1]], np.float32), sphere_mat=np.array([5], np.int32))
    assert ra.validate_scene(bad) == api.RT_ERR_INVALID


def test_bvh_threshold_is_parsed_and_carried(tmp_path):
    s = _parse_text(tmp_path, SCN + "bvh_threshold 123\n")
    assert (s.bvh_threshold, s.bvh_depth) == (123, 7)
    d, keep = s.desc()
    assert (d.bvh_threshold, d.bvh_depth) == (123, 7)
    assert _parse_text(tmp_path, "sphere 0 0 3 1\n").bvh_threshold == 0
    p = tmp_path / "t.npz"
    s.save(str(p))
    assert ra.SceneArrays.load(str(p)).bvh_threshold == 123


# ---------------------------------------------------------------- the command line, up to the point where a GPU is needed
CLI = os.path.join(ROOT, "raytracer_amd", "raytrace")


def _run_cli(args, cwd):
    import subprocess
    r = subprocess.run([CLI] + list(args), cwd=cwd, capture_output=True, text=True, timeout=120)
    return r.returncode, r.stdout


def test_cli_error_exits_match_the_reference(tmp_path):
    # src/main.cpp:27-30: no argument -> message, exit(-1)
    rc, out = _run_cli([], tmp_path)
    assert rc == 255 and out == "No Scene File Specified\n"
    # src/parse.cpp:96-99 + src/main.cpp:41-44
    rc, out = _run_cli(["nothing_here.scn"], tmp_path)
    assert rc == 255 and out == "File could not be opened: nothing_here.scn\nScene Parse Failed\n"
    # src/parse.cpp:205-216: index checks name the vertex and the triangle being added
    (tmp_path / "bad.scn").write_text("vertex 0 0 0\nvertex 1 0 0\nvertex 0 1 0\ntriangle 0 1 2\ntriangle 0 1 3\n")
    rc, out = _run_cli(["bad.scn"], tmp_path)
    assert rc == 255 and out == "Error: Specified vertex (3) in triangle (1) does not exist\nScene Parse Failed\n"
    (tmp_path / "badn.scn").write_text("vertex 0 0 0\nvertex 1 0 0\nvertex 0 1 0\nnormal 0 0 1\nnormal_triangle 0 1 2 0 0 1\n")
    rc, out = _run_cli(["badn.scn"], tmp_path)
    assert rc == 255 and out == "Error: Specified normal (1) in triangle (0) does not exist\nScene Parse Failed\n"


@pytest.mark.skipif(not (os.path.isdir("/root/reference") and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libref_oracle.so"))),
                    reason="reference build / tree not present")
@pytest.mark.parametrize("rel", ["SphereExamples/spot_sphere.scn", "TriangleExamples/outdoor.scn", "ComplexExamples/test.scn"])
def test_cli_debug_dump_equals_the_reference_printscene(tmp_path, rel):
    """`--debug` prints what the reference's DEBUG build prints after parsing (src/main.cpp:50-53): compared
    with the reference's own printScene() on the same file.  ComplexExamples/test.scn has
    `film_resolution 1024 768`, which both must report as 1024 x 1024 (src/parse.cpp:119)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("on a GPU box the trace would run: covered by the gpu suite")
    from _oracle import Reference
    path = os.path.join("/root/reference", rel)
    ref = Reference()
    assert ref.load(path) > 0
    want = ref.print_scene()
    rc, out = _run_cli([path, "--debug"], tmp_path)
    assert rc == 255                       # no GPU here: the trace fails after the dump
    head, _, rest = out.partition("Printing\n")
    assert head.startswith("Scene parsed in ")
    dump, _, tail = rest.partition("Beginning Trace\n")
    assert dump == want
    assert tail.endswith("Scene Trace Failed\n")


def test_header_layout_seen_by_a_c_compiler_matches_the_ctypes_binding():
    """abi_example is built with gcc -std=c99 from include/rt_api.h alone; its --layout dump (sizeof / offsetof of
    every struct field) must agree with the ctypes declarations the tests and bench.py call through."""
    import ctypes
    import subprocess
    from raytracer_amd import api
    exe = os.path.join(ROOT, "raytracer_amd", "abi_example")
    assert os.path.exists(exe), "raytracer_amd/abi_example not built (make -C raytracer_amd/csrc)"
    out = subprocess.run([exe, "--layout"], check=True, capture_output=True, text=True).stdout.split("\n")
    structs = {"rt_light": api.RtLight, "rt_scene_desc": api.RtSceneDesc, "rt_options": api.RtOptions, "rt_stats": api.RtStats,
               "rt_tree_stats": api.RtTreeStats}
    seen = {k: set() for k in structs}
    for line in filter(None, out):
        name, *nums = line.split()
        if "." not in name:
            assert ctypes.sizeof(structs[name]) == int(nums[0]), line
            continue
        sname, field = name.split(".")
        desc = getattr(structs[sname], field)
        assert (desc.offset, desc.size) == (int(nums[0]), int(nums[1])), line
        seen[sname].add(field)
    for sname, st in structs.items():
        assert seen[sname] == {f for f, _ in st._fields_}, sname


def test_every_environment_switch_of_the_library_is_documented():
    """Every RT_* variable the native sources read with getenv appears in INTEGRATION.md's table."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = set()
    for f in glob.glob(os.path.join(root, "raytracer_amd", "csrc", "*.*")):
        if f.endswith((".hip", ".cpp", ".h", ".c")):
            names |= set(re.findall(r'getenv\("(RT_[A-Z_0-9]+)"\)', open(f).read()))
    assert len(names) > 10
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    missing = sorted(n for n in names if n not in doc)
    assert not missing, missing
