"""GPU suite (-m gpu): the drop-in executable end to end -- scene TEXT -> parser -> HIP render -> BMP --
as a fresh child process, exactly as `./raytrace scene_file` of the reference is run (src/main.cpp:15-83);
and the in-process multi-GPU entry point it uses for `--gpus N`."""
import json
import os
import re
import subprocess

import numpy as np
import pytest

import raytracer_amd as ra
from conftest import ROOT, bits

pytestmark = pytest.mark.gpu

CLI = os.path.join(ROOT, "raytracer_amd", "raytrace")

# comments, CRLF line endings (like ComplexExamples/dragon.scn), a comment tail, normal_triangle, spot lights
# whose angles make every fall-off branch run, film_resolution W H, text after ".bmp" in the output name
SCENE = "\r\n".join("""# drop-in check: a scene file in the reference's text format
camera 0 3 -30 0 -0.05 1 0 1 0 20
film_resolution 96 64
output_image cli_out.bmp.tail
sample_rate 3
max_depth 4
bvh_threshold 500
background .1 .2 .35
ambient_light .1 .1 .1   # trailing comment
material .2 .2 .2 .7 .6 .5 .3 .3 .3 20 0 0 0 1
sphere -5 -1 8 3.5
material 0 0 0 .3 .5 .7 .6 .6 .6 30 .8 .8 .8 1.4
sphere 5 0 4 3
max_vertices 7
material .1 .1 .1 .8 .8 .8 0 0 0 5 0 0 0 1
vertex -60 -5 -30
vertex 60 -5 -30
vertex 60 -5 90
vertex -60 -5 90
triangle 0 1 2
triangle 0 2 3
vertex -30 -5 40
vertex 30 -5 40
vertex 0 30 45
normal 0 0.2 -1
normal 0.3 0 -1
normal -0.3 0.1 -1
material 0 0 0 .6 .7 .6 .2 .2 .2 8 0 0 0 1
normal_triangle 4 5 6 0 1 2
spot_light 900 900 800 0 25 -5 0 -1 0 0.35 0.9
spot_light 400 100 100 -20 10 -10 1 -1 1 0.1 0.2
spot_light 200 300 200 15 12 -15 0 0 1 45 90
point_light 50 50 80 10 20 0
directional_light .2 .2 .2 -1 -1 .5
plane 0 0 0 0 1 0
""".split("\n"))


def run_cli(args, cwd):
    r = subprocess.run([CLI] + list(args), cwd=cwd, capture_output=True, text=True, timeout=300)
    return r.returncode, r.stdout, r.stderr


def expected_bmp(oracle, scene, tmp_path, name="expect.bmp"):
    rgb, _, cnt, _ = oracle.render(scene, threads=16)
    p = tmp_path / name
    ra.write_bmp24(str(p), rgb)
    return p.read_bytes(), cnt


def check_stdout(out, w, h):
    # the reference prints a line whenever the pixel index is a multiple of (int)(.1 * w * h)
    # (src/raytrace.cpp:83-84, :97-98): ten lines when w * h is a multiple of ten, else eleven
    tenper = int(.1 * (w * h))
    decile_lines = -(-(w * h) // tenper)
    lines = out.splitlines()
    assert re.fullmatch(r"Scene parsed in \d+d \d+h \d+m \d+\.\d{6}s", lines[0]), lines[0]
    assert lines[1:1 + decile_lines] == [f"{10 * k}%" for k in range(decile_lines)], lines
    assert lines[1 + decile_lines] == "100%"
    assert re.fullmatch(r"Scene traced in \d+d \d+h \d+m \d+\.\d{6}s", lines[2 + decile_lines]), lines
    return lines[3 + decile_lines:]


def test_cli_renders_the_reference_film_and_writes_the_reference_bmp(oracle, tmp_path):
    (tmp_path / "scene.scn").write_bytes(SCENE.encode())
    rc, out, err = run_cli(["scene.scn"], tmp_path)
    assert rc == 0, (out, err)
    assert check_stdout(out, 96, 96) == []
    # `film_resolution 96 64` is 96 x 96 in the reference (src/parse.cpp:119); name rule of src/main.cpp:65
    scene = ra.parse_scene_file(str(tmp_path / "scene.scn"), ref_compat=True)
    assert (scene.width, scene.height, scene.spp, scene.bvh_threshold) == (96, 96, 3, 500)
    assert scene.tri_flags.tolist() == [0, 0, 1] and len(scene.lights) == 5
    assert sorted(os.listdir(tmp_path)) == ["cli_out.bmp", "scene.scn"]
    want, cnt = expected_bmp(oracle, scene, tmp_path)
    got = (tmp_path / "cli_out.bmp").read_bytes()
    assert len(got) == 54 + 96 * 96 * 3 and got == want
    assert cnt.secondary > 0 and cnt.shadow > 0


def test_cli_flags_film_as_written_stats_and_extra_arguments(oracle, tmp_path):
    (tmp_path / "scene.scn").write_bytes(SCENE.encode())
    rc, out, err = run_cli(["scene.scn", "--film-as-written", "--stats", "--seed", "7", "stray"], tmp_path)
    assert rc == 0, (out, err)
    lines = out.splitlines()
    assert lines[0] == "Too many arguments. Defaulting to first one"         # src/main.cpp:34-37
    rest = check_stdout("\n".join(lines[1:]), 96, 64)
    scene = ra.parse_scene_file(str(tmp_path / "scene.scn"), ref_compat=False).copy(seed=7)
    assert (scene.width, scene.height) == (96, 64)
    want, cnt = expected_bmp(oracle, scene, tmp_path)
    assert (tmp_path / "cli_out.bmp").read_bytes() == want
    st = json.loads(rest[0])
    assert st["rays"] == {"primary": cnt.primary, "secondary": cnt.secondary, "shadow": cnt.shadow}
    assert st["accel"] == "brute" and st["gpus"] == 1                       # 5 primitives <= bvh_threshold 500
    assert st["ms"]["extend"] > 0 and st["ms"]["shadow"] > 0 and st["ms"]["render"] >= st["ms"]["extend"]
    assert st["tests_per_ray"]["closest_hit"]["primitives"] == 5.0
    wall = json.loads(rest[1])["wall_ms"]                                   # second line: where the run's wall clock went
    assert set(wall) == {"parse", "create", "load", "render", "convert", "write_bmp"} and all(v >= 0 for v in wall.values())
    assert wall["render"] >= st["ms"]["render"]
    # LBVH on the same file: same bytes, and the work counters then count node fetches
    rc, out, err = run_cli(["scene.scn", "--film-as-written", "--seed", "7", "--accel", "lbvh", "--stats-file", "st.json"], tmp_path)
    assert rc == 0, (out, err)
    assert (tmp_path / "cli_out.bmp").read_bytes() == want
    st, wall = [json.loads(l) for l in (tmp_path / "st.json").read_text().splitlines()]
    assert "wall_ms" in wall
    assert st["accel"] == "lbvh" and st["fetches_per_ray"]["closest_hit"]["nodes"] > 0


def test_cli_error_exits_on_a_gpu_box(tmp_path):
    rc, out, _ = run_cli([], tmp_path)
    assert rc == 255 and out == "No Scene File Specified\n"
    rc, out, _ = run_cli(["missing.scn"], tmp_path)
    assert rc == 255 and out == "File could not be opened: missing.scn\nScene Parse Failed\n"
    (tmp_path / "bad.scn").write_text("vertex 0 0 0\nvertex 1 0 0\ntriangle 0 1 2\n")
    rc, out, _ = run_cli(["bad.scn"], tmp_path)
    assert rc == 255 and out == "Error: Specified vertex (2) in triangle (0) does not exist\nScene Parse Failed\n"


def test_cli_debug_lines_and_progress_for_a_film_that_is_not_a_multiple_of_ten(oracle, tmp_path):
    # 33 x 33 = 1089 pixels, (int)(.1 * 1089) = 108: the reference prints ELEVEN lines 0% .. 100% before the final 100%
    text = "film_resolution 33 33\nsphere 0 0 5 1\npoint_light 9 9 9 0 3 0\nbackground .2 .2 .2\noutput_image noext\n"
    (tmp_path / "s.scn").write_text(text)
    rc, out, err = run_cli(["s.scn", "--debug"], tmp_path)
    assert rc == 0, (out, err)
    head, _, rest = out.partition("Printing\n")
    dump, _, tail = rest.partition("Beginning Trace\n")
    assert "Spheres:\nPosition: (0.000000 0.000000 5.000000)\nRadius: 1.000000\n" in dump and "No Triangles" in dump
    lines = tail.splitlines()
    assert lines[0].startswith("P1: (") and lines[1].startswith("P2: (") and lines[2].startswith("Plane Dist: ")
    assert lines[3:15] == [f"{10 * k}%" for k in range(11)] + ["100%"]
    assert lines[-1] == "Outputed image to file: noe"                         # no ".bmp": find() + 4 wraps to 3 characters
    scene = ra.parse_scene_file(str(tmp_path / "s.scn"))
    want, _ = expected_bmp(oracle, scene, tmp_path)
    assert (tmp_path / "noe").read_bytes() == want


def test_cli_gpus_flag_and_in_process_multi_gpu_render(oracle, tmp_path):
    """`--gpus N` = rt_render_multi: one context and one host thread per device.  This box has one GPU, so the
    library call is made with N contexts on device 0 (the threads and the shard exchange are the same); the
    frame must equal the whole-frame render byte for byte for every N.  The executable is run with --gpus 1."""
    (tmp_path / "scene.scn").write_bytes(SCENE.encode())
    scene = ra.parse_scene_file(str(tmp_path / "scene.scn"), ref_compat=False).copy(width=200, height=136, spp=2)
    with ra.Tracer() as t0:
        t0.load_scene(scene)
        whole, _, st_whole = t0.render()
        for n in (1, 2, 3, 5):
            others = [ra.Tracer(device=0) for _ in range(n - 1)]
            try:
                for t in others:
                    t.load_scene(scene)
                rgb, st = ra.render_multi([t0] + others)
            finally:
                for t in others:
                    t.close()
            assert np.array_equal(rgb, whole), n
            assert (st.rays_primary, st.rays_secondary, st.rays_shadow) == (st_whole.rays_primary, st_whole.rays_secondary, st_whole.rays_shadow)
        # mismatched scenes are an error, not a wrong picture
        with ra.Tracer() as t1:
            t1.load_scene(scene.copy(width=100))
            with pytest.raises(ra.RtError):
                ra.render_multi([t0, t1])
    rc, out, err = run_cli(["scene.scn", "--film-as-written", "--gpus", "1", "--film", "200", "136", "--spp", "2"], tmp_path)
    assert rc == 0, (out, err)
    p = tmp_path / "w.bmp"
    ra.write_bmp24(str(p), whole)
    assert (tmp_path / "cli_out.bmp").read_bytes() == p.read_bytes()
    rc, out, err = run_cli(["scene.scn", "--gpus", "64"], tmp_path)              # more devices than the box has
    assert rc == 255 and out.endswith("Scene Trace Failed\n")


def test_progress_callback_reports_every_batch(tracer):
    scene = ra.SceneArrays.load(os.path.join(ROOT, "tests", "golden", "scenes", "spheres1.npz")).copy(width=200, height=100)
    seen = []
    with ra.Tracer(batch_samples=32 * 32 * 6) as t:            # 28 tiles in batches of six
        t.load_scene(scene)
        t.set_progress_callback(lambda done, total: seen.append((done, total)))
        rgb, _, _ = t.render()
        t.set_progress_callback(None)
        n = len(seen)
        rgb2, _, _ = t.render()
        assert len(seen) == n and np.array_equal(rgb, rgb2)
    assert seen[0] == (0, 200 * 100) and seen[-1] == (200 * 100, 200 * 100)
    assert len(seen) == 1 + 5 and all(a[0] < b[0] for a, b in zip(seen, seen[1:]))


def test_c_abi_from_plain_c_writes_the_same_bmp(tmp_path):
    """raytracer_amd/abi_example is include/rt_api.h used from C99 (gcc, no C++): parse -> create -> load -> render ->
    BMP.  Its file must be the CLI's, byte for byte, and its ray count the library's."""
    exe = os.path.join(ROOT, "raytracer_amd", "abi_example")
    assert os.path.exists(exe), "raytracer_amd/abi_example not built (make -C raytracer_amd/csrc)"
    (tmp_path / "scene.scn").write_bytes(SCENE.encode())
    rc, out, err = run_cli(["scene.scn"], tmp_path)
    assert rc == 0, (out, err)
    r = subprocess.run([exe, "scene.scn", "from_c.bmp"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert (tmp_path / "from_c.bmp").read_bytes() == (tmp_path / "cli_out.bmp").read_bytes()
    scene = ra.parse_scene_file(str(tmp_path / "scene.scn"), ref_compat=True)
    with ra.Tracer() as t:
        t.load_scene(scene)
        _, _, st = t.render()
    assert r.stdout.strip() == f"96 x 96, {st.rays} rays -> from_c.bmp"
    # the CLI renders one frame per process and skips the timing of the walks unless asked: same file either way
    rc, out, err = run_cli(["scene.scn", "--calibrate", "--accel", "lbvh"], tmp_path)
    assert rc == 0, (out, err)
    assert (tmp_path / "from_c.bmp").read_bytes() == (tmp_path / "cli_out.bmp").read_bytes()
    # errors come back as codes with a message, the library never exits
    r = subprocess.run([exe, "missing.scn"], cwd=tmp_path, capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "missing.scn" in r.stderr
