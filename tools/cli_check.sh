#!/bin/bash
# End-to-end check of the drop-in command line on a GPU box: writes a scene file in the reference's
# text format, runs `raytracer_amd/raytrace scene_file` as the reference would be run, and compares
# the BMP it wrote with (a) the image the library renders from the same file through the Python
# binding and (b) the CPU oracle.  Run from a shell (not from a process that holds the GPU).
set -u
cd "$(dirname "$0")/.." || exit 1
work=$(mktemp -d /tmp/rt_cli.XXXXXX) || exit 1
cat > $work/scene.scn <<'SCN'
# drop-in check
camera 0 1.2 -6 0 -0.1 1 0 1 0 30
film_resolution 333 211
output_image cli_check.bmp
sample_rate 3
max_depth 4
background .1 .2 .35
ambient_light .2 .2 .2
material .8 .2 .2 .8 .2 .2 .4 .4 .4 16 0 0 0 1
sphere -1.2 .9 1 .9
material .1 .1 .1 .2 .2 .2 .1 .1 .1 32 .8 .8 .8 1.4
sphere 1.1 .8 0 .8
material .3 .6 .3 .3 .6 .3 .2 .2 .2 8 0 0 0 1
vertex -8 0 -8
vertex 8 0 -8
vertex 8 0 12
vertex -8 0 12
triangle 0 2 1
triangle 0 3 2
point_light 40 40 40 3 6 -4
directional_light .4 .4 .4 -1 -1 .5
spot_light 60 60 60 -4 5 -2 .6 -1 .4 20 40
SCN
( cd $work && timeout -k 10 120 "$OLDPWD/raytracer_amd/raytrace" scene.scn --stats > stdout.txt 2> stderr.txt ); rc=$?
echo "raytrace exit code $rc"
grep -v amdgpu.ids $work/stderr.txt | head -5
cat $work/stdout.txt
[ $rc -eq 0 ] && [ -f $work/cli_check.bmp ] || { echo "CLI FAILED"; exit 1; }
( cd $work && "$OLDPWD/raytracer_amd/raytrace" > none.txt 2>/dev/null; echo "no-argument exit code $? ($(cat none.txt))" )
( cd $work && "$OLDPWD/raytracer_amd/raytrace" missing.scn > miss.txt 2>/dev/null; echo "missing-file exit code $? ($(tr '\n' '|' < miss.txt))" )
timeout -k 10 300 python3 - $work <<'PY'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import raytracer_amd as ra
from _oracle import Oracle
work = sys.argv[1]
raw = open(os.path.join(work, "cli_check.bmp"), "rb").read()
assert raw[:2] == b"BM"
off = int.from_bytes(raw[10:14], "little"); w = int.from_bytes(raw[18:22], "little"); h = int.from_bytes(raw[22:26], "little")
row = (3 * w + 3) // 4 * 4
img = np.frombuffer(raw, np.uint8, offset=off).reshape(h, row)[:, :3 * w].reshape(h, w, 3)[::-1, :, ::-1]   # bottom-up BGR
scene = ra.parse_scene_file(os.path.join(work, "scene.scn"))
assert (scene.width, scene.height) == (w, h) == (333, 211)
with ra.Tracer() as t:
    t.load_scene(scene)
    rgb, _, st = t.render()
orgb, _, cnt, _ = Oracle().render(scene, threads=16)
print("BMP == library render:", bool(np.array_equal(img, rgb)), "| BMP == oracle:", bool(np.array_equal(img, orgb)),
      "| rays", st.rays, "oracle", cnt.primary + cnt.secondary + cnt.shadow)
assert np.array_equal(img, rgb) and np.array_equal(img, orgb)
print("CLI CHECK OK")
PY
