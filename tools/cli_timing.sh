#!/bin/bash
# What a user of the drop-in executable sees: wall-clock split of `./raytrace scene` runs (fresh process each):
# parse / contexts / scene upload + hierarchies / frame (pool allocation + render + copy out) / pixel conversion / BMP file.
#   usage: tools/cli_timing.sh            -> gpurun_out/cli_timing.txt   (needs a GPU)
# The scene text is written from the committed arrays (tests/golden/scenes/*.npz) with the same writer the fixtures use.
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/cli_timing.txt
tmp=$(mktemp -d)
python3 - "$tmp" <<'PY' || exit 1
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import raytracer_amd as ra
from make_fixtures import scene_to_scn
for name, w, h in (("dragon", 640, 480), ("plant_h", 660, 512), ("spheres1", 640, 480)):
    s = ra.SceneArrays.load(f"tests/golden/scenes/{name}.npz").copy(width=w, height=h)
    s.output_name = os.path.join(sys.argv[1], name + ".bmp")
    open(os.path.join(sys.argv[1], name + ".scn"), "w").write(scene_to_scn(s))
PY
: > $out
run() {   # label, then the command line after ./raytrace
  label=$1; shift
  for rep in 1 2 3; do
    t0=$(date +%s.%N)
    RT_VERBOSE=${RT_VERBOSE:-} timeout -k 5 120 raytracer_amd/raytrace "$@" --stats > $tmp/out.txt 2> $tmp/err.txt || { echo "$label failed"; tail -3 $tmp/out.txt $tmp/err.txt; exit 1; }
    t1=$(date +%s.%N)
    echo "$label run $rep: process wall $(python3 -c "print(f'{($t1-$t0)*1e3:.1f}')") ms; $(grep wall_ms $tmp/out.txt); device $(grep -o '"render": [0-9.]*' $tmp/out.txt | head -1)" >> $out
    grep -h "rt pool\|rt bvh" $tmp/err.txt | cut -c1-400 >> $out
  done
}
run "dragon.scn as the reference renders it (640x640, 1 spp)" $tmp/dragon.scn
run "dragon.scn K3 film (3840x2160, 16 spp)" $tmp/dragon.scn --film 3840 2160 --spp 16
run "plant_h.scn (660x660, 1 spp)" $tmp/plant_h.scn
run "spheres1.scn K1 (640x640: film_resolution absent -> 640x480)" $tmp/spheres1.scn
rm -rf $tmp
cat $out
