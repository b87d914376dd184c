#!/bin/bash
# Kernel timeline (rocprofv3 --kernel-trace) of the last of a few frames of one workload: which kernel ran when,
# on which queue, how long.   usage: tools/frame_timeline.sh <tag> <world> <bench workload: k1|k2|k3|k4>
#   -> gpurun_out/timeline_<tag>.txt   (world > 1: rank 0's shard of that many, rendered on this GPU)
tag=$1; world=$2; wl=$3
root="$(cd "$(dirname "$0")/.." && pwd)"
out=$root/gpurun_out/trace_$tag; rm -rf $out; mkdir -p $out
cat > $out/run.py <<PY
import os, sys
sys.path.insert(0, "$root")
import torch
import raytracer_amd as ra
import bench
sname, w, h, spp, depth, desc = bench.WORKLOADS["$wl"]
scene = ra.SceneArrays.load("$root/tests/golden/scenes/" + sname + ".npz").copy(width=w, height=h, spp=spp, seed=1)
if depth is not None: scene.max_depth = depth
t = ra.Tracer(device=0); t.load_scene(scene)
buf = torch.empty(max(ra.shard_bytes(scene.width, scene.height, $world), w * h * 3), dtype=torch.uint8, device="cuda:0")
for _ in range(4):
    if $world > 1: t.render_shard(0, $world, buf.data_ptr())
    else: t.render_device(buf.data_ptr())
torch.cuda.synchronize(); t.close()
PY
cd /tmp; export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv -d $out -o k -- python3 $out/run.py > $out/log.txt 2>&1 || { tail $out/log.txt; exit 1; }
cd $root; python3 - $out "$tag" <<'PY' > gpurun_out/timeline_${tag}.txt
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = max(i for i, r in enumerate(rows) if 'k_init_counters' in r['Kernel_Name'])
t0 = int(rows[idx]['Start_Timestamp'])
print(f"# {sys.argv[2]}: last of four frames; microseconds from the frame's first kernel")
print("# kernel                          queue    start      end  duration   grid")
for r in rows[idx:]:
    n = r['Kernel_Name'].replace('void ', '').split('(')[0][:32]
    s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
    print(f"{n:32s} {r['Queue_Id']:>5s} {s:8.1f} {e:8.1f} {e - s:8.1f} {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>8s}")
PY
tail -2 gpurun_out/timeline_${tag}.txt
