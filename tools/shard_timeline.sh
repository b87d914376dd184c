#!/bin/bash
# Kernel timeline (rocprofv3 --kernel-trace) of the last of three frames of rank 0's shard of a K3 frame split
# over N GPUs, rendered on this GPU: which kernel ran when, on which queue.  Shows the two-stream overlap of
# the thin generations.   usage: tools/shard_timeline.sh [N=8]   -> gpurun_out/shard_timeline_N.txt
world=${1:-8}
root="$(cd "$(dirname "$0")/.." && pwd)"
out=$root/gpurun_out/shard_trace_$world; rm -rf $out; mkdir -p $out
cat > $out/run.py <<PY
import os, sys
sys.path.insert(0, "$root")
import torch
import raytracer_amd as ra
scene = ra.SceneArrays.load("$root/tests/golden/scenes/dragon.npz").copy(width=3840, height=2160, spp=16, seed=1)
t = ra.Tracer(device=0); t.load_scene(scene)
buf = torch.empty(ra.shard_bytes(scene.width, scene.height, $world), dtype=torch.uint8, device="cuda:0")
for _ in range(3): t.render_shard(0, $world, buf.data_ptr())
torch.cuda.synchronize(); t.close()
PY
cd /tmp; export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv -d $out -o k -- python3 $out/run.py > $out/log.txt 2>&1 || { tail $out/log.txt; exit 1; }
cd $root; python3 - $out $world <<'PY' > gpurun_out/shard_timeline_${world}.txt
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = max(i for i, r in enumerate(rows) if 'k_init_counters' in r['Kernel_Name'])
t0 = int(rows[idx]['Start_Timestamp'])
print(f"# K3 (dragon 3840x2160 16 spp), rank 0's shard of {sys.argv[2]}, last of three frames; microseconds from the frame's first kernel")
print("# kernel                  queue    start      end  duration")
for r in rows[idx:]:
    n = r['Kernel_Name'].replace('void ', '').split('(')[0][:24]
    s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
    print(f"{n:24s} {r['Queue_Id']:>5s} {s:8.1f} {e:8.1f} {e - s:8.1f}")
PY
tail -3 gpurun_out/shard_timeline_${world}.txt
