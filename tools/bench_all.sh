#!/bin/bash
# One bench line per workload of the measured table (K1, K2, K4 with the reference timed beside them on the host cores --
# K1 on its WHOLE frame, it is the reference's own CPU case; K5 at 1080p / 4 spp without: the reference needs half a
# second per ray there; K3 is the default bench line and has its own profile), written to gpurun_out/bench_all.jsonl on the GPU box.
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
: > gpurun_out/bench_all.jsonl
for w in "k1 --steps 100" "k2 --steps 50" "k4 --steps 50" "k5 --scale 1.0 --film 1920 1080 --spp 4 --steps 5 --no-cpu-baseline"; do
  timeout -k 10 400 python3 -u bench.py --workload $w --warmup 2 >> gpurun_out/bench_all.jsonl 2> gpurun_out/bench_all.err || { echo "bench $w failed"; tail -5 gpurun_out/bench_all.err; exit 1; }
done
timeout -k 10 600 python3 -u tools/k5_shard.py 8 2 >> gpurun_out/bench_all.jsonl 2>> gpurun_out/bench_all.err || { echo "k5 shard failed"; tail -5 gpurun_out/bench_all.err; exit 1; }
timeout -k 10 300 python3 -u tools/sim_scaling.py 5 2>> gpurun_out/bench_all.err | grep '^{' >> gpurun_out/bench_all.jsonl
wc -l gpurun_out/bench_all.jsonl
