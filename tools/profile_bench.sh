#!/bin/bash
# Profiles one bench workload on the GPU box and leaves everything under gpurun_out/prof_<tag>/:
#   bench.json                 the plain bench line (HIP-event timings, no profiler attached)
#   trace/                     rocprofv3 --kernel-trace --stats of the same command
#   pmc_fetch/ pmc_write/ pmc_sq/   one counter pass each (never combined with a trace domain)
# usage: tools/profile_bench.sh <tag> <workload args...>     e.g.  tools/profile_bench.sh k3 --workload k3
set -u
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p $out
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
# the CPU baseline leg belongs to the default (K3) line; on K5 the brute-force oracle needs minutes per row
extra=""; [ "$tag" != "k3" ] && extra="--no-cpu-baseline"
timeout -k 10 400 python3 -u bench.py "$@" --steps 5 --warmup 2 $extra > $out/bench.json 2> $out/bench.err || { echo "bench failed"; tail -5 $out/bench.err; exit 1; }
echo "bench done"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o k -- python3 bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline > $out/trace.log 2>&1 || { echo "trace failed"; tail -5 $out/trace.log; exit 1; }
echo "trace done"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE TCC_REQ_sum --output-format csv -d $out/pmc_fetch -o k -- python3 bench.py "$@" --steps 1 --warmup 1 --no-cpu-baseline > $out/pmc_fetch.log 2>&1 || { echo "pmc fetch failed"; tail -5 $out/pmc_fetch.log; exit 1; }
echo "pmc fetch done"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_write -o k -- python3 bench.py "$@" --steps 1 --warmup 1 --no-cpu-baseline > $out/pmc_write.log 2>&1 || { echo "pmc write failed"; tail -5 $out/pmc_write.log; exit 1; }
echo "pmc write done"
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $out/pmc_sq -o k -- python3 bench.py "$@" --steps 1 --warmup 1 --no-cpu-baseline > $out/pmc_sq.log 2>&1 || { echo "pmc sq failed"; tail -5 $out/pmc_sq.log; exit 1; }
echo "pmc sq done"
# keep only the small csv files (the merge back is size-limited)
find $out -name "*.db" -delete 2>/dev/null
du -sh $out
