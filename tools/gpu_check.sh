#!/bin/bash
# One GPU-box call: the GPU suite (plus any test file named as argument, run last), then a K3 bench line.
# Every step has a time limit of its own and the chain stops at the first failure.
set -u
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests "$@" -x -v -m gpu --timeout 300 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1
rc=$?
tail -5 gpurun_out/gpu_tests.log
[ $rc -eq 0 ] || { grep -E "FAILED|Error|error" gpurun_out/gpu_tests.log | head -20; exit $rc; }
timeout -k 10 300 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_k3.json 2> gpurun_out/bench_k3.err
rc=$?
cat gpurun_out/bench_k3.json
[ $rc -eq 0 ] || { tail -20 gpurun_out/bench_k3.err; exit $rc; }
