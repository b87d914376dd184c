mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1
rc=$?; echo "pytest exit $rc"; tail -5 gpurun_out/gpu_tests.log
[ $rc -eq 0 ] || exit 1
for cfg in "0 0 4194304" "0 0 16777216" "1 1 16777216" "6 6 16777216" "1 1 67108864"; do
  set -- $cfg
  RT_PACKET_GENS=$1 RT_PACKET_SHADOW_GENS=$2 RT_BATCH_SAMPLES=$3 timeout -k 10 120 python -u bench.py --workload k3 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/exp.json 2> gpurun_out/exp.err || { echo "bench failed"; tail -5 gpurun_out/exp.err; exit 1; }
  python -c "
import sys,json
d=json.loads(open('gpurun_out/exp.json').read()); r=d['roofline']
print('packet gens $1/$2 batch $3: ms/frame', d['ms_per_step'], 'Mrays/s', d['value'], r['stage_ms_per_frame'])" || exit 1
done
