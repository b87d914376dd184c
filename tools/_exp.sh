mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1
rc=$?; echo "pytest exit $rc"; tail -5 gpurun_out/gpu_tests.log
[ $rc -eq 0 ] || exit 1
for w in k3 k2 k1 k4; do
timeout -k 10 120 python -u bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/exp.json 2> gpurun_out/exp.err || { echo "bench failed"; tail -5 gpurun_out/exp.err; exit 1; }
  python -c "
import sys,json
d=json.loads(open('gpurun_out/exp.json').read()); r=d['roofline']
print('$w ms/frame', d['ms_per_step'], 'Mrays/s', d['value'], r['stage_ms_per_frame'], d['config']['rays_per_frame'])" || exit 1
done
