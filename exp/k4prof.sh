export RT_PROFILE=1 RT_COUNT_WORK=1
timeout -k 10 100 python3 bench.py --workload k4 --steps 3 --warmup 2 --no-cpu-baseline 2> gpurun_out/k4prof.err | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_frame'], d['roofline']['tests_per_ray'], d['roofline']['fetches_per_ray'], d['config'].get('rays_per_frame'))"
grep "rt profile" gpurun_out/k4prof.err | tail -2
