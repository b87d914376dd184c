for t in "$@"; do
  if [ $t = main ]; then unset RT_LIB; else export RT_LIB=$PWD/exp/lib_$t.so; fi
  for r in 1 2; do
  timeout -k 10 100 python3 bench.py --workload k4 --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$t k4', d['ms_per_frame'], d['roofline']['avg_launch_ms'], d['roofline']['tests_per_ray'])" || exit 1
  done
  timeout -k 10 200 python3 tools/sim_scaling.py 20 2>/dev/null | cut -c1-70 | tail -1 || exit 1
done
