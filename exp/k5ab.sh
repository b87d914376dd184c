for t in main p16 p24 p32; do
  if [ $t = main ]; then unset RT_LIB; else export RT_LIB=$PWD/exp/lib_$t.so; fi
  timeout -k 10 200 python3 bench.py --workload k5 --scale 1.0 --film 1920 1080 --spp 4 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$t', d['ms_per_frame'], d['roofline']['avg_launch_ms'], d['roofline']['fetches_per_ray'])" || exit 1
done
