mkdir -p gpurun_out
export RT_VERBOSE=1 RT_STAGE_TIMES=1
timeout -k 10 280 python -u bench.py --workload k3 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/exp.json 2> gpurun_out/exp.err || { tail -5 gpurun_out/exp.err; exit 1; }
grep -v amdgpu.ids gpurun_out/exp.err | cut -c1-400 | tail -60
