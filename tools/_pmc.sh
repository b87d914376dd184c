mkdir -p gpurun_out/pmcx; rm -rf gpurun_out/pmcx/*
export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_BRANCH --output-format csv -d gpurun_out/pmcx/a -o k -- python3 bench.py --workload k3 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmcx/a.log 2>&1 || { echo fail a; tail -3 gpurun_out/pmcx/a.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmcx/b -o k -- python3 bench.py --workload k3 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmcx/b.log 2>&1 || { echo fail b; tail -3 gpurun_out/pmcx/b.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_REQ SQC_TC_STALL --output-format csv -d gpurun_out/pmcx/c -o k -- python3 bench.py --workload k3 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmcx/c.log 2>&1 || { echo fail c; tail -3 gpurun_out/pmcx/c.log; exit 1; }
echo done
