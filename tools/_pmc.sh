mkdir -p gpurun_out/pmck5; rm -rf gpurun_out/pmck5/*
export TMPDIR=/tmp
W="--workload k5 --film 1920 1080 --spp 4"
timeout -s KILL 280 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/pmck5/c -o k -- python3 bench.py $W --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmck5/c.log 2>&1 || { echo fail c; tail -3 gpurun_out/pmck5/c.log; exit 1; }
timeout -s KILL 280 rocprofv3 --pmc FETCH_SIZE TCC_REQ_sum --output-format csv -d gpurun_out/pmck5/a -o k -- python3 bench.py $W --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmck5/a.log 2>&1 || { echo fail a; tail -3 gpurun_out/pmck5/a.log; exit 1; }
echo done
