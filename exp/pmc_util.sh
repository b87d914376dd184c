cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -s KILL 300 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d gpurun_out/pmc_util -o k -- python3 bench.py --workload k5 --scale 1.0 --film 1920 1080 --spp 4 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_util.log 2>&1 || { tail -5 gpurun_out/pmc_util.log; exit 1; }
find gpurun_out/pmc_util -name "*.db" -delete
echo done
