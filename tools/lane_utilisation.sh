#!/bin/bash
# Fraction of the vector lanes that do work in every traversal launch of one bench run:
#   SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)   per dispatch, with its duration and instruction count
# (one counter pass, no trace domain).   usage: tools/lane_utilisation.sh <tag> <bench.py arguments...>
#   -> gpurun_out/lane_util_<tag>.txt        e.g.  tools/lane_utilisation.sh k5 --workload k5 --scale 1.0 --film 1920 1080 --spp 4
tag=$1; shift
root="$(cd "$(dirname "$0")/.." && pwd)"
out=$root/gpurun_out/pmc_util_$tag; rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
cd $root
timeout -s KILL 300 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $out -o k -- python3 bench.py "$@" --steps 1 --warmup 1 --no-cpu-baseline > $out/log.txt 2>&1 || { tail -5 $out/log.txt; exit 1; }
find $out -name "*.db" -delete
python3 - $out "$tag" "$*" <<'PY' > gpurun_out/lane_util_$tag.txt
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
by = {}
for r in csv.DictReader(open(f)):
    if not re.search(r'k_(extend|shadow|shade)<', r['Kernel_Name']): continue
    key = (int(r['Dispatch_Id']), re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', ''))
    by.setdefault(key, {})[r['Counter_Name']] = float(r['Counter_Value'])
    by[key]['us'] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
keys = sorted(by)
print(f"# {sys.argv[2]}: bench.py {sys.argv[3]}; the last 24 traversal / shade launches of the run (counter pass: launches run one at a time)")
print("# dispatch kernel                         us   VALU wave-instructions   lanes busy")
for k in keys[-24:]:
    v = by[k]
    print(f"{k[0]:6d} {k[1]:28s} {v['us']:8.0f} {v['SQ_INSTS_VALU'] / 1e6:12.1f} M  {v['SQ_THREAD_CYCLES_VALU'] / (v['SQ_ACTIVE_INST_VALU'] * 64):12.3f}")
PY
tail -30 gpurun_out/lane_util_$tag.txt
