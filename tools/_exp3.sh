out=$GRAFT_REPO_ROOT/gpurun_out/w8pmc; rm -rf $out; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
export RT_PACKET_GENS=9 RT_PACKET_SHADOW_GENS=9
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $out -o k -- python3 $GRAFT_REPO_ROOT/tools/_w8.py 8 > $out/log.txt 2>&1 || { tail $out/log.txt; exit 1; }
cd $GRAFT_REPO_ROOT; python3 - <<'PY'
import csv, glob
from collections import defaultdict
f = glob.glob('gpurun_out/w8pmc/**/*counter_collection.csv', recursive=True)[0]
d = defaultdict(dict)
for r in csv.DictReader(open(f)):
    if 'k_extend' in r['Kernel_Name'] or 'k_shadow' in r['Kernel_Name']:
        key = (int(r['Dispatch_Id']), r['Kernel_Name'][5:20])
        d[key][r['Counter_Name']] = float(r['Counter_Value']); d[key]['us'] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for k in sorted(d)[-12:]:
    v = d[k]
    print(k, {n: round(x, 1) for n, x in v.items()})
PY
