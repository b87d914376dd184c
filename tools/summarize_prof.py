#!/usr/bin/env python3
"""Summarise rocprofv3 output directories into small, commit-able text files.

  python tools/summarize_prof.py <gpurun_out/prof dir> <profiles/prefix> [workload]

Reads <dir>/*_trace/**/_kernel_stats.csv (kernel-trace --stats) and every
<dir>/*/**/_counter_collection.csv (one --pmc pass each) and writes
  <prefix>_kernel_stats.csv   per-kernel calls / total / average duration (our kernels only)
  <prefix>_pmc.json           per kernel and counter: sum over dispatches and mean per dispatch
FETCH_SIZE / WRITE_SIZE are reported in KiB by rocprofv3; on gfx950 FETCH_SIZE reads half
of a wide coalesced stream, so `hbm_read_bytes_corrected` doubles it (upper bound for
narrow accesses)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*", "", name).replace("void ", "").strip()
    return name if len(name) < 60 else name[:57] + "..."


def main():
    src, prefix = sys.argv[1], sys.argv[2]
    os.makedirs(os.path.dirname(prefix) or ".", exist_ok=True)
    for f in glob.glob(os.path.join(src, "*trace*", "**", "*_kernel_stats.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        with open(prefix + "_kernel_stats.csv", "w") as out:
            out.write("kernel,calls,total_ms,avg_us,percent,min_us,max_us\n")
            for r in rows:
                n = short(r["Name"])
                if "rocprim" in n:
                    n = "rocprim::" + ("radix_sort_block_sort" if "block_sort" in r["Name"] else "merge_sort_block_merge")
                out.write(f"{n},{r['Calls']},{int(r['TotalDurationNs']) / 1e6:.3f},{float(r['AverageNs']) / 1e3:.2f},"
                          f"{r['Percentage']},{int(r['MinNs']) / 1e3:.2f},{int(r['MaxNs']) / 1e3:.2f}\n")
    pmc = defaultdict(lambda: defaultdict(lambda: [0.0, 0, 0.0]))
    for f in glob.glob(os.path.join(src, "*", "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            c = pmc[k][r["Counter_Name"]]
            c[0] += float(r["Counter_Value"]); c[1] += 1
            c[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    res = {}
    for k, cs in pmc.items():
        if k.startswith("__amd") or "rocprim" in k:
            continue
        res[k] = {}
        for c, (s, n, ms) in cs.items():
            res[k][c] = {"sum": s, "dispatches": n, "mean_per_dispatch": s / max(1, n), "kernel_ms_in_this_pass": round(ms, 3)}
        if "FETCH_SIZE" in cs:
            res[k]["hbm_read_bytes_corrected"] = {"sum": cs["FETCH_SIZE"][0] * 1024 * 2,
                                                  "mean_per_dispatch": cs["FETCH_SIZE"][0] * 1024 * 2 / max(1, cs["FETCH_SIZE"][1])}
        if "WRITE_SIZE" in cs:
            res[k]["hbm_write_bytes"] = {"sum": cs["WRITE_SIZE"][0] * 1024,
                                         "mean_per_dispatch": cs["WRITE_SIZE"][0] * 1024 / max(1, cs["WRITE_SIZE"][1])}
        if "TCC_HIT_sum" in cs and "TCC_MISS_sum" in cs:
            h, m = cs["TCC_HIT_sum"][0], cs["TCC_MISS_sum"][0]
            res[k]["l2_hit_rate"] = h / max(1.0, h + m)
        if "SQ_ACTIVE_INST_VALU" in cs and "SQ_BUSY_CYCLES" in cs:
            res[k]["valu_active_per_wave_cycle"] = cs["SQ_ACTIVE_INST_VALU"][0] / max(1.0, cs["SQ_WAVE_CYCLES"][0])
    json.dump(res, open(prefix + "_pmc.json", "w"), indent=1, sort_keys=True)
    if len(sys.argv) > 3:
        # HBM bytes per launch of the traversal kernels (the variant launched most), for bench.py's
        # roofline.traffic: corrected FETCH_SIZE (x2, an upper bound for narrow accesses) + WRITE_SIZE
        tpath = os.path.join(os.path.dirname(prefix) or ".", "hbm_traffic.json")
        table = json.load(open(tpath)) if os.path.exists(tpath) else {}
        entry = {}
        for base in ("k_extend", "k_shadow"):
            cands = [(v["FETCH_SIZE"]["dispatches"], k) for k, v in res.items()
                     if k.startswith(base) and "FETCH_SIZE" in v and "WRITE_SIZE" in v]
            if cands:
                k = max(cands)[1]
                entry[base] = round(res[k]["hbm_read_bytes_corrected"]["mean_per_dispatch"] +
                                    res[k]["hbm_write_bytes"]["mean_per_dispatch"])
                entry[base + "_variant"] = k
        table[sys.argv[3]] = entry
        json.dump(table, open(tpath, "w"), indent=1, sort_keys=True)
    print(open(prefix + "_kernel_stats.csv").read() if os.path.exists(prefix + "_kernel_stats.csv") else "")
    print(json.dumps({k: {c: (round(v["mean_per_dispatch"], 1) if isinstance(v, dict) else round(v, 4)) for c, v in cs.items()}
                      for k, cs in res.items()}, indent=1))


if __name__ == "__main__":
    main()
