#!/usr/bin/env python3
"""Summarise rocprofv3 output directories into small, commit-able text files.

  python tools/summarize_prof.py <gpurun_out/prof dir> <profiles/prefix> [workload]

Reads <dir>/*_trace/**/_kernel_stats.csv (kernel-trace --stats) and every
<dir>/*/**/_counter_collection.csv (one --pmc pass each) and writes
  <prefix>_kernel_stats.csv   per-kernel calls / total / average duration (our kernels only)
  <prefix>_pmc.json           per kernel and counter: sum over dispatches and mean per dispatch
  <prefix>_timed_launches.json  average duration of the traversal launches of the timed frames only
  <prefix>_bench.json         (with a workload argument) the run's bench line, traffic filled from its own passes
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
    # The traversal kernels of the TIMED frames only.  A run also holds the calibration launches (a
    # quarter-size sample, every walk of both trees) and the counting frame (<.., true> variants), so
    # the plain per-kernel averages above mix launch sizes; these do not.
    gens, steps = 6, 5
    bj = os.path.join(src, "bench.json")
    if os.path.exists(bj):
        b = json.load(open(bj))
        gens, steps = int(b["config"]["max_depth"]) + 1, int(b["steps"])

    def traversal_rows(path, base):
        rows = [r for r in csv.DictReader(open(path))
                if re.search(r"\b" + base + r"<\d+, false>", r["Kernel_Name"])]
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        return rows

    timed = {}
    for f in glob.glob(os.path.join(src, "*trace*", "**", "*_kernel_trace.csv"), recursive=True):
        for base in ("k_extend", "k_shadow"):
            rows = traversal_rows(f, base)[-gens * steps:]
            if rows:
                us = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
                timed[base] = {"launches": len(us), "avg_us": round(sum(us) / len(us), 2),
                               "min_us": round(min(us), 2), "max_us": round(max(us), 2)}
    if len(sys.argv) > 3:
        # HBM bytes per launch for bench.py's roofline.traffic: corrected FETCH_SIZE (x2, an upper bound
        # for narrow accesses) + WRITE_SIZE, averaged over the launches of the LAST frame of each pass
        entry = {}
        for base in ("k_extend", "k_shadow"):
            total = 0.0
            for counter, scale in (("FETCH_SIZE", 2048.0), ("WRITE_SIZE", 1024.0)):
                for f in glob.glob(os.path.join(src, "*", "**", "*_counter_collection.csv"), recursive=True):
                    rows = [r for r in traversal_rows(f, base) if r["Counter_Name"] == counter][-gens:]
                    if rows:
                        total += scale * sum(float(r["Counter_Value"]) for r in rows) / len(rows)
                        entry[base + "_launches_averaged"] = len(rows)
            if total:
                entry[base] = round(total)
        tpath = os.path.join(os.path.dirname(prefix) or ".", "hbm_traffic.json")
        table = json.load(open(tpath)) if os.path.exists(tpath) else {}
        table[sys.argv[3]] = entry
        json.dump(table, open(tpath, "w"), indent=1, sort_keys=True)
        if os.path.exists(bj):
            # the bench line of this profile run, with the traffic its own counter passes measured
            b = json.load(open(bj))
            dominant = b["roofline"]["kernel"].split("<")[0]           # k_extend or k_shadow
            if dominant in entry:
                b["roofline"]["traffic"] = entry[dominant]
                if isinstance(b["roofline"].get("hbm"), dict):
                    b["roofline"]["hbm"]["traffic"] = entry[dominant]
                b["roofline"]["traffic_source"] = "the rocprofv3 PMC passes of this profile run (2 x TCC FETCH_SIZE + WRITE_SIZE per launch)"
            b["roofline"]["rocprofv3_kernel_trace"] = timed
            json.dump(b, open(prefix + "_bench.json", "w"))
    json.dump(timed, open(prefix + "_timed_launches.json", "w"), indent=1, sort_keys=True)
    print(open(prefix + "_kernel_stats.csv").read() if os.path.exists(prefix + "_kernel_stats.csv") else "")
    print(json.dumps(timed))


if __name__ == "__main__":
    main()
