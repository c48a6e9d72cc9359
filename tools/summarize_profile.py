#!/usr/bin/env python3
"""Turn the output of tools/profile_bench.sh (gpurun_out/prof_<tag>/) into the small files kept under
profiles/<round>/: kernel stats and trace summaries, and one HBM-traffic JSON per decode kernel with
the gfx950 correction applied (FETCH_SIZE counts half of the bytes read, WRITE_SIZE is exact; both in KiB).

usage: tools/summarize_profile.py <tag> <round-dir> [frames iters n edges]"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
frames, iters, n, edges = (int(v) for v in (sys.argv[3:7] if len(sys.argv) >= 7 else (1 << 20, 50, 4096, 12288)))
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles", rnd)
os.makedirs(dst, exist_ok=True)


def find(sub, name):
    hits = glob.glob(os.path.join(src, sub, "**", "*" + name), recursive=True)
    return hits[0] if hits else None


stats = find("trace", "kernel_stats.csv")
if stats:
    shutil.copy(stats, os.path.join(dst, tag + "_kernel_stats.csv"))
trace = find("trace", "kernel_trace.csv")
if trace:  # one line per launch of the decode kernels only
    with open(trace) as f, open(os.path.join(dst, tag + "_kernel_trace.csv"), "w") as g:
        rd = csv.DictReader(f)
        g.write("Kernel_Name,Start_Timestamp,End_Timestamp,Duration_ms\n")
        for row in rd:
            if "bp_" in row["Kernel_Name"]:
                g.write('"%s",%s,%s,%.3f\n' % (row["Kernel_Name"], row["Start_Timestamp"], row["End_Timestamp"],
                                                (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6))

raw = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    path = find("pmc_" + counter, "counter_collection.csv")
    agg = collections.defaultdict(list)
    if path:
        for row in csv.DictReader(open(path)):
            agg[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    raw[counter] = {k: {"launches": len(v), "mean_counter_value_KiB": sum(v) / len(v)} for k, v in agg.items()}

algorithmic = frames * iters * (4 * edges * 4 + n * 4)
for kernel in raw["FETCH_SIZE"]:
    if "bp_lds_kernel" not in kernel and "bp_decode_kernel" not in kernel:
        continue
    fetch = raw["FETCH_SIZE"][kernel]["mean_counter_value_KiB"] * 1024 * 2
    write = raw["WRITE_SIZE"].get(kernel, {"mean_counter_value_KiB": 0.0})["mean_counter_value_KiB"] * 1024
    name = "resident" if "bp_lds_kernel" in kernel else "stream"
    out = {
        "command": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE -- python3 bench.py --steps 2 --warmup 1 "
                   "--cpu-seconds 0 (separate passes)",
        "workload": {"frames": frames, "iters": iters, "n": n},
        "kernel": kernel,
        "fetch_bytes_per_launch_corrected_x2": fetch,
        "write_bytes_per_launch": write,
        "hbm_bytes_per_launch": fetch + write,
        "algorithmic_bytes_per_launch": float(algorithmic),
        "raw": {c: raw[c].get(kernel) for c in raw},
    }
    with open(os.path.join(dst, "%s_%s_hbm_traffic.json" % (tag, name)), "w") as f:
        json.dump(out, f, indent=1)
    print(name, "hbm bytes per launch %.4g, algorithmic %.4g" % (fetch + write, algorithmic))
