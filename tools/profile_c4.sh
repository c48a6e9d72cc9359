#!/bin/bash
# Runs on the GPU box: kernel trace + stats of one decode of the BG1-sized irregular code through the
# row-parallel sweeps, then its HBM traffic counters in separate passes; prints a summary.
# usage: tools/profile_c4.sh <tag>
set -u
TAG=${1:-c4}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/tools/run_c4_once.py" > "$OUT/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$C" -- python3 "$ROOT/tools/run_c4_once.py" > "$OUT/pmc_$C.log" 2>&1 || { echo "pmc $C failed"; tail -5 "$OUT/pmc_$C.log"; exit 1; }
done
grep -E "^(kernel|frames|algorithmic)" "$OUT/trace.log"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, os
out = sys.argv[1]
stats = glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)[0]
print("== rocprofv3 --kernel-trace --stats")
for i, row in enumerate(csv.DictReader(open(stats))):
    if i < 8:
        print("%-60s calls %5s  total %10.3f ms  avg %9.3f ms  %s %%" % (row["Name"][:60], row["Calls"], float(row["TotalDurationNs"]) / 1e6,
                                                                float(row["AverageNs"]) / 1e6, row["Percentage"]))
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(os.path.join(out, "pmc_" + c, "**", "*counter_collection.csv"), recursive=True)[0]
    agg = collections.defaultdict(float)
    for row in csv.DictReader(open(f)):
        if "rows_" in row["Kernel_Name"]:
            agg[row["Kernel_Name"].split("<")[0].split("(")[0]] += float(row["Counter_Value"])
    tot[c] = agg
print("== HBM bytes of the sweep kernels over the decode (FETCH_SIZE KiB x 1024 x 2, WRITE_SIZE KiB x 1024)")
grand = 0.0
for k in sorted(set(tot["FETCH_SIZE"]) | set(tot["WRITE_SIZE"])):
    r, w = tot["FETCH_SIZE"].get(k, 0.0) * 2048.0, tot["WRITE_SIZE"].get(k, 0.0) * 1024.0
    grand += r + w
    print("%-28s read %.4g B  write %.4g B" % (k, r, w))
print("total %.5g B" % grand)
PY
find "$OUT" -name "*.db" -delete
find "$OUT" -name "*kernel_trace.csv" -size +8M -delete
