#!/bin/bash
# Runs on the GPU box: kernel trace + stats of the default bench workload, then HBM traffic
# counters in separate passes (rocprofv3 does not split counters over passes).
# usage: tools/profile_bench.sh <tag> [extra bench args]
set -u
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --cpu-seconds 0 $*"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 400 rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$C" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_$C.log" 2>&1 || { echo "pmc $C failed"; tail -5 "$OUT/pmc_$C.log"; exit 1; }
done
find "$OUT" -name "*kernel_stats.csv" | head -1 | xargs -r head -12
for C in FETCH_SIZE WRITE_SIZE; do
  f=$(find "$OUT/pmc_$C" -name "*counter_collection.csv" | head -1)
  echo "== $C ($f)"
  python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    agg[(row["Kernel_Name"][:60], row["Counter_Name"])].append(float(row["Counter_Value"]))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:6]:
    print(k, "launches", len(v), "mean", sum(v) / len(v))
PY
done
# keep only the summaries small enough to merge back
find "$OUT" -name "*.db" -delete
find "$OUT" -name "*kernel_trace.csv" -size +8M -delete
