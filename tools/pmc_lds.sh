#!/bin/bash
# SQ / LDS counters of the frame-resident kernel on a short run (one pass per counter group).
set -u
TAG=${1:-lds}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 0 --cpu-seconds 0 --no-stream-leg --no-config-legs ${PMC_FRAMES:+--frames $PMC_FRAMES} $*"
i=0
for GROUP in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
             "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS SQ_WAVES SQ_INSTS_SMEM" \
             "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --pmc $GROUP --output-format csv -d "$OUT/g$i" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/g$i.log" 2>&1 || { echo "group $i failed"; tail -5 "$OUT/g$i.log"; continue; }
  f=$(find "$OUT/g$i" -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    if "bp_" in row["Kernel_Name"]:
        agg[(row["Kernel_Name"][:48], row["Counter_Name"])].append(float(row["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(k[0], k[1], "%.4g" % (sum(v) / len(v)))
PY
done
find "$OUT" -name "*.db" -delete
