#!/bin/bash
# usage: tools/pmc_run.sh <tag> <kernel-name-substring> -- <program and args>
# Collects SQ / LDS / GRBM counters for kernels whose name contains the substring, one rocprofv3
# pass per counter group (rocprofv3 does not split counters over passes).
set -u
TAG=$1; SUB=$2; shift 3
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
i=0
for GROUP in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
             "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS SQ_WAVES SQ_INSTS_SMEM" \
             "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd /tmp && TMPDIR=/tmp timeout -s KILL 300 rocprofv3 --pmc $GROUP --kernel-trace --output-format csv -d "$OUT/g$i" -- "$@" > "$OUT/g$i.log" 2>&1) || { echo "group $i failed"; tail -5 "$OUT/g$i.log"; continue; }
  f=$(find "$OUT/g$i" -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$SUB" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in row["Kernel_Name"]:
        agg[(row["Kernel_Name"][:52], row["Counter_Name"])].append(float(row["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(k[0], k[1], "n=%d" % len(v), "%.4g" % (sum(v) / len(v)))
PY
done
find "$OUT" -name "*.db" -delete
