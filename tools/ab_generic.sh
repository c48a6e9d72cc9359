#!/bin/bash
# A/B of two builds of the library on the generic frame-resident kernel (same box, same inputs).
# usage: tools/ab_generic.sh <baseline .so>   (the tree's own build is the candidate)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
BASE=$1
for CASE in qc_empty qc_full bg1_z32 bg1_z16 r36_2040; do
  for MODE in "" "early"; do
    echo "== $CASE $MODE"
    PERF_LIB=$BASE timeout -k 10 120 python3 $ROOT/tools/run_generic_profile.py $CASE 131072 $MODE 2>/dev/null | sed 's/^/base: /'
    timeout -k 10 120 python3 $ROOT/tools/run_generic_profile.py $CASE 131072 $MODE 2>/dev/null | sed 's/^/new:  /'
  done
done
