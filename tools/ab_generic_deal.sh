#!/bin/bash
# A/B of two builds of the library on the generic frame-resident kernel: a copy of the other build placed at
# tools/ab/libldpc_amd_old.so (not kept in the tree) against the tree's
set -e
OUT=gpurun_out/r03/ab_generic_deal.txt
mkdir -p gpurun_out/r03
: > $OUT
for c in bg1_z16 bg1_z32 qc_empty r36_2040; do
  for lib in tools/ab/libldpc_amd_old.so ""; do
    for mode in "" early; do
      echo "lib=${lib:-tree} $c $mode" >> $OUT
      PERF_LIB=$lib timeout -k 10 120 python tools/run_generic_profile.py $c 65536 $mode 2>/dev/null >> $OUT
    done
  done
done
cat $OUT
