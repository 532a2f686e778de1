#!/bin/bash
# per-kernel totals of the default bench command (three pairs in flight) under rocprofv3 --kernel-trace --stats, for prebuilt
# libraries tests/_ab/<name>.so:  bash tests/tools/gpu_r06_inflight.sh name1 name2 ...
export RSM_AB_OLD_LIBRARY=1
cp reconstruction_amd/librsm_mi355.so /tmp/keep.so
for n in "$@"; do
  cp tests/_ab/$n.so reconstruction_amd/librsm_mi355.so
  echo "== [$n]"
  bash tests/tools/gpu_stats_inflight.sh ab_$n --steps 6 --warmup 1 2>&1 | head -16
done
cp /tmp/keep.so reconstruction_amd/librsm_mi355.so
