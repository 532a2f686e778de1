#!/bin/bash
# extra PMC counters of the dominant kernel (one pair alone per pass) for prebuilt libraries tests/_ab/<name>.so:
#   COUNTERS="A B C" bash tests/tools/gpu_r06_pmc.sh name1 name2 ...
export TMPDIR=/tmp RSM_AB_OLD_LIBRARY=1
root=$PWD
cp reconstruction_amd/librsm_mi355.so /tmp/keep.so
for n in "$@"; do
  [ -f tests/_ab/$n.so ] && cp tests/_ab/$n.so reconstruction_amd/librsm_mi355.so || cp /tmp/keep.so reconstruction_amd/librsm_mi355.so   # (a name without a file: the tree's library)
  out=/tmp/pmc_r06; rm -rf $out; mkdir -p $out
  cd /tmp
  rocprofv3 --pmc $COUNTERS --kernel-trace -d $out -o pmc -- python $root/bench.py --pmc-child --inflight 1 --no-cpu-baseline --opt refine_split=0 > $out/log 2>&1
  cd $root
  echo "== [$n] $COUNTERS"
  python tests/tools/rocpd_pmc.py $(find $out -name "*.db") 2>/dev/null | python -c "
import csv,sys
for r in csv.reader(sys.stdin):
    if 'k_refine_skew<4, 1' in r[0]: print('%-40s %-26s %16.0f  (%s dispatches)' % (r[0][:40], r[1], float(r[3]), r[2]))"
done
cp /tmp/keep.so reconstruction_amd/librsm_mi355.so
