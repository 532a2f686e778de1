#!/bin/bash
# SQ counters of the time-skewed kernel for option sets (one pair alone per pass): bash tests/tools/gpu_pmc_ab.sh "optsA" "optsB" ...
export TMPDIR=/tmp
root=$PWD; mkdir -p gpurun_out
for o in "$@"; do
  args=""; for kv in $o; do args="$args --opt $kv"; done
  out=/tmp/pmc_ab; rm -rf $out; mkdir -p $out
  cd /tmp
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS --kernel-trace -d $out -o pmc -- python $root/bench.py --pmc-child --inflight 1 --no-cpu-baseline $args > $out/log 2>&1
  cd $root
  echo "== [$o]"
  python tests/tools/rocpd_pmc.py $(find $out -name "*.db") 2>/dev/null | python -c "
import csv,sys
for r in csv.reader(sys.stdin):
    if 'refine_skew' in r[0] and '4, 1>' in r[0]: print('%-44s %-22s %16.0f' % (r[0][:44], r[1], float(r[3])))"
done
