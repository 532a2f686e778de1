#!/bin/bash
# per-kernel times of the cloud filter on C2's cloud for option sets: bash tests/tools/gpu_r06_filter_opts.sh "filter_list=7" "filter_list=23" ...
export TMPDIR=/tmp OMP_NUM_THREADS=16
root=$PWD
for o in "$@"; do
  rm -rf /tmp/fs; cd /tmp
  rocprofv3 --kernel-trace --stats -d /tmp/fs -o fs -- python $root/tests/tools/gpu_filter_run.py 6 $o > /tmp/fs.log 2>&1
  cd $root
  echo "== [$o]"; grep "^filter" /tmp/fs.log | tail -3
  python tests/tools/rocpd_stats.py $(find /tmp/fs -name "*.db") 2>/dev/null | grep "k_sor\|normals\|k_xq\|radix_sort_onesweep_iteration" | python -c "
import csv,sys
for r in csv.reader(sys.stdin): print('   %-50s calls %4s avg %9.1f us  min %9.1f  max %9.1f' % (r[0][:50], r[1], float(r[3])/1e3, float(r[5])/1e3, float(r[6])/1e3))"
done
