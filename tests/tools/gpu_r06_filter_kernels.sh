#!/bin/bash
# per-kernel times of the cloud filter on C2's cloud for prebuilt libraries tests/_ab/<name>.so: bash tests/tools/gpu_r06_filter_kernels.sh name1 name2 ...
export TMPDIR=/tmp OMP_NUM_THREADS=16 RSM_AB_OLD_LIBRARY=1
root=$PWD
cp reconstruction_amd/librsm_mi355.so /tmp/keep.so
for n in "$@"; do
  [ -f tests/_ab/$n.so ] && cp tests/_ab/$n.so reconstruction_amd/librsm_mi355.so || cp /tmp/keep.so reconstruction_amd/librsm_mi355.so   # (a name without a file: the tree's library)
  rm -rf /tmp/fs; cd /tmp
  rocprofv3 --kernel-trace --stats -d /tmp/fs -o fs -- python $root/tests/tools/gpu_filter_run.py 6 > /tmp/fs.log 2>&1
  cd $root
  echo "== [$n]"; grep "^filter" /tmp/fs.log | tail -2
  python tests/tools/rocpd_stats.py $(find /tmp/fs -name "*.db") 2>/dev/null | grep "k_sor\|normals" | python -c "
import csv,sys
for r in csv.reader(sys.stdin): print('   %-50s calls %4s avg %9.1f us  min %9.1f  max %9.1f' % (r[0][:50], r[1], float(r[3])/1e3, float(r[5])/1e3, float(r[6])/1e3))"
done
cp /tmp/keep.so reconstruction_amd/librsm_mi355.so
