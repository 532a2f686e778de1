#!/bin/bash
# counters of the window kernels of the cloud filter (C2's cloud): COUNTERS="A B ..." bash tests/tools/gpu_r06_filter_pmc2.sh
export TMPDIR=/tmp OMP_NUM_THREADS=16
root=$PWD
for C in "$@"; do
  rm -rf /tmp/fp; cd /tmp
  rocprofv3 --pmc $C --kernel-trace -d /tmp/fp -o pmc -- python $root/tests/tools/gpu_filter_run.py 3 > /tmp/fp.log 2>&1
  cd $root
  echo "== $C"
  python tests/tools/rocpd_pmc.py $(find /tmp/fp -name "*.db") 2>/dev/null | python -c "
import csv,sys
for r in csv.reader(sys.stdin):
    if any(k in r[0] for k in ('k_sor_window<16, false', 'k_sor_window_list', 'k_sor_window_wave', 'k_sor_window_wg', 'k_normals_lattice')):
        print('%-34s %-32s %3s dispatches, mean %16.1f' % (r[0].replace('void ', '')[:34], r[1], r[2], float(r[3])))"
done
