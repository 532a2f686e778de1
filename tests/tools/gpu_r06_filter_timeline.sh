#!/bin/bash
# time line of one cloud-filter call on C2's cloud (kernels + copies, gaps between them)
export TMPDIR=/tmp OMP_NUM_THREADS=16
root=$PWD; rm -rf /tmp/ft; cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/ft -o ft -- python $root/tests/tools/gpu_filter_run.py 4 "$@" > /tmp/ft.log 2>&1
cd $root; grep "^filter" /tmp/ft.log | tail -2
python tests/tools/rocpd_timeline.py $(find /tmp/ft -name "*.db")
