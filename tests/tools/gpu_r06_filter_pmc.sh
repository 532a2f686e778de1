#!/bin/bash
# the cloud filter's kernels on C2's cloud: rocprofv3 kernel stats, then --pmc passes (counters in their own runs)
export TMPDIR=/tmp OMP_NUM_THREADS=16
root=$PWD; o=$root/gpurun_out/${OUT:-r06_filter}; mkdir -p $o
cd /tmp
rm -rf /tmp/fs; rocprofv3 --kernel-trace --stats -d /tmp/fs -o fs -- python $root/tests/tools/gpu_filter_run.py 6 > $o/stats_run.log 2>&1
cd $root
python tests/tools/rocpd_stats.py $(find /tmp/fs -name "*.db") 2>/dev/null | head -70 > $o/kernel_stats.txt
cat $o/kernel_stats.txt
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  rm -rf /tmp/fp; cd /tmp
  rocprofv3 --pmc $C --kernel-trace -d /tmp/fp -o pmc -- python $root/tests/tools/gpu_filter_run.py 3 > $o/pmc_run.log 2>&1
  cd $root
  echo "== $C"
  python tests/tools/rocpd_pmc.py $(find /tmp/fp -name "*.db") 2>/dev/null | grep "k_sor_window" 
done 2>&1 | tee $o/pmc.txt
