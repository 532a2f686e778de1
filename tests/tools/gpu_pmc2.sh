#!/bin/bash
# rocprofv3 PMC passes for a bench invocation; prints per-kernel means for kernels matching $1
filt=$1; shift
export OMP_NUM_THREADS=16 OMP_WAIT_POLICY=passive TMPDIR=/tmp
base=$PWD/gpurun_out/pmc2
rm -rf $base; mkdir -p $base
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1)); out=$base/g$i; mkdir -p $out
  (cd /tmp && rocprofv3 --pmc $grp --kernel-trace -d $out -o pmc -- python $OLDPWD/bench.py --no-cpu-baseline --steps 1 --warmup 0 --inflight 1 "$@" > $out/stdout.log 2>&1)
  echo "group $i rc=$?"
done
python tests/tools/rocpd_pmc.py --filter=$filt $(find $base -name "*.db") > gpurun_out/pmc2_summary.csv 2>gpurun_out/pmc2_err.log
cat gpurun_out/pmc2_summary.csv | head -60; tail -3 gpurun_out/pmc2_err.log
rm -rf $base  # the .db files exceed what gpurun merges back
