#!/bin/bash
# rocprofv3 PMC passes (one counter group per run, kernel-trace only) -> gpurun_out/pmc_<tag>/<group>/
tag=${1:-r01}; shift
export OMP_NUM_THREADS=16 OMP_WAIT_POLICY=passive TMPDIR=/tmp
base=$PWD/gpurun_out/pmc_$tag
rm -rf $base; mkdir -p $base
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1)); out=$base/g$i; mkdir -p $out
  rocprofv3 --pmc $grp --kernel-trace -d $out -o pmc -- python bench.py --no-cpu-baseline --steps 1 --warmup 0 "$@" > $out/stdout.log 2>&1
  echo "group $i ($grp) rc=$?"
done
ls -la $base/*/ | head -30
