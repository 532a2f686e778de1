#!/bin/bash
# round-3 (second half) profiles: kernel-trace stats of the default bench command and PMC passes of one pair alone.
# Summaries land in gpurun_out/prof_r03b/ (copy what is to be judged into profiles/).
export OMP_NUM_THREADS=16 OMP_WAIT_POLICY=passive TMPDIR=/tmp
root=$PWD
out=$root/gpurun_out/prof_r03b; rm -rf $out; mkdir -p $out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o r03b -- python $root/bench.py --no-cpu-baseline --measure-traffic 0 > $out/bench_under_rocprof.log 2>&1
echo "stats rc=$?"
f=$(find $out/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $out/r03b_kernel_stats.csv && head -8 $out/r03b_kernel_stats.csv | cut -c1-160
rm -rf $out/stats
tail -1 $out/bench_under_rocprof.log | cut -c1-300
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" "GRBM_GUI_ACTIVE"; do
  i=$((i+1)); o=$out/pmc/g$i; mkdir -p $o
  rocprofv3 --pmc $grp --kernel-trace -d $o -o pmc -- python $root/bench.py --no-cpu-baseline --measure-traffic 0 --steps 1 --warmup 0 --inflight 1 > $o/stdout.log 2>&1
  echo "pmc group $i rc=$?"
done
cd $root
python tests/tools/rocpd_pmc.py $(find $out/pmc -name "*.db") > $out/r03b_pmc_all_kernels.csv 2>$out/pmc_err.log
grep -E "refine_skew<4, 1|refine_sweep<1|k_ncc_dot4|kernel,counter" $out/r03b_pmc_all_kernels.csv > $out/r03b_pmc_main_kernels.csv
head -50 $out/r03b_pmc_main_kernels.csv
rm -rf $out/pmc
du -sh $out
