#!/bin/bash
# round-3 profiles: kernel-trace stats of the default bench command, PMC passes (FETCH_SIZE / WRITE_SIZE / L2 hit / SQ) of one
# pair alone, kernel stats of the cloud filter and of the NCC microbenchmark.  Summaries land in gpurun_out/prof_r03/.
export OMP_NUM_THREADS=16 OMP_WAIT_POLICY=passive TMPDIR=/tmp
root=$PWD
out=$root/gpurun_out/prof_r03; rm -rf $out; mkdir -p $out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o r03 -- python $root/bench.py --no-cpu-baseline > $out/bench_under_rocprof.log 2>&1
echo "stats rc=$?"
f=$(find $out/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $out/r03_kernel_stats.csv && head -8 $out/r03_kernel_stats.csv | cut -c1-160
find $out/stats -name "*kernel_trace.csv" -size +1M -delete; find $out/stats -name "*.db" -size +20M -delete
tail -1 $out/bench_under_rocprof.log | cut -c1-300
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1)); o=$out/pmc/g$i; mkdir -p $o
  rocprofv3 --pmc $grp --kernel-trace -d $o -o pmc -- python $root/bench.py --no-cpu-baseline --steps 1 --warmup 0 --inflight 1 > $o/stdout.log 2>&1
  echo "pmc group $i rc=$?"
done
cd $root
python tests/tools/rocpd_pmc.py $(find $out/pmc -name "*.db") > $out/r03_pmc_all_kernels.csv 2>$out/pmc_err.log
grep -E "refine_sweep<1|refine_sweep<0|k_ncc_dot4|k_sor_knn|kernel,counter" $out/r03_pmc_all_kernels.csv | head -40
find $out/pmc -name "*.db" -size +20M -delete
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/filter -o filt -- python $root/tests/tools/gpu_filter_c2.py 2 > $out/filter.log 2>&1
f=$(find $out/filter -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $out/r03_filter_kernel_stats.csv
find $out/filter -name "*kernel_trace.csv" -size +1M -delete; find $out/filter -name "*.db" -size +20M -delete
tail -2 $out/filter.log
cd $root
du -sh $out
