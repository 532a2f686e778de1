#!/bin/bash
# round-5 evidence: the default bench line, C3 / C4 / C5, the NCC micro, rocprofv3 kernel stats of the default bench command (three
# pairs in flight), PMC passes of one pair alone, the VALU budget.  Summaries land in gpurun_out/prof_r05/ (copied to profiles/).
export OMP_NUM_THREADS=16 OMP_WAIT_POLICY=passive TMPDIR=/tmp
root=$PWD
out=$root/gpurun_out/prof_r05; rm -rf $out; mkdir -p $out
python -u bench.py > $out/r05_bench.log 2> $out/r05_bench.err; echo "bench rc=$?"; tail -1 $out/r05_bench.log | cut -c1-260
for c in c3 c5 c4; do
  timeout 600 python -u bench.py --no-cpu-baseline --measure-traffic 0 --config $c --steps 5 --warmup 1 > $out/r05_bench_$c.log 2> $out/r05_bench_$c.err; echo "$c rc=$?"; tail -1 $out/r05_bench_$c.log | cut -c1-200
done
python tests/tools/gpu_ncc_micro.py > $out/r05_ncc_micro.log 2>&1; tail -6 $out/r05_ncc_micro.log
bash tests/tools/gpu_stats_inflight.sh r05 > $out/r05_stats_inflight.log 2>&1; cp gpurun_out/stats_r05/r05_kernel_stats.csv $out/ 2>/dev/null; cp gpurun_out/stats_r05/bench_under_rocprof.log $out/r05_bench_under_rocprof.log 2>/dev/null; head -12 $out/r05_stats_inflight.log
bash tests/tools/gpu_valu_budget.sh > $out/r05_valu_budget.log 2>&1; head -8 $out/r05_valu_budget.log
cd /tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" "GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1)); o=$out/pmc/g$i; mkdir -p $o
  rocprofv3 --pmc $grp --kernel-trace -d $o -o pmc -- python $root/bench.py --pmc-child --inflight 1 --no-cpu-baseline --opt refine_split=0 > $o/stdout.log 2>&1
  echo "pmc group $i rc=$?"
done
cd $root
# the cloud filter on C2's cloud: kernel trace with the pixel-window pass (probed radius) and without
bash tests/tools/gpu_r05_filter2.sh > $out/r05_filter_radii.log 2>&1; tail -5 $out/r05_filter_radii.log
bash tests/tools/gpu_r05_filter_prof.sh 1 > $out/r05_filter_kernels_window.log 2>&1; cp gpurun_out/filter_kernel_stats_1.csv $out/r05_filter_kernel_stats_window.csv
bash tests/tools/gpu_r05_filter_prof.sh 0 > $out/r05_filter_kernels_generic.log 2>&1; cp gpurun_out/filter_kernel_stats_0.csv $out/r05_filter_kernel_stats_generic.csv
python tests/tools/rocpd_pmc.py $(find $out/pmc -name "*.db") > $out/r05_pmc_all_kernels.csv 2>$out/pmc_err.log
grep -E "refine_skew<4, 1|refine_skew<4, 0|refine_sweep<1|k_ncc_dot4|k_ncc_rowgemm|k_ncc_slide|k_refine_first|kernel,counter" $out/r05_pmc_all_kernels.csv > $out/r05_pmc_main_kernels.csv
head -40 $out/r05_pmc_main_kernels.csv
rm -rf $out/pmc gpurun_out/stats_r05
du -sh $out
