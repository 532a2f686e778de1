#!/bin/bash
# round-4 evidence: the default bench line, C3 / C4 / C5, the NCC micro, rocprofv3 kernel stats of the default bench command (three
# pairs in flight), PMC passes of one pair alone, the VALU budget.  Summaries land in gpurun_out/prof_r04d/ (copied to profiles/).
export OMP_NUM_THREADS=16 OMP_WAIT_POLICY=passive TMPDIR=/tmp
root=$PWD
out=$root/gpurun_out/prof_r04d; rm -rf $out; mkdir -p $out
python -u bench.py > $out/r04d_bench.log 2> $out/r04d_bench.err; echo "bench rc=$?"; tail -1 $out/r04d_bench.log | cut -c1-260
for c in c3 c5 c4; do
  timeout 600 python -u bench.py --no-cpu-baseline --measure-traffic 0 --config $c --steps 5 --warmup 1 > $out/r04d_bench_$c.log 2> $out/r04d_bench_$c.err; echo "$c rc=$?"; tail -1 $out/r04d_bench_$c.log | cut -c1-200
done
python tests/tools/gpu_ncc_micro.py > $out/r04d_ncc_micro.log 2>&1; tail -6 $out/r04d_ncc_micro.log
bash tests/tools/gpu_stats_inflight.sh r04d > $out/r04d_stats_inflight.log 2>&1; cp gpurun_out/stats_r04d/r04d_kernel_stats.csv $out/ 2>/dev/null; cp gpurun_out/stats_r04d/bench_under_rocprof.log $out/r04d_bench_under_rocprof.log 2>/dev/null; head -12 $out/r04d_stats_inflight.log
bash tests/tools/gpu_valu_budget.sh > $out/r04d_valu_budget.log 2>&1; head -8 $out/r04d_valu_budget.log
cd /tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" "GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1)); o=$out/pmc/g$i; mkdir -p $o
  rocprofv3 --pmc $grp --kernel-trace -d $o -o pmc -- python $root/bench.py --pmc-child --inflight 1 --no-cpu-baseline --opt refine_split=0 > $o/stdout.log 2>&1
  echo "pmc group $i rc=$?"
done
cd $root
python tests/tools/rocpd_pmc.py $(find $out/pmc -name "*.db") > $out/r04d_pmc_all_kernels.csv 2>$out/pmc_err.log
grep -E "refine_skew<4, 1|refine_skew<4, 0|refine_sweep<1|k_ncc_dot4|k_ncc_rowgemm|k_ncc_slide|k_refine_first|kernel,counter" $out/r04d_pmc_all_kernels.csv > $out/r04d_pmc_main_kernels.csv
head -40 $out/r04d_pmc_main_kernels.csv
rm -rf $out/pmc gpurun_out/stats_r04d
du -sh $out
