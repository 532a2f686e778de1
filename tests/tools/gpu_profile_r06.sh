#!/bin/bash
# round-6 evidence: the default bench line (its own --pmc passes included), C3 / C4 / C5, rocprofv3 kernel stats of the default
# bench command (three pairs in flight), PMC passes of one pair alone (traffic, issue side, scalar / LDS side, instruction mix),
# the per-launch trace of the dominant kernel.  Summaries land in gpurun_out/prof_r06/ (copied to profiles/).
export OMP_NUM_THREADS=16 OMP_WAIT_POLICY=passive TMPDIR=/tmp
root=$PWD
out=$root/gpurun_out/prof_r06; rm -rf $out; mkdir -p $out
python -u bench.py > $out/r06_bench.log 2> $out/r06_bench.err; echo "bench rc=$?"; tail -1 $out/r06_bench.log | cut -c1-260
for c in c3 c5 c4; do
  timeout 600 python -u bench.py --no-cpu-baseline --measure-traffic 0 --config $c --steps 5 --warmup 1 > $out/r06_bench_$c.log 2> $out/r06_bench_$c.err; echo "$c rc=$?"; tail -1 $out/r06_bench_$c.log | cut -c1-200
done
bash tests/tools/gpu_stats_inflight.sh r06 > $out/r06_stats_inflight.log 2>&1; cp gpurun_out/stats_r06/r06_kernel_stats.csv $out/ 2>/dev/null; cp gpurun_out/stats_r06/bench_under_rocprof.log $out/r06_bench_under_rocprof.log 2>/dev/null; head -12 $out/r06_stats_inflight.log
bash tests/tools/gpu_valu_budget.sh > $out/r06_valu_budget.log 2>&1; head -8 $out/r06_valu_budget.log
cd /tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" "GRBM_GUI_ACTIVE" "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_IFETCH"; do
  i=$((i+1)); o=$out/pmc/g$i; mkdir -p $o
  rocprofv3 --pmc $grp --kernel-trace -d $o -o pmc -- python $root/bench.py --pmc-child --inflight 1 --no-cpu-baseline --opt refine_split=0 > $o/stdout.log 2>&1
  echo "pmc group $i rc=$?"
done
cd $root
python tests/tools/rocpd_pmc.py $(find $out/pmc -name "*.db") > $out/r06_pmc_all_kernels.csv 2>$out/pmc_err.log
grep -E "refine_skew<4, 1|refine_skew<4, 0|refine_sweep<1|refine_sweep<0|k_ncc_dot4|k_ncc_rowgemm|k_ncc_slide|k_refine_first|kernel,counter" $out/r06_pmc_all_kernels.csv > $out/r06_pmc_main_kernels.csv
head -60 $out/r06_pmc_main_kernels.csv
bash tests/tools/gpu_r06_trace.sh new > $out/r06_skew_launch_trace.log 2>&1; cat $out/r06_skew_launch_trace.log | cut -c1-300
rm -rf $out/pmc gpurun_out/stats_r06
du -sh $out
