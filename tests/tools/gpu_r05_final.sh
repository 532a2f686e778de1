#!/bin/bash
# the round's closing evidence with the final library: the whole GPU suite, the default bench line, the filter's kernel trace
export OMP_NUM_THREADS=16 OMP_WAIT_POLICY=passive TMPDIR=/tmp
mkdir -p gpurun_out/final
bash tests/tools/gpu_full.sh > gpurun_out/final/suite_tail.log 2>&1; cp gpurun_out/pytest_full.log gpurun_out/final/r05_gputest.log
python -u bench.py > gpurun_out/final/r05_bench.log 2> gpurun_out/final/r05_bench.err; echo "bench rc=$?"
bash tests/tools/gpu_r05_filter_prof.sh 1 > gpurun_out/final/r05_filter_kernels_window.log 2>&1; cp gpurun_out/filter_kernel_stats_1.csv gpurun_out/final/r05_filter_kernel_stats_window.csv
tail -3 gpurun_out/final/suite_tail.log; tail -1 gpurun_out/final/r05_bench.log | cut -c1-200
