#!/bin/bash
# the default bench command as the driver runs it (with the CPU baseline and the live PMC passes), log under gpurun_out/
mkdir -p gpurun_out
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-16} OMP_WAIT_POLICY=passive
( time python -u bench.py "$@" ) > gpurun_out/bench_full.log 2>&1
echo "rc=$?" >> gpurun_out/bench_full.log
tail -6 gpurun_out/bench_full.log | cut -c1-3000
