#!/bin/bash
# whole GPU suite, the default bench as the driver runs it, then the round's profiles
mkdir -p gpurun_out
export OMP_NUM_THREADS=16 OMP_WAIT_POLICY=passive
timeout 1500 python -u -m pytest tests -m gpu -x -q --durations=5 -p no:cacheprovider > gpurun_out/pytest_full.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_full.log
tail -10 gpurun_out/pytest_full.log
bash tests/tools/gpu_bench_full.sh
bash tests/tools/gpu_profile_r03b.sh > gpurun_out/prof_r03b.log 2>&1
tail -30 gpurun_out/prof_r03b.log
