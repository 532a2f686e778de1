#!/bin/bash
mkdir -p gpurun_out
export OMP_NUM_THREADS=16 OMP_WAIT_POLICY=passive
timeout 900 python -u -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 > gpurun_out/r2b_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2b_pytest.log
tail -15 gpurun_out/r2b_pytest.log
timeout 300 python -u bench.py --no-cpu-baseline --steps 10 --warmup 2 > gpurun_out/r2b_bench.log 2>&1
tail -1 gpurun_out/r2b_bench.log
