#!/bin/bash
# the whole GPU suite, as the driver runs it
mkdir -p gpurun_out
export OMP_NUM_THREADS=16 OMP_WAIT_POLICY=passive
timeout 2400 python -u -m pytest tests -m gpu -x -q --durations=8 -p no:cacheprovider > gpurun_out/pytest_full.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_full.log
tail -16 gpurun_out/pytest_full.log
