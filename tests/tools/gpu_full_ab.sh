#!/bin/bash
# whole GPU suite, then the skew A/B
mkdir -p gpurun_out
export OMP_NUM_THREADS=16 OMP_WAIT_POLICY=passive
timeout 1500 python -u -m pytest tests -m gpu -x -q --durations=8 -p no:cacheprovider > gpurun_out/pytest_full.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_full.log
tail -14 gpurun_out/pytest_full.log
SKIPTEST=1 bash tests/tools/gpu_skew.sh
