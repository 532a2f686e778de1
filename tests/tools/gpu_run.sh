#!/bin/bash
# GPU-box driver used during development: parity tests, then the bench, logs under gpurun_out/.
mkdir -p gpurun_out
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-16} OMP_WAIT_POLICY=passive
timeout ${1:-300} python -u -m pytest tests -m gpu -v --durations=10 -p no:cacheprovider > gpurun_out/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest.log
grep -E "passed|failed|rc=" gpurun_out/pytest.log | tail -3
shift
if [ -n "$1" ]; then
  timeout 900 python -u bench.py "$@" > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log
  tail -5 gpurun_out/bench.log
fi
