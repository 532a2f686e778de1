#!/bin/bash
# whole GPU suite + default bench (what the driver runs at round end)
mkdir -p gpurun_out
export OMP_NUM_THREADS=16 OMP_WAIT_POLICY=passive
timeout 1500 python -u -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 > gpurun_out/full_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/full_pytest.log
tail -14 gpurun_out/full_pytest.log
timeout 600 python -u bench.py > gpurun_out/full_bench.log 2>gpurun_out/full_bench.err; echo "bench rc=$?"
tail -1 gpurun_out/full_bench.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
