#!/bin/bash
mkdir -p gpurun_out
export OMP_NUM_THREADS=16 OMP_WAIT_POLICY=passive
timeout 900 python -u -m pytest tests -m gpu -q -x -p no:cacheprovider --deselect tests/test_gpu_fullsize.py::test_fullsize_c2_equals_the_oracle_bit_for_bit > gpurun_out/r2c_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2c_pytest.log
tail -6 gpurun_out/r2c_pytest.log
for i in 1 2 3; do
python -u bench.py --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']
print(d['value'], d['ms_per_step'], 'init', s['initial_match'], 'rematch', s['rematch'], 'top', s['refine_sweep_top'], 'low', s['refine_sweep'])"
done
python tests/gpu_micro.py 2>&1 | tail -12
