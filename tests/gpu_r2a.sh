#!/bin/bash
# round-2 development driver: banded-refine parity, then bench at several band budgets
mkdir -p gpurun_out
export OMP_NUM_THREADS=16 OMP_WAIT_POLICY=passive
timeout 600 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "band or fp64 or full_pair" -p no:cacheprovider > gpurun_out/r2a_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2a_pytest.log
tail -5 gpurun_out/r2a_pytest.log
for mb in 0 48 96 144 192; do
  timeout 300 python -u bench.py --no-cpu-baseline --steps 10 --warmup 2 --opt refine_band_mb=$mb > gpurun_out/r2a_bench_$mb.log 2>&1
  echo "band_mb=$mb rc=$?"; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r2a_bench_$mb.log").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["stage_ms_per_step"])
except Exception as e:
    print("parse fail", e); print(open("gpurun_out/r2a_bench_$mb.log").read()[-2000:])
PY
done
