#!/bin/bash
mkdir -p gpurun_out
export OMP_NUM_THREADS=16 OMP_WAIT_POLICY=passive
timeout 600 python -u -m pytest tests/test_gpu_configs.py -m gpu -q -x -k "in_flight" -p no:cacheprovider 2>&1 | tail -3
for cfg in "1 1" "2 1" "2 0" "3 1" "2 1" "2 0"; do
set -- $cfg
python -u bench.py --no-cpu-baseline --steps 10 --warmup 2 --inflight $1 --opt heavy_exclusive=$2 2>gpurun_out/r2e_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']; r=d['roofline']
print('inflight $1 excl $2', d['value'], d['ms_per_step'], 'launch', r['avg_launch_ms'], 'frac', r['frac'], 'timed', r['launches_timed_per_step'])" || tail -5 gpurun_out/r2e_err.log
done
