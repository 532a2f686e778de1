#!/bin/bash
mkdir -p gpurun_out
export OMP_NUM_THREADS=16 OMP_WAIT_POLICY=passive
timeout 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_edgecases.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
for cfg in "1 1" "1 0" "2 1" "2 0" "1 1" "2 1"; do
set -- $cfg
python -u bench.py --no-cpu-baseline --steps 10 --warmup 2 --inflight $1 --opt refine_prefill=$2 2>gpurun_out/r2h_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']; r=d['roofline']
print('inflight $1 prefill $2', d['value'], d['ms_per_pair'], 'launch', r['avg_launch_ms'], 'frac', r['frac'], 'top', s['refine_sweep_top'], 'low', s['refine_sweep'])" || tail -5 gpurun_out/r2h_err.log
done
