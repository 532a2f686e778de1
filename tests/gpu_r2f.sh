#!/bin/bash
mkdir -p gpurun_out
export OMP_NUM_THREADS=16 OMP_WAIT_POLICY=passive
timeout 600 python -u -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "two_sweeps or band or fp64" -p no:cacheprovider 2>&1 | tail -15
for cfg in "1 0" "1 32" "1 16" "1 48" "2 32"; do
set -- $cfg
python -u bench.py --no-cpu-baseline --steps 10 --warmup 2 --inflight $1 --opt refine_multi_from=$2 2>gpurun_out/r2f_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']; r=d['roofline']
print('inflight $1 multi_from $2', d['value'], d['ms_per_step'], 'launch', r['avg_launch_ms'], 'frac', r['frac'], 'top', s['refine_sweep_top'], 'low', s['refine_sweep'])" || tail -5 gpurun_out/r2f_err.log
done
