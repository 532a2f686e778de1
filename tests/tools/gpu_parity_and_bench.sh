#!/bin/bash
mkdir -p gpurun_out
export OMP_NUM_THREADS=16 OMP_WAIT_POLICY=passive
timeout 600 python -u -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
for f in 1 2 1 2; do
python -u bench.py --no-cpu-baseline --steps 10 --warmup 2 --inflight $f 2>gpurun_out/r2g_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']; r=d['roofline']
print('inflight $f', d['value'], d['ms_per_pair'], 'launch', r['avg_launch_ms'], 'frac', r['frac'], 'top', s['refine_sweep_top'], 'low', s['refine_sweep'])" || tail -5 gpurun_out/r2g_err.log
done
