#!/bin/bash
# quick loop: refine parity tests, then the bench (no CPU baseline, no PMC passes, no adapter) twice
mkdir -p gpurun_out
timeout 900 python -u -m pytest tests/test_gpu_parity.py -x -q -k "${1:-refine or skew or whole}" -p no:cacheprovider 2>&1 | tail -4
shift
for i in 1 2; do python -u bench.py --no-cpu-baseline --measure-traffic 0 --adapter-pairs 0 --steps 8 --warmup 2 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']; r=d['roofline']
print('value', d['value'], 'single', d['ms_single_pair'], 'top', s['refine_sweep_top'], 'low', s['refine_sweep'], 'skew launch alone', r['alone']['avg_launch_ms'], 'in flight', r['avg_launch_ms'])"; done
