#!/bin/bash
# round 4: second-way prefill in k_refine_first -- parity, then A/B of the option and of the time-skewed kernel's first sweep
mkdir -p gpurun_out
timeout 900 python -u -m pytest tests/test_gpu_parity.py -x -q -p no:cacheprovider 2>&1 | tail -4
run() { python -u bench.py --no-cpu-baseline --measure-traffic 0 --steps 6 --warmup 1 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']
print('$*'.ljust(60), 'value', d['value'], 'single', d['ms_single_pair'], 'top', s['refine_sweep_top'], 'low', s['refine_sweep'])"; }
run --opt refine_prefill=0
run
for f in 30 22 16 12 8; do run --opt refine_skew_from=$f; done
export NSW=37 NSK=28
bash tests/tools/gpu_trace_refine.sh 2>&1 | head -4 | cut -c1-700
