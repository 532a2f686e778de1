#!/bin/bash
# k_refine_skew: parity test, then A/B of the bench against the default on the same box.
mkdir -p gpurun_out
export OMP_NUM_THREADS=16 OMP_WAIT_POLICY=passive
[ -n "$SKIPTEST" ] || timeout 900 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "skewed" -p no:cacheprovider > gpurun_out/skew_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/skew_pytest.log
tail -15 gpurun_out/skew_pytest.log
run() {
  python -u bench.py --no-cpu-baseline --steps 6 --warmup 2 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']
print('$*'.ljust(70), d['value'], d['ms_per_step'], 'single', d.get('ms_single_pair'), 'top', s['refine_sweep_top'], 'low', s['refine_sweep'], 'light', s['refine_light_top'], 'skew', s.get('refine_skew_top'), 'frac', d['roofline']['frac'], d['roofline'].get('alone',{}).get('avg_launch_ms'))"
}
{
run
run --opt refine_skew_waves=1024
run --opt refine_skew_waves=1152
run --opt refine_skew_waves=896
run --opt refine_skew_waves=1024 --inflight 4
run
} > gpurun_out/skew_ab.log 2>&1
cat gpurun_out/skew_ab.log
