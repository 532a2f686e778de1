#!/bin/bash
# same-box A/B of prebuilt libraries tests/_ab/<name>.so with the dominant kernel's counters: bash tests/tools/gpu_r06_ab.sh reps name1 name2 ...
# (BENCH_OPTS: extra bench.py options; PMC=1: the --pmc passes too)
reps=$1; shift; export RSM_AB_OLD_LIBRARY=1
cp reconstruction_amd/librsm_mi355.so /tmp/keep.so
for r in $(seq $reps); do for n in "$@"; do
  [ -f tests/_ab/$n.so ] && cp tests/_ab/$n.so reconstruction_amd/librsm_mi355.so || cp /tmp/keep.so reconstruction_amd/librsm_mi355.so   # (a name without a file: the tree's library)
  python -u bench.py --no-cpu-baseline --measure-traffic ${PMC:-0} --adapter-pairs 0 --steps 8 --warmup 2 $BENCH_OPTS 2>/tmp/err_$n.log | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']; r=d['roofline']
    print('[$n]', 'value', d['value'], 'single', d['ms_single_pair'], 'top', s['refine_sweep_top'], 'low', s['refine_sweep'], 'skew alone', r['alone']['avg_launch_ms'], 'in flight', r['avg_launch_ms'], 'traffic', r.get('traffic'), 'valu', json.dumps(r.get('valu')), 'scalar', json.dumps(r.get('scalar')))
except Exception as e:
    print('[$n] FAILED', e); print(open('/tmp/err_$n.log').read()[-1500:])"
done; done
cp /tmp/keep.so reconstruction_amd/librsm_mi355.so
