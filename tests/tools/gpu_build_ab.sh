#!/bin/bash
# timing of builds with extra compiler flags (timing experiments: results may be invalid): gpu_build_ab.sh reps "flags" ...
reps=$1; shift
for i in $(seq $reps); do for v in "$@"; do
  touch reconstruction_amd/csrc/k_refine.hip
  make -s -C reconstruction_amd/csrc EXTRA="$v" all 2>&1 | grep -E "error" | head -3
  python -u bench.py --no-cpu-baseline --measure-traffic 0 --adapter-pairs 0 --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('[$v]', 'value', d['value'], 'single', d['ms_single_pair'], 'skew alone', r['alone']['avg_launch_ms'], 'in flight', r['avg_launch_ms'])"
done; done
