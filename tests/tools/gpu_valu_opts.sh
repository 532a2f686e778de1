#!/bin/bash
# VALU / SALU wave-instructions, VALU-busy share and the wave-cycle split of the dominant kernel for option sets (bench.py's own --pmc passes)
for o in "$@"; do
  args=""; for kv in $o; do args="$args --opt $kv"; done
  python -u bench.py --no-cpu-baseline --adapter-pairs 0 --steps 4 --warmup 1 $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; v=r.get('valu') or {}
print('[$o]', 'value', d['value'], 'single', d['ms_single_pair'], 'skew alone', r['alone']['avg_launch_ms'], 'in flight', r['avg_launch_ms'], 'valu', v.get('insts_valu'), 'salu', v.get('insts_salu'), 'active', v.get('active_frac'), v.get('wave_cycles_split'), 'pmc-pass ms', v.get('launch_ms_in_this_pass'))"
done
