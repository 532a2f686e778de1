#!/bin/bash
# timing builds of k_refine_skew1 (prebuilt variants): the per-phase split and the launch time
cp reconstruction_amd/librsm_mi355.so /tmp/shipped.so
for v in reconstruction_amd/variants/v_*.so; do
  cp $v reconstruction_amd/librsm_mi355.so; echo "== $(cat ${v%.so}.txt)"
  python -u bench.py --no-cpu-baseline --measure-traffic 0 --adapter-pairs 0 --steps 2 --warmup 1 --inflight 1 --opt refine_skew_variant=64 2>&1 | python -c "
import sys,json
n=0
for l in sys.stdin:
    if 'skew1time' in l:
        n+=1
        if n<=3: print(l.strip())
    elif l.startswith('{'):
        d=json.loads(l); print('single', d['ms_single_pair'], 'skew alone', d['roofline']['alone']['avg_launch_ms'] if 'alone' in d['roofline'] else d['roofline']['avg_launch_ms'])"
done
cp /tmp/shipped.so reconstruction_amd/librsm_mi355.so
