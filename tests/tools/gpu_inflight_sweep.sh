#!/bin/bash
# value against pairs in flight and hardware queues
for q in 4 8 16; do for f in 3 4 5; do
  GPU_MAX_HW_QUEUES=$q python -u bench.py --no-cpu-baseline --measure-traffic 0 --adapter-pairs 0 --steps 6 --warmup 2 --inflight $f "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('queues $q inflight $f value', d['value'], 'ms/pair', d['ms_per_pair'])"
done; done
