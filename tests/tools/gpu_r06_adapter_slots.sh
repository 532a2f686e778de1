#!/bin/bash
# the adapter loop with the cloud filter inside, for several numbers of pairs in flight: PAIRS=36 bash tests/tools/gpu_r06_adapter_slots.sh 5 7 9
# (environment: RSM_FILTER_LOW_PRIORITY=0/1 -- the filter's stream priority)
for n in "$@"; do
  python -u bench.py --no-cpu-baseline --measure-traffic 0 --steps 3 --warmup 1 --adapter-inflight $n --adapter-pairs ${PAIRS:-18} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); a=d['adapter']; f=d['adapter_with_filter']
print('slots $n: adapter', a['value'], 'steady', a['value_steady'], '| with filter', f['value'], 'steady', f['value_steady'], 'filter_ms', d['filter_ms'])"
done
