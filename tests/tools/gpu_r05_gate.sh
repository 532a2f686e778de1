#!/bin/bash
# adapter loop with the run gate: slots x max_running (prebuilt bench harness through bench.py's adapter leg)
for cfg in "5 3" "5 0" "6 3" "7 3" "4 3" "5 3"; do
  set -- $cfg
  RSM_ADAPTER_MAX_RUNNING=$2 python -u bench.py --no-cpu-baseline --measure-traffic 0 --steps 3 --warmup 1 --adapter-inflight $1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('slots $1 max_running $2: adapter', d.get('value_adapter_pcie_inclusive'), 'with filter', d.get('value_with_filter'), 'resident', d['value'])"
done
