export RSM_AB_OLD_LIBRARY=1
run() { python -u bench.py --no-cpu-baseline --measure-traffic 0 --adapter-pairs 0 --steps 8 --warmup 2 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('value', d['value'], 'ms/pair', d['ms_per_pair'], 'single', d['ms_single_pair'], 'skew alone', r['alone']['avg_launch_ms'], 'in flight', r['avg_launch_ms'])"; }
for o in "" "--opt heavy_exclusive=0" "--opt heavy_exclusive=2" "--opt heavy_lanes=1" "--inflight 2" "--inflight 4" "--opt refine_skew_from=18" "--opt refine_skew_from=26" "--opt refine_skew_min_px=200000" "--opt refine_skew_waves=3200" "--opt refine_skew_waves=2048"; do echo "[$o]"; run $o; done
