#!/bin/bash
mkdir -p gpurun_out
i=0
for o in "$@"; do
  i=$((i+1))
  timeout 600 python -u bench.py --steps 3 --warmup 1 --no-cpu-baseline $o > gpurun_out/bench_ab$i.log 2>&1
  echo "== $o"; tail -1 gpurun_out/bench_ab$i.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], {k:v for k,v in d['stage_ms_per_step'].items() if 'refine' in k})" || tail -5 gpurun_out/bench_ab$i.log
done
