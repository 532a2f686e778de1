#!/bin/bash
# A/B of tuning options on the GPU box: parity tests, then bench.py once per option set given as arguments.
mkdir -p gpurun_out
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-16} OMP_WAIT_POLICY=passive
timeout 600 python -u -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest.log
grep -E "passed|failed|rc=" gpurun_out/pytest.log | tail -3
i=0
for o in "$@"; do
  i=$((i+1))
  timeout 600 python -u bench.py --steps 5 --warmup 1 --no-cpu-baseline $o > gpurun_out/bench_ab$i.log 2>&1
  echo "== $o"; tail -1 gpurun_out/bench_ab$i.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], {k:v for k,v in d['stage_ms_per_step'].items() if 'refine' in k})" || tail -5 gpurun_out/bench_ab$i.log
done
