#!/bin/bash
# end-of-round evidence: whole GPU suite, the default bench, C3 / C5, the profiles
mkdir -p gpurun_out
export OMP_NUM_THREADS=16 OMP_WAIT_POLICY=passive
timeout 1500 python -u -m pytest tests -m gpu -x -q --durations=5 -p no:cacheprovider > gpurun_out/pytest_full.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_full.log
tail -4 gpurun_out/pytest_full.log
bash tests/tools/gpu_bench_full.sh | cut -c1-400
for c in c3 c5; do
  timeout 600 python -u bench.py --no-cpu-baseline --config $c --steps 5 --warmup 1 > gpurun_out/bench_$c.log 2>gpurun_out/bench_$c.err; echo "$c rc=$?"; tail -1 gpurun_out/bench_$c.log | cut -c1-200
done
python tests/tools/gpu_ncc_micro.py > gpurun_out/ncc_micro.log 2>&1; tail -11 gpurun_out/ncc_micro.log
bash tests/tools/gpu_profile_r03b.sh > gpurun_out/prof_r03b.log 2>&1
tail -5 gpurun_out/prof_r03b.log
