#!/bin/bash
mkdir -p gpurun_out
export OMP_NUM_THREADS=16 OMP_WAIT_POLICY=passive
for c in c3 c5; do
  timeout 600 python -u bench.py --no-cpu-baseline --config $c --steps 5 --warmup 1 > gpurun_out/r02_bench_$c.log 2>gpurun_out/r02_bench_$c.err; echo "$c rc=$?"; tail -1 gpurun_out/r02_bench_$c.log | cut -c1-700
done
# the 10-pair rig on one GPU through the C ABI's work queue (rsm_match_pairs, host buffers in and out: PCIe included)
python tests/tools/gpu_c3_queue.py 2>&1 | grep '^C3'
