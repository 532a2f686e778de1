#!/bin/bash
# round 5: xi goldens on the device, the adapter's new forms (16-byte records, filter on the GPU, device list), bench with the adapter figures
mkdir -p gpurun_out
export OMP_NUM_THREADS=16 OMP_WAIT_POLICY=passive
timeout 900 python -u -m pytest tests/test_gpu_golden.py tests/test_gpu_cpp_adapter.py tests/test_gpu_cloud_filter.py tests/test_gpu_multigpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -8
python -u bench.py --no-cpu-baseline --measure-traffic 0 --steps 8 --warmup 2 2>gpurun_out/r05b_err.log | tee gpurun_out/r05b_bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value', d['value'], 'single', d['ms_single_pair'], 'filter_ms', d.get('filter_ms'))
for k in ('adapter','adapter_with_filter','adapter_fp64_points'):
    a=d.get(k,{}); print(k, {kk:a.get(kk) for kk in ('value','ms_per_pair','pairs','points','kept','error')})"
tail -3 gpurun_out/r05b_err.log
