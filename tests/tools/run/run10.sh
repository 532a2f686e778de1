cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r03_bench.log 2>gpurun_out/r03_bench.err; grep '^{' gpurun_out/r03_bench.log | cut -c1-600
python bench.py --config c3 --no-cpu-baseline > gpurun_out/r03_bench_c3.log 2>&1; grep '^{' gpurun_out/r03_bench_c3.log | cut -c1-200
python bench.py --config c5 --no-cpu-baseline > gpurun_out/r03_bench_c5.log 2>&1; grep '^{' gpurun_out/r03_bench_c5.log | cut -c1-200
python -m pytest tests/test_gpu_configs.py -m gpu -q -s -k "ten_fullsize" 2>&1 | grep -E "C3 rig|passed|failed"
