cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ncc_ties.py tests/test_gpu_golden.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
python tests/tools/gpu_ncc_micro.py 2>&1 | tee gpurun_out/r3_ncc_micro.log
python bench.py --config c5 --steps 4 --no-cpu-baseline > gpurun_out/r3_bench_c5.log 2>&1; grep -o '"initial_match": [0-9.]*' gpurun_out/r3_bench_c5.log; grep -o '"value": [0-9.]*' gpurun_out/r3_bench_c5.log
