cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_golden.py tests/test_gpu_ncc_ties.py tests/test_gpu_parity.py tests/test_gpu_edgecases.py -m gpu -x -q 2>&1 | tail -8
python bench.py --config c5 --steps 4 --no-cpu-baseline > gpurun_out/r3_bench_c5.log 2>&1; tail -c 1800 gpurun_out/r3_bench_c5.log
python bench.py --config c5 --steps 4 --no-cpu-baseline --opt wide_rows=2 > gpurun_out/r3_bench_c5_mfma.log 2>&1; tail -c 900 gpurun_out/r3_bench_c5_mfma.log
