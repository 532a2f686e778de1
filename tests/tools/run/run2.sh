cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python tests/tools/gpu_filter_c2.py 3 > gpurun_out/r3_filter0.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_filter0 -o filt -- python tests/tools/gpu_filter_c2.py 1 > gpurun_out/r3_filter0_prof.log 2>&1
tail -3 gpurun_out/r3_filter0.log
find gpurun_out/prof_filter0 -name "*kernel_stats*" | head
python bench.py --steps 6 > gpurun_out/r3_bench1.log 2>&1; tail -c 1500 gpurun_out/r3_bench1.log
python -m pytest tests/test_gpu_configs.py tests/test_gpu_rectify.py -m gpu -x -q -s 2>&1 | tail -8
