cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_cloud_filter.py tests/test_gpu_rectify.py -m gpu -x -q 2>&1 | tail -8
python tests/tools/gpu_filter_c2.py 3 > gpurun_out/r3_filter1.log 2>&1
tail -3 gpurun_out/r3_filter1.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_filter1 -o filt -- python tests/tools/gpu_filter_c2.py 1 > gpurun_out/r3_filter1_prof.log 2>&1
find gpurun_out/prof_filter1 -name "*kernel_stats*" | head -3
