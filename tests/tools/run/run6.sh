cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ncc -o ncc -- python tests/tools/gpu_ncc_micro.py > gpurun_out/r3_ncc_micro_prof.log 2>&1
ls gpurun_out/prof_ncc | head
