cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r3_full_pytest.log 2>&1; tail -5 gpurun_out/r3_full_pytest.log
bash tests/tools/gpu_profile_r03.sh 2>&1 | tail -60
