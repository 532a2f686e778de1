#!/bin/bash
# round 5: the glibc-grade exp -- its parity tests, then a same-box A/B against round 4's library (tests/_ab/r04.so)
mkdir -p gpurun_out
export OMP_NUM_THREADS=16 OMP_WAIT_POLICY=passive
timeout 900 python -u -m pytest tests/test_gpu_golden.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -8
bash tests/tools/gpu_ab_so.sh 2 r04 expg
