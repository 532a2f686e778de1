#!/bin/bash
# round 5: the one-wave-per-strip time-skewed kernel (refine_skew_variant 64): parity, then same-box A/B against the shipped variant
python -m pytest tests/test_gpu_parity.py tests/test_gpu_edgecases.py -m gpu -x -q -k "time_skewed or whole_range" 2>&1 | tail -5
bash tests/tools/gpu_ab_opts.sh 2 "" "refine_skew_variant=64" "refine_skew_variant=64 refine_skew1_strips=4096" 2>&1
