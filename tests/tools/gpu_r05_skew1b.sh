#!/bin/bash
# round 5: k_refine_skew1 iteration: parity subset, A/B against the shipped variant, then the per-phase clock split (timing build)
python -m pytest tests/test_gpu_parity.py tests/test_gpu_edgecases.py -m gpu -x -q -k "time_skewed or whole_range" 2>&1 | tail -3
bash tests/tools/gpu_ab_opts.sh 2 "" "refine_skew_variant=64" 2>&1
if [ -f reconstruction_amd/variants/v_1.so ]; then
  cp reconstruction_amd/librsm_mi355.so /tmp/shipped.so; cp reconstruction_amd/variants/v_1.so reconstruction_amd/librsm_mi355.so
  python -u bench.py --no-cpu-baseline --measure-traffic 0 --adapter-pairs 0 --steps 1 --warmup 0 --inflight 1 --opt refine_skew_variant=64 2>&1 | grep skew1time | head -8
  cp /tmp/shipped.so reconstruction_amd/librsm_mi355.so
fi
