#!/bin/bash
# what 24-byte cache entries would cost the time-skewed kernel in residency alone: LDS padded by 9 216 B
BENCH_OPTS="" bash tests/tools/gpu_ab_build2.sh 2 "" "-DRF_SKEW_PAD=9216" "-DRF_SKEW_PAD=1024"
touch reconstruction_amd/csrc/k_refine.hip; make -s -C reconstruction_amd/csrc all
