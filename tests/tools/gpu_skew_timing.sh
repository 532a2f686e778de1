#!/bin/bash
# per-phase shader-clock split of k_refine_skew's step (RF_SKEW_TIMING build; printf from a few workgroups of launch 12)
mkdir -p gpurun_out
touch reconstruction_amd/csrc/k_refine.hip
make -s -C reconstruction_amd/csrc EXTRA=-DRF_SKEW_TIMING all
python -u bench.py --no-cpu-baseline --measure-traffic 0 --steps 1 --warmup 0 --inflight 1 "$@" 2>&1 | grep skewtime | sort | uniq | head -80 > gpurun_out/skew_timing.log
awk '{i+=$9; l+=$11; m+=$13; s+=$15; b+=$17; n++} END {print "mean over", n, "waves: issue", i/n, "lds", l/n, "math", m/n, "stage", s/n, "barrier", b/n}' gpurun_out/skew_timing.log
for w in 0 1 2 3; do grep "wave $w " gpurun_out/skew_timing.log | awk -v w=$w '{i+=$9; l+=$11; m+=$13; s+=$15; b+=$17; n++} END {print "wave", w, "n", n, ": issue", i/n, "lds", l/n, "math", m/n, "stage", s/n, "barrier", b/n}'; done
head -8 gpurun_out/skew_timing.log
