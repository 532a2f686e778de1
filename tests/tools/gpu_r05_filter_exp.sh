#!/bin/bash
# timing experiments on k_sor_window (compile-time, results invalid): bash tests/tools/gpu_r05_filter_exp.sh
for e in 0 1 2 3; do
  touch reconstruction_amd/csrc/k_filter.hip
  make -s -C reconstruction_amd/csrc EXTRA="-DWIN_EXP=$e" all 2>/dev/null || { echo build failed; continue; }
  echo "WIN_EXP=$e"
  bash tests/tools/gpu_r05_filter_prof.sh 16 2>&1 | grep "k_sor_window<16, false>"
done
touch reconstruction_amd/csrc/k_filter.hip; make -s -C reconstruction_amd/csrc all
