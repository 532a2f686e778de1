#!/bin/bash
# like build_variants.sh, the extra flags for ONE translation unit only: bash tests/tools/build_variants_tu.sh k_refine "FLAGS_1" "FLAGS_2" ...
tu=$1; shift
mkdir -p reconstruction_amd/variants; rm -f reconstruction_amd/variants/v_*
i=0
for e in "$@"; do
  i=$((i+1)); touch reconstruction_amd/csrc/$tu.hip
  make -s -C reconstruction_amd/csrc "EXTRA_$tu=$e" all 2>&1 | grep -E "error"; cp reconstruction_amd/librsm_mi355.so reconstruction_amd/variants/v_$i.so; echo "$tu: $e" > reconstruction_amd/variants/v_$i.txt
done
touch reconstruction_amd/csrc/$tu.hip; make -s -C reconstruction_amd/csrc all
