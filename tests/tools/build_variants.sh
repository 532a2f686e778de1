#!/bin/bash
# builds compile-time variants of the library HERE (hipcc cross-compiles) into reconstruction_amd/variants/v_<i>.so so that a gpurun call
# spends its minutes measuring, not compiling:  bash tests/tools/build_variants.sh <file to touch> "EXTRA_1" "EXTRA_2" ...
f=$1; shift
mkdir -p reconstruction_amd/variants; rm -f reconstruction_amd/variants/v_*.so
i=0
for e in "$@"; do
  i=$((i+1)); touch reconstruction_amd/csrc/$f
  make -s -C reconstruction_amd/csrc EXTRA="$e" all 2>&1 | grep -E "error" ; cp reconstruction_amd/librsm_mi355.so reconstruction_amd/variants/v_$i.so; echo "$e" > reconstruction_amd/variants/v_$i.txt
done
touch reconstruction_amd/csrc/$f; make -s -C reconstruction_amd/csrc all
