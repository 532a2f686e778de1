#!/bin/bash
# same-box A/B of prebuilt library variants on the cloud filter (C2's cloud): time and the tile kernel
cp reconstruction_amd/librsm_mi355.so /tmp/shipped.so
for r in 1 2; do
for v in reconstruction_amd/variants/v_*.so; do
  cp $v reconstruction_amd/librsm_mi355.so; echo "== [$(cat ${v%.so}.txt)]"
  python tests/tools/filt_info.py 2>&1 | grep -E "^rep 3|filter_list" | cut -c1-130 | tail -2
done
done
cp /tmp/shipped.so reconstruction_amd/librsm_mi355.so
