#!/bin/bash
# A/B of the C++ adapter's loop on C2 (18 pairs): page-locked input staging on / off, 5 / 6 / 7 slots
export OMP_NUM_THREADS=16
python -u - <<'PY'
import os, sys, json
sys.path.insert(0, os.getcwd())
import bench
from reconstruction_amd import synth, Context
cfgs = [synth.config_c2(pair=p) for p in range(3)]
with Context(0) as c:
    v_top = c.match_pair(cfgs[0], want_cloud=False).v_top
for extra in ("0",):
    for slots in (3, 4, 5, 6):
        os.environ["RSM_ADAPTER_FLAGS_EXTRA"] = extra
        r = bench.adapter_bench(cfgs, 18, slots, v_top)
        for k in ("records16", "gpu_filter"):
            d = r[k]
            print("flags+%s slots %d %-10s value %7.2f ms/pair %6.2f wait %.3f replay %.3f" % (extra, slots, k, d["value"], d["ms_per_pair"], d["caller_wait_s"], d["caller_replay_s"]))
PY
