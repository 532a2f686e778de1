#!/bin/bash
mkdir -p gpurun_out
export OMP_NUM_THREADS=16 OMP_WAIT_POLICY=passive
for c in c3 c5; do
  timeout 600 python -u bench.py --no-cpu-baseline --config $c --steps 5 --warmup 1 > gpurun_out/r02_bench_$c.log 2>gpurun_out/r02_bench_$c.err; echo "$c rc=$?"; tail -1 gpurun_out/r02_bench_$c.log | cut -c1-700
done
# the 10-pair rig on one GPU through the C ABI's work queue (rsm_match_pairs, host buffers in and out: PCIe included)
python - <<'PY'
import time, numpy as np
from reconstruction_amd import Context, match_pairs, synth
cfgs = [synth.config_c3(pair=p) for p in range(10)]
for nctx in (1, 2, 3):
    pool = [Context(0) for _ in range(nctx)]
    match_pairs(pool, cfgs[:nctx], want_cloud=True)          # warm-up (workspace allocation)
    t0 = time.perf_counter()
    res, st = match_pairs(pool, cfgs, want_cloud=True)
    dt = time.perf_counter() - t0
    v = sum(r.v_top for r in res)
    print("C3 10 pairs, %d context(s), host buffers in/out: %.1f ms per pair, %.1f Mdisp/s (PCIe-inclusive), statuses %s" % (nctx, dt / 10 * 1e3, v / dt / 1e6, set(st)), flush=True)
    for c in pool: c.close()
PY
