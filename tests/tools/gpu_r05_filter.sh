#!/bin/bash
# round 5: the pixel-window cloud filter -- its tests, then filter_ms on C2's cloud with the pass on / off and the adapter figures
mkdir -p gpurun_out
export OMP_NUM_THREADS=16 OMP_WAIT_POLICY=passive
timeout 900 python -u -m pytest tests/test_gpu_cloud_filter.py tests/test_gpu_cpp_adapter.py -m gpu -q -x -s -p no:cacheprovider 2>&1 | grep -v "^$" | tail -25
python -u - <<'PY'
import time, numpy as np, torch, sys
sys.path.insert(0, '.')
from reconstruction_amd import Context, synth
cfg = synth.config_c2(pair=0)
with Context(0) as ctx:
    ctx.match_pair(cfg, want_cloud=False)
    n = ctx.n_points
    rec = torch.empty((n, 16), dtype=torch.uint8, device="cuda:0"); nrm = torch.empty((n, 4), dtype=torch.float32, device="cuda:0")
    out = {}
    for w in (1, 0, 1):
        ctx.set_option("filter_window", w)
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            m, st = ctx.filter_last_cloud(rec.data_ptr(), nrm.data_ptr(), n, 100, 1.0, 2.5, (0.0, 0.0, 0.0))
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        info = ctx.filter_last_info()
        print("C2 cloud %d points: window %d -> %.2f ms, kept %d, undecided %d (%.2f %%), stats %s" % (n, w, dt * 1e3, m, info["undecided"], 100.0 * info["undecided"] / n, st))
        out[w] = (m, st["mean"], st["stddev"], st["threshold"], rec[:m].cpu().numpy().tobytes(), nrm[:m].cpu().numpy().tobytes())
    print("identical results with and without the window pass:", out[1] == out[0])
PY
