#!/bin/bash
# round 5: window radius sweep of the pixel-window cloud filter on C2's cloud (decided fraction, time)
export OMP_NUM_THREADS=16
python -u - <<'PY'
import time, numpy as np, torch, sys
sys.path.insert(0, '.')
from reconstruction_amd import Context, synth
cfg = synth.config_c2(pair=0)
with Context(0) as ctx:
    ctx.match_pair(cfg, want_cloud=False)
    n = ctx.n_points
    rec = torch.empty((n, 16), dtype=torch.uint8, device="cuda:0"); nrm = torch.empty((n, 4), dtype=torch.float32, device="cuda:0")
    ref = None
    for w in (0, 1, 12, 16):
        ctx.set_option("filter_window", w)
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            m, st = ctx.filter_last_cloud(rec.data_ptr(), nrm.data_ptr(), n, 100, 1.0, 2.5, (0.0, 0.0, 0.0))
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        info = ctx.filter_last_info()
        sig = (m, st["mean"], st["stddev"], st["threshold"], rec[:m].cpu().numpy().tobytes(), nrm[:m].cpu().numpy().tobytes())
        if ref is None: ref = sig
        print("window %2d: %.2f ms, kept %d, undecided %d (%.2f %%), same as generic: %s" % (w, dt * 1e3, m, info["undecided"], 100.0 * info["undecided"] / n, sig == ref))
PY
