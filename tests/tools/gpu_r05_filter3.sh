#!/bin/bash
# which list passes pay on C2's cloud: filter_list 0 (none) / 1 (24 px) / 2 (40 px) / 3 (both), probed tile radius
export OMP_NUM_THREADS=16
python -u - <<'PY'
import time, torch, sys
sys.path.insert(0, '.')
from reconstruction_amd import Context, synth
cfg = synth.config_c2(pair=0)
with Context(0) as ctx:
    ctx.match_pair(cfg, want_cloud=False)
    n = ctx.n_points
    rec = torch.empty((n, 16), dtype=torch.uint8, device="cuda:0"); nrm = torch.empty((n, 4), dtype=torch.float32, device="cuda:0")
    for w in (1, 16, 20):
        for fl in (0, 1, 2, 3):
            ctx.set_option("filter_window", w); ctx.set_option("filter_list", fl)
            for rep in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                m, st = ctx.filter_last_cloud(rec.data_ptr(), nrm.data_ptr(), n, 100, 1.0, 2.5, (0.0, 0.0, 0.0))
                torch.cuda.synchronize(); dt = time.perf_counter() - t0
            info = ctx.filter_last_info()
            print("window %2d list %d: %.2f ms, kept %d, left to the ladder %d" % (w, fl, dt * 1e3, m, info["undecided"]))
PY
