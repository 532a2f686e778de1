"""Development aid: the cloud filter on a full-size C2 cloud (time, survivors, exhaustive searches)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from reconstruction_amd import Context, synth
cfg = synth.config_c2(pair=0)
ctx = Context(0)
ctx.upload_pair(cfg); ctx.run_pair()
n = ctx.n_points
rec = torch.empty((n, 16), dtype=torch.uint8, device="cuda:0")
nd = torch.empty((n, 4), dtype=torch.float32, device="cuda:0")
for radius in (2.5, 0.5):
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m, st = ctx.filter_last_cloud(rec.data_ptr(), nd.data_ptr(), n, 100, 1.0, radius, (0, 0, 0))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("radius %.1f: n %d kept %d (%.1f %%)  %.1f ms  stats %s" % (radius, n, m, 100.0 * m / n, dt * 1e3, st), flush=True)
