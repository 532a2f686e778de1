"""Development aid: rsm_filter_last_cloud on the C2 cloud, timed; run under rocprofv3 --kernel-trace --stats
for the per-kernel split (profiles/r03_filter_*.csv)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from reconstruction_amd import Context, synth
cfg = synth.config_c2(pair=0)
ctx = Context(0)
ctx.upload_pair(cfg)
ctx.run_pair()
n = ctx.n_points
rec = torch.empty((n, 16), dtype=torch.uint8, device="cuda:0")
nrm = torch.empty((n, 4), dtype=torch.float32, device="cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for i in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m, st = ctx.filter_last_cloud(rec.data_ptr(), nrm.data_ptr(), n, 100, 1.0, 2.5, (0.0, 0.0, 0.0))
    torch.cuda.synchronize()
    print("filter_last_cloud: %d -> %d points, %.1f ms, stats %s" % (n, m, (time.perf_counter() - t0) * 1e3, st), flush=True)
