"""One C2 pair matched, then the per-pair cloud filter REPS times (for rocprofv3 passes: python tests/tools/gpu_filter_run.py [reps])."""
import sys, time
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from reconstruction_amd import Context, synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = synth.config_c2(pair=0)
with Context(0) as ctx:
    for a in sys.argv[2:]:
        k, v = a.split("=")
        ctx.set_option(k, int(v))
    ctx.match_pair(cfg, want_cloud=False)
    n = ctx.n_points
    rec = torch.empty((n, 16), dtype=torch.uint8, device="cuda:0"); nrm = torch.empty((n, 4), dtype=torch.float32, device="cuda:0")
    for rep in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m, st = ctx.filter_last_cloud(rec.data_ptr(), nrm.data_ptr(), n, 100, 1.0, 2.5, (0.0, 0.0, 0.0))
        torch.cuda.synchronize(); print("filter %.2f ms, kept %d" % ((time.perf_counter() - t0) * 1e3, m))
