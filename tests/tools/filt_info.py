import time, torch, sys
sys.path.insert(0, '/root/repo')
from reconstruction_amd import Context, synth
cfg = synth.config_c2(pair=0)
with Context(0) as ctx:
    ctx.match_pair(cfg, want_cloud=False)
    n = ctx.n_points
    rec = torch.empty((n, 16), dtype=torch.uint8, device="cuda:0"); nrm = torch.empty((n, 4), dtype=torch.float32, device="cuda:0")
    for fl in (23, 23):
        ctx.set_option("filter_list", fl)
        for rep in range(4):
            torch.cuda.synchronize(); t0 = time.time()
            m, st = ctx.filter_last_cloud(rec.data_ptr(), nrm.data_ptr(), n, 100, 1.0, 2.5, (0.0, 0.0, 0.0))
            torch.cuda.synchronize(); dt = time.time() - t0
            print("rep", rep, "ms %.2f" % ((time.time() - t0) * 1e3), ctx.filter_last_info()["radius"])
        print("filter_list", fl, "ms %.2f" % (dt * 1e3), ctx.filter_last_info(), st)
