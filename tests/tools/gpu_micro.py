"""NCC kernel microbenchmarks on the GPU box (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from reconstruction_amd import Context
ctx = Context(0)
for (W, H, r) in [(4096, 3072, 5), (2048, 1536, 5)]:
    for cands in (1, 5, 10, 25, 26, 129):
        ms = ctx.bench_ncc(W, H, r, cands, iters=3)
        px = (W - 2 * r) * (H - 2 * r)
        print("W=%d H=%d r=%d cands=%3d: %8.3f ms/launch  %8.1f GDE/s" % (W, H, r, cands, ms, px * cands / ms / 1e6), flush=True)
