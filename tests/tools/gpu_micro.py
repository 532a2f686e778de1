"""NCC kernel microbenchmarks on the GPU box (MDE/s = pixel x candidate evaluations per second of the
interval-argmax kernel alone, at 11x11 / 129 candidates and 15x15 / 257 candidates)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from reconstruction_amd import Context
ctx = Context(0)
for (W, H, r, cl) in [(4096, 3072, 5, (1, 5, 25, 129)), (4096, 3072, 7, (5, 129, 257))]:
    for cands in cl:
        for opt in (0, 1):
            if cands <= 160 and opt:
                continue   # the row GEMM only takes intervals wider than 160
            ctx.set_option("no_rowgemm", opt)
            ms = ctx.bench_ncc(W, H, r, cands, iters=3)
            px = (W - 2 * r) * (H - 2 * r)
            print("W=%d H=%d window %dx%d cands=%3d %s: %8.3f ms/launch  %8.1f GDE/s" % (W, H, 2 * r + 1, 2 * r + 1, cands,
                  "(k_ncc_wide)   " if opt else ("(k_ncc_rowgemm)" if cands > 160 else "(k_ncc_dot4)   "), ms, px * cands / ms / 1e6), flush=True)
ctx.set_option("no_rowgemm", 0)
